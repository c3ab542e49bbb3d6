// overlay.h -- headless version of the reference's per-splat debug drawing (main.cpp:441-485): the two
// principal axes, the 16-segment 1-sigma ellipse in the splat's colour, and the exact 1-sigma bounding box from
// the covariance (Form.pdf section 12), rasterised into an RGB8 image instead of pr::PrimVertex lines.
// Host-side diagnostic; not on the training path.
#pragma once

#include <cmath>
#include <cstdint>
#include <vector>

#include "image_io.h"
#include "splat2d.h"

namespace s2dio {

inline void draw_line(Image8* im, float x0, float y0, float x1, float y1, const uint8_t c[3])
{
    const float dx = x1 - x0, dy = y1 - y0;
    const int steps = (int)std::ceil(std::fmax(std::fabs(dx), std::fabs(dy))) + 1;
    if (steps > 1 << 16) return; // degenerate / huge: skip rather than loop
    for (int i = 0; i <= steps; i++) {
        const float t = (float)i / (float)steps;
        const int x = (int)std::floor(x0 + dx * t), y = (int)std::floor(y0 + dy * t);
        if (x < 0 || y < 0 || x >= im->w || y >= im->h) continue;
        uint8_t* p = &im->rgb[((size_t)y * im->w + x) * 3];
        p[0] = c[0]; p[1] = c[1]; p[2] = c[2];
    }
}

// scale: output pixels per image pixel (the GUI's viewScale, main.cpp:822); `im` must already be that size.
inline void draw_splat_overlay(Image8* im, const std::vector<s2d_splat>& splats, int scale, int stride = 1)
{
    const float S = (float)scale;
    for (size_t i = 0; i < splats.size(); i += (size_t)(stride < 1 ? 1 : stride)) {
        const s2d_splat& s = splats[i];
        // cov_of, main.cpp:206-221
        const float c = std::cos(s.rot), sn = std::sin(s.rot);
        const float l0 = s.sx * s.sx, l1 = s.sy * s.sy;
        const float s11 = l0 * c * c + l1 * sn * sn, s12 = (l0 - l1) * sn * c, s22 = l0 + l1 - s11;
        // eignValues, main.cpp:188-196
        const float mean = (s11 + s22) * 0.5f, det = s11 * s22 - s12 * s12;
        const float d = std::sqrt(std::fmax(mean * mean - det, 0.0f));
        const float lambda0 = mean + d, lambda1 = mean - d;
        // eigen_vectors_of_cov, main.cpp:223-234
        const float eps = 1e-15f;
        float ex = s11 < s22 ? s12 + eps : lambda0 - s22, ey = s11 < s22 ? lambda0 - s11 : s12 + eps;
        const float len = std::sqrt(ex * ex + ey * ey);
        if (!(len > 0.0f) || !std::isfinite(len)) continue;
        ex /= len; ey /= len;
        const float a0x = ex * std::sqrt(lambda0), a0y = ey * std::sqrt(lambda0);             // axis0, main.cpp:443
        const float a1x = -ey * std::sqrt(std::fmax(lambda1, 0.0f)), a1y = ex * std::sqrt(std::fmax(lambda1, 0.0f)); // axis1, :444
        const float px = s.pos[0] * S, py = s.pos[1] * S;
        const uint8_t white[3] = {255, 255, 255}, light[3] = {230, 230, 230}, grey[3] = {128, 128, 128};
        draw_line(im, px, py, px + a0x * S, py + a0y * S, white);                             // main.cpp:447-448
        draw_line(im, px, py, px + a1x * S, py + a1y * S, light);                             // main.cpp:450-451
        const uint8_t col[3] = {(uint8_t)(s.color[0] * 255.0f), (uint8_t)(s.color[1] * 255.0f), (uint8_t)(s.color[2] * 255.0f)};
        const int nvtx = 16;                                                                  // main.cpp:454-462
        for (int k = 0; k < nvtx; k++) {
            const float t0 = 6.2831853f * k / nvtx, t1 = 6.2831853f * (k + 1) / nvtx;
            draw_line(im, px + (a0x * std::sin(t0) + a1x * std::cos(t0)) * S, py + (a0y * std::sin(t0) + a1y * std::cos(t0)) * S,
                      px + (a0x * std::sin(t1) + a1x * std::cos(t1)) * S, py + (a0y * std::sin(t1) + a1y * std::cos(t1)) * S, col);
        }
        // exact bounding box from the covariance: sqrt(inv_cov[1][1]*det) = sqrt(s11), sqrt(inv_cov[0][0]*det) = sqrt(s22); main.cpp:464-477
        const float hx = std::sqrt(std::fmax(s11, 0.0f)) * S, hy = std::sqrt(std::fmax(s22, 0.0f)) * S;
        draw_line(im, px - hx, py - hy, px + hx, py - hy, grey);
        draw_line(im, px + hx, py - hy, px + hx, py + hy, grey);
        draw_line(im, px + hx, py + hy, px - hx, py + hy, grey);
        draw_line(im, px - hx, py + hy, px - hx, py - hy, grey);
    }
}

inline Image8 upscale(const Image8& in, int scale)
{
    Image8 out;
    out.w = in.w * scale;
    out.h = in.h * scale;
    out.rgb.resize((size_t)out.w * out.h * 3);
    for (int y = 0; y < out.h; y++)
        for (int x = 0; x < out.w; x++)
            std::memcpy(&out.rgb[((size_t)y * out.w + x) * 3], &in.rgb[((size_t)(y / scale) * in.w + x / scale) * 3], 3);
    return out;
}

} // namespace s2dio
