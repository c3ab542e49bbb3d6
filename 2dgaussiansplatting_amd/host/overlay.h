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

// One vertex as the reference hands it to pr::PrimVertex(glm::vec3, glm::u8vec3): scene coordinates (x, -y, 0) -- the
// reference's camera looks at a y-up plane, main.cpp:447 -- and an 8-bit colour.
struct OverlayVertex {
    float x, y, z;
    uint8_t r, g, b, pad;
};
constexpr int kOverlayVertices = 46; // per splat: 2 axes + 17 ellipse segments + 4 box sides, two vertices each

// The vertex list of main.cpp:419-477 for one splat, in the reference's call order and with its arithmetic: cov_of
// (:206-221), eignValues (:188-196), the inverse (:432-436), eigen_vectors_of_cov (:223-234; glm::normalize(v) =
// v * (1 / sqrt(dot(v, v)))), the axes (:443-451), the 16-gon (:454-462; pr::CircleGenerator = the angle-addition
// recurrence from (sin, cos) = (0, 1)), the box (:464-477).  tests/test_shared_math_host.py compares these bit for bit with
// the test oracle's restatement of the same lines.
inline void overlay_vertices(const s2d_splat& s, OverlayVertex* out)
{
    struct V3 {
        float x, y, z;
        V3 operator+(const V3& o) const { return {x + o.x, y + o.y, z + o.z}; }
        V3 operator*(float k) const { return {x * k, y * k, z * k}; }
    };
    int n = 0;
    const auto put = [&](const V3& p, unsigned r, unsigned g, unsigned b) {
        out[n++] = OverlayVertex{p.x, p.y, p.z, (uint8_t)r, (uint8_t)g, (uint8_t)b, 0};
    };
    const float ct = std::cos(s.rot), st = std::sin(s.rot);
    const float lam_x = s.sx * s.sx, lam_y = s.sy * s.sy;
    const float s11 = lam_x * ct * ct + lam_y * st * st;
    const float s12 = (lam_x - lam_y) * st * ct;
    const float s22 = lam_x + lam_y - s11;
    const float mean = (s11 + s22) * 0.5f;
    const float det = s11 * s22 - s12 * s12;
    const float disc = mean * mean - det;
    const float half_gap = std::sqrt(disc < 0.0f ? 0.0f : disc); // ss_max(x, 0) = (x < 0) ? 0 : x
    const float big = mean + half_gap, small = mean - half_gap;
    const float r_big = std::sqrt(big), r_small = std::sqrt(small);
    const float i00 = s22 / det, i11 = s11 / det;
    const float eps = 1e-15f;
    const bool tall = s11 < s22;
    const float ux = tall ? s12 + eps : big - s22, uy = tall ? big - s11 : s12 + eps;
    const float scale = 1.0f / std::sqrt(ux * ux + uy * uy);
    const float e0x = ux * scale, e0y = uy * scale;
    const V3 centre{s.pos[0], -s.pos[1], 0.0f};
    const V3 major{e0x * r_big, -(e0y * r_big), 0.0f};
    const V3 minor{-e0y * r_small, -(e0x * r_small), 0.0f};
    put(centre, 255, 255, 255);
    put(centre + major, 255, 255, 255);
    put(centre, 255, 255, 255);
    put(centre + minor, 230, 230, 230);
    const unsigned cr = (unsigned)(s.color[0] * 255.0f), cg = (unsigned)(s.color[1] * 255.0f), cb = (unsigned)(s.color[2] * 255.0f);
    const int sides = 16;
    const float dtheta = 3.14159265358979323846264338327950288f * 2.0f / sides;
    const float sd = std::sin(dtheta), cd = std::cos(dtheta);
    float sn = 0.0f, cs = 1.0f;
    for (int k = 0; k <= sides; k++) {
        put(centre + major * sn + minor * cs, cr, cg, cb);
        const float sn2 = sn * cd + cs * sd, cs2 = cs * cd - sn * sd;
        sn = sn2;
        cs = cs2;
        put(centre + major * sn + minor * cs, cr, cg, cb);
    }
    const float hx = std::sqrt(i11 * det), hy = std::sqrt(i00 * det);
    const V3 corner[4] = {{-hx, -hy, 0.0f}, {hx, -hy, 0.0f}, {hx, hy, 0.0f}, {-hx, hy, 0.0f}};
    for (int k = 0; k < 4; k++) {
        put(centre + corner[k], 128, 128, 128);
        put(centre + corner[(k + 1) & 3], 128, 128, 128);
    }
}

// scale: output pixels per image pixel (the GUI's viewScale, main.cpp:822); `im` must already be that size.  Every pair
// of vertices is one line (PrimitiveMode::Lines, main.cpp:416), drawn in its second vertex's colour; scene y is image -y.
// `dump` (optional) receives the vertices as drawn, kOverlayVertices per splat.
inline void draw_splat_overlay(Image8* im, const std::vector<s2d_splat>& splats, int scale, int stride = 1,
                               std::vector<OverlayVertex>* dump = nullptr)
{
    const float S = (float)scale;
    OverlayVertex v[kOverlayVertices];
    for (size_t i = 0; i < splats.size(); i += (size_t)(stride < 1 ? 1 : stride)) {
        overlay_vertices(splats[i], v);
        if (dump) dump->insert(dump->end(), v, v + kOverlayVertices);
        bool finite = true;
        for (const OverlayVertex& q : v) finite = finite && std::isfinite(q.x) && std::isfinite(q.y);
        if (!finite) continue;
        for (int k = 0; k + 1 < kOverlayVertices; k += 2) {
            const uint8_t c[3] = {v[k + 1].r, v[k + 1].g, v[k + 1].b};
            draw_line(im, v[k].x * S, -v[k].y * S, v[k + 1].x * S, -v[k + 1].y * S, c);
        }
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
