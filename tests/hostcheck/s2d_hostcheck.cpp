// s2d_hostcheck.cpp -- TEST SHIM.  Compiles the kernels' shared arithmetic header
// (2dgaussiansplatting_amd/csrc/s2d_math.h) for the host so that tests can compare it
// with the oracle on a machine without a GPU.  It is not a CPU fallback: the product
// library never links or calls this file.
#include "../../2dgaussiansplatting_amd/csrc/s2d_math.h"
#include "../../2dgaussiansplatting_amd/host/overlay.h" // the host program's vertex list of main.cpp:447-476 (row f3)

#include <cstring>

using namespace s2d;

extern "C" {

void hc_sincos(const float* x, int n, float* s, float* c)
{
    for (int i = 0; i < n; i++) { s[i] = sinf_ref(x[i]); c[i] = cosf_ref(x[i]); }
}

void hc_init(float* splats9, int n, int W, int H)
{
    for (int i = 0; i < n; i++) {
        Splat s = init_splat((uint32_t)i, W, H);
        std::memcpy(splats9 + 9 * (size_t)i, &s, sizeof(Splat));
    }
}

// Forward image by the kernels' per-pixel rule: for every pixel, walk the splats in index order,
// include the splat iff its row mask has the pixel's bit (row_mask16), blend while T >= 1/256.
void hc_forward(const float* splats9, int n, int W, int H, float* image0)
{
    Projected* pr = new Projected[n];
    for (int i = 0; i < n; i++) {
        Splat s;
        std::memcpy(&s, splats9 + 9 * (size_t)i, sizeof(Splat));
        pr[i] = project(s);
    }
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            float cr = 0.f, cg = 0.f, cb = 0.f, T = 1.f;
            int x0 = (x / kTile) * kTile;
            for (int i = 0; i < n; i++) {
                const Projected& p = pr[i];
                uint32_t m = row_mask16(p.pos_x, p.pos_y, p.a, p.b, p.d, p.begY, p.endY, y, x0, W);
                if (!((m >> (x - x0)) & 1u)) continue;
                if (T < kMinThroughput) continue;
                float vx, vy;
                float G = gauss_at((float)x + 0.5f, (float)y + 0.5f, p.pos_x, p.pos_y, p.a, p.b, p.d, &vx, &vy);
                float alpha = G * p.opacity;
                cr += T * p.col_r * alpha;
                cg += T * p.col_g * alpha;
                cb += T * p.col_b * alpha;
                T *= (1.0f - alpha);
            }
            float* o = image0 + 4 * ((size_t)y * W + x);
            o[0] = cr; o[1] = cg; o[2] = cb; o[3] = 1.0f;
        }
    delete[] pr;
}

// One Adam scalar (main.cpp:144-156) through the shared helper.
float hc_adam(float* m, float* v, float value, float g, float lr, float b1t, float b2t)
{
    return adam_optimize(*m, *v, value, g, lr, b1t, b2t);
}

// Bounding rectangle test: does the conservative x extent used for binning (pos_x +- (hx + 1)) contain
// every column the exact per-row ranges visit?  Returns the number of violations over all rows.
int hc_check_bounds(const float* splats9, int n, int W, int H)
{
    int bad = 0;
    for (int i = 0; i < n; i++) {
        Splat s;
        std::memcpy(&s, splats9 + 9 * (size_t)i, sizeof(Splat));
        Projected p = project(s);
        float lo = p.pos_x - p.hx - 1.0f, hi = p.pos_x + p.hx + 1.0f;
        for (int y = p.begY; y <= p.endY; y++) {
            if (y < 0 || y >= H) continue;
            int bx, ex;
            if (!row_range(p.pos_x, p.pos_y, p.a, p.b, p.d, y, &bx, &ex)) continue;
            if (bx > ex) continue;
            int cb = bx < 0 ? 0 : bx, ce = ex > W - 1 ? W - 1 : ex;
            if (cb > ce) continue;
            if ((float)cb < lo - 1.0f || (float)ce > hi) bad++;
        }
    }
    return bad;
}

// The host program's overlay vertices (host/overlay.h overlay_vertices) for n splats: 46 x (3 floats, 3 bytes) each.
void hc_overlay_vertices(const float* splats9, int n, float* xyz, unsigned char* rgb)
{
    for (int i = 0; i < n; i++) {
        s2d_splat sp;
        std::memcpy(&sp, splats9 + 9 * (size_t)i, sizeof(sp));
        s2dio::OverlayVertex v[s2dio::kOverlayVertices];
        s2dio::overlay_vertices(sp, v);
        for (int k = 0; k < s2dio::kOverlayVertices; k++) {
            const size_t at = (size_t)i * s2dio::kOverlayVertices + (size_t)k;
            xyz[3 * at] = v[k].x, xyz[3 * at + 1] = v[k].y, xyz[3 * at + 2] = v[k].z;
            rgb[3 * at] = v[k].r, rgb[3 * at + 1] = v[k].g, rgb[3 * at + 2] = v[k].b;
        }
    }
}

} // extern "C"
