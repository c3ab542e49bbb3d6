/*
 * s2d_oracle.h -- CPU restatement of the reference's training iteration.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product path:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library, and only as the checker / the timed CPU baseline.  The product
 * (2dgaussiansplatting_amd/csrc) never calls into it and has no CPU fallback.
 *
 * What it restates: /root/reference/main.cpp:414-807 (forward rasteriser,
 * analytic backward, Adam + constraints, finite guard, MSE) and the init at
 * main.cpp:280-305, expression by expression, evaluation order preserved,
 * compiled with `gcc -O2 -ffp-contract=off` and no -march (SURVEY.md §8c:
 * those are the flags under which the verbatim reference is deterministic).
 *
 * Pinning: the reference cannot be built in-repo (it needs the un-vendored
 * prlib submodule; writing stand-in headers is not allowed), and it ships no
 * tests.  The oracle is pinned against the known-answer vectors captured from
 * the verbatim reference during the survey (MSE traces
 * for three configurations up to 300 iterations, init positions, iteration-0
 * framebuffer checksum + sha256) -- see tests/test_oracle_kat.py.
 */
#ifndef S2D_ORACLE_H
#define S2D_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* main.cpp:85-93  struct Splat {vec2 pos; float sx, sy, rot; vec3 color; float opacity;} */
typedef struct {
    float pos_x, pos_y;
    float sx, sy, rot;
    float col_r, col_g, col_b;
    float opacity;
} s2do_splat;

/* main.cpp:139-166  struct Adam {m_m, m_v}; struct SplatAdam {pos[2], sx, sy, rot, color[3], opacity} */
typedef struct { float m, v; } s2do_adam;
typedef struct {
    s2do_adam pos[2];
    s2do_adam sx, sy, rot;
    s2do_adam color[3];
    s2do_adam opacity;
} s2do_splat_adam;

/* pair counters: visited = pixel-loop bodies entered after the x/y
 * clip, active = those that passed the T >= 1/256 test. */
typedef struct { uint64_t visited, active; } s2do_counters;

/* main.cpp:17-24 */
void s2do_pcg3d(uint32_t v[3]);

/* main.cpp:280-305: deterministic init of all splats, Adam state zeroed. */
void s2do_init(s2do_splat* splats, s2do_splat_adam* adams, int n, int W, int H);

/* main.cpp:414-546 restricted to rows [y0, y1).  image0 is W*H RGBA32F; rows in the
 * slab are cleared to (0,0,0,1), rasterised, then .w reset to 1 (main.cpp:543-546).
 * Rows outside the slab are not touched.  counters may be NULL. */
void s2do_forward_rows(const s2do_splat* splats, int n, int W, int H, int y0, int y1,
                       float* image0, s2do_counters* counters);

/* main.cpp:548-712 restricted to rows [y0, y1).  image1 is W*H RGBA32F scratch
 * (rows in the slab are cleared here).  dsplats (n records) is ACCUMULATED into:
 * the caller zeroes it (main.cpp:550 value-initialises it). */
void s2do_backward_rows(const s2do_splat* splats, int n, int W, int H, int y0, int y1,
                        const float* image0, const float* image_ref, float* image1,
                        s2do_splat* dsplats, s2do_counters* counters);

/* Same pass; additionally accumulates (into caller-zeroed n*9 doubles, record order) dsum = the sum of the
 * SAME fp32 per-pixel contributions carried in double, and dabs = the sum of their absolute values.  Used by
 * the parity tests to judge fp32 summation-order differences against the conditioning of each sum. */
void s2do_backward_rows_stats(const s2do_splat* splats, int n, int W, int H, int y0, int y1,
                              const float* image0, const float* image_ref, float* image1,
                              s2do_splat* dsplats, double* dsum, double* dabs);

/* main.cpp:714-785.  Multiplies *beta1t, *beta2t first (718-719), applies the nine
 * scalar Adam updates (opacity only if optimize_opacity), clamps, then the finite
 * guard.  Returns 0, or 1 where the reference would abort() (752-785). */
int s2do_adam_step(s2do_splat* splats, s2do_splat_adam* adams, const s2do_splat* dsplats,
                   int n, int W, int H, float* beta1t, float* beta2t,
                   int optimize_opacity, float training_rate);

/* main.cpp:796-805 restricted to rows [y0, y1): returns the un-normalised double sum
 * of lengthSquared(d*255); s2do_mse divides by (H*W*3). */
double s2do_sqerr_rows(const float* image0, const float* image_ref, int W, int H, int y0, int y1);
double s2do_mse(const float* image0, const float* image_ref, int W, int H);

/* One whole iteration (main.cpp:414-807) on all rows; *mse_out gets the value the
 * reference prints for this iteration.  image0/image1 are caller-provided scratch
 * of W*H*4 floats.  Returns s2do_adam_step's status. */
int s2do_step(s2do_splat* splats, s2do_splat_adam* adams, int n, int W, int H,
              const float* image_ref, float* image0, float* image1, s2do_splat* dsplats,
              float* beta1t, float* beta2t, int optimize_opacity, double* mse_out);

/* Same iteration with forward/backward split into `nthreads` row slabs run on
 * pthreads (the "all host cores" CPU baseline of SURVEY.md §8d).  Per-slab partial
 * gradients are summed in slab order, so results differ from s2do_step only by
 * fp32 summation order of the gradients. */
int s2do_step_mt(s2do_splat* splats, s2do_splat_adam* adams, int n, int W, int H,
                 const float* image_ref, float* image0, float* image1, s2do_splat* dsplats,
                 float* beta1t, float* beta2t, int optimize_opacity, double* mse_out,
                 int nthreads);

/* The reference's libm calls (main.cpp:212-213, 568-569), exposed so the tests can
 * check the device trig against exactly what the oracle used. */
float s2do_cosf(float x);
float s2do_sinf(float x);

/* main.cpp:49-83 */
float s2do_exp_approx(float x);
/* main.cpp:155 as an MSVC build evaluates it (sqrt(float) -> float, all-fp32 update) instead of the g++ form
 * (double quotient) the known-answer vectors were captured with. */
void s2do_set_adam_fp32(int on);
/* main.cpp:51: switch exp_approx to expf for every later call ("use this for numerical varidation"). */
void s2do_set_exact_exp(int on);

/* The per-splat debug drawing of the forward loop, main.cpp:419-477: the vertices the
 * reference hands to pr::PrimVertex(glm::vec3, glm::u8vec3) for splat s, in call order -- 2 axis segments (main.cpp:447-451),
 * 17 segments of the 16-gon (the loop runs i = 0..nvtx inclusive, main.cpp:457-462) and the 4 sides of the 1-sigma box
 * (main.cpp:464-477): S2DO_OVERLAY_VERTICES = 46 vertices per splat.  xyz gets 3 floats per vertex in the reference's
 * scene coordinates (x, -y, 0); rgb 3 bytes per vertex.  Restated from main.cpp: cov_of :206-221, eignValues :188-196,
 * eigen_vectors_of_cov :223-234, the inverse :432-436.  Restated from third-party code ABSENT from /root/reference
 * (prlib, un-vendored submodule, pinned commit not recoverable): pr::CircleGenerator, as the
 * angle-addition recurrence its published source implements (sin, cos start at 0, 1; step(): s' = s*cd + c*sd,
 * c' = c*cd - s*sd with sd, cd = sin, cos of the step angle), and glm::normalize(v) = v * (1 / sqrt(dot(v, v))).
 * The axes and the box depend on main.cpp's own arithmetic only; the 16-gon's vertices are "parity unpinned" with
 * respect to prlib's arithmetic. */
#define S2DO_OVERLAY_VERTICES 46
void s2do_overlay_vertices(const s2do_splat* splats, int n, float* xyz, uint8_t* rgb);

#ifdef __cplusplus
}
#endif
#endif
