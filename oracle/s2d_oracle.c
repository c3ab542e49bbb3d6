/*
 * s2d_oracle.c -- CPU restatement of /root/reference/main.cpp:414-807 (+ init 280-305).
 *
 * TEST INFRASTRUCTURE ONLY (see s2d_oracle.h).  Build: `make -C oracle`
 * (gcc -O2 -ffp-contract=off, no -march: every float operation below is one IEEE
 * binary32 operation, in the reference's evaluation order).
 *
 * Each block cites the reference lines it follows.  Where glm is involved the
 * semantics restated are: mat2 is column-major, m[col][row];
 * mat2/float divides element-wise; mat2*vec2 = (m00*x + m10*y, m01*x + m11*y);
 * dot(vec2) = x*x' + y*y'; dot(vec3) = (x*x' + y*y') + z*z'; mix(x,y,a) =
 * x*(1-a) + y*a; clamp(x,lo,hi) = min(max(x,lo),hi).
 */
#include "s2d_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define SPLAT_BOUNDS 3.0f                 /* main.cpp:7 */
#define MIN_THROUGHPUT (1.0f / 256.0f)    /* main.cpp:8 */

static const float ADAM_BETA1 = 0.9f;     /* main.cpp:136 */
static const float ADAM_BETA2 = 0.99f;    /* main.cpp:137 */

float s2do_cosf(float x) { return cosf(x); }
float s2do_sinf(float x) { return sinf(x); }

/* main.cpp:17-24 */
void s2do_pcg3d(uint32_t v[3])
{
    v[0] = v[0] * 1664525u + 1013904223u;
    v[1] = v[1] * 1664525u + 1013904223u;
    v[2] = v[2] * 1664525u + 1013904223u;
    v[0] += v[1] * v[2]; v[1] += v[2] * v[0]; v[2] += v[0] * v[1];
    v[0] ^= v[0] >> 16u; v[1] ^= v[1] >> 16u; v[2] ^= v[2] >> 16u;
    v[0] += v[1] * v[2]; v[1] += v[2] * v[0]; v[2] += v[0] * v[1];
}

/* main.cpp:34-37 */
static inline float sign_of(float v) { return v < 0.0f ? -1.0f : 1.0f; }
/* main.cpp:38-48 */
static inline float ss_max(float x, float y) { return (x < y) ? y : x; }
static inline float ss_min(float x, float y) { return (y < x) ? y : x; }

/* main.cpp:51: "return expf( x ); // use this for numerical varidation" -- a process-wide switch of this test
 * library (never set while the known-answer tests run): when on, exp_approx IS expf, as in the reference with that
 * line un-commented. */
static int g_exact_exp = 0;
void s2do_set_exact_exp(int on) { g_exact_exp = on; }

/* main.cpp:49-83 */
float s2do_exp_approx(float x)
{
    if (g_exact_exp)
        return expf(x); /* main.cpp:51 */
    x = 1.0f + x / 8.0f;
    if (x < 0.00001814586175896693021059036255f) /* avoid subnormal */
        return 0.0f;
    x *= x;
    x *= x;
    x *= x;
    return x;
}

/* main.cpp:171-185 */
static int solve_quadratic(float xs[2], float a, float b, float c)
{
    float det = b * b - 4.0f * a * c;
    if (det < 0.0f)
        return 0;
    float k = (-b - sign_of(b) * sqrtf(det)) / 2.0f;
    float x0 = k / a;
    float x1 = c / k;
    xs[0] = ss_min(x0, x1);
    xs[1] = ss_max(x0, x1);
    return 2;
}

static inline float mixf(float x, float y, float a) { return x * (1.0f - a) + y * a; }
/* glm::clamp = min(max(x,lo),hi) with glm::max(x,y) = (x<y)?y:x and glm::min(x,y) = (y<x)?y:x:
 * a NaN x stays NaN (so the finite guard at main.cpp:752-785 can see it). */
static inline float clampf(float x, float lo, float hi)
{
    float t = (x < lo) ? lo : x;
    return (hi < t) ? hi : t;
}

/* main.cpp:280-305 */
void s2do_init(s2do_splat* splats, s2do_splat_adam* adams, int n, int W, int H)
{
    if (adams) memset(adams, 0, sizeof(s2do_splat_adam) * (size_t)n); /* 285-286 */
    const float denom = (float)0xFFFFFFFFu; /* glm::vec3(0xFFFFFFFF): 4294967296.0f */
    const float pi = 3.14159265358979323846264338327950288f; /* glm::pi<float>() */
    for (int i = 0; i < n; i++) {
        uint32_t a[3] = { (uint32_t)i, 0u, 0xFFFFFFFFu };
        uint32_t b[3] = { (uint32_t)i, 1u, 0xFFFFFFFFu };
        s2do_pcg3d(a);
        s2do_pcg3d(b);
        float r0x = (float)a[0] / denom, r0y = (float)a[1] / denom;
        float r1x = (float)b[0] / denom, r1y = (float)b[1] / denom, r1z = (float)b[2] / denom;
        s2do_splat s;
        s.pos_x = mixf(r0x, (float)W - 1, r0x); /* 294 (sic: x argument is r0.x) */
        s.pos_y = mixf(r0y, (float)H - 1, r0y); /* 295 */
        s.sx = mixf(6.0f, 10.0f, r1x);          /* 296 */
        s.sy = mixf(6.0f, 10.0f, r1y);          /* 297 */
        s.rot = pi * r1z;                       /* 300 */
        s.col_r = s.col_g = s.col_b = 0.5f;     /* 301 */
        s.opacity = 1.0f;                       /* 302 */
        splats[i] = s;
    }
}

/* Per-splat set-up shared by both passes: main.cpp:423-436 / 556-569, 489-491 / 573-575. */
typedef struct {
    float a, b, c, d;      /* inv_cov[0][0], [1][0], [0][1], [1][1] */
    float cosT, sinT;
    int begY, endY;
} splat_setup;

static inline splat_setup setup_of(const s2do_splat* s)
{
    splat_setup o;
    /* cov_of, main.cpp:206-221 */
    float cosTheta = s2do_cosf(s->rot);
    float sinTheta = s2do_sinf(s->rot);
    float lambda0 = s->sx * s->sx;
    float lambda1 = s->sy * s->sy;
    float s11 = lambda0 * cosTheta * cosTheta + lambda1 * sinTheta * sinTheta;
    float s12 = (lambda0 - lambda1) * sinTheta * cosTheta;
    float s22 = lambda0 + lambda1 - s11;
    /* det: eignValues main.cpp:191 (fwd) == inline main.cpp:560 (bwd) */
    float det = s11 * s22 - s12 * s12;
    /* inverse, main.cpp:432-436 / 561-565: mat2(cov11, -cov01, -cov10, cov00) / det */
    float inv00 = s22 / det;
    float inv01 = -s12 / det;
    float inv10 = -s12 / det;
    float inv11 = s11 / det;
    o.a = inv00; o.b = inv10; o.c = inv01; o.d = inv11;
    o.cosT = cosTheta; o.sinT = sinTheta;
    /* y range, main.cpp:489-491 / 573-575 (C float->int truncation) */
    float hsize_invCovY = sqrtf(inv00 * det) * SPLAT_BOUNDS;
    o.begY = (int)(s->pos_y - hsize_invCovY);
    o.endY = (int)(s->pos_y + hsize_invCovY);
    return o;
}

/* main.cpp:498-509 / 582-593 */
static inline void row_range(const s2do_splat* s, const splat_setup* u, int y, int* begX, int* endX)
{
    float vy = (y + 0.5f) - s->pos_y;
    float xs[2];
    *begX = -1;
    *endX = -1;
    if (solve_quadratic(xs, u->a, 2.0f * u->b * vy, u->d * vy * vy - SPLAT_BOUNDS * SPLAT_BOUNDS)) {
        *begX = (int)(s->pos_x + xs[0]);
        *endX = (int)(s->pos_x + xs[1]);
    }
}

/* main.cpp:414-546, rows [y0,y1) */
void s2do_forward_rows(const s2do_splat* splats, int n, int W, int H, int y0, int y1,
                       float* image0, s2do_counters* counters)
{
    uint64_t visited = 0, active = 0;
    if (y0 < 0) y0 = 0;
    if (y1 > H) y1 = H;
    for (int y = y0; y < y1; y++)
        for (int x = 0; x < W; x++) { /* 414 */
            float* p = image0 + 4 * ((size_t)y * W + x);
            p[0] = 0.0f; p[1] = 0.0f; p[2] = 0.0f; p[3] = 1.0f;
        }

    for (int i = 0; i < n; i++) { /* 419 */
        s2do_splat s = splats[i];
        splat_setup u = setup_of(&s);
        for (int y = u.begY; y <= u.endY; y++) { /* 492 */
            if (y < 0 || H <= y) continue;      /* 494 */
            if (y < y0 || y1 <= y) continue;    /* slab restriction (not in reference) */
            int begX, endX;
            row_range(&s, &u, y, &begX, &endX);
            for (int x = begX; x <= endX; x++) { /* 511 */
                if (x < 0 || W <= x) continue;  /* 513 */
                visited++;
                float* color = image0 + 4 * ((size_t)y * W + x); /* 517 */
                float T = color[3];
                if (T < MIN_THROUGHPUT) continue; /* 520 */
                active++;
                float vx = (x + 0.5f) - s.pos_x; /* 523-524 */
                float vy = (y + 0.5f) - s.pos_y;
                float mx = u.a * vx + u.b * vy;  /* inv_cov * v */
                float my = u.c * vx + u.d * vy;
                float d2 = vx * mx + vy * my;    /* 526 */
                float alpha = s2do_exp_approx(-0.5f * d2) * s.opacity; /* 527 */
                color[0] += T * s.col_r * alpha; /* 529-531 */
                color[1] += T * s.col_g * alpha;
                color[2] += T * s.col_b * alpha;
                color[3] *= (1.0f - alpha);      /* 533 */
            }
        }
    }
    for (int y = y0; y < y1; y++) /* 543-546 */
        for (int x = 0; x < W; x++)
            image0[4 * ((size_t)y * W + x) + 3] = 1.0f;
    if (counters) { counters->visited += visited; counters->active += active; }
}

/* main.cpp:548-712, rows [y0,y1).  STATS (compile-time constant after inlining) additionally accumulates
 * every fp32 contribution into double sums (dsum: the same terms summed without fp32 rounding) and their
 * absolute values (dabs: the scale against which a summation-order difference has to be judged). */
static inline __attribute__((always_inline)) void backward_rows_impl(
    const s2do_splat* splats, int n, int W, int H, int y0, int y1, const float* image0, const float* image_ref,
    float* image1, s2do_splat* dsplats, s2do_counters* counters, const int STATS, double* dsum, double* dabs)
{
    uint64_t visited = 0, active = 0;
    if (y0 < 0) y0 = 0;
    if (y1 > H) y1 = H;
    for (int y = y0; y < y1; y++)
        for (int x = 0; x < W; x++) { /* 549 */
            float* p = image1 + 4 * ((size_t)y * W + x);
            p[0] = 0.0f; p[1] = 0.0f; p[2] = 0.0f; p[3] = 1.0f;
        }

    for (int i = 0; i < n; i++) { /* 552 */
        s2do_splat s = splats[i];
        splat_setup u = setup_of(&s);
        float cosTheta = u.cosT, sinTheta = u.sinT; /* 567-569 */
        s2do_splat* dS = &dsplats[i];
        for (int y = u.begY; y <= u.endY; y++) { /* 576 */
            if (y < 0 || H <= y) continue;      /* 578 */
            if (y < y0 || y1 <= y) continue;    /* slab restriction (not in reference) */
            int begX, endX;
            row_range(&s, &u, y, &begX, &endX);
            for (int x = begX; x <= endX; x++) { /* 595 */
                if (x < 0 || W <= x) continue;  /* 597 */
                visited++;
                size_t px = 4 * ((size_t)y * W + x);
                float* color = image1 + px;     /* 601 */
                float T = color[3];
                if (T < MIN_THROUGHPUT) continue; /* 604 */
                active++;
                float vx = (x + 0.5f) - s.pos_x; /* 607-608 */
                float vy = (y + 0.5f) - s.pos_y;
                float mx = u.a * vx + u.b * vy;
                float my = u.c * vx + u.d * vy;
                float d2 = vx * mx + vy * my;    /* 609 */
                float G = s2do_exp_approx(-0.5f * d2); /* 610 */
                float alpha = G * s.opacity;     /* 611 */

                const float* finalColor = image0 + px; /* 613 */
                const float* ref = image_ref + px;
                /* dL/dc, 616-620 */
                float dL_dC_x = finalColor[0] - ref[0];
                float dL_dC_y = finalColor[1] - ref[1];
                float dL_dC_z = finalColor[2] - ref[2];
                {
                    float dC_dc = alpha * T;
                    dS->col_r += dL_dC_x * dC_dc;
                    dS->col_g += dL_dC_y * dC_dc;
                    dS->col_b += dL_dC_z * dC_dc;
                }
                /* color accumulation, 623-625 */
                color[0] += T * s.col_r * alpha;
                color[1] += T * s.col_g * alpha;
                color[2] += T * s.col_b * alpha;

                /* 627-630 */
                float Sx = finalColor[0] - color[0];
                float Sy = finalColor[1] - color[1];
                float Sz = finalColor[2] - color[2];
                float den = 1.0f - alpha + 1.0e-15f;
                float dC_dalpha_x = s.col_r * T - Sx / den;
                float dC_dalpha_y = s.col_g * T - Sy / den;
                float dC_dalpha_z = s.col_b * T - Sz / den;
                float dL_dalpha_x = dL_dC_x * dC_dalpha_x;
                float dL_dalpha_y = dL_dC_y * dC_dalpha_y;
                float dL_dalpha_z = dL_dC_z * dC_dalpha_z;
                float dL_dalpha_rgb = dL_dalpha_x + dL_dalpha_y + dL_dalpha_z;
                {
                    float a = u.a, b = u.b, c = u.c, d = u.d; /* 635-638 */
                    float dalpha_dx = 0.5f * alpha * (2.0f * a * vx + (b + c) * vy); /* 639 */
                    float dalpha_dy = 0.5f * alpha * (2.0f * d * vy + (b + c) * vx); /* 640 */
                    dS->pos_x += dL_dalpha_rgb * dalpha_dx; /* 654-655 */
                    dS->pos_y += dL_dalpha_rgb * dalpha_dy;

                    float vxx = vx * vx, vxy = vx * vy, vyy = vy * vy;
                    /* 657-662: alpha/(s^3) * dot(vec3, vec3), dot = (x+y)+z */
                    float dalpha_dsx = alpha / (s.sx * s.sx * s.sx) *
                        (((cosTheta * cosTheta) * vxx + (2.0f * sinTheta * cosTheta) * vxy) + (sinTheta * sinTheta) * vyy);
                    float dalpha_dsy = alpha / (s.sy * s.sy * s.sy) *
                        (((sinTheta * sinTheta) * vxx + (-2.0f * sinTheta * cosTheta) * vxy) + (cosTheta * cosTheta) * vyy);
                    dS->sx += dL_dalpha_rgb * dalpha_dsx; /* 677-678 */
                    dS->sy += dL_dalpha_rgb * dalpha_dsy;

                    /* 680-685 */
                    float dalpha_dtheta =
                        alpha * (s.sx * s.sx - s.sy * s.sy) / (s.sx * s.sx * s.sy * s.sy) *
                        ((cosTheta * cosTheta - sinTheta * sinTheta) * vx * vy - sinTheta * cosTheta * (vx * vx - vy * vy));
                    dS->rot += (dL_dalpha_x + dL_dalpha_y + dL_dalpha_z) * dalpha_dtheta;

                    float dalpha_do = G; /* 703-704 */
                    dS->opacity += dL_dalpha_rgb * dalpha_do;

                    if (STATS) {
                        const float t[9] = { dL_dalpha_rgb * dalpha_dx, dL_dalpha_rgb * dalpha_dy,
                                             dL_dalpha_rgb * dalpha_dsx, dL_dalpha_rgb * dalpha_dsy,
                                             (dL_dalpha_x + dL_dalpha_y + dL_dalpha_z) * dalpha_dtheta,
                                             dL_dC_x * (alpha * T), dL_dC_y * (alpha * T), dL_dC_z * (alpha * T),
                                             dL_dalpha_rgb * dalpha_do };
                        for (int k = 0; k < 9; k++) {
                            dsum[(size_t)i * 9 + k] += (double)t[k];
                            dabs[(size_t)i * 9 + k] += fabs((double)t[k]);
                        }
                    }
                }
                color[3] *= (1.0f - alpha); /* 707 */
            }
        }
    }
    if (counters) { counters->visited += visited; counters->active += active; }
}

void s2do_backward_rows(const s2do_splat* splats, int n, int W, int H, int y0, int y1,
                        const float* image0, const float* image_ref, float* image1,
                        s2do_splat* dsplats, s2do_counters* counters)
{
    backward_rows_impl(splats, n, W, H, y0, y1, image0, image_ref, image1, dsplats, counters, 0, NULL, NULL);
}

void s2do_backward_rows_stats(const s2do_splat* splats, int n, int W, int H, int y0, int y1,
                              const float* image0, const float* image_ref, float* image1,
                              s2do_splat* dsplats, double* dsum, double* dabs)
{
    backward_rows_impl(splats, n, W, H, y0, y1, image0, image_ref, image1, dsplats, NULL, 1, dsum, dabs);
}

/* main.cpp:144-156.  `sqrt` at :155 is unqualified: under g++/clang++ on Linux (where
 * the survey's known-answer vectors were captured) it binds ::sqrt(double), so the
 * quotient and the final subtraction are evaluated in double and rounded to float on
 * return; `s * m_hat` is still a float product. */
/* The reference is an MSVC program; there the unqualified sqrt binds the float overload and the expression is fp32
 * throughout.  s2do_set_adam_fp32(1) selects that form (process-wide switch of this test library, off while the
 * known-answer tests run): it lets the tests measure what the unverifiable choice costs. */
static int g_adam_fp32 = 0;
void s2do_set_adam_fp32(int on) { g_adam_fp32 = on; }

static inline float adam_optimize(s2do_adam* st, float value, float g, float alpha, float beta1t, float beta2t)
{
    float s = alpha;
    float m = ADAM_BETA1 * st->m + (1.0f - ADAM_BETA1) * g;
    float v = ADAM_BETA2 * st->v + (1.0f - ADAM_BETA2) * g * g;
    st->m = m;
    st->v = v;
    float m_hat = m / (1.0f - beta1t);
    float v_hat = v / (1.0f - beta2t);
    const float ADAM_E = 1.0e-15f;
    float sm = s * m_hat;
    if (g_adam_fp32)
        return value - sm / (sqrtf(v_hat) + ADAM_E);
    return (float)((double)value - (double)sm / (sqrt((double)v_hat) + (double)ADAM_E));
}

/* main.cpp:714-785 */
int s2do_adam_step(s2do_splat* splats, s2do_splat_adam* adams, const s2do_splat* dsplats,
                   int n, int W, int H, float* beta1t, float* beta2t,
                   int optimize_opacity, float training_rate)
{
    *beta1t *= ADAM_BETA1; /* 718-719 */
    *beta2t *= ADAM_BETA2;
    float b1 = *beta1t, b2 = *beta2t;
    for (int i = 0; i < n; i++) {
        s2do_splat* s = &splats[i];
        s2do_splat_adam* a = &adams[i];
        const s2do_splat* d = &dsplats[i];
        s->col_r = adam_optimize(&a->color[0], s->col_r, d->col_r, training_rate, b1, b2); /* 723-725 */
        s->col_g = adam_optimize(&a->color[1], s->col_g, d->col_g, training_rate, b1, b2);
        s->col_b = adam_optimize(&a->color[2], s->col_b, d->col_b, training_rate, b1, b2);
        s->pos_x = adam_optimize(&a->pos[0], s->pos_x, d->pos_x, training_rate, b1, b2);   /* 727-728 */
        s->pos_y = adam_optimize(&a->pos[1], s->pos_y, d->pos_y, training_rate, b1, b2);
        s->sx = adam_optimize(&a->sx, s->sx, d->sx, training_rate, b1, b2);                /* 730-731 */
        s->sy = adam_optimize(&a->sy, s->sy, d->sy, training_rate, b1, b2);
        s->rot = adam_optimize(&a->rot, s->rot, d->rot, training_rate, b1, b2);            /* 733 */
        if (optimize_opacity)                                                              /* 735-738 */
            s->opacity = adam_optimize(&a->opacity, s->opacity, d->opacity, training_rate, b1, b2);
        /* constraints, 741-749 */
        s->pos_x = clampf(s->pos_x, 0.0f, (float)W - 1);
        s->pos_y = clampf(s->pos_y, 0.0f, (float)H - 1);
        s->sx = clampf(s->sx, 1.0f, 1024.0f);
        s->sy = clampf(s->sy, 1.0f, 1024.0f);
        s->col_r = clampf(s->col_r, 0.0f, 1.0f);
        s->col_g = clampf(s->col_g, 0.0f, 1.0f);
        s->col_b = clampf(s->col_b, 0.0f, 1.0f);
        s->opacity = clampf(s->opacity, 0.1f, 1.0f);
    }
    /* finite guard, 752-785: pos.y and opacity are NOT checked by the reference */
    for (int i = 0; i < n; i++) {
        const s2do_splat* s = &splats[i];
        if (!isfinite(s->col_r) || !isfinite(s->col_g) || !isfinite(s->col_b) ||
            !isfinite(s->sx) || !isfinite(s->sy) || !isfinite(s->rot) || !isfinite(s->pos_x))
            return 1;
    }
    return 0;
}

/* main.cpp:796-805 */
double s2do_sqerr_rows(const float* image0, const float* image_ref, int W, int H, int y0, int y1)
{
    double mse = 0.0;
    if (y0 < 0) y0 = 0;
    if (y1 > H) y1 = H;
    for (int y = y0; y < y1; y++)
        for (int x = 0; x < W; x++) {
            size_t px = 4 * ((size_t)y * W + x);
            float dx = (image0[px + 0] - image_ref[px + 0]) * 255.0f;
            float dy = (image0[px + 1] - image_ref[px + 1]) * 255.0f;
            float dz = (image0[px + 2] - image_ref[px + 2]) * 255.0f;
            mse += dx * dx + dy * dy + dz * dz; /* lengthSquared(vec3), main.cpp:131-134 */
        }
    return mse;
}

double s2do_mse(const float* image0, const float* image_ref, int W, int H)
{
    double mse = s2do_sqerr_rows(image0, image_ref, W, H, 0, H);
    mse /= (H * W * 3);
    return mse;
}

int s2do_step(s2do_splat* splats, s2do_splat_adam* adams, int n, int W, int H,
              const float* image_ref, float* image0, float* image1, s2do_splat* dsplats,
              float* beta1t, float* beta2t, int optimize_opacity, double* mse_out)
{
    s2do_forward_rows(splats, n, W, H, 0, H, image0, NULL);
    memset(dsplats, 0, sizeof(s2do_splat) * (size_t)n); /* 550 */
    s2do_backward_rows(splats, n, W, H, 0, H, image0, image_ref, image1, dsplats, NULL);
    int st = s2do_adam_step(splats, adams, dsplats, n, W, H, beta1t, beta2t, optimize_opacity, 0.05f /* 715 */);
    if (mse_out) *mse_out = s2do_mse(image0, image_ref, W, H);
    return st;
}

/* ---- row-slab threaded variant (CPU baseline on all host cores) ---- */
typedef struct {
    const s2do_splat* splats; int n, W, H, y0, y1;
    float* image0; const float* image_ref; float* image1; s2do_splat* dpart;
    int pass;
} slab_job;

static void* slab_run(void* p)
{
    slab_job* j = (slab_job*)p;
    if (j->pass == 0)
        s2do_forward_rows(j->splats, j->n, j->W, j->H, j->y0, j->y1, j->image0, NULL);
    else
        s2do_backward_rows(j->splats, j->n, j->W, j->H, j->y0, j->y1, j->image0, j->image_ref, j->image1, j->dpart, NULL);
    return NULL;
}

int s2do_step_mt(s2do_splat* splats, s2do_splat_adam* adams, int n, int W, int H,
                 const float* image_ref, float* image0, float* image1, s2do_splat* dsplats,
                 float* beta1t, float* beta2t, int optimize_opacity, double* mse_out,
                 int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > H) nthreads = H;
    slab_job* jobs = (slab_job*)calloc((size_t)nthreads, sizeof(slab_job));
    pthread_t* th = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
    s2do_splat* parts = (s2do_splat*)calloc((size_t)nthreads * (size_t)n, sizeof(s2do_splat));
    for (int pass = 0; pass < 2; pass++) {
        for (int t = 0; t < nthreads; t++) {
            slab_job* j = &jobs[t];
            j->splats = splats; j->n = n; j->W = W; j->H = H;
            j->y0 = (int)((long long)H * t / nthreads);
            j->y1 = (int)((long long)H * (t + 1) / nthreads);
            j->image0 = image0; j->image_ref = image_ref; j->image1 = image1;
            j->dpart = parts + (size_t)t * n; j->pass = pass;
            pthread_create(&th[t], NULL, slab_run, j);
        }
        for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    }
    memset(dsplats, 0, sizeof(s2do_splat) * (size_t)n);
    for (int t = 0; t < nthreads; t++) {
        const float* src = (const float*)(parts + (size_t)t * n);
        float* dst = (float*)dsplats;
        for (size_t k = 0; k < (size_t)n * 9; k++) dst[k] += src[k];
    }
    free(parts); free(th); free(jobs);
    int st = s2do_adam_step(splats, adams, dsplats, n, W, H, beta1t, beta2t, optimize_opacity, 0.05f);
    if (mse_out) *mse_out = s2do_mse(image0, image_ref, W, H);
    return st;
}

/* ------------------------------------------------------------------------------------------------------------------
 * The per-splat debug drawing, main.cpp:419-477 (see s2d_oracle.h).  Every expression below is the reference's, in its
 * order; glm::vec3 + glm::vec3 * float + glm::vec3 * float evaluates left to right, component by component.
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct { float x, y, z; } ov3;
static inline ov3 ov_add(ov3 a, ov3 b) { ov3 r = { a.x + b.x, a.y + b.y, a.z + b.z }; return r; }
static inline ov3 ov_mul(ov3 a, float s) { ov3 r = { a.x * s, a.y * s, a.z * s }; return r; }

static void ov_emit(float** xyz, uint8_t** rgb, ov3 p, unsigned r, unsigned g, unsigned b)
{
    (*xyz)[0] = p.x; (*xyz)[1] = p.y; (*xyz)[2] = p.z;
    (*rgb)[0] = (uint8_t)r; (*rgb)[1] = (uint8_t)g; (*rgb)[2] = (uint8_t)b; /* glm::u8vec3 from ints / from a uvec3 */
    *xyz += 3;
    *rgb += 3;
}

void s2do_overlay_vertices(const s2do_splat* splats, int n, float* xyz, uint8_t* rgb)
{
    const float pi = 3.14159265358979323846264338327950288f; /* glm::pi<float>() */
    for (int i = 0; i < n; i++) {
        const s2do_splat s = splats[i];                       /* main.cpp:421 */
        /* cov_of, main.cpp:206-221 */
        const float cosTheta = cosf(s.rot), sinTheta = sinf(s.rot);
        const float l0 = s.sx * s.sx, l1 = s.sy * s.sy;
        const float s11 = l0 * cosTheta * cosTheta + l1 * sinTheta * sinTheta;
        const float s12 = (l0 - l1) * sinTheta * cosTheta;
        const float s22 = l0 + l1 - s11;
        /* eignValues, main.cpp:188-196 (mat[0][0] = s11, mat[1][1] = s22, mat[1][0] = mat[0][1] = s12) */
        const float mean = (s11 + s22) * 0.5f;
        const float det = s11 * s22 - s12 * s12;
        const float d = sqrtf(ss_max(mean * mean - det, 0.0f));
        const float lambda0 = mean + d, lambda1 = mean - d;
        const float sqrt_of_lambda0 = sqrtf(lambda0), sqrt_of_lambda1 = sqrtf(lambda1); /* main.cpp:429-430 */
        /* inv_cov = mat2(cov[1][1], -cov[0][1], -cov[1][0], cov[0][0]) / det, main.cpp:432-436 */
        const float inv00 = s22 / det, inv11 = s11 / det;
        /* eigen_vectors_of_cov, main.cpp:223-234 */
        const float eps = 1e-15f;
        float ex, ey;
        if (s11 < s22) { ex = s12 + eps; ey = lambda0 - s11; }
        else           { ex = lambda0 - s22; ey = s12 + eps; }
        const float inv_len = 1.0f / sqrtf(ex * ex + ey * ey); /* glm::normalize: v * inversesqrt(dot(v, v)) */
        const float e0x = ex * inv_len, e0y = ey * inv_len;
        const float e1x = -e0y, e1y = e0x;
        /* main.cpp:443-444 */
        const float axis0x = e0x * sqrt_of_lambda0, axis0y = e0y * sqrt_of_lambda0;
        const float axis1x = e1x * sqrt_of_lambda1, axis1y = e1y * sqrt_of_lambda1;
        const ov3 P = { s.pos_x, -s.pos_y, 0.0f };
        const ov3 A0 = { axis0x, -axis0y, 0.0f }, A1 = { axis1x, -axis1y, 0.0f };
        /* axes, main.cpp:447-451 */
        ov_emit(&xyz, &rgb, P, 255, 255, 255);
        ov_emit(&xyz, &rgb, ov_add(P, A0), 255, 255, 255);
        ov_emit(&xyz, &rgb, P, 255, 255, 255);
        ov_emit(&xyz, &rgb, ov_add(P, A1), 230, 230, 230);
        /* ellipse, main.cpp:454-462; glm::uvec3 col = s.color * 255.0f, then u8vec3 at the call */
        const int nvtx = 16;
        const float step = pi * 2.0f / nvtx;
        const float sd = sinf(step), cd = cosf(step);          /* pr::CircleGenerator (prlib, absent: see header) */
        float cs = 0.0f, cc = 1.0f;
        const unsigned cr = (unsigned)(s.col_r * 255.0f), cg = (unsigned)(s.col_g * 255.0f), cb = (unsigned)(s.col_b * 255.0f);
        for (int k = 0; k <= nvtx; k++) {
            ov_emit(&xyz, &rgb, ov_add(ov_add(P, ov_mul(A0, cs)), ov_mul(A1, cc)), cr, cg, cb);
            const float ns = cs * cd + cc * sd, nc = cc * cd - cs * sd;
            cs = ns;
            cc = nc;
            ov_emit(&xyz, &rgb, ov_add(ov_add(P, ov_mul(A0, cs)), ov_mul(A1, cc)), cr, cg, cb);
        }
        /* the exact 1-sigma bounding box, main.cpp:464-477 */
        const float hx = sqrtf(inv11 * det), hy = sqrtf(inv00 * det);
        const ov3 vs[4] = { { -hx, -hy, 0.0f }, { +hx, -hy, 0.0f }, { +hx, +hy, 0.0f }, { -hx, +hy, 0.0f } };
        for (int k = 0; k < 4; k++) {
            ov_emit(&xyz, &rgb, ov_add(P, vs[k]), 128, 128, 128);
            ov_emit(&xyz, &rgb, ov_add(P, vs[(k + 1) % 4]), 128, 128, 128);
        }
    }
}
