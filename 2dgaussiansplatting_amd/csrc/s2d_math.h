// s2d_math.h -- per-splat and per-pixel arithmetic shared by every kernel.
//
// Everything here is a pure function of its arguments, written once and compiled
//   * by hipcc for gfx950 (the product), with -ffp-contract=off, and
//   * by g++ into tests' libs2d_hostcheck.so (unit tests of this arithmetic against
//     the oracle on a machine without a GPU -- a test shim, never a fallback).
// Each float expression keeps the reference's evaluation order so that the
// discrete decisions (pixel inclusion, T < 1/256 early-out, exp cut-off, clamps)
// come out exactly as in /root/reference/main.cpp.
#pragma once

#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define S2D_HD __host__ __device__ __forceinline__
#else
#define S2D_HD static inline
#endif

namespace s2d {

constexpr float kSplatBounds = 3.0f;             // SPLAT_BOUNDS, main.cpp:7
constexpr float kMinThroughput = 1.0f / 256.0f;  // MIN_THROUGHPUT, main.cpp:8
constexpr float kAdamBeta1 = 0.9f;               // main.cpp:136
constexpr float kAdamBeta2 = 0.99f;              // main.cpp:137
constexpr int kTile = 16;                        // tile edge in pixels (one 256-thread workgroup per tile)

// ----------------------------------------------------------------------------------------------
// sinf / cosf.
// The reference calls std::cosf / std::sinf (main.cpp:212-213, 568-569).  Under glibc 2.35 on an
// FMA+AVX2 x86-64 host those resolve to __sinf_fma / __cosf_fma: the "sincosf" algorithm of the
// ARM optimized routines (double-precision range reduction by pi/2 and two degree-3/4 minimax
// polynomials), compiled with FMA contraction.  This is a restatement of that published algorithm
// with the contractions written out as explicit fma() calls; tools/check_trig_exhaustive.cpp shows
// it returns the same bits as this container's libm for every float with |x| < 120 (2.2e9 values,
// 0 mismatches).  |x| >= 120 (the algorithm's large-argument path, unreachable for `rot` in
// practice: init is [0, pi] and Adam moves it <= ~0.12 per step) is evaluated from the double
// sin/cos and rounded once.
// ----------------------------------------------------------------------------------------------
S2D_HD float sincos_poly(double x, double x2, bool neg_cos, int n)
{
    // cosine coefficients change sign with the quadrant (table [1] of the published algorithm);
    // the sine's sign travels in x.
    const double c0 = neg_cos ? -0x1p0 : 0x1p0;
    const double c1 = neg_cos ? 0x1.ffffffd0c621cp-2 : -0x1.ffffffd0c621cp-2;
    const double c2 = neg_cos ? -0x1.55553e1068f19p-5 : 0x1.55553e1068f19p-5;
    const double c3 = neg_cos ? 0x1.6c087e89a359dp-10 : -0x1.6c087e89a359dp-10;
    const double c4 = neg_cos ? -0x1.99343027bf8c3p-16 : 0x1.99343027bf8c3p-16;
    const double s1 = -0x1.555545995a603p-3;
    const double s2 = 0x1.1107605230bc4p-7;
    const double s3 = -0x1.994eb3774cf24p-13;
    if ((n & 1) == 0) {
        double x3 = x * x2;
        double t1 = ::fma(x2, s3, s2);
        double x7 = x3 * x2;
        double s = ::fma(x3, s1, x);
        return (float)::fma(x7, t1, s);
    } else {
        double x4 = x2 * x2;
        double t2 = ::fma(x2, c4, c3);
        double t1 = ::fma(x2, c1, c0);
        double x6 = x4 * x2;
        double c = ::fma(x4, c2, t1);
        return (float)::fma(x6, t2, c);
    }
}

S2D_HD uint32_t f32_bits(float f)
{
    union { float f; uint32_t u; } v;
    v.f = f;
    return v.u;
}

S2D_HD uint32_t abstop12(float x) { return (f32_bits(x) >> 20) & 0x7ffu; }

// which = 0: sin, 1: cos
S2D_HD float sincos_f32(float y, int which)
{
    double x = (double)y;
    const uint32_t top = abstop12(y);
    if (top < abstop12(0x1.921FB6p-1f)) { // |y| < pi/4
        double x2 = x * x;
        if (top < abstop12(0x1p-12f))
            return which ? 1.0f : y;
        return sincos_poly(x, x2, false, which);
    }
    if (top < abstop12(120.0f)) {
        const double hpi_inv = 0x1.45F306DC9C883p+23; // 2/pi * 2^24
        const double hpi = 0x1.921FB54442D18p0;       // pi/2
        double r = x * hpi_inv;
        int n = ((int32_t)r + 0x800000) >> 24;
        x = ::fma(-(double)n, hpi, x);
        double s = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0; // sign[n & 3] = {1,-1,-1,1}
        return sincos_poly(x * s, x * x, (n & 2) != 0, n ^ which);
    }
    // large / non-finite arguments: not on any realistic training path (see header comment)
    return which ? (float)::cos(x) : (float)::sin(x);
}

S2D_HD float sinf_ref(float y) { return sincos_f32(y, 0); }
S2D_HD float cosf_ref(float y) { return sincos_f32(y, 1); }

// ----------------------------------------------------------------------------------------------
// small helpers, main.cpp:34-83, 171-185
// ----------------------------------------------------------------------------------------------
S2D_HD float sign_of(float v) { return v < 0.0f ? -1.0f : 1.0f; }           // main.cpp:34-37
S2D_HD float ss_max(float x, float y) { return (x < y) ? y : x; }           // main.cpp:38-42
S2D_HD float ss_min(float x, float y) { return (y < x) ? y : x; }           // main.cpp:44-48
S2D_HD float glm_mix(float x, float y, float a) { return x * (1.0f - a) + y * a; }
S2D_HD float glm_clamp(float x, float lo, float hi)
{
    float t = (x < lo) ? lo : x; // glm::max(x, lo)
    return (hi < t) ? hi : t;    // glm::min(t, hi); a NaN x stays NaN
}

// main.cpp:49-83.  x/8 is exact in binary, so it is written as a multiply.
S2D_HD float exp_approx(float x)
{
    x = 1.0f + x * 0.125f;
    if (x < 0.00001814586175896693021059036255f)
        return 0.0f;
    x *= x;
    x *= x;
    x *= x;
    return x;
}

// exp_approx(-0.5f * d2), main.cpp:527 / :610, in one fused step: -0.5*d2 and the division by 8 are exact
// scalings by powers of two, so 1 + (-0.5*d2)/8 has a single rounding -- the one fmaf(d2, -1/16, 1) performs.
// (If the scaling underflows, both forms give exactly 1.)  Bit-identical to the two-step form; checked against
// the oracle by tests/test_shared_math_host.py.
S2D_HD float gauss_from_d2(float d2)
{
    float x = ::fmaf(d2, -0.0625f, 1.0f);
    if (x < 0.00001814586175896693021059036255f)
        return 0.0f;
    x *= x;
    x *= x;
    x *= x;
    return x;
}

// gauss_from_d2 with the cut-off handed back as a predicate instead of applied: returns x^8 (x = 1 - d2/16) and
// *nonzero = !(x < cut-off).  gauss_from_d2(d2) == (*nonzero ? result : 0) bit for bit, NaN included; the raster
// kernels AND the predicate into the lane mask that selects alpha, saving one select per blended pixel.
S2D_HD float gauss_pow8(float d2, bool* nonzero)
{
    float x = ::fmaf(d2, -0.0625f, 1.0f);
    *nonzero = !(x < 0.00001814586175896693021059036255f);
    x *= x;
    x *= x;
    x *= x;
    return x;
}

// C float -> int conversion as the reference's x86-64 build performs it (cvttss2si): truncation
// toward zero, and the "integer indefinite" value INT_MIN for NaN / out-of-range inputs.
S2D_HD int cvt_trunc(float f)
{
    if (!(f > -2147483904.0f && f < 2147483648.0f))
        return (int)0x80000000u;
    return (int)f;
}

// sqrtf: IEEE correctly rounded on both sides (hipcc default: -fhip-fp32-correctly-rounded-divide-sqrt)
S2D_HD float sqrt_f32(float x) { return ::sqrtf(x); }

// ----------------------------------------------------------------------------------------------
// Projection of one splat: main.cpp:423-436 (forward) == :556-575 (backward), + :489-491.
// ----------------------------------------------------------------------------------------------
struct Splat { // == struct Splat, main.cpp:85-93
    float pos_x, pos_y, sx, sy, rot, col_r, col_g, col_b, opacity;
};

struct Projected {
    float pos_x, pos_y;
    float a, b, d;          // inv_cov[0][0], inv_cov[1][0] (== inv_cov[0][1] bit for bit), inv_cov[1][1]
    float col_r, col_g, col_b, opacity;
    int begY, endY;         // main.cpp:490-491
    float cosT, sinT, sx, sy;
    float hx;               // 3*sqrt(s11): half-width of the footprint's bounding box (binning only)
};

S2D_HD Projected project(const Splat& s)
{
    Projected p;
    // cov_of, main.cpp:206-221
    float cosTheta = cosf_ref(s.rot);
    float sinTheta = sinf_ref(s.rot);
    float lambda0 = s.sx * s.sx;
    float lambda1 = s.sy * s.sy;
    float s11 = lambda0 * cosTheta * cosTheta + lambda1 * sinTheta * sinTheta;
    float s12 = (lambda0 - lambda1) * sinTheta * cosTheta;
    float s22 = lambda0 + lambda1 - s11;
    float det = s11 * s22 - s12 * s12;  // main.cpp:191 / :560
    // main.cpp:432-436 / :561-565: mat2(cov11, -cov01, -cov10, cov00) / det, element-wise division
    p.a = s22 / det;
    p.b = -s12 / det;
    p.d = s11 / det;
    p.pos_x = s.pos_x; p.pos_y = s.pos_y;
    p.col_r = s.col_r; p.col_g = s.col_g; p.col_b = s.col_b; p.opacity = s.opacity;
    p.cosT = cosTheta; p.sinT = sinTheta; p.sx = s.sx; p.sy = s.sy;
    // main.cpp:489-491 / :573-575
    float hsize_invCovY = sqrt_f32(p.a * det) * kSplatBounds;
    p.begY = cvt_trunc(s.pos_y - hsize_invCovY);
    p.endY = cvt_trunc(s.pos_y + hsize_invCovY);
    // not in the reference: conservative x half-extent for binning.  The exact per-row ranges
    // (row_range below) lie inside pos_x +- 3*sqrt(s11) up to rounding; callers add a 1-pixel skirt.
    p.hx = sqrt_f32(ss_max(s11, 0.0f)) * kSplatBounds;
    return p;
}

// main.cpp:171-185 + :498-509 / :582-593: inclusive pixel-column range of the splat on row y.
// Returns false where solve_quadratic finds no root (begX = endX = -1 in the reference: no pixel).
S2D_HD bool row_range(float pos_x, float pos_y, float a, float b, float d, int y, int* begX, int* endX)
{
    float vy = ((float)y + 0.5f) - pos_y;
    float qb = 2.0f * b * vy;
    float qc = d * vy * vy - kSplatBounds * kSplatBounds;
    float det = qb * qb - 4.0f * a * qc;
    if (det < 0.0f)
        return false;
    float k = (-qb - sign_of(qb) * sqrt_f32(det)) / 2.0f;
    float x0 = k / a;
    float x1 = qc / k;
    float lo = ss_min(x0, x1);
    float hi = ss_max(x0, x1);
    *begX = cvt_trunc(pos_x + lo);
    *endX = cvt_trunc(pos_x + hi);
    return true;
}

// 16-bit inclusion mask of the splat on row y for the tile whose first column is x0:
// bit i set <=> pixel (x0+i, y) is visited by the reference's loops (main.cpp:492-514).
// The caller guarantees 0 <= y < H.
S2D_HD uint32_t row_mask16(float pos_x, float pos_y, float a, float b, float d, int begY, int endY,
                           int y, int x0, int W)
{
    if (y < begY || y > endY)
        return 0u;
    int begX, endX;
    if (!row_range(pos_x, pos_y, a, b, d, y, &begX, &endX))
        return 0u;
    int lo = begX > x0 ? begX : x0;
    int xmax = x0 + (kTile - 1);
    if (xmax > W - 1) xmax = W - 1;
    int hi = endX < xmax ? endX : xmax;
    if (lo > hi)
        return 0u;
    uint32_t upto_hi = (2u << (uint32_t)(hi - x0)) - 1u;  // bits 0..hi-x0
    uint32_t below_lo = (1u << (uint32_t)(lo - x0)) - 1u; // bits 0..lo-x0-1
    return upto_hi & ~below_lo;
}

// ----------------------------------------------------------------------------------------------
// Per-pixel blend, main.cpp:523-533 (forward) == :607-611, :623-625, :707 (backward).
// Returns G = exp_approx(-d2/2) and the pixel offset v; the caller applies alpha = G * opacity.
// ----------------------------------------------------------------------------------------------
S2D_HD float gauss_at(float px, float py, float pos_x, float pos_y, float a, float b, float d,
                      float* vx_out, float* vy_out)
{
    float vx = px - pos_x;
    float vy = py - pos_y;
    float mx = a * vx + b * vy; // inv_cov * v   (inv_cov[1][0] == inv_cov[0][1])
    float my = b * vx + d * vy;
    float d2 = vx * mx + vy * my;
    *vx_out = vx;
    *vy_out = vy;
    return gauss_from_d2(d2);
}

// ----------------------------------------------------------------------------------------------
// init(), main.cpp:17-24 + :280-305
// ----------------------------------------------------------------------------------------------
S2D_HD void pcg3d(uint32_t& x, uint32_t& y, uint32_t& z)
{
    x = x * 1664525u + 1013904223u;
    y = y * 1664525u + 1013904223u;
    z = z * 1664525u + 1013904223u;
    x += y * z; y += z * x; z += x * y;
    x ^= x >> 16u; y ^= y >> 16u; z ^= z >> 16u;
    x += y * z; y += z * x; z += x * y;
}

S2D_HD Splat init_splat(uint32_t i, int W, int H)
{
    const float denom = 4294967296.0f; // glm::vec3(0xFFFFFFFF) in fp32
    const float pi = 3.14159265358979323846264338327950288f;
    uint32_t ax = i, ay = 0u, az = 0xFFFFFFFFu;
    uint32_t bx = i, by = 1u, bz = 0xFFFFFFFFu;
    pcg3d(ax, ay, az);
    pcg3d(bx, by, bz);
    float r0x = (float)ax / denom, r0y = (float)ay / denom;
    float r1x = (float)bx / denom, r1y = (float)by / denom, r1z = (float)bz / denom;
    Splat s;
    s.pos_x = glm_mix(r0x, (float)W - 1.0f, r0x); // main.cpp:294 (sic)
    s.pos_y = glm_mix(r0y, (float)H - 1.0f, r0y); // main.cpp:295
    s.sx = glm_mix(6.0f, 10.0f, r1x);
    s.sy = glm_mix(6.0f, 10.0f, r1y);
    s.rot = pi * r1z;
    s.col_r = 0.5f; s.col_g = 0.5f; s.col_b = 0.5f;
    s.opacity = 1.0f;
    return s;
}

// ----------------------------------------------------------------------------------------------
// Adam::optimize, main.cpp:144-156.  `sqrt` at :155 is unqualified.  In the Linux build the known-answer vectors
// come from (g++ / clang++) it binds ::sqrt(double): the quotient and the subtraction are
// evaluated in double and rounded to float once -- the default here, because that is what the oracle can be pinned
// to.  The reference itself is an MSVC program, where <cmath> puts the float overload of sqrt in the global
// namespace and the whole expression is fp32: `fp32_quotient` (S2D_CFG_ADAM_FP32) selects that form.  The two differ
// by at most one ulp of the parameter plus ~3 ulp of the 0.05 update per step (tests/test_oracle_kat.py).  m, v, m_hat, v_hat, s*m_hat are fp32 either way.
// ----------------------------------------------------------------------------------------------
S2D_HD float adam_optimize(float& m_m, float& m_v, float value, float g, float alpha, float beta1t, float beta2t,
                           bool fp32_quotient = false)
{
    float m = kAdamBeta1 * m_m + (1.0f - kAdamBeta1) * g;
    float v = kAdamBeta2 * m_v + (1.0f - kAdamBeta2) * g * g;
    m_m = m;
    m_v = v;
    float m_hat = m / (1.0f - beta1t);
    float v_hat = v / (1.0f - beta2t);
    const float ADAM_E = 1.0e-15f;
    float sm = alpha * m_hat;
    if (fp32_quotient)
        return value - sm / (sqrt_f32(v_hat) + ADAM_E); // MSVC: sqrt(float) -> float
    return (float)((double)value - (double)sm / (::sqrt((double)v_hat) + (double)ADAM_E));
}

} // namespace s2d
