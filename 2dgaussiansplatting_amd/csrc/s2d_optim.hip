// s2d_optim.hip -- init(), the Adam step with constraints and finite guard, and small utilities.
#include <hip/hip_fp16.h>

#include "s2d_device.h"

namespace s2d {

// init(), main.cpp:280-305: one thread per splat; Adam state zeroed (main.cpp:285-286).
__global__ __launch_bounds__(256) void init_splats_kernel(float* __restrict__ splats, float* __restrict__ adams, int n,
                                                          int W, int H)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Splat s = init_splat((uint32_t)i, W, H);
    float* o = splats + (size_t)i * 9;
    o[0] = s.pos_x; o[1] = s.pos_y; o[2] = s.sx; o[3] = s.sy; o[4] = s.rot;
    o[5] = s.col_r; o[6] = s.col_g; o[7] = s.col_b; o[8] = s.opacity;
    float* a = adams + (size_t)i * 18;
#pragma unroll
    for (int k = 0; k < 18; k++) a[k] = 0.0f;
}

__device__ __forceinline__ bool finite_f32(float x) { return (f32_bits(x) & 0x7f800000u) != 0x7f800000u; }

// main.cpp:721-785 for one splat per thread.  The scalar order of Splat (pos.xy, sx, sy, rot, color.rgb,
// opacity) and of SplatAdam (pos[2], sx, sy, rot, color[3], opacity) is the same, so scalar k of the splat
// pairs with Adam slot k.  The nine updates are independent, so the reference's update order (color, pos,
// sx, sy, rot, opacity; main.cpp:723-738) does not matter.  Also re-zeroes the gradient record
// (main.cpp:550 value-initialises dSplats every iteration).
// With proj != nullptr the thread goes on to project the updated splat (main.cpp:423-436, 489-491 of the NEXT
// iteration's forward) and to check it against the rectangle its tile lists were built from, so that the next
// iteration needs no separate projection pass over the parameters.
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ splats, float* __restrict__ adams,
                                                   float* __restrict__ grads, const uint32_t* __restrict__ held_ids,
                                                   const uint32_t* __restrict__ held_count, int n, Geometry g,
                                                   float beta1t,
                                                   float beta2t, float lr, int optimize_opacity, int iteration,
                                                   DeviceStatus* __restrict__ status, ProjRec* __restrict__ proj,
                                                   const TileRect* __restrict__ rects, int check_stamp,
                                                   int* __restrict__ host_stamp, SqerrJob sq)
{
    const int W = g.W, H = g.H;
    // The reference abort()s at the first non-finite parameter (main.cpp:752-785): later iterations do nothing.
    // (Strictly earlier: blocks of the detecting launch itself, which stores `iteration`, must all finish their work.)
    if (status->first_nonfinite_iter < iteration) return;
    // MSE of the iteration (main.cpp:796-805) from the tile errors the backward pass left, by this launch's first
    // workgroups (block-uniform branch: all 256 threads take it together)
    if (sq.tile_sqerr != nullptr && blockIdx.x < (unsigned)kSqerrChunks)
        sqerr_reduce(sq.tile_sqerr, sq.num_tiles, sq.out, sq.scratch, (int)blockIdx.x, min((int)gridDim.x, kSqerrChunks));
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (held_ids) { // slab ownership (s2d_halo.hip): only the splats this rank holds, from their compact list
        if ((uint32_t)i >= *held_count) return;
        i = (int)held_ids[i];
    }
    float* sp = splats + (size_t)i * 9;
    float* ad = adams + (size_t)i * 18;
    float* gr = grads + (size_t)i * 9;
    float v[9];
#pragma unroll
    for (int k = 0; k < 9; k++) {
        v[k] = sp[k];
        if (k < 8 || (optimize_opacity & 1)) { // main.cpp:735-738; bit 1 of the argument: fp32 Adam quotient
            float m_m = ad[2 * k], m_v = ad[2 * k + 1];
            v[k] = adam_optimize(m_m, m_v, v[k], gr[k], lr, beta1t, beta2t, (optimize_opacity & 2) != 0);
            ad[2 * k] = m_m;
            ad[2 * k + 1] = m_v;
        }
        gr[k] = 0.0f;
    }
    // constraints, main.cpp:741-749
    v[0] = glm_clamp(v[0], 0.0f, (float)W - 1.0f);
    v[1] = glm_clamp(v[1], 0.0f, (float)H - 1.0f);
    v[2] = glm_clamp(v[2], 1.0f, 1024.0f);
    v[3] = glm_clamp(v[3], 1.0f, 1024.0f);
    v[5] = glm_clamp(v[5], 0.0f, 1.0f);
    v[6] = glm_clamp(v[6], 0.0f, 1.0f);
    v[7] = glm_clamp(v[7], 0.0f, 1.0f);
    v[8] = glm_clamp(v[8], 0.1f, 1.0f);
#pragma unroll
    for (int k = 0; k < 9; k++) sp[k] = v[k];
    // finite guard, main.cpp:752-785: color.xyz, sx, sy, rot, pos.x (pos.y and opacity are not checked)
    const bool ok = finite_f32(v[5]) && finite_f32(v[6]) && finite_f32(v[7]) && finite_f32(v[2]) &&
                    finite_f32(v[3]) && finite_f32(v[4]) && finite_f32(v[0]);
    if (!ok) {
        atomicOr(&status->nonfinite, 1);
        atomicMin(&status->first_nonfinite_iter, iteration);
    }
    if (proj) {
        // A rank that owns a row slab only ever reads the records of splats that can touch its rows.  A splat whose
        // 3-sigma circle (plus the 1-pixel skirt) stays clear of the slab has an empty exact rectangle, which every
        // binned rectangle covers: skip its projection (7/8 of the splats at 8 ranks).  NaNs fall through.
        const float reach = 3.0f * fmaxf(v[2], v[3]) + 2.0f;
        if (v[1] + reach < (float)g.row_begin || v[1] - reach > (float)g.row_end) {
            // ... but the re-used tile lists may still name it (it was inside when they were built, and one Adam step
            // can carry it out by any distance for a large training_rate or loaded moments): leave a record with
            // an empty row range (begY > endY) behind, so that the raster kernels see no footprint instead of its
            // stale one.
            proj[i].q2 = make_float4(v[8], as_f(1), as_f(0), 0.0f);
            return;
        }
        Splat s;
        s.pos_x = v[0]; s.pos_y = v[1]; s.sx = v[2]; s.sy = v[3]; s.rot = v[4];
        s.col_r = v[5]; s.col_g = v[6]; s.col_b = v[7]; s.opacity = v[8];
        const Projected p = project(s);
        proj[i] = pack_proj(p);
        if (!rect_still_covers(p, g, rects[i])) raise_rebin(status, check_stamp, host_stamp);
    }
}

// ref(x,y) = (x/W, 1 - x/W, y/H, 1): main.cpp:261-267's commented generator plus a blue ramp.
__device__ __forceinline__ uint2 pack_half4(float4 c)
{
    const __half2 a = __floats2half2_rn(c.x, c.y), b = __floats2half2_rn(c.z, c.w);
    uint2 v;
    v.x = *reinterpret_cast<const uint32_t*>(&a);
    v.y = *reinterpret_cast<const uint32_t*>(&b);
    return v;
}

template <bool HALF>
__global__ __launch_bounds__(256) void synthetic_target_kernel(void* __restrict__ image_ref, int W, int H)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= W || y >= H) return;
    const float fx = (float)x / (float)W;
    const float4 c = make_float4(fx, 1.0f - fx, (float)y / (float)H, 1.0f);
    if (HALF) reinterpret_cast<uint2*>(image_ref)[(size_t)y * W + x] = pack_half4(c);
    else reinterpret_cast<float4*>(image_ref)[(size_t)y * W + x] = c;
}

__global__ __launch_bounds__(256) void convert_f32_to_f16_kernel(const float4* __restrict__ src, uint2* __restrict__ dst,
                                                                 size_t pixels)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < pixels) dst[i] = pack_half4(src[i]);
}

__global__ __launch_bounds__(256) void convert_f16_to_f32_kernel(const uint2* __restrict__ src, float4* __restrict__ dst,
                                                                 size_t pixels)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pixels) return;
    const uint2 v = src[i];
    const float2 a = __half22float2(*reinterpret_cast<const __half2*>(&v.x));
    const float2 b = __half22float2(*reinterpret_cast<const __half2*>(&v.y));
    dst[i] = make_float4(a.x, a.y, b.x, b.y);
}

__global__ __launch_bounds__(256) void test_sincos_kernel(const float* __restrict__ x, int n, float* __restrict__ s,
                                                          float* __restrict__ c)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    s[i] = sinf_ref(x[i]);
    c[i] = cosf_ref(x[i]);
}

hipError_t launch_init_splats(float* splats, float* adams, int n, int W, int H, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(init_splats_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, splats, adams, n, W, H);
    return hipGetLastError();
}

hipError_t launch_adam(float* splats, float* adams, float* grads, const uint32_t* held_ids, const uint32_t* held_count,
                       int n, Geometry g, float beta1t, float beta2t,
                       float lr, int optimize_opacity, int iteration, DeviceStatus* status, ProjRec* proj,
                       const TileRect* rects, int check_stamp, int* host_stamp, SqerrJob sq, hipStream_t stream)
{
    if (n <= 0) return hipSuccess; // (callers queue the standalone squared-error reduction themselves when n == 0)
    hipLaunchKernelGGL(adam_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, splats, adams, grads, held_ids, held_count, n, g, beta1t,
                       beta2t, lr, optimize_opacity, iteration, status, proj, rects, check_stamp, host_stamp, sq);
    return hipGetLastError();
}

hipError_t launch_synthetic_target(void* image_ref, bool half_images, int W, int H, hipStream_t stream)
{
    if (half_images)
        hipLaunchKernelGGL(synthetic_target_kernel<true>, dim3((W + 255) / 256, H), dim3(256), 0, stream, image_ref, W, H);
    else
        hipLaunchKernelGGL(synthetic_target_kernel<false>, dim3((W + 255) / 256, H), dim3(256), 0, stream, image_ref, W, H);
    return hipGetLastError();
}

hipError_t launch_convert_f32_to_f16(const float4* src, void* dst, size_t pixels, hipStream_t stream)
{
    if (!pixels) return hipSuccess;
    hipLaunchKernelGGL(convert_f32_to_f16_kernel, dim3((unsigned)((pixels + 255) / 256)), dim3(256), 0, stream, src,
                       reinterpret_cast<uint2*>(dst), pixels);
    return hipGetLastError();
}

hipError_t launch_convert_f16_to_f32(const void* src, float4* dst, size_t pixels, hipStream_t stream)
{
    if (!pixels) return hipSuccess;
    hipLaunchKernelGGL(convert_f16_to_f32_kernel, dim3((unsigned)((pixels + 255) / 256)), dim3(256), 0, stream,
                       reinterpret_cast<const uint2*>(src), dst, pixels);
    return hipGetLastError();
}

hipError_t launch_test_sincos(const float* x, int n, float* s, float* c, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(test_sincos_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, x, n, s, c);
    return hipGetLastError();
}

} // namespace s2d
