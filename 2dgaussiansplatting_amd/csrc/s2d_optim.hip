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

// main.cpp:721-785 for one splat: the nine Adam updates, the constraints and the finite guard, on values held in
// registers.  The scalar order of Splat (pos.xy, sx, sy, rot, color.rgb, opacity) and of SplatAdam (pos[2], sx, sy, rot,
// color[3], opacity) is the same, so scalar k of the splat pairs with Adam slot k.  The nine updates are independent,
// so the reference's update order (color, pos, sx, sy, rot, opacity; main.cpp:723-738) does not matter.
// mode: bit 0 = optimizeOpacity (main.cpp:735-738), bit 1 = fp32 Adam quotient (S2D_CFG_ADAM_FP32).
__device__ __forceinline__ void adam_update_one(float (&v)[9], float (&mv)[18], const float (&gr)[9], int W, int H, float beta1t,
                                                float beta2t, float lr, int mode, int iteration, DeviceStatus* status)
{
#pragma unroll
    for (int k = 0; k < 9; k++)
        if (k < 8 || (mode & 1))
            v[k] = adam_optimize(mv[2 * k], mv[2 * k + 1], v[k], gr[k], lr, beta1t, beta2t, (mode & 2) != 0);
    // constraints, main.cpp:741-749
    v[0] = glm_clamp(v[0], 0.0f, (float)W - 1.0f);
    v[1] = glm_clamp(v[1], 0.0f, (float)H - 1.0f);
    v[2] = glm_clamp(v[2], 1.0f, 1024.0f);
    v[3] = glm_clamp(v[3], 1.0f, 1024.0f);
    v[5] = glm_clamp(v[5], 0.0f, 1.0f);
    v[6] = glm_clamp(v[6], 0.0f, 1.0f);
    v[7] = glm_clamp(v[7], 0.0f, 1.0f);
    v[8] = glm_clamp(v[8], 0.1f, 1.0f);
    // finite guard, main.cpp:752-785: color.xyz, sx, sy, rot, pos.x (pos.y and opacity are not checked)
    const bool ok = finite_f32(v[5]) && finite_f32(v[6]) && finite_f32(v[7]) && finite_f32(v[2]) &&
                    finite_f32(v[3]) && finite_f32(v[4]) && finite_f32(v[0]);
    if (!ok) {
        atomicOr(&status->nonfinite, 1);
        atomicMin(&status->first_nonfinite_iter, iteration);
    }
}

// Projection of the UPDATED splat for the next iteration's raster (main.cpp:423-436, 489-491) and the check against
// the rectangle its tile lists were built from, so that the next iteration needs no separate pass over the parameters.
__device__ __forceinline__ void project_updated(const float (&v)[9], int i, const Geometry& g, DeviceStatus* status,
                                                ProjRec* __restrict__ proj, const TileRect* __restrict__ rects,
                                                int check_stamp, int* __restrict__ host_stamp)
{
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

// The reference abort()s at the first non-finite parameter (main.cpp:752-785): later iterations do nothing.  (Strictly
// earlier: blocks of the detecting launch itself, which stores `iteration`, must all finish their work.)  Then the MSE of
// the iteration (main.cpp:796-805) from the tile errors the backward pass left, by the launch's first workgroups
// (block-uniform: all 256 threads take it together).  Returns false when the launch is to do nothing.
__device__ __forceinline__ bool adam_prologue(const DeviceStatus* status, int iteration, const SqerrJob& sq)
{
    if (status->first_nonfinite_iter < iteration) return false;
    if (sq.tile_sqerr != nullptr && blockIdx.x < (unsigned)kSqerrChunks)
        sqerr_reduce(sq.tile_sqerr, sq.num_tiles, sq.out, sq.scratch, (int)blockIdx.x, min((int)gridDim.x, kSqerrChunks));
    return true;
}

// Adam launch: one splat per thread, 256 records per block, every array moved through LDS so that global memory is
// accessed in runs of consecutive dwords instead of one record per lane.  (A thread reading its own 36-byte record
// dword by dword makes every load instruction touch 18 cache lines per wave, 36 for the 72-byte moments: the
// record-by-record form of this kernel ran at 1.8 TB/s, 192 us at 10^6 splats.)
//   * all splats in index order (ids == nullptr): a block's records are contiguous -- whole float4 lines;
//   * slab OWNERSHIP (s2d_halo.hip): the block walks 256 entries of the rank's compact, ascending list of held splats;
//     element e of the block's copy is dword e % w of record ids[e / w], so consecutive lanes still read consecutive
//     dwords of a record (and usually of neighbouring records).
// Also re-zeroes the gradient records (main.cpp:550 value-initialises dSplats every iteration).
template <int WIDTH>
__device__ __forceinline__ void lds_fill(float* lds, const float* __restrict__ src, const uint32_t* s_ids, int base, int cnt)
{
    const int floats = cnt * WIDTH;
    if (s_ids == nullptr) { // contiguous and 16-byte aligned: a block starts at a multiple of 256 records
        const float* p = src + (size_t)base * WIDTH;
        const int vec = floats >> 2;
        for (int q = threadIdx.x; q < vec; q += 256) reinterpret_cast<float4*>(lds)[q] = reinterpret_cast<const float4*>(p)[q];
        for (int q = (vec << 2) + threadIdx.x; q < floats; q += 256) lds[q] = p[q];
    } else {
        for (int q = threadIdx.x; q < floats; q += 256) lds[q] = src[(size_t)s_ids[q / WIDTH] * WIDTH + q % WIDTH];
    }
}

template <int WIDTH>
__device__ __forceinline__ void lds_drain(float* __restrict__ dst, const float* lds, const uint32_t* s_ids, int base, int cnt)
{
    const int floats = cnt * WIDTH;
    if (s_ids == nullptr) {
        float* p = dst + (size_t)base * WIDTH;
        const int vec = floats >> 2;
        for (int q = threadIdx.x; q < vec; q += 256) reinterpret_cast<float4*>(p)[q] = reinterpret_cast<const float4*>(lds)[q];
        for (int q = (vec << 2) + threadIdx.x; q < floats; q += 256) p[q] = lds[q];
    } else {
        for (int q = threadIdx.x; q < floats; q += 256) dst[(size_t)s_ids[q / WIDTH] * WIDTH + q % WIDTH] = lds[q];
    }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ splats, float* __restrict__ adams,
                                                   float* __restrict__ grads, const uint32_t* __restrict__ held_ids,
                                                   const uint32_t* __restrict__ held_count, int n, Geometry g,
                                                   float beta1t, float beta2t, float lr, int mode, int iteration,
                                                   DeviceStatus* __restrict__ status, ProjRec* __restrict__ proj,
                                                   const TileRect* __restrict__ rects, int check_stamp,
                                                   int* __restrict__ host_stamp, uint8_t* __restrict__ dormant, SqerrJob sq,
                                                   int compact)
{
    __shared__ __attribute__((aligned(16))) float buf[256 * 18];
    __shared__ uint32_t s_idbuf[256];
    if (!adam_prologue(status, iteration, sq)) return;
    const int total = held_ids ? (int)min(*held_count, (uint32_t)n) : n;
    const int base = blockIdx.x * 256, cnt = min(256, total - base), t = threadIdx.x;
    if (cnt <= 0) return;
    const bool mine = t < cnt;
    const uint32_t* s_ids = nullptr;
    int i = base + t;
    if (held_ids) { // only the splats this rank holds, from their compact list
        if (mine) {
            i = (int)held_ids[base + t];
            s_idbuf[t] = (uint32_t)i;
        }
        s_ids = s_idbuf;
        __syncthreads();
    }
    // compact: record base + t of `splats` / `adams` IS splat ids[t]'s (the rank's held splats in a compact array of their
    // own: whole lines, like the all-splats case); the gradient records stay where the raster kernels' atomics put them
    const uint32_t* const s_ids_state = compact ? nullptr : s_ids;
    float v[9], mv[18], gr[9];
    // gradients in, zeros out
    lds_fill<9>(buf, grads, s_ids, base, cnt);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 9; k++) gr[k] = mine ? buf[t * 9 + k] : 0.0f;
    // A splat no live pixel sees receives a zero gradient; if its moments are zero as well (dormant[i]: they were after its
    // last full update, and nothing but this kernel has touched it since) the whole of main.cpp:721-750 leaves it exactly as
    // it is: m = v = 0, the step 0 / (0 + 1e-15), the clamps already applied.  Splats are blended in index order, so the
    // hidden ones sit together at the high indices (about 60 % of 10^6 on a 4096^2 image): a block made of them has nothing
    // to read, write, project or check beyond the gradients it has just looked at.
    bool live = false;
    if (mine) {
#pragma unroll
        for (int k = 0; k < 9; k++) live = live || (gr[k] != 0.0f);
        if (!live) live = dormant == nullptr || dormant[i] == 0;
    }
    if (!__syncthreads_or(live)) return;
    for (int q = t; q < 256 * 9 / 4; q += 256) reinterpret_cast<float4*>(buf)[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    lds_drain<9>(grads, buf, s_ids, base, cnt);
    __syncthreads();
    // parameters in
    lds_fill<9>(buf, splats, s_ids_state, base, cnt);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 9; k++) v[k] = mine ? buf[t * 9 + k] : 0.0f;
    __syncthreads();
    // moments in
    lds_fill<18>(buf, adams, s_ids_state, base, cnt);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 18; k++) mv[k] = mine ? buf[t * 18 + k] : 0.0f;
    if (mine) {
        adam_update_one(v, mv, gr, g.W, g.H, beta1t, beta2t, lr, mode, iteration, status);
        if (dormant) { // all eighteen moments zero (+0 or -0): the next zero gradient changes nothing
            uint32_t any = 0u;
#pragma unroll
            for (int k = 0; k < 18; k++) any |= f32_bits(mv[k]) << 1;
            dormant[i] = any == 0u ? 1 : 0;
        }
        // moments out (each thread rewrites only its own record of the block's copy; the opacity slot goes back
        // unchanged when the checkbox is off)
#pragma unroll
        for (int k = 0; k < 18; k++) buf[t * 18 + k] = mv[k];
    }
    __syncthreads();
    lds_drain<18>(adams, buf, s_ids_state, base, cnt);
    __syncthreads();
    // parameters out
    if (mine) {
#pragma unroll
        for (int k = 0; k < 9; k++) buf[t * 9 + k] = v[k];
    }
    __syncthreads();
    lds_drain<9>(splats, buf, s_ids_state, base, cnt);
    if (proj && mine) project_updated(v, i, g, status, proj, rects, check_stamp, host_stamp);
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

// image_ref holds rows [row_begin, row_end) of the image (the context's slab).
template <bool HALF>
__global__ __launch_bounds__(256) void synthetic_target_kernel(void* __restrict__ image_ref, int W, int H, int row_begin,
                                                               int row_end)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = row_begin + (int)blockIdx.y;
    if (x >= W || y >= row_end) return;
    const float fx = (float)x / (float)W;
    const float4 c = make_float4(fx, 1.0f - fx, (float)y / (float)H, 1.0f);
    const size_t at = (size_t)(y - row_begin) * W + x;
    if (HALF) reinterpret_cast<uint2*>(image_ref)[at] = pack_half4(c);
    else reinterpret_cast<float4*>(image_ref)[at] = c;
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
                       const TileRect* rects, int check_stamp, int* host_stamp, uint8_t* dormant, SqerrJob sq,
                       bool compact, hipStream_t stream)
{
    if (n <= 0) return hipSuccess; // (callers queue the standalone squared-error reduction themselves when n == 0)
    hipLaunchKernelGGL(adam_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, splats, adams, grads, held_ids, held_count, n, g,
                       beta1t, beta2t, lr, optimize_opacity, iteration, status, proj, rects, check_stamp, host_stamp, dormant, sq,
                       (compact && held_ids) ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_synthetic_target(void* image_ref, bool half_images, int W, int H, int row_begin, int row_end, hipStream_t stream)
{
    const dim3 grid((W + 255) / 256, row_end - row_begin);
    if (half_images)
        hipLaunchKernelGGL(synthetic_target_kernel<true>, grid, dim3(256), 0, stream, image_ref, W, H, row_begin, row_end);
    else
        hipLaunchKernelGGL(synthetic_target_kernel<false>, grid, dim3(256), 0, stream, image_ref, W, H, row_begin, row_end);
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
