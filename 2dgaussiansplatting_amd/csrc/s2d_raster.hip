// s2d_raster.hip -- tile-binned forward rasteriser and analytic backward pass.
//
// One 256-thread workgroup (4 wave64) per 16x16 image tile, one thread per pixel; wave w owns
// rows 4w..4w+3 of the tile, so lane = 16*(row & 3) + column.  A tile walks its splat list
// (ascending splat index == the reference's blend order, main.cpp:419/:552) in batches of 64
// entries staged in LDS:
//   * 4 threads per entry load the 64-byte projected record and evaluate the reference's exact
//     per-row column range (solve_quadratic + int truncation, main.cpp:498-509) for 4 rows each,
//     producing one 64-bit lane mask per (entry, wave): bit l set <=> the reference's loops visit
//     that pixel for that splat.  The quadratic is solved once per (entry, row), not per pixel.
//   * the blend loop then reads one wave-uniform mask per entry, skips the entry with a scalar
//     branch when no live lane is covered, and otherwise evaluates main.cpp:523-533 per lane.
// A pixel whose throughput fell below 1/256 never works again (main.cpp:520); when all 256 pixels
// of the tile are in that state the workgroup stops walking its list (tile retirement).
//
// Forward arithmetic is the reference's, operation for operation (-ffp-contract=off), so the
// framebuffer is bit-identical to the oracle's.  The backward pass recomputes T and the running
// colour the same way, then reduces each splat's nine partial gradients over the wave with DPP,
// over the four waves through per-wave LDS slots (plain stores, fixed order), and issues one global
// float-atomic burst per (tile, splat) into the N x 9 gradient array.
#include "s2d_device.h"

namespace s2d {

constexpr int B = kRasterBatch;

// Blocks b and b+8 share an XCD (round-robin dispatch): give each XCD a contiguous run of tiles so
// that neighbouring tiles, which share most of their splat records, hit the same L2.  Speed only.
__device__ __forceinline__ int tile_of_block(int bid, int num_tiles)
{
    const int per = (num_tiles + 7) >> 3;
    const int t = (bid & 7) * per + (bid >> 3);
    return t < num_tiles ? t : -1;
}

__device__ __forceinline__ unsigned long long wave_uniform_u64(unsigned long long v)
{
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// 64-bit lane mask of entry (q0, q1, begY, endY) for the wave that owns rows y_first..y_first+3.
__device__ __forceinline__ unsigned long long wave_mask_of(const float4& q0, const float4& q1, int begY, int endY,
                                                           int y_first, int x0, int W, int row_end)
{
    unsigned long long m = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int yy = y_first + k;
        uint32_t rm = 0;
        if (yy < row_end) rm = row_mask16(q0.x, q0.y, q0.z, q0.w, q1.x, begY, endY, yy, x0, W);
        m |= (unsigned long long)rm << (16 * k);
    }
    return m;
}

// ---------------------------------------------------------------------------------------------------
// forward, main.cpp:414-546
// ---------------------------------------------------------------------------------------------------
template <bool COUNT>
__global__ __launch_bounds__(256) void raster_forward_kernel(const uint32_t* __restrict__ tile_off,
                                                             const uint32_t* __restrict__ list,
                                                             const ProjRec* __restrict__ proj,
                                                             float4* __restrict__ image0, Geometry g,
                                                             PairCounters* __restrict__ counters)
{
    __shared__ float4 s_q0[B];
    __shared__ float4 s_q1[B];
    __shared__ float s_op[B];
    __shared__ unsigned long long s_mask[B * 4];

    const int tile = tile_of_block(blockIdx.x, g.num_tiles);
    if (tile < 0) return;
    const int tx = tile % g.tiles_x;
    const int ty = tile / g.tiles_x + g.trow0;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int x = tx * kTile + (tid & 15);
    const int y = ty * kTile + (tid >> 4);
    const bool inside = x < g.W && y < g.row_end;
    const float px = (float)x + 0.5f, py = (float)y + 0.5f; // main.cpp:523

    float cr = 0.0f, cg = 0.0f, cb = 0.0f, T = 1.0f;       // main.cpp:414: (0,0,0,1)
    bool alive = inside;
    unsigned long long n_vis = 0, n_act = 0, n_staged = 0, n_exec = 0;

    const uint32_t beg = tile_off[tile], end = tile_off[tile + 1];
    const int se = tid >> 2, sub = tid & 3;
    for (uint32_t base = beg; base < end; base += B) {
        const int cnt = (int)min((uint32_t)B, end - base);
        if (se < cnt) {
            const ProjRec* r = proj + list[base + se];
            const float4 q0 = r->q0, q1 = r->q1, q2 = r->q2;
            s_mask[se * 4 + sub] = wave_mask_of(q0, q1, __float_as_int(q2.y), __float_as_int(q2.z),
                                                ty * kTile + sub * 4, tx * kTile, g.W, g.row_end);
            if (sub == 0) {
                s_q0[se] = q0;
                s_q1[se] = q1;
                s_op[se] = q2.x;
            }
        }
        __syncthreads();
        if (COUNT) n_staged += (tid == 0) ? (unsigned long long)cnt : 0ull;
        unsigned long long alive_mask = __ballot(alive);
        if (alive_mask != 0ull || COUNT) {
            for (int e = 0; e < cnt; e++) {
                const unsigned long long wm = wave_uniform_u64(s_mask[e * 4 + w]);
                if (COUNT) n_vis += (wm >> lane) & 1ull;
                if ((wm & alive_mask) == 0ull) continue;
                if (COUNT) n_exec += (lane == 0);
                if (((wm >> lane) & 1ull) && alive) { // main.cpp:511-521
                    const float4 q0 = s_q0[e], q1 = s_q1[e];
                    float vx, vy;
                    const float G = gauss_at(px, py, q0.x, q0.y, q0.z, q0.w, q1.x, &vx, &vy);
                    const float alpha = G * s_op[e];        // main.cpp:527
                    cr += T * q1.y * alpha;                  // main.cpp:529-531
                    cg += T * q1.z * alpha;
                    cb += T * q1.w * alpha;
                    T *= (1.0f - alpha);                     // main.cpp:533
                    alive = !(T < kMinThroughput);           // main.cpp:520, evaluated for the next splat
                    if (COUNT) n_act++;
                }
                alive_mask = __ballot(alive);
            }
        }
        if (!__syncthreads_or(alive ? 1 : 0)) break;
    }
    if (inside) image0[(size_t)y * g.W + x] = make_float4(cr, cg, cb, 1.0f); // .w reset, main.cpp:543-546
    if (COUNT) {
        atomicAdd(&counters->fwd_visited, n_vis);
        atomicAdd(&counters->fwd_active, n_act);
        if (tid == 0) atomicAdd(&counters->fwd_staged, n_staged);
        if (lane == 0) atomicAdd(&counters->fwd_wave_execs, n_exec);
    }
}

// ---------------------------------------------------------------------------------------------------
// Wave-wide sums of nine values by DPP: afterwards lane 63 holds the nine totals.
// Six steps per value: xor-1 and xor-2 inside each quad, half-mirror and mirror inside each row of 16
// lanes (every lane of a row then holds the row's sum), row_bcast15 into rows 1 and 3, row_bcast31 into
// rows 2 and 3.  Written as ONE asm block in step-major order: hipcc packs the source-level adds into
// v_pk_add_f32, which cannot carry a DPP modifier (3 instructions per step instead of 1), and in this order
// every register is read again only nine instructions after it was written, so the DPP read-after-VALU-write
// wait states are covered without padding (the leading s_nop covers the producers of the inputs).
// All 64 lanes are active here (wave-uniform control flow).
// ---------------------------------------------------------------------------------------------------
#define S2D_DPP9(ctrl)                                  \
    "v_add_f32_dpp %0, %0, %0 " ctrl "\n"               \
    "v_add_f32_dpp %1, %1, %1 " ctrl "\n"               \
    "v_add_f32_dpp %2, %2, %2 " ctrl "\n"               \
    "v_add_f32_dpp %3, %3, %3 " ctrl "\n"               \
    "v_add_f32_dpp %4, %4, %4 " ctrl "\n"               \
    "v_add_f32_dpp %5, %5, %5 " ctrl "\n"               \
    "v_add_f32_dpp %6, %6, %6 " ctrl "\n"               \
    "v_add_f32_dpp %7, %7, %7 " ctrl "\n"               \
    "v_add_f32_dpp %8, %8, %8 " ctrl "\n"

__device__ __forceinline__ void wave_sum9_to_lane63(float& a0, float& a1, float& a2, float& a3, float& a4, float& a5,
                                                    float& a6, float& a7, float& a8)
{
    asm volatile("s_nop 1\n"
                 S2D_DPP9("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                 S2D_DPP9("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                 S2D_DPP9("row_half_mirror row_mask:0xf bank_mask:0xf")
                 S2D_DPP9("row_mirror row_mask:0xf bank_mask:0xf")
                 S2D_DPP9("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 S2D_DPP9("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 "s_nop 1\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(a8));
}

// ---------------------------------------------------------------------------------------------------
// backward, main.cpp:548-712, + the squared error of main.cpp:796-805
// ---------------------------------------------------------------------------------------------------
template <bool COUNT>
__global__ __launch_bounds__(256) void raster_backward_kernel(const uint32_t* __restrict__ tile_off,
                                                              const uint32_t* __restrict__ list,
                                                              const ProjRec* __restrict__ proj,
                                                              const float4* __restrict__ image0,
                                                              const float4* __restrict__ image_ref,
                                                              float* __restrict__ grads,
                                                              double* __restrict__ tile_sqerr, Geometry g,
                                                              PairCounters* __restrict__ counters)
{
    __shared__ float4 s_q0[B];
    __shared__ float4 s_q1[B];
    __shared__ float4 s_e0[B]; // cc, 2sc, ss, 1/sx^3
    __shared__ float4 s_e1[B]; // 1/sy^3, (sx^2-sy^2)/(sx^2 sy^2), cc-ss, sc
    __shared__ float s_op[B];
    __shared__ unsigned long long s_mask[B * 4];
    __shared__ uint32_t s_idx[2][B];
    // per-wave partial gradients of the batch: written once per (wave, entry) by lane 63 with plain stores,
    // summed over the 4 waves in a fixed order by the flush.  12 floats per slot keep float4 stores aligned.
    __shared__ float4 s_part[4][B][3];
    __shared__ unsigned long long s_touched[4]; // bit e: wave w wrote slot e in this batch
    __shared__ double s_red[4];

    const int tile = tile_of_block(blockIdx.x, g.num_tiles);
    if (tile < 0) return;
    const int tx = tile % g.tiles_x;
    const int ty = tile / g.tiles_x + g.trow0;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int x = tx * kTile + (tid & 15);
    const int y = ty * kTile + (tid >> 4);
    const bool inside = x < g.W && y < g.row_end;
    const float px = (float)x + 0.5f, py = (float)y + 0.5f;

    float4 fin = make_float4(0.f, 0.f, 0.f, 0.f), ref = make_float4(0.f, 0.f, 0.f, 0.f);
    if (inside) {
        fin = image0[(size_t)y * g.W + x];    // finalColor, main.cpp:613
        ref = image_ref[(size_t)y * g.W + x];
    }
    const float dLr = fin.x - ref.x, dLg = fin.y - ref.y, dLb = fin.z - ref.z; // dL_dC, main.cpp:616

    // squared error of this tile (main.cpp:801-802): float per pixel, double across pixels
    {
        const float ex = dLr * 255.0f, ey = dLg * 255.0f, ez = dLb * 255.0f;
        double e2 = inside ? (double)(ex * ex + ey * ey + ez * ez) : 0.0;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) e2 += __shfl_down(e2, d, 64);
        if (lane == 0) s_red[w] = e2;
    }
    __syncthreads();
    if (tid == 0) tile_sqerr[tile] = ((s_red[0] + s_red[1]) + s_red[2]) + s_red[3];

    float cr = 0.0f, cg = 0.0f, cb = 0.0f, T = 1.0f; // image1 = (0,0,0,1), main.cpp:549
    bool alive = inside;
    unsigned long long n_vis = 0, n_act = 0, n_staged = 0, n_exec = 0;

    const uint32_t beg = tile_off[tile], end = tile_off[tile + 1];
    const int se = tid >> 2, sub = tid & 3;
    int par = 0;
    for (uint32_t base = beg; base < end; base += B, par ^= 1) {
        const int cnt = (int)min((uint32_t)B, end - base);
        if (se < cnt) {
            const uint32_t idx = list[base + se];
            const ProjRec* r = proj + idx;
            const float4 q0 = r->q0, q1 = r->q1, q2 = r->q2;
            s_mask[se * 4 + sub] = wave_mask_of(q0, q1, __float_as_int(q2.y), __float_as_int(q2.z),
                                                ty * kTile + sub * 4, tx * kTile, g.W, g.row_end);
            if (sub == 0) {
                const float4 q3 = r->q3;
                const float cosT = q2.w, sinT = q3.x, sx = q3.y, sy = q3.z;
                const float cc = cosT * cosT, ss = sinT * sinT, sc = sinT * cosT;
                const float sx2 = sx * sx, sy2 = sy * sy;
                s_q0[se] = q0;
                s_q1[se] = q1;
                s_op[se] = q2.x;
                s_e0[se] = make_float4(cc, 2.0f * sinT * cosT, ss, 1.0f / (sx2 * sx));
                s_e1[se] = make_float4(1.0f / (sy2 * sy), (sx2 - sy2) / (sx2 * sy * sy), cc - ss, sc);
                s_idx[par][se] = idx;
            }
        }
        __syncthreads();
        if (COUNT) n_staged += (tid == 0) ? (unsigned long long)cnt : 0ull;
        unsigned long long alive_mask = __ballot(alive);
        unsigned long long touched = 0ull;
        if (alive_mask != 0ull || COUNT) {
            for (int e = 0; e < cnt; e++) {
                const unsigned long long wm = wave_uniform_u64(s_mask[e * 4 + w]);
                if (COUNT) n_vis += (wm >> lane) & 1ull;
                if ((wm & alive_mask) == 0ull) continue;
                touched |= 1ull << e;
                if (COUNT) n_exec += (lane == 0);
                float g_px = 0.f, g_py = 0.f, g_sx = 0.f, g_sy = 0.f, g_rot = 0.f;
                float g_r = 0.f, g_g = 0.f, g_b = 0.f, g_op = 0.f;
                if (((wm >> lane) & 1ull) && alive) { // main.cpp:595-605
                    const float4 q0 = s_q0[e], q1 = s_q1[e];
                    const float4 e0 = s_e0[e], e1 = s_e1[e];
                    const float a = q0.z, b = q0.w, d = q1.x;
                    float vx, vy;
                    const float G = gauss_at(px, py, q0.x, q0.y, a, b, d, &vx, &vy); // main.cpp:607-610
                    const float alpha = G * s_op[e];                                   // main.cpp:611
                    const float dC_dc = alpha * T;                                     // main.cpp:618
                    g_r = dLr * dC_dc;
                    g_g = dLg * dC_dc;
                    g_b = dLb * dC_dc;
                    cr += T * q1.y * alpha;                                            // main.cpp:623-625
                    cg += T * q1.z * alpha;
                    cb += T * q1.w * alpha;
                    // c*T - S/(1-alpha) cancels to a small remainder (T_final-sized) out of T-sized terms, so
                    // the quotient is taken exactly as the reference takes it (IEEE division), main.cpp:627-628
                    const float den = 1.0f - alpha + 1.0e-15f;
                    const float dCa_r = q1.y * T - (fin.x - cr) / den;                // S = final - colour
                    const float dCa_g = q1.z * T - (fin.y - cg) / den;
                    const float dCa_b = q1.w * T - (fin.z - cb) / den;
                    const float gs = (dLr * dCa_r + dLg * dCa_g) + dLb * dCa_b;       // dL_dalpha_rgb, :629-630
                    const float bc = b + b;                                            // b + c, :639-640
                    g_px = gs * (0.5f * alpha * (2.0f * a * vx + bc * vy));
                    g_py = gs * (0.5f * alpha * (2.0f * d * vy + bc * vx));
                    const float vxx = vx * vx, vxy = vx * vy, vyy = vy * vy;
                    g_sx = gs * (alpha * e0.w * ((e0.x * vxx + e0.y * vxy) + e0.z * vyy)); // :657-659
                    g_sy = gs * (alpha * e1.x * ((e0.z * vxx - e0.y * vxy) + e0.x * vyy)); // :660-662
                    g_rot = gs * (alpha * e1.y * (e1.z * vx * vy - e1.w * (vxx - vyy)));   // :680-685
                    g_op = gs * G;                                                     // main.cpp:703-704
                    T *= (1.0f - alpha);                                               // main.cpp:707
                    alive = !(T < kMinThroughput);
                    if (COUNT) n_act++;
                }
                wave_sum9_to_lane63(g_px, g_py, g_sx, g_sy, g_rot, g_r, g_g, g_b, g_op);
                if (lane == 63) { // order of the record: pos.xy, sx, sy, rot, color.rgb, opacity (main.cpp:85-93)
                    s_part[w][e][0] = make_float4(g_px, g_py, g_sx, g_sy);
                    s_part[w][e][1] = make_float4(g_rot, g_r, g_g, g_b);
                    s_part[w][e][2].x = g_op;
                }
                alive_mask = __ballot(alive);
            }
        }
        if (lane == 0) s_touched[w] = touched;
        const int any = __syncthreads_or(alive ? 1 : 0);
        // one float-atomic burst per (tile, splat): 9 consecutive floats of grads[idx]
        for (int i = tid; i < cnt * 9; i += 256) {
            const int e = i / 9, k = i - e * 9;
            float v = 0.0f;
            bool any_w = false;
#pragma unroll
            for (int ww = 0; ww < 4; ww++)
                if ((s_touched[ww] >> e) & 1ull) {
                    v += reinterpret_cast<const float*>(&s_part[ww][e][0])[k];
                    any_w = true;
                }
            if (any_w && v != 0.0f) atomicAdd(grads + (size_t)s_idx[par][e] * 9 + k, v);
        }
        if (!any) break;
    }
    if (COUNT) {
        atomicAdd(&counters->bwd_visited, n_vis);
        atomicAdd(&counters->bwd_active, n_act);
        if (tid == 0) atomicAdd(&counters->bwd_staged, n_staged);
        if (lane == 0) atomicAdd(&counters->bwd_wave_execs, n_exec);
    }
}

// One block, fixed summation order (deterministic MSE trace): 1024 threads, 8 loads in flight per thread.
__global__ __launch_bounds__(1024) void sqerr_finalize_kernel(const double* __restrict__ tile_sqerr, int num_tiles,
                                                              double* __restrict__ out)
{
    __shared__ double s[1024];
    double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int base = threadIdx.x; base < num_tiles; base += 8 * 1024) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int i = base + j * 1024;
            if (i < num_tiles) a[j] += tile_sqerr[i];
        }
    }
    s[threadIdx.x] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    __syncthreads();
    for (int d = 512; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) s[threadIdx.x] += s[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = s[0];
}

static inline unsigned raster_grid(int num_tiles) { return (unsigned)(((num_tiles + 7) / 8) * 8); }

hipError_t launch_raster_forward(const uint32_t* tile_off, const uint32_t* list, const ProjRec* proj,
                                 float4* image0, Geometry g, PairCounters* counters, hipStream_t stream)
{
    if (g.num_tiles <= 0) return hipSuccess;
    if (counters)
        hipLaunchKernelGGL(raster_forward_kernel<true>, dim3(raster_grid(g.num_tiles)), dim3(256), 0, stream, tile_off,
                           list, proj, image0, g, counters);
    else
        hipLaunchKernelGGL(raster_forward_kernel<false>, dim3(raster_grid(g.num_tiles)), dim3(256), 0, stream,
                           tile_off, list, proj, image0, g, counters);
    return hipGetLastError();
}

hipError_t launch_raster_backward(const uint32_t* tile_off, const uint32_t* list, const ProjRec* proj,
                                  const float4* image0, const float4* image_ref, float* grads,
                                  double* tile_sqerr, Geometry g, PairCounters* counters, hipStream_t stream)
{
    if (g.num_tiles <= 0) return hipSuccess;
    if (counters)
        hipLaunchKernelGGL(raster_backward_kernel<true>, dim3(raster_grid(g.num_tiles)), dim3(256), 0, stream,
                           tile_off, list, proj, image0, image_ref, grads, tile_sqerr, g, counters);
    else
        hipLaunchKernelGGL(raster_backward_kernel<false>, dim3(raster_grid(g.num_tiles)), dim3(256), 0, stream,
                           tile_off, list, proj, image0, image_ref, grads, tile_sqerr, g, counters);
    return hipGetLastError();
}

hipError_t launch_sqerr_finalize(const double* tile_sqerr, int num_tiles, double* out, hipStream_t stream)
{
    hipLaunchKernelGGL(sqerr_finalize_kernel, dim3(1), dim3(1024), 0, stream, tile_sqerr, num_tiles, out);
    return hipGetLastError();
}

} // namespace s2d
