// s2d_device.h -- internal declarations shared by the HIP translation units.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "s2d_math.h"

namespace s2d {

// ---- data layout in HBM -----------------------------------------------------------------------
// Projected splat: what the raster kernels need per list entry, one 64-byte record (4 x float4)
// so that a list entry is staged with four 16-byte loads from one cache line.
//   q0 = (pos_x, pos_y, a, b)          a,b,d = inverse covariance (b == c bit for bit)
//   q1 = (d, col_r, col_g, col_b)
//   q2 = (opacity, bits(begY), bits(endY), cosT)
//   q3 = (sinT, sx, sy, hx)
struct alignas(16) ProjRec {
    float4 q0, q1, q2, q3;
};
static_assert(sizeof(ProjRec) == 64, "ProjRec must be 64 bytes");

// Inclusive tile rectangle of a splat, in tile units LOCAL to the context's slab; empty when n == 0.
struct TileRect {
    uint16_t tx0, tx1, ty0, ty1;
};

struct Geometry {
    int W, H;           // image size
    int row_begin;      // slab [row_begin, row_end) in pixel rows; row_begin % 16 == 0
    int row_end;
    int tiles_x;        // ceil(W / 16)
    int trow0;          // row_begin / 16: first global tile row of the slab
    int tiles_y;        // tile rows in the slab
    int num_tiles;      // tiles_x * tiles_y
};

struct PairCounters {
    unsigned long long fwd_visited, fwd_active, bwd_visited, bwd_active, fwd_staged, bwd_staged, fwd_wave_execs,
        bwd_wave_execs;
    unsigned long long bwd_lane_hist[65]; // executed (wave, entry) pairs by number of active lanes
    unsigned long long fwd_staged_hit;    // staged entries with at least one pixel of the tile inside their ranges
    unsigned long long fwd_rows_hit;      // (staged entry, tile row) pairs with a non-empty column range
    unsigned long long bwd_quadrant_execs; // executed (wave, entry) pairs weighted by their live 4x4 quadrants (1..4)
};

// Device-resident status word(s), written by kernels, read by the host at synchronisation points.
struct DeviceStatus {
    int nonfinite;            // 1 once a checked parameter became non-finite (main.cpp:752-785)
    int first_nonfinite_iter; // iteration at which that first happened (INT_MAX if never)
    int rebin_needed;         // sequence number of the last containment check in which a splat's exact tile
                              // rectangle had left its binned one (a stamp, never cleared: no memset per iteration)
    int pad;
};

// A squared-error reduction riding on another launch (tile_sqerr == nullptr: none): the Adam launch's first
// workgroups (large images), or the raster launch's last tile (images of at most kSqerrSmallTiles tiles).
struct SqerrJob {
    const double* tile_sqerr;
    int num_tiles;
    double* out;
    double* scratch;
};

#if defined(__HIPCC__)
__device__ __forceinline__ float as_f(int v) { return __int_as_float(v); }

// A containment check found a splat outside its binned rectangle: stamp the device word the optimistically launched
// forward kernel compares against, and the host-mapped word the host reads once the checking kernel has completed
// (every writer stores the same value, so plain stores suffice).
__device__ __forceinline__ void raise_rebin(DeviceStatus* status, int stamp, int* host_stamp)
{
    __hip_atomic_store(&status->rebin_needed, stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (host_stamp) __hip_atomic_store(host_stamp, stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ ProjRec pack_proj(const Projected& p)
{
    ProjRec rec;
    rec.q0 = make_float4(p.pos_x, p.pos_y, p.a, p.b);
    rec.q1 = make_float4(p.d, p.col_r, p.col_g, p.col_b);
    rec.q2 = make_float4(p.opacity, as_f(p.begY), as_f(p.endY), p.cosT);
    // what the backward pass needs of the scales, once per splat instead of once per (tile, splat): 1/sx^3, 1/sy^3
    // (main.cpp:657-662, :677-678) and (sx^2 - sy^2)/(sx^2 sy^2) (main.cpp:680-685), in the reference's operation order
    const float sx2 = p.sx * p.sx, sy2 = p.sy * p.sy;
    rec.q3 = make_float4(p.sinT, 1.0f / (sx2 * p.sx), 1.0f / (sy2 * p.sy), (sx2 - sy2) / (sx2 * p.sy * p.sy));
    return rec;
}

// Conservative tile rectangle (local to the slab) of a projected splat, inflated by `margin` pixels.
// Columns: the exact per-row ranges (row_range) lie within pos_x +- hx up to rounding and the
// truncation toward zero of negative values; a 1-pixel skirt covers both.
__device__ __forceinline__ bool tile_rect_of(const Projected& p, const Geometry& g, float margin, TileRect* r)
{
    const int m = (int)margin;
    // rows: [begY, endY] from the reference, clipped to the slab
    long long y0 = (long long)p.begY - m, y1 = (long long)p.endY + m;
    if (p.begY == (int)0x80000000u || p.endY == (int)0x80000000u) return false; // NaN / out of range
    if (y0 < g.row_begin) y0 = g.row_begin;
    if (y1 > g.row_end - 1) y1 = g.row_end - 1;
    if (y0 > y1) return false;
    float xlo = p.pos_x - p.hx - 1.0f - margin;
    float xhi = p.pos_x + p.hx + 1.0f + margin;
    if (!(xlo <= xhi)) return false; // NaN
    if (xhi < 0.0f || xlo > (float)(g.W - 1)) return false;
    xlo = fmaxf(xlo, 0.0f);
    xhi = fminf(xhi, (float)(g.W - 1));
    r->tx0 = (uint16_t)((int)xlo >> 4);
    r->tx1 = (uint16_t)((int)xhi >> 4);
    r->ty0 = (uint16_t)(((int)y0 >> 4) - g.trow0);
    r->ty1 = (uint16_t)(((int)y1 >> 4) - g.trow0);
    return true;
}

// Do the tile lists built from rectangle `binned` still list this splat in every tile it can touch now?
// (its exact rectangle, margin 0, must lie inside the binned one)
__device__ __forceinline__ bool rect_still_covers(const Projected& p, const Geometry& g, const TileRect& binned)
{
    TileRect r;
    if (!tile_rect_of(p, g, 0.0f, &r)) return true; // touches nothing
    return binned.tx0 <= binned.tx1 && r.tx0 >= binned.tx0 && r.tx1 <= binned.tx1 && r.ty0 >= binned.ty0 &&
           r.ty1 <= binned.ty1;
}
#endif

constexpr float kAnisoMargin = 1.0f;   // extra binning margin per pixel of |sx - sy| (list re-use)
constexpr int kRasterBatch = 64;       // list entries staged in LDS per batch
constexpr int kSortItemsPerThread = 16;
constexpr int kSortBlock = 256;
constexpr int kSortItemsPerBlock = kSortBlock * kSortItemsPerThread; // 4096
constexpr int kScanItemsPerThread = 8;
constexpr int kScanBlock = 256;
constexpr int kScanItemsPerBlock = kScanBlock * kScanItemsPerThread; // 2048

// ---- launchers (each queues kernels on `stream`, never synchronises) ------------------------------
// scan / sort (s2d_scan_sort.hip)
size_t scan_temp_words(int64_t n);
// exclusive prefix sum of in[0..n) into out[0..n) (may alias); *total_dev (device) receives the sum, and so does
// *total_host_mapped (host memory mapped into the device's address space, written by the kernel itself: complete once an
// event recorded behind this call has completed).
hipError_t exclusive_scan_u32(const uint32_t* in, uint32_t* out, int64_t n, uint32_t* temp, uint32_t* total_dev,
                              hipStream_t stream, uint32_t* total_host_mapped = nullptr);
size_t sort_temp_words(int64_t n);
// Stable LSD radix sort by the low key_bits of keys.  Result pointers (one of the two buffers each) are
// returned through keys_out / vals_out.
// tile_first != nullptr (1 << key_bits words, all 0xFFFFFFFF on entry): the last pass leaves the sorted KEYS unwritten and
// records tile_first[k] = position of the first pair with key k instead (launch_tile_offsets_from_first turns that into
// the list boundaries): one 4-byte stream less to write and none to read back.  key_bits == 0 sorts nothing: not for that.
hipError_t sort_pairs_u32(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, int64_t n,
                          int key_bits, uint32_t* temp, uint32_t** keys_out, uint32_t** vals_out, uint32_t* tile_first,
                          hipStream_t stream);
// tile_off[t] = first position with key >= t for t in [0, num_keys], from tile_first (see sort_pairs_u32);
// temp: tile_first_temp_words(num_keys) words
size_t tile_first_temp_words(int num_keys);
hipError_t launch_tile_offsets_from_first(const uint32_t* tile_first, int num_keys, uint32_t total, uint32_t* temp,
                                          uint32_t* tile_off, hipStream_t stream);

// binning (s2d_binning.hip)
// Projects every splat; mode 0: also writes rects[] (inflated by `margin` pixels) and counts[];
// mode 1: checks that the exact rectangle lies inside rects[] and raises status->rebin_needed otherwise.
// (mode 0, row_counts != nullptr: also the number of tile rows each rectangle covers)
hipError_t launch_project(const float* splats, const uint8_t* held, int n, Geometry g, float margin, int mode, ProjRec* proj,
                          TileRect* rects, uint32_t* counts, uint32_t* row_counts, DeviceStatus* status, int check_stamp,
                          int* host_stamp, hipStream_t stream);
// one (tile row, splat) entry per row of every splat's rectangle, in splat order, at the scanned row offsets
hipError_t launch_emit_row_entries(const TileRect* rects, const uint32_t* row_offsets, const uint32_t* row_counts, int n,
                                   uint32_t* keys, uint32_t* vals, uint32_t capacity, hipStream_t stream);
hipError_t launch_emit_pairs(const TileRect* rects, const uint32_t* offsets, const uint32_t* counts, int n, Geometry g,
                             uint32_t* keys, uint32_t* vals, uint32_t capacity, hipStream_t stream);
// (key_mask: the bits of a key that count -- row entries carry their column range above the row)
hipError_t launch_tile_offsets(const uint32_t* sorted_keys, uint32_t num_pairs, int num_tiles,
                               uint32_t* tile_off, hipStream_t stream, uint32_t key_mask = 0xFFFFFFFFu);

// tile lists in two levels (s2d_tilelists.hip): from the (splat, tile row) entries sorted by row -- `entries` holds the splat
// indices, row_off[0 .. tiles_y] where each row's entries begin -- to tile_off[0 .. tiles] and the lists themselves.
// Images of up to kTlMaxColumns tile columns.  workspace: tl_workspace_words(...) words; chunk_base: tiles_y + 1 words.
constexpr int kTlMaxColumns = 512;
constexpr int kTlRowBits = 12; // a row entry's key: tile row (H <= 65536) | tx0 << 12 | tx1 << 21 (nine bits each)
size_t tl_workspace_words(uint64_t entries, int tiles_x, int tiles_y);
hipError_t launch_tile_lists_from_rows(const uint32_t* entries, const uint32_t* entry_keys, uint64_t num_entries, const uint32_t* row_off,
                                       Geometry g, uint32_t* chunk_base, uint32_t* workspace, uint32_t* tile_off, uint32_t* list,
                                       hipStream_t stream);

// raster (s2d_raster.hip)
// abort_stamp != 0: when status->rebin_needed equals it at kernel start the launch does nothing -- the lists it would
// walk are stale and the host rebuilds them and launches again.  Every kernel of an iteration does nothing once a
// parameter went non-finite in an earlier iteration (status->first_nonfinite_iter < iteration).
hipError_t launch_raster_forward(const uint32_t* tile_off, const uint32_t* list, const ProjRec* proj, void* image0,
                                 bool half_images, unsigned long long* wave_masks, Geometry g, const DeviceStatus* status,
                                 int abort_stamp, int iteration, PairCounters* counters, bool count, bool exact_exp,
                                 hipStream_t stream);
// Deterministic gradient accumulation (S2D_CFG_DETERMINISTIC): instead of float atomics every tile stores its
// partial gradient of a splat into the slot offsets[splat] + (position of the tile in the splat's emission
// rectangle), stamped with the iteration and announced in the splat's `touched` word; a gather kernel then sums each
// splat's written slots in slot order.
constexpr int kDetStride = 12; // floats per slot: nine gradients, padded so that a slot is three aligned 16-byte words
struct DetGather {
    const TileRect* rects;
    const uint32_t* offsets;
    const uint32_t* counts;
    float* data;      // [pair capacity][kDetStride]
    uint32_t* stamp;  // [pair capacity]
    uint32_t* touched; // [n]: per splat, which of its slots were written in this pass (zero between passes)
    uint32_t now;     // iteration + 1 (never 0: 0 marks a slot that was never written)
    int n;
};
hipError_t launch_raster_backward(const uint32_t* tile_off, const uint32_t* list, const ProjRec* proj,
                                  const void* image0, const void* image_ref, bool half_images,
                                  const unsigned long long* wave_masks, float* grads, double* tile_sqerr, Geometry g,
                                  bool need_opacity_grad, const DetGather* dg, const DeviceStatus* status, int iteration,
                                  PairCounters* counters, bool count, bool exact_exp, hipStream_t stream);
// Forward + backward walk of every tile in one launch (same results as the two launches above).
hipError_t launch_raster_fused(const uint32_t* tile_off, const uint32_t* list, const ProjRec* proj, void* image0,
                               const void* image_ref, bool half_images, unsigned long long* wave_masks, float* grads,
                               double* tile_sqerr, Geometry g, bool need_opacity_grad, const DetGather* dg,
                               const DeviceStatus* status, int abort_stamp, int iteration, bool write_image, bool exact_exp,
                               SqerrJob sq, hipStream_t stream);
// Index-range rendering (scenes beyond one set of lists; s2d_raster.hip "chunked", s2d_api.hip chunked_raster): the lists are
// those of ONE index range of the splats (proj / grads / the DetGather arrays point at the range's first splat, list words
// count from it); `state` carries (r, g, b, T) per pixel from range to range; first: start from (0, 0, 0, 1).
hipError_t launch_raster_forward_chunk(const uint32_t* tile_off, const uint32_t* list, const ProjRec* proj, void* image0,
                                       bool half_images, float4* state, bool first, unsigned long long* wave_masks, Geometry g,
                                       const DeviceStatus* status, int iteration, uint32_t* any_alive, bool exact_exp,
                                       hipStream_t stream);
hipError_t launch_raster_backward_chunk(const uint32_t* tile_off, const uint32_t* list, const ProjRec* proj, const void* image0,
                                        const void* image_ref, bool half_images, float4* state, bool first,
                                        unsigned long long* wave_masks, float* grads, double* tile_sqerr, Geometry g,
                                        bool need_opacity_grad, const DetGather* dg, const DeviceStatus* status, int iteration,
                                        bool exact_exp, hipStream_t stream);
// slab ownership (s2d_halo.hip): `held` == nullptr means every splat is held (single rank, or replicated state)
hipError_t launch_halo_masks(const float* splats, const uint8_t* held, int n, int world, const int* row_bounds, float margin,
                             uint32_t* masks, hipStream_t stream);
// also builds ids[0 .. *count_dev): the held splats in ascending order (scan_work: n words, scan_temp: scan_temp_words(n))
hipError_t launch_halo_commit(const uint32_t* masks, int n, int rank, uint8_t* held, uint32_t* ids, uint32_t* count_dev,
                              uint32_t* scan_work, uint32_t* scan_temp, hipStream_t stream);
// compact[h][0..w) = base[ids[h]][0..w) for h < *count_dev (to_compact), or the other way round
hipError_t launch_compact_copy(float* base, int w, const uint32_t* ids, const uint32_t* count_dev, int n, float* compact, bool to_compact,
                               hipStream_t stream);
hipError_t launch_rows_gather(const float* base, int w, const int* ids, int count, int n, float* out, hipStream_t stream);
hipError_t launch_rows_scatter(float* base, int w, const int* ids, int count, int n, const float* in, hipStream_t stream);
hipError_t launch_grads_combine(float* grads, const int* rows, int n_rows, const int* src, int world, const float* recv,
                                int n, hipStream_t stream);
// scratch: kSqerrScratchDoubles doubles, zero before the first launch
constexpr int kSqerrChunks = 64;
constexpr int kSqerrScratchDoubles = kSqerrChunks + 1;
constexpr int kSqerrSmallTiles = 1024; // up to here ONE workgroup adds the tile errors in one pass (sqerr_sum_small)


#if defined(__HIPCC__)
__device__ __forceinline__ double block_sum_256(double v, double* s)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d, 64);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    const double r = ((s[0] + s[1]) + s[2]) + s[3];
    __syncthreads(); // s may be written again by the caller's next sum
    return r;
}

// Small images: the whole sum by the calling workgroup, one pass, one fixed order (thread t adds tiles t, t + 256,
// t + 512, t + 768, then the block sum).  Every path that sums a small image's tile errors uses this, so the MSE does
// not depend on the path.  The tile errors were written by other workgroups: agent-scope loads.
__device__ __forceinline__ double sqerr_sum_small(const double* tile_sqerr, int num_tiles, double* s4)
{
    double a = 0.0;
    for (int i = (int)threadIdx.x; i < num_tiles; i += 256)
        a += __hip_atomic_load(tile_sqerr + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return block_sum_256(a, s4);
}

// Sum of the per-tile squared errors in a FIXED order (a deterministic MSE trace, main.cpp:796-805), shared between
// workgroups: the tiles are cut into kSqerrChunks contiguous chunks; the calling 256-thread workgroup reduces chunks
// first_chunk, first_chunk + chunk_stride, ... into scratch[chunk]; the workgroup that completes the last chunk (a
// ticket counter behind the partials) adds the kSqerrChunks partials, again in a fixed order, writes *out and re-arms
// the counter.  The result does not depend on how many workgroups share the chunks.  Every thread of the workgroup
// must call this.
__device__ __forceinline__ void sqerr_reduce(const double* __restrict__ tile_sqerr, int num_tiles, double* __restrict__ out,
                                             double* scratch, int first_chunk, int chunk_stride)
{
    __shared__ double s[4];
    __shared__ bool last;
    if (num_tiles <= kSqerrSmallTiles) { // one workgroup does it all (the one that was handed chunk 0)
        if (first_chunk != 0) return;
        const double total = sqerr_sum_small(tile_sqerr, num_tiles, s);
        if (threadIdx.x == 0) *out = total;
        return;
    }
    unsigned long long* ticket = reinterpret_cast<unsigned long long*>(scratch + kSqerrChunks);
    const int chunk = (num_tiles + kSqerrChunks - 1) / kSqerrChunks;
    unsigned long long mine = 0;
    for (int b = first_chunk; b < kSqerrChunks; b += chunk_stride, mine++) {
        const int beg = b * chunk, end = min(beg + chunk, num_tiles);
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        for (int i = beg + (int)threadIdx.x; i < end; i += 1024) {
            a0 += tile_sqerr[i];
            if (i + 256 < end) a1 += tile_sqerr[i + 256];
            if (i + 512 < end) a2 += tile_sqerr[i + 512];
            if (i + 768 < end) a3 += tile_sqerr[i + 768];
        }
        const double part = block_sum_256((a0 + a1) + (a2 + a3), s);
        if (threadIdx.x == 0) __hip_atomic_store(scratch + b, part, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0) {
        __threadfence();
        last = mine != 0 && atomicAdd(ticket, mine) + mine == (unsigned long long)kSqerrChunks;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    const double p = threadIdx.x < kSqerrChunks
                         ? __hip_atomic_load(scratch + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                         : 0.0;
    const double total = block_sum_256(p, s);
    if (threadIdx.x == 0) {
        *out = total;
        *ticket = 0ull;
    }
}
#endif
hipError_t launch_sqerr_finalize(const double* tile_sqerr, int num_tiles, double* out, double* scratch,
                                 const DeviceStatus* status, int iteration, hipStream_t stream);

// optimiser / init (s2d_optim.hip)
hipError_t launch_init_splats(float* splats, float* adams, int n, int W, int H, hipStream_t stream);
// proj != nullptr: also project the UPDATED splat for the next iteration and check it against rects[]
// (what project_kernel mode 1 would do), raising status->rebin_needed.
// dormant (n bytes, or nullptr): dormant[i] = 1 while every Adam moment of splat i is zero -- maintained by the kernel,
// cleared by whoever else writes splats or moments; a block whose splats are all dormant and received zero gradients skips
// the step, which would leave them bit for bit as they are.
// compact (only with held_ids): splats / adams are the COMPACT arrays of the held splats, record h = splat held_ids[h]:
// whole lines instead of one gathered record per splat; gradients, projection and `dormant` stay indexed by splat id.
hipError_t launch_adam(float* splats, float* adams, float* grads, const uint32_t* held_ids, const uint32_t* held_count, int n,
                       Geometry g, float beta1t, float beta2t,
                       float lr, int optimize_opacity, int iteration, DeviceStatus* status, ProjRec* proj,
                       const TileRect* rects, int check_stamp, int* host_stamp, uint8_t* dormant, SqerrJob sq,
                       bool compact, hipStream_t stream);
// image_ref: rows [row_begin, row_end) of the W x H target
hipError_t launch_synthetic_target(void* image_ref, bool half_images, int W, int H, int row_begin, int row_end, hipStream_t stream);
// RGBA32F <-> 4 x fp16 (round to nearest even) for images that cross the boundary as floats
hipError_t launch_convert_f32_to_f16(const float4* src, void* dst, size_t pixels, hipStream_t stream);
hipError_t launch_convert_f16_to_f32(const void* src, float4* dst, size_t pixels, hipStream_t stream);
hipError_t launch_test_sincos(const float* x, int n, float* s, float* c, hipStream_t stream);

} // namespace s2d
