// s2d_binning.hip -- projection of the splats and construction of the per-tile lists.
//
// project_kernel   1 thread / splat: covariance, inverse, row range (main.cpp:423-436, 489-491 ==
//                  556-575) -> 64-byte ProjRec; conservative tile rectangle -> TileRect + pair count.
// emit_kernel      256 splats / block: writes their (tile, splat) pairs at the scanned offsets, in splat
//                  order, so the stable sort by tile leaves every tile's list in index order.
// tile_offsets     boundaries of the sorted key array -> tile_off[0..tiles].
#include "s2d_device.h"

namespace s2d {

__global__ __launch_bounds__(256) void project_kernel(const float* __restrict__ splats,
                                                      const uint8_t* __restrict__ held, int n, Geometry g,
                                                      float margin, int mode, ProjRec* __restrict__ proj,
                                                      TileRect* __restrict__ rects, uint32_t* __restrict__ counts,
                                                      uint32_t* __restrict__ row_counts,
                                                      DeviceStatus* __restrict__ status, int check_stamp,
                                                      int* __restrict__ host_stamp)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (held && !held[i]) { // slab ownership: another rank's splat, the local copy is stale -- it has no footprint here
        if (mode == 0) {
            TileRect e;
            e.tx0 = 1; e.tx1 = 0; e.ty0 = 1; e.ty1 = 0;
            rects[i] = e;
            counts[i] = 0u;
            if (row_counts) row_counts[i] = 0u;
        }
        return;
    }
    const float* sp = splats + (size_t)i * 9;
    Splat s;
    s.pos_x = sp[0]; s.pos_y = sp[1]; s.sx = sp[2]; s.sy = sp[3]; s.rot = sp[4];
    s.col_r = sp[5]; s.col_g = sp[6]; s.col_b = sp[7]; s.opacity = sp[8];
    const Projected p = project(s);
    proj[i] = pack_proj(p);

    TileRect r;
    r.tx0 = 1; r.tx1 = 0; r.ty0 = 1; r.ty1 = 0;
    if (mode == 0) {
        uint32_t cnt = 0;
        // Footprints of anisotropic splats change fastest (Adam turns `rot` by up to ~0.05 rad per step, which
        // moves a bounding box by ~3*|sx - sy|*0.05 px): give them a proportionally wider margin.
        const float m_eff = margin > 0.0f ? margin + kAnisoMargin * fabsf(p.sx - p.sy) : 0.0f;
        if (tile_rect_of(p, g, m_eff, &r)) cnt = (uint32_t)(r.tx1 - r.tx0 + 1) * (uint32_t)(r.ty1 - r.ty0 + 1);
        else { r.tx0 = 1; r.tx1 = 0; r.ty0 = 1; r.ty1 = 0; }
        rects[i] = r;
        counts[i] = cnt;
        if (row_counts) row_counts[i] = cnt ? (uint32_t)(r.ty1 - r.ty0 + 1) : 0u; // tile rows it covers (s2d_tilelists.hip)
    } else {
        if (!rect_still_covers(p, g, rects[i])) raise_rebin(status, check_stamp, host_stamp);
    }
}

// Pairs of 256 consecutive splats per block, written in output order: the block's pairs occupy the contiguous range
// [offsets[first], offsets[first] + total); thread t takes positions t, t + 256, ... of it, finds the owning splat by a
// binary search over the block's offsets (LDS) and the tile from the position inside that splat's rectangle (row-major,
// the order the stable sort then keeps).  Consecutive lanes write consecutive words.  (One thread per splat writing its
// ~20 pairs one after the other made every store instruction touch 64 different cache lines: 180 us at 20 M pairs.)
// ROWS: one entry per (splat, tile row) instead of one per (splat, tile): counts = rows per splat, key = the row in the
// low kTlRowBits bits with the rectangle's column range above it (bits 12..20 tx0, 21..29 tx1: the column pass of
// s2d_tilelists.hip then reads its entries' ranges beside their splat indices instead of gathering rectangles).
template <bool ROWS>
__global__ __launch_bounds__(256) void emit_kernel(const TileRect* __restrict__ rects,
                                                   const uint32_t* __restrict__ offsets,
                                                   const uint32_t* __restrict__ counts, int n, int tiles_x,
                                                   uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                   uint32_t capacity)
{
    __shared__ uint32_t s_off[257]; // offsets of the block's splats relative to the block's first pair
    __shared__ TileRect s_rect[256];
    const int first = blockIdx.x * 256, t = threadIdx.x, i = first + t;
    const uint32_t base = offsets[first];
    uint32_t my_off = 0u, my_cnt = 0u;
    if (i < n) {
        my_off = offsets[i] - base;
        my_cnt = counts[i];
        s_rect[t] = rects[i];
    }
    s_off[t] = (i < n) ? my_off : 0xFFFFFFFFu;
    const int last = min(256, n - first) - 1;
    if (t == last) s_off[256] = my_off + my_cnt; // total pairs of the block (later entries stay 0xFFFFFFFF)
    __syncthreads();
    const uint32_t total = s_off[256];
    for (uint32_t q = t; q < total; q += 256) {
        // owner: the last splat k of the block with s_off[k] <= q (splats without pairs share an offset with their
        // successor and are skipped by taking the LAST such k that has pairs: its range [s_off[k], s_off[k+1]) holds q)
        int lo = 0, hi = last;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (s_off[mid] <= q) lo = mid; else hi = mid - 1;
        }
        const TileRect r = s_rect[lo];
        const uint32_t local = q - s_off[lo], w = (uint32_t)(r.tx1 - r.tx0 + 1);
        const uint32_t o = base + q;
        if (o < capacity) {
            if (ROWS) {
                keys[o] = (r.ty0 + local) | ((uint32_t)r.tx0 << kTlRowBits) | ((uint32_t)r.tx1 << (kTlRowBits + 9));
            } else {
                const uint32_t ty = r.ty0 + local / w, tx = r.tx0 + local % w;
                keys[o] = ty * (uint32_t)tiles_x + tx;
            }
            vals[o] = (uint32_t)(first + lo);
        }
    }
}

// tile_off[t] = first position p with sorted_keys[p] >= t; tile_off[num_tiles] = num_pairs.
__global__ __launch_bounds__(256) void tile_offsets_kernel(const uint32_t* __restrict__ sorted_keys,
                                                           uint32_t num_pairs, int num_tiles,
                                                           uint32_t* __restrict__ tile_off, uint32_t key_mask)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p > num_pairs) return;
    const int cur = (p < num_pairs) ? (int)(sorted_keys[p] & key_mask) : num_tiles;
    const int prev = (p > 0) ? (int)(sorted_keys[p - 1] & key_mask) : -1;
    for (int t = prev + 1; t <= cur; t++) tile_off[t] = p;
}

hipError_t launch_project(const float* splats, const uint8_t* held, int n, Geometry g, float margin, int mode, ProjRec* proj,
                          TileRect* rects, uint32_t* counts, uint32_t* row_counts, DeviceStatus* status, int check_stamp,
                          int* host_stamp, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(project_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, splats, held, n, g, margin, mode, proj,
                       rects, counts, row_counts, status, check_stamp, host_stamp);
    return hipGetLastError();
}

hipError_t launch_emit_pairs(const TileRect* rects, const uint32_t* offsets, const uint32_t* counts, int n, Geometry g,
                             uint32_t* keys, uint32_t* vals, uint32_t capacity, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(emit_kernel<false>, dim3((n + 255) / 256), dim3(256), 0, stream, rects, offsets, counts, n, g.tiles_x, keys,
                       vals, capacity);
    return hipGetLastError();
}

hipError_t launch_emit_row_entries(const TileRect* rects, const uint32_t* row_offsets, const uint32_t* row_counts, int n,
                                   uint32_t* keys, uint32_t* vals, uint32_t capacity, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(emit_kernel<true>, dim3((n + 255) / 256), dim3(256), 0, stream, rects, row_offsets, row_counts, n, 1, keys,
                       vals, capacity);
    return hipGetLastError();
}

hipError_t launch_tile_offsets(const uint32_t* sorted_keys, uint32_t num_pairs, int num_tiles,
                               uint32_t* tile_off, hipStream_t stream, uint32_t key_mask)
{
    const uint32_t threads = num_pairs + 1;
    hipLaunchKernelGGL(tile_offsets_kernel, dim3((threads + 255) / 256), dim3(256), 0, stream, sorted_keys, num_pairs,
                       num_tiles, tile_off, key_mask);
    return hipGetLastError();
}

} // namespace s2d
