// s2d_halo.hip -- device side of slab ownership (SURVEY.md section 8e "neighbour-only exchange").
//
// With row slabs a splat only matters to the ranks whose rows it can reach.  Instead of replicating the optimiser
// state and all-reducing N x 9 gradients, a rank HOLDS a splat while
//     [pos.y - reach - margin, pos.y + reach + margin]  meets its rows,   reach = 3*max(sx, sy) + 2
// (the circle that bounds the reference's y-range, main.cpp:489-491, for any rotation, plus the 1-pixel skirt the
// projection uses).  Holders keep bit-identical copies: each adds the holders' partial gradients in ascending rank
// order (grads_combine_kernel) and applies the same Adam step.  The host (distributed.HaloStep) moves the rows
// between ranks with torch.distributed; these kernels only classify, gather, scatter and combine rows.
#include <algorithm>

#include "s2d_device.h"

namespace s2d {

struct SlabBounds {
    int world;
    int row[33]; // row[q] .. row[q + 1] are the rows of rank q
};

// masks[i] bit q: rank q holds splat i according to the CURRENT parameters; 0 for splats this rank does not hold
// (their local copy is stale).  Every holder computes the same mask from its identical copy.
__global__ __launch_bounds__(256) void halo_masks_kernel(const float* __restrict__ splats, const uint8_t* __restrict__ held,
                                                         int n, SlabBounds b, float margin, uint32_t* __restrict__ masks)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (held && !held[i]) {
        masks[i] = 0u;
        return;
    }
    const float* sp = splats + (size_t)i * 9;
    const float y = sp[1];
    const float reach = 3.0f * fmaxf(sp[2], sp[3]) + 2.0f + margin;
    uint32_t m = 0u;
    for (int q = 0; q < b.world; q++)
        if (y + reach >= (float)b.row[q] && y - reach <= (float)b.row[q + 1]) m |= 1u << q;
    if (m == 0u) { // non-finite parameters: keep the protocol consistent, the finite guard reports the failure
        int q = 0;
        while (q + 1 < b.world && !(y < (float)b.row[q + 1])) q++;
        m = 1u << q;
    }
    masks[i] = m;
}

__global__ __launch_bounds__(256) void halo_commit_kernel(const uint32_t* __restrict__ masks, int n, int rank,
                                                          uint8_t* __restrict__ held, uint32_t* __restrict__ flags)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t h = (masks[i] >> rank) & 1u;
    held[i] = (uint8_t)h;
    flags[i] = h; // scanned into positions of the compact id list
}

// ids[pos[i]] = i for held splats: the ascending list the Adam kernel walks, so that its waves are full
__global__ __launch_bounds__(256) void held_ids_kernel(const uint8_t* __restrict__ held, const uint32_t* __restrict__ pos, int n,
                                                       uint32_t* __restrict__ ids)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && held[i]) ids[pos[i]] = (uint32_t)i;
}

// out[j][0..w) = base[ids[j]][0..w)
__global__ __launch_bounds__(256) void rows_gather_kernel(const float* __restrict__ base, int w, const int* __restrict__ ids,
                                                          int count, int n, float* __restrict__ out)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count * w) return;
    const int j = t / w, k = t - j * w;
    const int i = ids[j];
    out[t] = (i >= 0 && i < n) ? base[(size_t)i * w + k] : 0.0f;
}

// base[ids[j]][0..w) = in[j][0..w)   (ids are distinct)
__global__ __launch_bounds__(256) void rows_scatter_kernel(float* __restrict__ base, int w, const int* __restrict__ ids,
                                                           int count, int n, const float* __restrict__ in)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count * w) return;
    const int j = t / w, k = t - j * w;
    const int i = ids[j];
    if (i >= 0 && i < n) base[(size_t)i * w + k] = in[t];
}

// grads[rows[u]] = sum over the holders q = 0 .. world-1, IN THAT ORDER, of q's partial gradient:
//   src[u][q] == -1: q does not hold the row;  == -2: q is this rank (the partial already in grads);
//   >= 0: row index into recv (what q sent).  The same sequence of additions on every holder => identical bits.
__global__ __launch_bounds__(256) void grads_combine_kernel(float* __restrict__ grads, const int* __restrict__ rows, int n_rows,
                                                            const int* __restrict__ src, int world,
                                                            const float* __restrict__ recv, int n)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_rows * 9) return;
    const int u = t / 9, k = t - u * 9;
    const int i = rows[u];
    if (i < 0 || i >= n) return;
    float acc = 0.0f;
    bool first = true;
    for (int q = 0; q < world; q++) {
        const int s = src[(size_t)u * world + q];
        if (s == -1) continue;
        const float v = (s == -2) ? grads[(size_t)i * 9 + k] : recv[(size_t)s * 9 + k];
        acc = first ? v : acc + v;
        first = false;
    }
    if (!first) grads[(size_t)i * 9 + k] = acc;
}

// Compact copies of the held splats' records (s2d_api.hip "compact held state"): out[h] = base[ids[h]] and back, for
// h < *count (the launch covers n, the upper bound known to the host).
__global__ __launch_bounds__(256) void compact_gather_kernel(const float* __restrict__ base, int w, const uint32_t* __restrict__ ids,
                                                             const uint32_t* __restrict__ count, float* __restrict__ out)
{
    const long long total = (long long)*count * w; // (the host knows only the upper bound n: a fixed grid strides over what there is)
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int h = (int)(t / w), k = (int)(t - (long long)h * w);
        out[t] = base[(size_t)ids[h] * w + k];
    }
}

__global__ __launch_bounds__(256) void compact_scatter_kernel(float* __restrict__ base, int w, const uint32_t* __restrict__ ids,
                                                              const uint32_t* __restrict__ count, const float* __restrict__ in)
{
    const long long total = (long long)*count * w;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int h = (int)(t / w), k = (int)(t - (long long)h * w);
        base[(size_t)ids[h] * w + k] = in[t];
    }
}

hipError_t launch_compact_copy(float* base, int w, const uint32_t* ids, const uint32_t* count_dev, int n, float* compact, bool to_compact,
                               hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    const long long total = (long long)n * w;
    const unsigned blocks = (unsigned)std::min<long long>((total + 255) / 256, 4096); // 16 per CU: enough to stream, few to retire empty
    if (to_compact)
        hipLaunchKernelGGL(compact_gather_kernel, dim3(blocks), dim3(256), 0, stream, base, w, ids, count_dev, compact);
    else
        hipLaunchKernelGGL(compact_scatter_kernel, dim3(blocks), dim3(256), 0, stream, base, w, ids, count_dev, compact);
    return hipGetLastError();
}

hipError_t launch_halo_masks(const float* splats, const uint8_t* held, int n, int world, const int* row_bounds, float margin,
                             uint32_t* masks, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    SlabBounds b;
    b.world = world;
    for (int q = 0; q <= world; q++) b.row[q] = row_bounds[q];
    hipLaunchKernelGGL(halo_masks_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, splats, held, n, b, margin, masks);
    return hipGetLastError();
}

hipError_t launch_halo_commit(const uint32_t* masks, int n, int rank, uint8_t* held, uint32_t* ids, uint32_t* count_dev,
                              uint32_t* scan_work, uint32_t* scan_temp, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(halo_commit_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, masks, n, rank, held, scan_work);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = exclusive_scan_u32(scan_work, scan_work, n, scan_temp, count_dev, stream); // count_dev = number of held splats
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(held_ids_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, held, scan_work, n, ids);
    return hipGetLastError();
}

hipError_t launch_rows_gather(const float* base, int w, const int* ids, int count, int n, float* out, hipStream_t stream)
{
    if (count <= 0) return hipSuccess;
    const long long total = (long long)count * w;
    hipLaunchKernelGGL(rows_gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, base, w, ids, count, n, out);
    return hipGetLastError();
}

hipError_t launch_rows_scatter(float* base, int w, const int* ids, int count, int n, const float* in, hipStream_t stream)
{
    if (count <= 0) return hipSuccess;
    const long long total = (long long)count * w;
    hipLaunchKernelGGL(rows_scatter_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, base, w, ids, count, n, in);
    return hipGetLastError();
}

hipError_t launch_grads_combine(float* grads, const int* rows, int n_rows, const int* src, int world, const float* recv,
                                int n, hipStream_t stream)
{
    if (n_rows <= 0) return hipSuccess;
    const long long total = (long long)n_rows * 9;
    hipLaunchKernelGGL(grads_combine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, grads, rows, n_rows,
                       src, world, recv, n);
    return hipGetLastError();
}

} // namespace s2d
