// s2d_scan_sort.hip -- exclusive prefix sum and stable LSD radix sort of (tile, splat) pairs.
//
// Why a sort exists at all: the reference blends splats in INDEX order (main.cpp:419, :552;
// Form.pdf p.2 "order is pre-defined"), so every tile's list must be ascending in splat index.
// Pairs are emitted in splat order, so a STABLE sort by tile id alone yields lists that are
// already index-ordered: ceil(log2(tiles)/8) passes of an 8-bit digit (2 passes at 4096x4096).
//
// All integer work, HBM-bound: coalesced 4-byte streams in, scattered 4-byte stores out.
#include "s2d_device.h"

namespace s2d {

// ---------------------------------------------------------------------------------------------------
// block-level helpers (256 threads = 4 wave64)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_inclusive_scan_u32(uint32_t v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// Exclusive prefix of v over the block's 256 threads; *total = block sum.  s_wave: 4 words of LDS.
__device__ __forceinline__ uint32_t block_exclusive_scan_u32(uint32_t v, uint32_t* total, uint32_t* s_wave)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = wave_inclusive_scan_u32(v);
    if (lane == 63) s_wave[w] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t t = s_wave[i];
        if (i < w) base += t;
        tot += t;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// ---------------------------------------------------------------------------------------------------
// exclusive scan: reduce per block -> scan block sums (one block) -> scan per block + base.
// *total_dev saturates at 0xFFFFFFFF when the sum does not fit 32 bits (the prefixes are then meaningless).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kScanBlock) void scan_reduce_kernel(const uint32_t* __restrict__ in, int64_t n,
                                                                 uint32_t* __restrict__ block_sums)
{
    __shared__ uint32_t s_wave[4];
    const int64_t base = (int64_t)blockIdx.x * kScanItemsPerBlock;
    uint32_t sum = 0;
#pragma unroll
    for (int j = 0; j < kScanItemsPerThread; j++) {
        int64_t i = base + (int64_t)j * kScanBlock + threadIdx.x;
        if (i < n) sum += in[i];
    }
    uint32_t total;
    block_exclusive_scan_u32(sum, &total, s_wave);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(kScanBlock) void scan_top_kernel(uint32_t* __restrict__ block_sums, int64_t nb,
                                                              uint32_t* __restrict__ total_out)
{
    __shared__ uint32_t s_wave[4];
    __shared__ unsigned long long s_wide[kScanBlock / 64];
    uint32_t carry = 0;
    unsigned long long wide = 0; // this thread's share of the grand total, carried in 64 bits
    for (int64_t base = 0; base < nb; base += kScanBlock) {
        int64_t i = base + threadIdx.x;
        uint32_t v = (i < nb) ? block_sums[i] : 0u;
        wide += v;
        uint32_t tot;
        uint32_t ex = block_exclusive_scan_u32(v, &tot, s_wave);
        if (i < nb) block_sums[i] = carry + ex;
        carry += tot;
    }
    // The 32-bit prefix sums wrap silently when the grand total does not fit (tile lists of > 2^32 pairs: a few 10^4
    // splats at the sx/sy clamp on a 4096^2 image).  Report a SATURATED total then, so the caller sees "too many"
    // instead of a small wrapped number.
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) wide += __shfl_down(wide, d, 64);
    if ((threadIdx.x & 63) == 0) s_wide[threadIdx.x >> 6] = wide;
    __syncthreads();
    if (threadIdx.x == 0 && total_out) {
        unsigned long long all = 0;
        for (int k = 0; k < kScanBlock / 64; k++) all += s_wide[k];
        *total_out = all > 0xFFFFFFFFull ? 0xFFFFFFFFu : carry;
    }
}

// in and out may be the same array (each thread reads its items before it writes them)
__global__ __launch_bounds__(kScanBlock) void scan_apply_kernel(const uint32_t* in, uint32_t* out, int64_t n,
                                                                const uint32_t* block_sums)
{
    __shared__ uint32_t s_wave[4];
    // thread t owns the contiguous items [t*8, t*8+8) of the block's 2048
    const int64_t first = (int64_t)blockIdx.x * kScanItemsPerBlock + (int64_t)threadIdx.x * kScanItemsPerThread;
    uint32_t v[kScanItemsPerThread];
    uint32_t sum = 0;
#pragma unroll
    for (int j = 0; j < kScanItemsPerThread; j++) {
        int64_t i = first + j;
        v[j] = (i < n) ? in[i] : 0u;
        sum += v[j];
    }
    uint32_t total;
    uint32_t run = block_exclusive_scan_u32(sum, &total, s_wave) + block_sums[blockIdx.x];
#pragma unroll
    for (int j = 0; j < kScanItemsPerThread; j++) {
        int64_t i = first + j;
        if (i < n) out[i] = run;
        run += v[j];
    }
}

static inline int64_t scan_blocks(int64_t n) { return (n + kScanItemsPerBlock - 1) / kScanItemsPerBlock; }

size_t scan_temp_words(int64_t n) { return (size_t)scan_blocks(n) + 4; }

hipError_t exclusive_scan_u32(const uint32_t* in, uint32_t* out, int64_t n, uint32_t* temp, uint32_t* total_dev,
                              hipStream_t stream)
{
    if (n <= 0) {
        if (total_dev) return hipMemsetAsync(total_dev, 0, sizeof(uint32_t), stream);
        return hipSuccess;
    }
    const int64_t nb = scan_blocks(n);
    hipLaunchKernelGGL(scan_reduce_kernel, dim3((unsigned)nb), dim3(kScanBlock), 0, stream, in, n, temp);
    hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(kScanBlock), 0, stream, temp, nb, total_dev);
    hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)nb), dim3(kScanBlock), 0, stream, in, out, n, temp);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// LSD radix sort pass, 8-bit digit.
// Block b owns the contiguous items [b*4096, (b+1)*4096); wave w of the block owns the contiguous
// quarter of that, which it walks in 16 rounds of 64 consecutive items, so memory order ==
// (block, wave, round, lane) order and a rank computed in that order is stable.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kSortBlock) void radix_hist_kernel(const uint32_t* __restrict__ keys, int64_t n,
                                                                int shift, uint32_t* __restrict__ ghist, int nblk)
{
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kSortItemsPerBlock;
#pragma unroll
    for (int j = 0; j < kSortItemsPerThread; j++) {
        int64_t i = base + (int64_t)j * kSortBlock + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    ghist[(size_t)threadIdx.x * nblk + blockIdx.x] = h[threadIdx.x];
}

__global__ __launch_bounds__(kSortBlock) void radix_scatter_kernel(const uint32_t* __restrict__ keys_in,
                                                                   const uint32_t* __restrict__ vals_in,
                                                                   uint32_t* __restrict__ keys_out,
                                                                   uint32_t* __restrict__ vals_out, int64_t n,
                                                                   int shift, const uint32_t* __restrict__ ghist_scanned,
                                                                   int nblk)
{
    __shared__ uint32_t s_cnt[4 * 256]; // per-wave running digit counts, then per-wave exclusive bases
    __shared__ uint32_t s_gbase[256];   // global start of (digit, this block)
    __shared__ uint32_t s_boff[256];    // start of the digit's run in the block's sorted order
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_key[kSortItemsPerBlock], s_val[kSortItemsPerBlock];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < 4 * 256; i += kSortBlock) s_cnt[i] = 0;
    s_gbase[tid] = ghist_scanned[(size_t)tid * nblk + blockIdx.x];
    __syncthreads();

    const int64_t wave_base = (int64_t)blockIdx.x * kSortItemsPerBlock + (int64_t)w * (kSortItemsPerBlock / 4);
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    uint32_t key[kSortItemsPerThread], val[kSortItemsPerThread], rank[kSortItemsPerThread];

#pragma unroll
    for (int j = 0; j < kSortItemsPerThread; j++) {
        const int64_t i = wave_base + (int64_t)j * 64 + lane;
        const bool valid = i < n;
        key[j] = valid ? keys_in[i] : 0u;
        val[j] = valid ? vals_in[i] : 0u;
        const uint32_t digit = (key[j] >> shift) & 255u;
        // lanes of this wave holding the same digit
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const bool bit = (digit >> b) & 1u;
            const uint64_t m = __ballot(valid && bit);
            peers &= bit ? m : ~m;
        }
        const uint32_t before = (uint32_t)__popcll(peers & lt_mask);
        uint32_t prior = 0;
        if (valid) prior = s_cnt[w * 256 + digit];
        // the read above is issued for all lanes before this store (same wave, LDS is in order)
        if (valid && before == 0) s_cnt[w * 256 + digit] = prior + (uint32_t)__popcll(peers);
        rank[j] = prior + before;
    }
    __syncthreads();
    // per digit: exclusive prefix over the 4 waves, and the block's total
    uint32_t digit_total;
    {
        uint32_t run = 0;
#pragma unroll
        for (int ww = 0; ww < 4; ww++) {
            uint32_t t = s_cnt[ww * 256 + tid];
            s_cnt[ww * 256 + tid] = run;
            run += t;
        }
        digit_total = run;
    }
    // where each digit's run starts inside the block's sorted order
    {
        uint32_t unused;
        s_boff[tid] = block_exclusive_scan_u32(digit_total, &unused, s_wave);
    }
    __syncthreads();
    // Sort the block's items in LDS first (stable: rank order is memory order), then write them out in sorted order:
    // consecutive threads then store consecutive words of a digit's run instead of 64 unrelated words per instruction.
#pragma unroll
    for (int j = 0; j < kSortItemsPerThread; j++) {
        const int64_t i = wave_base + (int64_t)j * 64 + lane;
        if (i < n) {
            const uint32_t digit = (key[j] >> shift) & 255u;
            const uint32_t p = s_boff[digit] + s_cnt[w * 256 + digit] + rank[j];
            s_key[p] = key[j];
            s_val[p] = val[j];
        }
    }
    __syncthreads();
    const int64_t block_base = (int64_t)blockIdx.x * kSortItemsPerBlock;
    const uint32_t count = (uint32_t)min((int64_t)kSortItemsPerBlock, n - block_base);
    for (uint32_t p = tid; p < count; p += kSortBlock) {
        const uint32_t k = s_key[p];
        const uint32_t digit = (k >> shift) & 255u;
        const uint32_t dest = s_gbase[digit] + (p - s_boff[digit]);
        keys_out[dest] = k;
        vals_out[dest] = s_val[p];
    }
}

static inline int64_t sort_blocks(int64_t n) { return (n + kSortItemsPerBlock - 1) / kSortItemsPerBlock; }

size_t sort_temp_words(int64_t n)
{
    const int64_t nblk = sort_blocks(n);
    return (size_t)(256 * nblk) + scan_temp_words(256 * nblk) + 4;
}

hipError_t sort_pairs_u32(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, int64_t n,
                          int key_bits, uint32_t* temp, uint32_t** keys_out, uint32_t** vals_out,
                          hipStream_t stream)
{
    uint32_t *kin = keys_a, *vin = vals_a, *kout = keys_b, *vout = vals_b;
    if (n > 0) {
        const int64_t nblk = sort_blocks(n);
        uint32_t* ghist = temp;
        uint32_t* scan_temp = temp + 256 * nblk;
        for (int shift = 0; shift < key_bits; shift += 8) {
            hipLaunchKernelGGL(radix_hist_kernel, dim3((unsigned)nblk), dim3(kSortBlock), 0, stream, kin, n, shift,
                               ghist, (int)nblk);
            hipError_t e = exclusive_scan_u32(ghist, ghist, 256 * nblk, scan_temp, nullptr, stream);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(radix_scatter_kernel, dim3((unsigned)nblk), dim3(kSortBlock), 0, stream, kin, vin,
                               kout, vout, n, shift, ghist, (int)nblk);
            uint32_t* t;
            t = kin; kin = kout; kout = t;
            t = vin; vin = vout; vout = t;
        }
    }
    *keys_out = kin;
    *vals_out = vin;
    return hipGetLastError();
}

} // namespace s2d
