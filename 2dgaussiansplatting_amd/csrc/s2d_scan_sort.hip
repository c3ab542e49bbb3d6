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
                                                              uint32_t* __restrict__ total_out,
                                                              uint32_t* __restrict__ total_host)
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
    if (threadIdx.x == 0 && (total_out || total_host)) {
        unsigned long long all = 0;
        for (int k = 0; k < kScanBlock / 64; k++) all += s_wide[k];
        const uint32_t total = all > 0xFFFFFFFFull ? 0xFFFFFFFFu : carry;
        if (total_out) *total_out = total;
        // host-mapped word: the host reads it behind an event recorded after this kernel, no copy engine in between
        if (total_host) __hip_atomic_store(total_host, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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
                              hipStream_t stream, uint32_t* total_host_mapped)
{
    if (n <= 0) {
        if (total_host_mapped) *total_host_mapped = 0u; // nothing is queued that could write it
        if (total_dev) return hipMemsetAsync(total_dev, 0, sizeof(uint32_t), stream);
        return hipSuccess;
    }
    const int64_t nb = scan_blocks(n);
    hipLaunchKernelGGL(scan_reduce_kernel, dim3((unsigned)nb), dim3(kScanBlock), 0, stream, in, n, temp);
    hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(kScanBlock), 0, stream, temp, nb, total_dev, total_host_mapped);
    hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)nb), dim3(kScanBlock), 0, stream, in, out, n, temp);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// LSD radix sort pass, 8-bit digit.
// Block b owns the contiguous items [b*4096, (b+1)*4096); wave w of the block owns the contiguous
// quarter of that, which it walks in 16 rounds of 64 consecutive items, so memory order ==
// (block, wave, round, lane) order and a rank computed in that order is stable.
// ---------------------------------------------------------------------------------------------------
// dmask: the bits of the digit that belong to the key (the last pass of a key whose width is not a multiple of 8 must
// not look at what the caller keeps above the key).
__global__ __launch_bounds__(kSortBlock) void radix_hist_kernel(const uint32_t* __restrict__ keys, int64_t n,
                                                                int shift, uint32_t dmask, uint32_t* __restrict__ ghist, int nblk)
{
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kSortItemsPerBlock;
#pragma unroll
    for (int j = 0; j < kSortItemsPerThread; j++) {
        int64_t i = base + (int64_t)j * kSortBlock + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & dmask], 1u);
    }
    __syncthreads();
    ghist[(size_t)threadIdx.x * nblk + blockIdx.x] = h[threadIdx.x];
}

// LAST (the pass of the highest digit, when the caller wants list boundaries instead of sorted keys): the keys are not
// written out; instead tile_first[k] = position of the first pair of key k, by atomicMin over the blocks that hold pairs of
// k.  A block's pairs of one digit leave it in input order (the sort is stable) and the input of the last pass is ordered
// by all lower bits, so inside a digit's run of the block the full keys never decrease: a key starts where it differs from
// its predecessor in the run (or opens the run).
template <bool LAST>
__global__ __launch_bounds__(kSortBlock) void radix_scatter_kernel(const uint32_t* __restrict__ keys_in,
                                                                   const uint32_t* __restrict__ vals_in,
                                                                   uint32_t* __restrict__ keys_out,
                                                                   uint32_t* __restrict__ vals_out, int64_t n,
                                                                   int shift, uint32_t dmask, const uint32_t* __restrict__ ghist_scanned,
                                                                   int nblk, uint32_t* __restrict__ tile_first)
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
        const uint32_t digit = (key[j] >> shift) & dmask;
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
            const uint32_t digit = (key[j] >> shift) & dmask;
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
        const uint32_t digit = (k >> shift) & dmask;
        const uint32_t dest = s_gbase[digit] + (p - s_boff[digit]);
        vals_out[dest] = s_val[p];
        if (LAST) {
            if (p == s_boff[digit] || s_key[p - 1] != k) atomicMin(&tile_first[k], dest);
        } else {
            keys_out[dest] = k;
        }
    }
}

// tile_off[t] = first position whose key is >= t = min over k >= t of tile_first[k] (0xFFFFFFFF: no pair of that key),
// tile_off[num_keys] = total.  Two small launches over chunks of 1024 keys: the chunks' minima, then every chunk's suffix
// minimum seeded with the minimum of the chunks behind it.
constexpr int kFirstChunk = 1024;

__device__ __forceinline__ uint32_t block_min_1024(uint32_t v, uint32_t* s16)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = min(v, (uint32_t)__shfl_down((int)v, d, 64));
    if ((threadIdx.x & 63) == 0) s16[threadIdx.x >> 6] = v;
    __syncthreads();
    uint32_t m = s16[0];
#pragma unroll
    for (int k = 1; k < kFirstChunk / 64; k++) m = min(m, s16[k]);
    __syncthreads();
    return m;
}

__global__ __launch_bounds__(kFirstChunk) void tile_first_chunk_min_kernel(const uint32_t* __restrict__ tile_first, int num_keys,
                                                                           uint32_t* __restrict__ chunk_min)
{
    __shared__ uint32_t s16[kFirstChunk / 64];
    const int k = blockIdx.x * kFirstChunk + threadIdx.x;
    const uint32_t m = block_min_1024(k < num_keys ? tile_first[k] : 0xFFFFFFFFu, s16);
    if (threadIdx.x == 0) chunk_min[blockIdx.x] = m;
}

__global__ __launch_bounds__(kFirstChunk) void tile_offsets_from_first_kernel(const uint32_t* __restrict__ tile_first, int num_keys,
                                                                              const uint32_t* __restrict__ chunk_min, int num_chunks,
                                                                              uint32_t total, uint32_t* __restrict__ tile_off)
{
    __shared__ uint32_t s16[kFirstChunk / 64];
    __shared__ uint32_t s_v[kFirstChunk];
    const int t = threadIdx.x, k = blockIdx.x * kFirstChunk + t;
    // everything behind this chunk
    uint32_t behind = total;
    for (int c = (int)blockIdx.x + 1 + t; c < num_chunks; c += kFirstChunk) behind = min(behind, chunk_min[c]);
    behind = block_min_1024(behind, s16);
    // suffix minimum inside the chunk (Hillis-Steele)
    s_v[t] = k < num_keys ? tile_first[k] : 0xFFFFFFFFu;
    __syncthreads();
    for (int d = 1; d < kFirstChunk; d <<= 1) {
        const uint32_t other = (t + d < kFirstChunk) ? s_v[t + d] : 0xFFFFFFFFu;
        __syncthreads();
        s_v[t] = min(s_v[t], other);
        __syncthreads();
    }
    if (k < num_keys) tile_off[k] = min(s_v[t], behind);
    if (k == num_keys) tile_off[num_keys] = total; // (num_keys a multiple of the chunk: written by thread 0 of an extra block)
}

static inline int64_t sort_blocks(int64_t n) { return (n + kSortItemsPerBlock - 1) / kSortItemsPerBlock; }

size_t sort_temp_words(int64_t n)
{
    const int64_t nblk = sort_blocks(n);
    return (size_t)(256 * nblk) + scan_temp_words(256 * nblk) + 4;
}

hipError_t sort_pairs_u32(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, int64_t n,
                          int key_bits, uint32_t* temp, uint32_t** keys_out, uint32_t** vals_out, uint32_t* tile_first,
                          hipStream_t stream)
{
    uint32_t *kin = keys_a, *vin = vals_a, *kout = keys_b, *vout = vals_b;
    if (n > 0) {
        const int64_t nblk = sort_blocks(n);
        uint32_t* ghist = temp;
        uint32_t* scan_temp = temp + 256 * nblk;
        for (int shift = 0; shift < key_bits; shift += 8) {
            const uint32_t dmask = key_bits - shift >= 8 ? 255u : (1u << (key_bits - shift)) - 1u;
            hipLaunchKernelGGL(radix_hist_kernel, dim3((unsigned)nblk), dim3(kSortBlock), 0, stream, kin, n, shift, dmask,
                               ghist, (int)nblk);
            hipError_t e = exclusive_scan_u32(ghist, ghist, 256 * nblk, scan_temp, nullptr, stream);
            if (e != hipSuccess) return e;
            if (tile_first && shift + 8 >= key_bits)
                hipLaunchKernelGGL(radix_scatter_kernel<true>, dim3((unsigned)nblk), dim3(kSortBlock), 0, stream, kin, vin,
                                   kout, vout, n, shift, dmask, ghist, (int)nblk, tile_first);
            else
                hipLaunchKernelGGL(radix_scatter_kernel<false>, dim3((unsigned)nblk), dim3(kSortBlock), 0, stream, kin, vin,
                                   kout, vout, n, shift, dmask, ghist, (int)nblk, tile_first);
            uint32_t* t;
            t = kin; kin = kout; kout = t;
            t = vin; vin = vout; vout = t;
        }
    }
    *keys_out = kin; // (not written by the last pass when tile_first was asked for)
    *vals_out = vin;
    return hipGetLastError();
}

size_t tile_first_temp_words(int num_keys) { return (size_t)(num_keys / kFirstChunk + 2); }

hipError_t launch_tile_offsets_from_first(const uint32_t* tile_first, int num_keys, uint32_t total, uint32_t* temp,
                                          uint32_t* tile_off, hipStream_t stream)
{
    const int chunks = num_keys / kFirstChunk + 1; // covers key num_keys itself (the end marker) too
    hipLaunchKernelGGL(tile_first_chunk_min_kernel, dim3(chunks), dim3(kFirstChunk), 0, stream, tile_first, num_keys, temp);
    hipLaunchKernelGGL(tile_offsets_from_first_kernel, dim3(chunks), dim3(kFirstChunk), 0, stream, tile_first, num_keys, temp,
                       chunks, total, tile_off);
    return hipGetLastError();
}

} // namespace s2d
