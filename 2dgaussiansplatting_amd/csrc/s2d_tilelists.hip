// s2d_tilelists.hip -- the per-tile lists in two levels instead of two radix passes over all (tile, splat) pairs.
//
// What has to come out (s2d_scan_sort.hip says why): for every tile the splats whose binned rectangle covers it, ascending
// in splat index (the reference's blend order, main.cpp:419).  The generic builder emits all P pairs (20 M at 4096^2 /
// 10^6 splats) in splat order and sorts them by tile id with two stable 8-bit radix passes: every pass reads and writes
// 8 bytes per pair.  Here the splat order is carried through two counting sorts of different sizes:
//
//   level 1, per (splat, tile ROW): a splat covers rows ty0..ty1 -- M entries, M = P / (columns per splat) ~ P / 4.5.  The
//            entries are emitted in splat order and sorted by row with the generic stable radix sort (n = M): every tile
//            row now has its splats in ascending order.
//   level 2, per tile row, by COLUMN: an entry covers columns tx0..tx1 of its row (carried through the row sort in the
//            upper bits of its key, so this level reads entries and ranges as two coalesced streams).  The rows are cut into chunks of C
//            consecutive entries; a chunk's workgroup marks, in a bitmap in LDS (one bit per (column, entry)), which
//            entries cover which column.  Pass A counts the entries per (chunk, column); a running sum down each row's
//            chunks gives every chunk its place inside the column's tile list and the tile's size, whose exclusive scan
//            over the tiles is tile_off; pass B builds the bitmap, ranks every (entry, column) by the set bits in front
//            of it and writes the splat index to tile_off + place + rank, through LDS so that consecutive lanes store
//            consecutive words of a column's run.
//   No pair is ever written except into its final place (4 bytes), none is read: the pairs exist only as bits in LDS.
//
// Limits: tiles_x <= kTlMaxColumns (the bitmap must fit LDS); wider images take the generic builder.
#include "s2d_device.h"

namespace s2d {

#ifndef S2D_TL_CHUNK_NARROW
#define S2D_TL_CHUNK_NARROW 512 // entries per chunk, images of up to 256 tile columns
#endif
#ifndef S2D_TL_CHUNK_WIDE
#define S2D_TL_CHUNK_WIDE 256   // ... of up to kTlMaxColumns
#endif
// pairs of a chunk staged in LDS before they are written out: six per entry (a chunk with more writes straight to memory)
template <int C> struct TlStage { static constexpr int value = 6 * C; };

__host__ __device__ inline int tl_chunk_entries(int tiles_x) { return tiles_x <= 256 ? S2D_TL_CHUNK_NARROW : S2D_TL_CHUNK_WIDE; }

// What a chunk's workgroup needs to know about itself.
struct TlChunk {
    int row;            // tile row (local to the slab); -1: no such chunk
    uint32_t e0;        // first entry (position in the row-sorted entry array)
    int cnt;            // entries in the chunk
    uint32_t hist_base; // index of (this chunk, column 0) in the histogram: chunks in order, a row of tiles_x counts each
    uint32_t nch;       // chunks of the row
};

// chunk_base[ty] = chunks of the rows before ty (a row of len entries has ceil(len / C) chunks); chunk_base[tiles_y] = all.
// One workgroup; rows <= 4096 (H <= 65536).
__global__ __launch_bounds__(1024) void tl_chunk_table_kernel(const uint32_t* __restrict__ row_off, int tiles_y, int C,
                                                              uint32_t* __restrict__ chunk_base)
{
    __shared__ uint32_t s_sum[1024];
    const int t = threadIdx.x, per = (tiles_y + 1023) / 1024;
    const int beg = min(t * per, tiles_y), end = min(beg + per, tiles_y);
    uint32_t mine = 0;
    for (int r = beg; r < end; r++) mine += (row_off[r + 1] - row_off[r] + (uint32_t)C - 1u) / (uint32_t)C;
    s_sum[t] = mine;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) { // inclusive Hillis-Steele over the threads' sums
        const uint32_t other = t >= d ? s_sum[t - d] : 0u;
        __syncthreads();
        s_sum[t] += other;
        __syncthreads();
    }
    uint32_t run = s_sum[t] - mine;
    for (int r = beg; r < end; r++) {
        chunk_base[r] = run;
        run += (row_off[r + 1] - row_off[r] + (uint32_t)C - 1u) / (uint32_t)C;
    }
    if (t == 1023) chunk_base[tiles_y] = s_sum[1023];
}

// One descriptor per possible chunk, so that a chunk's workgroup starts with one load instead of a serial search.
__global__ __launch_bounds__(256) void tl_chunk_desc_kernel(const uint32_t* __restrict__ chunk_base, const uint32_t* __restrict__ row_off,
                                                            int tiles_x, int tiles_y, int C, unsigned max_chunks, TlChunk* __restrict__ desc)
{
    const unsigned b = blockIdx.x * 256u + threadIdx.x;
    if (b >= max_chunks) return;
    TlChunk c;
    c.row = -1; c.e0 = 0u; c.cnt = 0; c.hist_base = 0u; c.nch = 0u;
    if (b < chunk_base[tiles_y]) {
        int lo = 0, hi = tiles_y - 1; // the last row with chunk_base[row] <= b (rows without chunks share their successor's base)
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (chunk_base[mid] <= b) lo = mid; else hi = mid - 1;
        }
        const uint32_t cr = b - chunk_base[lo];
        c.row = lo;
        c.nch = chunk_base[lo + 1] - chunk_base[lo];
        c.e0 = row_off[lo] + cr * (uint32_t)C;
        c.cnt = (int)min((uint32_t)C, row_off[lo + 1] - c.e0);
        c.hist_base = b * (uint32_t)tiles_x;
    }
    desc[b] = c;
}

// Pass A: how many entries of the chunk cover each column (LDS counters).
__device__ __forceinline__ int tl_tx0(uint32_t key) { return (int)((key >> kTlRowBits) & 511u); }
__device__ __forceinline__ int tl_tx1(uint32_t key) { return (int)((key >> (kTlRowBits + 9)) & 511u); }

template <int TXMAX>
__global__ __launch_bounds__(256) void tl_hist_kernel(const uint32_t* __restrict__ entry_keys,
                                                      const TlChunk* __restrict__ desc, int tiles_x, uint32_t* __restrict__ hist)
{
    __shared__ uint32_t cnt[TXMAX];
    const TlChunk c = desc[blockIdx.x];
    if (c.row < 0) return;
    for (int tx = threadIdx.x; tx < tiles_x; tx += 256) cnt[tx] = 0u;
    __syncthreads();
    for (int e = threadIdx.x; e < c.cnt; e += 256) {
        const uint32_t key = entry_keys[c.e0 + e];
        for (int tx = tl_tx0(key); tx <= tl_tx1(key); tx++) atomicAdd(cnt + tx, 1u);
    }
    __syncthreads();
    for (int tx = threadIdx.x; tx < tiles_x; tx += 256) hist[c.hist_base + (uint32_t)tx] = cnt[tx];
}

// Down the chunks of one tile row: hist[chunk][tx] becomes the number of entries of EARLIER chunks of the row that cover
// column tx (the chunk's place in that tile's list), tile_count[row * tiles_x + tx] the tile's list length.
__global__ __launch_bounds__(256) void tl_column_prefix_kernel(uint32_t* __restrict__ hist, const uint32_t* __restrict__ chunk_base,
                                                               int tiles_x, uint32_t* __restrict__ tile_count)
{
    const int row = blockIdx.y, tx = blockIdx.x * 256 + threadIdx.x;
    if (tx >= tiles_x) return;
    uint32_t run = 0;
    for (uint32_t b = chunk_base[row]; b < chunk_base[row + 1]; b++) {
        const uint32_t v = hist[(size_t)b * tiles_x + tx];
        hist[(size_t)b * tiles_x + tx] = run;
        run += v;
    }
    tile_count[row * tiles_x + tx] = run;
}

// Pass B: every (entry, column) pair of the chunk to its place in the column's tile list.
// Bitmap of the chunk: word (w, column) holds entries 32 w .. 32 w + 31; laid out [w][column] so that lanes (consecutive
// entries, one w) hitting different columns hit different banks, and a thread walking one column over w reads beside its
// neighbours' columns.
template <int TXMAX, int C>
__global__ __launch_bounds__(256) void tl_scatter_kernel(const uint32_t* __restrict__ entries, const uint32_t* __restrict__ entry_keys,
                                                         const TlChunk* __restrict__ desc, const uint32_t* __restrict__ place,
                                                         const uint32_t* __restrict__ tile_off, int tiles_x,
                                                         uint32_t* __restrict__ list)
{
    constexpr int words = C / 32;
    __shared__ uint32_t bm[words * TXMAX];   // which entries cover which column
    __shared__ uint16_t pre[words * TXMAX];  // per column: entries covering it in front of each word
    __shared__ uint32_t boff[TXMAX + 1];     // start of each column's run in the chunk's output
    __shared__ uint32_t gbase[TXMAX];        // the column's position in its tile list for this chunk
    constexpr int kTlStage = TlStage<C>::value;
    __shared__ uint32_t stage[kTlStage];     // the chunk's output in column order ...
    __shared__ uint16_t stage_col[kTlStage]; // ... and the column of each element
    __shared__ uint32_t s_wave[4];
    const TlChunk c = desc[blockIdx.x];
    if (c.row < 0) return;
    const int t = threadIdx.x;
    constexpr int kPer = C / 256; // entries per thread: their indices and rectangles stay in registers between the phases
    uint32_t my_splat[kPer];
    int my_tx0[kPer], my_tx1[kPer];
#pragma unroll
    for (int k = 0; k < kPer; k++) {
        const int e = k * 256 + t;
        my_splat[k] = e < c.cnt ? entries[c.e0 + e] : 0u;
        const uint32_t key = e < c.cnt ? entry_keys[c.e0 + e] : (1u << kTlRowBits); // (tx0 = 1 > tx1 = 0: covers nothing)
        my_tx0[k] = tl_tx0(key);
        my_tx1[k] = tl_tx1(key);
    }
    for (int q = t; q < words * tiles_x; q += 256) bm[q] = 0u;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kPer; k++) {
        const int e = k * 256 + t;
        uint32_t* row = bm + (e >> 5) * tiles_x;
        const uint32_t bit = 1u << (e & 31);
        for (int tx = my_tx0[k]; tx <= my_tx1[k]; tx++) atomicOr(row + tx, bit);
    }
    __syncthreads();
    // per column: the count in front of every word, the total, the base position
    uint32_t mine[2] = {0u, 0u}; // (TXMAX <= 512: at most two columns per thread)
    for (int k = 0, tx = t; tx < tiles_x; tx += 256, k++) {
        uint32_t run = 0;
#pragma unroll 4
        for (int w = 0; w < words; w++) {
            pre[w * tiles_x + tx] = (uint16_t)run;
            run += (uint32_t)__popc(bm[w * tiles_x + tx]);
        }
        mine[k] = run;
        gbase[tx] = tile_off[c.row * tiles_x + tx] + place[c.hist_base + (uint32_t)tx];
    }
    // exclusive scan of the column totals in column order: columns t (k = 0) come before columns 256 + t (k = 1)
    {
        const int lane = t & 63, w = t >> 6;
        uint32_t total0 = 0;
        for (int k = 0; k < 2; k++) {
            uint32_t inc = mine[k];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(inc, d, 64);
                if (lane >= d) inc += o;
            }
            if (lane == 63) s_wave[w] = inc;
            __syncthreads();
            uint32_t base = 0, tot = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (i < w) base += s_wave[i];
                tot += s_wave[i];
            }
            __syncthreads();
            const int tx = k * 256 + t;
            if (tx < tiles_x) boff[tx] = total0 + base + inc - mine[k];
            total0 += tot;
        }
        if (t == 0) boff[tiles_x] = total0;
    }
    __syncthreads();
    const uint32_t pairs = boff[tiles_x];
    const bool staged = pairs <= (uint32_t)kTlStage; // (a chunk of very wide splats writes straight to memory)
#pragma unroll
    for (int k = 0; k < kPer; k++) {
        const int e = k * 256 + t;
        const uint32_t splat = my_splat[k];
        const int w = e >> 5;
        const uint32_t below = (1u << (e & 31)) - 1u;
        for (int tx = my_tx0[k]; tx <= my_tx1[k]; tx++) {
            const uint32_t rank = (uint32_t)pre[w * tiles_x + tx] + (uint32_t)__popc(bm[w * tiles_x + tx] & below);
            if (staged) {
                stage[boff[tx] + rank] = splat;
                stage_col[boff[tx] + rank] = (uint16_t)tx;
            } else {
                list[gbase[tx] + rank] = splat;
            }
        }
    }
    if (!staged) return;
    __syncthreads();
    for (uint32_t p = t; p < pairs; p += 256) { // consecutive lanes: consecutive words of a column's run
        const uint32_t tx = stage_col[p];
        list[gbase[tx] + (p - boff[tx])] = stage[p];
    }
}

// Workspace: the per-(chunk, column) counts, the per-tile counts with their scan workspace, one descriptor per possible chunk.
size_t tl_max_chunks(uint64_t entries, int tiles_x, int tiles_y) { return (size_t)(entries / (uint64_t)tl_chunk_entries(tiles_x)) + (size_t)tiles_y + 1; }
size_t tl_workspace_words(uint64_t entries, int tiles_x, int tiles_y)
{
    const size_t chunks = tl_max_chunks(entries, tiles_x, tiles_y), tiles = (size_t)tiles_x * tiles_y;
    return chunks * (size_t)tiles_x + tiles + scan_temp_words((int64_t)tiles) + chunks * (sizeof(TlChunk) / sizeof(uint32_t)) + 8;
}

hipError_t launch_tile_lists_from_rows(const uint32_t* entries, const uint32_t* entry_keys, uint64_t num_entries, const uint32_t* row_off,
                                       Geometry g, uint32_t* chunk_base, uint32_t* workspace, uint32_t* tile_off, uint32_t* list,
                                       hipStream_t stream)
{
    if (g.num_tiles <= 0) return hipSuccess;
    if (g.tiles_x > kTlMaxColumns) return hipErrorInvalidValue;
    const int C = tl_chunk_entries(g.tiles_x);
    const unsigned chunks = (unsigned)tl_max_chunks(num_entries, g.tiles_x, g.tiles_y);
    uint32_t* hist = workspace;
    uint32_t* tile_count = hist + (size_t)chunks * g.tiles_x;
    uint32_t* scan_temp = tile_count + g.num_tiles;
    TlChunk* desc = reinterpret_cast<TlChunk*>(scan_temp + scan_temp_words((int64_t)g.num_tiles));
    hipLaunchKernelGGL(tl_chunk_table_kernel, dim3(1), dim3(1024), 0, stream, row_off, g.tiles_y, C, chunk_base);
    hipLaunchKernelGGL(tl_chunk_desc_kernel, dim3((chunks + 255) / 256), dim3(256), 0, stream, chunk_base, row_off, g.tiles_x, g.tiles_y, C,
                       chunks, desc);
    if (g.tiles_x <= 256)
        hipLaunchKernelGGL((tl_hist_kernel<256>), dim3(chunks), dim3(256), 0, stream, entry_keys, desc, g.tiles_x, hist);
    else
        hipLaunchKernelGGL((tl_hist_kernel<kTlMaxColumns>), dim3(chunks), dim3(256), 0, stream, entry_keys, desc, g.tiles_x, hist);
    hipLaunchKernelGGL(tl_column_prefix_kernel, dim3((g.tiles_x + 255) / 256, g.tiles_y), dim3(256), 0, stream, hist, chunk_base, g.tiles_x,
                       tile_count);
    // tile_off[t] = pairs of the tiles before t; tile_off[tiles] = all of them (the scan's total)
    hipError_t e = exclusive_scan_u32(tile_count, tile_off, (int64_t)g.num_tiles, scan_temp, tile_off + g.num_tiles, stream);
    if (e != hipSuccess) return e;
    if (g.tiles_x <= 256)
        hipLaunchKernelGGL((tl_scatter_kernel<256, S2D_TL_CHUNK_NARROW>), dim3(chunks), dim3(256), 0, stream, entries, entry_keys, desc, hist,
                           tile_off, g.tiles_x, list);
    else
        hipLaunchKernelGGL((tl_scatter_kernel<kTlMaxColumns, S2D_TL_CHUNK_WIDE>), dim3(chunks), dim3(256), 0, stream, entries, entry_keys, desc,
                           hist, tile_off, g.tiles_x, list);
    return hipGetLastError();
}

} // namespace s2d
