// s2d_api.hip -- the C ABI of include/splat2d.h: context, device memory, iteration sequencing.
//
// One iteration (main.cpp:414-809) of s2d_step is TWO launches on the context's stream:
//   raster_fused (forward walk + backward walk + per-tile squared error of every tile)
//   -> adam (+ the sum of the tile errors, + projection of the updated splats and the containment check for the next
//      iteration)
// s2d_forward / s2d_backward / s2d_adam_step queue the passes one by one (raster_forward, raster_backward, sqerr_finalize).
// When the tile lists have to be (re)built:  project -> count scan -> emit -> radix sort -> tile offsets.
// The host never makes the GPU wait: it reads the 4-byte containment flag after launching the raster kernel
// optimistically, and the 4-byte pair count only when lists are rebuilt.
#include "../../include/splat2d.h"
#include "../../include/splat2d_test.h"

#include <algorithm>
#include <vector>
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <new>

#include "s2d_device.h"

using namespace s2d;

struct s2d_ctx {
    s2d_config cfg{};
    Geometry g{};
    int n = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    float lr = 0.05f;

    // parameters / optimiser state / gradients (AoS, the reference's layouts)
    float* d_splats = nullptr;   // n * 9
    float* d_adams = nullptr;    // n * 18
    uint8_t* d_dormant = nullptr; // n: 1 = all of the splat's Adam moments are zero (adam_kernel keeps it; cleared with every outside write)
    float* d_grads_own = nullptr;
    float* d_grads = nullptr;    // buffer in use (own or bound)
    // projection + binning
    ProjRec* d_proj = nullptr;
    TileRect* d_rects = nullptr;
    uint32_t* d_counts = nullptr;
    uint32_t* d_offsets = nullptr;
    uint32_t* d_scan_temp = nullptr;
    uint32_t* d_total = nullptr;
    uint32_t* d_keys[2] = {nullptr, nullptr};
    uint32_t* d_vals[2] = {nullptr, nullptr};
    uint32_t* d_sort_temp = nullptr;
    unsigned long long* d_wave_masks = nullptr; // 4 x u64 per listed pair: forward -> backward lane masks
    bool deterministic = false;                 // S2D_CFG_DETERMINISTIC
    float* d_det_data = nullptr;                // [pair capacity][9] per-(tile, splat) partial gradients
    uint32_t* d_det_stamp = nullptr;            // [pair capacity]
    uint32_t* d_det_touched = nullptr;          // [n]: which of a splat's slots the current pass wrote (zero between passes)
    uint32_t det_epoch = 0;                     // stamps written so far (monotone; 0 = never)
    uint64_t pair_capacity = 0;
    uint32_t* d_tile_off = nullptr;
    uint32_t* d_tile_first = nullptr; // per tile id (padded to a power of two): position of its first pair (last radix pass)
    // tile lists in two levels (s2d_tilelists.hip): (splat, tile row) entries sorted by row, then per-row counting sort by column
    bool two_level = false;            // tiles_x <= kTlMaxColumns and not S2D_CFG_GENERIC_BINNING
    uint32_t* d_row_counts = nullptr;  // per splat: tile rows its rectangle covers
    uint32_t* d_row_offsets = nullptr; // ... scanned
    uint32_t* d_row_off = nullptr;     // [tiles_y + 1]: where each tile row's entries begin
    uint32_t* d_chunk_base = nullptr;  // [tiles_y + 1]
    uint32_t* d_tl_hist = nullptr;     // per (row, column, chunk) counts + scan workspace
    size_t tl_hist_capacity = 0;       // words
    uint32_t* d_list = nullptr; // == one of d_vals after the sort
    // Index-range ("chunked") rendering: when the (tile, splat) pairs of a scene exceed chunk_pairs -- at the latest 2^32 - 65536,
    // what 32-bit list positions can address -- the splats are cut into consecutive index ranges of at most that many
    // pairs, and the lists of one range at a time are built and walked front to back (chunked_forward / chunked_backward)
    uint64_t chunk_pairs = 1ull << 30;   // S2D_CHUNK_PAIRS overrides (tests force the path on small scenes)
    std::vector<int> chunks;             // range k = splats [chunks[k], chunks[k+1]); empty: one set of lists
    int chunks_used = 0;                 // ranges the last forward pass walked before every pixel was saturated
    int chunk_built = -1;                // the range whose lists are in the buffers now
    float4* d_state = nullptr;           // per pixel of the slab: (r, g, b, T) carried from range to range
    uint32_t* d_chunk_alive = nullptr;   // != 0: some pixel is still above the throughput cut-off after this range
    uint64_t pairs = 0;
    uint64_t rebins = 0;
    bool lists_valid = false;
    bool proj_fresh = false; // d_proj and d_status->rebin_needed describe the CURRENT parameters
    hipEvent_t ev_total = nullptr; // recorded behind the copy of the pair count to the host (rebuild_lists)
    hipEvent_t ev_flag = nullptr;  // recorded behind the kernel that ran the latest containment check
    int check_seq = 1;             // its sequence number (both stamp words start at 0: nothing matches before a check): the stamp that kernel writes if a splat left its rectangle
    int* h_rebin_stamp = nullptr;  // host-mapped copy of that stamp (written by the kernel, read after ev_flag)
    int rebin_interval = 1;
    int since_rebin = 0;
    float margin = 0.0f;
    // images
    // image0 / imageRef (main.cpp:310, :254): the rows [row_begin, row_end) of this context's slab only -- a context
    // never touches another row, so a 1/8 slab of 8192^2 holds 2 x 134 MB instead of 2 x 1.07 GB
    void* d_image0 = nullptr;  // RGBA32F, or 4 x fp16 per pixel with S2D_CFG_FP16_IMAGES
    void* d_ref = nullptr;
    bool half_images = false;
    size_t pixel_bytes = sizeof(float4);
    double* d_tile_sqerr = nullptr;
    uint32_t* d_held_ids = nullptr;   // ... and their ascending id list, *d_held_count long, for the Adam kernel
    uint32_t* d_held_count = nullptr;
    uint32_t* d_held_work = nullptr;  // n words of scan workspace
    // Compact held state: with slab ownership the Adam step touches only the splats the rank holds -- a seventh of them at
    // eight ranks, scattered through the id-indexed arrays (36- and 72-byte records, a cache line or two each).  Their
    // parameters and moments are therefore kept in compact arrays in the order of d_held_ids, which the Adam kernel reads
    // and writes in whole lines; the id-indexed arrays are brought up to date (compact_flush) before anything else reads
    // them -- a projection pass, a hold-set refresh, a row transfer, a read-back -- and the compact copy is made afresh
    // (compact_load) whenever the held set or the id-indexed arrays change from outside.
    float* d_csplats = nullptr;  // n * 9 (capacity: every splat)
    float* d_cadams = nullptr;   // n * 18
    bool compact_live = false;   // the compact arrays mirror the held splats
    bool compact_dirty = false;  // ... and are ahead of the id-indexed arrays (Adam steps since the last flush)
    bool compact_enabled = true; // S2D_COMPACT_HELD=0 turns it off (A/B)
    uint8_t* d_held = nullptr; // slab ownership: 1 = this rank holds (updates) the splat; nullptr = all (s2d_halo_commit)
    double* d_sqerr_trace = nullptr;
    int trace_cap = 1 << 16;
    DeviceStatus* d_status = nullptr;
    PairCounters* d_counters = nullptr;
    // pinned host mirrors
    uint32_t* h_total = nullptr;
    DeviceStatus* h_status = nullptr;
    double* h_trace = nullptr;          // kHostTrace squared errors: s2d_step reads its trace and the status word in ONE round trip

    // host-side state of the reference's main()
    float beta1t = 1.0f, beta2t = 1.0f; // main.cpp:274-275
    int iterations = 0;                 // main.cpp:278
    float good_beta1t = 1.0f, good_beta2t = 1.0f; // the three above at the last point known to be finite
    int good_iterations = 0;
    bool have_target = false;
    bool have_forward = false;  // image0 holds the framebuffer of the CURRENT parameters (s2d_backward reads it)
    bool have_backward = false;
    bool sqerr_deferred = false; // the squared-error reduction of the last backward pass rides on the next Adam launch
    int last_sqerr_slot = -1;
    char err[512] = {0};
};

namespace {

constexpr int kHostTrace = 4096;

int fail(s2d_ctx* c, int code, const char* fmt, ...)
{
    if (c) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(c->err, sizeof(c->err), fmt, ap);
        va_end(ap);
    }
    return code;
}

#define S2D_HIP(c, expr)                                                                              \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            return fail((c), S2D_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

template <typename T>
hipError_t dev_alloc(T** p, size_t count)
{
    return hipMalloc((void**)p, std::max<size_t>(count, 1) * sizeof(T));
}

int use_device(s2d_ctx* c)
{
    S2D_HIP(c, hipSetDevice(c->device));
    return S2D_OK;
}

int key_bits_for(int num_tiles)
{
    int bits = 0;
    while ((1 << bits) < num_tiles) bits++;
    return bits;
}

int ensure_pair_capacity(s2d_ctx* c, uint64_t need)
{
    if (need <= c->pair_capacity) return S2D_OK;
    if (need >= 0xFFFF0000ull) return fail(c, S2D_E_NOMEM, "tile lists need %llu pairs (> 2^32)", (unsigned long long)need);
    uint64_t cap = std::max<uint64_t>(need + need / 4 + 4096, 1 << 16);
    if (cap > 0xFFFF0000ull) cap = 0xFFFF0000ull;
    S2D_HIP(c, hipStreamSynchronize(c->stream));
    for (int k = 0; k < 2; k++) {
        if (c->d_keys[k]) S2D_HIP(c, hipFree(c->d_keys[k]));
        if (c->d_vals[k]) S2D_HIP(c, hipFree(c->d_vals[k]));
        c->d_keys[k] = c->d_vals[k] = nullptr;
    }
    if (c->d_sort_temp) S2D_HIP(c, hipFree(c->d_sort_temp));
    if (c->d_wave_masks) S2D_HIP(c, hipFree(c->d_wave_masks));
    if (c->d_det_data) S2D_HIP(c, hipFree(c->d_det_data));
    if (c->d_det_stamp) S2D_HIP(c, hipFree(c->d_det_stamp));
    c->d_sort_temp = nullptr;
    c->d_wave_masks = nullptr;
    c->d_det_data = nullptr;
    c->d_det_stamp = nullptr;
    c->pair_capacity = 0;
    for (int k = 0; k < 2; k++) {
        S2D_HIP(c, dev_alloc(&c->d_keys[k], cap));
        S2D_HIP(c, dev_alloc(&c->d_vals[k], cap));
    }
    S2D_HIP(c, dev_alloc(&c->d_sort_temp, sort_temp_words((int64_t)cap)));
    S2D_HIP(c, dev_alloc(&c->d_wave_masks, (size_t)cap * 4));
    if (c->deterministic) {
        S2D_HIP(c, dev_alloc(&c->d_det_data, (size_t)cap * kDetStride));
        S2D_HIP(c, dev_alloc(&c->d_det_stamp, (size_t)cap));
        S2D_HIP(c, hipMemsetAsync(c->d_det_stamp, 0, (size_t)cap * sizeof(uint32_t), c->stream));
    }
    c->pair_capacity = cap;
    return S2D_OK;
}

// The workspace of the two-level builder grows with the number of (splat, tile row) entries.
int ensure_tl_hist(s2d_ctx* c, uint64_t entries)
{
    const size_t need = tl_workspace_words(entries, c->g.tiles_x, c->g.tiles_y);
    if (need <= c->tl_hist_capacity) return S2D_OK;
    S2D_HIP(c, hipStreamSynchronize(c->stream));
    if (c->d_tl_hist) S2D_HIP(c, hipFree(c->d_tl_hist));
    c->d_tl_hist = nullptr;
    c->tl_hist_capacity = 0;
    const size_t cap = need + need / 4 + 4096;
    S2D_HIP(c, dev_alloc(&c->d_tl_hist, cap));
    c->tl_hist_capacity = cap;
    return S2D_OK;
}

// (Re)build the per-tile lists from the current parameters.  The projection has already been queued with mode 0.
// Two builders with the same result (every tile's list ascending in splat index): the two-level one of s2d_tilelists.hip
// (images of up to kTlMaxColumns tile columns), and the generic one -- all (tile, splat) pairs emitted in splat order and
// radix-sorted by tile -- for wider images and on request (S2D_CFG_GENERIC_BINNING).
// first / count: the index range of the splats to list (count < 0: all of them).  A range's lists hold indices RELATIVE to
// its first splat -- every per-splat array is handed over from that splat on -- and so do the scanned offsets.
// Returns kNeedChunks (and builds nothing) when all splats were asked for and their pairs exceed chunk_pairs.
constexpr int kNeedChunks = -100;
int rebuild_lists(s2d_ctx* c, int first = 0, int count = -1)
{
    const bool whole = count < 0;
    const int n = whole ? c->n : count;
    const TileRect* const rects = c->d_rects + first;
    const uint32_t *const counts = c->d_counts + first, *const row_counts = c->d_row_counts ? c->d_row_counts + first : nullptr;
    uint32_t *const offsets = c->d_offsets + first, *const row_offsets = c->d_row_offsets ? c->d_row_offsets + first : nullptr;
    // the scans' last kernels store their totals into host-mapped memory themselves (no copy engine between two kernels)
    S2D_HIP(c, exclusive_scan_u32(counts, offsets, n, c->d_scan_temp, c->d_total, c->stream, c->h_total));
    if (c->two_level)
        S2D_HIP(c, exclusive_scan_u32(row_counts, row_offsets, n, c->d_scan_temp, c->d_total + 1, c->stream, c->h_total + 1));
    S2D_HIP(c, hipEventRecord(c->ev_total, c->stream));
    // The emission needs the offsets, not the totals (it never writes past the buffers' capacity): queue it behind the
    // scans and wait for the SCANS only, so the host reads the totals and queues the rest while the emission runs instead
    // of the device idling through the host's round trip (~30 us per rebuild).  Only when the pairs outgrow the buffers
    // (rare: they are sized with a quarter to spare) is the emission queued again.
    auto emit = [&]() -> hipError_t {
        if (c->two_level) // (there are never more entries than pairs: the pair buffers hold them)
            return launch_emit_row_entries(rects, row_offsets, row_counts, n, c->d_keys[0], c->d_vals[0],
                                           (uint32_t)c->pair_capacity, c->stream);
        return launch_emit_pairs(rects, offsets, counts, n, c->g, c->d_keys[0], c->d_vals[0], (uint32_t)c->pair_capacity,
                                 c->stream);
    };
    S2D_HIP(c, emit());
    S2D_HIP(c, hipEventSynchronize(c->ev_total));
    const uint64_t total = *(volatile uint32_t*)c->h_total; // saturates at 0xFFFFFFFF instead of wrapping (scan_top_kernel)
    if (whole && total > c->chunk_pairs) return kNeedChunks; // (the emission queued above wrote within the buffers' capacity: harmless)
    if (total >= 0xFFFF0000ull)
        return fail(c, S2D_E_NOMEM, "the tile lists of splats %d..%d need more than 2^32 - 65536 (tile, splat) pairs", first, first + n - 1);
    if (total > c->pair_capacity) {
        int rc = ensure_pair_capacity(c, total);
        if (rc != S2D_OK) return rc;
        S2D_HIP(c, emit());
    }
    uint32_t *k_out = nullptr, *v_out = nullptr;
    if (c->two_level) {
        const uint64_t entries = *(volatile uint32_t*)(c->h_total + 1);
        if (int rc = ensure_tl_hist(c, entries)) return rc;
        const int row_bits = key_bits_for(c->g.tiles_y);
        if (row_bits > 0) { // level 1: the entries by tile row, and where every row begins
            // (sorted keys written out and compared: with only tiles_y distinct keys the last pass's atomicMin per (block,
            // key) would pile thousands of atomics on each of a few hundred words)
            S2D_HIP(c, sort_pairs_u32(c->d_keys[0], c->d_vals[0], c->d_keys[1], c->d_vals[1], (int64_t)entries, row_bits,
                                      c->d_sort_temp, &k_out, &v_out, nullptr, c->stream));
            S2D_HIP(c, launch_tile_offsets(k_out, (uint32_t)entries, c->g.tiles_y, c->d_row_off, c->stream, (1u << kTlRowBits) - 1u));
        } else { // one tile row: the emission order is the row's order
            const uint32_t two[2] = {0u, (uint32_t)entries};
            S2D_HIP(c, hipMemcpyAsync(c->d_row_off, two, sizeof(two), hipMemcpyHostToDevice, c->stream));
            S2D_HIP(c, hipStreamSynchronize(c->stream)); // (`two` lives on this stack frame)
            v_out = c->d_vals[0];
            k_out = c->d_keys[0];
        }
        // level 2: every row's entries by column, straight into the lists (the value buffer the sort finished with is free)
        uint32_t* list = v_out == c->d_vals[0] ? c->d_vals[1] : c->d_vals[0];
        S2D_HIP(c, launch_tile_lists_from_rows(v_out, k_out, entries, c->d_row_off, c->g, c->d_chunk_base, c->d_tl_hist,
                                               c->d_tile_off, list, c->stream));
        c->d_list = list;
    } else {
        const int key_bits = key_bits_for(c->g.num_tiles);
        if (key_bits > 0) {
            // the last radix pass records where each tile's pairs begin instead of writing the sorted keys out
            S2D_HIP(c, hipMemsetAsync(c->d_tile_first, 0xFF, ((size_t)1 << key_bits) * sizeof(uint32_t), c->stream));
            S2D_HIP(c, sort_pairs_u32(c->d_keys[0], c->d_vals[0], c->d_keys[1], c->d_vals[1], (int64_t)total, key_bits,
                                      c->d_sort_temp, &k_out, &v_out, c->d_tile_first, c->stream));
            S2D_HIP(c, launch_tile_offsets_from_first(c->d_tile_first, c->g.num_tiles, (uint32_t)total,
                                                      c->d_tile_first + ((size_t)1 << key_bits), c->d_tile_off, c->stream));
        } else { // a single tile: nothing to sort
            S2D_HIP(c, sort_pairs_u32(c->d_keys[0], c->d_vals[0], c->d_keys[1], c->d_vals[1], (int64_t)total, key_bits,
                                      c->d_sort_temp, &k_out, &v_out, nullptr, c->stream));
            S2D_HIP(c, launch_tile_offsets(k_out, (uint32_t)total, c->g.num_tiles, c->d_tile_off, c->stream));
        }
        c->d_list = v_out;
    }
    c->pairs = total;
    c->rebins++;
    c->lists_valid = whole; // a range's lists are walked once and replaced by the next range's
    c->since_rebin = 0;
    return S2D_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Index-range ("chunked") rendering.  The reference's loops have no limit on the number of (pixel, splat) pairs
// (main.cpp:492-536); 32-bit list positions have one, and long before it the list and mask buffers have a price.  A scene
// beyond chunk_pairs is rendered range by range: cut where the running pair count would pass the budget, build the
// lists of one range, walk them, carry the per-pixel (colour, T) to the next range.  Blend order is index order
// (main.cpp:419), so the cut changes no operation: the framebuffer is bit for bit the unchunked one, and so is every
// gradient term (the sums differ in the order the atomics arrive, as always).  Lists are rebuilt every pass: this is the
// path for scenes that do not fit, not a fast one.
// ---------------------------------------------------------------------------------------------------------------------
int plan_chunks(s2d_ctx* c)
{
    std::vector<uint32_t> cnt((size_t)c->n);
    S2D_HIP(c, hipMemcpyAsync(cnt.data(), c->d_counts, cnt.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    S2D_HIP(c, hipStreamSynchronize(c->stream));
    c->chunks.assign(1, 0);
    uint64_t acc = 0;
    for (int i = 0; i < c->n; i++) {
        if (acc > 0 && acc + cnt[(size_t)i] > c->chunk_pairs) {
            c->chunks.push_back(i);
            acc = 0;
        }
        acc += cnt[(size_t)i];
    }
    c->chunks.push_back(c->n);
    if (!c->d_state) S2D_HIP(c, dev_alloc(&c->d_state, (size_t)c->g.W * (size_t)(c->g.row_end - c->g.row_begin)));
    if (!c->d_chunk_alive) S2D_HIP(c, dev_alloc(&c->d_chunk_alive, 1));
    return S2D_OK;
}

int build_chunk(s2d_ctx* c, int k)
{
    if (c->chunk_built == k) return S2D_OK;
    c->chunk_built = -1;
    if (int rc = rebuild_lists(c, c->chunks[(size_t)k], c->chunks[(size_t)k + 1] - c->chunks[(size_t)k])) return rc;
    c->chunk_built = k;
    return S2D_OK;
}

// Forward pass over the ranges (main.cpp:414-546); stops behind the range after which no pixel of the slab is above the
// throughput cut-off any more (main.cpp:520: nothing later could change a pixel).
int chunked_forward(s2d_ctx* c)
{
    const bool exact = (c->cfg.flags & S2D_CFG_EXACT_EXP) != 0;
    const int K = (int)c->chunks.size() - 1;
    c->chunks_used = 0;
    c->chunk_built = -1;
    for (int k = 0; k < K; k++) {
        if (int rc = build_chunk(c, k)) return rc;
        S2D_HIP(c, hipMemsetAsync(c->d_chunk_alive, 0, sizeof(uint32_t), c->stream));
        S2D_HIP(c, launch_raster_forward_chunk(c->d_tile_off, c->d_list, c->d_proj + c->chunks[(size_t)k], c->d_image0, c->half_images,
                                               c->d_state, k == 0, c->d_wave_masks, c->g, c->d_status, c->iterations,
                                               c->d_chunk_alive, exact, c->stream));
        c->chunks_used = k + 1;
        if (k + 1 < K) {
            S2D_HIP(c, hipMemcpyAsync(c->h_total + 2, c->d_chunk_alive, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
            S2D_HIP(c, hipStreamSynchronize(c->stream));
            if (*(volatile uint32_t*)(c->h_total + 2) == 0u) break;
        }
    }
    return S2D_OK;
}

// Backward pass over the same ranges (main.cpp:548-712), from a fresh per-pixel state; image0 holds the final colours.
int chunked_backward(s2d_ctx* c, bool need_opacity_grad)
{
    const bool exact = (c->cfg.flags & S2D_CFG_EXACT_EXP) != 0;
    for (int k = 0; k < c->chunks_used; k++) {
        if (int rc = build_chunk(c, k)) return rc;
        const int first = c->chunks[(size_t)k], count = c->chunks[(size_t)k + 1] - first;
        DetGather dg{};
        if (c->deterministic) {
            c->det_epoch++;
            dg = DetGather{c->d_rects + first, c->d_offsets + first, c->d_counts + first, c->d_det_data, c->d_det_stamp,
                           c->d_det_touched + first, c->det_epoch, count};
        }
        S2D_HIP(c, launch_raster_backward_chunk(c->d_tile_off, c->d_list, c->d_proj + first, c->d_image0, c->d_ref, c->half_images,
                                                c->d_state, k == 0, c->d_wave_masks, c->d_grads + (size_t)first * 9, c->d_tile_sqerr,
                                                c->g, need_opacity_grad, c->deterministic ? &dg : nullptr, c->d_status,
                                                c->iterations, exact, c->stream));
    }
    return S2D_OK;
}

// First raster launch of an iteration, on lists believed (optimistic) or known to cover the current parameters:
// the forward kernel alone, or the fused forward + backward kernel.
struct RasterJob {
    bool fused = false;        // forward + backward walk in one launch
    bool need_opacity_grad = true;
    bool write_image = true;   // fused only: store image0 (nothing but s2d_get_image reads it)
    bool sum_sqerr = false;    // fused only, small images: the launch's last tile also adds up the tile errors
};

int launch_raster(s2d_ctx* c, bool optimistic, const RasterJob& job)
{
    const int abort_stamp = optimistic ? c->check_seq : 0;
    const bool count = (c->cfg.flags & S2D_CFG_COUNT_PAIRS) != 0, exact = (c->cfg.flags & S2D_CFG_EXACT_EXP) != 0;
    if (!job.fused) {
        S2D_HIP(c, launch_raster_forward(c->d_tile_off, c->d_list, c->d_proj, c->d_image0, c->half_images, c->d_wave_masks,
                                         c->g, c->d_status, abort_stamp, c->iterations, c->d_counters, count, exact, c->stream));
        return S2D_OK;
    }
    DetGather dg{};
    if (c->deterministic) {
        c->det_epoch++; // a fresh stamp per backward pass (slots of earlier passes become invalid)
        dg = DetGather{c->d_rects, c->d_offsets, c->d_counts, c->d_det_data, c->d_det_stamp, c->d_det_touched, c->det_epoch, c->n};
    }
    S2D_HIP(c, launch_raster_fused(c->d_tile_off, c->d_list, c->d_proj, c->d_image0, c->d_ref, c->half_images, c->d_wave_masks,
                                   c->d_grads, c->d_tile_sqerr, c->g, job.need_opacity_grad, c->deterministic ? &dg : nullptr,
                                   c->d_status, abort_stamp, c->iterations, job.write_image, exact,
                                   job.sum_sqerr ? SqerrJob{c->d_tile_sqerr, c->g.num_tiles,
                                                            c->d_sqerr_trace + c->iterations % c->trace_cap,
                                                            c->d_tile_sqerr + c->g.num_tiles}
                                                 : SqerrJob{nullptr, 0, nullptr, nullptr},
                                   c->stream));
    return S2D_OK;
}

// Compact held state (see s2d_ctx): bring the id-indexed parameter / moment arrays up to date ...
int compact_flush(s2d_ctx* c)
{
    if (!c->compact_live || !c->compact_dirty) return S2D_OK;
    S2D_HIP(c, launch_compact_copy(c->d_splats, 9, c->d_held_ids, c->d_held_count, c->n, c->d_csplats, false, c->stream));
    S2D_HIP(c, launch_compact_copy(c->d_adams, 18, c->d_held_ids, c->d_held_count, c->n, c->d_cadams, false, c->stream));
    c->compact_dirty = false;
    return S2D_OK;
}

// ... and make the compact copy afresh from them (the held set, or the arrays, changed from outside).
int compact_load(s2d_ctx* c)
{
    c->compact_live = false;
    c->compact_dirty = false;
    if (!c->d_held || !c->compact_enabled || c->n <= 0) return S2D_OK;
    if (!c->d_csplats) {
        S2D_HIP(c, dev_alloc(&c->d_csplats, (size_t)c->n * 9));
        S2D_HIP(c, dev_alloc(&c->d_cadams, (size_t)c->n * 18));
    }
    S2D_HIP(c, launch_compact_copy(c->d_splats, 9, c->d_held_ids, c->d_held_count, c->n, c->d_csplats, true, c->stream));
    S2D_HIP(c, launch_compact_copy(c->d_adams, 18, c->d_held_ids, c->d_held_count, c->n, c->d_cadams, true, c->stream));
    c->compact_live = true;
    return S2D_OK;
}

// Project the splats, make sure the tile lists cover them, run the forward raster (or the fused forward + backward).
//
// Steady state (lists re-used): the projection of the current parameters and the containment check were produced by
// the Adam kernel of the previous iteration, which stamps a device word and a host-mapped word with the check's
// sequence number if some splat left its binned rectangle.  The raster kernel is launched OPTIMISTICALLY: it
// compares the device word with that sequence number and does nothing on a match, and the host reads its word only
// after the launch (waiting for the CHECKING kernel, not the raster kernel), so the GPU never waits for the host and
// no flag has to be copied or cleared.  If the check failed the lists are rebuilt and the raster kernel is
// launched again.  (In deterministic mode the gather pass queued behind a voided fused launch adds nothing: the
// slots carry no stamp of that pass.)
int queue_raster(s2d_ctx* c, const RasterJob& job)
{
    if (!c->have_target) return fail(c, S2D_E_STATE, "no target image set (s2d_set_target)");
    const bool scheduled = !c->lists_valid || c->rebin_interval <= 1 || c->since_rebin >= c->rebin_interval;
    bool rebuild = scheduled;
    if (!scheduled) {
        if (!c->proj_fresh) { // parameters changed without a fused projection: project + check now
            if (int rc = compact_flush(c)) return rc;
            c->check_seq++;
            S2D_HIP(c, launch_project(c->d_splats, c->d_held, c->n, c->g, 0.0f, 1, c->d_proj, c->d_rects, c->d_counts, nullptr, c->d_status,
                                      c->check_seq, c->h_rebin_stamp, c->stream));
            S2D_HIP(c, hipEventRecord(c->ev_flag, c->stream));
            c->proj_fresh = true;
        }
        if (int rc = launch_raster(c, true, job)) return rc;
        S2D_HIP(c, hipEventSynchronize(c->ev_flag)); // the checking kernel, not the raster kernel
        rebuild = *(volatile int*)c->h_rebin_stamp == c->check_seq;
    }
    if (rebuild) {
        if (int rc = compact_flush(c)) return rc;
        S2D_HIP(c, launch_project(c->d_splats, c->d_held, c->n, c->g, c->margin, 0, c->d_proj, c->d_rects, c->d_counts,
                                  c->two_level ? c->d_row_counts : nullptr, c->d_status, 0, nullptr, c->stream));
        c->chunks.clear();
        int rc = rebuild_lists(c);
        if (rc == kNeedChunks) {
            // more pairs than one set of lists may hold: render by index ranges (every pass rebuilds: lists_valid stays false)
            if (c->cfg.flags & S2D_CFG_COUNT_PAIRS)
                return fail(c, S2D_E_NOMEM, "pair counting (S2D_CFG_COUNT_PAIRS) is not available for scenes beyond %llu (tile, splat) pairs",
                            (unsigned long long)c->chunk_pairs);
            c->lists_valid = false;
            c->proj_fresh = true;
            c->check_seq++;
            if ((rc = plan_chunks(c)) != S2D_OK) return rc;
            if ((rc = chunked_forward(c)) != S2D_OK) return rc;
            if (job.fused && (rc = chunked_backward(c, job.need_opacity_grad)) != S2D_OK) return rc;
            c->have_forward = true; // the forward pass over the ranges always stores image0
            c->have_backward = false;
            return S2D_OK;
        }
        if (rc != S2D_OK) return rc;
        c->proj_fresh = true;
        c->check_seq++; // the new lists cover the current parameters: a stamp that asked for them matches nothing now
        if (int rc2 = launch_raster(c, false, job)) return rc2;
    }
    c->have_forward = !job.fused || job.write_image; // a fused launch told not to store image0 leaves an older frame there
    c->have_backward = false;
    return S2D_OK;
}

int queue_forward(s2d_ctx* c) { return queue_raster(c, RasterJob{}); }

// Sum of the tile errors of the backward pass just queued -> ring slot of this iteration.  defer: leave it to the next
// Adam launch, whose first workgroups do it on the way (one dispatch less per iteration); whoever wants the value
// before that (s2d_get_mse, s2d_get_sqerr_trace) flushes it with the standalone kernel (flush_sqerr).
int queue_sqerr(s2d_ctx* c, bool defer = false)
{
    const int slot = c->iterations % c->trace_cap;
    c->last_sqerr_slot = slot;
    c->have_backward = true;
    // (worth it only when the Adam launch has a workgroup per chunk: a 4-workgroup launch would walk 16 chunks each)
    c->sqerr_deferred = defer && (c->n + 255) / 256 >= kSqerrChunks;
    if (c->sqerr_deferred) return S2D_OK;
    S2D_HIP(c, launch_sqerr_finalize(c->d_tile_sqerr, c->g.num_tiles, c->d_sqerr_trace + slot, c->d_tile_sqerr + c->g.num_tiles,
                                     c->d_status, c->iterations, c->stream));
    return S2D_OK;
}

int queue_backward(s2d_ctx* c, bool need_opacity_grad)
{
    if (!c->have_forward) return fail(c, S2D_E_STATE, "s2d_backward needs s2d_forward on the current parameters");
    if (!c->chunks.empty()) { // the forward pass went over index ranges: so does this one
        if (int rc = chunked_backward(c, need_opacity_grad)) return rc;
        return queue_sqerr(c);
    }
    DetGather dg{};
    if (c->deterministic) {
        c->det_epoch++; // a fresh stamp per backward pass (slots of earlier passes become invalid)
        dg = DetGather{c->d_rects, c->d_offsets, c->d_counts, c->d_det_data, c->d_det_stamp, c->d_det_touched, c->det_epoch, c->n};
    }
    S2D_HIP(c, launch_raster_backward(c->d_tile_off, c->d_list, c->d_proj, c->d_image0, c->d_ref, c->half_images,
                                      c->d_wave_masks, c->d_grads,
                                      c->d_tile_sqerr, c->g, need_opacity_grad, c->deterministic ? &dg : nullptr,
                                      c->d_status, c->iterations, c->d_counters, (c->cfg.flags & S2D_CFG_COUNT_PAIRS) != 0,
                                      (c->cfg.flags & S2D_CFG_EXACT_EXP) != 0, c->stream));
    return queue_sqerr(c);
}

int flush_sqerr(s2d_ctx* c)
{
    if (!c->sqerr_deferred) return S2D_OK;
    c->sqerr_deferred = false;
    S2D_HIP(c, launch_sqerr_finalize(c->d_tile_sqerr, c->g.num_tiles, c->d_sqerr_trace + c->last_sqerr_slot,
                                     c->d_tile_sqerr + c->g.num_tiles, c->d_status, c->iterations, c->stream));
    return S2D_OK;
}

// Forward + backward (+ squared error) of the current parameters through the fused kernel.  Pair counting is a
// property of the separate kernels only, so a counting context takes those.
int queue_forward_backward(s2d_ctx* c, bool need_opacity_grad, bool write_image, bool defer_sqerr = true)
{
    if (c->cfg.flags & S2D_CFG_COUNT_PAIRS) {
        if (int rc = queue_forward(c)) return rc;
        return queue_backward(c, need_opacity_grad);
    }
    RasterJob job;
    job.fused = true;
    job.need_opacity_grad = need_opacity_grad;
    job.write_image = write_image;
    // Small scene (few tiles, and too few splats for the Adam launch to have a workgroup per chunk of tile errors): the
    // raster launch's last tile adds up the tile errors itself.  (Where the Adam launch can do it on the way that is
    // cheaper still: 535x426 / 50 k measured 6.9 % slower with the in-raster sum.)
    job.sum_sqerr = c->g.num_tiles <= kSqerrSmallTiles && (c->n + 255) / 256 < kSqerrChunks;
    if (int rc = queue_raster(c, job)) return rc;
    if (job.sum_sqerr && c->chunks.empty()) { // nothing left to queue
        c->last_sqerr_slot = c->iterations % c->trace_cap;
        c->have_backward = true;
        c->sqerr_deferred = false;
        return S2D_OK;
    }
    return queue_sqerr(c, defer_sqerr);
}

int queue_adam(s2d_ctx* c, uint32_t flags)
{
    c->beta1t *= kAdamBeta1; // main.cpp:718-719
    c->beta2t *= kAdamBeta2;
    // With re-usable lists the Adam kernel also projects the updated splats and checks them against their binned
    // rectangles (what the next forward needs), which saves a pass over the parameters per iteration.
    const bool fuse = c->lists_valid && c->rebin_interval > 1;
    if (fuse) c->check_seq++;
    const bool compact = c->compact_live && c->d_held_ids != nullptr;
    S2D_HIP(c, launch_adam(compact ? c->d_csplats : c->d_splats, compact ? c->d_cadams : c->d_adams, c->d_grads, c->d_held_ids,
                           c->d_held_count, c->n, c->g, c->beta1t, c->beta2t,
                           c->lr,
                           ((flags & S2D_STEP_OPTIMIZE_OPACITY) ? 1 : 0) | ((c->cfg.flags & S2D_CFG_ADAM_FP32) ? 2 : 0),
                           c->iterations, c->d_status,
                           fuse ? c->d_proj : nullptr, c->d_rects, c->check_seq, c->h_rebin_stamp, c->d_dormant,
                           c->sqerr_deferred ? SqerrJob{c->d_tile_sqerr, c->g.num_tiles, c->d_sqerr_trace + c->last_sqerr_slot,
                                                        c->d_tile_sqerr + c->g.num_tiles}
                                             : SqerrJob{nullptr, 0, nullptr, nullptr},
                           compact, c->stream));
    if (compact) c->compact_dirty = true;
    c->sqerr_deferred = false;
    if (fuse) S2D_HIP(c, hipEventRecord(c->ev_flag, c->stream));
    c->proj_fresh = fuse;
    c->iterations++; // main.cpp:809
    c->since_rebin++;
    c->have_forward = false;
    c->have_backward = false;
    return S2D_OK;
}

// Splats or Adam moments are about to be written from outside the Adam kernel: nothing is known to be dormant any more
// (the next step of every splat is a full one, which also applies the constraints to whatever was loaded).
int clear_dormant(s2d_ctx* c)
{
    if (c->n > 0) S2D_HIP(c, hipMemsetAsync(c->d_dormant, 0, (size_t)c->n, c->stream));
    return S2D_OK;
}

// New parameters (init / set_splats): a non-finite event of the old ones no longer stops the queue.
int reset_status(s2d_ctx* c)
{
    static const DeviceStatus fresh{0, INT_MAX, 0, 0}; // rebin_needed 0 matches no check (sequence numbers start at 1)
    S2D_HIP(c, hipMemcpyAsync(c->d_status, &fresh, sizeof(DeviceStatus), hipMemcpyHostToDevice, c->stream));
    return S2D_OK;
}

// The status word has been copied to h_status and the stream synchronised: act on it.
int judge_status(s2d_ctx* c);

int check_status(s2d_ctx* c)
{
    S2D_HIP(c, hipMemcpyAsync(c->h_status, c->d_status, sizeof(DeviceStatus), hipMemcpyDeviceToHost, c->stream));
    S2D_HIP(c, hipStreamSynchronize(c->stream));
    return judge_status(c);
}

int judge_status(s2d_ctx* c)
{
    if (c->h_status->nonfinite) {
        // The kernels queued behind the failing Adam step did nothing: put the host-side counters back to where the
        // device stopped (that step's update is the last thing that happened, as at the reference's abort()).
        const int k = c->h_status->first_nonfinite_iter;
        if (k >= c->good_iterations && k < c->iterations) {
            float b1 = c->good_beta1t, b2 = c->good_beta2t;
            for (int i = c->good_iterations; i <= k; i++) { b1 *= kAdamBeta1; b2 *= kAdamBeta2; } // main.cpp:718-719
            c->beta1t = b1;
            c->beta2t = b2;
            c->iterations = k + 1;
            c->have_forward = c->have_backward = false;
        }
        return fail(c, S2D_E_NONFINITE, "non-finite parameter after iteration %d (the reference abort()s, main.cpp:752-785)", k);
    }
    c->good_beta1t = c->beta1t;
    c->good_beta2t = c->beta2t;
    c->good_iterations = c->iterations;
    return S2D_OK;
}

double mse_norm(const s2d_ctx* c) { return (double)((long long)c->g.H * c->g.W * 3); }

size_t slab_pixels(const s2d_ctx* c) { return (size_t)c->g.W * (size_t)(c->g.row_end - c->g.row_begin); }

} // namespace

extern "C" {

int s2d_abi_version(void) { return S2D_ABI_VERSION; }

int s2d_create(const s2d_config* cfg, s2d_ctx** out)
{
    if (!cfg || !out || cfg->struct_size != sizeof(s2d_config)) return S2D_E_INVALID;
    *out = nullptr;
    if (cfg->width <= 0 || cfg->height <= 0 || cfg->n_splats < 0 || cfg->width > 65536 || cfg->height > 65536)
        return S2D_E_INVALID;
    int rb = cfg->row_begin, re = cfg->row_end;
    if (rb == 0 && re == 0) re = cfg->height;
    if (rb < 0 || re > cfg->height || rb >= re || (rb % kTile) != 0) return S2D_E_INVALID;
    if ((cfg->flags & S2D_CFG_EXACT_EXP) && (cfg->flags & (S2D_CFG_COUNT_PAIRS | S2D_CFG_FP16_IMAGES))) return S2D_E_INVALID;

    s2d_ctx* c = new (std::nothrow) s2d_ctx();
    if (!c) return S2D_E_NOMEM;
    *out = c; // handed out even on failure so that s2d_last_error works; caller destroys it
    c->cfg = *cfg;
    c->n = cfg->n_splats;
    c->device = cfg->device;
    c->lr = cfg->training_rate > 0.0f ? cfg->training_rate : 0.05f; // main.cpp:715
    c->rebin_interval = cfg->rebin_interval > 0 ? cfg->rebin_interval : INT_MAX; // default: rebuild on violation only
    c->margin = c->rebin_interval > 1 ? (cfg->rebin_margin > 0.0f ? cfg->rebin_margin : 2.0f) : 0.0f;

    Geometry& g = c->g;
    g.W = cfg->width; g.H = cfg->height;
    g.row_begin = rb; g.row_end = re;
    g.tiles_x = (g.W + kTile - 1) / kTile;
    g.trow0 = rb / kTile;
    g.tiles_y = (re + kTile - 1) / kTile - g.trow0;
    g.num_tiles = g.tiles_x * g.tiles_y;

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(c, S2D_E_HIP, "no HIP device available (%s): this library has no CPU fallback", hipGetErrorString(e));
    if (c->device < 0 || c->device >= ndev) return fail(c, S2D_E_INVALID, "device %d out of range (%d devices)", c->device, ndev);
    S2D_HIP(c, hipSetDevice(c->device));
    if (cfg->stream) {
        c->stream = (hipStream_t)cfg->stream;
    } else {
        S2D_HIP(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }

    const size_t n = std::max<size_t>((size_t)c->n, 1);           // >= 1 so that n == 0 still has buffers
    const size_t px = (size_t)g.W * (size_t)(g.row_end - g.row_begin); // pixels of the slab: all this context stores
    S2D_HIP(c, dev_alloc(&c->d_splats, n * 9));
    S2D_HIP(c, dev_alloc(&c->d_adams, n * 18));
    S2D_HIP(c, dev_alloc(&c->d_dormant, n));
    S2D_HIP(c, dev_alloc(&c->d_grads_own, n * 9));
    c->d_grads = c->d_grads_own;
    S2D_HIP(c, dev_alloc(&c->d_proj, n));
    S2D_HIP(c, dev_alloc(&c->d_rects, n));
    S2D_HIP(c, dev_alloc(&c->d_counts, n));
    S2D_HIP(c, dev_alloc(&c->d_offsets, n));
    S2D_HIP(c, dev_alloc(&c->d_scan_temp, scan_temp_words((int64_t)n)));
    S2D_HIP(c, dev_alloc(&c->d_total, 4)); // [0] pairs, [1] (splat, tile row) entries
    S2D_HIP(c, dev_alloc(&c->d_tile_off, (size_t)g.num_tiles + 1));
    S2D_HIP(c, dev_alloc(&c->d_tile_first, ((size_t)1 << key_bits_for(g.num_tiles)) + tile_first_temp_words(g.num_tiles))); // + chunk minima
    c->two_level = g.tiles_x <= kTlMaxColumns && !(cfg->flags & S2D_CFG_GENERIC_BINNING);
    if (const char* e = getenv("S2D_COMPACT_HELD")) c->compact_enabled = atoi(e) != 0;
    if (const char* e = getenv("S2D_CHUNK_PAIRS")) { // pairs per index range (tests; default 2^30, never beyond 32-bit positions)
        const unsigned long long v = strtoull(e, nullptr, 10);
        if (v > 0) c->chunk_pairs = std::min<unsigned long long>(v, 0xFFFF0000ull - 1);
    }
    if (c->two_level) {
        S2D_HIP(c, dev_alloc(&c->d_row_counts, n));
        S2D_HIP(c, dev_alloc(&c->d_row_offsets, n));
        S2D_HIP(c, dev_alloc(&c->d_row_off, (size_t)g.tiles_y + 1));
        S2D_HIP(c, dev_alloc(&c->d_chunk_base, (size_t)g.tiles_y + 1));
    }
    c->deterministic = (cfg->flags & S2D_CFG_DETERMINISTIC) != 0;
    if (c->deterministic) {
        S2D_HIP(c, dev_alloc(&c->d_det_touched, n));
        S2D_HIP(c, hipMemset(c->d_det_touched, 0, n * sizeof(uint32_t)));
    }
    c->half_images = (cfg->flags & S2D_CFG_FP16_IMAGES) != 0;
    c->pixel_bytes = c->half_images ? 8 : sizeof(float4);
    S2D_HIP(c, hipMalloc(&c->d_image0, px * c->pixel_bytes));
    S2D_HIP(c, hipMalloc(&c->d_ref, px * c->pixel_bytes));
    S2D_HIP(c, dev_alloc(&c->d_tile_sqerr, (size_t)g.num_tiles + kSqerrScratchDoubles)); // + finalize scratch
    S2D_HIP(c, hipMemset(c->d_tile_sqerr + g.num_tiles, 0, kSqerrScratchDoubles * sizeof(double)));
    S2D_HIP(c, dev_alloc(&c->d_sqerr_trace, (size_t)c->trace_cap));
    S2D_HIP(c, dev_alloc(&c->d_status, 1));
    S2D_HIP(c, dev_alloc(&c->d_counters, 1));
    S2D_HIP(c, hipEventCreateWithFlags(&c->ev_flag, hipEventDisableTiming));
    S2D_HIP(c, hipEventCreateWithFlags(&c->ev_total, hipEventDisableTiming));
    S2D_HIP(c, hipHostMalloc((void**)&c->h_total, 64, hipHostMallocMapped));
    S2D_HIP(c, hipHostMalloc((void**)&c->h_status, sizeof(DeviceStatus), hipHostMallocDefault));
    S2D_HIP(c, hipHostMalloc((void**)&c->h_trace, kHostTrace * sizeof(double), hipHostMallocDefault));
    S2D_HIP(c, hipHostMalloc((void**)&c->h_rebin_stamp, 64, hipHostMallocMapped));
    *c->h_rebin_stamp = 0;

    S2D_HIP(c, hipMemsetAsync(c->d_splats, 0, n * 9 * sizeof(float), c->stream));
    S2D_HIP(c, hipMemsetAsync(c->d_adams, 0, n * 18 * sizeof(float), c->stream));
    S2D_HIP(c, hipMemsetAsync(c->d_dormant, 0, n, c->stream));
    S2D_HIP(c, hipMemsetAsync(c->d_grads_own, 0, n * 9 * sizeof(float), c->stream));
    S2D_HIP(c, hipMemsetAsync(c->d_image0, 0, px * c->pixel_bytes, c->stream));
    S2D_HIP(c, hipMemsetAsync(c->d_ref, 0, px * c->pixel_bytes, c->stream));
    S2D_HIP(c, hipMemsetAsync(c->d_sqerr_trace, 0, (size_t)c->trace_cap * sizeof(double), c->stream));
    S2D_HIP(c, hipMemsetAsync(c->d_counters, 0, sizeof(PairCounters), c->stream));
    DeviceStatus st0{0, INT_MAX, 0, 0};
    *c->h_status = st0;
    S2D_HIP(c, hipMemcpyAsync(c->d_status, c->h_status, sizeof(DeviceStatus), hipMemcpyHostToDevice, c->stream));
    S2D_HIP(c, hipStreamSynchronize(c->stream));
    // ~16 tiles per splat at init() scales; grown on demand
    int rc = ensure_pair_capacity(c, std::max<uint64_t>((uint64_t)n * 20, 1 << 16));
    if (rc != S2D_OK) return rc;
    return S2D_OK;
}

void s2d_destroy(s2d_ctx* c)
{
    if (!c) return;
    if (hipSetDevice(c->device) == hipSuccess) {
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        void* ptrs[] = {c->d_splats, c->d_adams, c->d_dormant, c->d_grads_own, c->d_proj, c->d_rects, c->d_counts, c->d_offsets,
                        c->d_scan_temp, c->d_total, c->d_keys[0], c->d_keys[1], c->d_vals[0], c->d_vals[1],
                        c->d_sort_temp, c->d_wave_masks, c->d_det_data, c->d_det_stamp, c->d_det_touched, c->d_tile_off, c->d_tile_first, c->d_row_counts, c->d_row_offsets, c->d_row_off,
                        c->d_chunk_base, c->d_tl_hist, c->d_image0, c->d_ref, c->d_tile_sqerr, c->d_held, c->d_held_ids, c->d_held_count, c->d_held_work, c->d_sqerr_trace,
                        c->d_status, c->d_counters, c->d_state, c->d_chunk_alive, c->d_csplats, c->d_cadams};
        for (void* p : ptrs)
            if (p) (void)hipFree(p);
        if (c->ev_flag) (void)hipEventDestroy(c->ev_flag);
        if (c->ev_total) (void)hipEventDestroy(c->ev_total);
        if (c->h_total) (void)hipHostFree(c->h_total);
        if (c->h_status) (void)hipHostFree(c->h_status);
        if (c->h_trace) (void)hipHostFree(c->h_trace);
        if (c->h_rebin_stamp) (void)hipHostFree(c->h_rebin_stamp);
        if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    }
    delete c;
}

const char* s2d_last_error(const s2d_ctx* c) { return c ? c->err : "null context"; }

int s2d_set_target(s2d_ctx* c, const float* rgba32f)
{
    if (!c || !rgba32f) return S2D_E_INVALID;
    if (int rc = use_device(c)) return rc;
    // the caller hands over the whole image (main.cpp:254-259); a slab context uploads and keeps its own rows only
    const size_t px = slab_pixels(c), bytes = px * sizeof(float4);
    const float* src = rgba32f + (size_t)c->g.row_begin * c->g.W * 4;
    if (c->half_images) { // floats cross the boundary; the device keeps them as fp16 (round to nearest even)
        float4* tmp = nullptr;
        S2D_HIP(c, hipMalloc((void**)&tmp, bytes));
        hipError_t e = hipMemcpyAsync(tmp, src, bytes, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = launch_convert_f32_to_f16(tmp, c->d_ref, px, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        (void)hipFree(tmp);
        S2D_HIP(c, e);
    } else {
        S2D_HIP(c, hipMemcpyAsync(c->d_ref, src, bytes, hipMemcpyHostToDevice, c->stream));
        S2D_HIP(c, hipStreamSynchronize(c->stream));
    }
    c->have_target = true;
    c->have_forward = c->have_backward = false;
    return S2D_OK;
}

int s2d_set_target_synthetic(s2d_ctx* c)
{
    if (!c) return S2D_E_INVALID;
    if (int rc = use_device(c)) return rc;
    S2D_HIP(c, launch_synthetic_target(c->d_ref, c->half_images, c->g.W, c->g.H, c->g.row_begin, c->g.row_end, c->stream));
    c->have_target = true;
    c->have_forward = c->have_backward = false;
    return S2D_OK;
}

int s2d_init_splats(s2d_ctx* c)
{
    if (!c) return S2D_E_INVALID;
    if (int rc = use_device(c)) return rc;
    if (int rc = flush_sqerr(c)) return rc;
    S2D_HIP(c, launch_init_splats(c->d_splats, c->d_adams, c->n, c->g.W, c->g.H, c->stream));
    if (int rc = compact_load(c)) return rc; // (every record is new: nothing of the old compact copy is worth flushing)
    if (int rc = clear_dormant(c)) return rc;
    if (c->n > 0) S2D_HIP(c, hipMemsetAsync(c->d_grads, 0, (size_t)c->n * 9 * sizeof(float), c->stream));
    if (int rc = reset_status(c)) return rc;
    c->beta1t = c->good_beta1t = 1.0f; // main.cpp:283-284
    c->beta2t = c->good_beta2t = 1.0f;
    c->iterations = c->good_iterations = 0; // main.cpp:281
    c->lists_valid = false;
    c->proj_fresh = false;
    c->have_forward = c->have_backward = false;
    return S2D_OK;
}

int s2d_set_splats(s2d_ctx* c, const s2d_splat* splats)
{
    if (!c || (!splats && c->n)) return S2D_E_INVALID;
    if (int rc = use_device(c)) return rc;
    if (int rc = compact_flush(c)) return rc; // the moments of the held splats must not be lost with the compact copy
    S2D_HIP(c, hipMemcpyAsync(c->d_splats, splats, (size_t)c->n * sizeof(s2d_splat), hipMemcpyHostToDevice, c->stream));
    if (int rc = compact_load(c)) return rc;
    if (int rc = clear_dormant(c)) return rc;
    if (int rc = reset_status(c)) return rc;
    S2D_HIP(c, hipStreamSynchronize(c->stream));
    c->lists_valid = false;
    c->proj_fresh = false;
    c->have_forward = c->have_backward = false;
    return S2D_OK;
}

int s2d_get_splats(s2d_ctx* c, s2d_splat* splats)
{
    if (!c || (!splats && c->n)) return S2D_E_INVALID;
    if (int rc = use_device(c)) return rc;
    if (int rc = compact_flush(c)) return rc;
    S2D_HIP(c, hipMemcpyAsync(splats, c->d_splats, (size_t)c->n * sizeof(s2d_splat), hipMemcpyDeviceToHost, c->stream));
    S2D_HIP(c, hipStreamSynchronize(c->stream));
    return S2D_OK;
}

int s2d_set_adam(s2d_ctx* c, const s2d_splat_adam* adams, float beta1t, float beta2t, int32_t iterations)
{
    if (!c || (!adams && c->n) || iterations < 0) return S2D_E_INVALID;
    if (int rc = use_device(c)) return rc;
    if (int rc = flush_sqerr(c)) return rc; // (its ring slot is named by the iteration count about to change)
    if (int rc = compact_flush(c)) return rc; // the parameters of the held splats must not be lost with the compact copy
    S2D_HIP(c, hipMemcpyAsync(c->d_adams, adams, (size_t)c->n * sizeof(s2d_splat_adam), hipMemcpyHostToDevice, c->stream));
    if (int rc = compact_load(c)) return rc;
    if (int rc = clear_dormant(c)) return rc;
    S2D_HIP(c, hipStreamSynchronize(c->stream));
    c->beta1t = c->good_beta1t = beta1t;
    c->beta2t = c->good_beta2t = beta2t;
    c->iterations = c->good_iterations = iterations;
    return S2D_OK;
}

int s2d_get_adam(s2d_ctx* c, s2d_splat_adam* adams, float* beta1t, float* beta2t, int32_t* iterations)
{
    if (!c) return S2D_E_INVALID;
    if (int rc = use_device(c)) return rc;
    if (adams) {
        if (int rc = compact_flush(c)) return rc;
        S2D_HIP(c, hipMemcpyAsync(adams, c->d_adams, (size_t)c->n * sizeof(s2d_splat_adam), hipMemcpyDeviceToHost, c->stream));
        S2D_HIP(c, hipStreamSynchronize(c->stream));
    }
    if (beta1t) *beta1t = c->beta1t;
    if (beta2t) *beta2t = c->beta2t;
    if (iterations) *iterations = c->iterations;
    return S2D_OK;
}

int s2d_forward(s2d_ctx* c)
{
    if (!c) return S2D_E_INVALID;
    if (int rc = use_device(c)) return rc;
    return queue_forward(c);
}

int s2d_get_image_rows(s2d_ctx* c, float* rgba32f_rows)
{
    if (!c || !rgba32f_rows) return S2D_E_INVALID;
    if (int rc = use_device(c)) return rc;
    const size_t px = slab_pixels(c), bytes = px * sizeof(float4);
    if (c->half_images) {
        float4* tmp = nullptr;
        S2D_HIP(c, hipMalloc((void**)&tmp, bytes));
        hipError_t e = launch_convert_f16_to_f32(c->d_image0, tmp, px, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(rgba32f_rows, tmp, bytes, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        (void)hipFree(tmp);
        S2D_HIP(c, e);
        return S2D_OK;
    }
    S2D_HIP(c, hipMemcpyAsync(rgba32f_rows, c->d_image0, bytes, hipMemcpyDeviceToHost, c->stream));
    S2D_HIP(c, hipStreamSynchronize(c->stream));
    return S2D_OK;
}

int s2d_get_image(s2d_ctx* c, float* rgba32f)
{
    if (!c || !rgba32f) return S2D_E_INVALID;
    // a full-size image goes back (main.cpp:794 uploads all of image0): this context's rows, zeros elsewhere
    const size_t row_floats = (size_t)c->g.W * 4;
    std::memset(rgba32f, 0, (size_t)c->g.row_begin * row_floats * sizeof(float));
    std::memset(rgba32f + (size_t)c->g.row_end * row_floats, 0, (size_t)(c->g.H - c->g.row_end) * row_floats * sizeof(float));
    return s2d_get_image_rows(c, rgba32f + (size_t)c->g.row_begin * row_floats);
}

int s2d_forward_backward(s2d_ctx* c, uint32_t flags)
{
    if (!c) return S2D_E_INVALID;
    if (int rc = use_device(c)) return rc;
    return queue_forward_backward(c, !(flags & S2D_BWD_SKIP_OPACITY_GRAD), !(flags & S2D_FB_SKIP_IMAGE));
}

int s2d_backward(s2d_ctx* c, uint32_t flags)
{
    if (!c) return S2D_E_INVALID;
    if (int rc = use_device(c)) return rc;
    return queue_backward(c, !(flags & S2D_BWD_SKIP_OPACITY_GRAD));
}

int s2d_get_grads(s2d_ctx* c, s2d_splat* dsplats)
{
    if (!c || (!dsplats && c->n)) return S2D_E_INVALID;
    if (int rc = use_device(c)) return rc;
    S2D_HIP(c, hipMemcpyAsync(dsplats, c->d_grads, (size_t)c->n * sizeof(s2d_splat), hipMemcpyDeviceToHost, c->stream));
    S2D_HIP(c, hipStreamSynchronize(c->stream));
    return S2D_OK;
}

int s2d_adam_step(s2d_ctx* c, uint32_t flags)
{
    if (!c) return S2D_E_INVALID;
    if (int rc = use_device(c)) return rc;
    if (int rc = queue_adam(c, flags)) return rc;
    return S2D_OK;
}

int s2d_step(s2d_ctx* c, int32_t iters, uint32_t flags, double* mse_out)
{
    if (!c || iters < 0) return S2D_E_INVALID;
    if (int rc = use_device(c)) return rc;
    const double norm = mse_norm(c);
    const int call_first_iter = c->iterations;
    int done = 0;
    bool status_read = false;
    while (done < iters) {
        const int chunk = std::min(iters - done, c->trace_cap);
        const int first_iter = c->iterations;
        for (int k = 0; k < chunk; k++) {
            // image0 is stored by the last iteration of the call only: nothing else could observe the others
            const bool last = done + k + 1 == iters;
            if (int rc = queue_forward_backward(c, (flags & S2D_STEP_OPTIMIZE_OPACITY) != 0, last, true)) return rc;
            if (int rc = queue_adam(c, flags)) return rc;
        }
        const bool last_chunk = done + chunk == iters;
        if (mse_out && last_chunk && chunk <= kHostTrace) {
            // the usual call (a frame, or a batch of frames, of the host loop): trace and status word in one round trip
            if (int rc = flush_sqerr(c)) return rc;
            for (int got = 0; got < chunk;) {
                const int slot = (first_iter + got) % c->trace_cap, run = std::min(chunk - got, c->trace_cap - slot);
                S2D_HIP(c, hipMemcpyAsync(c->h_trace + got, c->d_sqerr_trace + slot, (size_t)run * sizeof(double),
                                          hipMemcpyDeviceToHost, c->stream));
                got += run;
            }
            S2D_HIP(c, hipMemcpyAsync(c->h_status, c->d_status, sizeof(DeviceStatus), hipMemcpyDeviceToHost, c->stream));
            S2D_HIP(c, hipStreamSynchronize(c->stream));
            for (int k = 0; k < chunk; k++) mse_out[done + k] = c->h_trace[k] / norm; // main.cpp:805
            status_read = true;
        } else if (mse_out) {
            if (int rc = s2d_get_sqerr_trace(c, first_iter, chunk, mse_out + done)) return rc;
            for (int k = 0; k < chunk; k++) mse_out[done + k] /= norm; // main.cpp:805
        }
        done += chunk;
    }
    const int rc = status_read ? judge_status(c) : check_status(c);
    if (rc == S2D_E_NONFINITE && mse_out) {
        // The reference abort()s right after the Adam step of that iteration (main.cpp:752-785): its trace ends with
        // that iteration's line.  The kernels of the later iterations queued here did nothing; their entries are NaN.
        const int last_valid = c->h_status->first_nonfinite_iter - call_first_iter;
        for (int k = std::max(last_valid + 1, 0); k < iters; k++) mse_out[k] = std::nan("");
    }
    return rc;
}

int s2d_get_mse(s2d_ctx* c, double* mse)
{
    if (!c || !mse) return S2D_E_INVALID;
    if (int rc = use_device(c)) return rc;
    if (c->last_sqerr_slot < 0) return fail(c, S2D_E_STATE, "no backward pass has run yet");
    if (int rc = flush_sqerr(c)) return rc;
    double v = 0.0;
    S2D_HIP(c, hipMemcpyAsync(&v, c->d_sqerr_trace + c->last_sqerr_slot, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    S2D_HIP(c, hipStreamSynchronize(c->stream));
    *mse = v / mse_norm(c);
    return S2D_OK;
}

int s2d_bind_grads_device(s2d_ctx* c, void* grads_device)
{
    if (!c) return S2D_E_INVALID;
    if ((uintptr_t)grads_device & 15u) return fail(c, S2D_E_INVALID, "the gradient buffer must be 16-byte aligned");
    c->d_grads = grads_device ? (float*)grads_device : c->d_grads_own;
    return S2D_OK;
}

void* s2d_grads_device_ptr(s2d_ctx* c) { return c ? (void*)c->d_grads : nullptr; }

void* s2d_stream(s2d_ctx* c) { return c ? (void*)c->stream : nullptr; }

// ---- slab ownership (s2d_halo.hip): all pointers below are device pointers of the caller ----

int s2d_halo_masks(s2d_ctx* c, int32_t world, const int32_t* row_bounds, float margin_rows, uint32_t* masks_device)
{
    if (!c || !row_bounds || !masks_device || world < 1 || world > 32 || !(margin_rows >= 0.0f)) return S2D_E_INVALID;
    for (int q = 0; q < world; q++)
        if (row_bounds[q] > row_bounds[q + 1]) return S2D_E_INVALID;
    if (int rc = use_device(c)) return rc;
    if (int rc = compact_flush(c)) return rc;
    S2D_HIP(c, launch_halo_masks(c->d_splats, c->d_held, c->n, world, row_bounds, margin_rows, masks_device, c->stream));
    return S2D_OK;
}

int s2d_halo_commit(s2d_ctx* c, const uint32_t* masks_device, int32_t rank, int32_t added)
{
    if (!c || rank < 0 || rank > 31) return S2D_E_INVALID;
    if (int rc = use_device(c)) return rc;
    if (int rc = compact_flush(c)) return rc; // the held set is about to change: the id-indexed arrays take over
    c->compact_live = false;
    if (!masks_device) { // back to holding every splat (the caller has made this context's copy complete again)
        if (c->d_held) {
            S2D_HIP(c, hipStreamSynchronize(c->stream));
            void* ptrs[] = {c->d_held, c->d_held_ids, c->d_held_work, c->d_held_count};
            for (void* q : ptrs) (void)hipFree(q);
            c->d_held = nullptr;
            c->d_held_ids = c->d_held_work = c->d_held_count = nullptr;
            c->lists_valid = false;
            c->proj_fresh = false;
            c->have_forward = c->have_backward = false;
        }
        return S2D_OK;
    }
    const bool first = c->d_held == nullptr;
    if (first) {
        S2D_HIP(c, dev_alloc(&c->d_held, (size_t)c->n));
        S2D_HIP(c, dev_alloc(&c->d_held_ids, (size_t)c->n));
        S2D_HIP(c, dev_alloc(&c->d_held_work, (size_t)c->n));
        S2D_HIP(c, dev_alloc(&c->d_held_count, 4));
    }
    S2D_HIP(c, launch_halo_commit(masks_device, c->n, rank, c->d_held, c->d_held_ids, c->d_held_count, c->d_held_work,
                                  c->d_scan_temp, c->stream));
    if (int rc = compact_load(c)) return rc;
    if (added || first) {
        // splats arrived: project the held ones and rebuild the tile lists before the next forward
        c->lists_valid = false;
        c->proj_fresh = false;
        c->have_forward = false;
        c->have_backward = false;
    }
    return S2D_OK;
}

static int rows_base(s2d_ctx* c, int32_t what, float** base, int* w)
{
    switch (what) {
    case S2D_ROWS_GRADS: *base = c->d_grads; *w = 9; return S2D_OK;
    case S2D_ROWS_SPLATS: *base = c->d_splats; *w = 9; return S2D_OK;
    case S2D_ROWS_ADAM: *base = c->d_adams; *w = 18; return S2D_OK;
    default: return S2D_E_INVALID;
    }
}

int s2d_rows_gather(s2d_ctx* c, int32_t what, const int32_t* ids_device, int32_t count, float* out_device)
{
    if (!c || count < 0 || (count > 0 && (!ids_device || !out_device))) return S2D_E_INVALID;
    float* base;
    int w;
    if (int rc = rows_base(c, what, &base, &w)) return rc;
    if (int rc = use_device(c)) return rc;
    if (what != S2D_ROWS_GRADS)
        if (int rc = compact_flush(c)) return rc;
    S2D_HIP(c, launch_rows_gather(base, w, ids_device, count, c->n, out_device, c->stream));
    return S2D_OK;
}

int s2d_rows_scatter(s2d_ctx* c, int32_t what, const int32_t* ids_device, int32_t count, const float* in_device)
{
    if (!c || count < 0 || (count > 0 && (!ids_device || !in_device))) return S2D_E_INVALID;
    float* base;
    int w;
    if (int rc = rows_base(c, what, &base, &w)) return rc;
    if (int rc = use_device(c)) return rc;
    if (what != S2D_ROWS_GRADS)
        if (int rc = compact_flush(c)) return rc;
    S2D_HIP(c, launch_rows_scatter(base, w, ids_device, count, c->n, in_device, c->stream));
    if (what == S2D_ROWS_SPLATS || what == S2D_ROWS_ADAM) {
        if (c->compact_live)
            if (int rc = compact_load(c)) return rc; // rows of held splats may be among them
        if (int rc = clear_dormant(c)) return rc;
    }
    if (what == S2D_ROWS_SPLATS) { // parameters changed behind the projection
        c->proj_fresh = false;
        c->have_forward = false;
        c->have_backward = false;
    }
    return S2D_OK;
}

int s2d_grads_combine(s2d_ctx* c, const int32_t* rows_device, int32_t n_rows, const int32_t* src_device, int32_t world,
                      const float* recv_device)
{
    if (!c || n_rows < 0 || world < 1 || world > 32 || (n_rows > 0 && (!rows_device || !src_device))) return S2D_E_INVALID;
    if (int rc = use_device(c)) return rc;
    S2D_HIP(c, launch_grads_combine(c->d_grads, rows_device, n_rows, src_device, world, recv_device, c->n, c->stream));
    return S2D_OK;
}

int s2d_get_sqerr_trace(s2d_ctx* c, int32_t first_iteration, int32_t count, double* out)
{
    if (!c || !out || count < 0 || first_iteration < 0 || count > c->trace_cap) return S2D_E_INVALID;
    if (int rc = use_device(c)) return rc;
    if (int rc = flush_sqerr(c)) return rc;
    int done = 0;
    while (done < count) {
        const int slot = (first_iteration + done) % c->trace_cap;
        const int run = std::min(count - done, c->trace_cap - slot);
        S2D_HIP(c, hipMemcpyAsync(out + done, c->d_sqerr_trace + slot, (size_t)run * sizeof(double),
                                  hipMemcpyDeviceToHost, c->stream));
        done += run;
    }
    S2D_HIP(c, hipStreamSynchronize(c->stream));
    return S2D_OK;
}

int s2d_synchronize(s2d_ctx* c)
{
    if (!c) return S2D_E_INVALID;
    if (int rc = use_device(c)) return rc;
    return check_status(c);
}

int s2d_get_stats(s2d_ctx* c, s2d_stats* out)
{
    // the struct as its first version (ABI 2) ends with bwd_quadrant_execs: a caller must have at least that
    if (!c || !out || out->struct_size < (uint32_t)(offsetof(s2d_stats, bwd_quadrant_execs) + sizeof(uint64_t))) return S2D_E_INVALID;
    if (int rc = use_device(c)) return rc;
    PairCounters pc;
    S2D_HIP(c, hipMemcpyAsync(&pc, c->d_counters, sizeof(pc), hipMemcpyDeviceToHost, c->stream));
    S2D_HIP(c, hipMemcpyAsync(c->h_status, c->d_status, sizeof(DeviceStatus), hipMemcpyDeviceToHost, c->stream));
    S2D_HIP(c, hipStreamSynchronize(c->stream));
    // filled in a full-size copy, handed back at the caller's size: a caller built against an older, shorter struct is
    // never written past its end
    const uint32_t caller_size = std::min<uint32_t>(out->struct_size, (uint32_t)sizeof(s2d_stats));
    s2d_stats full;
    s2d_stats* const dst = out;
    out = &full;
    std::memset(out, 0, sizeof(*out));
    out->struct_size = caller_size;
    out->pairs_binned = c->pairs;
    out->pairs_capacity = c->pair_capacity;
    out->rebins = c->rebins;
    out->fwd_visited = pc.fwd_visited; out->fwd_active = pc.fwd_active;
    out->bwd_visited = pc.bwd_visited; out->bwd_active = pc.bwd_active;
    out->fwd_staged = pc.fwd_staged; out->bwd_staged = pc.bwd_staged;
    out->fwd_wave_execs = pc.fwd_wave_execs; out->bwd_wave_execs = pc.bwd_wave_execs;
    for (int k = 0; k < 65; k++) out->bwd_lane_hist[k] = pc.bwd_lane_hist[k];
    out->fwd_staged_hit = pc.fwd_staged_hit;
    out->fwd_rows_hit = pc.fwd_rows_hit;
    out->bwd_quadrant_execs = pc.bwd_quadrant_execs;
    out->iterations = c->iterations;
    out->first_nonfinite_iteration = c->h_status->nonfinite ? c->h_status->first_nonfinite_iter : -1;
    std::memcpy(dst, &full, caller_size);
    return S2D_OK;
}

int s2d_get_rebuild_count(const s2d_ctx* c, uint64_t* rebuilds)
{
    if (!c || !rebuilds) return S2D_E_INVALID;
    *rebuilds = c->rebins;
    return S2D_OK;
}

int s2d_debug_get_tile_lists(s2d_ctx* c, int32_t* tiles_x, int32_t* tiles_y, uint32_t* offsets,
                             int64_t offsets_capacity, uint32_t* list, int64_t list_capacity)
{
    if (!c) return S2D_E_INVALID;
    if (int rc = use_device(c)) return rc;
    if (!c->lists_valid) return fail(c, S2D_E_STATE, "tile lists not built yet (run s2d_forward)");
    if (tiles_x) *tiles_x = c->g.tiles_x;
    if (tiles_y) *tiles_y = c->g.tiles_y;
    if (offsets) {
        if (offsets_capacity < c->g.num_tiles + 1) return S2D_E_INVALID;
        S2D_HIP(c, hipMemcpyAsync(offsets, c->d_tile_off, (size_t)(c->g.num_tiles + 1) * sizeof(uint32_t),
                                  hipMemcpyDeviceToHost, c->stream));
    }
    if (list) {
        if (list_capacity < (int64_t)c->pairs) return S2D_E_INVALID;
        S2D_HIP(c, hipMemcpyAsync(list, c->d_list, (size_t)c->pairs * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    }
    S2D_HIP(c, hipStreamSynchronize(c->stream));
    return S2D_OK;
}

// ---- test hooks -----------------------------------------------------------------------------------
#define S2D_HIP0(expr)                         \
    do {                                       \
        if ((expr) != hipSuccess) { rc = S2D_E_HIP; goto done; } \
    } while (0)

int s2d_test_sincos(int32_t device, const float* x, int32_t n, float* sin_out, float* cos_out)
{
    if (!x || !sin_out || !cos_out || n < 0) return S2D_E_INVALID;
    int rc = S2D_OK;
    float *dx = nullptr, *ds = nullptr, *dc = nullptr;
    const size_t bytes = std::max<size_t>(n, 1) * sizeof(float);
    S2D_HIP0(hipSetDevice(device));
    S2D_HIP0(hipMalloc((void**)&dx, bytes));
    S2D_HIP0(hipMalloc((void**)&ds, bytes));
    S2D_HIP0(hipMalloc((void**)&dc, bytes));
    S2D_HIP0(hipMemcpy(dx, x, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    S2D_HIP0(launch_test_sincos(dx, n, ds, dc, nullptr));
    S2D_HIP0(hipDeviceSynchronize());
    S2D_HIP0(hipMemcpy(sin_out, ds, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    S2D_HIP0(hipMemcpy(cos_out, dc, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
done:
    if (dx) (void)hipFree(dx);
    if (ds) (void)hipFree(ds);
    if (dc) (void)hipFree(dc);
    return rc;
}

int s2d_test_sort_pairs(int32_t device, uint32_t* keys, uint32_t* values, int64_t n, int32_t key_bits)
{
    if (!keys || !values || n < 0 || key_bits < 0 || key_bits > 32) return S2D_E_INVALID;
    int rc = S2D_OK;
    uint32_t *k[2] = {nullptr, nullptr}, *v[2] = {nullptr, nullptr}, *temp = nullptr, *ko = nullptr, *vo = nullptr;
    const size_t bytes = std::max<size_t>((size_t)n, 1) * sizeof(uint32_t);
    S2D_HIP0(hipSetDevice(device));
    for (int i = 0; i < 2; i++) {
        S2D_HIP0(hipMalloc((void**)&k[i], bytes));
        S2D_HIP0(hipMalloc((void**)&v[i], bytes));
    }
    S2D_HIP0(hipMalloc((void**)&temp, sort_temp_words(n) * sizeof(uint32_t)));
    S2D_HIP0(hipMemcpy(k[0], keys, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice));
    S2D_HIP0(hipMemcpy(v[0], values, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice));
    S2D_HIP0(sort_pairs_u32(k[0], v[0], k[1], v[1], n, key_bits, temp, &ko, &vo, nullptr, nullptr));
    S2D_HIP0(hipDeviceSynchronize());
    S2D_HIP0(hipMemcpy(keys, ko, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    S2D_HIP0(hipMemcpy(values, vo, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
done:
    for (int i = 0; i < 2; i++) {
        if (k[i]) (void)hipFree(k[i]);
        if (v[i]) (void)hipFree(v[i]);
    }
    if (temp) (void)hipFree(temp);
    return rc;
}

int s2d_test_exclusive_scan(int32_t device, uint32_t* data, int64_t n, uint64_t* total)
{
    if (!data || n < 0) return S2D_E_INVALID;
    int rc = S2D_OK;
    uint32_t *d = nullptr, *temp = nullptr, *tot = nullptr, htot = 0;
    S2D_HIP0(hipSetDevice(device));
    S2D_HIP0(hipMalloc((void**)&d, std::max<size_t>((size_t)n, 1) * sizeof(uint32_t)));
    S2D_HIP0(hipMalloc((void**)&temp, scan_temp_words(n) * sizeof(uint32_t)));
    S2D_HIP0(hipMalloc((void**)&tot, sizeof(uint32_t)));
    S2D_HIP0(hipMemcpy(d, data, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice));
    S2D_HIP0(exclusive_scan_u32(d, d, n, temp, tot, nullptr));
    S2D_HIP0(hipDeviceSynchronize());
    S2D_HIP0(hipMemcpy(data, d, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    S2D_HIP0(hipMemcpy(&htot, tot, sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (total) *total = htot;
done:
    if (d) (void)hipFree(d);
    if (temp) (void)hipFree(temp);
    if (tot) (void)hipFree(tot);
    return rc;
}

} // extern "C"
