// s2d_raster.hip -- tile-binned forward rasteriser and analytic backward pass.
//
// One 256-thread workgroup (4 wave64) per 16x16 image tile, one thread per pixel.  Each wave owns an 8x8 pixel
// block of the tile, lane = 8 * (row in block) + (column in block).  A tile walks its splat list (ascending splat index ==
// the reference's blend order, main.cpp:419/:552) twice -- forward_tile, then backward_tile -- in batches of 64 entries
// staged in LDS:
//   * forward walk: 4 threads per entry load the 64-byte projected record and evaluate the reference's exact per-row
//     column range (solve_quadratic + int truncation, main.cpp:498-509) for 4 tile rows each, producing one
//     64-bit lane mask per (entry, wave): bit l set <=> the reference's loops visit that pixel for that
//     splat.  The quadratic is solved once per (entry, row), not per pixel.  The masks are also stored (32 B per
//     staged pair); the backward walk, which goes through exactly the same batches, loads them instead of solving
//     the quadratics again.
//   * after the barrier lane l of every wave fetches the mask of entry l; a ballot of "mask touches a live pixel" is
//     the set of entries this wave has to look at, and the blend loop iterates over its set bits only
//     (scalar bit tricks + v_readlane), skipping an entry with a scalar branch when no lane is live any more.
// A pixel whose throughput fell below 1/256 never works again (main.cpp:520); when all 256 pixels of the tile
// are in that state the workgroup stops walking its list (tile retirement).
//
// Kernels: raster_fused_kernel runs both walks of a tile in one launch, the final colours staying in registers
// (what s2d_step and s2d_forward_backward queue); raster_forward_kernel / raster_backward_kernel run one walk each
// through image0 (s2d_forward / s2d_backward, and the counting diagnostics).
//
// The alpha / throughput / colour arithmetic is the reference's, operation for operation
// (-ffp-contract=off), in both walks, so the framebuffer is bit-identical to the oracle's and the backward
// walk makes exactly the forward's per-pixel decisions.  The backward walk then sums each splat's nine
// partial gradients over the wave through an LDS transpose (wave_sum8_lds), over the four waves in one LDS slot per
// entry, and issues one global float-atomic burst per (tile, splat) into the N x 9 gradient array -- or,
// with S2D_CFG_DETERMINISTIC, keeps one slot per wave, adds them in a fixed order and stores the partial into the
// tile's own slot of that splat, and a gather kernel sums each splat's slots in a fixed order (bitwise
// reproducible gradients).
#include <hip/hip_fp16.h>

#include "s2d_device.h"

namespace s2d {

constexpr int B = kRasterBatch;
static_assert(B == 64, "lane l of a wave holds the mask of batch entry l");

constexpr int kWaveW = 8;              // pixel columns per wave block (8x8 blocks: measured better than 16x4 strips,
constexpr int kWaveH = 64 / kWaveW;    // profiles/r01: 34 vs 37 executed entries per wave, 36 vs 33 active lanes)
constexpr int kWavesX = kTile / kWaveW;

// Block -> tile: dispatch order.  (An XCD-contiguous remap and a longest-list-first order were measured slower,
// profiles/r01/README.md: these kernels are VALU-issue-bound, not L2-bound.)
__device__ __forceinline__ int tile_of_block(int bid, const Geometry& g) { return bid < g.num_tiles ? bid : -1; }

__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int src_lane)
{
    const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, src_lane);
    const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), src_lane);
    return ((unsigned long long)hi << 32) | lo;
}

// Framebuffer / target pixel in HBM: RGBA32F (the reference's Image2DRGBA32, 16 B) or, with S2D_CFG_FP16_IMAGES,
// four IEEE halves (8 B, round-to-nearest-even on store).  Arithmetic is fp32 either way.
template <bool HALF>
__device__ __forceinline__ float4 load_pixel(const void* base, size_t i)
{
    if (HALF) {
        const uint2 v = reinterpret_cast<const uint2*>(base)[i];
        const float2 a = __half22float2(*reinterpret_cast<const __half2*>(&v.x));
        const float2 b = __half22float2(*reinterpret_cast<const __half2*>(&v.y));
        return make_float4(a.x, a.y, b.x, b.y);
    }
    return reinterpret_cast<const float4*>(base)[i];
}

template <bool HALF>
__device__ __forceinline__ void store_pixel(void* base, size_t i, float4 c)
{
    if (HALF) {
        const __half2 a = __floats2half2_rn(c.x, c.y), b = __floats2half2_rn(c.z, c.w);
        uint2 v;
        v.x = *reinterpret_cast<const uint32_t*>(&a);
        v.y = *reinterpret_cast<const uint32_t*>(&b);
        reinterpret_cast<uint2*>(base)[i] = v;
    } else {
        reinterpret_cast<float4*>(base)[i] = c;
    }
}

// A pair of fp32 values -- (vx,vy), (mx,my), the (r,g) colour channels, the two covariance dot products.  The blend
// and gradient arithmetic below is written on such pairs with every product and sum in the reference's order.
// A plain struct whose operators are two scalar VALU instructions each: on gfx950 a v_pk_{mul,add,fma}_f32 occupies a
// SIMD for ~4.3 cycles, a v_{mul,add,fma}_f32 for ~2.4 (tools/microbench/valu_rates.hip, profiles/r01/valu_rates.txt),
// so packed fp32 buys nothing and the register-pair assembly it needs costs (395 vs 413 it/s, profiles/r01/v8_*);
// the build passes -fno-slp-vectorize so LLVM does not re-pack these.
struct f2 {
    float x, y;
};
__device__ __forceinline__ f2 operator+(f2 a, f2 b) { return f2{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ f2 operator-(f2 a, f2 b) { return f2{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ f2 operator*(f2 a, f2 b) { return f2{a.x * b.x, a.y * b.y}; }
__device__ __forceinline__ f2 operator*(f2 a, float b) { return f2{a.x * b, a.y * b}; }
__device__ __forceinline__ f2 operator*(float a, f2 b) { return f2{a * b.x, a * b.y}; }
__device__ __forceinline__ f2 operator-(f2 a) { return f2{-a.x, -a.y}; }
__device__ __forceinline__ f2& operator+=(f2& a, f2 b) { a = a + b; return a; }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return f2{__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y)}; }
__device__ __forceinline__ f2 mk2(float a, float b)
{
    f2 r;
    r.x = a;
    r.y = b;
    return r;
}

// Diagnostic counters (S2D_CFG_COUNT_PAIRS): a per-lane count is summed over the wave first and added by one lane (the build
// switches LLVM's atomic optimizer off, _build.py: nothing does this behind our back any more).
__device__ __forceinline__ void count_add(unsigned long long* counter, unsigned long long v, int lane)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d, 64);
    if (lane == 0) atomicAdd(counter, v);
}

// Pixel of thread tid inside the tile.
__device__ __forceinline__ void pixel_of_thread(int tid, int* lx, int* ly)
{
    const int w = tid >> 6, lane = tid & 63;
    *lx = (w % kWavesX) * kWaveW + (lane % kWaveW);
    *ly = (w / kWavesX) * kWaveH + (lane / kWaveW);
}

// Staging thread (se = entry, sub = 0..3) evaluates tile rows sub*4 .. sub*4+3 of entry se and deposits the
// bits into the per-(wave, entry) lane masks, laid out s_mask[wave][entry] as 2 x 32-bit words each.
__device__ __forceinline__ int stage_masks(uint32_t* s_mask32, int se, int sub, const float4& q0, const float4& q1,
                                            int begY, int endY, int y_tile, int x_tile, int W, int row_end)
{
    uint32_t rm[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int yy = y_tile + sub * 4 + k;
        rm[k] = 0;
        if (yy < row_end) rm[k] = row_mask16(q0.x, q0.y, q0.z, q0.w, q1.x, begY, endY, yy, x_tile, W);
    }
    // waves (wy, 0) and (wy, 1) with wy = sub >> 1; rows (sub & 1) * 4 + k of the 8x8 block; lane = 8 * row + column
    const int wy = sub >> 1, half = sub & 1;
#pragma unroll
    for (int wx = 0; wx < 2; wx++) {
        const int sh = 8 * wx;
        const uint32_t word = ((rm[0] >> sh) & 0xFFu) | (((rm[1] >> sh) & 0xFFu) << 8) |
                              (((rm[2] >> sh) & 0xFFu) << 16) | (((rm[3] >> sh) & 0xFFu) << 24);
        s_mask32[((wy * 2 + wx) * B + se) * 2 + half] = word;
    }
    return (rm[0] != 0u) + (rm[1] != 0u) + (rm[2] != 0u) + (rm[3] != 0u); // non-empty rows (diagnostic counter)
}

// ---------------------------------------------------------------------------------------------------
// forward, main.cpp:414-546
// ---------------------------------------------------------------------------------------------------
// EXACT (S2D_CFG_EXACT_EXP): G = expf(-d2/2) instead of exp_approx -- the switch the reference keeps at main.cpp:51
// "for numerical varidation": the analytic gradients are those of the TRUE exponential, so only in this mode is the
// backward pass the derivative of the forward pass (tests/test_fd_end_to_end.py checks exactly that by finite differences).
template <bool EXACT>
__device__ __forceinline__ float gauss_of(float d2, bool* nonzero)
{
    if (EXACT) {
        *nonzero = true;
        return expf(-0.5f * d2); // main.cpp:51, :527
    }
    return gauss_pow8(d2, nonzero);
}

// What a thread needs to know about its pixel and its place in the workgroup.
struct TileCtx {
    int tile, tx, ty;   // tile index (local to the slab), tile column, GLOBAL tile row
    int tid, lane, w;   // thread, lane, wave within the workgroup
    int x, y;           // pixel
    bool inside;        // pixel inside the image and the slab
    f2 pxy;             // pixel centre, main.cpp:523
};

__device__ __forceinline__ TileCtx tile_ctx(int tile, const Geometry& g)
{
    TileCtx c;
    c.tile = tile;
    c.tx = tile % g.tiles_x;
    c.ty = tile / g.tiles_x + g.trow0;
    c.tid = threadIdx.x;
    c.lane = c.tid & 63;
    c.w = c.tid >> 6;
    int lx, ly;
    pixel_of_thread(c.tid, &lx, &ly);
    c.x = c.tx * kTile + lx;
    c.y = c.ty * kTile + ly;
    c.inside = c.x < g.W && c.y < g.row_end;
    c.pxy = mk2((float)c.x + 0.5f, (float)c.y + 0.5f);
    return c;
}

// Where the thread's pixel sits in image0 / imageRef: the context stores only the rows of its slab.
__device__ __forceinline__ size_t pixel_index(const TileCtx& c, const Geometry& g) { return (size_t)(c.y - g.row_begin) * g.W + c.x; }

// LDS of the forward walk: per-entry record, three 16-B rows at one LDS address (one address register for the blend
// loop's reads):  [0] pos.x, pos.y, a, b   [1] b, d, col_r, col_g   [2] col_b, opacity, -, -
// (b twice: (a,b) and (b,d) are the two columns of inv_cov)
struct FwdShared {
    float4 rec[B][3];
    unsigned long long mask[4 * B]; // [wave][entry]
    int4 alive;                     // per wave: does it still have a live pixel (block_any_alive)
};

// "Does any wave of the workgroup still have a live pixel?"  The answer of a wave is already uniform (its alive mask
// sits in scalar registers), so one lane posts it and everybody reads the four flags after the barrier --
// __syncthreads_or would first reduce the flag over the 64 lanes with eight DPP steps.  The flags are rewritten only
// behind the next batch's staging barrier.
__device__ __forceinline__ bool block_any_alive(int4* flags, int w, int lane, unsigned long long alive_mask)
{
    if (lane == 0) reinterpret_cast<int*>(flags)[w] = alive_mask != 0ull ? 1 : 0;
    __syncthreads();
    const int4 f = *flags;
    return ((f.x | f.y) | (f.z | f.w)) != 0;
}

// One tile's forward walk (main.cpp:419-536 for its pixels): leaves the final colour of this thread's pixel in
// (crg, cb) and the lane masks of every staged pair in wave_masks.
// CHUNK (scenes whose (tile, splat) pairs do not fit one set of lists, s2d_api.hip chunked_raster): the list holds the
// splats of one INDEX RANGE only; the walk continues from the pixel's state after the ranges before it -- (crg, cb) and
// *T_io on entry -- and leaves the state for the range after it.  The reference's loop is front to back in index order
// (main.cpp:419), so cutting it at any index and carrying (colour, T) across the cut changes no operation.
template <bool COUNT, bool EXACT, bool CHUNK = false>
__device__ __forceinline__ void forward_tile(FwdShared& s, const TileCtx& c, const uint32_t* __restrict__ tile_off,
                                             const uint32_t* __restrict__ list, const ProjRec* __restrict__ proj,
                                             unsigned long long* __restrict__ wave_masks, const Geometry& g,
                                             PairCounters* __restrict__ counters, f2& crg, float& cb, float* T_io = nullptr)
{
    const int tid = c.tid, lane = c.lane, w = c.w;
    const f2 pxy = c.pxy;
    float T = 1.0f;
    if (CHUNK) {
        T = *T_io;
    } else {
        crg = mk2(0.0f, 0.0f);                              // main.cpp:414: (0,0,0,1)
        cb = 0.0f;
    }
    // Pixels still above the throughput cut-off (main.cpp:520), as ONE wave-uniform 64-bit mask in scalar registers.
    // (A per-lane bool here costs ~15 scalar instructions per blended entry to merge with exec, and the CU's
    // single scalar unit -- not the SIMDs -- then bounds the loop; measured, profiles/r01/valu_rates.txt.)
    unsigned long long alive_mask = __ballot(CHUNK ? c.inside && !(T < kMinThroughput) : c.inside);
    unsigned long long n_vis = 0, n_act = 0, n_staged = 0, n_exec = 0, n_rows_hit = 0, n_staged_hit = 0;

    const uint32_t beg = tile_off[c.tile], end = tile_off[c.tile + 1];
    const int se = tid >> 2, sub = tid & 3;
    for (uint32_t base = beg; base < end; base += B) {
        const int cnt = (int)min((uint32_t)B, end - base);
        if (se < cnt) {
            const ProjRec* r = proj + list[base + se];
            const float4 q0 = r->q0, q1 = r->q1, q2 = r->q2;
            const int rows_hit = stage_masks(reinterpret_cast<uint32_t*>(s.mask), se, sub, q0, q1, __float_as_int(q2.y),
                                             __float_as_int(q2.z), c.ty * kTile, c.tx * kTile, g.W, g.row_end);
            if (COUNT) {
                n_rows_hit += rows_hit;
                const int entry_rows = rows_hit + __shfl_xor(rows_hit, 1) + __shfl_xor(rows_hit, 2); // the entry's 4 threads
                n_staged_hit += (sub == 0 && entry_rows > 0) ? 1 : 0;
            }
            if (sub == 0) {
                s.rec[se][0] = q0;
                s.rec[se][1] = make_float4(q0.w, q1.x, q1.y, q1.z);
                s.rec[se][2] = make_float4(q1.w, q2.x, 0.0f, 0.0f);
            }
        }
        __syncthreads();
        // keep the lane masks for the backward walk, which goes through exactly these batches (32 B per staged pair)
        if (se < cnt) wave_masks[(size_t)(base + se) * 4 + sub] = s.mask[sub * B + se];
        if (COUNT) n_staged += (tid == 0) ? (unsigned long long)cnt : 0ull;
        if (alive_mask != 0ull || COUNT) {
            const unsigned long long my_mask = (lane < cnt) ? s.mask[w * B + lane] : 0ull;
            // entries that touch a pixel of this wave's block that is still alive now (the counting build walks
            // every entry that touches the block at all, for its "visited" statistic); pixels only ever die, so
            // nothing is missed, and the per-entry test below still sees the mask of the moment
            unsigned long long cand = __ballot(COUNT ? my_mask != 0ull : (my_mask & alive_mask) != 0ull);
            while (cand != 0ull) {
                const int e = __builtin_ctzll(cand);
                cand &= cand - 1ull;
                const unsigned long long wm = readlane_u64(my_mask, e);
                if (COUNT) n_vis += (wm >> lane) & 1ull;
                const unsigned long long act = wm & alive_mask; // visited (main.cpp:511-514) and not cut off (:520)
                if (act == 0ull) continue;
                if (COUNT) n_exec += (lane == 0);
                // Branch-free body: lanes outside `act` run the same instructions with alpha forced to 0, which
                // makes c += (T*c)*0 and T *= 1 exact no-ops.  The scalar mask itself is the select predicate.
                const float4 q0 = s.rec[e][0], q1 = s.rec[e][1], q2 = s.rec[e][2];
                const f2 v = pxy - mk2(q0.x, q0.y);                          // main.cpp:523-524
                const f2 m = mk2(q0.z, q0.w) * v.x + mk2(q1.x, q1.y) * v.y;  // inv_cov * v: (a vx + b vy, b vx + d vy)
                const f2 vm = v * m;
                bool nonzero;
                const float G = gauss_of<EXACT>(vm.x + vm.y, &nonzero);      // main.cpp:526-527
                const unsigned long long on = act & __ballot(nonzero);
                const float alpha = __builtin_amdgcn_inverse_ballot_w64(on) ? G * q2.y : 0.0f;
                crg += (T * mk2(q1.z, q1.w)) * alpha;                        // main.cpp:529-530: (T*c)*alpha
                cb += T * q2.x * alpha;                                      // main.cpp:531
                T *= (1.0f - alpha);                                         // main.cpp:533
                alive_mask &= __ballot(!(T < kMinThroughput));               // main.cpp:520, for the next splat
                if (COUNT) n_act += (act >> lane) & 1ull;
            }
        }
        if (!block_any_alive(&s.alive, w, lane, alive_mask)) break; // every pixel of the tile saturated: retire it
    }
    if (CHUNK) *T_io = T;
    if (COUNT) {
        count_add(&counters->fwd_visited, n_vis, lane);
        count_add(&counters->fwd_active, n_act, lane);
        if (tid == 0) atomicAdd(&counters->fwd_staged, n_staged);
        if (lane == 0) atomicAdd(&counters->fwd_wave_execs, n_exec);
        count_add(&counters->fwd_rows_hit, n_rows_hit, lane);
        count_add(&counters->fwd_staged_hit, n_staged_hit, lane);
    }
}

// Optimistic launch: the host queues the first raster kernel of an iteration before it has seen the containment flag
// the previous Adam (or projection) kernel produced.  If some splat left its binned rectangle the lists are stale:
// do nothing; the host rebuilds them and launches again.  The flag is final before the kernel starts (stream order).
// abort_stamp 0: lists known to be current.  A parameter that went non-finite in an EARLIER iteration stops the
// run where the reference abort()s (main.cpp:752-785): every later kernel of the queue does nothing.
__device__ __forceinline__ bool launch_is_void(const DeviceStatus* status, int abort_stamp, int iteration)
{
    return (abort_stamp != 0 && status->rebin_needed == abort_stamp) || status->first_nonfinite_iter < iteration;
}

template <bool COUNT, bool HALF, bool EXACT>
__global__ __launch_bounds__(256) void raster_forward_kernel(const uint32_t* __restrict__ tile_off,
                                                             const uint32_t* __restrict__ list,
                                                             const ProjRec* __restrict__ proj,
                                                             void* __restrict__ image0,
                                                             unsigned long long* __restrict__ wave_masks, Geometry g,
                                                             const DeviceStatus* __restrict__ status, int abort_stamp,
                                                             int iteration, PairCounters* __restrict__ counters)
{
    __shared__ FwdShared s;
    if (launch_is_void(status, abort_stamp, iteration)) return;
    const int tile = tile_of_block(blockIdx.x, g);
    if (tile < 0) return;
    const TileCtx c = tile_ctx(tile, g);
    f2 crg;
    float cb;
    forward_tile<COUNT, EXACT>(s, c, tile_off, list, proj, wave_masks, g, counters, crg, cb);
    if (c.inside) store_pixel<HALF>(image0, pixel_index(c, g), make_float4(crg.x, crg.y, cb, 1.0f)); // .w reset, main.cpp:543-546
}

// ---------------------------------------------------------------------------------------------------
// Wave-wide sums of eight of the nine partial gradients, through LDS.
// On gfx950 a v_permlane*_swap occupies the SIMD for ~8.8 cycles and a DPP add for ~4.6 against 2.4 for a plain
// v_add_f32 (tools/microbench/valu_rates.hip), so a swap/DPP butterfly costs ~90 SIMD cycles per (wave, entry) of a
// VALU-bound kernel (profiles/r01/v9_bench_lds_reduce.json), while the LDS pipe idles.  Here every lane stores its 8 partials component-major into a
// wave-private scratch (element n = 64*c + lane), reads back elements 8*lane .. 8*lane+7 -- eight lanes' worth of
// component lane>>3 -- with two 16-B reads, adds them pairwise (7 plain adds) and finishes inside its 8-lane
// group with three DPP adds: ~31 SIMD cycles.  Afterwards every lane of group g = lane >> 3 holds the total of
// component g.  Element n lives at dword n + 4*(n >> 7): a ds_read_b128 is served in four groups of sixteen lanes --
// {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS) -- over 64 banks, and with
// 32-byte strides the sixteen lanes of a group would cover the banks twice; shifting the runs of lanes 16-31 (32-47,
// 48-63) by one (two, three) 16-byte pads puts the lanes of every group on sixteen different quads of banks.  (Rounds 1-3
// padded every 32 elements, which is 2-way conflicted under this grouping, as is a pad every 64: SQ_LDS_BANK_CONFLICT
// 85 M -> 15 M cycles per launch, +0.7...1.3 %, profiles/r03/ab_lds_and_salu.txt.)  The stores are 32 consecutive dwords
// per half-wave either way.  Same-wave LDS accesses complete in issue order, so no barrier is needed -- only the compiler
// is told not to reorder.
// The ninth value (opacity gradient) takes the DPP chain to lane 63, interleaved with the group sum.
// ---------------------------------------------------------------------------------------------------
constexpr int kPadShift = 7; // element n lives at dword n + 4 * (n >> kPadShift)
static_assert(kPadShift >= 6, "a pad inside a component's 64 elements would make the store address lane-dependent");
constexpr int red_at(int n) { return n + 4 * (n >> kPadShift); }
constexpr int kRedDwords = red_at(512); // per wave

template <bool NINTH>
__device__ __forceinline__ float wave_sum8_lds(float* sw, int lane, float a0, float a1, float a2, float a3, float a4,
                                               float a5, float a6, float a7, float& a8)
{
    float* wp = sw + lane; // (a component's 64 elements never straddle a pad)
    wp[red_at(0 * 64)] = a0;
    wp[red_at(1 * 64)] = a1;
    wp[red_at(2 * 64)] = a2;
    wp[red_at(3 * 64)] = a3;
    wp[red_at(4 * 64)] = a4;
    wp[red_at(5 * 64)] = a5;
    wp[red_at(6 * 64)] = a6;
    wp[red_at(7 * 64)] = a7;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const float4* rp = reinterpret_cast<const float4*>(sw + 8 * lane + 4 * (lane >> (kPadShift - 3)));
    const float4 u = rp[0], v = rp[1];
    float t = ((u.x + u.y) + (u.z + u.w)) + ((v.x + v.y) + (v.z + v.w));
    // the reads must have returned before the next entry's stores may be issued by the compiler
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (NINTH) {
        asm volatile("s_nop 1\n"
                     "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                     "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                     "s_nop 0\n"
                     "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                     "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                     "s_nop 0\n"
                     "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n"
                     "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n"
                     "s_nop 1\n"
                     "v_add_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n"
                     "s_nop 1\n"
                     "v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
                     "s_nop 1\n"
                     "v_add_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
                     "s_nop 1\n"
                     : "+v"(t), "+v"(a8));
    } else {
        asm volatile("s_nop 1\n"
                     "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                     "s_nop 1\n"
                     "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                     "s_nop 1\n"
                     "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n"
                     "s_nop 1\n"
                     : "+v"(t));
    }
    return t;
}

// num / den as IEEE division would round it, from the hardware reciprocal r = v_rcp_f32(den) (1 ulp) shared by several
// numerators: quotient estimate, then ONE correction q += (num - den*q) * r whose residual is exact (fma).  The estimate
// is off by up to ~1.5 ulp.  Measured on 3 x 10^8 operand pairs of this blend's range, den = 1e-15 and quotients next to
// rounding midpoints included (tools/check_recip_division.py, profiles/r04/r04_recip_division_check.txt): with a correctly
// rounded r the corrected value is the IEEE quotient in every trial; with r one ulp off, 7-9 quotients per 10^8 differ from
// IEEE division (1-2 per 10^8 after a second correction) -- at ~10^9 quotients per iteration of 4096^2 / 1 M that is at most a
// few dozen one ulp off per iteration.  Rounds 1-3 applied two corrections: six more instructions per executed (wave, entry),
// same parity statistics, -2.9 % (profiles/r03/ab_one_division_correction.txt).  No per-division scaling for denormal / huge
// operands: they cannot occur here (den in {1e-15} u [2^-24, 1], |num| <~ 1).
// The quotient should be the one the reference computes:
// c*T - S/(1-alpha) cancels down to a T_final-sized remainder, which magnifies a last-place difference in the
// quotient by T/T_final (10^3..10^5 in flat image regions); the gradient bars are what is asserted.
__device__ __forceinline__ float div_by_recip(float num, float den, float r)
{
    const float q = num * r;
    return __builtin_fmaf(__builtin_fmaf(-den, q, num), r, q);
}

// ---------------------------------------------------------------------------------------------------
// backward, main.cpp:548-712, + the squared error of main.cpp:796-805
//
// Per (pixel, splat) the reference adds nine terms (main.cpp:619, :654-655, :677-678, :685, :704).  They are
// evaluated with the reference's own expressions: its three-term dot products (cc vx^2 + 2sc vx vy + ss vy^2,
// ...) and the channel sum dL . dC/dalpha can cancel, and then carry rounding noise that only the same
// operation order reproduces (the algebraically equal alpha*u^2/sx^3 with u = cos*vx + sin*vy is cheaper and
// more accurate, but differs from the reference by up to 1e-4 of a splat's summed |terms| when its few live
// pixels lie near u = 0).  Only factors that are plain products are regrouped or precomputed per entry
// (1/sx^3, 1/sy^3, (sx^2-sy^2)/(sx^2 sy^2), 0.5*alpha*(2a vx + (b+c) vy) = alpha*mx): a few ulp per term.
// ---------------------------------------------------------------------------------------------------
// Where a tile puts its partial gradient of a splat in deterministic mode.
struct DetSlots {
    const TileRect* rects;     // per splat: the rectangle its pairs were emitted from
    const uint32_t* offsets;   // per splat: first emission slot
    float* data;               // [pairs][kDetStride]: nine floats per slot, padded to three 16-byte words
    uint32_t* stamp;           // [pairs]: iteration + 1 of the last write
    uint32_t* touched;         // [splats]: bit j = slot offsets[i] + j was written in this pass (bit 31: some slot >= 31)
    uint32_t now;              // iteration + 1
};

// LDS of the backward walk.  Entries per staged batch: 64, or 32 in deterministic mode, whose four per-wave slot sets would
// otherwise lift the workgroup from 18.9 to ~25 KB of LDS -- six instead of eight workgroups per CU, which alone costs
// ~14 % (profiles/r03/r03_bound_experiments.txt); with half-size batches it is 17.1 KB.  The forward walk stages 64 either
// way: the lane masks it leaves behind are indexed by list position, and a walk that looks at its pixels' throughput
// every 32 entries stops no later than one that looks every 64.
template <bool DET>
struct BwdShared {
    static constexpr int kBatch = DET ? 32 : B;
    float4 q0[kBatch]; // pos.x, pos.y, a, b
    float4 q1[kBatch]; // b, d, col_r, col_g
    float4 q2[kBatch]; // col_b, opacity, (sx^2-sy^2)/(sx^2 sy^2), sin*cos
    float4 e0[kBatch]; // cc, ss, 2sc, cc - ss
    float4 e1[kBatch]; // ss, cc, 1/sx^3, 1/sy^3
    unsigned long long mask[4 * kBatch]; // [wave][entry]
    // Partial gradients of the batch, 9 floats per slot (+3 pad where one slot per entry is shared).  Deterministic mode:
    // one slot per wave, plain stores, summed over the 4 waves in a fixed order by the flush.  Otherwise the four waves add into ONE slot
    // per entry with ds_add_f32 (eight lanes, eight addresses per wave and entry): the order of those four
    // additions is as free as the order of the global atomics that follow, and 9 KB less LDS per workgroup is
    // one to two more resident workgroups per CU.  The flush zeroes what it read.
    static constexpr int kPartStride = DET ? 9 : 12; // floats per slot (deterministic mode packs its four slots per entry tightly)
    __attribute__((aligned(16))) float part[(DET ? 4 : 1) * kBatch * kPartStride];
    // Splat index of each staged entry: the flush addresses gradients without going back to the list.  Two copies, by batch
    // parity: a batch's flush runs behind the last barrier of the batch, so fast threads already stage the NEXT batch
    // (and overwrite these words) while slow ones still flush.  Deterministic mode also keeps the entry's slot for this
    // tile and which of the splat's slots that is.
    uint32_t idx[2][kBatch];
    uint32_t slot[2][DET ? kBatch : 1];
    uint32_t local[2][DET ? kBatch : 1];
    unsigned long long touched[4]; // bit e: wave w wrote slot e in this batch
    double red[4];
    int4 alive;                    // per wave: does it still have a live pixel (block_any_alive)
    int last_tile;                 // this workgroup stored the launch's last tile error: it adds them all up
    __attribute__((aligned(16))) float xpose[4][kRedDwords]; // wave-private transpose scratch
};

// One tile's backward walk (main.cpp:552-711 for its pixels) from the pixel's final colour `fin` and target `ref`:
// adds the tile's partial gradients into grads (or its deterministic slots) and stores the tile's squared error.
// CHUNK: as in forward_tile -- the list is one index range of the splats, *state_io (running colour r, g, b and T of
// main.cpp:601-625, :707) is the pixel's state after the ranges before it on entry and after this range on return.
template <bool COUNT, bool NEED_OP, bool DET, bool EXACT, bool CHUNK = false>
__device__ __forceinline__ void backward_tile(BwdShared<DET>& s, const TileCtx& c, const float4 fin, const float4 ref,
                                              const uint32_t* __restrict__ tile_off, const uint32_t* __restrict__ list,
                                              const ProjRec* __restrict__ proj,
                                              const unsigned long long* __restrict__ wave_masks,
                                              float* __restrict__ grads, double* __restrict__ tile_sqerr,
                                              const Geometry& g, const DetSlots& det, PairCounters* __restrict__ counters,
                                              const SqerrJob& sq, float4* state_io = nullptr)
{
    constexpr int BB = BwdShared<DET>::kBatch; // entries per staged batch
    const int tid = c.tid, lane = c.lane, w = c.w;
    const bool inside = c.inside;
    const f2 pxy = c.pxy;
    // after wave_sum8_lds the 8-lane group holds the total of component lane >> 3, and lane 63 the ninth sum (slot 8)
    const bool op_lane = NEED_OP && !DET && lane == 63;
    const int part_slot = op_lane ? 8 : lane >> 3;
    const bool adds = (lane & 7) == 0 || op_lane;
    const float dLr = fin.x - ref.x, dLg = fin.y - ref.y, dLb = fin.z - ref.z; // dL_dC, main.cpp:616
    const f2 dLrg = mk2(dLr, dLg), fin_rg = mk2(fin.x, fin.y);

    // squared error of this tile (main.cpp:801-802): float per pixel, double across pixels
    {
        const float ex = dLr * 255.0f, ey = dLg * 255.0f, ez = dLb * 255.0f;
        double e2 = inside ? (double)(ex * ex + ey * ey + ez * ez) : 0.0;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) e2 += __shfl_down(e2, d, 64);
        if (lane == 0) s.red[w] = e2;
    }
    if (!DET)
        for (int i = tid; i < BB * 3; i += 256) reinterpret_cast<float4*>(s.part)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    if (sq.out == nullptr) {
        if (tid == 0) tile_sqerr[c.tile] = ((s.red[0] + s.red[1]) + s.red[2]) + s.red[3];
    } else {
        // Small images: the iteration's squared error (main.cpp:796-805) is summed here, by the workgroup whose tile
        // error completes the set (a ticket behind the stores), instead of by a launch of its own.
        if (tid == 0) {
            __hip_atomic_store(tile_sqerr + c.tile, ((s.red[0] + s.red[1]) + s.red[2]) + s.red[3], __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            __threadfence();
            unsigned long long* ticket = reinterpret_cast<unsigned long long*>(sq.scratch + kSqerrChunks);
            s.last_tile = atomicAdd(ticket, 1ull) == (unsigned long long)(sq.num_tiles - 1) ? 1 : 0;
        }
        __syncthreads();
        if (s.last_tile) { // block-uniform
            __threadfence();
            const double total = sqerr_sum_small(tile_sqerr, sq.num_tiles, s.red);
            if (tid == 0) {
                *sq.out = total;
                *reinterpret_cast<unsigned long long*>(sq.scratch + kSqerrChunks) = 0ull;
            }
        }
    }

    f2 crg = mk2(0.0f, 0.0f);                        // image1 = (0,0,0,1), main.cpp:549
    float cb = 0.0f, T = 1.0f;
    if (CHUNK) {
        crg = mk2(state_io->x, state_io->y);
        cb = state_io->z;
        T = state_io->w;
    }
    unsigned long long alive_mask = __ballot(CHUNK ? inside && !(T < kMinThroughput) : inside); // wave-uniform, scalar registers (see the forward kernel)
    unsigned long long n_vis = 0, n_act = 0, n_staged = 0, n_exec = 0;

    const uint32_t beg = tile_off[c.tile], end = tile_off[c.tile + 1];
    const int se = tid >> 2, sub = tid & 3;
    for (uint32_t base = beg; base < end; base += BB) {
        const int cnt = (int)min((uint32_t)BB, end - base);
        const int pb = (int)(((base - beg) / (uint32_t)BB) & 1u); // which copy of the per-entry index words this batch uses
        if (se < cnt) {
            // the forward pass of this iteration staged the same batch and left its lane masks behind
            s.mask[sub * BB + se] = wave_masks[(size_t)(base + se) * 4 + sub];
            if (sub == 0) {
                const uint32_t idx = list[base + se];
                s.idx[pb][se] = idx;
                if (DET) { // the slot of this tile in the splat's emission rectangle (row-major), looked up beside the record
                    const TileRect r = det.rects[idx];
                    const uint32_t local = (uint32_t)(c.ty - g.trow0 - r.ty0) * (uint32_t)(r.tx1 - r.tx0 + 1) + (uint32_t)(c.tx - r.tx0);
                    s.slot[pb][se] = det.offsets[idx] + local;
                    s.local[pb][se] = min(local, 31u);
                }
                const ProjRec* r = proj + idx;
                const float4 q0 = r->q0, q1 = r->q1, q2 = r->q2;
                const float4 q3 = r->q3; // sin, 1/sx^3, 1/sy^3, (sx^2-sy^2)/(sx^2 sy^2): divided once per splat (pack_proj)
                const float cosT = q2.w, sinT = q3.x;
                const float cc = cosT * cosT, ss = sinT * sinT, sc2 = 2.0f * sinT * cosT;
                s.q0[se] = q0;
                s.q1[se] = make_float4(q0.w, q1.x, q1.y, q1.z);
                s.q2[se] = make_float4(q1.w, q2.x, q3.w, sinT * cosT);
                s.e0[se] = make_float4(cc, ss, sc2, cc - ss);
                s.e1[se] = make_float4(ss, cc, q3.y, q3.z);
            }
        }
        __syncthreads();
        if (COUNT) n_staged += (tid == 0) ? (unsigned long long)cnt : 0ull;
        unsigned long long touched = 0ull;
        if (alive_mask != 0ull || COUNT) {
            const unsigned long long my_mask = (lane < cnt) ? s.mask[w * BB + lane] : 0ull;
            unsigned long long cand = __ballot(COUNT ? my_mask != 0ull : (my_mask & alive_mask) != 0ull); // as in the forward walk
            while (cand != 0ull) {
                const int e = __builtin_ctzll(cand);
                cand &= cand - 1ull;
                const unsigned long long wm = readlane_u64(my_mask, e);
                if (COUNT) n_vis += (wm >> lane) & 1ull;
                const unsigned long long act_mask = wm & alive_mask;
                if (act_mask == 0ull) continue;
                touched |= 1ull << e;
                if (COUNT) n_exec += (lane == 0);
                // Lanes this splat does not visit (main.cpp:595-598) or whose pixel is already below the throughput
                // cut-off (main.cpp:604) run the same instructions with alpha forced to 0: then c += T*c*0 and
                // T *= 1 are exact no-ops and every gradient term below is a multiple of alpha, i.e. exactly 0 --
                // no divergent region, no zero-initialisation of the nine partials.
                if (COUNT) {
                    n_act += (act_mask >> lane) & 1ull;
                    const int na = __popcll(act_mask);
                    if (lane == 0) atomicAdd(&counters->bwd_lane_hist[na], 1ull);
                    // how many of the block's four 4x4 quadrants have a live covered pixel (lane = 8*row + column)
                    const uint32_t lo = (uint32_t)act_mask, hi = (uint32_t)(act_mask >> 32);
                    const int nq = ((lo & 0x0F0F0F0Fu) != 0u) + ((lo & 0xF0F0F0F0u) != 0u) + ((hi & 0x0F0F0F0Fu) != 0u) +
                                   ((hi & 0xF0F0F0F0u) != 0u);
                    if (lane == 0) atomicAdd(&counters->bwd_quadrant_execs, (unsigned long long)nq);
                }
                float g_px, g_py, g_sx, g_sy, g_rot, g_r, g_g, g_b, g_op = 0.f;
                {
                    const float4 q0 = s.q0[e], q1 = s.q1[e], q2 = s.q2[e];
                    const float4 e0 = s.e0[e], e1 = s.e1[e];
                    // ---- the reference's operations, in its order (decides T, alive, the running colour) ----
                    const f2 v = pxy - mk2(q0.x, q0.y);                              // main.cpp:607-608
                    const f2 m = mk2(q0.z, q0.w) * v.x + mk2(q1.x, q1.y) * v.y;      // inv_cov * v
                    const f2 vm = v * m;
                    bool nonzero;
                    const float G = gauss_of<EXACT>(vm.x + vm.y, &nonzero);          // main.cpp:609-610
                    // one scalar mask selects alpha (and dOpacity): visited, alive, and G not cut to 0
                    const bool on = __builtin_amdgcn_inverse_ballot_w64(act_mask & __ballot(nonzero));
                    const float alpha = on ? G * q2.y : 0.0f;                        // main.cpp:611
                    const f2 Trg = T * mk2(q1.z, q1.w);
                    const float Tb = T * q2.x;
                    crg += Trg * alpha;                                              // main.cpp:623-625
                    cb += Tb * alpha;
                    // ---- gradient terms ----
                    const float dC_dc = alpha * T;                                   // main.cpp:618
                    const f2 g_rg = dLrg * dC_dc;
                    g_r = g_rg.x;
                    g_g = g_rg.y;
                    g_b = dLb * dC_dc;
                    // S / (1 - alpha + 1e-15), main.cpp:627-628: three quotients over one denominator (div_by_recip)
                    const float den = 1.0f - alpha + 1.0e-15f;
                    const float rd = __builtin_amdgcn_rcpf(den); // 1 ulp; the correction below does the rest
                    const f2 S_rg = fin_rg - crg;                                    // S = final - colour
                    const f2 nden2 = mk2(-den, -den), rd2 = mk2(rd, rd);
                    f2 q_rg = S_rg * rd;
                    q_rg = fma2(fma2(nden2, q_rg, S_rg), rd2, q_rg);
                    const f2 dCa_rg = Trg - q_rg;
                    const float dCa_b = Tb - div_by_recip(fin.z - cb, den, rd);
                    // the three channel products may cancel: same products, same order as main.cpp:629-630
                    const f2 pr = dLrg * dCa_rg;
                    const float gs = (pr.x + pr.y) + dLb * dCa_b;
                    const float ga = gs * alpha;
                    const f2 g_pos = ga * m;                                         // main.cpp:639-640, :654-655
                    g_px = g_pos.x;
                    g_py = g_pos.y;
                    const f2 vv = v.x * v;                                           // vx*vx, vx*vy
                    const float vyy = v.y * v.y;
                    // both covariance dot products (main.cpp:657-662): (cc, ss)*vxx +- 2sc*vxy, then + (ss, cc)*vyy
                    // (x + (-2sc)*vxy and x - 2sc*vxy are the same IEEE operation)
                    const float sc2vxy = e0.z * vv.y;
                    const f2 dots = mk2(e0.x * vv.x + sc2vxy, e0.y * vv.x - sc2vxy) + mk2(e1.x, e1.y) * vyy;
                    const f2 g_s = (ga * mk2(e1.z, e1.w)) * dots;                    // main.cpp:657-662, :677-678
                    g_sx = g_s.x;
                    g_sy = g_s.y;
                    g_rot = (ga * q2.z) * (e0.w * v.x * v.y - q2.w * (vv.x - vyy)); // main.cpp:680-685, e0.w = cc - ss
                    if (NEED_OP) g_op = on ? gs * G : 0.0f;                           // main.cpp:703-704
                    T *= (1.0f - alpha);                                             // main.cpp:707
                    alive_mask &= __ballot(!(T < kMinThroughput));
                }
                // order of the record: pos.xy, sx, sy, rot, color.rgb, opacity (main.cpp:85-93)
                const float tot = wave_sum8_lds<NEED_OP>(s.xpose[w], lane, g_px, g_py, g_sx, g_sy, g_rot, g_r, g_g, g_b, g_op);
                // slot of (wave, entry, component), 12 dwords per entry.  The entry's share of the index is computed on the
                // scalar unit explicitly: left to itself the compiler folds e * 12 + lane term into a per-lane
                // v_mad_u64_u32 in the blend loop.
                float* const part = s.part;
                int pe;
                asm("s_mul_i32 %0, %1, %2" : "=s"(pe) : "s"((DET ? __builtin_amdgcn_readfirstlane(w) : 0) * BB + e), "n"(BwdShared<DET>::kPartStride));
                if (DET) {
                    if ((lane & 7) == 0) part[pe + part_slot] = tot;
                    if (NEED_OP && lane == 63) part[pe + 8] = g_op;
                } else {
                    const float val = op_lane ? g_op : tot; // one LDS instruction for the eight totals and the ninth sum
                    if (adds) __hip_atomic_fetch_add(part + (pe + part_slot), val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        if (DET && lane == 0) s.touched[w] = touched;
        const bool any = block_any_alive(&s.alive, w, lane, alive_mask);
        // one burst per (tile, splat): 9 consecutive floats -- float atomics into grads[idx], or (deterministic
        // mode) plain stores into this tile's own slot of the splat, summed later in a fixed order
        for (int i = tid; i < cnt * 9; i += 256) {
            const int e = i / 9, k = i - e * 9;
            if (!NEED_OP && k == 8 && !DET) continue; // dSplats.opacity left at zero on request
            float v = 0.0f;
            bool any_w = false;
            if (DET) {
#pragma unroll
                for (int ww = 0; ww < (DET ? 4 : 1); ww++)
                    if ((s.touched[ww] >> e) & 1ull) {
                        v += s.part[(ww * BB + e) * BwdShared<DET>::kPartStride + k];
                        any_w = true;
                    }
            } else {
                float* slot = s.part + e * BwdShared<DET>::kPartStride + k;
                v = *slot;
                *slot = 0.0f; // the next batch's waves add after the staging barrier
                any_w = true;
            }
            if (DET) {
                if (any_w) {
                    const uint32_t slot = s.slot[pb][e];
                    det.data[(size_t)slot * kDetStride + k] = (!NEED_OP && k == 8) ? 0.0f : v;
                    if (k == 0) {
                        det.stamp[slot] = det.now;
                        atomicOr(det.touched + s.idx[pb][e], 1u << s.local[pb][e]); // which slots the gather pass has to read
                    }
                }
            } else if (any_w && v != 0.0f) {
                atomicAdd(grads + (size_t)s.idx[pb][e] * 9 + k, v);
            }
        }
        if (!any) break;
    }
    if (CHUNK) *state_io = make_float4(crg.x, crg.y, cb, T);
    if (COUNT) {
        count_add(&counters->bwd_visited, n_vis, lane);
        count_add(&counters->bwd_active, n_act, lane);
        if (tid == 0) atomicAdd(&counters->bwd_staged, n_staged);
        if (lane == 0) atomicAdd(&counters->bwd_wave_execs, n_exec);
    }
}

template <bool COUNT, bool NEED_OP, bool HALF, bool DET, bool EXACT>
__global__ __launch_bounds__(256) void raster_backward_kernel(const uint32_t* __restrict__ tile_off,
                                                              const uint32_t* __restrict__ list,
                                                              const ProjRec* __restrict__ proj,
                                                              const void* __restrict__ image0,
                                                              const void* __restrict__ image_ref,
                                                              const unsigned long long* __restrict__ wave_masks,
                                                              float* __restrict__ grads,
                                                              double* __restrict__ tile_sqerr, Geometry g,
                                                              DetSlots det, const DeviceStatus* __restrict__ status,
                                                              int iteration, PairCounters* __restrict__ counters)
{
    __shared__ BwdShared<DET> s;
    if (launch_is_void(status, 0, iteration)) return; // the reference abort()ed in an earlier iteration
    const int tile = tile_of_block(blockIdx.x, g);
    if (tile < 0) return;
    const TileCtx c = tile_ctx(tile, g);
    float4 fin = make_float4(0.f, 0.f, 0.f, 0.f), ref = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c.inside) {
        fin = load_pixel<HALF>(image0, pixel_index(c, g));    // finalColor, main.cpp:613
        ref = load_pixel<HALF>(image_ref, pixel_index(c, g));
    }
    backward_tile<COUNT, NEED_OP, DET, EXACT>(s, c, fin, ref, tile_off, list, proj, wave_masks, grads, tile_sqerr, g, det, counters,
                                              SqerrJob{nullptr, 0, nullptr, nullptr});
}

// Forward and backward walk of a tile in ONE launch (what s2d_step and s2d_forward_backward queue): a tile's backward
// pass needs nothing but its own pixels' final colours, which are still in registers when the forward walk ends.  One
// dispatch, one ramp-down tail and one read of image0 (16 B per pixel) less per iteration than the two kernels above,
// which remain for callers that run the passes separately.  The lane masks still travel through wave_masks (written
// and read back by the same workgroup, so they rarely leave L2).  With S2D_CFG_FP16_IMAGES the backward walk sees the
// final colour as stored, i.e. rounded to fp16, exactly like the separate kernels.  image0 is written only when
// `write_image` is set (s2d_step: the last iteration of the call; nothing else reads it).
template <bool NEED_OP, bool HALF, bool DET, bool EXACT>
__global__ __launch_bounds__(256, 8) void raster_fused_kernel(const uint32_t* __restrict__ tile_off,
                                                           const uint32_t* __restrict__ list,
                                                           const ProjRec* __restrict__ proj, void* __restrict__ image0,
                                                           const void* __restrict__ image_ref,
                                                           unsigned long long* __restrict__ wave_masks,
                                                           float* __restrict__ grads, double* __restrict__ tile_sqerr,
                                                           Geometry g, DetSlots det,
                                                           const DeviceStatus* __restrict__ status, int abort_stamp,
                                                           int iteration, int write_image, SqerrJob sq)
{
    constexpr size_t kBytes = sizeof(BwdShared<DET>) > sizeof(FwdShared) ? sizeof(BwdShared<DET>) : sizeof(FwdShared);
    __shared__ __attribute__((aligned(16))) unsigned char smem[kBytes]; // the two walks use the same LDS one after the other
    if (launch_is_void(status, abort_stamp, iteration)) return;
    const int tile = tile_of_block(blockIdx.x, g);
    if (tile < 0) return;
    const TileCtx c = tile_ctx(tile, g);
    f2 crg;
    float cb;
    forward_tile<false, EXACT>(*reinterpret_cast<FwdShared*>(smem), c, tile_off, list, proj, wave_masks, g, nullptr, crg, cb);
    float4 fin = make_float4(crg.x, crg.y, cb, 1.0f), ref = make_float4(0.f, 0.f, 0.f, 0.f);
    if (HALF) { // what the backward pass would read back from the fp16 framebuffer
        const __half2 a = __floats2half2_rn(fin.x, fin.y), b = __floats2half2_rn(fin.z, fin.w);
        const float2 fa = __half22float2(a), fb = __half22float2(b);
        fin = make_float4(fa.x, fa.y, fb.x, fb.y);
    }
    if (c.inside) {
        if (write_image) store_pixel<HALF>(image0, pixel_index(c, g), fin); // .w reset, main.cpp:543-546
        ref = load_pixel<HALF>(image_ref, pixel_index(c, g));
    } else {
        fin = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads(); // the backward walk re-uses the LDS the forward walk's last flag exchange may still be reading
    backward_tile<false, NEED_OP, DET, EXACT>(*reinterpret_cast<BwdShared<DET>*>(smem), c, fin, ref, tile_off, list, proj,
                                              wave_masks, grads, tile_sqerr, g, det, nullptr, sq);
}

// ---------------------------------------------------------------------------------------------------
// Index-range ("chunked") rendering, for scenes with more (tile, splat) pairs than one set of lists may hold
// (s2d_api.hip chunked_raster).  The splats are cut into consecutive index ranges; the lists of one range at a time are
// built and walked, front to back like the reference's loops (main.cpp:419, :552), and the per-pixel state is carried
// from range to range in `state` (fp32 whatever the image format): (r, g, b, T).
//   forward pass : raster_forward_chunk_kernel per range; each stores the state and image0 (.w = 1, main.cpp:543-546),
//                  so image0 is final after the last range -- or after the range behind which no pixel is alive any
//                  more (*any_alive stays 0: the host skips the ranges that follow, in both passes).
//   backward pass: raster_backward_chunk_kernel per range, from a fresh state: the forward walk of the range (only for its
//                  lane masks -- the backward walk reads them back, exactly as in the fused kernel) and the backward
//                  walk from the same state, with the FINAL colours of the forward pass out of image0.
// ---------------------------------------------------------------------------------------------------
template <bool HALF, bool EXACT>
__global__ __launch_bounds__(256) void raster_forward_chunk_kernel(const uint32_t* __restrict__ tile_off,
                                                                   const uint32_t* __restrict__ list,
                                                                   const ProjRec* __restrict__ proj, void* __restrict__ image0,
                                                                   float4* __restrict__ state, int first,
                                                                   unsigned long long* __restrict__ wave_masks, Geometry g,
                                                                   const DeviceStatus* __restrict__ status, int iteration,
                                                                   uint32_t* __restrict__ any_alive)
{
    __shared__ FwdShared s;
    if (launch_is_void(status, 0, iteration)) return;
    const int tile = tile_of_block(blockIdx.x, g);
    if (tile < 0) return;
    const TileCtx c = tile_ctx(tile, g);
    float4 st = make_float4(0.0f, 0.0f, 0.0f, 1.0f); // main.cpp:414
    if (!first && c.inside) st = state[pixel_index(c, g)];
    f2 crg = mk2(st.x, st.y);
    float cb = st.z, T = st.w;
    forward_tile<false, EXACT, true>(s, c, tile_off, list, proj, wave_masks, g, nullptr, crg, cb, &T);
    if (c.inside) {
        state[pixel_index(c, g)] = make_float4(crg.x, crg.y, cb, T);
        store_pixel<HALF>(image0, pixel_index(c, g), make_float4(crg.x, crg.y, cb, 1.0f)); // .w reset, main.cpp:543-546
    }
    if (__ballot(c.inside && !(T < kMinThroughput)) != 0ull && c.lane == 0) atomicOr(any_alive, 1u);
}

template <bool NEED_OP, bool HALF, bool DET, bool EXACT>
__global__ __launch_bounds__(256) void raster_backward_chunk_kernel(const uint32_t* __restrict__ tile_off,
                                                                    const uint32_t* __restrict__ list,
                                                                    const ProjRec* __restrict__ proj,
                                                                    const void* __restrict__ image0,
                                                                    const void* __restrict__ image_ref,
                                                                    float4* __restrict__ state, int first,
                                                                    unsigned long long* __restrict__ wave_masks,
                                                                    float* __restrict__ grads, double* __restrict__ tile_sqerr,
                                                                    Geometry g, DetSlots det,
                                                                    const DeviceStatus* __restrict__ status, int iteration)
{
    constexpr size_t kBytes = sizeof(BwdShared<DET>) > sizeof(FwdShared) ? sizeof(BwdShared<DET>) : sizeof(FwdShared);
    __shared__ __attribute__((aligned(16))) unsigned char smem[kBytes];
    if (launch_is_void(status, 0, iteration)) return;
    const int tile = tile_of_block(blockIdx.x, g);
    if (tile < 0) return;
    const TileCtx c = tile_ctx(tile, g);
    float4 st = make_float4(0.0f, 0.0f, 0.0f, 1.0f); // image1 = (0,0,0,1), main.cpp:549
    float4 fin = make_float4(0.f, 0.f, 0.f, 0.f), ref = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c.inside) {
        if (!first) st = state[pixel_index(c, g)];
        fin = load_pixel<HALF>(image0, pixel_index(c, g));    // finalColor, main.cpp:613: of ALL ranges
        ref = load_pixel<HALF>(image_ref, pixel_index(c, g));
    }
    {   // the lane masks of this range's batches, from the state the backward walk starts from (results discarded)
        f2 crg = mk2(st.x, st.y);
        float cb = st.z, T = st.w;
        forward_tile<false, EXACT, true>(*reinterpret_cast<FwdShared*>(smem), c, tile_off, list, proj, wave_masks, g, nullptr, crg, cb, &T);
    }
    __syncthreads(); // the backward walk re-uses the LDS the forward walk's last flag exchange may still be reading
    backward_tile<false, NEED_OP, DET, EXACT, true>(*reinterpret_cast<BwdShared<DET>*>(smem), c, fin, ref, tile_off, list, proj,
                                                    wave_masks, grads, tile_sqerr, g, det, nullptr,
                                                    SqerrJob{nullptr, 0, nullptr, nullptr}, &st);
    if (c.inside) state[pixel_index(c, g)] = st;
}

// Deterministic mode: gradient of splat i = sum of the partials its tiles stored this iteration, in emission
// (tile row-major) order -- the same order whatever the dispatch order of the tiles was.  A splat's tiles announce
// themselves in touched[i] (bit j = its j-th emission slot; a tile reaches a splat's entry only while some pixel of it is
// still live, so most of a splat's ~20 slots stay unwritten and a hidden splat has none): the pass reads one word per
// splat and then only the slots that hold something, instead of every stamp of every slot.  Slots from the 32nd on share
// bit 31 and are told apart by their stamps.  The word is cleared for the next pass.
__device__ __forceinline__ void det_add_slot(float (&acc)[9], const float* __restrict__ data, uint32_t slot)
{
    const float4* d = reinterpret_cast<const float4*>(data + (size_t)slot * kDetStride); // three 16-byte loads per slot
    const float4 a = d[0], b = d[1], c = d[2];
    acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
    acc[4] += b.x; acc[5] += b.y; acc[6] += b.z; acc[7] += b.w;
    acc[8] += c.x;
}

__global__ __launch_bounds__(256) void gather_grads_kernel(const uint32_t* __restrict__ offsets,
                                                           const uint32_t* __restrict__ counts, int n,
                                                           const float* __restrict__ data,
                                                           const uint32_t* __restrict__ stamp, uint32_t* __restrict__ touched,
                                                           uint32_t now, float* __restrict__ grads)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t m = touched[i];
    if (m == 0u) return; // no tile wrote anything for this splat: its gradient record stays as it is (zero)
    touched[i] = 0u;
    float acc[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const uint32_t o = offsets[i];
    const bool tail = (m >> 31) != 0u;
    m &= 0x7FFFFFFFu;
    while (m != 0u) { // ascending slot order, four slots' loads in flight at a time
        uint32_t js[4];
        int cnt = 0;
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (m != 0u) {
                js[q] = (uint32_t)__builtin_ctz(m);
                m &= m - 1u;
                cnt = q + 1;
            }
        float4 v[4][3];
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (q < cnt) {
                const float4* d = reinterpret_cast<const float4*>(data + (size_t)(o + js[q]) * kDetStride);
                v[q][0] = d[0]; v[q][1] = d[1]; v[q][2] = d[2];
            }
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (q < cnt) {
                acc[0] += v[q][0].x; acc[1] += v[q][0].y; acc[2] += v[q][0].z; acc[3] += v[q][0].w;
                acc[4] += v[q][1].x; acc[5] += v[q][1].y; acc[6] += v[q][1].z; acc[7] += v[q][1].w;
                acc[8] += v[q][2].x;
            }
    }
    if (tail) {
        const uint32_t c = counts[i];
        for (uint32_t s = o + 31u; s < o + c; s++)
            if (stamp[s] == now) det_add_slot(acc, data, s);
    }
    float* gr = grads + (size_t)i * 9;
#pragma unroll
    for (int k = 0; k < 9; k++) gr[k] += acc[k]; // += : the buffer is zero here unless the caller accumulates slabs
}

// Sum of the per-tile squared errors in a fixed order (deterministic MSE trace), two stages in one launch:
// kSqerrChunks blocks each reduce a contiguous chunk to partial[b] (sqerr_reduce, s2d_device.h); the block that finishes last (ticket counter)
// adds the partials, again in a fixed order, and re-arms the counter.  scratch = kSqerrChunks doubles + one
// 64-bit counter, zero before the first launch (s2d_api.hip allocates it behind tile_sqerr).
__global__ __launch_bounds__(256) void sqerr_finalize_kernel(const double* __restrict__ tile_sqerr, int num_tiles,
                                                             double* __restrict__ out, double* scratch,
                                                             const DeviceStatus* __restrict__ status, int iteration)
{
    if (status->first_nonfinite_iter < iteration) return;
    sqerr_reduce(tile_sqerr, num_tiles, out, scratch, (int)blockIdx.x, kSqerrChunks);
}

static inline unsigned raster_grid(int num_tiles) { return (unsigned)(((num_tiles + 7) / 8) * 8); }

hipError_t launch_raster_forward(const uint32_t* tile_off, const uint32_t* list, const ProjRec* proj, void* image0,
                                 bool half_images, unsigned long long* wave_masks, Geometry g, const DeviceStatus* status,
                                 int abort_stamp, int iteration, PairCounters* counters, bool count, bool exact_exp,
                                 hipStream_t stream)
{
    if (g.num_tiles <= 0) return hipSuccess;
    const dim3 grid(raster_grid(g.num_tiles)), block(256);
#define S2D_LAUNCH_FWD(C, H) \
    hipLaunchKernelGGL((raster_forward_kernel<C, H, false>), grid, block, 0, stream, tile_off, list, proj, image0, wave_masks, g, status, abort_stamp, iteration, counters)
    if (exact_exp) { // validation mode: plain fp32 images, no pair counting (s2d_create rejects the combinations)
        hipLaunchKernelGGL((raster_forward_kernel<false, false, true>), grid, block, 0, stream, tile_off, list, proj, image0,
                           wave_masks, g, status, abort_stamp, iteration, counters);
    } else if (count) {
        if (half_images) S2D_LAUNCH_FWD(true, true); else S2D_LAUNCH_FWD(true, false);
    } else {
        if (half_images) S2D_LAUNCH_FWD(false, true); else S2D_LAUNCH_FWD(false, false);
    }
#undef S2D_LAUNCH_FWD
    return hipGetLastError();
}

hipError_t launch_raster_backward(const uint32_t* tile_off, const uint32_t* list, const ProjRec* proj,
                                  const void* image0, const void* image_ref, bool half_images,
                                  const unsigned long long* wave_masks, float* grads, double* tile_sqerr, Geometry g,
                                  bool need_opacity_grad, const DetGather* dg, const DeviceStatus* status, int iteration,
                                  PairCounters* counters, bool count, bool exact_exp, hipStream_t stream)
{
    if (g.num_tiles <= 0) return hipSuccess;
    DetSlots det{nullptr, nullptr, nullptr, nullptr, nullptr, 0u};
    if (dg) det = DetSlots{dg->rects, dg->offsets, dg->data, dg->stamp, dg->touched, dg->now};
    const dim3 grid(raster_grid(g.num_tiles)), block(256);
#define S2D_LAUNCH_BWD(C, O, H, D)                                                                                       \
    hipLaunchKernelGGL((raster_backward_kernel<C, O, H, D, false>), grid, block, 0, stream, tile_off, list, proj, image0, image_ref, \
                       wave_masks, grads, tile_sqerr, g, det, status, iteration, counters)
#define S2D_LAUNCH_BWD_D(C, O, H) do { if (dg) S2D_LAUNCH_BWD(C, O, H, true); else S2D_LAUNCH_BWD(C, O, H, false); } while (0)
#define S2D_LAUNCH_BWD_H(C, O) do { if (half_images) S2D_LAUNCH_BWD_D(C, O, true); else S2D_LAUNCH_BWD_D(C, O, false); } while (0)
    if (exact_exp) {
#define S2D_LAUNCH_BWD_X(O, D)                                                                                           \
    hipLaunchKernelGGL((raster_backward_kernel<false, O, false, D, true>), grid, block, 0, stream, tile_off, list, proj, image0, \
                       image_ref, wave_masks, grads, tile_sqerr, g, det, status, iteration, counters)
        if (need_opacity_grad) { if (dg) S2D_LAUNCH_BWD_X(true, true); else S2D_LAUNCH_BWD_X(true, false); }
        else { if (dg) S2D_LAUNCH_BWD_X(false, true); else S2D_LAUNCH_BWD_X(false, false); }
#undef S2D_LAUNCH_BWD_X
    } else if (count) {
        if (need_opacity_grad) S2D_LAUNCH_BWD_H(true, true); else S2D_LAUNCH_BWD_H(true, false);
    } else {
        if (need_opacity_grad) S2D_LAUNCH_BWD_H(false, true); else S2D_LAUNCH_BWD_H(false, false);
    }
#undef S2D_LAUNCH_BWD_H
#undef S2D_LAUNCH_BWD_D
#undef S2D_LAUNCH_BWD
    if (dg && dg->n > 0)
        hipLaunchKernelGGL(gather_grads_kernel, dim3((dg->n + 255) / 256), dim3(256), 0, stream, dg->offsets, dg->counts,
                           dg->n, dg->data, dg->stamp, dg->touched, dg->now, grads);
    return hipGetLastError();
}

hipError_t launch_raster_fused(const uint32_t* tile_off, const uint32_t* list, const ProjRec* proj, void* image0,
                               const void* image_ref, bool half_images, unsigned long long* wave_masks, float* grads,
                               double* tile_sqerr, Geometry g, bool need_opacity_grad, const DetGather* dg,
                               const DeviceStatus* status, int abort_stamp, int iteration, bool write_image, bool exact_exp,
                               SqerrJob sq, hipStream_t stream)
{
    if (g.num_tiles <= 0) return hipSuccess;
    DetSlots det{nullptr, nullptr, nullptr, nullptr, nullptr, 0u};
    if (dg) det = DetSlots{dg->rects, dg->offsets, dg->data, dg->stamp, dg->touched, dg->now};
    const dim3 grid(raster_grid(g.num_tiles)), block(256);
    const int wi = write_image ? 1 : 0;
#define S2D_LAUNCH_FUSED(O, H, D, X)                                                                                       \
    hipLaunchKernelGGL((raster_fused_kernel<O, H, D, X>), grid, block, 0, stream, tile_off, list, proj, image0, image_ref, \
                       wave_masks, grads, tile_sqerr, g, det, status, abort_stamp, iteration, wi, sq)
#define S2D_LAUNCH_FUSED_D(O, H, X) do { if (dg) S2D_LAUNCH_FUSED(O, H, true, X); else S2D_LAUNCH_FUSED(O, H, false, X); } while (0)
    if (exact_exp) {
        if (need_opacity_grad) S2D_LAUNCH_FUSED_D(true, false, true); else S2D_LAUNCH_FUSED_D(false, false, true);
    } else if (half_images) {
        if (need_opacity_grad) S2D_LAUNCH_FUSED_D(true, true, false); else S2D_LAUNCH_FUSED_D(false, true, false);
    } else {
        if (need_opacity_grad) S2D_LAUNCH_FUSED_D(true, false, false); else S2D_LAUNCH_FUSED_D(false, false, false);
    }
#undef S2D_LAUNCH_FUSED_D
#undef S2D_LAUNCH_FUSED
    if (dg && dg->n > 0)
        hipLaunchKernelGGL(gather_grads_kernel, dim3((dg->n + 255) / 256), dim3(256), 0, stream, dg->offsets, dg->counts,
                           dg->n, dg->data, dg->stamp, dg->touched, dg->now, grads);
    return hipGetLastError();
}

hipError_t launch_raster_forward_chunk(const uint32_t* tile_off, const uint32_t* list, const ProjRec* proj, void* image0,
                                       bool half_images, float4* state, bool first, unsigned long long* wave_masks, Geometry g,
                                       const DeviceStatus* status, int iteration, uint32_t* any_alive, bool exact_exp,
                                       hipStream_t stream)
{
    if (g.num_tiles <= 0) return hipSuccess;
    const dim3 grid(raster_grid(g.num_tiles)), block(256);
    const int f = first ? 1 : 0;
#define S2D_LAUNCH_FC(H, X) \
    hipLaunchKernelGGL((raster_forward_chunk_kernel<H, X>), grid, block, 0, stream, tile_off, list, proj, image0, state, f, wave_masks, g, status, iteration, any_alive)
    if (exact_exp) S2D_LAUNCH_FC(false, true);
    else if (half_images) S2D_LAUNCH_FC(true, false);
    else S2D_LAUNCH_FC(false, false);
#undef S2D_LAUNCH_FC
    return hipGetLastError();
}

hipError_t launch_raster_backward_chunk(const uint32_t* tile_off, const uint32_t* list, const ProjRec* proj, const void* image0,
                                        const void* image_ref, bool half_images, float4* state, bool first,
                                        unsigned long long* wave_masks, float* grads, double* tile_sqerr, Geometry g,
                                        bool need_opacity_grad, const DetGather* dg, const DeviceStatus* status, int iteration,
                                        bool exact_exp, hipStream_t stream)
{
    if (g.num_tiles <= 0) return hipSuccess;
    DetSlots det{nullptr, nullptr, nullptr, nullptr, nullptr, 0u};
    if (dg) det = DetSlots{dg->rects, dg->offsets, dg->data, dg->stamp, dg->touched, dg->now};
    const dim3 grid(raster_grid(g.num_tiles)), block(256);
    const int f = first ? 1 : 0;
#define S2D_LAUNCH_BC(O, H, D, X)                                                                                              \
    hipLaunchKernelGGL((raster_backward_chunk_kernel<O, H, D, X>), grid, block, 0, stream, tile_off, list, proj, image0, image_ref, \
                       state, f, wave_masks, grads, tile_sqerr, g, det, status, iteration)
#define S2D_LAUNCH_BC_D(O, H, X) do { if (dg) S2D_LAUNCH_BC(O, H, true, X); else S2D_LAUNCH_BC(O, H, false, X); } while (0)
    if (exact_exp) {
        if (need_opacity_grad) S2D_LAUNCH_BC_D(true, false, true); else S2D_LAUNCH_BC_D(false, false, true);
    } else if (half_images) {
        if (need_opacity_grad) S2D_LAUNCH_BC_D(true, true, false); else S2D_LAUNCH_BC_D(false, true, false);
    } else {
        if (need_opacity_grad) S2D_LAUNCH_BC_D(true, false, false); else S2D_LAUNCH_BC_D(false, false, false);
    }
#undef S2D_LAUNCH_BC_D
#undef S2D_LAUNCH_BC
    if (dg && dg->n > 0)
        hipLaunchKernelGGL(gather_grads_kernel, dim3((dg->n + 255) / 256), dim3(256), 0, stream, dg->offsets, dg->counts,
                           dg->n, dg->data, dg->stamp, dg->touched, dg->now, grads);
    return hipGetLastError();
}

hipError_t launch_sqerr_finalize(const double* tile_sqerr, int num_tiles, double* out, double* scratch,
                                 const DeviceStatus* status, int iteration, hipStream_t stream)
{
    hipLaunchKernelGGL(sqerr_finalize_kernel, dim3(kSqerrChunks), dim3(256), 0, stream, tile_sqerr, num_tiles, out, scratch,
                       status, iteration);
    return hipGetLastError();
}

} // namespace s2d
