// s2d_multi.hip -- several GPUs behind ONE handle (include/splat2d.h "s2d_multi_*"; SURVEY.md section 8b/8e: "device
// list ... multi-GPU fan-out is internal").
//
// A reference-side caller keeps its single-threaded frame loop (main.cpp:334) and gets N GPUs by swapping s2d_ctx for
// s2d_multi: the image is cut into N row slabs (whole 16-pixel tile rows), every device gets an ordinary context for
// its slab and one worker thread that keeps the context's calls in order; the caller's thread only hands out commands
// and adds up the slabs' squared errors.  Two ways to keep the devices consistent:
//
//  * slab ownership (default): a device holds -- projects, lists, updates -- only the splats that can reach its rows.
//    Per iteration the holders of a shared splat swap its partial gradient rows (s2d_rows_gather into a send buffer,
//    a peer-to-peer copy over xGMI queued on the RECEIVER's stream behind the sender's event, s2d_grads_combine in
//    rank order, so every holder forms the same bits); every 64 iterations the hold sets are refreshed from the
//    current parameters and the 27-float state of splats that drift into a neighbour's reach is handed over.  No
//    collective library involved: neighbours talk to neighbours, ~1 MB per neighbour and iteration (8 ranks, 4096^2 / 1 M) instead of a 36 MB all-reduce.
//  * replicated state (S2D_MULTI_REPLICATED, north_star's scheme): splats and Adam state on every device, the N x 9 fp32
//    gradient arrays summed in place by an RCCL all-reduce (ncclAllReduce on each context's own stream, between
//    s2d_forward_backward and s2d_adam_step), the identical Adam step everywhere.  RCCL is loaded with dlopen when
//    such a handle is created, so other users of the library never load it.
//
// S2D_MULTI_SHARE_GPU (rehearsal on a box with fewer GPUs than ranks): all ranks on the first listed device; the peer
// copies become plain device copies, the all-reduce is staged through pinned host memory in rank order.
#include "../../include/splat2d.h"
#include "../../include/splat2d_test.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h> // types only; the entry points are resolved at run time

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr; // frees a communicator whose collective can no longer complete
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

bool load_rccl(Rccl* r, std::string* why)
{
    static std::mutex m;
    static Rccl cached;
    std::lock_guard<std::mutex> lk(m);
    if (!cached.lib) {
        // One RCCL per process, on the process's one HIP runtime: first an image that is already
        // mapped -- PyTorch's wheel bundles RCCL as torch/lib/librccl.so with SONAME librccl.so.1, and a process that has
        // imported torch must not get the system's copy beside it -- and only then a load by name (the system's librccl.so.1
        // needs libamdhip64.so.7, which binds to whichever runtime the process already has, by SONAME).
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            cached.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
            if (cached.lib) break;
        }
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (cached.lib) break;
            cached.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!cached.lib) {
            *why = std::string("cannot load librccl: ") + dlerror();
            return false;
        }
        cached.CommInitAll = (decltype(cached.CommInitAll))dlsym(cached.lib, "ncclCommInitAll");
        cached.CommDestroy = (decltype(cached.CommDestroy))dlsym(cached.lib, "ncclCommDestroy");
        cached.CommAbort = (decltype(cached.CommAbort))dlsym(cached.lib, "ncclCommAbort");
        cached.AllReduce = (decltype(cached.AllReduce))dlsym(cached.lib, "ncclAllReduce");
        cached.GetErrorString = (decltype(cached.GetErrorString))dlsym(cached.lib, "ncclGetErrorString");
        if (!cached.CommInitAll || !cached.CommDestroy || !cached.AllReduce || !cached.GetErrorString) {
            *why = "librccl lacks ncclCommInitAll / ncclAllReduce";
            cached.lib = nullptr;
            return false;
        }
    }
    *r = cached;
    return true;
}

// Reusable barrier of the rank threads; abort() releases everybody for good once a rank has failed.  A wait is bounded:
// a rank that does not arrive within `timeout_ms` (it stopped answering -- stuck in a device wait, or dead) breaks the
// barrier for everybody instead of leaving the others asleep, and `timed_out` says so.
struct Barrier {
    std::mutex m;
    std::condition_variable cv;
    int n = 1, waiting = 0, generation = 0;
    std::atomic<bool> broken{false};
    std::atomic<bool> timed_out{false};
    std::atomic<int> timeout_ms{0}; // 0: wait without bound
    uint64_t here = 0;              // bit r: rank r is waiting in the current generation
    std::atomic<uint64_t> missing{0}; // when a wait timed out: the ranks that had not arrived at that moment
    void wait(int rank)
    {
        std::unique_lock<std::mutex> lk(m);
        if (broken) return;
        const int gen = generation;
        here |= 1ull << (rank & 63);
        if (++waiting == n) {
            waiting = 0;
            here = 0;
            generation++;
            cv.notify_all();
            return;
        }
        const auto ready = [&] { return gen != generation || broken; };
        const int ms = timeout_ms.load();
        if (ms <= 0) {
            cv.wait(lk, ready);
        } else if (!cv.wait_for(lk, std::chrono::milliseconds(ms), ready)) {
            missing = ~here & ((n >= 64) ? ~0ull : ((1ull << n) - 1ull)); // who is not here NOW (they may show up a moment later)
            timed_out = true;
            broken = true;
            cv.notify_all();
        }
    }
    void abort()
    {
        std::lock_guard<std::mutex> lk(m);
        broken = true;
        cv.notify_all();
    }
    void reset()
    {
        std::lock_guard<std::mutex> lk(m);
        broken = false;
        timed_out = false;
        waiting = 0;
        here = 0;
        missing = 0;
    }
};

enum Command { CMD_NONE = 0, CMD_STEP, CMD_HOLD, CMD_FORWARD, CMD_QUIT };
enum Scheme { SCHEME_NONE = 0, SCHEME_OWNERSHIP = 1, SCHEME_REPLICATED = 2 };
constexpr int kStopped = -1; // a rank that stopped because another one failed (never reported to the caller)

// What a rank thread is doing, for the report when one stops answering (s2d_multi_last_error names it).
enum Phase { PH_IDLE = 0, PH_RASTER, PH_EXCHANGE, PH_ALLREDUCE, PH_ADAM, PH_REFRESH, PH_DRAIN, PH_HOLD, PH_FORWARD, PH_DONE };
const char* phase_name(int p)
{
    static const char* const names[] = {"idle", "raster launch", "gradient exchange", "all-reduce", "Adam step", "hold-set refresh",
                                        "waiting for its stream", "making the hold sets", "forward", "done"};
    return p >= 0 && p <= PH_DONE ? names[p] : "?";
}

struct Progress {                 // one per rank, written by its thread only
    std::atomic<uint64_t> ticks{0};   // bumped at every phase change: the watchdog of run_command looks for movement
    std::atomic<int> phase{PH_IDLE};
    std::atomic<int> iteration{0};
};

// Device array that only grows (freed and re-allocated while the rank's stream is idle).
template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t count)
    {
        if (count <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = std::max<size_t>(count + count / 4, 256);
        const hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

// Slab ownership, one rank's side (the host logic of distributed.HaloStep, here inside the library).
struct HaloRank {
    std::vector<uint32_t> mask;           // per splat: bit q = rank q holds it; 0 = this rank does not (its copy is stale)
    std::vector<int32_t> held;            // ascending ids with mask != 0
    std::vector<int32_t> splits, offsets; // gradient rows swapped with each peer per iteration; where its segment starts
    int total = 0, n_rows = 0;
    bool any_exchange = false;            // some rank swaps something: everybody takes part in the per-iteration barrier
    DevBuf<int32_t> d_send_ids, d_rows, d_src;
    DevBuf<float> d_send[2], d_recv;      // two send buffers: a peer may still be copying iteration k while k + 1 is gathered
    DevBuf<uint32_t> d_mask;              // n words: s2d_halo_masks output / s2d_halo_commit input
    hipEvent_t ev_sent[2] = {nullptr, nullptr};
    unsigned seq = 0;                     // exchanges since the hold sets were made (its parity picks the send buffer)
    // a refresh: state rows on their way to ranks that newly hold a splat
    std::vector<uint32_t> fresh;                 // masks from the current parameters (0 where not held)
    std::vector<std::vector<int32_t>> out_ids;   // [destination rank]
    std::vector<std::vector<uint32_t>> out_mask; // ... and the hold-set word the destination starts with
    std::vector<int32_t> out_off;                // first payload row of each destination
    DevBuf<int32_t> d_pay_ids, d_in_ids;
    DevBuf<float> d_pay_sp, d_pay_ad, d_in_sp, d_in_ad;
    long long handed = 0;                 // state rows sent or received so far (diagnostic)
    void release()
    {
        d_send_ids.release(), d_rows.release(), d_src.release(), d_send[0].release(), d_send[1].release(), d_recv.release();
        d_mask.release(), d_pay_ids.release(), d_in_ids.release(), d_pay_sp.release(), d_pay_ad.release(), d_in_sp.release();
        d_in_ad.release();
        for (hipEvent_t& e : ev_sent) {
            if (e) (void)hipEventDestroy(e);
            e = nullptr;
        }
    }
};

} // namespace

struct s2d_multi {
    int world = 0, W = 0, H = 0, n = 0;
    bool share_gpu = false;
    int scheme = SCHEME_NONE;
    std::vector<int> devices;
    std::vector<s2d_ctx*> ctx;
    std::vector<int> row_begin, row_end;
    Rccl rccl;
    // One communicator per rank.  A slot is emptied (exchange(nullptr)) by whoever aborts or destroys it, so exactly one
    // thread ever frees a communicator and nobody can pick up a freed one (a rank takes its OWN slot's value only).
    std::unique_ptr<std::atomic<ncclComm_t>[]> comms;
    int n_comms = 0;
    std::vector<float*> host_grads; // replicated + share-gpu: pinned copies of the ranks' partial gradients; [0] receives the sum
    // slab ownership
    std::vector<HaloRank> halo;
    std::vector<int32_t> bounds;    // world + 1 row bounds
    int interval = 64;              // iterations between refreshes of the hold sets
    float margin = 0.0f;            // rows; must outlast one interval of Adam steps
    bool hold_valid = false;        // hold sets exist (otherwise every context holds, and has, everything)
    int hold_age = 0;               // iterations since they were made
    std::atomic<int> late{0};
    // sent_seq[r]: exchanges (since the hold sets were planned) whose gather and hipEventRecord rank r's thread has ISSUED.
    // A receiver may call hipStreamWaitEvent on r's event of exchange k only once r has recorded it -- the one thing the
    // rank threads tell each other per iteration, pairwise and without sleeping (no barrier: the streams order the rest)
    std::unique_ptr<std::atomic<unsigned>[]> sent_seq;
    std::atomic<bool> comms_aborted{false};   // a stop was published during the current command: no rank starts a collective any more
    std::atomic<bool> collective_lost{false}; // some communicator really was aborted
    std::atomic<bool> timed_out{false};       // some wait ran out: the ranks no longer agree on where the run stands
    bool dead = false;              // a collective was aborted or a rank stopped answering: the handle must be re-created
    bool failed_step = false;       // the last step failed: the ranks stand at different iterations until the state is set afresh
    bool stuck = false;             // ... and some worker never came back: its thread and context are abandoned, not freed
    // A rank that stops answering (a device wait that never ends, a thread that died) must not hang the caller: every
    // wait of one rank for another, and for its own stream, gives up after this long without progress (milliseconds;
    // S2D_MULTI_STALL_TIMEOUT_MS or s2d_multi_set_stall_timeout; 0 = wait for ever, the behaviour up to round 3)
    std::atomic<int> stall_ms{30000};
    std::unique_ptr<Progress[]> progress;
    std::vector<hipEvent_t> ev_prog;      // [rank * 3 + which], see prog_event()
    std::vector<char> rank_done;          // guarded by m
    // test hook (include/splat2d_test.h): rank `stall_rank` stops before its exchange of iteration `stall_iter`
    int stall_rank = -1, stall_iter = -1, stall_for_ms = 0;
    // command hand-out
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cv_cmd, cv_done;
    int cmd = CMD_NONE, cmd_seq = 0, done_count = 0;
    int step_iters = 0;
    uint32_t step_flags = 0;
    int step_first_iter = 0;
    Barrier barrier;
    std::vector<int> rank_rc;
    std::vector<std::string> rank_msg; // failures of this file's own HIP calls (the contexts keep theirs)
    std::vector<std::vector<double>> sqerr; // [rank][iteration of the call]: partial squared errors
    int iterations = 0;
    char err[1280] = {0};
};

namespace {

int mfail(s2d_multi* m, int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(m->err, sizeof(m->err), fmt, ap);
    va_end(ap);
    return code;
}

int rank_fail(s2d_multi* m, int rank, int code, const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    m->rank_msg[(size_t)rank] = buf;
    return code;
}

#define MHIP(m, rank, call)                                                                               \
    do {                                                                                                  \
        const hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess) return rank_fail((m), (rank), S2D_E_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

// Rows [r0, r1) of rank `rank`: whole 16-pixel tile rows, as even as possible (== distributed.slab_rows).
void slab_rows(int height, int rank, int world, int* r0, int* r1)
{
    const int tile_rows = (height + 15) / 16;
    *r0 = (int)((long long)tile_rows * rank / world) * 16;
    *r1 = (int)((long long)tile_rows * (rank + 1) / world) * 16;
    if (*r1 > height) *r1 = height;
}

inline void at_phase(s2d_multi* m, int rank, int phase, int iteration)
{
    Progress& P = m->progress[(size_t)rank];
    P.phase.store(phase, std::memory_order_relaxed);
    P.iteration.store(iteration, std::memory_order_relaxed);
    P.ticks.fetch_add(1, std::memory_order_release);
}

// Where every rank's thread was last seen, for the reports below; and the rank furthest behind -- when a wait runs out
// because of a rank that stopped answering, that is the one (the waiting ranks are iterations ahead of it, or at the same
// iteration in a later phase).
std::string phase_table(s2d_multi* m)
{
    std::string out;
    int worst = 0;
    long long worst_key = -1;
    for (int q = 0; q < m->world; q++) {
        const Progress& P = m->progress[(size_t)q];
        const int ph = P.phase.load(), it = P.iteration.load();
        char one[96];
        snprintf(one, sizeof(one), "%srank %d: %s, iteration %d", out.empty() ? "" : "; ", q, phase_name(ph), it);
        out += one;
        const long long key = ph == PH_DONE ? (1LL << 60) : (long long)it * 16 + ph;
        if (worst_key < 0 || key < worst_key) {
            worst_key = key;
            worst = q;
        }
    }
    char tail[64];
    snprintf(tail, sizeof(tail), "; furthest behind: rank %d (device %d)", worst, m->devices[(size_t)worst]);
    return out + tail;
}

// This rank's communicator, taken out of its slot and aborted (RCCL then ends the kernels of this rank that wait for a
// peer which will never arrive).  Whoever empties the slot owns the communicator: no second abort, no destroy afterwards.
void abort_own_collective(s2d_multi* m, int rank)
{
    if (m->scheme != SCHEME_REPLICATED || m->share_gpu || rank >= m->n_comms) return;
    const ncclComm_t c = m->comms[(size_t)rank].exchange(nullptr);
    if (!c) return;
    m->collective_lost.store(true);
    if (m->rccl.CommAbort) (void)m->rccl.CommAbort(c);
}

// A rank has failed (or stopped answering): publish the stop FIRST -- no rank may start another collective, every rank
// thread leaves its waits -- and only then give up this rank's own communicator.  The other ranks abort theirs when they
// notice (stopped()): a communicator is never freed under the thread that may be submitting to it.
void stop_everybody(s2d_multi* m, int rank)
{
    m->comms_aborted.store(true);
    m->barrier.abort();
    abort_own_collective(m, rank);
}

// Called by a rank thread inside its waits: has somebody stopped the command?  If collectives were declared lost
// (stop_everybody: a rank failed on the way or stopped answering) this rank gives up its communicator too.  A broken
// barrier alone does not mean that: the finite guard (main.cpp:752-785) breaks it at the END of a rank's share, with every
// collective of the call queued by everybody, and the handle stays usable.
inline bool stopped(s2d_multi* m, int rank)
{
    if (!m->barrier.broken.load(std::memory_order_acquire)) return false;
    if (m->comms_aborted.load()) abort_own_collective(m, rank);
    return true;
}

// The rendezvous of the rank threads, for a rank: S2D_OK, kStopped (somebody else failed), or S2D_E_STATE when the wait itself ran out -- some rank
// never arrived; the first rank to notice reports it.
int meet_rank(s2d_multi* m, int r, const char* where)
{
    m->barrier.wait(r);
    if (!m->barrier.broken) return S2D_OK;
    if (m->barrier.timed_out.exchange(false)) {
        const std::string table = phase_table(m);
        std::string who;
        const uint64_t missing = m->barrier.missing.load();
        for (int q = 0; q < m->world; q++)
            if ((missing >> q) & 1ull) {
                char one[48];
                snprintf(one, sizeof(one), "%srank %d (device %d)", who.empty() ? "" : ", ", q, m->devices[(size_t)q]);
                who += one;
            }
        m->timed_out.store(true);
        stop_everybody(m, r);
        return rank_fail(m, r, S2D_E_STATE, "%s did not reach the rendezvous of the %s within %d ms (the ranks now: %s)", who.c_str(), where,
                         m->barrier.timeout_ms.load(), table.c_str());
    }
    (void)stopped(m, r);
    return kStopped;
}

// Wait, with a bound, until event `ev` (recorded on rank r's stream) has completed.  hipStreamSynchronize would wait for
// ever behind a kernel that never ends (a collective whose peer is gone, a device that stopped); an event and a poll
// let the thread notice a stop published by another rank, abort its own collective, and give up with a report.
int wait_event(s2d_multi* m, int r, hipEvent_t ev, const char* what)
{
    const int limit = m->stall_ms.load();
    const auto t0 = std::chrono::steady_clock::now();
    bool aborted = false;
    for (unsigned spins = 0;; spins++) {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) return aborted ? kStopped : S2D_OK;
        if (e != hipErrorNotReady) return rank_fail(m, r, S2D_E_HIP, "hipEventQuery while %s: %s", what, hipGetErrorString(e));
        if (!aborted && stopped(m, r)) aborted = true; // keep polling: with its collective aborted the stream drains
        const long long waited = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
        if (limit > 0 && waited > (aborted ? 2LL * limit : (long long)limit)) {
            if (aborted) return kStopped; // somebody else already reports; this stream is left as it is
            const Progress& P = m->progress[(size_t)r];
            const std::string table = phase_table(m);
            m->timed_out.store(true);
            stop_everybody(m, r);
            return rank_fail(m, r, S2D_E_STATE, "rank %d (device %d): the device did not finish the work queued on its stream within %d ms "
                                                "while %s (iteration %d) -- its own kernels, or a collective / peer wait for a rank that "
                                                "stopped answering (%s)", r, m->devices[(size_t)r], limit, what, P.iteration.load(), table.c_str());
        }
        if (spins < 2000) std::this_thread::yield();
        else std::this_thread::sleep_for(std::chrono::microseconds(spins < 20000 ? 20 : 200));
    }
}

// Three events per rank: [0], [1] mark every 64th iteration of a step (rank_step waits for the one before last, so a
// wait never covers more than 128 iterations of device work and never leaves the device idle); [2] is for one-off drains.
int drain(s2d_multi* m, int r, hipEvent_t ev, hipStream_t stream, const char* what)
{
    if (ev == nullptr) { // before the events exist (creation failed half-way): the plain wait
        MHIP(m, r, hipStreamSynchronize(stream));
        return S2D_OK;
    }
    MHIP(m, r, hipEventRecord(ev, stream));
    return wait_event(m, r, ev, what);
}

inline hipEvent_t prog_event(s2d_multi* m, int r, int which) { return m->ev_prog.empty() ? nullptr : m->ev_prog[(size_t)r * 3 + (size_t)which]; }
inline int drain_now(s2d_multi* m, int r, const char* what) { return drain(m, r, prog_event(m, r, 2), (hipStream_t)s2d_stream(m->ctx[(size_t)r]), what); }


// Device-to-device copy between two ranks' buffers, queued on `stream` (the receiver's): over xGMI between two GPUs,
// an ordinary device copy when the ranks share one.
hipError_t rank_copy(s2d_multi* m, void* dst, int dst_rank, const void* src, int src_rank, size_t bytes, hipStream_t stream)
{
    if (bytes == 0) return hipSuccess;
    const int dd = m->devices[(size_t)dst_rank], sd = m->devices[(size_t)src_rank];
    if (dd == sd) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream);
    return hipMemcpyPeerAsync(dst, dd, src, sd, bytes, stream);
}

// ---------------------------------------------------------------------------------------------------------------------
// replicated state: the sum RCCL would form, through host memory (S2D_MULTI_SHARE_GPU): every rank copies its partial
// gradients out, rank 0 adds them in rank order, every rank copies the sum back in.
// ---------------------------------------------------------------------------------------------------------------------
int staged_all_reduce(s2d_multi* m, int rank, float* grads, size_t count, hipStream_t stream)
{
    MHIP(m, rank, hipMemcpyAsync(m->host_grads[(size_t)rank], grads, count * sizeof(float), hipMemcpyDeviceToHost, stream));
    if (int rc = drain_now(m, rank, "copying its gradients out for the staged all-reduce")) return rc;
    if (int rc = meet_rank(m, rank, "staged all-reduce (partials out)")) return rc;
    if (rank == 0)
        for (int q = 1; q < m->world; q++) {
            const float* src = m->host_grads[(size_t)q];
            float* dst = m->host_grads[0];
            for (size_t k = 0; k < count; k++) dst[k] += src[k];
        }
    if (int rc = meet_rank(m, rank, "staged all-reduce (sum formed)")) return rc;
    MHIP(m, rank, hipMemcpyAsync(grads, m->host_grads[0], count * sizeof(float), hipMemcpyHostToDevice, stream));
    if (int rc = drain_now(m, rank, "copying the summed gradients in")) return rc;
    return meet_rank(m, rank, "staged all-reduce (sum read)"); // nobody overwrites its host copy before everybody has read the sum
}

// ---------------------------------------------------------------------------------------------------------------------
// slab ownership
// ---------------------------------------------------------------------------------------------------------------------
// Exchange lists for the rank's current hold set: which gradient rows go to which peer (ascending ids per peer -- the
// peer derives the same list from its identical mask words), and for every shared row where each holder's partial sits
// in the receive buffer (s2d_grads_combine's table).
int plan(s2d_multi* m, int r)
{
    HaloRank& H = m->halo[(size_t)r];
    const int world = m->world;
    const uint32_t me = 1u << r;
    hipStream_t stream = (hipStream_t)s2d_stream(m->ctx[(size_t)r]);
    std::vector<std::vector<int32_t>> peer((size_t)world);
    std::vector<int32_t> rows, src;
    for (const int32_t i : H.held) {
        uint32_t others = H.mask[(size_t)i] & ~me;
        if (!others) continue;
        const size_t u = rows.size();
        rows.push_back(i);
        src.resize((u + 1) * (size_t)world, -1);
        src[u * (size_t)world + (size_t)r] = -2;
        while (others) {
            const int p = __builtin_ctz(others);
            others &= others - 1u;
            src[u * (size_t)world + (size_t)p] = (int32_t)peer[(size_t)p].size(); // + the peer's offset, below
            peer[(size_t)p].push_back(i);
        }
    }
    H.splits.assign((size_t)world, 0);
    H.offsets.assign((size_t)world, 0);
    int total = 0;
    for (int p = 0; p < world; p++) {
        H.offsets[(size_t)p] = total;
        H.splits[(size_t)p] = (int32_t)peer[(size_t)p].size();
        total += H.splits[(size_t)p];
    }
    for (size_t u = 0; u < rows.size(); u++)
        for (int p = 0; p < world; p++)
            if (src[u * (size_t)world + (size_t)p] >= 0) src[u * (size_t)world + (size_t)p] += H.offsets[(size_t)p];
    std::vector<int32_t> send_ids;
    send_ids.reserve((size_t)total);
    for (int p = 0; p < world; p++) send_ids.insert(send_ids.end(), peer[(size_t)p].begin(), peer[(size_t)p].end());
    H.total = total;
    H.n_rows = (int)rows.size();
    H.seq = 0;
    m->sent_seq[(size_t)r].store(0u, std::memory_order_release); // every rank thread is behind a barrier here, none is waiting on it
    MHIP(m, r, H.d_send_ids.reserve((size_t)total));
    MHIP(m, r, H.d_send[0].reserve((size_t)total * 9));
    MHIP(m, r, H.d_send[1].reserve((size_t)total * 9));
    MHIP(m, r, H.d_recv.reserve((size_t)total * 9));
    MHIP(m, r, H.d_rows.reserve(rows.size()));
    MHIP(m, r, H.d_src.reserve(src.size()));
    if (total) MHIP(m, r, hipMemcpyAsync(H.d_send_ids.p, send_ids.data(), (size_t)total * 4, hipMemcpyHostToDevice, stream));
    if (!rows.empty()) {
        MHIP(m, r, hipMemcpyAsync(H.d_rows.p, rows.data(), rows.size() * 4, hipMemcpyHostToDevice, stream));
        MHIP(m, r, hipMemcpyAsync(H.d_src.p, src.data(), src.size() * 4, hipMemcpyHostToDevice, stream));
    }
    return drain_now(m, r, "uploading its exchange plan"); // the host vectors go away
}

// After the plans of all ranks are in (behind a barrier): does anybody swap anything, and do the two sides of every
// pair agree on how much?
int settle_plans(s2d_multi* m, int r)
{
    HaloRank& H = m->halo[(size_t)r];
    H.any_exchange = false;
    for (int p = 0; p < m->world; p++) {
        const HaloRank& P = m->halo[(size_t)p];
        H.any_exchange = H.any_exchange || P.total > 0;
        if (p != r && P.splits[(size_t)r] != H.splits[(size_t)p])
            return rank_fail(m, r, S2D_E_STATE, "slab ownership: ranks %d and %d disagree on the rows they share (%d vs %d)", r, p,
                             H.splits[(size_t)p], P.splits[(size_t)r]);
    }
    return S2D_OK;
}

// Hold sets from scratch; every context holds a complete, identical copy of the splats and the Adam state.
int hold_fresh(s2d_multi* m, int r)
{
    HaloRank& H = m->halo[(size_t)r];
    s2d_ctx* c = m->ctx[(size_t)r];
    hipStream_t stream = (hipStream_t)s2d_stream(c);
    const size_t n = (size_t)m->n;
    MHIP(m, r, hipSetDevice(m->devices[(size_t)r]));
    for (hipEvent_t& e : H.ev_sent)
        if (!e) MHIP(m, r, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (int rc = s2d_halo_commit(c, nullptr, r, 0)) return rc; // hold everything: the masks below cover every splat
    MHIP(m, r, H.d_mask.reserve(n));
    H.fresh.resize(n);
    H.mask.assign(n, 0u);
    H.held.clear();
    H.out_ids.assign((size_t)m->world, {});
    H.out_mask.assign((size_t)m->world, {});
    H.out_off.assign((size_t)m->world, 0);
    if (int rc = s2d_halo_masks(c, m->world, m->bounds.data(), m->margin, H.d_mask.p)) return rc;
    MHIP(m, r, hipMemcpyAsync(H.fresh.data(), H.d_mask.p, n * 4, hipMemcpyDeviceToHost, stream));
    if (int rc = drain_now(m, r, "reading the hold-set words")) return rc;
    for (size_t i = 0; i < n; i++)
        if ((H.fresh[i] >> r) & 1u) {
            H.mask[i] = H.fresh[i];
            H.held.push_back((int32_t)i);
        }
    MHIP(m, r, hipMemcpyAsync(H.d_mask.p, H.mask.data(), n * 4, hipMemcpyHostToDevice, stream));
    if (int rc = s2d_halo_commit(c, H.d_mask.p, r, 1)) return rc;
    if (int rc = plan(m, r)) return rc;
    if (int rc = meet_rank(m, r, "hold sets")) return rc;
    return settle_plans(m, r);
}

// Refresh the hold sets from the current parameters (every `interval` iterations).  A splat that has come within
// reach + margin of a rank that does not hold it yet is handed over -- parameters and Adam moments, by its
// lowest-ranked holder -- before it can touch that rank's rows; a holder it has left drops it.
int refresh(s2d_multi* m, int r)
{
    HaloRank& H = m->halo[(size_t)r];
    s2d_ctx* c = m->ctx[(size_t)r];
    hipStream_t stream = (hipStream_t)s2d_stream(c);
    const int world = m->world;
    const size_t n = (size_t)m->n;
    const uint32_t me = 1u << r;
    if (int rc = drain_now(m, r, "finishing the iterations before a hold-set refresh")) return rc; // up to `interval` iterations of device work
    if (int rc = s2d_synchronize(c)) return rc; // the finite guard; and every copy this rank queued has landed
    if (int rc = s2d_halo_masks(c, world, m->bounds.data(), m->margin, H.d_mask.p)) return rc;
    MHIP(m, r, hipMemcpyAsync(H.fresh.data(), H.d_mask.p, n * 4, hipMemcpyDeviceToHost, stream));
    if (int rc = drain_now(m, r, "reading the hold-set words")) return rc;
    for (int q = 0; q < world; q++) {
        H.out_ids[(size_t)q].clear();
        H.out_mask[(size_t)q].clear();
    }
    std::vector<int32_t> keep;
    keep.reserve(H.held.size());
    for (const int32_t i : H.held) {
        const uint32_t old = H.mask[(size_t)i], now = H.fresh[(size_t)i];
        if ((old & (0u - old)) == me) { // the lowest-ranked old holder hands the splat to its new holders
            uint32_t arriving = now & ~old;
            while (arriving) {
                const int q = __builtin_ctz(arriving);
                arriving &= arriving - 1u;
                H.out_ids[(size_t)q].push_back(i);
                H.out_mask[(size_t)q].push_back(now);
            }
        }
        if (now & me) {
            H.mask[(size_t)i] = now;
            keep.push_back(i);
        } else {
            H.mask[(size_t)i] = 0u;
        }
    }
    // outgoing state rows, one segment per destination
    std::vector<int32_t> pay_ids;
    for (int q = 0; q < world; q++) {
        H.out_off[(size_t)q] = (int32_t)pay_ids.size();
        pay_ids.insert(pay_ids.end(), H.out_ids[(size_t)q].begin(), H.out_ids[(size_t)q].end());
    }
    const int k_out = (int)pay_ids.size();
    if (k_out) {
        MHIP(m, r, H.d_pay_ids.reserve((size_t)k_out));
        MHIP(m, r, H.d_pay_sp.reserve((size_t)k_out * 9));
        MHIP(m, r, H.d_pay_ad.reserve((size_t)k_out * 18));
        MHIP(m, r, hipMemcpyAsync(H.d_pay_ids.p, pay_ids.data(), (size_t)k_out * 4, hipMemcpyHostToDevice, stream));
        if (int rc = s2d_rows_gather(c, S2D_ROWS_SPLATS, H.d_pay_ids.p, k_out, H.d_pay_sp.p)) return rc;
        if (int rc = s2d_rows_gather(c, S2D_ROWS_ADAM, H.d_pay_ids.p, k_out, H.d_pay_ad.p)) return rc;
        if (int rc = drain_now(m, r, "gathering the state rows it hands over")) return rc;
    }
    if (int rc = meet_rank(m, r, "hold-set refresh (state rows out)")) return rc; // every rank's outgoing rows are in place, every stream is idle
    // incoming state rows, in sender order
    std::vector<int32_t> in_ids;
    std::vector<uint32_t> in_mask;
    for (int p = 0; p < world; p++)
        if (p != r) {
            const HaloRank& P = m->halo[(size_t)p];
            in_ids.insert(in_ids.end(), P.out_ids[(size_t)r].begin(), P.out_ids[(size_t)r].end());
            in_mask.insert(in_mask.end(), P.out_mask[(size_t)r].begin(), P.out_mask[(size_t)r].end());
        }
    const int k_in = (int)in_ids.size();
    int late = 0;
    if (k_in) {
        MHIP(m, r, H.d_in_ids.reserve((size_t)k_in));
        MHIP(m, r, H.d_in_sp.reserve((size_t)k_in * 9));
        MHIP(m, r, H.d_in_ad.reserve((size_t)k_in * 18));
        MHIP(m, r, hipMemcpyAsync(H.d_in_ids.p, in_ids.data(), (size_t)k_in * 4, hipMemcpyHostToDevice, stream));
        size_t off = 0;
        for (int p = 0; p < world; p++)
            if (p != r) {
                const HaloRank& P = m->halo[(size_t)p];
                const size_t cnt = P.out_ids[(size_t)r].size(), from = (size_t)P.out_off[(size_t)r];
                if (!cnt) continue;
                MHIP(m, r, rank_copy(m, H.d_in_sp.p + off * 9, r, P.d_pay_sp.p + from * 9, p, cnt * 9 * sizeof(float), stream));
                MHIP(m, r, rank_copy(m, H.d_in_ad.p + off * 18, r, P.d_pay_ad.p + from * 18, p, cnt * 18 * sizeof(float), stream));
                off += cnt;
            }
        std::vector<float> sp((size_t)k_in * 9);
        MHIP(m, r, hipMemcpyAsync(sp.data(), H.d_in_sp.p, sp.size() * sizeof(float), hipMemcpyDeviceToHost, stream));
        if (int rc = drain_now(m, r, "fetching the state rows handed to it")) return rc;
        // a splat that arrives already touching this rank's rows was rasterised here without being listed: the margin
        // did not outlast the interval
        const float r0 = (float)m->bounds[(size_t)r], r1 = (float)m->bounds[(size_t)r + 1];
        for (int j = 0; j < k_in; j++) {
            const float* q = sp.data() + (size_t)j * 9;
            const float reach = 3.0f * std::max(q[2], q[3]) + 2.0f;
            if (q[1] + reach >= r0 && q[1] - reach <= r1) late++;
        }
    }
    if (late) m->late.fetch_add(late);
    if (int rc = meet_rank(m, r, "hold-set refresh (state rows in)")) return rc; // everybody has fetched its rows and reported late arrivals
    if (m->late.load() > 0)        // fatal on every rank alike
        return rank_fail(m, r, S2D_E_STATE, "slab ownership: %d splat(s) reached the rows of a rank before their state was handed "
                                            "over (%d on rank %d): the margin of %.1f rows did not outlast %d iterations",
                         m->late.load(), late, r, (double)m->margin, m->interval);
    if (k_in) {
        if (int rc = s2d_rows_scatter(c, S2D_ROWS_SPLATS, H.d_in_ids.p, k_in, H.d_in_sp.p)) return rc;
        if (int rc = s2d_rows_scatter(c, S2D_ROWS_ADAM, H.d_in_ids.p, k_in, H.d_in_ad.p)) return rc;
        for (int j = 0; j < k_in; j++) H.mask[(size_t)in_ids[(size_t)j]] = in_mask[(size_t)j];
        keep.insert(keep.end(), in_ids.begin(), in_ids.end());
        std::sort(keep.begin(), keep.end());
    }
    H.held.swap(keep);
    H.handed += k_out + k_in;
    MHIP(m, r, hipMemcpyAsync(H.d_mask.p, H.mask.data(), n * 4, hipMemcpyHostToDevice, stream));
    if (int rc = s2d_halo_commit(c, H.d_mask.p, r, k_in > 0)) return rc; // departures alone leave the tile lists valid
    if (int rc = plan(m, r)) return rc;
    if (int rc = meet_rank(m, r, "hold-set refresh (plans)")) return rc; // plans are in; the outgoing buffers may be re-used
    return settle_plans(m, r);
}

// Has rank p's thread issued (gathered + recorded the event of) exchange number `seq`?  Spins without sleeping at first:
// the threads queue an iteration in tens of microseconds and run at the same pace, so the wait is short.  A failed rank
// ends it (kStopped); a rank that does not answer within the stall limit ends it too, with a report that names it
// (S2D_E_STATE): the reference's only failure policy is abort() (main.cpp:752-785), the boundary turns "a rank stopped
// answering" into a status like the rest.
int wait_issued(s2d_multi* m, int r, int p, unsigned seq)
{
    const std::atomic<unsigned>& a = m->sent_seq[(size_t)p];
    const int limit = m->stall_ms.load();
    std::chrono::steady_clock::time_point t0;
    for (unsigned spins = 0; a.load(std::memory_order_acquire) < seq; spins++) {
        if (stopped(m, r)) return kStopped;
        if (spins <= 64) continue;
        if (spins == 65) t0 = std::chrono::steady_clock::now();
        if (spins < 4096) {
            std::this_thread::yield();
            continue;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(50));
        if (limit > 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(limit)) {
            const Progress& P = m->progress[(size_t)p];
            const int it = m->progress[(size_t)r].iteration.load();
            m->timed_out.store(true);
            stop_everybody(m, r);
            return rank_fail(m, r, S2D_E_STATE, "rank %d (device %d) stopped answering: it has not issued gradient exchange %u (iteration %d) "
                                                "%d ms after rank %d asked for it; it was last seen at '%s', iteration %d", p,
                             m->devices[(size_t)p], seq, it, limit, r, phase_name(P.phase.load()), P.iteration.load());
        }
    }
    return S2D_OK;
}

// The iteration's exchange: the partial gradient rows of splats this rank shares go to their other holders, theirs
// come here, and every holder adds the partials in rank order.  Stream-ordered: the rank gathers its rows into one of
// two send buffers and records an event; on its OWN stream it waits for each neighbour's event of the same exchange and
// copies its segment out of the neighbour's buffer.  The rank threads do not meet: a receiver only makes sure, pairwise,
// that the sender's thread has already recorded that event (sent_seq) -- all `iters` iterations of a call are queued
// without a barrier, the refreshes stay the only rendezvous.  Buffer b of exchange k is gathered again at k + 2: by then
// this rank's stream has waited for every neighbour's event k + 1, which that neighbour recorded behind its copy of k
// (exchanges are symmetric: who receives from a rank also sends to it), and the neighbour's thread called
// hipStreamWaitEvent for k before it recorded k + 1, so re-recording the event at k + 2 cannot overtake that call.
int exchange_grads(s2d_multi* m, int r)
{
    HaloRank& H = m->halo[(size_t)r];
    if (!H.any_exchange) return S2D_OK;
    s2d_ctx* c = m->ctx[(size_t)r];
    hipStream_t stream = (hipStream_t)s2d_stream(c);
    const unsigned seq = ++H.seq;
    const int b = (int)((seq - 1u) & 1u);
    if (H.total) {
        if (int rc = s2d_rows_gather(c, S2D_ROWS_GRADS, H.d_send_ids.p, H.total, H.d_send[b].p)) return rc;
        MHIP(m, r, hipEventRecord(H.ev_sent[b], stream));
    }
    m->sent_seq[(size_t)r].store(seq, std::memory_order_release);
    for (int p = 0; p < m->world; p++) {
        const int cnt = H.splits[(size_t)p];
        if (p == r || cnt == 0) continue;
        if (int rc = wait_issued(m, r, p, seq)) return rc;
        const HaloRank& P = m->halo[(size_t)p];
        MHIP(m, r, hipStreamWaitEvent(stream, P.ev_sent[b], 0));
        MHIP(m, r, rank_copy(m, H.d_recv.p + (size_t)H.offsets[(size_t)p] * 9, r, P.d_send[b].p + (size_t)P.offsets[(size_t)r] * 9, p,
                             (size_t)cnt * 9 * sizeof(float), stream));
    }
    if (H.n_rows)
        if (int rc = s2d_grads_combine(c, H.d_rows.p, H.n_rows, H.d_src.p, m->world, H.d_recv.p)) return rc;
    return S2D_OK;
}

// One rank's share of s2d_multi_step: `iters` frames of main.cpp:334 on its rows.
//
// Replicated state over RCCL: a rank that fails before it has queued its share of a collective leaves the others waiting
// inside theirs for good.  The failing rank publishes the stop and aborts ITS communicator (stop_everybody); every other
// rank aborts its own as soon as one of its bounded waits sees the stop (RCCL then ends the kernels that wait for the
// missing rank), and no rank submits a collective once comms_aborted is set.  The handle is dead afterwards --
// s2d_multi_destroy and a new s2d_multi_create bring it back.
int rank_step(s2d_multi* m, int rank)
{
    s2d_ctx* c = m->ctx[(size_t)rank];
    const uint32_t bwd_flags = (m->step_flags & S2D_STEP_OPTIMIZE_OPACITY) ? 0u : S2D_BWD_SKIP_OPACITY_GRAD; // main.cpp:735
    float* grads = (float*)s2d_grads_device_ptr(c);
    hipStream_t stream = (hipStream_t)s2d_stream(c);
    const size_t count = (size_t)m->n * 9;
    int rc = S2D_OK;
    int marks = 0; // progress marks recorded so far in this call
    for (int k = 0; k < m->step_iters && rc == S2D_OK && !m->barrier.broken; k++) {
        const bool last = k + 1 == m->step_iters;
        const int it = m->step_first_iter + k;
        at_phase(m, rank, PH_RASTER, it);
        rc = s2d_forward_backward(c, bwd_flags | (last ? 0u : S2D_FB_SKIP_IMAGE));
        if (rc != S2D_OK) break;
        if (rank == m->stall_rank && it == m->stall_iter) { // test hook: this rank's thread stops answering here
            const auto t0 = std::chrono::steady_clock::now();
            while (!m->barrier.broken && (m->stall_for_ms < 0 || std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(m->stall_for_ms)))
                std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
        // the only exchange of the iteration
        if (m->scheme == SCHEME_OWNERSHIP) {
            at_phase(m, rank, PH_EXCHANGE, it);
            rc = exchange_grads(m, rank);
        } else if (m->scheme == SCHEME_REPLICATED) { // (a handle on one device goes through RCCL too: same code whatever N)
            at_phase(m, rank, PH_ALLREDUCE, it);
            if (m->share_gpu) {
                if (m->world > 1) rc = staged_all_reduce(m, rank, grads, count, stream);
            } else {
                // this rank's own slot: nobody frees what it holds while comms_aborted is clear, and once it is set no
                // collective is started any more
                const ncclComm_t comm = m->comms_aborted.load() ? nullptr : m->comms[(size_t)rank].load();
                if (m->comms_aborted.load()) { // a stop was published (the barrier breaks a moment later): not this rank's failure
                    abort_own_collective(m, rank);
                    rc = kStopped;
                } else if (!comm) {
                    rc = rank_fail(m, rank, S2D_E_STATE, "no communicator");
                } else if (m->rccl.AllReduce(grads, grads, count, ncclFloat, ncclSum, comm, stream) != ncclSuccess)
                    rc = rank_fail(m, rank, S2D_E_HIP, "ncclAllReduce failed");
            }
        }
        if (rc == S2D_OK) {
            at_phase(m, rank, PH_ADAM, it);
            rc = s2d_adam_step(c, m->step_flags);
        }
        if (rc == S2D_OK && m->scheme == SCHEME_OWNERSHIP && (m->hold_age + k + 1) % m->interval == 0) {
            at_phase(m, rank, PH_REFRESH, it);
            rc = refresh(m, rank);
        } else if (rc == S2D_OK && m->scheme != SCHEME_OWNERSHIP && (k + 1) % 64 == 0 && !last && prog_event(m, rank, 0)) {
            // no refresh paces these schemes: mark every 64th iteration on the stream and wait for the mark BEFORE the one
            // just recorded, so the host never runs more than 128 iterations ahead of the device, the device never waits
            // for the host, and a device that stopped is noticed within the stall limit
            MHIP(m, rank, hipEventRecord(prog_event(m, rank, marks & 1), stream));
            marks++;
            if (marks >= 2) {
                at_phase(m, rank, PH_DRAIN, it);
                rc = wait_event(m, rank, prog_event(m, rank, marks & 1), "running a batch of 64 iterations");
            }
        }
    }
    std::vector<double>& mine = m->sqerr[(size_t)rank];
    mine.assign((size_t)m->step_iters, 0.0);
    // A rank that failed on the way (not the finite guard below, which every replica sees alike and which leaves every
    // collective queued): the others may already sit in an all-reduce that will never get this rank's share
    if (rc != S2D_OK && rc != kStopped) stop_everybody(m, rank);
    if (rc == S2D_OK) {
        at_phase(m, rank, PH_DRAIN, m->step_first_iter + m->step_iters);
        rc = drain_now(m, rank, "finishing the iterations of this call");
    }
    if (rc == S2D_OK && m->step_iters > 0) rc = s2d_get_sqerr_trace(c, m->step_first_iter, m->step_iters, mine.data());
    if (rc == S2D_OK) rc = s2d_synchronize(c); // the finite guard, main.cpp:752-785
    if (rc != S2D_OK) m->barrier.abort();      // the other ranks stop at their next wait instead of sitting in it
    return rc;
}

void worker_main(s2d_multi* m, int rank)
{
    int seen = 0;
    for (;;) {
        int cmd;
        {
            std::unique_lock<std::mutex> lk(m->m);
            m->cv_cmd.wait(lk, [&] { return m->cmd_seq != seen; });
            seen = m->cmd_seq;
            cmd = m->cmd;
        }
        if (cmd == CMD_QUIT) {
            (void)hipSetDevice(m->devices[(size_t)rank]);
            m->halo[(size_t)rank].release();
            return;
        }
        int rc = S2D_OK;
        if (cmd == CMD_STEP) rc = rank_step(m, rank);
        if (cmd == CMD_HOLD) {
            at_phase(m, rank, PH_HOLD, 0);
            rc = hold_fresh(m, rank);
            if (rc != S2D_OK) m->barrier.abort();
        }
        if (cmd == CMD_FORWARD) { // the rank's rows of image0 from the current parameters (it holds every splat that reaches them)
            at_phase(m, rank, PH_FORWARD, 0);
            rc = s2d_forward(m->ctx[(size_t)rank]);
            if (rc == S2D_OK) rc = drain_now(m, rank, "rendering its rows");
            if (rc == S2D_OK) rc = s2d_synchronize(m->ctx[(size_t)rank]);
        }
        at_phase(m, rank, PH_DONE, 0);
        {
            std::lock_guard<std::mutex> lk(m->m);
            m->rank_rc[(size_t)rank] = rc;
            m->rank_done[(size_t)rank] = 1;
            m->done_count++;
        }
        m->cv_done.notify_one();
    }
}

// Hand `cmd` to every worker and wait for all of them -- with a watchdog: the ranks' own waits are bounded (wait_issued,
// wait_event, the barrier), but a rank can also sit inside a runtime call this file cannot bound (a synchronous copy in
// a context's list rebuild behind a kernel that never ends, a collective's submission).  When NO rank has changed phase
// for twice the stall limit the caller stops everybody, gives the ranks one more limit to come back, and otherwise
// returns without them: the handle is then `stuck` -- its abandoned threads and contexts are never freed (a thread
// inside a runtime call cannot be cancelled) -- and every later call is refused.
void run_command(s2d_multi* m, int cmd)
{
    for (std::string& s : m->rank_msg) s.clear();
    m->late = 0;
    m->comms_aborted.store(false); // (a dead handle never gets here)
    m->barrier.reset();
    m->barrier.timeout_ms = m->stall_ms.load() > 0 ? 2 * m->stall_ms.load() : 0;
    {
        std::lock_guard<std::mutex> lk(m->m);
        m->cmd = cmd;
        m->cmd_seq++;
        m->done_count = 0;
        std::fill(m->rank_done.begin(), m->rank_done.end(), 0);
    }
    m->cv_cmd.notify_all();
    std::unique_lock<std::mutex> lk(m->m);
    const auto all_done = [&] { return m->done_count == m->world; };
    uint64_t seen = 0;
    auto moved = std::chrono::steady_clock::now();
    bool stopping = false;
    for (;;) {
        const int limit = m->stall_ms.load();
        if (limit <= 0) {
            m->cv_done.wait(lk, all_done);
            return;
        }
        if (m->cv_done.wait_for(lk, std::chrono::milliseconds(std::max(10, std::min(250, limit / 4))), all_done)) return;
        uint64_t ticks = (uint64_t)m->done_count;
        for (int r = 0; r < m->world; r++) ticks += m->progress[(size_t)r].ticks.load(std::memory_order_acquire);
        const auto now = std::chrono::steady_clock::now();
        if (ticks != seen) {
            seen = ticks;
            moved = now;
            continue;
        }
        const long long quiet = std::chrono::duration_cast<std::chrono::milliseconds>(now - moved).count();
        if (!stopping && quiet > 2LL * limit + 1000) {
            stopping = true;
            moved = now;
            m->timed_out.store(true);
            m->comms_aborted.store(true);
            m->barrier.abort();
            // last resort for a rank that sits inside a collective's submission or behind its kernel and cannot look up:
            // take its communicator out of its slot and abort it from here
            if (m->scheme == SCHEME_REPLICATED && !m->share_gpu)
                for (int r = 0; r < m->n_comms; r++)
                    if (!m->rank_done[(size_t)r]) {
                        const ncclComm_t c = m->comms[(size_t)r].exchange(nullptr);
                        if (c) m->collective_lost.store(true);
                        if (c && m->rccl.CommAbort) (void)m->rccl.CommAbort(c);
                    }
        } else if (stopping && quiet > (long long)limit + 1000) {
            std::string who;
            for (int r = 0; r < m->world; r++)
                if (!m->rank_done[(size_t)r]) {
                    const Progress& P = m->progress[(size_t)r];
                    char one[128];
                    snprintf(one, sizeof(one), "%srank %d (device %d) at '%s', iteration %d", who.empty() ? "" : "; ", r, m->devices[(size_t)r],
                             phase_name(P.phase.load()), P.iteration.load());
                    who += one;
                    m->rank_rc[(size_t)r] = S2D_E_STATE; // (under m->m, like the workers' own writes; their message strings are theirs)
                }
            m->stuck = true;
            m->dead = true;
            snprintf(m->err, sizeof(m->err), "no rank made progress for %d ms and the stop was not answered: %s; the handle is abandoned "
                                             "(its threads and device memory are not freed)", 3 * limit + 2000, who.c_str());
            return;
        }
    }
}

int first_failure(s2d_multi* m, const char* what)
{
    if (m->stuck) return S2D_E_STATE; // run_command wrote the report
    for (int r = 0; r < m->world; r++) {
        const int rc = m->rank_rc[(size_t)r];
        if (rc != S2D_OK && rc != kStopped)
            return mfail(m, rc, "%s on rank %d (device %d): %s", what, r, m->devices[(size_t)r],
                         m->rank_msg[(size_t)r].empty() ? s2d_last_error(m->ctx[(size_t)r]) : m->rank_msg[(size_t)r].c_str());
    }
    for (int r = 0; r < m->world; r++)
        if (m->rank_rc[(size_t)r] != S2D_OK) return mfail(m, S2D_E_STATE, "%s: rank %d stopped without a failing rank", what, r);
    return S2D_OK;
}

// Slab ownership: make the hold sets if the replicas were (re)filled since the last time.
int ensure_hold(s2d_multi* m)
{
    if (m->scheme != SCHEME_OWNERSHIP || m->hold_valid) return S2D_OK;
    run_command(m, CMD_HOLD);
    if (int rc = first_failure(m, "making the hold sets")) return rc;
    m->hold_valid = true;
    m->hold_age = 0;
    return S2D_OK;
}

// The s2d_multi_* calls select devices (hipSetDevice is per thread) from the CALLER's thread: put its device back on the
// way out, a caller that shares the thread with other HIP code (torch) does not expect it to have moved.
struct DeviceGuard {
    int prev = -1;
    DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

int refuse_dead(s2d_multi* m)
{
    if (m->stuck) return S2D_E_STATE; // keeps the watchdog's report
    return mfail(m, S2D_E_STATE, "%s; s2d_multi_destroy this handle and create a new one",
                 m->collective_lost.load() ? "a collective of this handle was aborted after a rank failed: its communicators are gone"
                                           : "a rank of this handle stopped answering and the ranks no longer agree on where the run stands");
}

inline bool lowest_holder(uint32_t mask, int r) { return mask != 0u && (mask & (0u - mask)) == (1u << r); }

// Slab ownership: the complete parameter or Adam array, every row from its lowest-ranked holder.
template <typename Row, typename Get>
int assemble(s2d_multi* m, Row* out, Get get)
{
    std::vector<Row> tmp((size_t)m->n);
    for (int r = 0; r < m->world; r++) {
        if (int rc = get(m->ctx[(size_t)r], tmp.data())) return mfail(m, rc, "reading rank %d: %s", r, s2d_last_error(m->ctx[(size_t)r]));
        const HaloRank& H = m->halo[(size_t)r];
        for (const int32_t i : H.held)
            if (lowest_holder(H.mask[(size_t)i], r)) out[(size_t)i] = tmp[(size_t)i];
    }
    return S2D_OK;
}

} // namespace

extern "C" {

int s2d_multi_create(const s2d_config* cfg, const int32_t* devices, int32_t n_devices, uint32_t flags, s2d_multi** out)
{
    if (!cfg || !out || !devices || n_devices < 1 || n_devices > 32 || cfg->struct_size != sizeof(s2d_config)) return S2D_E_INVALID;
    *out = nullptr;
    if (cfg->row_begin != 0 || cfg->row_end != 0 || cfg->stream != nullptr) return S2D_E_INVALID; // the handle cuts the slabs itself
    DeviceGuard guard;
    s2d_multi* m = new (std::nothrow) s2d_multi();
    if (!m) return S2D_E_NOMEM;
    *out = m; // handed out even on failure so that s2d_multi_last_error works; the caller destroys it
    m->world = n_devices;
    m->W = cfg->width;
    m->H = cfg->height;
    m->n = cfg->n_splats;
    m->share_gpu = (flags & S2D_MULTI_SHARE_GPU) != 0;
    m->scheme = (flags & S2D_MULTI_REPLICATED) ? SCHEME_REPLICATED : (n_devices > 1 ? SCHEME_OWNERSHIP : SCHEME_NONE);
    m->devices.assign(devices, devices + n_devices);
    if (m->share_gpu)
        for (int& d : m->devices) d = devices[0];
    if ((m->H + 15) / 16 < m->world) return mfail(m, S2D_E_INVALID, "%d devices for %d tile rows", m->world, (m->H + 15) / 16);
    m->ctx.assign((size_t)m->world, nullptr);
    m->row_begin.resize((size_t)m->world);
    m->row_end.resize((size_t)m->world);
    m->rank_rc.assign((size_t)m->world, S2D_OK);
    m->rank_msg.resize((size_t)m->world);
    m->sqerr.resize((size_t)m->world);
    m->halo.resize((size_t)m->world);
    m->barrier.n = m->world;
    m->sent_seq.reset(new std::atomic<unsigned>[(size_t)m->world]);
    for (int r = 0; r < m->world; r++) m->sent_seq[(size_t)r].store(0u);
    m->progress.reset(new Progress[(size_t)m->world]);
    m->rank_done.assign((size_t)m->world, 0);
    if (const char* e = getenv("S2D_MULTI_STALL_TIMEOUT_MS")) m->stall_ms = std::max(0, atoi(e));
    m->bounds.resize((size_t)m->world + 1);
    for (int r = 0; r < m->world; r++) {
        s2d_config c = *cfg;
        c.device = m->devices[(size_t)r];
        slab_rows(m->H, r, m->world, &c.row_begin, &c.row_end);
        m->row_begin[(size_t)r] = c.row_begin;
        m->row_end[(size_t)r] = c.row_end;
        m->bounds[(size_t)r] = c.row_begin;
        const int rc = s2d_create(&c, &m->ctx[(size_t)r]);
        if (rc != S2D_OK)
            return mfail(m, rc, "s2d_create for device %d, rows %d..%d: %s", c.device, c.row_begin, c.row_end,
                         m->ctx[(size_t)r] ? s2d_last_error(m->ctx[(size_t)r]) : "rejected configuration");
    }
    m->bounds[(size_t)m->world] = m->H;
    // Adam moves a parameter by at most lr * |m^| / sqrt(v^) <= 2.35 * lr per step for beta = (0.9, 0.99); pos.y moves by
    // that and reach = 3 * max(sx, sy) + 2 by three times that: the margin must outlast one refresh interval
    const float lr = cfg->training_rate > 0.0f ? cfg->training_rate : 0.05f;
    m->margin = std::max(8.0f, 1.1f * 2.35f * 4.0f * lr * (float)m->interval);
    if (m->scheme == SCHEME_REPLICATED && !m->share_gpu) {
        std::string why;
        if (!load_rccl(&m->rccl, &why)) return mfail(m, S2D_E_HIP, "%s", why.c_str());
        std::vector<ncclComm_t> made((size_t)m->world, nullptr);
        const ncclResult_t nrc = m->rccl.CommInitAll(made.data(), m->world, m->devices.data());
        if (nrc == ncclSuccess) {
            m->comms.reset(new std::atomic<ncclComm_t>[(size_t)m->world]);
            for (int r = 0; r < m->world; r++) m->comms[(size_t)r].store(made[(size_t)r]);
            m->n_comms = m->world;
        } else {
            return mfail(m, S2D_E_HIP, "ncclCommInitAll over %d devices: %s (RCCL takes one rank per GPU; S2D_MULTI_SHARE_GPU rehearses "
                                       "on fewer)", m->world, m->rccl.GetErrorString(nrc));
        }
    } else if (m->scheme == SCHEME_REPLICATED && m->world > 1) {
        m->host_grads.assign((size_t)m->world, nullptr);
        if (hipSetDevice(m->devices[0]) != hipSuccess) return mfail(m, S2D_E_HIP, "hipSetDevice(%d)", m->devices[0]);
        for (int r = 0; r < m->world; r++)
            if (hipHostMalloc((void**)&m->host_grads[(size_t)r], (size_t)m->n * 9 * sizeof(float) + 16, hipHostMallocDefault) != hipSuccess)
                return mfail(m, S2D_E_NOMEM, "host staging buffers for %d ranks", m->world);
    } else if (m->scheme == SCHEME_OWNERSHIP) {
        // one rank per GPU, and direct peer-to-peer copies where the fabric allows them (the copies work without, staged)
        for (int a = 0; a < m->world; a++)
            for (int b = 0; b < m->world; b++) {
                const int da = m->devices[(size_t)a], db = m->devices[(size_t)b];
                if (a == b || da == db) {
                    if (a != b && !m->share_gpu) return mfail(m, S2D_E_INVALID, "device %d listed twice (S2D_MULTI_SHARE_GPU rehearses several ranks on one GPU)", da);
                    continue;
                }
                int can = 0;
                if (hipSetDevice(da) != hipSuccess) return mfail(m, S2D_E_HIP, "hipSetDevice(%d)", da);
                if (hipDeviceCanAccessPeer(&can, da, db) == hipSuccess && can) {
                    const hipError_t e = hipDeviceEnablePeerAccess(db, 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError(); // copies are staged then
                }
            }
    }
    // the events of the bounded stream waits (three per rank, prog_event())
    m->ev_prog.assign((size_t)m->world * 3, nullptr);
    for (int r = 0; r < m->world; r++) {
        if (hipSetDevice(m->devices[(size_t)r]) != hipSuccess) return mfail(m, S2D_E_HIP, "hipSetDevice(%d)", m->devices[(size_t)r]);
        for (int k = 0; k < 3; k++)
            if (hipEventCreateWithFlags(&m->ev_prog[(size_t)r * 3 + (size_t)k], hipEventDisableTiming) != hipSuccess)
                return mfail(m, S2D_E_HIP, "hipEventCreate on device %d", m->devices[(size_t)r]);
    }
    for (int r = 0; r < m->world; r++) m->workers.emplace_back(worker_main, m, r);
    return S2D_OK;
}

void s2d_multi_destroy(s2d_multi* m)
{
    if (!m) return;
    DeviceGuard guard;
    if (m->stuck) {
        // some worker sits inside a runtime call that never returned: it cannot be joined, and what it may still touch
        // -- the handle, its context, its device memory -- cannot be freed under it.  Abandon all of it.
        for (auto& t : m->workers) t.detach();
        return;
    }
    if (!m->workers.empty()) {
        {
            std::lock_guard<std::mutex> lk(m->m);
            m->cmd = CMD_QUIT;
            m->cmd_seq++;
        }
        m->cv_cmd.notify_all();
        for (auto& t : m->workers) t.join();
    }
    for (int r = 0; r < m->n_comms; r++)
        if (const ncclComm_t c = m->comms[(size_t)r].exchange(nullptr)) m->rccl.CommDestroy(c);
    for (size_t k = 0; k < m->ev_prog.size(); k++)
        if (m->ev_prog[k] && hipSetDevice(m->devices[k / 3]) == hipSuccess) (void)hipEventDestroy(m->ev_prog[k]);
    for (float* p : m->host_grads)
        if (p) (void)hipHostFree(p);
    for (s2d_ctx* c : m->ctx)
        if (c) s2d_destroy(c);
    delete m;
}

const char* s2d_multi_last_error(const s2d_multi* m) { return m ? m->err : "null handle"; }

int s2d_multi_device_count(const s2d_multi* m) { return m ? m->world : 0; }

int s2d_multi_set_stall_timeout(s2d_multi* m, int32_t milliseconds)
{
    if (!m || milliseconds < 0) return S2D_E_INVALID;
    m->stall_ms = milliseconds;
    return S2D_OK;
}

int s2d_multi_device_info(s2d_multi* m, int32_t rank, int32_t* device, int32_t* row_begin, int32_t* row_end, char* pci_bus_id,
                          int32_t pci_capacity, char* name, int32_t name_capacity)
{
    if (!m || rank < 0 || rank >= m->world) return S2D_E_INVALID;
    const int dev = m->devices[(size_t)rank];
    if (device) *device = dev;
    if (row_begin) *row_begin = m->row_begin[(size_t)rank];
    if (row_end) *row_end = m->row_end[(size_t)rank];
    if (pci_bus_id && pci_capacity > 0) {
        pci_bus_id[0] = 0;
        if (hipDeviceGetPCIBusId(pci_bus_id, pci_capacity, dev) != hipSuccess) {
            (void)hipGetLastError();
            pci_bus_id[0] = 0;
        }
    }
    if (name && name_capacity > 0) {
        hipDeviceProp_t prop;
        name[0] = 0;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) snprintf(name, (size_t)name_capacity, "%s", prop.name);
        else (void)hipGetLastError();
    }
    return S2D_OK;
}

int s2d_test_multi_stall(s2d_multi* m, int32_t rank, int32_t iteration, int32_t milliseconds)
{
    if (!m) return S2D_E_INVALID;
    m->stall_rank = rank;
    m->stall_iter = iteration;
    m->stall_for_ms = milliseconds;
    return S2D_OK;
}

int s2d_multi_exchange_info(s2d_multi* m, int64_t* out4)
{
    if (!m || !out4) return S2D_E_INVALID;
    out4[0] = m->scheme;
    out4[1] = out4[2] = out4[3] = 0;
    if (m->scheme == SCHEME_OWNERSHIP && m->hold_valid)
        for (const HaloRank& H : m->halo) {
            out4[1] += H.total;
            out4[2] += H.handed;
            out4[3] += (int64_t)H.held.size();
        }
    else
        out4[3] = (int64_t)m->n * m->world;
    return S2D_OK;
}

// The calls below address every replica in turn from the caller's thread (the workers are idle between commands).
#define S2D_EACH(m, what, call)                                                                                      \
    do {                                                                                                             \
        for (int r_ = 0; r_ < (m)->world; r_++) {                                                                    \
            s2d_ctx* c = (m)->ctx[(size_t)r_];                                                                       \
            const int rc_ = (call);                                                                                  \
            if (rc_ != S2D_OK)                                                                                       \
                return mfail((m), rc_, "%s on rank %d (device %d): %s", what, r_, (m)->devices[(size_t)r_], s2d_last_error(c)); \
        }                                                                                                            \
    } while (0)

int s2d_multi_set_target(s2d_multi* m, const float* rgba32f)
{
    if (!m || !rgba32f) return S2D_E_INVALID;
    DeviceGuard guard;
    S2D_EACH(m, "s2d_set_target", s2d_set_target(c, rgba32f));
    return S2D_OK;
}

int s2d_multi_set_target_synthetic(s2d_multi* m)
{
    if (!m) return S2D_E_INVALID;
    DeviceGuard guard;
    S2D_EACH(m, "s2d_set_target_synthetic", s2d_set_target_synthetic(c));
    return S2D_OK;
}

int s2d_multi_init_splats(s2d_multi* m)
{
    if (!m) return S2D_E_INVALID;
    DeviceGuard guard;
    S2D_EACH(m, "s2d_init_splats", s2d_init_splats(c)); // every replica: the same deterministic init(), main.cpp:280-305
    m->iterations = 0;
    m->failed_step = false;
    m->hold_valid = false; // every replica is complete again: new hold sets at the next step
    return S2D_OK;
}

int s2d_multi_get_adam(s2d_multi* m, s2d_splat_adam* adams, float* beta1t, float* beta2t, int32_t* iterations)
{
    if (!m) return S2D_E_INVALID;
    DeviceGuard guard;
    if (m->scheme == SCHEME_OWNERSHIP && m->hold_valid && adams) {
        if (int rc = assemble(m, adams, [](s2d_ctx* c, s2d_splat_adam* p) { return s2d_get_adam(c, p, nullptr, nullptr, nullptr); })) return rc;
        adams = nullptr;
    }
    if (int rc = s2d_get_adam(m->ctx[0], adams, beta1t, beta2t, iterations)) return mfail(m, rc, "s2d_get_adam: %s", s2d_last_error(m->ctx[0]));
    return S2D_OK;
}

int s2d_multi_set_splats(s2d_multi* m, const s2d_splat* splats)
{
    if (!m || (!splats && m->n)) return S2D_E_INVALID;
    DeviceGuard guard;
    if (m->scheme == SCHEME_OWNERSHIP && m->hold_valid) {
        // the replicas are about to hold everything again: complete their Adam state first (a rank has current
        // moments only for the splats it holds)
        std::vector<s2d_splat_adam> full((size_t)m->n);
        float b1 = 0.f, b2 = 0.f;
        int32_t it = 0;
        if (int rc = s2d_multi_get_adam(m, full.data(), &b1, &b2, &it)) return rc;
        S2D_EACH(m, "s2d_set_adam", s2d_set_adam(c, full.data(), b1, b2, it));
    }
    S2D_EACH(m, "s2d_set_splats", s2d_set_splats(c, splats));
    m->hold_valid = false;
    return S2D_OK;
}

int s2d_multi_get_splats(s2d_multi* m, s2d_splat* splats)
{
    if (!m || (!splats && m->n)) return S2D_E_INVALID;
    DeviceGuard guard;
    if (m->scheme == SCHEME_OWNERSHIP && m->hold_valid)
        return assemble(m, splats, [](s2d_ctx* c, s2d_splat* p) { return s2d_get_splats(c, p); });
    if (int rc = s2d_get_splats(m->ctx[0], splats)) return mfail(m, rc, "s2d_get_splats: %s", s2d_last_error(m->ctx[0]));
    return S2D_OK; // the replicas are bit-identical: any one of them
}

int s2d_multi_set_adam(s2d_multi* m, const s2d_splat_adam* adams, float beta1t, float beta2t, int32_t iterations)
{
    if (!m || (!adams && m->n) || iterations < 0) return S2D_E_INVALID;
    DeviceGuard guard;
    S2D_EACH(m, "s2d_set_adam", s2d_set_adam(c, adams, beta1t, beta2t, iterations)); // complete on every rank; hold sets unaffected
    m->iterations = iterations;
    m->failed_step = false; // every rank's counters are alike again (the caller sets the splats as well: include/splat2d.h)
    return S2D_OK;
}

int s2d_multi_step(s2d_multi* m, int32_t iters, uint32_t flags, double* mse_out)
{
    if (!m || iters < 0 || iters > (1 << 16)) return S2D_E_INVALID;
    if (m->dead) return refuse_dead(m);
    if (m->failed_step)
        return mfail(m, S2D_E_STATE, "the last s2d_multi_step failed and left the ranks at different iterations: s2d_multi_init_splats, or "
                                     "s2d_multi_set_splats + s2d_multi_set_adam, first");
    DeviceGuard guard;
    if (int rc = ensure_hold(m)) return rc;
    m->step_iters = iters;
    m->step_flags = flags;
    m->step_first_iter = m->iterations;
    run_command(m, CMD_STEP);
    if (int rc = first_failure(m, "s2d_multi_step")) {
        // Where the run stands now: a non-finite stop winds the counters of the rank that holds the splat back to the
        // failing iteration (s2d_api.hip judge_status); with slab ownership the other ranks did not see it and are ahead.
        // The earliest count is the iteration at which the reference abort()ed; the state is for inspection only, and
        // s2d_multi_set_adam (which sets every rank's counters alike) comes before any further step.
        int32_t earliest = INT32_MAX;
        for (s2d_ctx* c : m->ctx) {
            int32_t it = 0;
            if (s2d_get_adam(c, nullptr, nullptr, nullptr, &it) == S2D_OK) earliest = std::min(earliest, it);
        }
        if (earliest != INT32_MAX) m->iterations = earliest;
        if (m->collective_lost.load() || m->timed_out.load()) m->dead = true;
        m->failed_step = true;
        return rc;
    }
    m->iterations += iters;
    m->hold_age += iters;
    if (mse_out) {
        const double norm = (double)((long long)m->H * m->W * 3);
        for (int k = 0; k < iters; k++) {
            double sum = 0.0;
            for (int r = 0; r < m->world; r++) sum += m->sqerr[(size_t)r][(size_t)k]; // slab order: a fixed order
            mse_out[k] = sum / norm; // main.cpp:805
        }
    }
    return S2D_OK;
}

int s2d_multi_forward(s2d_multi* m)
{
    if (!m) return S2D_E_INVALID;
    if (m->dead) return refuse_dead(m);
    DeviceGuard guard;
    // slab ownership: after set_splats / init_splats the hold sets (and with them each context's list of the splats it
    // projects and rasterises) are stale until they are made afresh -- a splat that now reaches a rank's rows but was not
    // in its old set would be missing from that slab of the image
    if (int rc = ensure_hold(m)) return rc;
    run_command(m, CMD_FORWARD);
    return first_failure(m, "s2d_multi_forward");
}

int s2d_multi_get_image(s2d_multi* m, float* rgba32f)
{
    if (!m || !rgba32f) return S2D_E_INVALID;
    DeviceGuard guard;
    // every context holds (and returns) its own rows; together they tile the image
    const size_t row = (size_t)m->W * 4;
    for (int r = 0; r < m->world; r++)
        if (int rc = s2d_get_image_rows(m->ctx[(size_t)r], rgba32f + row * (size_t)m->row_begin[(size_t)r]))
            return mfail(m, rc, "s2d_get_image_rows on rank %d: %s", r, s2d_last_error(m->ctx[(size_t)r]));
    return S2D_OK;
}

} // extern "C"
