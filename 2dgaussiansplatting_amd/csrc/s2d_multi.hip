// s2d_multi.hip -- several GPUs behind ONE handle (include/splat2d.h "s2d_multi_*"; SURVEY.md section 8b/8e: "device
// list ... multi-GPU fan-out is internal").
//
// A reference-side caller keeps its single-threaded frame loop (main.cpp:334) and gets N GPUs by swapping s2d_ctx for
// s2d_multi: the image is cut into N row slabs (whole 16-pixel tile rows), every device gets an ordinary context for
// its slab with the splats and the Adam state replicated, and per iteration every device rasterises its rows forward
// and backward, the N x 9 fp32 gradient arrays are summed in place by an RCCL all-reduce over xGMI (ncclAllReduce on
// each context's own stream, between s2d_forward_backward and s2d_adam_step), and every device applies the identical
// Adam step -- so the replicas stay bit-identical without ever exchanging parameters or framebuffers (north_star's
// scheme).  One worker thread per device keeps its context's calls in order; the caller's thread only hands out
// commands and adds up the slabs' squared errors.  RCCL is loaded with dlopen when the first multi handle is created,
// so single-GPU users of the library never load it.
//
// S2D_MULTI_SHARE_GPU (rehearsal on a box with fewer GPUs than ranks -- RCCL takes one rank per GPU): all ranks on the
// first listed device, the gradient sum staged through pinned host memory in rank order.
#include "../../include/splat2d.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h> // types only; the entry points are resolved at run time

#include <atomic>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
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
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

bool load_rccl(Rccl* r, std::string* why)
{
    static std::mutex m;
    static Rccl cached;
    std::lock_guard<std::mutex> lk(m);
    if (!cached.lib) {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            cached.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (cached.lib) break;
        }
        if (!cached.lib) {
            *why = std::string("cannot load librccl: ") + dlerror();
            return false;
        }
        cached.CommInitAll = (decltype(cached.CommInitAll))dlsym(cached.lib, "ncclCommInitAll");
        cached.CommDestroy = (decltype(cached.CommDestroy))dlsym(cached.lib, "ncclCommDestroy");
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

// Reusable barrier of the rank threads; abort() releases everybody for good once a rank has failed.
struct Barrier {
    std::mutex m;
    std::condition_variable cv;
    int n = 1, waiting = 0, generation = 0;
    bool broken = false;
    void wait()
    {
        std::unique_lock<std::mutex> lk(m);
        if (broken) return;
        const int gen = generation;
        if (++waiting == n) {
            waiting = 0;
            generation++;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return gen != generation || broken; });
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
        waiting = 0;
    }
};

enum Command { CMD_NONE = 0, CMD_STEP, CMD_QUIT };

} // namespace

struct s2d_multi {
    int world = 0, W = 0, H = 0, n = 0;
    bool share_gpu = false;
    std::vector<int> devices;
    std::vector<s2d_ctx*> ctx;
    std::vector<int> row_begin, row_end;
    Rccl rccl;
    std::vector<ncclComm_t> comms;
    std::vector<float*> host_grads; // share-gpu: pinned copies of the ranks' partial gradients; [0] receives the sum
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
    std::vector<std::vector<double>> sqerr; // [rank][iteration of the call]: partial squared errors
    int iterations = 0;
    char err[512] = {0};
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

// Rows [r0, r1) of rank `rank`: whole 16-pixel tile rows, as even as possible (== distributed.slab_rows).
void slab_rows(int height, int rank, int world, int* r0, int* r1)
{
    const int tile_rows = (height + 15) / 16;
    *r0 = (int)((long long)tile_rows * rank / world) * 16;
    *r1 = (int)((long long)tile_rows * (rank + 1) / world) * 16;
    if (*r1 > height) *r1 = height;
}

// The sum RCCL would form, through host memory: every rank copies its partial gradients out, rank 0 adds them in rank
// order, every rank copies the sum back in.
bool staged_all_reduce(s2d_multi* m, int rank, float* grads, size_t count, hipStream_t stream)
{
    bool ok = hipMemcpyAsync(m->host_grads[(size_t)rank], grads, count * sizeof(float), hipMemcpyDeviceToHost, stream) == hipSuccess &&
              hipStreamSynchronize(stream) == hipSuccess;
    m->barrier.wait();
    if (rank == 0)
        for (int q = 1; q < m->world; q++) {
            const float* src = m->host_grads[(size_t)q];
            float* dst = m->host_grads[0];
            for (size_t k = 0; k < count; k++) dst[k] += src[k];
        }
    m->barrier.wait();
    ok = ok && hipMemcpyAsync(grads, m->host_grads[0], count * sizeof(float), hipMemcpyHostToDevice, stream) == hipSuccess &&
         hipStreamSynchronize(stream) == hipSuccess;
    m->barrier.wait(); // nobody overwrites its host copy before everybody has read the sum
    return ok;
}

// One rank's share of s2d_multi_step: `iters` frames of main.cpp:334 on its rows.
int rank_step(s2d_multi* m, int rank)
{
    s2d_ctx* c = m->ctx[(size_t)rank];
    const uint32_t bwd_flags = (m->step_flags & S2D_STEP_OPTIMIZE_OPACITY) ? 0u : S2D_BWD_SKIP_OPACITY_GRAD; // main.cpp:735
    float* grads = (float*)s2d_grads_device_ptr(c);
    hipStream_t stream = (hipStream_t)s2d_stream(c);
    const size_t count = (size_t)m->n * 9;
    int rc = S2D_OK;
    for (int k = 0; k < m->step_iters && rc == S2D_OK && !m->barrier.broken; k++) {
        const bool last = k + 1 == m->step_iters;
        rc = s2d_forward_backward(c, bwd_flags | (last ? 0u : S2D_FB_SKIP_IMAGE));
        if (rc != S2D_OK) break;
        {
            // the only exchange of the iteration: the sum of the slabs' partial gradients, in place (a handle on one
            // device goes through RCCL too: same code whatever the device count)
            if (m->share_gpu) {
                if (m->world > 1 && !staged_all_reduce(m, rank, grads, count, stream)) rc = S2D_E_HIP;
            } else if (m->rccl.AllReduce(grads, grads, count, ncclFloat, ncclSum, m->comms[(size_t)rank], stream) != ncclSuccess) {
                rc = S2D_E_HIP;
            }
        }
        if (rc == S2D_OK) rc = s2d_adam_step(c, m->step_flags);
    }
    std::vector<double>& mine = m->sqerr[(size_t)rank];
    mine.assign((size_t)m->step_iters, 0.0);
    if (rc == S2D_OK && m->step_iters > 0) rc = s2d_get_sqerr_trace(c, m->step_first_iter, m->step_iters, mine.data());
    if (rc == S2D_OK) rc = s2d_synchronize(c); // the finite guard, main.cpp:752-785
    if (rc != S2D_OK) m->barrier.abort();      // the other ranks stop at their next barrier instead of waiting
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
        if (cmd == CMD_QUIT) return;
        int rc = S2D_OK;
        if (cmd == CMD_STEP) rc = rank_step(m, rank);
        {
            std::lock_guard<std::mutex> lk(m->m);
            m->rank_rc[(size_t)rank] = rc;
            m->done_count++;
        }
        m->cv_done.notify_one();
    }
}

// Hand `cmd` to every worker and wait for all of them.
void run_command(s2d_multi* m, int cmd)
{
    {
        std::lock_guard<std::mutex> lk(m->m);
        m->cmd = cmd;
        m->cmd_seq++;
        m->done_count = 0;
    }
    m->cv_cmd.notify_all();
    std::unique_lock<std::mutex> lk(m->m);
    m->cv_done.wait(lk, [&] { return m->done_count == m->world; });
}

int first_failure(s2d_multi* m, const char* what)
{
    for (int r = 0; r < m->world; r++)
        if (m->rank_rc[(size_t)r] != S2D_OK)
            return mfail(m, m->rank_rc[(size_t)r], "%s on rank %d (device %d): %s", what, r, m->devices[(size_t)r],
                         s2d_last_error(m->ctx[(size_t)r]));
    return S2D_OK;
}

} // namespace

extern "C" {

int s2d_multi_create(const s2d_config* cfg, const int32_t* devices, int32_t n_devices, uint32_t flags, s2d_multi** out)
{
    if (!cfg || !out || !devices || n_devices < 1 || n_devices > 64 || cfg->struct_size != sizeof(s2d_config)) return S2D_E_INVALID;
    *out = nullptr;
    if (cfg->row_begin != 0 || cfg->row_end != 0 || cfg->stream != nullptr) return S2D_E_INVALID; // the handle cuts the slabs itself
    s2d_multi* m = new (std::nothrow) s2d_multi();
    if (!m) return S2D_E_NOMEM;
    *out = m; // handed out even on failure so that s2d_multi_last_error works; the caller destroys it
    m->world = n_devices;
    m->W = cfg->width;
    m->H = cfg->height;
    m->n = cfg->n_splats;
    m->share_gpu = (flags & S2D_MULTI_SHARE_GPU) != 0;
    m->devices.assign(devices, devices + n_devices);
    if (m->share_gpu)
        for (int& d : m->devices) d = devices[0];
    if ((m->H + 15) / 16 < m->world) return mfail(m, S2D_E_INVALID, "%d devices for %d tile rows", m->world, (m->H + 15) / 16);
    m->ctx.assign((size_t)m->world, nullptr);
    m->row_begin.resize((size_t)m->world);
    m->row_end.resize((size_t)m->world);
    m->rank_rc.assign((size_t)m->world, S2D_OK);
    m->sqerr.resize((size_t)m->world);
    m->barrier.n = m->world;
    for (int r = 0; r < m->world; r++) {
        s2d_config c = *cfg;
        c.device = m->devices[(size_t)r];
        slab_rows(m->H, r, m->world, &c.row_begin, &c.row_end);
        m->row_begin[(size_t)r] = c.row_begin;
        m->row_end[(size_t)r] = c.row_end;
        const int rc = s2d_create(&c, &m->ctx[(size_t)r]);
        if (rc != S2D_OK)
            return mfail(m, rc, "s2d_create for device %d, rows %d..%d: %s", c.device, c.row_begin, c.row_end,
                         m->ctx[(size_t)r] ? s2d_last_error(m->ctx[(size_t)r]) : "rejected configuration");
    }
    if (!m->share_gpu) {
        std::string why;
        if (!load_rccl(&m->rccl, &why)) return mfail(m, S2D_E_HIP, "%s", why.c_str());
        m->comms.assign((size_t)m->world, nullptr);
        const ncclResult_t nrc = m->rccl.CommInitAll(m->comms.data(), m->world, m->devices.data());
        if (nrc != ncclSuccess) {
            m->comms.clear();
            return mfail(m, S2D_E_HIP, "ncclCommInitAll over %d devices: %s (RCCL takes one rank per GPU; S2D_MULTI_SHARE_GPU rehearses "
                                       "on fewer)", m->world, m->rccl.GetErrorString(nrc));
        }
    } else if (m->world > 1) {
        m->host_grads.assign((size_t)m->world, nullptr);
        if (hipSetDevice(m->devices[0]) != hipSuccess) return mfail(m, S2D_E_HIP, "hipSetDevice(%d)", m->devices[0]);
        for (int r = 0; r < m->world; r++)
            if (hipHostMalloc((void**)&m->host_grads[(size_t)r], (size_t)m->n * 9 * sizeof(float) + 16, hipHostMallocDefault) != hipSuccess)
                return mfail(m, S2D_E_NOMEM, "host staging buffers for %d ranks", m->world);
    }
    for (int r = 0; r < m->world; r++) m->workers.emplace_back(worker_main, m, r);
    return S2D_OK;
}

void s2d_multi_destroy(s2d_multi* m)
{
    if (!m) return;
    if (!m->workers.empty()) {
        {
            std::lock_guard<std::mutex> lk(m->m);
            m->cmd = CMD_QUIT;
            m->cmd_seq++;
        }
        m->cv_cmd.notify_all();
        for (auto& t : m->workers) t.join();
    }
    for (ncclComm_t c : m->comms)
        if (c) m->rccl.CommDestroy(c);
    for (float* p : m->host_grads)
        if (p) (void)hipHostFree(p);
    for (s2d_ctx* c : m->ctx)
        if (c) s2d_destroy(c);
    delete m;
}

const char* s2d_multi_last_error(const s2d_multi* m) { return m ? m->err : "null handle"; }

int s2d_multi_device_count(const s2d_multi* m) { return m ? m->world : 0; }

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
    S2D_EACH(m, "s2d_set_target", s2d_set_target(c, rgba32f));
    return S2D_OK;
}

int s2d_multi_set_target_synthetic(s2d_multi* m)
{
    if (!m) return S2D_E_INVALID;
    S2D_EACH(m, "s2d_set_target_synthetic", s2d_set_target_synthetic(c));
    return S2D_OK;
}

int s2d_multi_init_splats(s2d_multi* m)
{
    if (!m) return S2D_E_INVALID;
    S2D_EACH(m, "s2d_init_splats", s2d_init_splats(c)); // every replica: the same deterministic init(), main.cpp:280-305
    m->iterations = 0;
    return S2D_OK;
}

int s2d_multi_set_splats(s2d_multi* m, const s2d_splat* splats)
{
    if (!m || (!splats && m->n)) return S2D_E_INVALID;
    S2D_EACH(m, "s2d_set_splats", s2d_set_splats(c, splats));
    return S2D_OK;
}

int s2d_multi_get_splats(s2d_multi* m, s2d_splat* splats)
{
    if (!m || (!splats && m->n)) return S2D_E_INVALID;
    if (int rc = s2d_get_splats(m->ctx[0], splats)) return mfail(m, rc, "s2d_get_splats: %s", s2d_last_error(m->ctx[0]));
    return S2D_OK; // the replicas are bit-identical: any one of them
}

int s2d_multi_set_adam(s2d_multi* m, const s2d_splat_adam* adams, float beta1t, float beta2t, int32_t iterations)
{
    if (!m || (!adams && m->n) || iterations < 0) return S2D_E_INVALID;
    S2D_EACH(m, "s2d_set_adam", s2d_set_adam(c, adams, beta1t, beta2t, iterations));
    m->iterations = iterations;
    return S2D_OK;
}

int s2d_multi_get_adam(s2d_multi* m, s2d_splat_adam* adams, float* beta1t, float* beta2t, int32_t* iterations)
{
    if (!m) return S2D_E_INVALID;
    if (int rc = s2d_get_adam(m->ctx[0], adams, beta1t, beta2t, iterations)) return mfail(m, rc, "s2d_get_adam: %s", s2d_last_error(m->ctx[0]));
    return S2D_OK;
}

int s2d_multi_step(s2d_multi* m, int32_t iters, uint32_t flags, double* mse_out)
{
    if (!m || iters < 0 || iters > (1 << 16)) return S2D_E_INVALID;
    m->step_iters = iters;
    m->step_flags = flags;
    m->step_first_iter = m->iterations;
    m->barrier.reset();
    run_command(m, CMD_STEP);
    if (int rc = first_failure(m, "s2d_multi_step")) {
        int32_t it = 0; // where the replicas stand now (a non-finite stop winds the counters back, s2d_api.hip)
        if (s2d_get_adam(m->ctx[0], nullptr, nullptr, nullptr, &it) == S2D_OK) m->iterations = it;
        return rc;
    }
    m->iterations += iters;
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

int s2d_multi_get_image(s2d_multi* m, float* rgba32f)
{
    if (!m || !rgba32f) return S2D_E_INVALID;
    // every context returns a full-size image that is zero outside its slab: take each one's rows
    std::vector<float> tmp;
    const size_t row = (size_t)m->W * 4;
    for (int r = 0; r < m->world; r++) {
        float* dst = rgba32f;
        if (r > 0) {
            tmp.resize(row * (size_t)m->H);
            dst = tmp.data();
        }
        if (int rc = s2d_get_image(m->ctx[(size_t)r], dst)) return mfail(m, rc, "s2d_get_image on rank %d: %s", r, s2d_last_error(m->ctx[(size_t)r]));
        if (r > 0)
            std::memcpy(rgba32f + row * (size_t)m->row_begin[(size_t)r], tmp.data() + row * (size_t)m->row_begin[(size_t)r],
                        row * (size_t)(m->row_end[(size_t)r] - m->row_begin[(size_t)r]) * sizeof(float));
    }
    return S2D_OK;
}

} // extern "C"
