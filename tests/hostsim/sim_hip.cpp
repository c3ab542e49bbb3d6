// sim_hip.cpp -- the host simulation of the HIP runtime calls csrc/s2d_multi.hip makes (see include/hip/hip_runtime.h).
// TEST INFRASTRUCTURE ONLY.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>

struct SimStream {
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    unsigned long long enqueued = 0, done = 0;
    bool quit = false;
    std::atomic<bool> abandon{false}; // the stream is being destroyed: event waits give up
    int device = 0;
    std::thread worker;
};

struct SimEvent {
    std::mutex m;
    std::condition_variable cv;
    unsigned long long recorded = 0, completed = 0;
};

namespace {
std::atomic<int> g_devices{8};
thread_local int t_device = 0;
std::atomic<unsigned long long> g_ops{0}, g_peer{0}, g_waits{0}, g_records{0};

void stream_main(SimStream* s)
{
    for (;;) {
        std::function<void()> op;
        {
            std::unique_lock<std::mutex> lk(s->m);
            s->cv.wait(lk, [&] { return s->quit || !s->q.empty(); });
            if (s->q.empty()) return; // quit, and drained
            op = std::move(s->q.front());
            s->q.pop_front();
        }
        op();
        {
            std::lock_guard<std::mutex> lk(s->m);
            s->done++;
        }
        s->cv.notify_all();
    }
}
} // namespace

hipStream_t sim_stream_create(int device)
{
    SimStream* s = new SimStream();
    s->device = device;
    s->worker = std::thread(stream_main, s);
    return s;
}

void sim_stream_destroy(hipStream_t s)
{
    if (!s) return;
    s->abandon.store(true);
    {
        std::lock_guard<std::mutex> lk(s->m);
        s->quit = true;
    }
    s->cv.notify_all();
    s->worker.join();
    delete s;
}

void sim_enqueue(hipStream_t s, std::function<void()> op)
{
    {
        std::lock_guard<std::mutex> lk(s->m);
        s->q.push_back(std::move(op));
        s->enqueued++;
    }
    g_ops.fetch_add(1, std::memory_order_relaxed);
    s->cv.notify_all();
}

namespace {
std::atomic<int> g_fail_dev{-1}, g_fail_iter{-1}, g_fail_ar_rank{-1};
std::atomic<long long> g_fail_ar_call{-1};
std::atomic<long long> g_ar_calls[64];
} // namespace

void sim_fail_forward_backward(int device, int iteration)
{
    g_fail_dev.store(device);
    g_fail_iter.store(iteration);
}

void sim_fail_allreduce(int rank, long long call)
{
    for (auto& c : g_ar_calls) c.store(0);
    g_fail_ar_rank.store(rank);
    g_fail_ar_call.store(call);
}

namespace {
std::mutex g_block_m;
std::condition_variable g_block_cv;
int g_block_dev = -1, g_block_iter = -1;
bool g_block_released = false;
} // namespace

void sim_block_forward_backward(int device, int iteration)
{
    std::lock_guard<std::mutex> lk(g_block_m);
    g_block_dev = device;
    g_block_iter = iteration;
    g_block_released = false;
}

void sim_release_blocked()
{
    {
        std::lock_guard<std::mutex> lk(g_block_m);
        g_block_released = true;
        g_block_dev = g_block_iter = -1;
    }
    g_block_cv.notify_all();
}

void sim_maybe_block(int device, int iteration)
{
    std::unique_lock<std::mutex> lk(g_block_m);
    if (device != g_block_dev || iteration != g_block_iter) return;
    g_block_cv.wait(lk, [] { return g_block_released; });
}

int sim_forward_backward_fails(int device, int iteration) { return device == g_fail_dev.load() && iteration == g_fail_iter.load(); }

int sim_allreduce_fails(int rank)
{
    if (rank < 0 || rank >= 64) return 0;
    const long long k = g_ar_calls[rank].fetch_add(1);
    return rank == g_fail_ar_rank.load() && k == g_fail_ar_call.load();
}

int sim_device_count() { return g_devices.load(); }
void sim_set_device_count(int n) { g_devices.store(n); }
SimCounters sim_counters() { return SimCounters{g_ops.load(), g_peer.load(), g_waits.load(), g_records.load()}; }

extern "C" {

const char* hipGetErrorString(hipError_t e)
{
    switch (e) {
    case hipSuccess: return "hipSuccess";
    case hipErrorInvalidValue: return "hipErrorInvalidValue";
    case hipErrorOutOfMemory: return "hipErrorOutOfMemory";
    case hipErrorNotReady: return "hipErrorNotReady";
    case hipErrorPeerAccessAlreadyEnabled: return "hipErrorPeerAccessAlreadyEnabled";
    }
    return "hipError?";
}

hipError_t hipGetLastError(void) { return hipSuccess; }

hipError_t hipSetDevice(int device)
{
    if (device < 0 || device >= g_devices.load()) return hipErrorInvalidValue;
    t_device = device;
    return hipSuccess;
}

hipError_t hipGetDevice(int* device)
{
    *device = t_device;
    return hipSuccess;
}

hipError_t hipGetDeviceProperties(hipDeviceProp_t* prop, int device)
{
    snprintf(prop->name, sizeof(prop->name), "simulated device %d", device);
    return hipSuccess;
}

hipError_t hipDeviceGetPCIBusId(char* out, int len, int device)
{
    snprintf(out, (size_t)len, "0000:%02x:00.0", device);
    return hipSuccess;
}

hipError_t hipDeviceCanAccessPeer(int* can, int, int)
{
    *can = 1;
    return hipSuccess;
}

hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }

hipError_t hipMalloc(void** p, size_t bytes)
{
    *p = malloc(bytes ? bytes : 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}

hipError_t hipFree(void* p)
{
    free(p);
    return hipSuccess;
}

hipError_t hipHostMalloc(void** p, size_t bytes, unsigned) { return hipMalloc(p, bytes); }
hipError_t hipHostFree(void* p) { return hipFree(p); }

hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind, hipStream_t stream)
{
    if (!stream) return hipErrorInvalidValue;
    sim_enqueue(stream, [=] { memcpy(dst, src, bytes); });
    return hipSuccess;
}

hipError_t hipMemcpyPeerAsync(void* dst, int, const void* src, int, size_t bytes, hipStream_t stream)
{
    if (!stream) return hipErrorInvalidValue;
    g_peer.fetch_add(1, std::memory_order_relaxed);
    sim_enqueue(stream, [=] { memcpy(dst, src, bytes); });
    return hipSuccess;
}

hipError_t hipStreamSynchronize(hipStream_t s)
{
    if (!s) return hipErrorInvalidValue;
    std::unique_lock<std::mutex> lk(s->m);
    const unsigned long long target = s->enqueued;
    s->cv.wait(lk, [&] { return s->done >= target; });
    return hipSuccess;
}

hipError_t hipEventCreateWithFlags(hipEvent_t* ev, unsigned)
{
    *ev = new SimEvent();
    return hipSuccess;
}

hipError_t hipEventCreate(hipEvent_t* ev) { return hipEventCreateWithFlags(ev, 0); }

hipError_t hipEventDestroy(hipEvent_t ev)
{
    delete ev;
    return hipSuccess;
}

hipError_t hipEventRecord(hipEvent_t ev, hipStream_t stream)
{
    if (!ev || !stream) return hipErrorInvalidValue;
    unsigned long long seq;
    {
        std::lock_guard<std::mutex> lk(ev->m);
        seq = ++ev->recorded;
    }
    g_records.fetch_add(1, std::memory_order_relaxed);
    sim_enqueue(stream, [ev, seq] {
        {
            std::lock_guard<std::mutex> lk(ev->m);
            if (ev->completed < seq) ev->completed = seq;
        }
        ev->cv.notify_all();
    });
    return hipSuccess;
}

hipError_t hipEventQuery(hipEvent_t ev)
{
    if (!ev) return hipErrorInvalidValue;
    std::lock_guard<std::mutex> lk(ev->m);
    return ev->completed >= ev->recorded ? hipSuccess : hipErrorNotReady;
}

hipError_t hipStreamWaitEvent(hipStream_t stream, hipEvent_t ev, unsigned)
{
    if (!ev || !stream) return hipErrorInvalidValue;
    unsigned long long target;
    {
        std::lock_guard<std::mutex> lk(ev->m);
        target = ev->recorded; // the latest record at the time of THIS call
    }
    g_waits.fetch_add(1, std::memory_order_relaxed);
    sim_enqueue(stream, [stream, ev, target] {
        std::unique_lock<std::mutex> lk(ev->m);
        while (ev->completed < target) {
            if (stream->abandon.load()) return;
            ev->cv.wait_for(lk, std::chrono::milliseconds(20));
        }
    });
    return hipSuccess;
}

} // extern "C"
