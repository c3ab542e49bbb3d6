// sim_rccl.cpp -- host simulation of the RCCL calls csrc/s2d_multi.hip makes (SONAME librccl.so.1: the library finds it
// already mapped, as it finds PyTorch's bundled RCCL).  TEST INFRASTRUCTURE ONLY.
//
// ncclAllReduce queues, on the caller's stream, an operation that waits until every rank of the group has queued its share
// and then leaves the sum -- formed in rank order -- in the rank's receive buffer.  A rank whose peer never arrives waits
// for ever, like the kernel of a real collective, until ITS communicator is aborted (ncclCommAbort).  Using a communicator
// after it was aborted or destroyed is what a real RCCL would answer with a crash: here it ends the test.
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

struct SimGroup {
    int n = 0;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    unsigned long long generation = 0;
    std::vector<const float*> send;
    std::vector<float> sum[2];
};

struct SimComm {
    SimGroup* group = nullptr;
    int rank = 0;
    std::atomic<bool> aborted{false}, destroyed{false};
};

extern "C" {

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist)
{
    for (int a = 0; a < ndev; a++)
        for (int b = a + 1; b < ndev; b++)
            if (devlist[a] == devlist[b]) return ncclInvalidArgument; // one rank per device, like RCCL
    SimGroup* g = new SimGroup();
    g->n = ndev;
    g->send.assign((size_t)ndev, nullptr);
    for (int r = 0; r < ndev; r++) {
        comms[r] = new SimComm();
        comms[r]->group = g;
        comms[r]->rank = r;
    }
    return ncclSuccess;
}

// (communicators and groups are never freed: an operation still queued on some stream may hold them)
ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    if (comm->aborted.load() || comm->destroyed.exchange(true)) {
        fprintf(stderr, "SIM RCCL: ncclCommDestroy on a communicator that was already aborted or destroyed\n");
        abort();
    }
    return ncclSuccess;
}

ncclResult_t ncclCommAbort(ncclComm_t comm)
{
    if (comm->destroyed.load() || comm->aborted.exchange(true)) {
        fprintf(stderr, "SIM RCCL: ncclCommAbort on a communicator that was already aborted or destroyed\n");
        abort();
    }
    {
        std::lock_guard<std::mutex> lk(comm->group->m); // pairs with the waiters' predicate checks
    }
    comm->group->cv.notify_all();
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t, ncclRedOp_t, ncclComm_t comm, hipStream_t stream)
{
    if (comm->aborted.load() || comm->destroyed.load()) {
        fprintf(stderr, "SIM RCCL: ncclAllReduce submitted on an aborted / destroyed communicator (a use-after-free on real RCCL)\n");
        abort();
    }
    if (sim_allreduce_fails(comm->rank)) return ncclUnhandledCudaError; // injected: this rank's share is never queued
    sim_enqueue(stream, [=] {
        SimGroup* g = comm->group;
        std::unique_lock<std::mutex> lk(g->m);
        const unsigned long long gen = g->generation;
        g->send[(size_t)comm->rank] = static_cast<const float*>(send);
        if (++g->arrived == g->n) {
            std::vector<float>& s = g->sum[gen & 1u];
            s.assign(count, 0.0f);
            for (int r = 0; r < g->n; r++) { // rank order: every rank receives the same bits
                const float* src = g->send[(size_t)r];
                if (r == 0) memcpy(s.data(), src, count * sizeof(float));
                else for (size_t k = 0; k < count; k++) s[k] += src[k];
            }
            g->arrived = 0;
            g->generation++;
            g->cv.notify_all();
        } else {
            while (g->generation == gen && !comm->aborted.load()) g->cv.wait_for(lk, std::chrono::milliseconds(20));
            if (g->generation == gen) { // aborted: this rank's share is withdrawn, nothing is delivered
                g->arrived--;
                return;
            }
        }
        // the sum of collective `gen` stays valid until collective gen + 2 completes, which needs this rank's share of gen + 1
        const std::vector<float>& s = g->sum[gen & 1u];
        lk.unlock();
        memcpy(recv, s.data(), count * sizeof(float));
    });
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "ncclSuccess" : "simulated RCCL error"; }

} // extern "C"
