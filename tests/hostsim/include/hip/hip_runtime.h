// hip/hip_runtime.h of the HOST SIMULATION (tests/hostsim): the handful of runtime calls csrc/s2d_multi.hip makes, backed by
// threads instead of a GPU, so that its host protocol -- worker threads, the breakable barrier, pairwise sequence counters,
// stream event waits, two send buffers -- can run under ThreadSanitizer on a machine without a GPU.
//
// TEST INFRASTRUCTURE ONLY.  Not a HIP implementation and not a CPU fallback of the product: only tests/hostsim builds
// against it (g++ -x c++ -fsanitize=thread -I tests/hostsim/include ... csrc/s2d_multi.hip).
//
// Semantics that matter to the protocol under test, and that the simulation keeps:
//  * a stream is an in-order queue run by its own thread; every *Async call only enqueues (copies too -- stricter than HIP,
//    which stages pageable host memory synchronously);
//  * hipEventRecord takes effect when the stream reaches it; hipStreamWaitEvent waits for the record that was the latest
//    AT THE TIME OF THE CALL (an event re-recorded later does not move a wait already queued); hipEventQuery answers for
//    the latest record;
//  * device memory is host memory; a "peer copy" is a memcpy run by the destination's stream thread.
#pragma once

#include <cstddef>
#include <cstdint>

typedef enum hipError_t {
    hipSuccess = 0,
    hipErrorInvalidValue = 1,
    hipErrorOutOfMemory = 2,
    hipErrorNotReady = 600,
    hipErrorPeerAccessAlreadyEnabled = 704,
} hipError_t;

typedef enum hipMemcpyKind {
    hipMemcpyHostToHost = 0,
    hipMemcpyHostToDevice = 1,
    hipMemcpyDeviceToHost = 2,
    hipMemcpyDeviceToDevice = 3,
    hipMemcpyDefault = 4,
} hipMemcpyKind;

struct SimStream;
struct SimEvent;
typedef SimStream* hipStream_t;
typedef SimEvent* hipEvent_t;

#define hipEventDisableTiming 0x2
#define hipHostMallocDefault 0x0

struct hipDeviceProp_t {
    char name[256];
};

extern "C" {
const char* hipGetErrorString(hipError_t e);
hipError_t hipGetLastError(void);
hipError_t hipSetDevice(int device);
hipError_t hipGetDevice(int* device);
hipError_t hipGetDeviceProperties(hipDeviceProp_t* prop, int device);
hipError_t hipDeviceGetPCIBusId(char* out, int len, int device);
hipError_t hipDeviceCanAccessPeer(int* can, int device, int peer);
hipError_t hipDeviceEnablePeerAccess(int peer, unsigned flags);
hipError_t hipMalloc(void** p, size_t bytes);
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned flags);
hipError_t hipHostFree(void* p);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t stream);
hipError_t hipMemcpyPeerAsync(void* dst, int dst_device, const void* src, int src_device, size_t bytes, hipStream_t stream);
hipError_t hipStreamSynchronize(hipStream_t stream);
hipError_t hipEventCreate(hipEvent_t* ev);
hipError_t hipEventCreateWithFlags(hipEvent_t* ev, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t ev);
hipError_t hipEventRecord(hipEvent_t ev, hipStream_t stream);
hipError_t hipEventQuery(hipEvent_t ev);
hipError_t hipStreamWaitEvent(hipStream_t stream, hipEvent_t ev, unsigned flags);
}

// ---- the simulation's own interface (sim_hip.cpp), for the simulated contexts and collectives --------------------------
#include <functional>
hipStream_t sim_stream_create(int device);
void sim_stream_destroy(hipStream_t s);
void sim_enqueue(hipStream_t s, std::function<void()> op); // runs on the stream's thread, in order
int sim_device_count();
void sim_set_device_count(int n);
// failure injection (multi_sim_main.cpp): s2d_forward_backward of the context on `device` fails at `iteration` (< 0: never);
// the `call`-th ncclAllReduce submitted for rank `rank` returns an error (< 0: never)
void sim_fail_forward_backward(int device, int iteration);
void sim_fail_allreduce(int rank, long long call);
// s2d_forward_backward of the context on `device` does not RETURN at `iteration` (a runtime call that hangs) until
// sim_release_blocked() is called
void sim_block_forward_backward(int device, int iteration);
void sim_release_blocked();
void sim_maybe_block(int device, int iteration);
int sim_forward_backward_fails(int device, int iteration);
int sim_allreduce_fails(int rank);
// counters for the test's own assertions
struct SimCounters {
    unsigned long long ops, peer_copies, event_waits, event_records;
};
SimCounters sim_counters();
