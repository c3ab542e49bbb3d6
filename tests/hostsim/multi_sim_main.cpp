// multi_sim_main.cpp -- scenarios that drive csrc/s2d_multi.hip (compiled as host C++ against tests/hostsim) under
// ThreadSanitizer: 2 / 4 / 8 rank threads, slab ownership and replicated state side by side over more than 200 iterations
// with hold-set refreshes, a get_splats / set_splats in the middle, ranks slowed down at random, an injected rank that
// stops answering (both schemes), and a non-finite stop.  Prints one line per scenario; exit code 0 = all good.
// TEST INFRASTRUCTURE ONLY.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/splat2d.h"
#include "../../include/splat2d_test.h"

static int g_failures = 0;
#define EXPECT(cond, ...)                                  \
    do {                                                   \
        if (!(cond)) {                                     \
            fprintf(stderr, "FAILED %s:%d: ", __FILE__, __LINE__); \
            fprintf(stderr, __VA_ARGS__);                  \
            fprintf(stderr, "\n");                         \
            g_failures++;                                  \
        }                                                  \
    } while (0)

struct Run {
    std::vector<double> trace;
    std::vector<s2d_splat> splats;
    std::vector<s2d_splat_adam> adams;
    int64_t info[4] = {0, 0, 0, 0};
};

static s2d_config config(int W, int H, int n)
{
    s2d_config c;
    memset(&c, 0, sizeof(c));
    c.struct_size = sizeof(c);
    c.width = W;
    c.height = H;
    c.n_splats = n;
    return c;
}

static s2d_multi* make(int world, int W, int H, int n, uint32_t flags)
{
    const s2d_config cfg = config(W, H, n);
    std::vector<int32_t> dev((size_t)world);
    for (int r = 0; r < world; r++) dev[(size_t)r] = (flags & S2D_MULTI_SHARE_GPU) ? 0 : r;
    s2d_multi* m = nullptr;
    const int rc = s2d_multi_create(&cfg, dev.data(), world, flags, &m);
    if (rc != S2D_OK) {
        fprintf(stderr, "s2d_multi_create(%d ranks, flags %u): %d %s\n", world, flags, rc, m ? s2d_multi_last_error(m) : "");
        exit(2);
    }
    return m;
}

// 230 iterations in uneven calls (refreshes at 64, 128, 192 fall inside calls and on their edges), the state read back and
// written again in the middle; with `jitter` a different rank's thread is slowed down in every call.
static Run train(int world, int W, int H, int n, uint32_t flags, bool jitter)
{
    Run out;
    s2d_multi* m = make(world, W, H, n, flags);
    EXPECT(s2d_multi_set_target_synthetic(m) == S2D_OK, "set_target_synthetic");
    EXPECT(s2d_multi_init_splats(m) == S2D_OK, "init");
    const int calls[] = {50, 14, 1, 35, 30, 70, 30};
    int it = 0, call = 0;
    for (const int k : calls) {
        if (jitter) (void)s2d_test_multi_stall(m, (call * 5 + 1) % world, it + k / 2, 2 + call % 3);
        std::vector<double> mse((size_t)k);
        const int rc = s2d_multi_step(m, k, 0, mse.data());
        EXPECT(rc == S2D_OK, "s2d_multi_step(%d) at iteration %d, %d ranks: %d %s", k, it, world, rc, s2d_multi_last_error(m));
        if (rc != S2D_OK) break;
        out.trace.insert(out.trace.end(), mse.begin(), mse.end());
        it += k;
        call++;
        if (it == 100) { // the gathered state goes back in: the run must go on as if nothing had happened
            std::vector<s2d_splat> sp((size_t)n);
            EXPECT(s2d_multi_get_splats(m, sp.data()) == S2D_OK, "get_splats");
            EXPECT(s2d_multi_set_splats(m, sp.data()) == S2D_OK, "set_splats");
        }
    }
    out.splats.resize((size_t)n);
    out.adams.resize((size_t)n);
    float b1, b2;
    int32_t its = 0;
    EXPECT(s2d_multi_get_splats(m, out.splats.data()) == S2D_OK, "get_splats");
    EXPECT(s2d_multi_get_adam(m, out.adams.data(), &b1, &b2, &its) == S2D_OK && its == it, "get_adam: %d iterations, expected %d", its, it);
    EXPECT(s2d_multi_exchange_info(m, out.info) == S2D_OK, "exchange_info");
    std::vector<float> img((size_t)W * H * 4);
    EXPECT(s2d_multi_forward(m) == S2D_OK && s2d_multi_get_image(m, img.data()) == S2D_OK, "forward / get_image");
    s2d_multi_destroy(m);
    return out;
}

static void compare_schemes(int world, uint32_t extra = 0)
{
    const int W = 64, H = 32 * world, n = 60 * world;
    const int before = g_failures;
    const Run own = train(world, W, H, n, extra, true);
    const Run rep = train(world, W, H, n, extra | S2D_MULTI_REPLICATED, false);
    EXPECT(own.trace.size() == 230 && rep.trace.size() == 230, "trace lengths %zu %zu", own.trace.size(), rep.trace.size());
    EXPECT(own.info[0] == 1 && rep.info[0] == 2, "schemes %lld %lld", (long long)own.info[0], (long long)rep.info[0]);
    // (two slabs of 32 rows: every splat is within reach + margin of both -- from four ranks on, no rank holds everything)
    EXPECT(own.info[1] > 0 && (world == 2 || own.info[3] < (int64_t)n * world), "ownership shares rows (%lld) and holds less than everything (%lld)",
           (long long)own.info[1], (long long)own.info[3]);
    // both schemes add the ranks' partial gradients in ascending rank order: the same additions, the same bits
    size_t bad = 0;
    for (size_t k = 0; k < own.trace.size() && k < rep.trace.size(); k++) bad += own.trace[k] != rep.trace[k];
    EXPECT(bad == 0, "%zu of %zu MSE values differ between slab ownership and replicated state (%d ranks)", bad, own.trace.size(), world);
    EXPECT(memcmp(own.splats.data(), rep.splats.data(), own.splats.size() * sizeof(s2d_splat)) == 0, "parameters differ (%d ranks)", world);
    const float* a = reinterpret_cast<const float*>(own.adams.data());
    const float* b = reinterpret_cast<const float*>(rep.adams.data());
    bad = 0;
    for (size_t k = 0; k < own.adams.size() * 18; k++) bad += !(a[k] == b[k]);
    EXPECT(bad == 0, "%zu Adam moments differ (%d ranks)", bad, world);
    EXPECT(std::isfinite(own.trace.back()) && own.trace.back() < own.trace.front(), "the run trains: %g -> %g", own.trace.front(), own.trace.back());
    printf("%s: %d ranks%s, 230 iterations, ownership == replicated bit for bit (mse %.4f -> %.4f, %lld rows swapped per iteration, %lld hand-overs)\n",
           g_failures == before ? "ok" : "NOT ok", world, (extra & S2D_MULTI_SHARE_GPU) ? " sharing one device (device copies, host-staged sums)" : "",
           own.trace.front(), own.trace.back(), (long long)own.info[1], (long long)own.info[2]);
    if (world == 2 && !extra) { // ... and both follow one context on the whole image (only the order of the gradient sums differs)
        const Run one = train(1, W, H, n, 0, false);
        double worst = 0.0;
        for (size_t k = 0; k < one.trace.size() && k < own.trace.size(); k++)
            worst = std::fmax(worst, std::fabs(one.trace[k] - own.trace[k]) / one.trace[k]);
        EXPECT(worst <= 1e-3, "two ranks against one context: trace differs by %g", worst);
        printf("%s: 2 ranks follow one context's trace to %.1e\n", worst <= 1e-3 ? "ok" : "NOT ok", worst);
    }
}

// A rank's thread stops answering in the middle of a call: the call must come back with S2D_E_STATE inside a few stall
// limits, say which rank and where, refuse further work, and the handle must destroy.
static void stalled_rank(int world, uint32_t flags, int stall_rank, int at_iteration, int iters)
{
    const int W = 64, H = 32 * world, n = 40 * world;
    const int before = g_failures;
    char want[64];
    s2d_multi* m = make(world, W, H, n, flags);
    EXPECT(s2d_multi_set_target_synthetic(m) == S2D_OK && s2d_multi_init_splats(m) == S2D_OK, "set up");
    EXPECT(s2d_multi_set_stall_timeout(m, 300) == S2D_OK, "set_stall_timeout");
    EXPECT(s2d_test_multi_stall(m, stall_rank, at_iteration, -1) == S2D_OK, "stall hook");
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = s2d_multi_step(m, iters, 0, nullptr);
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const std::string msg = s2d_multi_last_error(m);
    EXPECT(rc == S2D_E_STATE, "a stalled rank gives S2D_E_STATE, got %d (%s)", rc, msg.c_str());
    const int dev = (flags & S2D_MULTI_SHARE_GPU) ? 0 : stall_rank;
    if ((flags & S2D_MULTI_REPLICATED) && (flags & S2D_MULTI_SHARE_GPU)) snprintf(want, sizeof(want), "rank %d (device %d) did not reach the rendezvous", stall_rank, dev);
    else if (flags & S2D_MULTI_REPLICATED) snprintf(want, sizeof(want), "furthest behind: rank %d", stall_rank);
    else snprintf(want, sizeof(want), "rank %d (device %d) stopped answering", stall_rank, dev);
    EXPECT(msg.find(want) != std::string::npos, "the report names the rank: %s", msg.c_str());
    EXPECT(secs < 20.0, "the call came back after %.1f s", secs);
    EXPECT(s2d_multi_step(m, 1, 0, nullptr) == S2D_E_STATE, "a handle whose ranks disagree refuses further steps");
    s2d_multi_destroy(m);
    printf("%s: %d ranks (%s), rank %d stops answering at iteration %d -> S2D_E_STATE after %.2f s: %s\n",
           g_failures == before ? "ok" : "NOT ok", world, (flags & S2D_MULTI_REPLICATED) ? "replicated, RCCL" : "ownership", stall_rank, at_iteration, secs, msg.c_str());
}

// One rank's launch fails (ownership and RCCL) or its all-reduce submission does: the call must return THAT rank's error --
// not a timeout of the others --, quickly; with RCCL the others' communicators are aborted (they sit in collectives the failed
// rank never joins) and the handle is dead; a slab-ownership handle stays usable after a restart.
static void failing_rank(int world, uint32_t flags, int bad_rank, int at_iteration, bool in_allreduce)
{
    const int W = 64, H = 32 * world, n = 40 * world;
    const int before = g_failures;
    s2d_multi* m = make(world, W, H, n, flags);
    EXPECT(s2d_multi_set_target_synthetic(m) == S2D_OK && s2d_multi_init_splats(m) == S2D_OK, "set up");
    EXPECT(s2d_multi_set_stall_timeout(m, 3000) == S2D_OK, "set_stall_timeout");
    if (in_allreduce) sim_fail_allreduce(bad_rank, at_iteration);
    else sim_fail_forward_backward(bad_rank, at_iteration);
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = s2d_multi_step(m, 150, 0, nullptr);
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const std::string msg = s2d_multi_last_error(m);
    sim_fail_allreduce(-1, -1);
    sim_fail_forward_backward(-1, -1);
    char want[32];
    snprintf(want, sizeof(want), "on rank %d", bad_rank);
    EXPECT(rc == S2D_E_HIP && msg.find(want) != std::string::npos, "the failing rank's own error is reported: %d %s", rc, msg.c_str());
    EXPECT(msg.find(in_allreduce ? "ncclAllReduce failed" : "simulated device fault") != std::string::npos, "... with its cause: %s", msg.c_str());
    EXPECT(secs < 2.5, "without waiting for a stall limit: %.2f s", secs);
    const int again = s2d_multi_step(m, 1, 0, nullptr); // refused either way: the ranks stand at different iterations
    EXPECT(again == S2D_E_STATE, "a step right after a failed one is refused: %d %s", again, s2d_multi_last_error(m));
    if (flags & S2D_MULTI_REPLICATED) {
        EXPECT(std::string(s2d_multi_last_error(m)).find("create a new one") != std::string::npos, "RCCL: the communicators are gone: %s",
               s2d_multi_last_error(m));
    } else {
        EXPECT(s2d_multi_init_splats(m) == S2D_OK && s2d_multi_step(m, 70, 0, nullptr) == S2D_OK, "ownership: usable after a restart: %s",
               s2d_multi_last_error(m));
    }
    s2d_multi_destroy(m);
    printf("%s: %d ranks (%s), rank %d fails in its %s at iteration %d -> %s after %.2f s\n", g_failures == before ? "ok" : "NOT ok", world,
           (flags & S2D_MULTI_REPLICATED) ? "replicated, RCCL" : "ownership", bad_rank, in_allreduce ? "all-reduce submission" : "raster launch",
           at_iteration, msg.c_str(), secs);
}

// A rank sits inside a runtime call that never returns: it cannot answer the stop.  The caller's watchdog (run_command) must
// notice that nobody moves, stop the others, give up on the one that does not come back, and return; the handle is then
// abandoned (destroy returns at once, nothing of it is freed under the sleeping thread).
static void rank_inside_a_call_that_never_returns(int world, uint32_t flags, int bad_rank, int at_iteration)
{
    const int W = 64, H = 32 * world, n = 40 * world;
    const int before = g_failures;
    s2d_multi* m = make(world, W, H, n, flags);
    EXPECT(s2d_multi_set_target_synthetic(m) == S2D_OK && s2d_multi_init_splats(m) == S2D_OK, "set up");
    EXPECT(s2d_multi_set_stall_timeout(m, 500) == S2D_OK, "set_stall_timeout");
    sim_block_forward_backward(bad_rank, at_iteration);
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = s2d_multi_step(m, 12, 0, nullptr);
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const std::string msg = s2d_multi_last_error(m);
    char want[48];
    snprintf(want, sizeof(want), "rank %d (device %d) at 'raster launch'", bad_rank, bad_rank);
    EXPECT(rc == S2D_E_STATE && msg.find("abandoned") != std::string::npos && msg.find(want) != std::string::npos, "%d %s", rc, msg.c_str());
    EXPECT(secs < 15.0, "the call came back after %.1f s", secs);
    EXPECT(s2d_multi_step(m, 1, 0, nullptr) == S2D_E_STATE, "an abandoned handle refuses");
    const auto t1 = std::chrono::steady_clock::now();
    s2d_multi_destroy(m);
    EXPECT(std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count() < 1.0, "destroy of an abandoned handle returns at once");
    sim_release_blocked(); // let the sleeping thread run out (it finds the stop and leaves; the abandoned handle is still there)
    std::this_thread::sleep_for(std::chrono::milliseconds(300));
    printf("%s: %d ranks (%s), rank %d inside a call that never returns -> abandoned after %.2f s: %s\n", g_failures == before ? "ok" : "NOT ok", world,
           (flags & S2D_MULTI_REPLICATED) ? "replicated, RCCL" : "ownership", bad_rank, secs, msg.c_str());
}

static void nonfinite(int world, uint32_t flags)
{
    const int W = 64, H = 32 * world, n = 40 * world;
    const int before = g_failures;
    s2d_multi* m = make(world, W, H, n, flags);
    EXPECT(s2d_multi_set_target_synthetic(m) == S2D_OK && s2d_multi_init_splats(m) == S2D_OK, "set up");
    EXPECT(s2d_multi_step(m, 3, 0, nullptr) == S2D_OK, "three iterations: %s", s2d_multi_last_error(m));
    std::vector<s2d_splat_adam> ad((size_t)n);
    float b1, b2;
    int32_t it;
    EXPECT(s2d_multi_get_adam(m, ad.data(), &b1, &b2, &it) == S2D_OK, "get_adam");
    ad[5].rot.m = INFINITY;
    EXPECT(s2d_multi_set_adam(m, ad.data(), b1, b2, it) == S2D_OK, "set_adam");
    const int rc = s2d_multi_step(m, 70, 0, nullptr); // runs into a refresh on the way
    EXPECT(rc == S2D_E_NONFINITE, "a non-finite parameter gives S2D_E_NONFINITE, got %d (%s)", rc, s2d_multi_last_error(m));
    EXPECT(s2d_multi_init_splats(m) == S2D_OK && s2d_multi_step(m, 5, 0, nullptr) == S2D_OK, "usable again after init: %s", s2d_multi_last_error(m));
    s2d_multi_destroy(m);
    printf("%s: %d ranks (%s), non-finite stop reported and survived\n", g_failures == before ? "ok" : "NOT ok", world, (flags & S2D_MULTI_REPLICATED) ? "replicated" : "ownership");
}

int main(int argc, char** argv)
{
    sim_set_device_count(8);
    const bool quick = argc > 1 && std::string(argv[1]) == "quick";
    for (const int world : {2, 4, 8}) {
        if (quick && world != 4) continue;
        compare_schemes(world);
    }
    compare_schemes(3, S2D_MULTI_SHARE_GPU);          // the rehearsal mode of one-GPU boxes: device copies, host-staged sums
    stalled_rank(4, 0, 2, 70, 100);                    // ownership: the neighbours wait for its exchange
    stalled_rank(2, 0, 1, 3, 10);
    stalled_rank(3, S2D_MULTI_REPLICATED, 1, 5, 20);   // RCCL: the others sit in an all-reduce it never joins
    stalled_rank(8, S2D_MULTI_REPLICATED, 7, 130, 200);
    stalled_rank(3, S2D_MULTI_REPLICATED | S2D_MULTI_SHARE_GPU, 2, 4, 12); // host-staged sums: the others wait at its rendezvous
    rank_inside_a_call_that_never_returns(3, 0, 1, 5);
    rank_inside_a_call_that_never_returns(4, S2D_MULTI_REPLICATED, 3, 6);
    failing_rank(4, 0, 1, 37, false);
    failing_rank(4, S2D_MULTI_REPLICATED, 2, 9, false);
    failing_rank(3, S2D_MULTI_REPLICATED, 0, 70, true);
    nonfinite(4, 0);
    nonfinite(4, S2D_MULTI_REPLICATED);
    const SimCounters c = sim_counters();
    printf("simulated runtime: %llu stream operations, %llu peer copies, %llu event records, %llu event waits\n", c.ops, c.peer_copies,
           c.event_records, c.event_waits);
    if (g_failures) fprintf(stderr, "%d check(s) failed\n", g_failures);
    return g_failures ? 1 : 0;
}
