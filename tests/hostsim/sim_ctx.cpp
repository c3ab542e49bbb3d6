// sim_ctx.cpp -- the single-device context (include/splat2d.h s2d_*) as csrc/s2d_multi.hip uses it, simulated on the host:
// every call queues an operation on the context's simulated stream, and the operations compute with the CPU oracle
// (oracle/s2d_oracle.c) -- forward / backward over the rank's rows for the splats it holds, the Adam step on them, the row
// gathers / scatters / the rank-ordered combine of slab ownership.  TEST INFRASTRUCTURE ONLY: this is how the multi-device
// HOST PROTOCOL runs under ThreadSanitizer without a GPU; it is not a CPU path of the product (nothing under
// 2dgaussiansplatting_amd/ builds or loads it).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/splat2d.h"
#include "../../oracle/s2d_oracle.h"

struct s2d_ctx {
    int W = 0, H = 0, n = 0, device = 0, r0 = 0, r1 = 0;
    float lr = 0.05f;
    hipStream_t stream = nullptr;
    // "device" state: touched by the stream's operations only (and by host calls after a synchronise)
    std::vector<s2do_splat> splats;
    std::vector<s2do_splat_adam> adams;
    std::vector<float> grads;   // n x 9
    std::vector<uint8_t> held;  // empty: every splat
    std::vector<float> image0, ref, image1; // W x H x 4 (only the rank's rows are used)
    std::vector<double> trace;  // squared error per iteration % size
    std::atomic<int> nonfinite{0};
    std::atomic<int> first_nonfinite{INT_MAX};
    // host-side state of main() (touched by the calling thread only)
    float beta1t = 1.0f, beta2t = 1.0f;
    int iterations = 0;
    bool have_target = false;
    char err[256] = {0};
};

namespace {

void sync(s2d_ctx* c) { (void)hipStreamSynchronize(c->stream); }

std::vector<int> held_ids(const s2d_ctx* c)
{
    std::vector<int> ids;
    ids.reserve((size_t)c->n);
    for (int i = 0; i < c->n; i++)
        if (c->held.empty() || c->held[(size_t)i]) ids.push_back(i);
    return ids;
}

// forward (+ backward) of the rank's rows over the splats it holds, in index order (the reference's blend order)
void raster(s2d_ctx* c, int iteration, bool backward)
{
    if (c->first_nonfinite.load() < iteration) return; // the reference abort()ed earlier: every later kernel does nothing
    const std::vector<int> ids = held_ids(c);
    const int m = (int)ids.size();
    std::vector<s2do_splat> cs((size_t)std::max(m, 1));
    for (int j = 0; j < m; j++) cs[(size_t)j] = c->splats[(size_t)ids[(size_t)j]];
    s2do_forward_rows(cs.data(), m, c->W, c->H, c->r0, c->r1, c->image0.data(), nullptr);
    if (!backward) return;
    std::vector<s2do_splat> cg((size_t)std::max(m, 1));
    memset(cg.data(), 0, cg.size() * sizeof(s2do_splat));
    s2do_backward_rows(cs.data(), m, c->W, c->H, c->r0, c->r1, c->image0.data(), c->ref.data(), c->image1.data(), cg.data(), nullptr);
    for (int j = 0; j < m; j++) {
        const float* g = reinterpret_cast<const float*>(&cg[(size_t)j]);
        float* dst = c->grads.data() + (size_t)ids[(size_t)j] * 9;
        for (int k = 0; k < 9; k++) dst[k] += g[k];
    }
    c->trace[(size_t)iteration % c->trace.size()] = s2do_sqerr_rows(c->image0.data(), c->ref.data(), c->W, c->H, c->r0, c->r1);
}

void adam(s2d_ctx* c, int iteration, float b1, float b2, int optimize_opacity)
{
    if (c->first_nonfinite.load() < iteration) return;
    const std::vector<int> ids = held_ids(c);
    const int m = (int)ids.size();
    std::vector<s2do_splat> cs((size_t)std::max(m, 1)), cg((size_t)std::max(m, 1));
    std::vector<s2do_splat_adam> ca((size_t)std::max(m, 1));
    for (int j = 0; j < m; j++) {
        const size_t i = (size_t)ids[(size_t)j];
        cs[(size_t)j] = c->splats[i];
        ca[(size_t)j] = c->adams[i];
        memcpy(&cg[(size_t)j], c->grads.data() + i * 9, sizeof(s2do_splat));
    }
    const int bad = s2do_adam_step(cs.data(), ca.data(), cg.data(), m, c->W, c->H, &b1, &b2, optimize_opacity, c->lr);
    for (int j = 0; j < m; j++) {
        const size_t i = (size_t)ids[(size_t)j];
        c->splats[i] = cs[(size_t)j];
        c->adams[i] = ca[(size_t)j];
        memset(c->grads.data() + i * 9, 0, 9 * sizeof(float)); // re-zeroed like main.cpp:550
    }
    if (bad) {
        c->nonfinite.store(1);
        int expect = INT_MAX;
        c->first_nonfinite.compare_exchange_strong(expect, iteration);
    }
}

} // namespace

extern "C" {

int s2d_abi_version(void) { return S2D_ABI_VERSION; }

int s2d_create(const s2d_config* cfg, s2d_ctx** out)
{
    if (!cfg || !out || cfg->struct_size != sizeof(s2d_config) || cfg->width <= 0 || cfg->height <= 0 || cfg->n_splats < 0) return S2D_E_INVALID;
    if (cfg->device < 0 || cfg->device >= sim_device_count()) return S2D_E_HIP;
    s2d_ctx* c = new s2d_ctx();
    *out = c;
    c->W = cfg->width;
    c->H = cfg->height;
    c->n = cfg->n_splats;
    c->device = cfg->device;
    c->r0 = cfg->row_begin;
    c->r1 = cfg->row_end ? cfg->row_end : cfg->height;
    if (cfg->training_rate > 0.0f) c->lr = cfg->training_rate;
    c->stream = sim_stream_create(c->device);
    c->splats.resize((size_t)c->n);
    c->adams.resize((size_t)c->n);
    memset(c->splats.data(), 0, c->splats.size() * sizeof(s2do_splat));
    memset(c->adams.data(), 0, c->adams.size() * sizeof(s2do_splat_adam));
    c->grads.assign((size_t)c->n * 9, 0.0f);
    const size_t px = (size_t)c->W * c->H * 4;
    c->image0.assign(px, 0.0f);
    c->ref.assign(px, 0.0f);
    c->image1.assign(px, 0.0f);
    c->trace.assign(4096, 0.0);
    return S2D_OK;
}

void s2d_destroy(s2d_ctx* c)
{
    if (!c) return;
    sim_stream_destroy(c->stream);
    delete c;
}

const char* s2d_last_error(const s2d_ctx* c) { return c ? c->err : "null context"; }
void* s2d_stream(s2d_ctx* c) { return c->stream; }
void* s2d_grads_device_ptr(s2d_ctx* c) { return c->grads.data(); }

int s2d_set_target(s2d_ctx* c, const float* rgba)
{
    sync(c);
    memcpy(c->ref.data(), rgba, c->ref.size() * sizeof(float));
    c->have_target = true;
    return S2D_OK;
}

int s2d_set_target_synthetic(s2d_ctx* c)
{
    sync(c);
    for (int y = 0; y < c->H; y++)
        for (int x = 0; x < c->W; x++) {
            float* p = c->ref.data() + 4 * ((size_t)y * c->W + x);
            p[0] = (float)x / (float)c->W;
            p[1] = 1.0f - (float)x / (float)c->W;
            p[2] = (float)y / (float)c->H;
            p[3] = 1.0f;
        }
    c->have_target = true;
    return S2D_OK;
}

int s2d_init_splats(s2d_ctx* c)
{
    sync(c);
    s2do_init(c->splats.data(), c->adams.data(), c->n, c->W, c->H);
    std::fill(c->grads.begin(), c->grads.end(), 0.0f);
    c->beta1t = c->beta2t = 1.0f;
    c->iterations = 0;
    c->nonfinite.store(0);
    c->first_nonfinite.store(INT_MAX);
    return S2D_OK;
}

int s2d_set_splats(s2d_ctx* c, const s2d_splat* s)
{
    sync(c);
    memcpy(c->splats.data(), s, (size_t)c->n * sizeof(s2d_splat));
    c->nonfinite.store(0);
    c->first_nonfinite.store(INT_MAX);
    return S2D_OK;
}

int s2d_get_splats(s2d_ctx* c, s2d_splat* s)
{
    sync(c);
    memcpy(s, c->splats.data(), (size_t)c->n * sizeof(s2d_splat));
    return S2D_OK;
}

int s2d_set_adam(s2d_ctx* c, const s2d_splat_adam* a, float b1, float b2, int32_t it)
{
    sync(c);
    memcpy(c->adams.data(), a, (size_t)c->n * sizeof(s2d_splat_adam));
    c->beta1t = b1;
    c->beta2t = b2;
    c->iterations = it;
    return S2D_OK;
}

int s2d_get_adam(s2d_ctx* c, s2d_splat_adam* a, float* b1, float* b2, int32_t* it)
{
    sync(c);
    if (a) memcpy(a, c->adams.data(), (size_t)c->n * sizeof(s2d_splat_adam));
    if (b1) *b1 = c->beta1t;
    if (b2) *b2 = c->beta2t;
    if (it) *it = c->iterations;
    return S2D_OK;
}

int s2d_forward(s2d_ctx* c)
{
    const int it = c->iterations;
    sim_enqueue(c->stream, [c, it] { raster(c, it, false); });
    return S2D_OK;
}

int s2d_forward_backward(s2d_ctx* c, uint32_t)
{
    if (!c->have_target) {
        snprintf(c->err, sizeof(c->err), "no target image set (s2d_set_target)");
        return S2D_E_STATE;
    }
    const int it = c->iterations;
    sim_maybe_block(c->device, it); // injected: a runtime call that does not return
    if (sim_forward_backward_fails(c->device, it)) { // injected: a launch that the runtime refuses on this device only
        snprintf(c->err, sizeof(c->err), "hipLaunchKernel failed: simulated device fault (device %d, iteration %d)", c->device, it);
        return S2D_E_HIP;
    }
    sim_enqueue(c->stream, [c, it] { raster(c, it, true); });
    return S2D_OK;
}

int s2d_adam_step(s2d_ctx* c, uint32_t flags)
{
    const int it = c->iterations;
    const float b1 = c->beta1t, b2 = c->beta2t; // s2do_adam_step multiplies them first, main.cpp:718-719
    c->beta1t *= 0.9f;
    c->beta2t *= 0.99f;
    c->iterations++;
    const int op = (flags & S2D_STEP_OPTIMIZE_OPACITY) ? 1 : 0;
    sim_enqueue(c->stream, [c, it, b1, b2, op] { adam(c, it, b1, b2, op); });
    return S2D_OK;
}

int s2d_synchronize(s2d_ctx* c)
{
    sync(c);
    if (c->nonfinite.load()) {
        const int k = c->first_nonfinite.load();
        if (k + 1 < c->iterations) { // wind the host-side counters back to the failing step, like the product
            float b1 = 1.0f, b2 = 1.0f;
            for (int i = 0; i <= k; i++) { b1 *= 0.9f; b2 *= 0.99f; }
            c->beta1t = b1;
            c->beta2t = b2;
            c->iterations = k + 1;
        }
        snprintf(c->err, sizeof(c->err), "non-finite parameter after iteration %d (the reference abort()s, main.cpp:752-785)", k);
        return S2D_E_NONFINITE;
    }
    return S2D_OK;
}

int s2d_get_sqerr_trace(s2d_ctx* c, int32_t first, int32_t count, double* out)
{
    sync(c);
    for (int k = 0; k < count; k++) out[k] = c->trace[(size_t)(first + k) % c->trace.size()];
    return S2D_OK;
}

int s2d_get_image_rows(s2d_ctx* c, float* rows)
{
    sync(c);
    memcpy(rows, c->image0.data() + (size_t)c->r0 * c->W * 4, (size_t)(c->r1 - c->r0) * c->W * 4 * sizeof(float));
    return S2D_OK;
}

// ---- slab ownership: what csrc/s2d_halo.hip does on the device ----
int s2d_halo_masks(s2d_ctx* c, int32_t world, const int32_t* bounds, float margin, uint32_t* masks)
{
    std::vector<int32_t> b(bounds, bounds + world + 1);
    sim_enqueue(c->stream, [c, world, b, margin, masks] {
        for (int i = 0; i < c->n; i++) {
            if (!c->held.empty() && !c->held[(size_t)i]) {
                masks[i] = 0u;
                continue;
            }
            const s2do_splat& s = c->splats[(size_t)i];
            const float y = s.pos_y, reach = 3.0f * std::fmax(s.sx, s.sy) + 2.0f + margin;
            uint32_t m = 0u;
            for (int q = 0; q < world; q++)
                if (y + reach >= (float)b[(size_t)q] && y - reach <= (float)b[(size_t)q + 1]) m |= 1u << q;
            if (m == 0u) {
                int q = 0;
                while (q + 1 < world && !(y < (float)b[(size_t)q + 1])) q++;
                m = 1u << q;
            }
            masks[i] = m;
        }
    });
    return S2D_OK;
}

int s2d_halo_commit(s2d_ctx* c, const uint32_t* masks, int32_t rank, int32_t)
{
    sim_enqueue(c->stream, [c, masks, rank] {
        if (!masks) {
            c->held.clear();
            return;
        }
        c->held.resize((size_t)c->n);
        for (int i = 0; i < c->n; i++) c->held[(size_t)i] = (uint8_t)((masks[i] >> rank) & 1u);
    });
    return S2D_OK;
}

static float* rows_base(s2d_ctx* c, int what, int* w)
{
    *w = what == S2D_ROWS_ADAM ? 18 : 9;
    if (what == S2D_ROWS_GRADS) return c->grads.data();
    if (what == S2D_ROWS_SPLATS) return reinterpret_cast<float*>(c->splats.data());
    return reinterpret_cast<float*>(c->adams.data());
}

int s2d_rows_gather(s2d_ctx* c, int32_t what, const int32_t* ids, int32_t count, float* out)
{
    sim_enqueue(c->stream, [c, what, ids, count, out] {
        int w;
        const float* base = rows_base(c, what, &w);
        for (int j = 0; j < count; j++) {
            const int i = ids[j];
            for (int k = 0; k < w; k++) out[(size_t)j * w + k] = (i >= 0 && i < c->n) ? base[(size_t)i * w + k] : 0.0f;
        }
    });
    return S2D_OK;
}

int s2d_rows_scatter(s2d_ctx* c, int32_t what, const int32_t* ids, int32_t count, const float* in)
{
    sim_enqueue(c->stream, [c, what, ids, count, in] {
        int w;
        float* base = rows_base(c, what, &w);
        for (int j = 0; j < count; j++) {
            const int i = ids[j];
            if (i < 0 || i >= c->n) continue;
            for (int k = 0; k < w; k++) base[(size_t)i * w + k] = in[(size_t)j * w + k];
        }
    });
    return S2D_OK;
}

int s2d_grads_combine(s2d_ctx* c, const int32_t* rows, int32_t n_rows, const int32_t* src, int32_t world, const float* recv)
{
    sim_enqueue(c->stream, [c, rows, n_rows, src, world, recv] {
        for (int u = 0; u < n_rows; u++) {
            float* g = c->grads.data() + (size_t)rows[u] * 9;
            float acc[9];
            bool first = true;
            for (int q = 0; q < world; q++) { // ascending rank order: every holder forms the same bits
                const int sidx = src[(size_t)u * world + q];
                if (sidx == -1) continue;
                const float* p = sidx == -2 ? g : recv + (size_t)sidx * 9;
                for (int k = 0; k < 9; k++) acc[k] = first ? p[k] : acc[k] + p[k];
                first = false;
            }
            for (int k = 0; k < 9; k++) g[k] = acc[k];
        }
    });
    return S2D_OK;
}

} // extern "C"
