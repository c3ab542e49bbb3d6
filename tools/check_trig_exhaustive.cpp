// Exhaustive check of the kernels' sinf/cosf restatement (s2d_math.h) against this machine's libm:
// every float with |x| < 120, both functions.  Build-container tool, not part of the product.
//   g++ -O2 -ffp-contract=off -mfma -std=c++17 tools/check_trig_exhaustive.cpp -o /tmp/check_trig -lpthread && /tmp/check_trig
// Result in the build container (glibc 2.35, FMA+AVX2 host): "mismatch sin 0 cos 0".
#include "../2dgaussiansplatting_amd/csrc/s2d_math.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <pthread.h>

struct Job { uint64_t lo, hi; uint64_t bad_s = 0, bad_c = 0; uint32_t first = 0; };

static void* run(void* a)
{
    Job* j = (Job*)a;
    for (uint64_t u = j->lo; u < j->hi; u++) {
        uint32_t v = (uint32_t)u;
        float f;
        std::memcpy(&f, &v, 4);
        if (!(std::fabs(f) < 120.0f)) continue;
        float a1 = sinf(f), a2 = s2d::sinf_ref(f), b1 = cosf(f), b2 = s2d::cosf_ref(f);
        if (s2d::f32_bits(a1) != s2d::f32_bits(a2)) { if (!j->bad_s && !j->bad_c) j->first = v; j->bad_s++; }
        if (s2d::f32_bits(b1) != s2d::f32_bits(b2)) { if (!j->bad_s && !j->bad_c) j->first = v; j->bad_c++; }
    }
    return nullptr;
}

int main()
{
    const int NT = 8;
    pthread_t th[NT];
    Job js[NT];
    const uint64_t total = 1ull << 32;
    for (int t = 0; t < NT; t++) {
        js[t].lo = total * t / NT;
        js[t].hi = total * (t + 1) / NT;
        pthread_create(&th[t], nullptr, run, &js[t]);
    }
    uint64_t bs = 0, bc = 0;
    for (int t = 0; t < NT; t++) {
        pthread_join(th[t], nullptr);
        bs += js[t].bad_s;
        bc += js[t].bad_c;
        if (js[t].bad_s || js[t].bad_c) std::printf("first mismatch in chunk %d: 0x%08x\n", t, js[t].first);
    }
    std::printf("mismatch sin %llu cos %llu\n", (unsigned long long)bs, (unsigned long long)bc);
    return (bs || bc) ? 1 : 0;
}
