// decode_fuzz.cpp -- TEST HARNESS.  Feeds mutated image files to the host program's readers (host/image_io.h: .s2di, PPM, PNG;
// host/jpeg_decode.h: baseline + progressive JPEG) in ONE process built with -fsanitize=address,undefined: a malformed file must
// be rejected (false) or decoded, never read or write out of bounds.  usage: decode_fuzz seed_file kind mutations rng_seed
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../2dgaussiansplatting_amd/host/image_io.h"

static uint64_t rng_state;
static uint32_t rnd()
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 16);
}

int main(int argc, char** argv)
{
    if (argc < 5) return 2;
    std::vector<uint8_t> seed;
    if (!s2dio::read_file(argv[1], &seed) || seed.empty()) return 2;
    const std::string kind = argv[2];
    const int n = std::atoi(argv[3]);
    rng_state = 0x9E3779B97F4A7C15ull ^ (uint64_t)std::atoll(argv[4]);
    int accepted = 0, rejected = 0;
    for (int k = 0; k < n; k++) {
        std::vector<uint8_t> d = seed;
        const int what = (int)(rnd() % 6);
        if (what == 0) d.resize(rnd() % d.size());                                     // truncated
        else if (what == 1) for (int j = 0; j < 1 + (int)(rnd() % 8); j++) d[rnd() % d.size()] ^= (uint8_t)(1u << (rnd() % 8));   // bit flips
        else if (what == 2) for (int j = 0; j < 1 + (int)(rnd() % 4); j++) d[rnd() % d.size()] = (uint8_t)rnd();                    // bytes
        else if (what == 3) { const size_t a = rnd() % d.size(); for (size_t j = a; j < d.size() && j < a + 2 + rnd() % 64; j++) d[j] = 0xFF; } // marker soup
        else if (what == 4) { const size_t a = rnd() % d.size(), b = rnd() % d.size(); if (a < b) d.erase(d.begin() + (long)a, d.begin() + (long)b); }  // a hole
        else { const size_t a = rnd() % std::min<size_t>(d.size(), 64); d[a] = (uint8_t)rnd(); }                                   // header damage
        s2dio::Image8 im;
        bool ok = false;
        if (kind == "jpg") {
            int w = 0, h = 0;
            std::vector<uint8_t> rgb;
            ok = s2dio::load_jpeg(d, &w, &h, &rgb);
            if (ok && rgb.size() != (size_t)w * (size_t)h * 3) return 3;   // accepted => consistent
        } else if (kind == "png") {
            ok = s2dio::load_png(d, &im);
            if (ok && im.rgb.size() != (size_t)im.w * (size_t)im.h * 3) return 3;
        } else if (kind == "ppm") {
            ok = s2dio::load_ppm(d, &im);
            if (ok && im.rgb.size() != (size_t)im.w * (size_t)im.h * 3) return 3;
        } else {
            ok = s2dio::load_s2di(d, &im);
            if (ok && im.rgb.size() != (size_t)im.w * (size_t)im.h * 3) return 3;
        }
        (ok ? accepted : rejected)++;
    }
    std::printf("%s: %d mutations, %d decoded, %d rejected\n", kind.c_str(), n, accepted, rejected);
    return 0;
}
