// splat2d_train.cpp -- headless C++ host loop on top of the C ABI (include/splat2d.h).
//
// Mirrors the shape of the reference's main() (/root/reference/main.cpp:236-856) without its window:
//   load imageRef (main.cpp:253-259)  ->  NSplat splats (:271-272)  ->  init() (:280-307)  ->
//   loop { forward (:414-546); backward (:548-712); Adam + constraints (:714-785); MSE (:796-805);
//          printf("%d itr, mse %.4f\n") (:807); iterations++ (:809) }
// The three passes run on the MI355X through s2d_step(); this file owns only what main() owns besides them:
// the image, the flags the GUI exposed ("Optimize opacity" :825, "Restart" :828-831) and the trace line.
// It is plain C++ (g++), links the library, and has no compute of its own.
//
//   splat2d_train --image tests/golden/squirrel_cls_mini_268x213.s2di --splats 1024 --iters 300
//   splat2d_train --synthetic 4096x4096 --splats 1000000 --iters 100 --quiet
//   splat2d_train --synthetic 4096x4096 --splats 1000000 --iters 100 --quiet --gpus 8
//
// --gpus N: the same loop, with every option, on a multi-device handle (s2d_multi_*): the image cut
// into N row slabs, one context per GPU, and between the backward pass and the Adam step either the holders of a
// boundary splat swap its gradient rows by peer-to-peer copies (slab ownership, default) or RCCL all-reduces all
// N x 9 gradients over xGMI (--exchange dense, north_star's scheme) -- all inside the library.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "image_io.h"
#include "overlay.h"
#include "splat2d.h"

namespace {

struct Options {
    std::string image;          // .s2di fixture (raw RGB8, see tools/make_image_fixtures.py), binary .ppm or .png
    int syn_w = 0, syn_h = 0;   // --synthetic WxH
    int n_splats = 1024;        // NSplat, main.cpp:271
    int iters = 300;
    int batch = 1;              // iterations queued per s2d_step call (1 = print after every iteration, like the reference)
    bool optimize_opacity = false;
    int opacity_from = 0;       // iteration from which the flag is on (the GUI checkbox can be ticked mid-run)
    int restart_at = -1;        // press "Restart" before this iteration (main.cpp:828-831)
    bool quiet = false;
    std::string out_image;      // --out-image file.(png|ppm|s2di): image0 after the last iteration (main.cpp:794)
    std::string overlay;        // --overlay file: image0 upscaled by overlay_scale with the splat debug drawing (main.cpp:441-485)
    int overlay_scale = 2;      // viewScale, main.cpp:822
    int overlay_stride = 1;     // draw every k-th splat
    std::string overlay_dump;   // --overlay-vertices file: the pr::PrimVertex list of main.cpp:447-476 as drawn (binary, see below)
    std::string convert_in, convert_out; // --convert in out: image conversion only (no GPU)
    std::string load_ckpt, save_ckpt; // the five objects of main.cpp:272-278: splats, splatAdams, beta1t, beta2t, iterations
    int device = 0;
    int rebin_interval = 0;
    float lr = 0.0f;            // --lr: trainingRate, main.cpp:715 (0 = the reference's 0.05)
    bool deterministic = false; // --deterministic: bitwise reproducible gradient sums (S2D_CFG_DETERMINISTIC)
    int stall_ms = -1;          // --stall-timeout-ms (multi-device handle): how long a rank may not answer before the step fails
    int gpus = 1;               // --gpus N: devices device .. device + N - 1, one row slab each, RCCL all-reduce of the gradients
    bool share_gpu = false;     // --share-gpu: all N ranks on --device (rehearsal on a box with fewer GPUs than ranks)
    bool replicated = false;    // --exchange dense: replicated state + RCCL all-reduce of all gradients (default: slab ownership)
                                // (RCCL takes one rank per GPU): a rehearsal of the N-rank host logic on a box with fewer GPUs
};

// Checkpoint = the state main() keeps between frames (main.cpp:272-278), in the reference's own layouts.
struct CkptHeader {
    char magic[4];      // "S2DC"
    uint32_t n_splats, width, height;
    float beta1t, beta2t;
    int32_t iterations;
};

int usage()
{
    std::fprintf(stderr,
                 "usage: splat2d_train (--image file.s2di|.ppm|.png|.jpg | --synthetic WxH) [--splats N] [--iters K] [--batch B]\n"
                 "                     [--optimize-opacity [--opacity-from IT]] [--restart-at IT] [--out-image file.png|.ppm]\n"
                 "                     [--overlay file [--overlay-scale S] [--overlay-stride K] [--overlay-vertices file]]\n"
                 "       splat2d_train --convert in.(s2di|ppm|png|jpg) out.(s2di|ppm|png)\n"
                 "                     [--load-checkpoint file] [--save-checkpoint file]\n"
                 "                     [--lr RATE] [--deterministic] [--device D] [--gpus N [--exchange halo|dense] [--share-gpu]\n"
                 "                     [--stall-timeout-ms MS]] [--rebin-interval R] [--quiet]\n");
    return 2;
}

// One context, or a multi-device handle, behind the same calls: what main() below talks to.
struct Session {
    s2d_ctx* ctx = nullptr;
    s2d_multi* multi = nullptr;
    bool is_multi = false;

    const char* last_error() const { return is_multi ? s2d_multi_last_error(multi) : s2d_last_error(ctx); }
    int create(const Options& o, int W, int H)
    {
        s2d_config cfg;
        std::memset(&cfg, 0, sizeof(cfg));
        cfg.struct_size = sizeof(cfg);
        cfg.width = W;
        cfg.height = H;
        cfg.n_splats = o.n_splats; // int NSplat = 1024; main.cpp:271
        cfg.rebin_interval = o.rebin_interval;
        cfg.training_rate = o.lr;
        if (o.deterministic) cfg.flags |= S2D_CFG_DETERMINISTIC;
        is_multi = o.gpus > 1 || std::getenv("S2D_TRAIN_FORCE_MULTI"); // (the variable sends --gpus 1 through the handle)
        if (!is_multi) {
            cfg.device = o.device;
            return s2d_create(&cfg, &ctx);
        }
        std::vector<int32_t> devs((size_t)o.gpus);
        for (int r = 0; r < o.gpus; r++) devs[(size_t)r] = o.device + (o.share_gpu ? 0 : r);
        return s2d_multi_create(&cfg, devs.data(), o.gpus,
                                (o.share_gpu ? S2D_MULTI_SHARE_GPU : 0u) | (o.replicated ? S2D_MULTI_REPLICATED : 0u), &multi);
    }
    void destroy()
    {
        if (ctx) s2d_destroy(ctx);
        if (multi) s2d_multi_destroy(multi);
        ctx = nullptr;
        multi = nullptr;
    }
    int set_target(const std::vector<float>& rgba)
    {
        if (rgba.empty()) return is_multi ? s2d_multi_set_target_synthetic(multi) : s2d_set_target_synthetic(ctx);
        return is_multi ? s2d_multi_set_target(multi, rgba.data()) : s2d_set_target(ctx, rgba.data());
    }
    int init() { return is_multi ? s2d_multi_init_splats(multi) : s2d_init_splats(ctx); }
    int set_splats(const s2d_splat* p) { return is_multi ? s2d_multi_set_splats(multi, p) : s2d_set_splats(ctx, p); }
    int get_splats(s2d_splat* p) { return is_multi ? s2d_multi_get_splats(multi, p) : s2d_get_splats(ctx, p); }
    int set_adam(const s2d_splat_adam* p, float b1, float b2, int32_t it)
    {
        return is_multi ? s2d_multi_set_adam(multi, p, b1, b2, it) : s2d_set_adam(ctx, p, b1, b2, it);
    }
    int get_adam(s2d_splat_adam* p, float* b1, float* b2, int32_t* it)
    {
        return is_multi ? s2d_multi_get_adam(multi, p, b1, b2, it) : s2d_get_adam(ctx, p, b1, b2, it);
    }
    int step(int iters, uint32_t flags, double* mse) { return is_multi ? s2d_multi_step(multi, iters, flags, mse) : s2d_step(ctx, iters, flags, mse); }
    int forward() { return is_multi ? s2d_multi_forward(multi) : s2d_forward(ctx); }
    int get_image(float* rgba) { return is_multi ? s2d_multi_get_image(multi, rgba) : s2d_get_image(ctx, rgba); }
    // how the devices were kept consistent, for the summary line
    std::string exchange_summary(const Options& o)
    {
        int64_t info[4] = {0, 0, 0, 0};
        if (!is_multi || s2d_multi_exchange_info(multi, info) != S2D_OK) return "";
        char how[240];
        if (info[0] == 2)
            std::snprintf(how, sizeof(how), ", %d ranks: row slabs; replicated state, %s of all gradients", o.gpus,
                          o.share_gpu ? "host-staged sum (ranks share one GPU)" : "RCCL all-reduce");
        else if (info[0] == 1)
            std::snprintf(how, sizeof(how), ", %d ranks: row slabs; slab ownership, %lld gradient rows swapped per iteration by %s, %lld state rows handed over",
                          o.gpus, (long long)info[1], o.share_gpu ? "device copies (ranks share one GPU)" : "peer-to-peer copies", (long long)info[2]);
        else
            std::snprintf(how, sizeof(how), ", %d ranks: one device, nothing to exchange", o.gpus);
        return how;
    }
};

#define CK(call)                                                                        \
    do {                                                                                \
        int rc_ = (call);                                                               \
        if (rc_ != S2D_OK) {                                                            \
            std::fprintf(stderr, "%s -> %d: %s\n", #call, rc_, S.last_error());         \
            S.destroy();                                                                \
            return rc_ == S2D_E_NONFINITE ? 3 : 1; /* the reference abort()s, main.cpp:752-785 */ \
        }                                                                               \
    } while (0)

} // namespace

int main(int argc, char** argv)
{
    Options o;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&](const char* what) -> const char* {
            if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", what); std::exit(2); }
            return argv[++i];
        };
        if (a == "--image") o.image = next("--image");
        else if (a == "--synthetic") { if (std::sscanf(next("--synthetic"), "%dx%d", &o.syn_w, &o.syn_h) != 2) return usage(); }
        else if (a == "--splats") o.n_splats = std::atoi(next("--splats"));
        else if (a == "--iters") o.iters = std::atoi(next("--iters"));
        else if (a == "--batch") o.batch = std::atoi(next("--batch"));
        else if (a == "--optimize-opacity") o.optimize_opacity = true;
        else if (a == "--opacity-from") o.opacity_from = std::atoi(next("--opacity-from"));
        else if (a == "--restart-at") o.restart_at = std::atoi(next("--restart-at"));
        else if (a == "--out-ppm" || a == "--out-image") o.out_image = next("--out-image");
        else if (a == "--overlay") o.overlay = next("--overlay");
        else if (a == "--overlay-scale") o.overlay_scale = std::atoi(next("--overlay-scale"));
        else if (a == "--overlay-stride") o.overlay_stride = std::atoi(next("--overlay-stride"));
        else if (a == "--overlay-vertices") o.overlay_dump = next("--overlay-vertices");
        else if (a == "--convert") { o.convert_in = next("--convert"); o.convert_out = next("--convert"); }
        else if (a == "--load-checkpoint") o.load_ckpt = next("--load-checkpoint");
        else if (a == "--save-checkpoint") o.save_ckpt = next("--save-checkpoint");
        else if (a == "--device") o.device = std::atoi(next("--device"));
        else if (a == "--gpus") o.gpus = std::atoi(next("--gpus"));
        else if (a == "--share-gpu") o.share_gpu = true;
        else if (a == "--exchange") {
            const std::string v = next("--exchange");
            if (v != "halo" && v != "dense") return usage();
            o.replicated = v == "dense";
        }
        else if (a == "--rebin-interval") o.rebin_interval = std::atoi(next("--rebin-interval"));
        else if (a == "--quiet") o.quiet = true;
        else if (a == "--lr") o.lr = (float)std::atof(next("--lr"));
        else if (a == "--deterministic") o.deterministic = true;
        else if (a == "--stall-timeout-ms") o.stall_ms = std::atoi(next("--stall-timeout-ms"));
        else return usage();
    }
    if (!o.convert_in.empty()) { // file conversion between .s2di / .ppm / .png; touches no GPU
        s2dio::Image8 im;
        if (!s2dio::load_image(o.convert_in, &im)) { std::fprintf(stderr, "cannot read %s\n", o.convert_in.c_str()); return 1; }
        if (!s2dio::save_image(o.convert_out, im)) { std::fprintf(stderr, "cannot write %s\n", o.convert_out.c_str()); return 1; }
        return 0;
    }
    if (o.image.empty() == (o.syn_w == 0) || o.n_splats < 0 || o.iters < 0 || o.batch < 1 || o.overlay_scale < 1) return usage();

    // imageRef, main.cpp:253-259
    int W = o.syn_w, H = o.syn_h;
    std::vector<float> imageRef;
    if (!o.image.empty()) {
        s2dio::Image8 im;
        if (!s2dio::load_image(o.image, &im)) {
            std::fprintf(stderr, "cannot read %s (.s2di, binary .ppm, 8-bit non-interlaced .png and Huffman .jpg are supported)\n", o.image.c_str());
            return 1;
        }
        W = im.w;
        H = im.h;
        const std::vector<uint8_t>& rgb = im.rgb;
        imageRef.resize((size_t)W * H * 4);
        for (size_t p = 0; p < (size_t)W * H; p++) { // Image2DRGBA8_to_Image2DRGBA32: byte / 255.0f
            for (int c = 0; c < 3; c++) imageRef[p * 4 + c] = (float)rgb[p * 3 + c] / 255.0f;
            imageRef[p * 4 + 3] = 1.0f;
        }
    }

    if (o.gpus < 1) return usage();
    Session S;
    CK(S.create(o, W, H));
    if (S.is_multi && o.stall_ms >= 0) CK(s2d_multi_set_stall_timeout(S.multi, o.stall_ms));
    if (S.is_multi) { // which GPU runs which rows (stderr: stdout is the reference's trace)
        for (int r = 0; r < s2d_multi_device_count(S.multi); r++) {
            int32_t dev = 0, r0 = 0, r1 = 0;
            char pci[64] = "", name[128] = "";
            if (s2d_multi_device_info(S.multi, r, &dev, &r0, &r1, pci, (int32_t)sizeof(pci), name, (int32_t)sizeof(name)) == S2D_OK)
                std::fprintf(stderr, "rank %d: device %d (%s, %s), rows %d..%d\n", r, dev, pci, name, r0, r1);
        }
    }
    CK(S.set_target(imageRef)); // (empty: the synthetic target is generated on the device)
    CK(S.init());               // init(); main.cpp:307

    int iterations = 0; // main.cpp:278
    if (!o.load_ckpt.empty()) {
        FILE* f = std::fopen(o.load_ckpt.c_str(), "rb");
        CkptHeader h;
        std::vector<s2d_splat> sp((size_t)o.n_splats);
        std::vector<s2d_splat_adam> ad((size_t)o.n_splats);
        const bool ok = f && std::fread(&h, sizeof(h), 1, f) == 1 && std::memcmp(h.magic, "S2DC", 4) == 0 &&
                        (int)h.n_splats == o.n_splats && (int)h.width == W && (int)h.height == H &&
                        std::fread(sp.data(), sizeof(s2d_splat), sp.size(), f) == sp.size() &&
                        std::fread(ad.data(), sizeof(s2d_splat_adam), ad.size(), f) == ad.size();
        if (f) std::fclose(f);
        if (!ok) {
            std::fprintf(stderr, "cannot load checkpoint %s (wrong size or format)\n", o.load_ckpt.c_str());
            S.destroy();
            return 1;
        }
        CK(S.set_splats(sp.data()));
        CK(S.set_adam(ad.data(), h.beta1t, h.beta2t, h.iterations));
        iterations = h.iterations;
        o.iters += iterations; // --iters counts iterations to run from the checkpoint
    }
    std::vector<double> mse((size_t)o.batch);
    int frames = 0; // iterations run by this process (the trace numbering restarts at Restart, this does not)
    const auto t0 = std::chrono::steady_clock::now();
    while (iterations < o.iters) { // while (pr::NextFrame() == false), main.cpp:334
        if (iterations == o.restart_at) { // ImGui::Button("Restart"), main.cpp:828-831: init() also sets
            CK(S.init());                 // iterations = 0 (main.cpp:281), so the trace restarts at "0 itr"
            o.iters -= iterations;        // --iters is the number of frames to run in total
            iterations = 0;
            o.restart_at = -1;
        }
        int k = o.batch;
        if (k > o.iters - iterations) k = o.iters - iterations;
        if (o.optimize_opacity && iterations < o.opacity_from && iterations + k > o.opacity_from) k = o.opacity_from - iterations;
        if (o.restart_at > iterations && iterations + k > o.restart_at) k = o.restart_at - iterations;
        const bool opacity_now = o.optimize_opacity && iterations >= o.opacity_from; // bool optimizeOpacity, main.cpp:317
        CK(S.step(k, opacity_now ? S2D_STEP_OPTIMIZE_OPACITY : 0u, mse.data()));
        if (!o.quiet)
            for (int j = 0; j < k; j++) std::printf("%d itr, mse %.4f\n", iterations + j, mse[(size_t)j]); // main.cpp:807
        iterations += k; // main.cpp:809
        frames += k;
    }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::fprintf(stderr, "%d iterations in %.3f s = %.2f it/s (%dx%d, %d splats%s)\n", frames, secs,
                 secs > 0 ? frames / secs : 0.0, W, H, o.n_splats, S.exchange_summary(o).c_str());
    int exit_code = 0; // an output file that cannot be written is an error, reported after everything else was tried

    if (!o.save_ckpt.empty()) {
        CkptHeader h;
        std::memcpy(h.magic, "S2DC", 4);
        h.n_splats = (uint32_t)o.n_splats; h.width = (uint32_t)W; h.height = (uint32_t)H;
        std::vector<s2d_splat> sp((size_t)o.n_splats);
        std::vector<s2d_splat_adam> ad((size_t)o.n_splats);
        CK(S.get_splats(sp.data()));
        CK(S.get_adam(ad.data(), &h.beta1t, &h.beta2t, &h.iterations));
        FILE* f = std::fopen(o.save_ckpt.c_str(), "wb");
        const bool ok = f && std::fwrite(&h, sizeof(h), 1, f) == 1 &&
                        std::fwrite(sp.data(), sizeof(s2d_splat), sp.size(), f) == sp.size() &&
                        std::fwrite(ad.data(), sizeof(s2d_splat_adam), ad.size(), f) == ad.size();
        if (f) std::fclose(f);
        if (!ok) {
            std::fprintf(stderr, "cannot write %s\n", o.save_ckpt.c_str());
            exit_code = 1;
        }
    }
    if (!o.out_image.empty() || !o.overlay.empty()) {
        std::vector<float> image0((size_t)W * H * 4);
        CK(S.forward());
        CK(S.get_image(image0.data())); // tex0->upload(image0), main.cpp:794
        const s2dio::Image8 im = s2dio::quantise(image0, W, H);
        if (!o.out_image.empty() && !s2dio::save_image(o.out_image, im)) {
            std::fprintf(stderr, "cannot write %s\n", o.out_image.c_str());
            exit_code = 1;
        }
        if (!o.overlay.empty()) { // the reference's splat visualisation, main.cpp:441-485
            std::vector<s2d_splat> sp((size_t)o.n_splats);
            CK(S.get_splats(sp.data()));
            s2dio::Image8 big = s2dio::upscale(im, o.overlay_scale);
            std::vector<s2dio::OverlayVertex> verts;
            s2dio::draw_splat_overlay(&big, sp, o.overlay_scale, o.overlay_stride, o.overlay_dump.empty() ? nullptr : &verts);
            if (!o.overlay_dump.empty()) {
                // "S2DV", int32 vertex count, then 16 bytes per vertex: float x, y, z (scene coordinates of main.cpp:447:
                // x, -y, 0), uint8 r, g, b, 0 -- 46 vertices per drawn splat, in the reference's PrimVertex order
                FILE* f = std::fopen(o.overlay_dump.c_str(), "wb");
                const int32_t count = (int32_t)verts.size();
                const bool ok = f && std::fwrite("S2DV", 1, 4, f) == 4 && std::fwrite(&count, 4, 1, f) == 1 &&
                                std::fwrite(verts.data(), sizeof(s2dio::OverlayVertex), verts.size(), f) == verts.size();
                if (f) std::fclose(f);
                if (!ok) {
                    std::fprintf(stderr, "cannot write %s\n", o.overlay_dump.c_str());
                    exit_code = 1;
                }
            }
            if (!s2dio::save_image(o.overlay, big)) {
                std::fprintf(stderr, "cannot write %s\n", o.overlay.c_str());
                exit_code = 1;
            }
        }
    }
    S.destroy();
    return exit_code;
}
