/*
 * splat2d.h -- C ABI of the MI355X-native 2D Gaussian splatting trainer.
 *
 * The reference (Ushio/2dGaussianSplatting, /root/reference/main.cpp) has no
 * plugin / FFI interface: the forward rasteriser, the analytic backward pass and
 * the Adam step are inline loops inside main()'s frame loop (main.cpp:334-851)
 * that communicate through main()'s locals.  This boundary is therefore drawn
 * around the locals those three passes own:
 *
 *   std::vector<Splat> splats          main.cpp:272      <-> s2d_set_splats / s2d_get_splats
 *   std::vector<SplatAdam> splatAdams  main.cpp:276      <-> s2d_set_adam / s2d_get_adam
 *   float beta1t, beta2t               main.cpp:274-275  <-> s2d_set_adam / s2d_get_adam
 *   int iterations                     main.cpp:278      <-> s2d_set_adam / s2d_get_adam
 *   Image2DRGBA32 imageRef             main.cpp:254-259  <-> s2d_set_target
 *   Image2DRGBA32 image0               main.cpp:310      <-> s2d_get_image
 *   std::vector<Splat> dSplats         main.cpp:550      <-> s2d_get_grads
 *   bool optimizeOpacity               main.cpp:317      <-> S2D_STEP_OPTIMIZE_OPACITY
 *   float trainingRate                 main.cpp:715      <-> s2d_config.training_rate
 *   abort() on non-finite params       main.cpp:752-785  <-> status S2D_E_NONFINITE
 *
 * Conventions: plain C, opaque handle, int status returns (0 = OK), no exceptions
 * or aborts across the boundary.  All `const T*` / `T*` arguments are HOST pointers
 * owned by the caller unless the name says `_device`.  A context is driven by one
 * host thread; it owns its device memory; work is queued on its HIP stream and
 * host-visible results are complete when the call that returns them returns.
 *
 * There is NO CPU fallback: every entry point that computes runs hand-written
 * HIP kernels for gfx950 and fails with S2D_E_HIP when no device is usable.
 */
#ifndef SPLAT2D_H
#define SPLAT2D_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever a struct that crosses the boundary changes layout or an entry point changes meaning; added entry points
 * alone do not bump it.  s2d_abi_version() returns the value the LIBRARY was built with: a caller compares it with this
 * macro before anything else (the Python binding does, load_library()).
 *   2: s2d_stats starts with struct_size (round 2 had grown the struct in place under version 1). */
#define S2D_ABI_VERSION 2

/* == struct Splat, main.cpp:85-93 (vec2 pos; float sx, sy, rot; vec3 color; float opacity): 36 bytes.
 * Also the gradient record (dSplats, main.cpp:550). */
typedef struct s2d_splat {
    float pos[2];
    float sx, sy, rot;
    float color[3];
    float opacity;
} s2d_splat;

/* == struct Adam, main.cpp:139-157 */
typedef struct s2d_adam { float m, v; } s2d_adam;

/* == struct SplatAdam, main.cpp:158-166: 72 bytes */
typedef struct s2d_splat_adam {
    s2d_adam pos[2];
    s2d_adam sx, sy, rot;
    s2d_adam color[3];
    s2d_adam opacity;
} s2d_splat_adam;

typedef enum s2d_status {
    S2D_OK = 0,
    S2D_E_INVALID = 1,    /* bad argument / bad config */
    S2D_E_HIP = 2,        /* HIP runtime error or no usable gfx950 device (see s2d_last_error) */
    S2D_E_NONFINITE = 3,  /* a parameter the reference checks (main.cpp:752-785) became non-finite */
    S2D_E_NOMEM = 4,
    S2D_E_STATE = 5       /* call order violated (e.g. backward before forward, no target set) */
} s2d_status;

/* s2d_step / s2d_adam_step flags */
#define S2D_STEP_OPTIMIZE_OPACITY 0x1u /* the "Optimize opacity" checkbox, main.cpp:317, :735-738, :825 */

/* s2d_backward / s2d_forward_backward flags */
#define S2D_BWD_SKIP_OPACITY_GRAD 0x1u /* leave dSplats.opacity at zero.  The reference always accumulates it
                                        * (main.cpp:704) but reads it only when "Optimize opacity" is on (main.cpp:735):
                                        * a caller whose next s2d_adam_step runs without S2D_STEP_OPTIMIZE_OPACITY
                                        * may skip it.  s2d_step does so by itself. */
#define S2D_FB_SKIP_IMAGE 0x2u         /* s2d_forward_backward only: do not store image0 (the backward walk takes the final
                                        * colours from registers); s2d_get_image then returns an older frame.  For training
                                        * loops that never look at it. */

/* s2d_config.flags */
#define S2D_CFG_COUNT_PAIRS 0x1u   /* count visited / active pixel-splat pairs in the raster kernels (diagnostic) */
#define S2D_CFG_FP16_IMAGES 0x2u   /* BASELINE configs[4] "fp16 color / fp32 grads": the framebuffer and the target are
                                    * held in HBM as 4 x fp16 per pixel (round to nearest even); all arithmetic, the
                                    * gradients and the optimiser stay fp32.  Images still cross this ABI as RGBA32F.
                                    * Not the reference's arithmetic: results equal the reference run with imageRef and
                                    * image0 rounded to fp16 (tests/test_gpu_parity.py::test_fp16_images_*). */
#define S2D_CFG_DETERMINISTIC 0x4u /* bitwise reproducible gradients: tiles store their per-splat partial sums into
                                    * private slots and a gather pass adds them in a fixed order, instead of float
                                    * atomics whose arrival order varies from run to run (the forward pass is
                                    * deterministic either way).  About a sixth slower at 4096^2 / 10^6 splats. */
#define S2D_CFG_EXACT_EXP 0x8u     /* validation mode: exp_approx returns expf(x), the switch the reference keeps at
                                    * main.cpp:51 "for numerical varidation".  The analytic gradients (main.cpp:639-704)
                                    * are those of the true exponential, so in this mode the backward pass is the
                                    * derivative of the forward pass and a finite-difference check closes
                                    * (tests/test_fd_end_to_end.py).  Not combinable with S2D_CFG_COUNT_PAIRS /
                                    * S2D_CFG_FP16_IMAGES. */
#define S2D_CFG_ADAM_FP32 0x10u    /* Adam::optimize (main.cpp:155) with the quotient and subtraction in fp32, as the
                                    * reference's own MSVC build evaluates the unqualified `sqrt` (float overload).
                                    * Default: the double-precision quotient of a g++/clang++ build, which is what the
                                    * known-answer vectors pin.  The two differ by <= 1 ulp
                                    * of the parameter + ~3 ulp of the update per step; 100-iteration MSE traces by 2e-5
                                    * (tests/test_oracle_kat.py). */

#define S2D_CFG_GENERIC_BINNING 0x20u /* diagnostic: build the tile lists with the generic builder (all (tile, splat) pairs
                                    * radix-sorted by tile) even where the two-level one applies (images of up to 512 tile
                                    * columns, csrc/s2d_tilelists.hip).  Same lists either way; wider images always use it. */

typedef struct s2d_config {
    uint32_t struct_size;   /* = sizeof(s2d_config) */
    int32_t width, height;  /* imageRef.width(), .height() (main.cpp:254) */
    int32_t n_splats;       /* NSplat (main.cpp:271) */
    int32_t device;         /* HIP device ordinal */
    /* Row slab [row_begin, row_end) of the image this context rasterises.
     * 0,0 = the whole image.  row_begin must be a multiple of 16 (the tile height). */
    int32_t row_begin, row_end;
    float training_rate;    /* 0 -> 0.05f (main.cpp:715) */
    uint32_t flags;         /* S2D_CFG_* */
    /* Tile lists may be re-used across iterations: they are built from tile rectangles inflated by
     * rebin_margin + |sx - sy| pixels, and every iteration a device-side check forces a rebuild BEFORE the raster runs if
     * any splat's exact rectangle left its binned one, so results do not depend on these two knobs.
     * rebin_interval: 0 -> library default (rebuild only when the check fires); 1 -> rebuild every iteration;
     * K > 1 -> additionally rebuild at least every K iterations.  rebin_margin: 0 -> default (2 pixels). */
    int32_t rebin_interval;
    float rebin_margin;
    void* stream;           /* hipStream_t to queue work on; NULL -> the context creates its own */
} s2d_config;

typedef struct s2d_stats {
    uint32_t struct_size;      /* IN: sizeof(s2d_stats) as the caller was compiled; s2d_get_stats writes no more than that */
    uint32_t reserved;
    uint64_t pairs_binned;     /* (tile, splat) pairs in the current tile lists (index-range rendering: of the range built last) */
    uint64_t pairs_capacity;
    uint64_t rebins;           /* times the tile lists were rebuilt */
    uint64_t fwd_visited, fwd_active; /* S2D_CFG_COUNT_PAIRS: pairs inside the reference's x/y ranges, and those with T >= 1/256 */
    uint64_t bwd_visited, bwd_active;
    uint64_t fwd_staged, bwd_staged;  /* list entries the raster kernels actually staged (after tile retirement) */
    uint64_t fwd_wave_execs, bwd_wave_execs; /* (wave, entry) pairs whose blend body ran (>= 1 live lane covered) */
    uint64_t bwd_lane_hist[65];              /* ... of the backward pass, by number of active lanes (0..64) */
    int32_t iterations;        /* == `iterations`, main.cpp:278 */
    int32_t first_nonfinite_iteration; /* -1 if none */
    uint64_t fwd_staged_hit;   /* S2D_CFG_COUNT_PAIRS: staged entries that cover >= 1 pixel of their tile ... */
    uint64_t fwd_rows_hit;     /* ... and (staged entry, tile row) pairs with a non-empty column range (of 16 per entry) */
    uint64_t bwd_quadrant_execs; /* S2D_CFG_COUNT_PAIRS: bwd_wave_execs weighted by the number (1..4) of 4x4 quadrants of the
                                  * wave's 8x8 block that hold a live covered pixel */
} s2d_stats;

typedef struct s2d_ctx s2d_ctx;

int s2d_abi_version(void);

/* Allocates device state for (width, height, n_splats).  Splats are zero until s2d_init_splats / s2d_set_splats. */
int s2d_create(const s2d_config* cfg, s2d_ctx** out);
void s2d_destroy(s2d_ctx* ctx);

/* imageRef (main.cpp:254-259): width*height RGBA32F, row-major, .rgb in [0,1]; copied.  A slab context
 * (row_begin/row_end) is handed the same full-size buffer and uploads and keeps only its own rows: device memory for
 * imageRef and image0 is (row_end - row_begin) * width pixels each. */
int s2d_set_target(s2d_ctx* ctx, const float* rgba32f);
/* Fills imageRef on the device with ref(x,y) = (x/W, 1 - x/W, y/H, 1): the reference's commented generator
 * (main.cpp:261-267) plus a blue ramp, evaluated in fp32. */
int s2d_set_target_synthetic(s2d_ctx* ctx);

/* init(), main.cpp:280-305: pcg3d-seeded splats, Adam state zeroed, beta powers = 1, iterations = 0. */
int s2d_init_splats(s2d_ctx* ctx);
int s2d_set_splats(s2d_ctx* ctx, const s2d_splat* splats);
int s2d_get_splats(s2d_ctx* ctx, s2d_splat* splats);
/* Checkpointable optimiser state (main.cpp:274-278). */
int s2d_set_adam(s2d_ctx* ctx, const s2d_splat_adam* adams, float beta1t, float beta2t, int32_t iterations);
int s2d_get_adam(s2d_ctx* ctx, s2d_splat_adam* adams, float* beta1t, float* beta2t, int32_t* iterations);

/* Forward rasteriser, main.cpp:414-546 (rows of this context's slab).
 * No scene the reference's loops can run is refused for its size: the (tile, splat) pairs of the tile lists are addressed
 * with 32 bits, and a scene with more of them than S2D_CHUNK_PAIRS (environment, read at s2d_create; default 2^30) is
 * rendered by consecutive INDEX RANGES of the splats -- the lists of one range at a time, front to back like main.cpp:419
 * and :552, the per-pixel colour and throughput carried from range to range -- with the same bits as one set of lists
 * (forward, backward and s2d_step alike; only S2D_CFG_COUNT_PAIRS contexts answer S2D_E_NOMEM there). */
int s2d_forward(s2d_ctx* ctx);
/* image0 as uploaded at main.cpp:794: width*height RGBA32F, .w = 1.  Rows outside the slab are returned as 0. */
int s2d_get_image(s2d_ctx* ctx, float* rgba32f);
/* The slab's rows of image0 alone: (row_end - row_begin) * width RGBA32F, first row = row_begin (what a multi-GPU host
 * stitches together; s2d_get_image of a whole-image context returns the same bytes). */
int s2d_get_image_rows(s2d_ctx* ctx, float* rgba32f_rows);

/* Backward pass, main.cpp:548-712: accumulates this slab's contribution into the gradient buffer
 * (which s2d_adam_step / s2d_step re-zero after use, like main.cpp:550).  Needs s2d_forward first. */
int s2d_backward(s2d_ctx* ctx, uint32_t flags);
/* s2d_forward + s2d_backward in one kernel launch per tile (same results: a tile's backward walk needs only its own
 * pixels' final colours).  What s2d_step queues; for callers that put their own work between backward and Adam (the
 * multi-GPU exchange).  flags: S2D_BWD_SKIP_OPACITY_GRAD, S2D_FB_SKIP_IMAGE. */
int s2d_forward_backward(s2d_ctx* ctx, uint32_t flags);
int s2d_get_grads(s2d_ctx* ctx, s2d_splat* dsplats);

/* Adam + constraints + finite guard, main.cpp:714-785, on the current gradient buffer; then iterations++ (809). */
int s2d_adam_step(s2d_ctx* ctx, uint32_t flags);

/* `iters` whole iterations (main.cpp:414-809): forward, backward, Adam, MSE.  mse_out (may be NULL) receives
 * iters doubles: the value the reference prints for each iteration (main.cpp:796-807).  For a slab context the
 * values are this slab's sum of squared errors divided by (H*W*3), i.e. partial MSEs that add up over slabs.
 * Returns S2D_E_NONFINITE where the reference would abort(). */
int s2d_step(s2d_ctx* ctx, int32_t iters, uint32_t flags, double* mse_out);

/* MSE (main.cpp:796-805) of the framebuffer produced by the last s2d_forward / s2d_backward pair. */
int s2d_get_mse(s2d_ctx* ctx, double* mse);

/* ---- Slab ownership for multi-GPU runs (no counterpart in the reference, which is single-process; DESIGN.md
 * section 7, SURVEY.md section 8e "neighbour-only exchange").  A rank HOLDS splat i while the rows
 * [pos.y - reach - margin, pos.y + reach + margin], reach = 3*max(sx, sy) + 2 (the circle bounding the y-range of
 * main.cpp:489-491), meet its row slab.  Only held splats are projected, listed and updated by s2d_adam_step; the
 * host keeps the holders' copies identical by exchanging gradient rows (and, when a splat drifts into another
 * rank's halo, its 27-float state).  All pointers are DEVICE pointers owned by the caller; work is queued on the
 * context's stream. */
#define S2D_ROWS_GRADS 0  /* 9 floats per row: the gradient buffer (s2d_bind_grads_device or internal) */
#define S2D_ROWS_SPLATS 1 /* 9 floats per row: s2d_splat */
#define S2D_ROWS_ADAM 2   /* 18 floats per row: s2d_splat_adam */
/* masks[i] bit q = rank q holds splat i under the current parameters (row_bounds: world+1 host ints, rank q owns
 * rows row_bounds[q] .. row_bounds[q+1]); 0 for splats this context does not hold. world <= 32. */
int s2d_halo_masks(s2d_ctx* ctx, int32_t world, const int32_t* row_bounds, float margin_rows, uint32_t* masks_device);
/* This context holds exactly the splats with bit `rank` set.  added != 0: splats this context did not hold before
 * are among them (their rows were written with s2d_rows_scatter): the tile lists are rebuilt before the next
 * forward.  Splats that merely left keep their (now empty) list entries until the next regular rebuild.
 * masks_device == NULL: the context holds every splat again (its copy of all rows must be current). */
int s2d_halo_commit(s2d_ctx* ctx, const uint32_t* masks_device, int32_t rank, int32_t added);
/* out[j] = row ids[j] of the chosen array / row ids[j] = in[j] (ids distinct; out-of-range ids read 0 / are skipped) */
int s2d_rows_gather(s2d_ctx* ctx, int32_t what, const int32_t* ids_device, int32_t count, float* out_device);
int s2d_rows_scatter(s2d_ctx* ctx, int32_t what, const int32_t* ids_device, int32_t count, const float* in_device);
/* grads[rows[u]] = sum over ranks q = 0..world-1, in that order, of rank q's partial: src[u*world + q] = -1 (q does
 * not hold the row), -2 (q is this rank: the partial already in the gradient buffer), or a row index into recv.
 * The fixed order makes the sum bit-identical on every holder. */
int s2d_grads_combine(s2d_ctx* ctx, const int32_t* rows_device, int32_t n_rows, const int32_t* src_device, int32_t world,
                      const float* recv_device);

/* ---- multi-GPU plumbing (one process per GPU; the caller's collective runs between s2d_forward_backward and
 * s2d_adam_step, on the context's stream) ---- */
/* Use caller-owned DEVICE memory (n_splats * 9 floats, layout s2d_splat[n]) as the gradient buffer, so the host
 * can all-reduce it in place (RCCL).  NULL -> back to the context's own buffer.  The buffer must be 16-byte
 * aligned, zero when bound, and is re-zeroed by s2d_adam_step. */
int s2d_bind_grads_device(s2d_ctx* ctx, void* grads_device);
/* Device address of the gradient buffer currently in use. */
void* s2d_grads_device_ptr(s2d_ctx* ctx);
/* The hipStream_t the context queues its work on (s2d_config.stream, or the one it created): a caller that puts its
 * own device work between two calls -- the RCCL all-reduce and the peer copies of s2d_multi -- queues it here. */
void* s2d_stream(s2d_ctx* ctx);
/* Per-iteration sums of squared errors of this slab kept on the device (ring of `capacity` doubles indexed by
 * iteration % capacity); lets a multi-GPU host reduce them once after many steps instead of every iteration. */
int s2d_get_sqerr_trace(s2d_ctx* ctx, int32_t first_iteration, int32_t count, double* out);
int s2d_synchronize(s2d_ctx* ctx);

/* ---- several GPUs behind one handle ("device list ...; multi-GPU fan-out is internal") -------------
 * s2d_multi keeps the single-threaded call pattern of main.cpp:334 and runs it on n_devices GPUs: the image is cut into
 * n_devices row slabs (whole 16-pixel tile rows), every device gets an ordinary context for its slab and a worker thread
 * inside the library; the caller needs none.  cfg: as for s2d_create, with device / row_begin / row_end / stream unused
 * (must be 0).  Keeping the devices consistent:
 *  - default, slab ownership: a device holds and updates only the splats that can reach its rows; per iteration the
 *    holders of a shared splat swap its partial gradient rows by peer-to-peer copies (xGMI) and add them in rank order;
 *    every 64 iterations the hold sets follow the parameters and the state of a splat entering a neighbour's reach is
 *    handed over.  get_splats / get_adam assemble the arrays from the holders.
 *  - S2D_MULTI_REPLICATED (north_star's scheme): splats and Adam state on every device, the N x 9 fp32 gradient arrays
 *    summed in place by an RCCL all-reduce (on each context's stream, between s2d_forward_backward and s2d_adam_step),
 *    the identical Adam step everywhere.  RCCL is loaded (dlopen) when such a handle is created.
 * With deterministic gradients (S2D_CFG_DETERMINISTIC) and sums formed in rank order the two give the same bits.
 * After a failed s2d_multi_step (S2D_E_NONFINITE: the reference abort()s there; or one rank's own failure, reported with that
 * rank's number and message) the ranks stand at different iterations and the state is for inspection only: further steps are
 * refused (S2D_E_STATE) until s2d_multi_init_splats, or s2d_multi_set_splats + s2d_multi_set_adam, set it afresh -- unless a
 * collective had to be aborted or a rank stopped answering (s2d_multi_set_stall_timeout): then the handle must be re-created. */
#define S2D_MULTI_SHARE_GPU 0x1u  /* rehearsal on a box with fewer GPUs than ranks: all ranks on devices[0]; peer copies
                                   * become device copies, the all-reduce is staged through pinned host memory in rank
                                   * order (RCCL takes one rank per GPU) */
#define S2D_MULTI_REPLICATED 0x2u /* replicated state + all-reduce of all gradients instead of slab ownership */
typedef struct s2d_multi s2d_multi;
int s2d_multi_create(const s2d_config* cfg, const int32_t* devices, int32_t n_devices, uint32_t flags, s2d_multi** out);
void s2d_multi_destroy(s2d_multi* m);
const char* s2d_multi_last_error(const s2d_multi* m);
int s2d_multi_device_count(const s2d_multi* m);
/* Which GPU runs which rows: HIP device ordinal, row slab [row_begin, row_end), PCI bus id ("0000:c1:00.0") and marketing
 * name of rank `rank`'s device (any output may be NULL; strings are truncated to their capacity).  What a host prints so
 * that a multi-GPU record proves N distinct devices took part. */
int s2d_multi_device_info(s2d_multi* m, int32_t rank, int32_t* device, int32_t* row_begin, int32_t* row_end, char* pci_bus_id,
                          int32_t pci_capacity, char* name, int32_t name_capacity);
/* A rank that stops answering must not hang the caller (the reference's only failure policy is abort(), main.cpp:752-785;
 * this boundary turns failures into statuses, and "a rank stopped answering" is one of them).  Every wait of one rank for
 * another -- for the event of its gradient exchange, at the rendezvous of a hold-set refresh -- and for its own stream
 * gives up after `milliseconds` without progress: the step returns S2D_E_STATE, s2d_multi_last_error names the rank, the
 * exchange / iteration and what it was last seen doing, communicators are aborted, and the handle refuses further work
 * until it is destroyed and created again.  A rank that does not even answer the stop (it sits inside a runtime call) is
 * abandoned after about three times the limit: the call still returns.  Default 30 000 (S2D_MULTI_STALL_TIMEOUT_MS in the
 * environment overrides it at creation); 0 = wait without bound. */
int s2d_multi_set_stall_timeout(s2d_multi* m, int32_t milliseconds);
int s2d_multi_set_target(s2d_multi* m, const float* rgba32f);         /* imageRef, main.cpp:254-259, to every replica */
int s2d_multi_set_target_synthetic(s2d_multi* m);
int s2d_multi_init_splats(s2d_multi* m);                              /* init(), main.cpp:280-305, on every replica */
int s2d_multi_set_splats(s2d_multi* m, const s2d_splat* splats);
int s2d_multi_get_splats(s2d_multi* m, s2d_splat* splats);            /* every row from its lowest-ranked holder */
int s2d_multi_set_adam(s2d_multi* m, const s2d_splat_adam* adams, float beta1t, float beta2t, int32_t iterations);
int s2d_multi_get_adam(s2d_multi* m, s2d_splat_adam* adams, float* beta1t, float* beta2t, int32_t* iterations);
/* `iters` whole iterations (main.cpp:414-809) on all devices; mse_out as for s2d_step (the slabs' squared errors are
 * added in slab order).  S2D_E_NONFINITE where the reference would abort(). */
int s2d_multi_step(s2d_multi* m, int32_t iters, uint32_t flags, double* mse_out);
/* Forward rasteriser (main.cpp:414-546) of the current parameters on every device's rows, e.g. for display (main.cpp:794). */
int s2d_multi_forward(s2d_multi* m);
/* image0 of the last s2d_multi_forward, or of the last iteration of the last s2d_multi_step, assembled from the slabs. */
int s2d_multi_get_image(s2d_multi* m, float* rgba32f);
/* out4: scheme in use (0 none: one device, 1 slab ownership, 2 replicated), gradient rows swapped per iteration (all
 * ranks), state rows handed over so far, splats held summed over the ranks (n_splats * n_devices when replicated). */
int s2d_multi_exchange_info(s2d_multi* m, int64_t* out4);

/* out->struct_size must be set by the caller (S2D_E_INVALID when it is smaller than the first version of the struct). */
int s2d_get_stats(s2d_ctx* ctx, s2d_stats* out);
/* s2d_stats.rebins without the device round trip s2d_get_stats makes (a host-side counter; never synchronises). */
int s2d_get_rebuild_count(const s2d_ctx* ctx, uint64_t* rebuilds);
const char* s2d_last_error(const s2d_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* SPLAT2D_H */
