/*
 * splat2d_test.h -- test and inspection hooks of libsplat2d_hip.so.
 *
 * Not part of the training path and not part of the drop-in boundary (include/splat2d.h): tests/ reach single device
 * routines (the trig restatement, the radix sort, the scan) and the tile lists through these.  Same conventions as
 * splat2d.h: plain C, host pointers, int status returns.
 */
#ifndef SPLAT2D_TEST_H
#define SPLAT2D_TEST_H

#include "splat2d.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Device trig used by the projection kernel, evaluated on the GPU for n host floats. */
int s2d_test_sincos(int32_t device, const float* x, int32_t n, float* sin_out, float* cos_out);
/* Stable LSD radix sort of (key, value) pairs by the low `key_bits` bits of key, on the GPU. */
int s2d_test_sort_pairs(int32_t device, uint32_t* keys, uint32_t* values, int64_t n, int32_t key_bits);
/* Exclusive prefix sum on the GPU; returns the total in *total. */
int s2d_test_exclusive_scan(int32_t device, uint32_t* data, int64_t n, uint64_t* total);
/* The tile lists the raster kernels walk, for inspection: offsets has tiles+1 entries. */
int s2d_debug_get_tile_lists(s2d_ctx* ctx, int32_t* tiles_x, int32_t* tiles_y, uint32_t* offsets,
                             int64_t offsets_capacity, uint32_t* list, int64_t list_capacity);

/* Failure injection for the multi-device handle: rank `rank`'s worker thread stops answering right after it has queued
 * the raster launch of iteration `iteration` (before its gradient exchange) for `milliseconds` (< 0: until another rank has
 * declared it gone).  rank < 0 clears the hook. */
int s2d_test_multi_stall(s2d_multi* m, int32_t rank, int32_t iteration, int32_t milliseconds);

#ifdef __cplusplus
}
#endif
#endif /* SPLAT2D_TEST_H */
