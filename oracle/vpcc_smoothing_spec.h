/*
 * vpcc_smoothing_spec.h — CPU SPECIFICATION of grid-based geometry and colour smoothing.
 * TEST INFRASTRUCTURE ONLY (same rules as vpcc_oracle.h: only tests/, smoke and bench's baseline leg).
 *
 * NOT a restatement of the reference: benclmnt/tmc2-rs has no smoothing — every hook is
 * `unimplemented!()` (src/decoder.rs:291-299, 630-658; src/codec.rs:498-500) and only the SEI syntax
 * is parsed (src/bitstream/reader.rs:1452-1505).  Upstream TMC2's PCCCodec.cpp is not available in this
 * environment either.  This file therefore DEFINES the behaviour of this repository's smoothing
 * ("build must publish its own CPU spec and test GPU against that"), in integer
 * arithmetic only so that CPU and GPU agree bit for bit.  PARITY WITH ANY REFERENCE: NONE EXISTS.
 *
 * Geometry smoothing "gs1" (grid size G, threshold T, 3-D bit depth B):
 *   cells per axis w = ceil(2^B / G); cell coordinate of a point = min(p / G, w - 1) per axis.
 *   Pass 1, over the ORIGINAL positions: per cell count, coordinate sums, min and max patch index.
 *   Pass 2, per point (reads only its own original position and the cell statistics):
 *     per axis: q = cell coordinate, r = p - q*G, s = (r < G/2) ? q - 1 : q   (lower cell of the pair),
 *               t = 2*(p - (s*G + G/2)) + 1, weights {lower: 2G - t, upper: t};
 *     over the 2x2x2 cells (s + d), skipping cells outside [0, w) and empty cells:
 *       W = wx*wy*wz;  num += W * sum[c];  den += W * count[c];  mixed |= (minpatch[c] != maxpatch[c]);
 *     if (mixed && den > 0):  C[a] = floor((16*num[a] + den/2) / den)      (centroid in 1/16 units)
 *                             d2 = sum_a (16*p[a] - C[a])^2
 *                             if (d2 > 256*T*T)  p[a] = min(65535, (C[a] + 8) >> 4)
 * Colour smoothing "cs1" (grid size Gc, thresholds Ts, Td), on the positions AFTER geometry smoothing
 * and the 8-bit RGB colours:
 *   Pass 1: per cell count, R/G/B sums, min and max patch index;  mean[c][ch] = floor(sum/count).
 *   Pass 2: same 2x2x2 neighbourhood and weights (with Gc); a neighbour cell takes part only if
 *     sum_ch |mean[c][ch] - mean[centre][ch]| <= Td (the point's own cell always does);
 *     if (mixed && den > 0):  m[ch] = floor((2*num[ch] + den) / (2*den));
 *                             if (sum_ch |colour[ch] - m[ch]| >= Ts)  colour = m.
 */
#ifndef VPCC_SMOOTHING_SPEC_H
#define VPCC_SMOOTHING_SPEC_H
#include "../include/vpcc_recon.h"
#ifdef __cplusplus
extern "C" {
#endif
/* In place.  patch_index[i] = partition entry of point i.  Returns 0, or VPCC_ERR_INVALID_ARG. */
int vpcc_spec_smooth_geometry(vpcc_point3* xyz, const uint16_t* patch_index, size_t n, uint32_t bitdepth,
                              uint32_t grid_size, uint32_t threshold);
int vpcc_spec_smooth_color(const vpcc_point3* xyz, vpcc_color3* rgb, const uint16_t* patch_index, size_t n,
                           uint32_t bitdepth, uint32_t grid_size, uint32_t threshold_smoothing,
                           uint32_t threshold_difference);
#ifdef __cplusplus
}
#endif
#endif
