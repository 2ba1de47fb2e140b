/* vpcc_colour.h — fast, exact evaluation of convert_yuv10_to_rgb8 (reference src/codec.rs:661-687).
 *
 * Shared, as the SAME source, by the gfx950 kernels and by tests/colour_exhaustive.c, which checks it
 * on the CPU against the reference formula for every 10-bit (Y,U,V) triplet.  Plain C99.
 *
 * The reference computes, in f64 and in this order, c = y + k*(v-512) ..., p = c/1023*255 and
 * clamp(floor(p)).  With T the exact rational value of p, |p - T| < 2e-13, and T is either an integer
 * (10 364 triplets) or at least 1/6 820 000 away from one.  Here T is accumulated with f64 FMAs on a
 * 2^-20 grid: adding 1.5*2^32 makes the low dword of the double the fixed-point number T*2^20 (two's
 * complement, 12.20 bits), so floor and clamp are integer shifts and min/max, and no division is left.
 * Every constant and every FMA rounds to that grid (<= 2 grid units in total); the constants carry a
 * +4 unit bias, so whenever the 20 fraction bits are >= 8 the integer part IS floor(T) and T is not an
 * integer — the reference's floor(p) is the same number.  Otherwise (8 of 2^20 fraction patterns,
 * which include all exact-integer cases) the caller evaluates the reference formula itself.
 */
#ifndef VPCC_COLOUR_H_
#define VPCC_COLOUR_H_

#include <stdint.h>

#if defined(__HIPCC__)
#define VPCC_COLOUR_FN __host__ __device__ __forceinline__
#else
#include <math.h>
#include <string.h>
#define VPCC_COLOUR_FN static inline
#endif

/* 255/1023 * {1, 1.57480, -0.18733, -0.46813, 1.85563}, correctly rounded from the exact rationals */
#define VPCC_C_AY 0x1.fe7f9fe7f9fe8p-3
#define VPCC_C_RV 0x1.91f76f85dd5ecp-2
#define VPCC_C_GU (-0x1.7e86d9d3f2bbfp-5)
#define VPCC_C_GV (-0x1.ddf5986eb5afep-4)
#define VPCC_C_BU 0x1.d9a5f03e360e2p-2
/* 1.5*2^32 + 4*2^-20 - 512 * (chroma coefficients of the channel) */
#define VPCC_K_R 0x1.7fffff3704488p+32
#define VPCC_K_G 0x1.80000053a720fp+32
#define VPCC_K_B 0x1.7fffff132d082p+32

typedef struct { double r, g, b; } vpcc_chroma_part;   /* shared by the pixels that use one chroma sample */

VPCC_COLOUR_FN vpcc_chroma_part vpcc_colour_chroma(uint32_t U, uint32_t V) {
  const double u = (double)U, v = (double)V;
  vpcc_chroma_part c;
  c.r = fma(v, VPCC_C_RV, VPCC_K_R);
  c.g = fma(v, VPCC_C_GV, fma(u, VPCC_C_GU, VPCC_K_G));
  c.b = fma(u, VPCC_C_BU, VPCC_K_B);
  return c;
}

VPCC_COLOUR_FN int32_t vpcc_colour_lo(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __double2loint(x);
#else
  uint64_t b;
  memcpy(&b, &x, 8);
  return (int32_t)(uint32_t)b;
#endif
}

VPCC_COLOUR_FN int32_t vpcc_colour_clamp(int32_t q) { return q < 0 ? 0 : (q > 255 ? 255 : q); }

/* r | g << 8 | b << 16.  *fraction_min is lowered to the smallest masked fraction field seen: the caller
 * must use the reference formula for every pixel that contributed when it ends up 0. */
VPCC_COLOUR_FN uint32_t vpcc_colour_luma(uint32_t Y, vpcc_chroma_part c, uint32_t* fraction_min) {
  const double y = (double)Y;
  const int32_t ir = vpcc_colour_lo(fma(y, VPCC_C_AY, c.r));
  const int32_t ig = vpcc_colour_lo(fma(y, VPCC_C_AY, c.g));
  const int32_t ib = vpcc_colour_lo(fma(y, VPCC_C_AY, c.b));
  const uint32_t fr = (uint32_t)ir & 0xFFFF8u, fg = (uint32_t)ig & 0xFFFF8u, fb = (uint32_t)ib & 0xFFFF8u;
  const uint32_t m = fr < fg ? (fr < fb ? fr : fb) : (fg < fb ? fg : fb);
  if (m < *fraction_min) *fraction_min = m;
  return (uint32_t)vpcc_colour_clamp(ir >> 20) | ((uint32_t)vpcc_colour_clamp(ig >> 20) << 8) |
         ((uint32_t)vpcc_colour_clamp(ib >> 20) << 16);
}

/* The same arithmetic for ONE triplet, with the three values the accumulations start from passed in (VPCC_K_R, _G, _B: the
 * general sequence keeps them in registers).  *fraction_min as in vpcc_colour_luma. */
VPCC_COLOUR_FN uint32_t vpcc_colour_one(uint32_t Y, uint32_t U, uint32_t V, double kr, double kg, double kb, uint32_t* fraction_min) {
  const double y = (double)Y, u = (double)U, v = (double)V;
  const int32_t ir = vpcc_colour_lo(fma(y, VPCC_C_AY, fma(v, VPCC_C_RV, kr)));
  const int32_t ig = vpcc_colour_lo(fma(y, VPCC_C_AY, fma(v, VPCC_C_GV, fma(u, VPCC_C_GU, kg))));
  const int32_t ib = vpcc_colour_lo(fma(y, VPCC_C_AY, fma(u, VPCC_C_BU, kb)));
  const uint32_t fr = (uint32_t)ir & 0xFFFF8u, fg = (uint32_t)ig & 0xFFFF8u, fb = (uint32_t)ib & 0xFFFF8u;
  const uint32_t m = fr < fg ? (fr < fb ? fr : fb) : (fg < fb ? fg : fb);
  if (m < *fraction_min) *fraction_min = m;
  return (uint32_t)vpcc_colour_clamp(ir >> 20) | ((uint32_t)vpcc_colour_clamp(ig >> 20) << 8) |
         ((uint32_t)vpcc_colour_clamp(ib >> 20) << 16);
}

#endif /* VPCC_COLOUR_H_ */
