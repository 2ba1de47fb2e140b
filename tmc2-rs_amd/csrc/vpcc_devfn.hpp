// vpcc_devfn.hpp — device-side helpers shared by the gfx950 kernels (wave64).
// Each function cites the reference lines (benclmnt/tmc2-rs) whose arithmetic it reproduces.
#pragma once

#include <hip/hip_runtime.h>

#include "vpcc_colour.h"
#include "vpcc_device.hpp"

namespace vpcc {

// Pointers read from a descriptor are generic ("flat") to the compiler.  flat_* instructions count
// on vmcnt AND lgkmcnt and return out of order, which serialises them against LDS traffic, so every
// plane/output access goes through an explicit global (address space 1) pointer.
#define VPCC_GLOBAL __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ const VPCC_GLOBAL T* gl(const T* p) {
  return (const VPCC_GLOBAL T*)p;
}
template <class T>
__device__ __forceinline__ VPCC_GLOBAL T* glw(T* p) {
  return (VPCC_GLOBAL T*)p;
}

// Whole-record load from global memory (descriptor tables).
template <class T>
__device__ __forceinline__ T gload(const T* p) {
  T r;
  __builtin_memcpy(&r, (const VPCC_GLOBAL void*)p, sizeof(T));
  return r;
}

template <class T>
__device__ __forceinline__ void gstore(T* p, const T& v) {
  __builtin_memcpy((VPCC_GLOBAL void*)p, &v, sizeof(T));
}

// ------------------------------------------------------------------ helpers
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// number of set bits of `mask` below this lane
__device__ __forceinline__ uint32_t mbcnt(uint64_t mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

struct Pt { uint16_t c[3]; };

// Patch::generate_point, src/decoder.rs:871-888.  Assignment order normal, tangent, bitangent
// as in the reference; `as u16` truncation.
__device__ __forceinline__ uint32_t normal_coord(const DevPatch& p, uint32_t depth) {
  return p.projection_mode == 0 ? depth + p.d1 : (p.d1 > depth ? p.d1 : depth) - depth;
}

__device__ __forceinline__ Pt make_point(const DevPatch& p, uint32_t u, uint32_t v, uint32_t depth) {
  Pt r;
  r.c[0] = r.c[1] = r.c[2] = 0;
  const uint16_t n = (uint16_t)normal_coord(p, depth);
  const uint16_t t = (uint16_t)(u * p.lod_x + p.u1);
  const uint16_t b = (uint16_t)(v * p.lod_y + p.v1);
#pragma unroll
  for (int a = 0; a < 3; ++a) {           // select instead of a runtime-indexed array (no scratch)
    uint16_t val = r.c[a];
    if (p.normal_axis == a) val = n;
    if (p.tangent_axis == a) val = t;
    if (p.bitangent_axis == a) val = b;
    r.c[a] = val;
  }
  return r;
}

// generate_points, src/codec.rs:517-565: D1 point from D0 point / second geometry sample.
__device__ __forceinline__ Pt make_point1(const DevFrame& f, const DevPatch& p, uint32_t u, uint32_t v,
                                          const Pt& p0, uint32_t d1) {
  if (f.absolute_d1) return make_point(p, u, v, d1);
  Pt r = p0;
#pragma unroll
  for (int a = 0; a < 3; ++a)
    if (p.normal_axis == a)
      r.c[a] = p.projection_mode == 0 ? (uint16_t)(r.c[a] + d1) : (uint16_t)(r.c[a] - d1);
  return r;
}

__device__ __forceinline__ bool same_point(const Pt& a, const Pt& b) {
  return a.c[0] == b.c[0] && a.c[1] == b.c[1] && a.c[2] == b.c[2];
}

// convert_yuv10_to_rgb8, src/codec.rs:661-687: IEEE f64, source order, no contraction.
__device__ __forceinline__ uint8_t clamp_u8(double x) {
  if (x < 0.) return 0;
  if (x > 255.) return 255;
  return (uint8_t)x;
}

__device__ __forceinline__ vpcc_color3 yuv10_to_rgb8(uint16_t y16, uint16_t u16, uint16_t v16) {
  const double offset = 512., scale = 1023.;
  const double y = (double)y16, u = (double)u16, v = (double)v16;
  const double r = y + 1.57480 * (v - offset);
  const double g = y - 0.18733 * (u - offset) - (0.46813 * (v - offset));
  const double b = y + 1.85563 * (u - offset);
  vpcc_color3 c;
  c.r = clamp_u8(__builtin_floor(r / scale * 255.));
  c.g = clamp_u8(__builtin_floor(g / scale * 255.));
  c.b = clamp_u8(__builtin_floor(b / scale * 255.));
  return c;
}

// Same result as yuv10_to_rgb8, bit for bit, without a division on the common path: vpcc_colour.h
// (f64 FMAs on a 2^-20 grid; checked on the whole 10-bit cube by tests/colour_exhaustive.c).  A fraction
// pattern that could hide an exact integer, or a sample above 10 bits, takes the reference formula.
__device__ __forceinline__ vpcc_color3 yuv10_to_rgb8_fast(uint16_t y16, uint16_t u16, uint16_t v16) {
  uint32_t fmin = ((uint32_t)(y16 | u16 | v16) > 1023u) ? 0u : 0xFFFFFFFFu;
  const uint32_t rgb = vpcc_colour_luma(y16, vpcc_colour_chroma(u16, v16), &fmin);
  if (fmin == 0u) return yuv10_to_rgb8(y16, u16, v16);   // rare
  vpcc_color3 c;
  c.r = (uint8_t)rgb; c.g = (uint8_t)(rgb >> 8); c.b = (uint8_t)(rgb >> 16);
  return c;
}

// Per-pixel work of the enumeration: returns the number of points (0,1,2) and the points.
struct PixelOut {
  Pt p0, p1;
  uint32_t x, y;
  uint32_t n;
};

__device__ __forceinline__ PixelOut eval_pixel(const DevFrame& f, const DevPatch& p, uint32_t u, uint32_t v) {
  PixelOut o;
  o.n = 0;
  const int32_t x = p.ax_u * (int32_t)u + p.ax_v * (int32_t)v + p.cx;   // host validated: inside the canvas
  const int32_t y = p.ay_u * (int32_t)u + p.ay_v * (int32_t)v + p.cy;
  o.x = (uint32_t)x;
  o.y = (uint32_t)y;
  // (a precision that is a power of two — every stream's — by shift: the two 32-bit divisions were a quarter of the
  // general sequence's vector instructions)
  const bool pow2 = (1u << f.prec_shift) == f.prec;
  const uint32_t oy = pow2 ? o.y >> f.prec_shift : o.y / f.prec, ox = pow2 ? o.x >> f.prec_shift : o.x / f.prec;
  // Occupancy and both depths are requested TOGETHER (the pixel lies inside the canvas and the planes cover it: validate_frame),
  // and the pixel's points are built without a branch: with the depths behind `if (occ == 0) return` every occupied pixel paid
  // two dependent round trips to memory, one per plane kind.
  const uint8_t occ = gl(f.occ)[oy * f.occ_stride + ox];                             // src/codec.rs:288-301, 393
  const uint32_t d0 = (uint32_t)(gl(f.geo[0])[o.y * f.geo_stride[0] + o.x] >> 2);  // depth / 4, codec.rs:534
  const uint32_t d1 = (uint32_t)(gl(f.geo[1])[o.y * f.geo_stride[1] + o.x] >> 2);  // (one map: the descriptor's alias of layer 0)
  o.p0 = make_point(p, u, v, d0);
  o.p1 = o.p0;
  uint32_t n = 1;
  if (f.map_count > 1) {                                                           // (uniform)
    o.p1 = make_point1(f, p, u, v, o.p0, d1);
    n = same_point(o.p0, o.p1) ? 1u : 2u;                                          // codec.rs:422-427
  }
  o.n = occ ? n : 0u;
  return o;
}

}  // namespace vpcc
