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

// ... and one whose address is the same in every lane: through the constant address space, i.e. scalar loads into SGPRs (for
// memory no kernel of the same launch writes: the scalar cache is not coherent with vector stores)
#define VPCC_CONST __attribute__((address_space(4)))
template <class T>
__device__ __forceinline__ T cload(const T* p) {
  static_assert(sizeof(T) % 4 == 0, "whole dwords");
  const VPCC_CONST uint32_t* q = (const VPCC_CONST uint32_t*)p;            // (dword by dword: a memcpy from this address space is
  uint32_t w[sizeof(T) / 4];                                               // lowered to vector loads + v_readfirstlane)
#pragma unroll
  for (uint32_t k = 0; k < sizeof(T) / 4; ++k) w[k] = q[k];
  T r;
  __builtin_memcpy(&r, w, sizeof(T));
  return r;
}
// Element at a BYTE offset below 2^32 from a base every lane shares: the address is {scalar base, 32-bit lane offset}, no 64-bit
// vector arithmetic (the general sequence spent a quarter of its vector instructions on v_mad_u64_u32 / v_lshl_add_u64).
template <class T>
__device__ __forceinline__ T ld32(const T* base, uint32_t byte_offset) {
  return *(const VPCC_GLOBAL T*)((const VPCC_GLOBAL unsigned char*)base + byte_offset);
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

// Patch::generate_point, src/decoder.rs:871-888, from the virtual block's record (pu, pv: the pixel's offsets inside the
// block).  Assignment order normal, tangent, bitangent as in the reference; `as u16` truncation.
__device__ __forceinline__ uint32_t normal_coord(const VBlock& b, uint32_t depth) {
  return (b.axes_mode >> 6) == 0 ? depth + b.d1 : (b.d1 > depth ? b.d1 : depth) - depth;
}

__device__ __forceinline__ Pt make_point(const VBlock& b, uint32_t pu, uint32_t pv, uint32_t depth) {
  Pt r;
  r.c[0] = r.c[1] = r.c[2] = 0;
  const uint16_t n = (uint16_t)normal_coord(b, depth);
  const uint16_t t = (uint16_t)(pu * b.lod_x + b.t0);
  const uint16_t bt = (uint16_t)(pv * b.lod_y + b.b0);
  const uint32_t na = b.axes_mode & 3u, ta = (b.axes_mode >> 2) & 3u, ba = (b.axes_mode >> 4) & 3u;
#pragma unroll
  for (uint32_t a = 0; a < 3; ++a) {      // select instead of a runtime-indexed array (no scratch)
    uint16_t val = r.c[a];
    if (na == a) val = n;
    if (ta == a) val = t;
    if (ba == a) val = bt;
    r.c[a] = val;
  }
  return r;
}

// generate_points, src/codec.rs:517-565: D1 point from D0 point / second geometry sample.
__device__ __forceinline__ Pt make_point1(const DevFrame& f, const VBlock& b, uint32_t pu, uint32_t pv, const Pt& p0, uint32_t d1) {
  if (f.absolute_d1) return make_point(b, pu, pv, d1);
  Pt r = p0;
  const uint32_t na = b.axes_mode & 3u;
#pragma unroll
  for (uint32_t a = 0; a < 3; ++a)
    if (na == a) r.c[a] = (b.axes_mode >> 6) == 0 ? (uint16_t)(r.c[a] + d1) : (uint16_t)(r.c[a] - d1);
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

// ... with the samples as the 32-bit values a 16-bit load leaves (no zero extension: saves three instructions), and the three
// constants that the FMAs start from held in registers by the caller (vpcc_colour_keys: the compiler otherwise rebuilds each
// 64-bit constant with two moves in front of every FMA that accumulates into it — six moves per conversion).
struct ColourKeys { double r, g, b; };
__device__ __forceinline__ ColourKeys vpcc_colour_keys() {
  ColourKeys k = {VPCC_K_R, VPCC_K_G, VPCC_K_B};
  asm volatile("" : "+v"(k.r), "+v"(k.g), "+v"(k.b));                       // opaque: not a literal any more
  return k;
}
__device__ __forceinline__ uint32_t yuv10_to_rgb8_packed(uint32_t y, uint32_t u, uint32_t v, const ColourKeys& k) {
  uint32_t fmin = (y | u | v) > 1023u ? 0u : 0xFFFFFFFFu;                   // a sample above 10 bits: the reference formula
  const uint32_t rgb = vpcc_colour_one(y, u, v, k.r, k.g, k.b, &fmin);
  if (fmin == 0u) {                                                         // ... or a fraction field that could hide an exact integer (rare)
    const vpcc_color3 c = yuv10_to_rgb8((uint16_t)y, (uint16_t)u, (uint16_t)v);
    return (uint32_t)c.r | ((uint32_t)c.g << 8) | ((uint32_t)c.b << 16);
  }
  return rgb;
}

// Per-pixel work of the enumeration: returns the number of points (0,1,2) and the points.
struct PixelOut {
  Pt p0, p1;
  uint32_t x, y;
  uint32_t n;
};

__device__ __forceinline__ PixelOut eval_pixel(const DevFrame& f, const VBlock& b, uint32_t pu, uint32_t pv) {
  PixelOut o;
  o.n = 0;
  // patch_to_canvas (src/decoder.rs:841-867) of the block's pixel (pu, pv); host validated: inside the canvas
  const int32_t cux = (int32_t)(b.coef & 3u) - 1, cvx = (int32_t)((b.coef >> 2) & 3u) - 1;
  const int32_t cuy = (int32_t)((b.coef >> 4) & 3u) - 1, cvy = (int32_t)(b.coef >> 6) - 1;
  o.x = (uint32_t)((int32_t)b.x0 + cux * (int32_t)pu + cvx * (int32_t)pv);
  o.y = (uint32_t)((int32_t)b.y0 + cuy * (int32_t)pu + cvy * (int32_t)pv);
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
  o.p0 = make_point(b, pu, pv, d0);
  o.p1 = o.p0;
  uint32_t n = 1;
  if (f.map_count > 1) {                                                           // (uniform)
    o.p1 = make_point1(f, b, pu, pv, o.p0, d1);
    n = same_point(o.p0, o.p1) ? 1u : 2u;                                          // codec.rs:422-427
  }
  o.n = occ ? n : 0u;
  return o;
}

}  // namespace vpcc
