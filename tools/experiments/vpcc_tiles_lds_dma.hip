// vpcc_tiles_lds_dma.hip — EXPERIMENTAL variant of vpcc_tiles.hip (round 4; `make lds-dma`, never the product): the attribute
// samples of an item do not pass through registers but are staged in LDS by LDS-DMA (global_load_lds_dwordx4), a whole step
// before the item is emitted.  Bit-exact (the GPU parity suite passes on it), 13 % fewer plane reads (1 250 vs 1 440 MB per
// 128 S-longdress frames: at the floor of the raster layout in emission order) — and 7 % SLOWER (0.482 vs 0.451 ms) once the
// memory system is balanced (vpcc_ctx_reserve): DESIGN.md 4.1.2, profiles/r04/ab_kernel.txt, ab_stage.txt.
//
// Single-pass, wave-per-tile reconstruction kernel for gfx950 (CDNA4, wave64).
//
// Production path for the common configuration (block size R = 16, Default/Swap patches, 8-byte
// aligned luma rows).  One launch per batch of frames; every plane is read from HBM once, every
// output byte written once.
//
// Work decomposition
//   item   = one virtual block that owns its canvas block and holds occupancy (k_plan_items, vpcc_kernels.hip), in
//            the reference's emission order (src/codec.rs:352-385);
//   group  = 16 consecutive items = one ticket and one look-back word = the work of one 256-thread
//            workgroup for one step;
//   wave   = items w, 4+w, 8+w, 12+w of the group, one after the other: the four waves work on four
//            CONSECUTIVE items at the same time (horizontal neighbours share every 128-byte line).
//            Lane l reads the 4 CANVAS pixels x0 + 4*(l&3)..+3 of canvas row y0 + (l>>2): 8 contiguous
//            bytes per plane.  For Default patches "lane order, then pixel order inside the lane" is the
//            reference's emission order; for Swap patches (u runs down the canvas column) the per-pixel
//            ranks are transposed through a 16x16 byte matrix in LDS instead of transposing the loads.
//
// Per workgroup: a two-stage pipeline over groups (tickets drawn dynamically per frame, a step ahead; the group being
// counted may already belong to the workgroup's next frame)
//   1. count the NEXT group: occupancy + both geometry layers of its 16 items (the geometry stays in
//      registers until the group is emitted); a D1 point is dropped when it equals the D0 point
//      (src/codec.rs:422-427); barrier; wave 0 publishes the group total;
//   2. every wave obtains the CURRENT group's output offset by decoupled look-back over the earlier
//      groups of the frame (its total was published one step ago; the words were read speculatively
//      behind the count's loads);
//   3. per item (unrolled, the next item's attribute samples prefetched): colour conversion of the lane's 8
//      samples (src/codec.rs:661-687; vpcc_colour.h), then every lane writes one 8-B record
//      {depth, du, dv, layer | r, g, b} per point into the wave's LDS slots at the point's rank
//      (= compaction in emission order); then lane <-> point: back-projection
//      (src/decoder.rs:871-888) and contiguous non-temporal stores (12 + 6 bytes per lane for two points).
//   What bounds it: with the planes really coming from HBM the kernel runs within 3 % of its
//   own memory skeleton at ~4 TB/s of memory-side traffic; the arithmetic (half of it the colour conversion) is hidden.
//   The counter is ONE in-order vmcnt for loads and stores: see "take delivery" below and DESIGN.md 4.1.1.
//
// Cross-workgroup ordering is placement-independent (cdna_hip_programming.md §6 Guideline 16):
// groups are drawn from a per-frame TICKET counter, so a look-back only waits for tickets that
// running workgroups hold; state words are 8-byte {status,value} granules moved with relaxed
// agent-scope atomics.  XCD-aware blockIdx mapping is used for L2 locality only.  Ticket counters and look-back
// words carry the launch generation: nothing is cleared between launches, and any number of workgroups may work on
// a frame — a workgroup whose own frames have run dry helps with the frames that are still open.
#include <hip/hip_runtime.h>
#include <algorithm>

#include <cstdlib>

#include "vpcc_device.hpp"
#include "vpcc_colour.h"
#include "vpcc_devfn.hpp"

namespace vpcc {

namespace {

// Look-back word: {generation:30 | status:2 | value:32}.  The generation is the launch counter of the
// gof: words written by earlier launches read as EMPTY, so nothing has to be cleared between launches.
// A few timing-only ablation switches and in-kernel stamps exist in the DIAGNOSTIC build only (`make diag`:
// -DVPCC_DIAGNOSTIC, libvpcc_recon_diag.so, used by tools/ — never by tests, bench.py or the
// product): in the product build `variant` is the constant 0 and every switch folds away.
#ifdef VPCC_DIAGNOSTIC
constexpr bool kDiagnostic = true;
#else
constexpr bool kDiagnostic = false;
#endif

#define VPCC_CONSTANT __attribute__((address_space(4)))
// The frame descriptors are host-written too: read through the constant address space, every field of a frame
// whose address is wave-uniform is a scalar load (a generic pointer that changes inside a loop would be loaded
// with VECTOR loads, each behind a vmcnt(0) — a wait for the wave's output stores).
typedef const VPCC_CONSTANT DevFrame CFrame;

constexpr uint64_t kStatusShift = 32;
constexpr uint64_t kGenShift = 34;
constexpr uint64_t kAggregate = 1ull << kStatusShift;
constexpr uint64_t kPrefix = 2ull << kStatusShift;
constexpr uint32_t kSpinLimit = 1u << 22;
#ifndef VPCC_FRAMES_IN_FLIGHT
#define VPCC_FRAMES_IN_FLIGHT 16
#endif
constexpr uint32_t kFramesInFlight = VPCC_FRAMES_IN_FLIGHT;    // frames of one XCD label worked on at a time (launches of more than 8 x this many frames run in rounds)

__device__ __forceinline__ uint64_t st_load(const uint64_t* p) {
  return __hip_atomic_load(gl(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_store(uint64_t* p, uint64_t v) {
  __hip_atomic_store(glw(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Ticket counter of a frame: {launch generation : 32 | groups drawn : 32}, drawn with one returning 64-bit add.  A
// value whose generation is not this launch's belongs to an earlier launch: the drawer resets the counter with a
// compare-and-swap and holds ticket 0 (or finds that another workgroup did, and draws again).  Nothing is cleared
// between launches, and ANY number of workgroups may draw from a frame — what lets a workgroup whose own frames have
// run dry help with the frames its XCD still works on.
__device__ __forceinline__ uint64_t ticket_add(uint64_t* w) {
  return __hip_atomic_fetch_add(glw(w), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ uint32_t ticket_settle(uint64_t* w, uint32_t gen, uint64_t drawn, uint32_t* error_flag) {      // one lane
  for (uint32_t tries = 0; (uint32_t)(drawn >> 32) != gen; ++tries) {
    uint64_t cur = __hip_atomic_load(gl(w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((uint32_t)(cur >> 32) == gen) { drawn = ticket_add(w); continue; }
    const uint64_t fresh = ((uint64_t)gen << 32) | 1ull;
    if (__hip_atomic_compare_exchange_strong(glw(w), &cur, fresh, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      return 0u;
    if (tries > kSpinLimit) {                                // never in a healthy run; reported by the host
      atomicOr(error_flag, 1u);
      return 0xFFFFFFFFu;                                    // (reads as "past the end")
    }
  }
  return (uint32_t)drawn;
}

// Exclusive prefix of group `g` within its frame.  One full wave; same result in every lane.
__device__ __forceinline__ uint32_t status_of(uint64_t s, uint32_t gen) {
  return (uint32_t)(s >> kGenShift) == gen ? (uint32_t)(s >> kStatusShift) & 3u : 0u;
}

// Wave-wide inclusive scan / sum with DPP row shifts and row broadcasts (gfx9 DPP controls; no LDS).
template <int kCtrl, int kRowMask>
__device__ __forceinline__ uint32_t dpp_add(uint32_t v) {
  return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, kCtrl, kRowMask, 0xF, kRowMask == 0xF);
}
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
  v = dpp_add<0x111, 0xF>(v);        // row_shr:1   (lanes shifted in from outside the row read 0)
  v = dpp_add<0x112, 0xF>(v);        // row_shr:2
  v = dpp_add<0x114, 0xF>(v);        // row_shr:4
  v = dpp_add<0x118, 0xF>(v);        // row_shr:8   -> scan inside each row of 16
  v = dpp_add<0x142, 0xA>(v);        // row_bcast:15 into rows 1 and 3
  v = dpp_add<0x143, 0xC>(v);        // row_bcast:31 into rows 2 and 3
  return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan(v), 63);
}

// `early` is this lane's word of the first 64 predecessors, read speculatively before the caller's
// count phase (the words of a pipelined group were published a step ago, so it is normally final).
__device__ uint32_t look_back_groups(CFrame& f, uint32_t g, uint32_t gen, uint64_t early) {
  uint32_t excl = 0;
  int32_t idx = (int32_t)g - 1;
  const uint32_t lane = lane_id();
  bool first = true;
  while (idx >= 0) {
    const int32_t my = idx - (int32_t)lane;
    uint64_t s = ((uint64_t)gen << kGenShift) | kPrefix;
    if (my >= 0) {
      uint32_t spins = 0;
      s = first ? early : st_load(f.scan_state + my);
      while (status_of(s, gen) == 0) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > kSpinLimit) {               // never reached in a healthy run; reported by the host
          atomicOr(f.error_flag, 1u);
          s = ((uint64_t)gen << kGenShift) | kPrefix;
          break;
        }
        s = st_load(f.scan_state + my);
      }
    }
    first = false;
    const uint64_t pm = __ballot(status_of(s, gen) == 2);
    const uint32_t firstp = pm ? (uint32_t)__builtin_ctzll(pm) : 64u;
    uint32_t v = lane <= firstp ? (uint32_t)s : 0u;
    v = wave_sum(v);
    excl += v;
    if (pm) break;
    idx -= 64;
  }
  return excl;
}

// Orders this wave's LDS writes before its later LDS reads across lanes (and vice versa).  A wave's
// DS instructions execute in issue order, so no counter wait is needed — only the compiler has to
// keep the accesses on their side of this point.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Workgroup barrier that orders LDS only: the waves of a workgroup hand each other nothing through global memory,
// and __syncthreads() — a fence over ALL address spaces — makes every wave wait for its outstanding output stores
// (s_waitcnt vmcnt(0)) at each step.
__device__ __forceinline__ void wg_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

struct Px4 { uint32_t lo, hi; };   // four u16 samples, pixel j in bits 16*(j&1) of (j<2 ? lo : hi)

template <int J>
__device__ __forceinline__ uint32_t px(const Px4& v) {
  return J == 0 ? (v.lo & 0xFFFFu) : J == 1 ? (v.lo >> 16) : J == 2 ? (v.hi & 0xFFFFu) : (v.hi >> 16);
}

// Scalar (wave-uniform) view of one work item: the eight descriptor dwords live in SGPRs.
struct Item {
  uint32_t x0, y0, patch, flags, axes, tb, bb, d1, lod_x, lod_y, sel_xy, sel_z;
};

// The item tables are written by the host before the launch and never by a kernel: with a
// wave-uniform address the constant address space makes this ONE s_load_dwordx8 — no vector load, no
// readfirstlane, and (unlike a vector load) it does not queue behind the wave's outstanding stores.
// (VPCC_CONSTANT: defined above.)
__device__ __forceinline__ Item load_item(const TileItem* p) {
  const VPCC_CONSTANT uint32_t* q = (const VPCC_CONSTANT uint32_t*)p;
  uint32_t w[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) w[k] = q[k];
  Item it;
  it.x0 = w[0] & 0xFFFFu; it.y0 = w[0] >> 16;
  it.patch = w[1] & 0xFFFFu; it.flags = (w[1] >> 16) & 0xFFu; it.axes = w[1] >> 24;
  it.tb = w[2]; it.bb = w[3]; it.d1 = w[4];
  it.lod_x = w[5] & 0xFFFFu; it.lod_y = w[5] >> 16;
  it.sel_xy = w[6]; it.sel_z = w[7];
  return it;
}

struct Samples {       // one item's geometry samples of the lane's 4 pixels (the attribute samples go to LDS: AttrTiles)
  Px4 g0, g1;          // geometry D0 / D1
  uint32_t occ;        // bit j: pixel j occupied
};

// Cache policy of the streaming accesses: the output STORES are non-temporal (they would otherwise push the
// plane lines that neighbouring items are about to share out of the XCD's 4 MB L2), every load is plain
// (non-temporal loads: 0.19 vs 0.15 ms per 32 S-longdress frames).
// Plane loads take a wave-uniform base and a 32-bit BYTE offset per lane (saddr + voffset addressing:
// no 64-bit address arithmetic in vector registers); tile_planes_aligned keeps planes below 4 GiB.
__device__ __forceinline__ Px4 load4_row(const uint16_t* base, uint32_t byte_off) {   // 8-B aligned by construction
  typedef uint32_t v2 __attribute__((ext_vector_type(2)));
  const v2 v = *(const VPCC_GLOBAL v2*)((const VPCC_GLOBAL unsigned char*)base + byte_off);
  return Px4{v.x, v.y};
}
__device__ __forceinline__ uint32_t load2(const uint16_t* base, uint32_t byte_off) {   // 4-B aligned by construction
  return *(const VPCC_GLOBAL uint32_t*)((const VPCC_GLOBAL unsigned char*)base + byte_off);
}
// Lane l always reads the 4 CANVAS pixels x0 + 4*(l&3) .. +3 of canvas row y0 + (l>>2): 8 contiguous
// bytes per plane, whatever the patch orientation.  For Default patches this is already the
// emission order; for Swap patches (u runs down the canvas column) the ranks are transposed through
// LDS instead of the loads (see pixel_ranks).
__device__ __forceinline__ void lane_origin(const Item& it, uint32_t lane, uint32_t& px0, uint32_t& py0) {
  px0 = it.x0 + 4u * (lane & 3u);
  py0 = it.y0 + (lane >> 2);
}

// Occupancy of the lane's 4 pixels through the low-resolution plane (src/codec.rs:288-301, 393).  The
// lane needs 4 / 2 / 1 consecutive bytes for occupancy_precision 1 / 2 / >= 4 (a power of two); they
// lie inside one aligned dword (tile_planes_aligned), which is what is loaded — an aligned dword
// that contains a valid byte never leaves the allocation's pages.
// Returns the aligned dword; `shift` is the bit position of the lane's first byte inside it.
__device__ __forceinline__ uint32_t load_occupancy_word(CFrame& f, const Item& it, uint32_t lane, uint32_t& shift) {
  uint32_t px0, py0;
  lane_origin(it, lane, px0, py0);
  const uint32_t off = __umul24(py0 >> f.prec_shift, f.occ_stride) + (px0 >> f.prec_shift);
  const uint32_t mis = ((uint32_t)(uintptr_t)f.occ + off) & 3u;            // position inside the aligned dword
  shift = 8u * mis;
  return *reinterpret_cast<const VPCC_GLOBAL uint32_t*>((const VPCC_GLOBAL unsigned char*)f.occ + (off - mis));
}
__device__ __forceinline__ uint32_t load_occupancy_raw(CFrame& f, const Item& it, uint32_t lane) {
  uint32_t shift;
  const uint32_t w = load_occupancy_word(f, it, lane, shift);
  return w >> shift;
}
// bit j: pixel j of the lane is occupied; pixel j reads byte j >> prec_shift (byte 0 for precision >= 4)
__device__ __forceinline__ uint32_t occupancy_bits(CFrame& f, uint32_t raw) {
  if (f.prec_shift >= 2u) return (raw & 0xFFu) ? 0xFu : 0u;     // the usual case (precision 4): one byte, four pixels
  const uint32_t sh = f.prec_shift;
  uint32_t bits = 0;
#pragma unroll
  for (uint32_t j = 0; j < 4; ++j) bits |= (((raw >> (8u * (j >> sh))) & 0xFFu) ? 1u : 0u) << j;
  return bits;
}

// Plane loads are UNCONDITIONAL (no exec-masked region, so loads of several items stay in flight
// together): a lane without an occupied pixel reads the block's first pixels instead and its samples are
// never used.  (Redirecting such lanes to the wave's first occupied lane, so that no unneeded 128-byte line
// is ever touched, changed neither the traffic counters nor the time and costs ~7 instructions per call.)
__device__ __forceinline__ void load_origin(const Item& it, uint32_t lane, uint32_t occ, uint32_t& px0, uint32_t& py0) {
  px0 = occ ? it.x0 + 4u * (lane & 3u) : it.x0;
  py0 = occ ? it.y0 + (lane >> 2) : it.y0;
}
__device__ __forceinline__ void load_geometry(CFrame& f, const Item& it, uint32_t lane, Samples& s) {
  uint32_t px0, py0;
  load_origin(it, lane, s.occ, px0, py0);
  const uint32_t off = (__umul24(py0, f.geo_stride[0]) + px0) * 2u;                // both layers: one video, one row pitch
  s.g0 = load4_row(f.geo[0], off);
  s.g1 = load4_row(f.geo[1], off);                                             // single map: an alias of layer 0
}

// Which D1 points duplicate their D0 point (src/codec.rs:422-427), one bit per pixel of the lane.
//   absolute D1: the two points differ only in the normal coordinate (src/decoder.rs:881-888),
//     depth + d1 (mode 0) or max(d1, depth) - depth = d1 - min(depth, d1) (mode 1): equal exactly when
//     min(depth0, D) == min(depth1, D) with D = d1 in mode 1 and "infinity" in mode 0 (depths are
//     < 2^14, so `as u16` cannot fold two different values) — unless a later assignment overwrites
//     the normal coordinate (degenerate axes, sel never picks the normal), then they are always equal;
//   relative D1: point0[normal_axis] +- d1 as u16 leaves the point unchanged only for d1 == 0.
// Both depths of a dword are handled at once with packed 16-bit operations.
typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_depth_min(uint32_t v, uint32_t dd) {
  u16x2 a = __builtin_bit_cast(u16x2, v);
  a = a >> (u16x2)(2);                                     // depth = sample / 4 (src/codec.rs:534, 548)
  a = __builtin_elementwise_min(a, __builtin_bit_cast(u16x2, dd));
  return __builtin_bit_cast(uint32_t, a);
}
__device__ __forceinline__ uint32_t zero_halves(uint32_t x) {   // bit 0: low half zero, bit 1: high half zero
  return ((x & 0xFFFFu) ? 0u : 1u) | ((x >> 16) ? 0u : 2u);
}
__device__ __forceinline__ bool normal_visible(const Item& it) {
  const uint32_t na = it.axes & 3u;
  return na != ((it.axes >> 2) & 3u) && na != ((it.axes >> 4) & 3u);
}
__device__ __forceinline__ uint32_t classify(CFrame& f, const Item& it, const Samples& s) {
  if (f.map_count < 2) return 0xFu;                      // single map: D0 only
  if (f.absolute_d1) {
    if (!normal_visible(it)) return 0xFu;
    const uint32_t D = (it.flags & kTileMode1) ? (it.d1 < 0xFFFFu ? it.d1 : 0xFFFFu) : 0xFFFFu;
    const uint32_t dd = D | (D << 16);
    return zero_halves(pk_depth_min(s.g0.lo, dd) ^ pk_depth_min(s.g1.lo, dd)) |
           (zero_halves(pk_depth_min(s.g0.hi, dd) ^ pk_depth_min(s.g1.hi, dd)) << 2);
  }
  return zero_halves(pk_depth_min(s.g1.lo, 0xFFFFFFFFu)) | (zero_halves(pk_depth_min(s.g1.hi, 0xFFFFFFFFu)) << 2);
}

// {x | y << 16, z} of a point as the reference builds it (src/decoder.rs:871-888): assignment order
// normal, tangent, bitangent, `as u16` truncation.  The axes are wave-uniform per item, so the
// assignment is two byte permutes with the item's selectors (vpcc_host.cpp).
struct PointConsts {
  uint32_t nmin;           // scalar: d1 in mode 1, 2^32-1 in mode 0
  int32_t nsign;           // scalar: -1 in mode 1, +1 in mode 0
  uint32_t lod_x, lod_y;   // scalar
  uint32_t sel_xy, sel_z;  // scalar byte-permute selectors
  uint32_t su, sv;         // scalar: bit positions of the patch-local offsets du, dv in a record (Default: du = column, dv = row
                           //   of the pixel in its canvas block; Swap: u runs down the canvas column, du = row, dv = column)
  uint32_t v_d1, v_tb, v_bb;   // vector copies of the addends
};
__device__ __forceinline__ PointConsts point_consts(const Item& it) {
  PointConsts c;
  c.nmin = (it.flags & kTileMode1) ? it.d1 : 0xFFFFFFFFu;
  c.nsign = (it.flags & kTileMode1) ? -1 : 1;
  c.lod_x = it.lod_x; c.lod_y = it.lod_y;
  c.sel_xy = it.sel_xy; c.sel_z = it.sel_z;
  c.su = (it.flags & kTileSwap) ? 20u : 16u; c.sv = (it.flags & kTileSwap) ? 16u : 20u;
  c.v_d1 = it.d1; c.v_tb = it.tb; c.v_bb = it.bb;
  c.nmin = __builtin_amdgcn_readfirstlane(c.nmin);            // opaque: one v_min_u32, not min + select on the mode
  asm volatile("" : "+v"(c.v_d1), "+v"(c.v_tb), "+v"(c.v_bb));
  return c;
}
// rx = depth | column << 16 | row << 20 | ... (the low three bytes of a point record)
__device__ __forceinline__ uint2 pack_point(const PointConsts& c, uint32_t rx) {
  const uint32_t depth = rx & 0xFFFFu, du = __builtin_amdgcn_ubfe(rx, c.su, 4u), dv = __builtin_amdgcn_ubfe(rx, c.sv, 4u);
  // normal coordinate: mode 0: depth + d1; mode 1: max(d1, depth) - depth = d1 - min(depth, d1)
  const uint32_t m = __builtin_elementwise_min(depth, c.nmin);               // depth < 2^14
  const uint32_t n = (uint32_t)__mul24((int32_t)m, c.nsign) + c.v_d1;
  const uint32_t t = __umul24(du, c.lod_x) + c.v_tb, b = __umul24(dv, c.lod_y) + c.v_bb;
  const uint32_t nt = __builtin_amdgcn_perm(t, n, 0x05040100u);             // n (low half) | t << 16
  return make_uint2(__builtin_amdgcn_perm(b, nt, c.sel_xy), __builtin_amdgcn_perm(b, nt, c.sel_z));
}

// D1 point in relative mode (src/codec.rs:551-559): point0 with +-d1 on coordinate index normal_axis.
__device__ __forceinline__ uint2 relative_point(uint32_t na, bool mode1, uint2 p0, uint32_t d1) {
  uint32_t c[3] = {p0.x & 0xFFFFu, p0.x >> 16, p0.y & 0xFFFFu};
#pragma unroll
  for (uint32_t a = 0; a < 3; ++a)
    if (na == a) c[a] = (mode1 ? c[a] - d1 : c[a] + d1) & 0xFFFFu;
  return make_uint2(c[0] | (c[1] << 16), c[2]);
}

// ---- attribute tiles in LDS ---------------------------------------------------------------------------------------
// The attribute samples of an item never pass through registers: the 16x16 luma tile of both layers and the 8x8
// chroma tiles of U and V of both layers (1 536 B) are copied from the raster planes straight into LDS by LDS-DMA
// (global_load_lds_dwordx4: a per-lane source address, 16 B per lane, destination = wave-uniform base + 16 * lane),
// for ALL sixteen items of a group in one burst, a whole step before the group is emitted: the four (luma) / eight
// (chroma) horizontally neighbouring blocks that share a 128-byte line ask for it while it is in flight or L2-hot,
// whatever their position in the group — with register loads one item ahead of emission (rounds 1-3) a line shared by
// two consecutive quads of items was fetched twice, 5.7 us apart (900 MB read for 512 MB of attribute samples).
// A wave stages the tiles of its own four items; nobody else reads them, so no barrier is involved: the DMA is issued
// behind the last store loop of a step, the count phase of the next step issues and consumes ordinary loads behind it,
// and the vector-memory counter retires in order — once those loads have arrived the tiles have landed.
// Layout per wave (6 KB), per pair of items 2k, 2k + 1 (3 KB): [luma 2k | chroma 2k | chroma 2k+1 | luma 2k+1] with a luma
// tile [layer][row 0..15][16 samples] (1 KB) and the chroma tiles [U0, V0, U1, V1][row 0..7][8 samples] (512 B): the two
// chroma tiles of a pair are one DMA instruction, and the tiles of ONE item are contiguous (its records overflow into them).
constexpr uint32_t kAttrLumaBytes = 1024u, kAttrChromaBytes = 512u;
constexpr uint32_t kAttrItemBytes = kAttrLumaBytes + kAttrChromaBytes;
constexpr uint32_t kAttrWaveBytes = kTileItemsPerWave * kAttrItemBytes;
__host__ __device__ constexpr uint32_t attr_luma_off(uint32_t i) { return 2u * kAttrItemBytes * (i >> 1) + ((i & 1u) ? 2048u : 0u); }
__host__ __device__ constexpr uint32_t attr_chroma_off(uint32_t i) { return 2u * kAttrItemBytes * (i >> 1) + 1024u + 512u * (i & 1u); }
__host__ __device__ constexpr uint32_t attr_item_off(uint32_t i) { return kAttrItemBytes * i; }   // first byte of item i's tiles
typedef __attribute__((address_space(3))) unsigned char LdsByte;
typedef uint32_t Rec __attribute__((ext_vector_type(2)));          // a point record (below); a plain vector type: usable through LDS pointers
typedef __attribute__((address_space(3))) Rec LdsRec;

// One LDS-DMA instruction: 16 bytes per active lane from `src` to LDS byte address `lds_base` + 16 * lane.  M0 (the
// destination base) is compiler-reserved: saved and restored inside the statement.  The compiler does not count this
// load (cdna_hip_programming.md, inline asm: "no VGPR destination: register-safe"); see above for what orders it.
__device__ __forceinline__ void lds_dma16(const VPCC_GLOBAL unsigned char* src, uint32_t lds_base) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(lds_base) : "memory");
}

// Stages the attribute tiles of the wave's four items of group `g` of frame `f`.  `occ4`: the lane's occupancy
// nibbles of those items (count phase).  A block row (luma) / row pair (chroma) without any occupied pixel is not
// fetched: its lanes read the block's first occupied row instead (their LDS bytes are never looked at) — what the
// redirect of unoccupied lanes did for the register loads.  An item without occupancy (or past the end of the frame)
// is skipped.
__device__ __forceinline__ void stage_attributes(CFrame& f, uint32_t g, uint32_t wave, uint32_t lane, uint32_t occ4,
                                                 uint32_t lds_wave, const uint32_t kItems = 0xFu) {   // bit i: stage the wave's i-th item
  if (!f.has_attr) return;
  // (opaque: everything derived from the lane id below is loop-invariant in the kernel's step loop, and the compiler would
  // keep it in registers — spilled ones — for the whole step instead of recomputing a few shifts here once per step)
  asm volatile("" : "+v"(lane));
  uint32_t x0[4], y0[4], lo[4], hi[4];                     // scalars: block origin, occupied pixel-lanes of the item (64 bits)
#pragma unroll
  for (uint32_t i = 0; i < 4; ++i) {
    const uint32_t idx = g * kTileItemsPerGroup + i * kTileWaves + wave;
    const uint32_t w0 = *(const VPCC_CONSTANT uint32_t*)(f.tiles + (idx < f.n_tiles ? idx : 0u));
    x0[i] = w0 & 0xFFFFu; y0[i] = w0 >> 16;
    const uint64_t m = (idx < f.n_tiles && ((kItems >> i) & 1u)) ? __ballot(((occ4 >> (4u * i)) & 0xFu) != 0u) : 0ull;
    lo[i] = (uint32_t)m; hi[i] = (uint32_t)(m >> 32);
  }
  {
    // luma: lane = layer * 32 + row * 2 + half; 16 bytes = 8 samples of row `row`
    const uint32_t row = (lane >> 1) & 15u, half = lane & 1u;
    // (the plane pointers are read as SCALARS and selected per lane afterwards: a select between two descriptor fields
    // with a per-lane condition becomes a vector load of the descriptor behind a vmcnt(0) — a wait for the stores)
    uint64_t ya = (uint64_t)f.attr_y[0], yb = (uint64_t)f.attr_y[1];
    asm volatile("" : "+s"(ya), "+s"(yb));
    const VPCC_GLOBAL unsigned char* plane = (const VPCC_GLOBAL unsigned char*)((lane >> 5) ? yb : ya);
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i) {
      const uint64_t m = ((uint64_t)hi[i] << 32) | lo[i];
      if (m == 0ull) continue;                                                     // wave-uniform
      const uint32_t first = (uint32_t)__builtin_ctzll(m) >> 2;                    // first occupied row
      const uint32_t r = ((uint32_t)(m >> (4u * row)) & 0xFu) ? row : first;
      const uint32_t off = (__umul24(y0[i] + r, f.attr_stride[0]) + x0[i] + 8u * half) * 2u;
      lds_dma16(plane + off, lds_wave + attr_luma_off(i));
    }
  }
  {
    // chroma: lane = item-of-pair * 32 + plane * 8 + row; 16 bytes = the 8 samples of chroma row `row` of the block
    const uint32_t which = lane >> 5, pl = (lane >> 3) & 3u, row = lane & 7u;
    uint64_t ua = (uint64_t)f.attr_u[0], ub = (uint64_t)f.attr_u[1], va = (uint64_t)f.attr_v[0], vb = (uint64_t)f.attr_v[1];
    asm volatile("" : "+s"(ua), "+s"(ub), "+s"(va), "+s"(vb));
    const uint64_t pu = (pl & 2u) ? ub : ua, pv = (pl & 2u) ? vb : va;
    const VPCC_GLOBAL unsigned char* plane = (const VPCC_GLOBAL unsigned char*)((pl & 1u) ? pv : pu);
#pragma unroll
    for (uint32_t k = 0; k < 2; ++k) {
      const uint64_t ma = ((uint64_t)hi[2 * k] << 32) | lo[2 * k], mb = ((uint64_t)hi[2 * k + 1] << 32) | lo[2 * k + 1];
      if ((ma | mb) == 0ull) continue;                                             // wave-uniform
      const uint32_t fa = ma ? (uint32_t)__builtin_ctzll(ma) >> 3 : 0u, fb = mb ? (uint32_t)__builtin_ctzll(mb) >> 3 : 0u;
      const uint64_t m = which ? mb : ma;
      const uint32_t r = ((uint32_t)(m >> (8u * row)) & 0xFFu) ? row : (which ? fb : fa);
      const uint32_t bx = which ? x0[2 * k + 1] : x0[2 * k], by = which ? y0[2 * k + 1] : y0[2 * k];
      const uint32_t off = (__umul24((by >> 1) + r, f.attr_cstride[0]) + (bx >> 1)) * 2u;
      if (m != 0ull) lds_dma16(plane + off, lds_wave + attr_chroma_off(2u * k));
    }
  }
}

// ---- convert_yuv10_to_rgb8 (src/codec.rs:661-687) of one point --------------------------------------------------
// vpcc_colour.h: f64 FMAs on a 2^-20 grid whose low dword is the fixed-point result — no division, no floor, no float
// compare.  When a fraction pattern could hide an exact integer (2.4e-5 of all triplets; checked on the whole 10-bit
// cube by tests/colour_exhaustive.c) or a sample exceeds 10 bits, the lane evaluates the reference formula itself.
// The conversion runs AFTER the compaction, per emitted point (58 % of the pixel-layers of an occupied block), on the
// samples the point's record leads to in the item's LDS tiles.
__device__ __forceinline__ uint32_t colour_exact(uint32_t Y, uint32_t U, uint32_t V) {
  const vpcc_color3 c = yuv10_to_rgb8((uint16_t)Y, (uint16_t)U, (uint16_t)V);
  return (uint32_t)c.r | ((uint32_t)c.g << 8) | ((uint32_t)c.b << 16);
}
// The lane's four pixels of one layer: pixels 0,1 and 2,3 share a chroma sample (src/decoder.rs:977), so the chroma
// part of each channel is evaluated once per pair.
__device__ __forceinline__ void colours4(const Px4& y, uint32_t u, uint32_t v, uint32_t rgb[4]) {
  const vpcc_chroma_part c01 = vpcc_colour_chroma(u & 0xFFFFu, v & 0xFFFFu);
  const vpcc_chroma_part c23 = vpcc_colour_chroma(u >> 16, v >> 16);
  uint32_t fmin = ((y.lo | y.hi | u | v) & 0xFC00FC00u) ? 0u : 0xFFFFFFFFu;   // a sample wider than 10 bits
  rgb[0] = vpcc_colour_luma(px<0>(y), c01, &fmin);
  rgb[1] = vpcc_colour_luma(px<1>(y), c01, &fmin);
  rgb[2] = vpcc_colour_luma(px<2>(y), c23, &fmin);
  rgb[3] = vpcc_colour_luma(px<3>(y), c23, &fmin);
  const bool amb = fmin == 0u;
  if (amb) {                                                       // rare
    rgb[0] = colour_exact(px<0>(y), u & 0xFFFFu, v & 0xFFFFu);
    rgb[1] = colour_exact(px<1>(y), u & 0xFFFFu, v & 0xFFFFu);
    rgb[2] = colour_exact(px<2>(y), u >> 16, v >> 16);
    rgb[3] = colour_exact(px<3>(y), u >> 16, v >> 16);
  }
}

// 8-B point record:
//   x: depth | c << 16 | layer << 24     (c: the pixel's position in its canvas block, 16 * row + column)
//   y: r | g << 8 | b << 16
// A D1 record always directly follows the D0 record of its pixel (relative D1 reads its depth there).
// Writes are unconditional: an absent point goes to the lane's dump slot.
// Where the records live: the first kRecordsOwn of an item in the wave's own slots (with the 64 dump slots behind
// them), the rest — an item has up to 512 — in the item's ATTRIBUTE TILE, which is dead once the colours of the item's
// pixels have been computed (1 536 B = 192 records): with 24 KB of tiles per workgroup, 512 slots of its own per wave
// would cost the fourth workgroup per CU (160 KB of LDS).
#ifndef VPCC_RECORDS_OWN
#define VPCC_RECORDS_OWN 320
#endif
constexpr uint32_t kRecordsOwn = VPCC_RECORDS_OWN;                    // 512: no record ever goes to the tile
constexpr uint32_t kSlotsPerWave = kRecordsOwn + 64u;
static_assert(kRecordsOwn == 512u || kRecordsOwn + kAttrItemBytes / 8u >= 512u, "an item has up to 512 points");
static_assert(kRecordsOwn * 8u >= 768u, "pixel_ranks' scratch");

// LDS byte address of record `r` (r < 512) or of dump slot kRecordsOwn + lane
struct RecordSpace {
  LdsByte* own;            // the wave's slots
  LdsByte* tile;           // the item's tiles - 8 * kRecordsOwn: record r >= kRecordsOwn lies at tile + 8 r
  __device__ __forceinline__ LdsRec* at(uint32_t r) const {
    if constexpr (kRecordsOwn >= 512u) return (LdsRec*)(own + 8u * r);
    return (LdsRec*)((r < kRecordsOwn ? own : tile) + 8u * r);
  }
  __device__ __forceinline__ LdsRec* dump(uint32_t lane) const { return (LdsRec*)(own + 8u * (kRecordsOwn + lane)); }
};

template <int J>
__device__ __forceinline__ void put_records(const Samples& s, uint32_t dup, uint32_t pixel, uint32_t rank, uint32_t lane,
                                            const uint32_t rgb0[4], const uint32_t rgb1[4], const RecordSpace& rs) {
  const bool occ = (s.occ >> J) & 1u, second = occ && !((dup >> J) & 1u);
  // depth = sample / 4 (src/codec.rs:534, 548): bits 2..15 of the pixel's half of the dword
  const uint32_t d0 = __builtin_amdgcn_ubfe(J < 2 ? s.g0.lo : s.g0.hi, 2u + 16u * (J & 1), 14u);
  const uint32_t d1 = __builtin_amdgcn_ubfe(J < 2 ? s.g1.lo : s.g1.hi, 2u + 16u * (J & 1), 14u);
  *(occ ? rs.at(rank) : rs.dump(lane)) = Rec{d0 | (pixel << 16), rgb0[J]};
  *(second ? rs.at(rank + 1u) : rs.dump(lane)) = Rec{d1 | (pixel << 16) | (1u << 24), rgb1[J]};
}

// Ranks of the lane's 4 pixels inside the item, emission order (src/codec.rs:382-385: v1 outer, u1 inner).
//   Default patch: u runs along the canvas row, so lane order then pixel order is the emission order.
//   Swap patch   : u runs down the canvas column.  The per-pixel counts go through a 16x16 byte matrix
//                  in LDS (the wave's own, not yet used slot area): every lane re-reads the counts of
//                  4 pixels that are consecutive in COLUMN-major order, scans, and scatters the ranks
//                  back to the row-major owners.
__device__ __forceinline__ void pixel_ranks(const Item& it, const Samples& s, uint32_t dup, uint32_t cnt,
                                            uint32_t lane, unsigned char* scratch, uint32_t rk[4]) {
  uint32_t c[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) c[j] = ((s.occ >> j) & 1u) ? (((dup >> j) & 1u) ? 1u : 2u) : 0u;
  if (!(it.flags & kTileSwap)) {
    const uint32_t first = wave_inclusive_scan(cnt) - cnt;
    rk[0] = first; rk[1] = rk[0] + c[0]; rk[2] = rk[1] + c[1]; rk[3] = rk[2] + c[2];
    return;
  }
  uint32_t* m32 = reinterpret_cast<uint32_t*>(scratch);                   // counts M[y][x], one byte each
  uint16_t* r16 = reinterpret_cast<uint16_t*>(scratch + 256);             // ranks  R[y][x], u16 each
  m32[lane] = c[0] | (c[1] << 8) | (c[2] << 16) | (c[3] << 24);            // row y = lane>>2, x = 4*(lane&3)..+3
  wave_sync();
  const uint32_t x = lane >> 2, yq = 4u * (lane & 3u);                     // column-major chunk: x, y = yq..yq+3
  uint32_t t[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) t[j] = scratch[(yq + j) * 16u + x];
  const uint32_t sum = t[0] + t[1] + t[2] + t[3];
  uint32_t r = wave_inclusive_scan(sum) - sum;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    r16[(yq + j) * 16u + x] = (uint16_t)r;
    r += t[j];
  }
  wave_sync();
  const uint2 back = *reinterpret_cast<const uint2*>(r16 + 4u * lane);    // R[lane>>2][4*(lane&3) .. +3]
  rk[0] = back.x & 0xFFFFu; rk[1] = back.x >> 16; rk[2] = back.y & 0xFFFFu; rk[3] = back.y >> 16;
  wave_sync();                                                             // scratch is overwritten by records next
}

// Diagnostic build only (variant bit 64): per-phase cycle sums of waves 0 and 8 of every group.
// Never read by the kernel; fetched with vpcc_debug_read_stamps().
#ifdef VPCC_DIAGNOSTIC
__device__ unsigned long long g_stamps[16];
// Variant bit 8192: {start, first ticket past the end seen, exit} of every workgroup (s_memtime), for the occupancy-over-time curve
// of a launch (tools/wg_timeline.py).
__device__ unsigned long long g_wg_times[8192][3];
#endif

__device__ __forceinline__ unsigned long long stamp_time() {      // constant 100 MHz clock, comparable across CUs
  unsigned long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define VPCC_STAMP(slot)                                                                  \
  if constexpr (kStamps) {                                                                  \
    const unsigned long long now_ = stamp();                                                \
    t_acc[slot] += now_ - t_prev;                                                            \
    t_prev = now_;                                                                          \
  }
#ifdef VPCC_DIAGNOSTIC
#define VPCC_STAMP_FLUSH()                                                                \
  if ((variant & 64u) && lane == 0) {                           \
    for (int q_ = 0; q_ < 10; ++q_) atomicAdd(&g_stamps[q_], t_acc[q_]);                    \
    atomicAdd(&g_stamps[15], 1ull);                                                         \
  }
#else
#define VPCC_STAMP_FLUSH()
#endif

typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef u32x3 u32x3_a2 __attribute__((aligned(2)));   // point pairs start at (base + k) * 6: 2-byte aligned only
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32_a2 __attribute__((aligned(2)));
typedef uint16_t u16_a2 __attribute__((aligned(2)));

template <class T>
__device__ __forceinline__ void out_store(VPCC_GLOBAL unsigned char* p, T v) {
#ifdef VPCC_PLAIN_STORES
  *(VPCC_GLOBAL T*)p = v;
#else
  __builtin_nontemporal_store(v, (VPCC_GLOBAL T*)p);
#endif
}

// Two points' 12 B at a 32-bit byte offset from a uniform base: consecutive lanes write consecutive pairs, a wave
// covers 768 contiguous bytes.
__device__ __forceinline__ void store_xyz2(VPCC_GLOBAL unsigned char* base, uint32_t byte_off, uint2 p, uint2 q) {
  u32x3 o;
  o.x = p.x; o.y = (p.y & 0xFFFFu) | (q.x << 16); o.z = (q.x >> 16) | (q.y << 16);
  out_store<u32x3_a2>(base + byte_off, o);
}

}  // namespace

// `variant_arg`: ignored by the product build.  Diagnostic build (VPCC_TILES_VARIANT): timing/traffic-only
// ablation bits: 1 skip the look-back wait (the groups' outputs spread over the frame's arrays), 8 skip the colour
// conversion, 32 skip the output stores, 64 in-kernel stamps, 256 no attribute loads, 512 no geometry loads,
// 8192 start / exit time of every workgroup.  Outputs of an ablated run are wrong by construction.
//
// Every workgroup is a short pipeline over groups: it draws a ticket, counts that
// group and publishes the group total BEFORE it looks back for and emits the group it counted one
// step earlier.  A group's look-back therefore happens a whole count phase (or a whole step) after
// its total was published, when the totals of the earlier tickets have long arrived: the wait that
// cost a quarter of the kernel in the count -> look back -> emit form is gone, and the plane loads
// of the next group overlap the stores of the current one across the waves of a CU.
// The geometry the count phase loaded stays in registers until the group is emitted one step later (16 VGPRs per
// group in flight): every plane byte is requested once.
// Register budget: 4 waves per SIMD.  Measured (tools/ab.sh): 3 waves 0.139, 4 waves 0.126, 5 waves (spills)
// 0.143 ms.
// Static wave priorities (s_setprio) for the phases that feed the memory pipeline or that other workgroups wait
// for: the store loop, count + publication of the group total, the issue of the next item's loads.  Each is
// worth 0.7-1.4 % (tools/ab.sh), together 2.5 %; holding the priority through the look-back as well costs 5 %.

// One item from geometry samples in registers and attribute tiles in LDS to its points in HBM: ranks, colours of the
// lane's pixels, compaction of the 8-B records through LDS, then lane <-> point (back-projection, contiguous stores at
// `base`).  `n` = the item's point count (wave-uniform, from the count phase), `dup` its duplicate nibble.
template <bool kStamps = false, class Hook>
__device__ __forceinline__ void emit_item(CFrame& f, const Item& it, const Samples& cur, uint32_t dup, uint32_t n,
                                          uint32_t base, uint32_t lane, const RecordSpace& rs, const LdsByte* luma,
                                          const LdsByte* chroma, uint32_t variant, Hook before_stores,
                                          unsigned long long* t_acc = nullptr) {
  [[maybe_unused]] unsigned long long t_prev = kStamps ? stamp() : 0ull;
  VPCC_GLOBAL unsigned char* const gx = (VPCC_GLOBAL unsigned char*)f.out_xyz;
  VPCC_GLOBAL unsigned char* const gc = (VPCC_GLOBAL unsigned char*)f.out_rgb;
  VPCC_GLOBAL uint16_t* const gp = glw(f.out_patch);
  typedef const LdsRec LdsU2;
  typedef __attribute__((address_space(3))) const uint32_t LdsU1;
  // An empty item (n == 0, wave-uniform) skips both halves.  The hook sits on the joined path between them, not
  // in either branch: the compiler's branch lowering leaves bypass edges that never run ("exec == 0") around
  // conditional code, and a delivery that such an edge skips counts as missing — the price is a vmcnt(0), a wait
  // for this wave's output stores, wherever the registers are next written (the loop latch: once per step).
  if (n != 0) {
    uint32_t rk[4];
    const uint32_t cnt = 2u * (uint32_t)__builtin_popcount(cur.occ) - (uint32_t)__builtin_popcount(dup);
    pixel_ranks(it, cur, dup, cnt, lane, (unsigned char*)rs.own, rk);
    uint32_t rgb0[4] = {0, 0, 0, 0}, rgb1[4] = {0, 0, 0, 0};
    if (f.has_attr && !(variant & 8u)) {
      // the lane's samples from the item's tiles: pixels 4 lane .. + 3 of the block (8 B per layer), and the chroma
      // samples of row (lane >> 2) / 2, columns 2 (lane & 3), + 1 (4 B per plane)
      const Rec ya = *(LdsU2*)(luma + 8u * lane), yb = *(LdsU2*)(luma + 512u + 8u * lane);
      const uint32_t co = (lane >> 3) * 16u + (lane & 3u) * 4u;
      const uint32_t u0 = *(LdsU1*)(chroma + co), v0 = *(LdsU1*)(chroma + co + 128u);
      const uint32_t u1 = *(LdsU1*)(chroma + co + 256u), v1 = *(LdsU1*)(chroma + co + 384u);
      colours4(Px4{ya.x, ya.y}, u0, v0, rgb0);
      if (f.map_count > 1) colours4(Px4{yb.x, yb.y}, u1, v1, rgb1);
    }
    const uint32_t pix0 = 4u * lane;                  // 16 * row + column of the lane's first pixel in the canvas block
    // (every lane has its samples: from here on the tiles may be overwritten — records beyond kRecordsOwn go there)
    put_records<0>(cur, dup, pix0, rk[0], lane, rgb0, rgb1, rs);
    put_records<1>(cur, dup, pix0 + 1u, rk[1], lane, rgb0, rgb1, rs);
    put_records<2>(cur, dup, pix0 + 2u, rk[2], lane, rgb0, rgb1, rs);
    put_records<3>(cur, dup, pix0 + 3u, rk[3], lane, rgb0, rgb1, rs);
    wave_sync();                                      // records written by other lanes are read below
  }
  VPCC_STAMP(7)
  before_stores();
  VPCC_STAMP(8)
  if (n == 0) return;
#ifndef VPCC_NO_STORE_PRIO
  __builtin_amdgcn_s_setprio(1);                    // the store loop feeds the memory pipeline: issue it ahead of arithmetic waves
#endif

  const uint32_t room = base < f.capacity ? f.capacity - base : 0u;   // never write past the caller's arrays
  const uint32_t nw = n < room ? n : room;
  // Per-item constants of the back-projection (src/decoder.rs:871-888).  The addends live in VGPRs (a VOP3
  // takes one scalar operand: with two, the compiler re-materialises one as v_mov inside the loop), the
  // mode-1 clamp is a plain unsigned min against d1 (mode 0: against 2^32-1).
  const PointConsts pc = point_consts(it);
  // The item's points B .. E-1 (absolute indices in the frame) leave in two parts:
  //   bulk  : the whole QUADS of points, [b4, e4) with b4 = B rounded up and e4 = E rounded down to a multiple of four.
  //           Two consecutive points per lane: one 12-B xyz store per lane, one 12-B rgb store per even lane for four
  //           points — never a partial pair or quad, hence no edge cases and exactly two store instructions per trip.
  //           The lane <-> point map is aligned to the ABSOLUTE point index (lane l of trip t: points abs0 + 128 t + 2 l, +1
  //           with abs0 a multiple of 128), so that every store instruction between the first and the last covers whole
  //           128-byte lines (128 points = 768 B of positions = 384 B of colours = 256 B of partition entries) and
  //           every dwordx3 is dword-aligned.
  //   edges : the up to three points in front of b4 and the up to three behind e4 — at most 54 bytes — with ONE byte-store
  //           instruction, lane <-> byte (lanes 0..26: head, 32..58: tail).
  const uint32_t B = base, E = base + nw;
  const uint32_t b4 = (B + 3u) & ~3u, e4 = E & ~3u;
  auto point_of = [&](uint32_t r, uint32_t rx) -> uint2 {
    uint2 p = pack_point(pc, rx);
    if (!f.absolute_d1 && (rx >> 24) != 0)                 // relative D1: the D0 record of the pixel precedes it
      p = relative_point(it.axes & 3u, it.flags & kTileMode1, pack_point(pc, (rs.at(r - 1u)->x & 0xFFFFu) | (rx & 0xFF0000u)), rx & 0xFFFFu);
    return p;
  };
  if (b4 < e4)
    for (uint32_t a = (b4 & ~127u) + 2u * lane; a < e4; a += 128u) {
      if (a < b4) continue;                                    // (first trip: lanes in front of the bulk; whole lane pairs)
      const uint32_t k = a - B;                                // record index of the lane's first point
      const Rec r0 = *rs.at(k), r1 = *rs.at(k + 1u);
      uint2 p0 = pack_point(pc, r0.x), p1 = pack_point(pc, r1.x);
      if (!f.absolute_d1) {                                    // wave-uniform
        if ((r0.x >> 24) != 0)
          p0 = relative_point(it.axes & 3u, it.flags & kTileMode1, pack_point(pc, (rs.at(k - 1u)->x & 0xFFFFu) | (r0.x & 0xFF0000u)), r0.x & 0xFFFFu);
        if ((r1.x >> 24) != 0)
          p1 = relative_point(it.axes & 3u, it.flags & kTileMode1, pack_point(pc, (r0.x & 0xFFFFu) | (r1.x & 0xFF0000u)), r1.x & 0xFFFFu);
      }
      if ((variant & 32u) && p0.x != 0xFFFFFFFEu) continue;               // ablation: all the arithmetic, no stores
      const uint32_t c2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r0.y, 0xF5, 0xF, 0xF, false);   // quad_perm:[1,1,3,3]
      const uint32_t c3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r1.y, 0xF5, 0xF, 0xF, false);
      store_xyz2(gx, a * 6u, p0, p1);
      if (gp) out_store<uint32_t>((VPCC_GLOBAL unsigned char*)gp + a * 2u, it.patch | (it.patch << 16));   // partition, codec.rs:452
      if (f.has_attr && !(lane & 1u)) {                        // an even lane: its own and its odd neighbour's colours
        u32x3 o;
        o.x = (r0.y & 0xFFFFFFu) | (r1.y << 24); o.y = ((r1.y >> 8) & 0xFFFFu) | (c2 << 16); o.z = ((c2 >> 16) & 0xFFu) | (c3 << 8);
        out_store<u32x3>(gc + a * 3u, o);
      }
    }
  {
    const uint32_t t0 = e4 > b4 ? e4 : b4;                     // the first point behind the bulk
    const uint32_t hn = (b4 < E ? b4 : E) - B, tn = E > t0 ? E - t0 : 0u;      // head / tail points, 0..3 each
    const uint32_t j = lane & 31u, pt = (j * 57u) >> 9, byte = j - 9u * pt;    // pt = j / 9 for j < 64
    const bool tail = lane >= 32u;
    const uint32_t ap = (tail ? t0 : B) + pt;                  // the lane's point, absolute
    const bool on = pt < (tail ? tn : hn) && (byte < 6u || f.has_attr) && !(variant & 32u);
    if (on) {
      const Rec rec = *rs.at(ap - B);
      const uint2 p = point_of(ap - B, rec.x);
      const uint32_t word = byte < 4u ? p.x : byte < 6u ? p.y : rec.y;
      const uint32_t sh = 8u * (byte < 4u ? byte : byte < 6u ? byte - 4u : byte - 6u);
      VPCC_GLOBAL unsigned char* dst = byte < 6u ? gx + (size_t)ap * 6u + byte : gc + (size_t)ap * 3u + (byte - 6u);
      out_store<uint8_t>(dst, (uint8_t)(word >> sh));
    }
    if (gp && (lane & 31u) < 3u && (lane & 31u) < (tail ? tn : hn) && !(variant & 32u))
      gp[(tail ? t0 : B) + (lane & 31u)] = (uint16_t)it.patch;
  }
  __builtin_amdgcn_s_setprio(0);
  wave_sync();                                      // the next item overwrites the slots
  VPCC_STAMP(9)
}

// Which item of its group a wave handles in its i-th turn: the four waves take four CONSECUTIVE items at the
// same time — in a Default patch these are horizontal neighbours that share every 128-byte line (a 16-pixel
// row is 32 B), so the line is requested once while it is in flight or L1/L2-hot instead of four times, 6 us
// apart (measured: profiles/r02).
__device__ __forceinline__ uint32_t item_in_group(uint32_t wave, uint32_t i) { return i * kTileWaves + wave; }

template <bool kStamps>
#ifndef VPCC_TILE_WAVES_PER_SIMD
#define VPCC_TILE_WAVES_PER_SIMD 4
#endif
__global__ __launch_bounds__(64 * kTileWaves) __attribute__((amdgpu_waves_per_eu(VPCC_TILE_WAVES_PER_SIMD, VPCC_TILE_WAVES_PER_SIMD)))
void k_recon_tiles(const DevFrame* __restrict__ frames, uint32_t first,
                                                     uint32_t count, uint32_t gen,
                                                     uint32_t variant_arg, const TileLaunchMap map) {
  const uint32_t variant = kDiagnostic ? variant_arg : 0u;
  // XCD-aware placement (speed only): ids equal mod 8 share an XCD/L2; a frame stays on one label.
  // The frames of one label are interleaved, so the workgroups of ONE frame start a few slots apart.
  // Launches of more than 128 frames run in ROUNDS: an XCD works on kFramesInFlight frames of its label at a time
  // (an equal share of its resident workgroups each); a workgroup whose ticket lies past the end of its frame goes on
  // to the frame kFramesInFlight places further down its label at once — its pipeline carries on across the change:
  // it counts the new frame's group while it emits the last group it holds of the old one.  Nobody waits for a
  // round to end.  Fewer frames in flight shared the XCD's 4 MB L2 better while planes were read in raster layout
  // (lines shared between neighbouring blocks), but more workgroups per frame mean longer look-back chains and more
  // changes of frame.  Measured on 128-frame launches: raster planes as hipMalloc places them 4 / 8 / 16 frames in
  // flight 0.5195 / 0.5155 / 0.5016 ms; raster planes on placed blocks 0.459 / 0.4531 / 0.4537 (and 91 MB fewer
  // reads with 8); TILED planes (a block's lines are its own) 0.4052 / 0.3989 / 0.3970, S-owlii 0.9546 (8) vs 0.9417
  // (16), reads all but equal (profiles/r03/ab_frames_in_flight*.txt, ab_tiled.txt): 16.
  const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
  const uint32_t frame_groups = (count + 7u) / 8u;
  const bool rounds = frame_groups > kFramesInFlight;
  uint32_t label_frame = rounds ? slot % kFramesInFlight : slot % frame_groups;   // (spreading a frame over all XCDs instead: 0.147 -> 0.164 ms)
  if (!rounds && map.slots) {                              // shares in proportion to the frames' sizes (TileLaunchMap)
    if (slot >= map.slots) return;
    const uint32_t v = map.frame_of_slot[xcd][slot];
    if (v == 0xFFu) return;
    label_frame = v;
  }
#ifdef VPCC_DIAGNOSTIC
  const unsigned long long wg_t0 = (variant & 8192u) ? stamp_time() : 0ull;
  unsigned long long wg_steps = 0;
#endif

  __shared__ uint32_t s_group, s_frame;
  __shared__ uint32_t s_tot[2][kTileItemsPerGroup];                     // point counts of the two groups in flight
  __shared__ __attribute__((aligned(16))) uint2 s_slots[kTileWaves][kSlotsPerWave];
  __shared__ __attribute__((aligned(16))) unsigned char s_attr[kTileWaves][kAttrWaveBytes];   // attribute tiles (stage_attributes)

  [[maybe_unused]] unsigned long long t_prev = kStamps ? stamp() : 0ull;
  [[maybe_unused]] unsigned long long t_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = lane_id();
  constexpr uint32_t K = kTileItemsPerWave;
  static_assert(K == 4, "one occupancy / duplicate nibble per item; four items of resident geometry");

  LdsByte* const slots = (LdsByte*)s_slots[wave];
  const LdsByte* const attr = (const LdsByte*)s_attr[wave];
  // the LDS byte address of the wave's tiles, for the DMA's M0 (an LDS pointer is its 32-bit offset)
  const uint32_t attr_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(LdsByte*)s_attr[wave]);

  // ---- cross-frame pipeline: the workgroup's "next" group may belong to a later frame than its "current" one
  if (xcd + 8u * label_frame >= count) return;
  CFrame* fn = (CFrame*)frames + (first + xcd + 8u * label_frame);     // frame of the group being counted
  CFrame* fc = fn;                                             // frame of the group being emitted
  uint32_t n_groups_next = (fn->n_tiles + kTileItemsPerGroup - 1u) / kTileItemsPerGroup, n_groups_cur = n_groups_next;
  uint32_t g_cur = 0, occ_cur = 0, dup_cur = 0, total_cur = 0, cb = 0;
  bool have_cur = false;
  // gn*: geometry of the group just counted (next), gc*: of the group being emitted (current)
  Px4 gn0[4] = {}, gn1[4] = {}, gc0[4] = {}, gc1[4] = {};
  // Once its own frames have run dry a workgroup HELPS: it draws from the frames of its XCD label that still have
  // groups to hand out (the last frames of the other teams, which end up to 12 % apart: frames differ in size).
  bool helping = false;
  uint32_t help_label = 0;                                   // offset of the label being helped from the workgroup's own
  uint64_t t_ahead = 0;
  if (threadIdx.x == 0) t_ahead = ticket_add(fn->ticket);
  asm volatile("" : "+v"(t_ahead));                            // delivered on every path into the loop: no wait at its top
  for (;;) {
    // ---- 1. the NEXT group.  Its ticket was drawn a step earlier (the round trip of the atomic is hidden behind the
    // previous group's item loop).  A workgroup whose ticket is past the end of its frame goes on to another frame AT
    // ONCE — the next one of its team, then any frame of its label that is not exhausted — and counts that frame's
    // group while it emits the last group it holds of the frame it leaves: a change of frame costs exposed ticket
    // round trips, not a count-only step.
    if (threadIdx.x == 0) s_group = ticket_settle(fn->ticket, gen, t_ahead, fn->error_flag);
    wg_sync_lds();
    uint32_t g_next = __builtin_amdgcn_readfirstlane(s_group);
    bool have_next = true;
    while (g_next >= n_groups_next) {
      wg_sync_lds();                                         // every wave has read s_group
      if (!helping) {
        label_frame += kFramesInFlight;
        if (!rounds || xcd + 8u * label_frame >= count) helping = true;
      }
      if (wave == 0) {
        uint32_t pick = helping ? 0xFFFFFFFFu : label_frame, pick_label = help_label;
        if (helping) {
          // Lane j looks at frame j of a label: exhausted = drawn from in this launch, and past its last group.  The
          // workgroup's own label first (its lines are in this XCD's L2), then the other XCDs' — they end up to 4 %
          // apart, and at the very end of a launch locality matters less than an idle CU.
          for (uint32_t tries = 0; tries < 8u && pick == 0xFFFFFFFFu; ++tries) {
            const uint32_t fj = ((xcd + pick_label) & 7u) + 8u * lane;
            bool open = false;
            if (lane < frame_groups && fj < count) {
              CFrame* fq = (CFrame*)frames + (first + fj);
              const uint64_t w = __hip_atomic_load(gl(fq->ticket), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              const uint32_t ng = (fq->n_tiles + kTileItemsPerGroup - 1u) / kTileItemsPerGroup;
              open = (uint32_t)(w >> 32) != gen || (uint32_t)w < ng;
            }
            const uint64_t m = __ballot(open);
            if (m) {                                           // the first open frame behind the one just left (wrapping)
              const uint32_t from = (label_frame + 1u) & 63u;
              const uint64_t hi = from ? m >> from << from : m;
              pick = (uint32_t)__builtin_ctzll(hi ? hi : m);
            } else {
              pick_label = (pick_label + 1u) & 7u;
              if (pick_label == 0u) break;                     // every label looked at: nothing left in the launch
            }
          }
        }
        if (lane == 0) {
          uint32_t t = 0;
          if (pick != 0xFFFFFFFFu) {
            CFrame* fq = (CFrame*)frames + (first + ((xcd + pick_label) & 7u) + 8u * pick);
            t = ticket_settle(fq->ticket, gen, ticket_add(fq->ticket), fq->error_flag);
          }
          s_group = t; s_frame = pick == 0xFFFFFFFFu ? pick : pick | (pick_label << 28);
        }
      }
      wg_sync_lds();
      const uint32_t pick = __builtin_amdgcn_readfirstlane(s_frame);
      if (pick == 0xFFFFFFFFu) { have_next = false; break; }
      label_frame = pick & 0x0FFFFFFFu;
      help_label = pick >> 28;
      fn = (CFrame*)frames + (first + ((xcd + help_label) & 7u) + 8u * label_frame);
      n_groups_next = (fn->n_tiles + kTileItemsPerGroup - 1u) / kTileItemsPerGroup;
      g_next = __builtin_amdgcn_readfirstlane(s_group);
    }
    // Speculative read of the current group's look-back words.  It is issued BEHIND the count phase's plane
    // loads: vmcnt retires in order, and this coherent load is the slowest of them.
    uint64_t early = 0;
    auto read_early = [&]() {
      // A wavefront-scope (i.e. plain, cacheable) load — may be served by this XCD's L2, even a stale L1 line: a look-back word only moves
      // EMPTY -> AGGREGATE -> PREFIX within a launch and carries the launch generation, so any older state
      // is safe to act on — a stale EMPTY just sends the lane to the coherent re-read in look_back_groups.
      if (have_cur && lane < g_cur)
        early = __hip_atomic_load(gl(fc->scan_state + (g_cur - 1u - lane)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    };
    VPCC_STAMP(0)

    // ---- 2. count it: occupancy + geometry of the wave's 4 items, all loads of a kind issued together.
    // The geometry stays in registers (gn0/gn1) until the group is emitted in the next step; kept besides are the
    // occupancy nibbles (occ_next) and the duplicate nibbles (dup_next), one nibble per item.
    uint32_t occ_next = 0, dup_next = 0;
    __builtin_amdgcn_s_setprio(2);                          // count + publish: other workgroups' look-backs wait for it
    if (have_next) {
      CFrame& f = *fn;
      uint32_t idx4[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) idx4[i] = g_next * kTileItemsPerGroup + item_in_group(wave, i);
      Item it4[4];
      Samples s4[4];
      uint32_t raw[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) it4[i] = load_item(f.tiles + (idx4[i] < f.n_tiles ? idx4[i] : 0u));
      uint32_t raw_shift[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) raw[i] = load_occupancy_word(f, it4[i], lane, raw_shift[i]);
      // complete on EVERY path (a group's items past the end of the frame never look at theirs): a load the
      // compiler believes pending at the top of the next step costs a vmcnt(0) there — a wait for the stores
      asm volatile("" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]));
#pragma unroll
      for (int i = 0; i < 4; ++i) raw[i] >>= raw_shift[i];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        s4[i].occ = idx4[i] < f.n_tiles ? occupancy_bits(f, raw[i]) : 0u;   // past the end: an empty copy of item 0
        if (variant & 512u) { s4[i].g0 = Px4{lane, 0u}; s4[i].g1 = Px4{0u, lane}; }
        else load_geometry(f, it4[i], lane, s4[i]);
      }
      read_early();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t dup = classify(f, it4[i], s4[i]) & s4[i].occ;
        gn0[i] = s4[i].g0; gn1[i] = s4[i].g1;
        // classify() does not touch the samples on every path (single map, degenerate axes): make the
        // loads complete on ALL paths, or the emit loop has to wait for vmcnt(0) — its own stores — at
        // the first move of these registers
        asm volatile("" : "+v"(gn0[i].lo), "+v"(gn0[i].hi), "+v"(gn1[i].lo), "+v"(gn1[i].hi));
        occ_next |= s4[i].occ << (4u * i);
        dup_next |= dup << (4u * i);
        const uint32_t cnt = wave_sum(2u * (uint32_t)__builtin_popcount(s4[i].occ) - (uint32_t)__builtin_popcount(dup));
        if (lane == 0) s_tot[cb ^ 1u][item_in_group(wave, i)] = cnt;
      }
    }
    if (!have_next) read_early();
    VPCC_STAMP(1)
    wg_sync_lds();
    VPCC_STAMP(2)

    // ---- 3. wave 0 publishes the next group's total.  Then EVERY wave looks back for the current group
    // (whose total has been public for a whole step: the words were read before the count phase), so no
    // wave waits for another one here; wave 0 also publishes the group's inclusive prefix.
    uint32_t total_next = 0;
    if (wave == 0 && have_next) {
      total_next = lane < kTileItemsPerGroup ? s_tot[cb ^ 1u][lane] : 0u;   // kTileItemsPerGroup <= 64
      total_next = wave_sum(total_next);
      if (lane == 0)
        st_store(fn->scan_state + g_next, ((uint64_t)gen << kGenShift) | (g_next == 0 ? kPrefix : kAggregate) | total_next);
    }
    if (threadIdx.x == 0 && have_next)                       // a workgroup that saw the end of its last frame draws no more
      t_ahead = ticket_add(fn->ticket);
    VPCC_STAMP(3)
    __builtin_amdgcn_s_setprio(0);

    // The speculative read is complete on EVERY path from here on (group 0 and a workgroup's first step never look
    // at it; all paths, the exit included, meet again in the loop's single latch): a load that the compiler
    // believes pending there costs a vmcnt(0) — a wait for the output stores — at the top of every step.
    asm volatile("" : "+v"(early));
    if (have_cur) {
      CFrame& f = *fc;
      // The look-back words were read before the count phase, whose loads have all been consumed: taking
      // delivery of them here waits for nothing.
      uint32_t excl = (variant & 1u) ? g_cur * (f.capacity / n_groups_cur) : 0u;   // ablation: no wait, outputs still spread over the frame
      if (g_cur != 0 && !(variant & 1u)) excl = look_back_groups(f, g_cur, gen, early);
      // the speculative read is complete on EVERY path from here on (group 0 never looks at it): a pending
      // load into registers the item loop reuses would cost a vmcnt(0) — a wait for the output stores — per item
      asm volatile("" : "+v"(early));
      const uint32_t first_item = g_cur * kTileItemsPerGroup + item_in_group(wave, 0);
      Item it = load_item(f.tiles + (first_item < f.n_tiles ? first_item : 0u));
      if (wave == 0 && lane == 0) {
        if (g_cur != 0 && !(variant & 1u))
          st_store(f.scan_state + g_cur, ((uint64_t)gen << kGenShift) | kPrefix | (uint64_t)(excl + total_cur));
        if (g_cur + 1u == n_groups_cur)                                    // tile.total_number_of_regular_points
          *glw(f.n_points) = (variant & 1u) ? (excl + total_cur < f.capacity ? excl + total_cur : f.capacity) : excl + total_cur;
      }
      VPCC_STAMP(4)
      uint32_t base = excl;
      for (uint32_t k = 0; k < item_in_group(wave, 0); ++k) base += s_tot[cb][k];

      // ---- 4. per item: compaction of the records through LDS, then lane <-> point (position, colour, stores).
      // The item's geometry has been in registers since it was counted a step ago, its attribute tiles in LDS since
      // the end of the previous step: the loop issues no plane load at all.
#pragma unroll
      for (uint32_t i = 0; i < K; ++i) {
        // scalar on purpose: a branch on a VGPR value is lowered with exec masks and an "exec == 0" bypass edge
        const uint32_t n = __builtin_amdgcn_readfirstlane(s_tot[cb][item_in_group(wave, i)]);
        const uint32_t next_item = g_cur * kTileItemsPerGroup + item_in_group(wave, i + 1u < K ? i + 1u : i);
        const Item nit = load_item(f.tiles + (next_item < f.n_tiles ? next_item : 0u));   // scalar: the descriptor of the wave's next item
        Samples cur;
        cur.occ = (occ_cur >> (4u * i)) & 0xFu;
        cur.g0 = gc0[i]; cur.g1 = gc1[i];                                  // unrolled: static indices
        {
          const uint32_t dup = (dup_cur >> (4u * i)) & 0xFu;                 // from the count phase
          const RecordSpace rs{slots, (LdsByte*)attr + attr_item_off(i) - 8u * kRecordsOwn};
          emit_item<kStamps>(f, it, cur, dup, n, base, lane, rs, attr + attr_luma_off(i), attr + attr_chroma_off(i), variant, [&]() {
            asm volatile("" : "+v"(t_ahead));
            // The tiles of the item emitted before this one are free: their successor — the same item of the group just
            // counted — is requested now, BEFORE this item's stores (behind them it would wait in the memory pipeline for
            // the whole burst to drain; all sixteen tiles at the end of the step: 0.532 ms, three quarters of them here in
            // the last item: 0.530, item by item: 0.527 — profiles/r04/ab_stage.txt).
            if (i >= 1u && have_next && !(variant & 256u)) stage_attributes(*fn, g_next, wave, lane, occ_next, attr_lds, 1u << (i - 1u));
            if (i + 1u == K) {
              // Hand the loop-carried registers over BEFORE the last item's stores: register copies made in
              // the loop latch, behind those stores, are preceded by a vmcnt(0) (with loads and stores both in
              // flight the compiler cannot count, and any register it believes pending costs a full wait).
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                gc0[j] = gn0[j]; gc1[j] = gn1[j];
                asm volatile("" : "+v"(gc0[j].lo), "+v"(gc0[j].hi), "+v"(gc1[j].lo), "+v"(gc1[j].hi));
              }
            }
          }, t_acc);
        }
        // the items between this one and the wave's next one (the other waves')
        for (uint32_t k = item_in_group(wave, i); k < item_in_group(wave, i + 1u) && k < kTileItemsPerGroup; ++k) base += s_tot[cb][k];
        it = nit;
      }
      VPCC_STAMP(5)
    } else if (have_next) {
      // first step of the workgroup: nothing to emit yet
#pragma unroll
      for (int j = 0; j < 4; ++j) { gc0[j] = gn0[j]; gc1[j] = gn1[j]; }
      asm volatile("" : "+v"(early));                        // (never read on this path, but pending in the compiler's books)
      asm volatile("" : "+v"(t_ahead));
    }
    // ---- 5. the attribute tiles of the wave's last item of the group just counted (the tiles of its first three were
    // requested in the item loop as their predecessors' became free): every wave stages and reads its own tiles, and
    // the data has a count phase's two round trips and three items to arrive.
    if (have_next && !(variant & 256u)) {
      __builtin_amdgcn_s_setprio(1);
      stage_attributes(*fn, g_next, wave, lane, occ_next, attr_lds, have_cur ? 0x8u : 0xFu);   // (a workgroup's first step: all four)
      __builtin_amdgcn_s_setprio(0);
    }
    if (!have_next) break;
#ifdef VPCC_DIAGNOSTIC
    ++wg_steps;
#endif
    g_cur = g_next;
    fc = fn;
    n_groups_cur = n_groups_next;
    occ_cur = occ_next;
    dup_cur = dup_next;
    total_cur = total_next;
    cb ^= 1u;
    have_cur = true;
  }
#ifdef VPCC_DIAGNOSTIC
  if ((variant & 8192u) && blockIdx.x < 8192u) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the workgroup's stores have left
    if (threadIdx.x == 0) {
      g_wg_times[blockIdx.x][0] = wg_t0;
      g_wg_times[blockIdx.x][1] = wg_steps;
      g_wg_times[blockIdx.x][2] = stamp_time();
    }
  }
#endif
  if constexpr (kStamps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  VPCC_STAMP(6)
  VPCC_STAMP_FLUSH()
}


}  // namespace vpcc

#ifdef VPCC_DIAGNOSTIC
extern "C" int vpcc_debug_read_wg_times(unsigned long long* out, int n_wgs) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(vpcc::g_wg_times), sizeof(unsigned long long) * 3 * (size_t)n_wgs) != hipSuccess;
}
extern "C" int vpcc_debug_read_stamps(unsigned long long* out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(vpcc::g_stamps), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[16] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(vpcc::g_stamps), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
#endif

namespace vpcc {

void launch_tiles(const DevFrame* d_frames, uint32_t first, uint32_t count, uint32_t max_groups, uint32_t gen,
                  const TileLaunchMap& map, uint32_t resident_per_xcd, void* stream) {
  if (!count || !max_groups) return;
  // groups each workgroup is expected to pipeline (tickets are drawn dynamically; this only sizes the grid)
  uint32_t depth = 3, variant = 0;
#ifdef VPCC_DIAGNOSTIC
  const uint32_t env_variant = [] {                          // read at every launch: tools switch ablations within a process
    const char* e = getenv("VPCC_TILES_VARIANT");
    return e ? (uint32_t)atoi(e) : 0u;
  }();
  static const uint32_t env_depth = [] {
    const char* e = getenv("VPCC_TILES_DEPTH");
    const int v = e ? atoi(e) : 3;
    return (uint32_t)(v < 1 ? 1 : v);
  }();
  variant = env_variant;
  depth = env_depth;
#endif
  const uint32_t frame_groups = (count + 7u) / 8u;
  const uint32_t wgs = (max_groups + depth - 1u) / depth;     // workgroups per frame
  uint32_t grid = map.slots ? 8u * map.slots : 8u * frame_groups * wgs;
  if (frame_groups > kFramesInFlight)                         // rounds: the resident workgroups and no more — but every team
    grid = 8u * std::max(kFramesInFlight,                     // of frames needs a workgroup, however small the device
                         std::min(resident_per_xcd * 4u / kTileWaves, kFramesInFlight * wgs));
#ifdef VPCC_DIAGNOSTIC
  if (variant & 64u) {
    hipLaunchKernelGGL(k_recon_tiles<true>, dim3(grid), dim3(64 * kTileWaves), 0, (hipStream_t)stream, d_frames, first, count,
                       gen, variant, map);
    return;
  }
#endif
  hipLaunchKernelGGL(k_recon_tiles<false>, dim3(grid), dim3(64 * kTileWaves), 0, (hipStream_t)stream, d_frames, first, count,
                     gen, variant, map);
}

__global__ void k_warm_tiles() {}
void launch_warm_tiles(void* stream) { hipLaunchKernelGGL(k_warm_tiles, dim3(1), dim3(64), 0, (hipStream_t)stream); }

}  // namespace vpcc
