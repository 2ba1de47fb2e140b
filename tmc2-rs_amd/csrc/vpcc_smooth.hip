// vpcc_smooth.hip — grid-based geometry and colour smoothing on gfx950.
//
// The reference implements neither (every hook is unimplemented!(): src/decoder.rs:291-299, 630-658,
// src/codec.rs:498-500); the behaviour is this repository's own integer specification "gs1"/"cs1",
// written down in oracle/vpcc_smoothing_spec.h and tested bit for bit against its CPU form.
//
// The unit of work is a QUAD — four consecutive points of a frame, held by one lane: their positions are 24 contiguous
// bytes, their colours 12, their partition entries 8, fetched with whole-dword loads; a wave's 64 quads are a CHUNK of 256
// consecutive points, four chunks a SPAN of kSmoothListSpan points, the unit of the cell lists.  Per pass (one pass serves both
// filters when they use one grid size):
//   k_smooth_stats      : per occupied grid cell {count, 3 sums, sum of patch indices, sum of their squares} — every field
//                         a sum, one 32-byte atomic request per delivery of a cell — into a dense w^3 grid that is all-zero
//                         between launches; every span leaves the list of the cells it delivered;
//   k_smooth_mark       : walks the lists; a cell that mixes patches (one in a hundred) gets its bit, a byte flag on each of
//                         the eight 2x2x2 neighbourhoods it belongs to (a neighbourhood is named by its lower corner) and one
//                         on each of the 27 cells around it;
//   k_smooth_spans      : which spans have anything to do: those with a listed cell that carries the second kind of flag
//                         (once per pass; the colour filter looks again itself where the geometry filter made cells mixed);
//   k_smooth_apply      : the filters.  Waves of the other spans leave at once; the rest derive each point's neighbourhood
//                         from its position, ONE flag load says whether any of its eight cells mixes patches, and only then
//                         are they read (16 bytes each) -> integer trilinear weights -> centroid / mean, thresholded
//                         replacement in place; a point the geometry filter moves takes its sums to its new cell;
//   k_smooth_moved_mark : both filters in one pass: the cells moved points left and entered are looked at again;
//   k_smooth_clear      : walks the lists again: zeroes exactly the listed cells and un-paints the flags (stores only).
// Round 4 rewrote them around the quad (round 3: a thread per point, a 4-byte cell index stored per point and read back
// by both filters, flags on the 3x3x3 block around a mixed cell): 2.19 -> 0.995 ms per 128 S-longdress frames, memory
// traffic 2.13 -> 1.16 x the algorithmic bytes.
#include <hip/hip_runtime.h>

#include "vpcc_device.hpp"
#include "vpcc_devfn.hpp"

namespace vpcc {

namespace {

constexpr uint32_t kCountMask = 0x7FFFFFFFu;                 // SmoothCell::count / SmoothColorCell::count below the mixed bit

// How a coordinate becomes a cell coordinate, min(p / G, w - 1): a shift when G is a power of two (kPow2); else without
// an integer division: (p + 0.5) * fl(1/G) is within 2^-24 * p/G of (p + 0.5)/G, which is at least 0.5/G away from
// every integer — more than that error as long as p * G < 2^23 — so truncation yields floor(p / G) exactly; larger
// grids take the division.
struct GridDims { uint32_t G, w, sh; };                      // sh = log2(G) when G is a power of two
template <bool kPow2>
__device__ __forceinline__ uint32_t cell_coord(uint32_t p, const GridDims& d) {
  uint32_t q;
  if (kPow2) {
    q = p >> d.sh;
  } else if (d.G < 128u) {                                   // p < 2^16  =>  p * G < 2^23
    const float r = 1.0f / (float)d.G;
    q = (uint32_t)__builtin_fmaf((float)p, r, 0.5f * r);
  } else {
    q = p / d.G;
  }
  return q < d.w ? q : d.w - 1u;
}
template <bool kPow2>
__device__ __forceinline__ uint32_t cell_key(uint32_t x, uint32_t y, uint32_t z, const GridDims& d) {
  return (cell_coord<kPow2>(z, d) * d.w + cell_coord<kPow2>(y, d)) * d.w + cell_coord<kPow2>(x, d);
}

// ---- a quad in memory ----------------------------------------------------------------------------------------------
typedef uint32_t u32x2a __attribute__((ext_vector_type(2), aligned(8)));
typedef uint32_t u32x3a __attribute__((ext_vector_type(3), aligned(4)));
struct QuadXyz { uint32_t d[6]; };                           // x0|y0<<16, z0|x1<<16, y1|z1<<16, x2|y2<<16, z2|x3<<16, y3|z3<<16
struct QuadIn { QuadXyz p; uint32_t patch[2]; uint32_t col[3]; };

// Positions of quad q (points 4q .. 4q+3).  The arrays end with four elements of padding (vpcc_runtime.hip), so a quad
// that begins inside an array can be read whole.
__device__ __forceinline__ QuadXyz load_quad_xyz(const vpcc_point3* base, uint32_t q) {
  const VPCC_GLOBAL unsigned char* b = (const VPCC_GLOBAL unsigned char*)base + (size_t)q * 24u;
  const u32x2a a0 = *(const VPCC_GLOBAL u32x2a*)b, a1 = *(const VPCC_GLOBAL u32x2a*)(b + 8), a2 = *(const VPCC_GLOBAL u32x2a*)(b + 16);
  QuadXyz r;
  r.d[0] = a0.x; r.d[1] = a0.y; r.d[2] = a1.x; r.d[3] = a1.y; r.d[4] = a2.x; r.d[5] = a2.y;
  return r;
}
__device__ __forceinline__ void unpack_xyz(const QuadXyz& r, uint32_t (&x)[4], uint32_t (&y)[4], uint32_t (&z)[4]) {
  x[0] = r.d[0] & 0xFFFFu; y[0] = r.d[0] >> 16; z[0] = r.d[1] & 0xFFFFu;
  x[1] = r.d[1] >> 16;     y[1] = r.d[2] & 0xFFFFu; z[1] = r.d[2] >> 16;
  x[2] = r.d[3] & 0xFFFFu; y[2] = r.d[3] >> 16; z[2] = r.d[4] & 0xFFFFu;
  x[3] = r.d[4] >> 16;     y[3] = r.d[5] & 0xFFFFu; z[3] = r.d[5] >> 16;
}
__device__ __forceinline__ void load_quad_patch(const uint16_t* base, uint32_t q, uint32_t (&out)[2]) {
  const u32x2a a = *(const VPCC_GLOBAL u32x2a*)((const VPCC_GLOBAL unsigned char*)base + (size_t)q * 8u);
  out[0] = a.x; out[1] = a.y;
}
__device__ __forceinline__ void load_quad_rgb(const vpcc_color3* base, uint32_t q, uint32_t (&out)[3]) {
  const u32x3a a = *(const VPCC_GLOBAL u32x3a*)((const VPCC_GLOBAL unsigned char*)base + (size_t)q * 12u);
  out[0] = a.x; out[1] = a.y; out[2] = a.z;
}
// r | g << 8 | b << 16 of the quad's four colours
__device__ __forceinline__ void unpack_rgb(const uint32_t (&c)[3], uint32_t (&rgb)[4]) {
  rgb[0] = c[0] & 0xFFFFFFu;
  rgb[1] = __builtin_amdgcn_alignbit(c[1], c[0], 24) & 0xFFFFFFu;
  rgb[2] = __builtin_amdgcn_alignbit(c[2], c[1], 16) & 0xFFFFFFu;
  rgb[3] = c[2] >> 8;
}

// value of the lane's partner inside its quad of four lanes (DPP quad_perm)
template <int kCtrl>
__device__ __forceinline__ uint32_t qperm(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, kCtrl, 0xF, 0xF, true);     // (every lane of a quad has its partner: nothing to preset)
}

// ---- statistics ----------------------------------------------------------------------------------------------------
// mode 0: sums of coordinates (geometry); mode 1: sums of R,G,B (colour); mode 2: both (colour sums into the colour cells).
// Points arrive in emission order, so the 1 024 points of a wave's kStatChunks consecutive chunks fall into a few dozen
// neighbouring cells (a block row of 16 pixels spans two or three cells of size 8), nearly always of ONE patch.  The low
// two bits of each cell coordinate give a cell a slot among 64 in LDS (two cells share a slot only if they are a
// multiple of four cells apart in every axis); the table holds cells of one patch at a time and lives as long as it can:
//   1. a lane adds up, in registers, those of its four points that share a cell — pairwise (1 into 0, 3 into 2), then
//      2 into 0: A A A A becomes one LEADER, A A B B two, the rest stay on their own (a cell may have several leaders);
//   2. every leader swaps its cell into its slot if the slot is free (LDS compare-and-swap); what the swap returns — free, or the
//      leader's own cell — says whether the slot is the leader's: the others stay PENDING;
//   3. the leaders that have their slot add their sums to it with LDS atomics (three 64-bit words: x | y, z | G, count | R | B);
//   4. while leaders are pending, the table is flushed — which frees every slot — and they try again (at least one per
//      contended slot succeeds each time); it is flushed for good after the wave's last chunk, and before points of
//      another patch (a chunk that holds several patches is worked on patch by patch);
//   flush: lane s owns slot s: the occupied slots are lined up, appended to the wave's cell list, and lane 4 j + t
//      carries 64-bit word t of the j-th cell: the four words of a cell leave as ONE 32-byte atomic request
//      ({count, s0}, {s1, s2}, sum of squared patch indices, {sum of patch indices, 0}: no carry crosses a pair's halves;
//      the table holds ONE patch, so the patch sums follow from the count), and lanes 4 j and 4 j + 1 the two words of its
//      colour cell.  Every field of a cell is a sum, so it does
//      not matter in how many parts a wave delivers a cell.
// The grids are all-zero between launches: k_smooth_clear zeroes exactly the listed cells afterwards — no dense memset
// (50 MB per frame at w = 128) per launch.
// History of the per-point form this replaces (rounds 1-3, per 32 S-longdress frames): one set of atomics per point
// 17.5 ms; runs of equal cells reduced by a segmented DPP scan with six atomics per run 2.7 ms (30 M atomics: the L2
// retires ~12 per ns); merged through an LDS table with ds_cmpst/ds_add per workgroup 0.62-0.70; per wave of 64 points
// with a de-duplication loop and one atomic instruction per cell 0.245; the slot scheme per 64 points, with a loop of
// wave-wide reductions for waves with a collision, 0.15-0.17 (it is the number of atomic REQUESTS that counts: the
// elected lanes issuing four 8-byte atomics each was 3 x slower).
struct StatTable {
  uint32_t key[64];            // cell index of the slot's cell, all-ones: free
  // The sums of a slot in THREE 64-bit words: it was six LDS adds per leader (x, y, z, count + R << 16, G, B) and a look at the slot
  // behind the compare-and-swap, on the CU's one LDS — INSTS_LDS 22.4 M -> 11.8 M per launch, the kernel 0.439 -> 0.379 ms (round 5).
  // No field reaches its neighbour: a table lives for kSmoothListSpan <= 1 024 points, so count < 2^11, a colour sum < 2^18, a
  // coordinate sum < 2^26.
  uint64_t xy[64];             // x | y << 32
  uint64_t zg[64];             // z | G << 32
  uint64_t crb[64];            // count | R << 16 | B << 40
  uint32_t order[64];          // the occupied slots, lined up
};
constexpr uint32_t kFreeSlot = 0xFFFFFFFFu;                  // (cells < 2^32 - 1: vpcc_gof_smooth)
// which of the six words a mode keeps: x, y, z, G, B (bits 0-4), count + R (always)
template <uint32_t kMode> constexpr uint32_t stat_words() { return kMode == 0 ? 0x07u : kMode == 1 ? 0x18u : 0x1Fu; }

struct StatDst {
  SmoothCell* grid;
  SmoothColorCell* cgrid;      // mode 2 only
  uint32_t* list;              // the wave's cell list (kSmoothListLen entries reserved)
};

// Delivers every occupied slot to the grid — the table's cells are of patch `pl` —, appends its cell to the list (`m`
// entries so far) and frees it.
template <uint32_t kMode>
__device__ __forceinline__ void stats_flush(StatTable& T, const StatDst& dst, uint32_t pl, uint32_t& m, uint32_t lane) {
  constexpr uint32_t kWords = stat_words<kMode>();
  const uint32_t own = T.key[lane];
  const bool occupied = own != kFreeSlot;
  const uint64_t em = __ballot(occupied);
  const uint32_t ncell = (uint32_t)__builtin_popcountll(em);
  if (ncell == 0u) return;
  const uint32_t rank = mbcnt(em);
  if (occupied) {
    T.order[rank] = lane;
    dst.list[m + rank] = own;
  }
  __builtin_amdgcn_wave_barrier();                           // (LDS operations of one wave execute in order)
  for (uint32_t base = 0; base < ncell; base += 16u) {
    const uint32_t j = base + (lane >> 2), t = lane & 3u;
    if (j < ncell) {
      const uint32_t sl = T.order[j];
      const uint32_t k = T.key[sl];
      const uint64_t cr = T.crb[sl], zg = T.zg[sl], xy = kMode == 1u ? 0ull : T.xy[sl];
      const uint32_t cnt = (uint32_t)cr & 0xFFFFu;
      const uint32_t s0 = kMode == 1u ? (uint32_t)(cr >> 16) & 0xFFFFFFu : (uint32_t)xy;                      // (mode 1: R, G, B)
      const uint32_t s1 = kMode == 1u ? (uint32_t)(zg >> 32) : (uint32_t)(xy >> 32), s2 = kMode == 1u ? (uint32_t)(cr >> 40) : (uint32_t)zg;
      const uint32_t sp = cnt * pl;
      const uint64_t val = t == 0 ? (uint64_t)cnt | ((uint64_t)s0 << 32)
                         : t == 1 ? (uint64_t)s1 | ((uint64_t)s2 << 32)
                         : t == 2 ? (uint64_t)sp * pl : (uint64_t)sp;
      atomicAdd(reinterpret_cast<unsigned long long*>(dst.grid + k) + t, (unsigned long long)val);
      if (kMode == 2u && t < 2u) {                           // the colour cell, from the same words of the table: lanes 0 and 1 of the four
        const uint64_t cval = t == 0 ? (cr & 0xFFFFull) | (((cr >> 16) & 0xFFFFFFull) << 32)         // {count, R}
                                     : (zg >> 32) | ((cr >> 40) << 32);                              // {G, B}
        atomicAdd(reinterpret_cast<unsigned long long*>(dst.cgrid + k) + t, (unsigned long long)cval);
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  if (occupied) {
    T.key[lane] = kFreeSlot;
    T.crb[lane] = 0ull;
    T.zg[lane] = 0ull;
    if (kWords & 3u) T.xy[lane] = 0ull;
  }
  __builtin_amdgcn_wave_barrier();
  m += ncell;
}

// One chunk: `tpatch` is the patch of the table's cells (wave-uniform; anything before the first point).
template <uint32_t kMode, bool kPow2>
__device__ __forceinline__ void stats_chunk(const QuadIn& in, uint32_t nvalid, const GridDims& gd, const StatDst& dst, StatTable& T,
                                            uint32_t& tpatch, uint32_t& m) {
  constexpr uint32_t kWords = stat_words<kMode>();
  constexpr uint32_t kLaneWords = kMode == 0 ? 0xFu : kMode == 1 ? 0x18u : 0x1Fu;     // in registers: x, y, z, count | R << 16, G | B << 16
  static_assert(kSmoothListSpan <= 1024u, "count < 2^16 below the red sum; sums of 1 024 bytes in 32 bits");
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t x[4], y[4], z[4];
  unpack_xyz(in.p, x, y, z);
  // the quad's colours, each in the low three bytes of a dword (the top byte is the next colour's: never looked at)
  const uint32_t rgb[4] = {in.col[0], __builtin_amdgcn_alignbit(in.col[1], in.col[0], 24),
                           __builtin_amdgcn_alignbit(in.col[2], in.col[1], 16), in.col[2] >> 8};
  const uint32_t patch[4] = {in.patch[0] & 0xFFFFu, in.patch[0] >> 16, in.patch[1] & 0xFFFFu, in.patch[1] >> 16};
  uint32_t cell[4], slot[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t cx = cell_coord<kPow2>(x[j], gd), cy = cell_coord<kPow2>(y[j], gd), cz = cell_coord<kPow2>(z[j], gd);
    cell[j] = (cz * gd.w + cy) * gd.w + cx;
    slot[j] = (cx & 3u) | ((cy & 3u) << 2) | ((cz & 3u) << 4);
  }
  // (per-lane flags are single variables, not arrays: as arrays that live across the loops below the compiler packs them into
  // bytes of one register and spends a fifth of the kernel's vector instructions packing and unpacking them)
  bool rem0 = nvalid > 0u, rem1 = nvalid > 1u, rem2 = nvalid > 2u, rem3 = nvalid > 3u;
  auto rem = [&](int j) -> bool& { return j == 0 ? rem0 : j == 1 ? rem1 : j == 2 ? rem2 : rem3; };
  do {
    // the patch of the first point that is left (the chunk's first pass: of its first point), and the points of that patch
    uint32_t P = 0;
    {
      const uint64_t r0 = __ballot(rem0), r1 = __ballot(rem1), r2 = __ballot(rem2), r3 = __ballot(rem3);
      if (r0) P = (uint32_t)__builtin_amdgcn_readlane((int)patch[0], (int)__builtin_ctzll(r0));
      else if (r1) P = (uint32_t)__builtin_amdgcn_readlane((int)patch[1], (int)__builtin_ctzll(r1));
      else if (r2) P = (uint32_t)__builtin_amdgcn_readlane((int)patch[2], (int)__builtin_ctzll(r2));
      else P = (uint32_t)__builtin_amdgcn_readlane((int)patch[3], (int)__builtin_ctzll(r3));
    }
    if (P != tpatch) {                                       // (wave-uniform)
      stats_flush<kMode>(T, dst, tpatch, m, lane);
      tpatch = P;
    }
    uint32_t key[4], W[4][5];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool sel = rem(j) && patch[j] == P;
      rem(j) = rem(j) && !sel;
      key[j] = sel ? cell[j] : kFreeSlot;                     // (a point that is not taken joins nobody and leads nothing)
      W[j][0] = x[j]; W[j][1] = y[j]; W[j][2] = z[j];
      W[j][3] = __builtin_amdgcn_perm(rgb[j], 1u, 0x0C040C00u);          // 1 | R << 16
      W[j][4] = __builtin_amdgcn_perm(rgb[j], 0u, 0x0C060C05u);          // G | B << 16
    }
    // 1. points of the lane that share a cell
    const bool s10 = key[1] == key[0], s32 = key[3] == key[2], s20 = key[2] == key[0];
    bool pend0 = key[0] != kFreeSlot, pend1 = key[1] != kFreeSlot && !s10, pend2 = key[2] != kFreeSlot && !s20, pend3 = key[3] != kFreeSlot && !s32;
    auto pend = [&](int j) -> bool& { return j == 0 ? pend0 : j == 1 ? pend1 : j == 2 ? pend2 : pend3; };
#define VPCC_JOIN(cond, to, from)                                                         \
    {                                                                                     \
      _Pragma("unroll") for (int q = 0; q < 5; ++q) if ((kLaneWords >> q) & 1u) W[to][q] += (cond) ? W[from][q] : 0u; \
    }
    VPCC_JOIN(s10, 0, 1)
    VPCC_JOIN(s32, 2, 3)
    VPCC_JOIN(s20, 0, 2)
#undef VPCC_JOIN
    // ... and neighbouring lanes whose first leaders share a cell: lane pairs, then pairs of pairs (the LDS serves the lanes of
    // one slot one after the other; a CU has ONE LDS, and it is this kernel's busiest unit)
    {
      // (the neighbour's cell is fetched by every lane, before any condition: a DPP read under `pend0 && ...` runs with the
      // lanes without a leader switched off, and reads 0 from them — the index of a cell)
      const uint32_t key1 = qperm<0xB1>(key[0]);                                // lane ^ 1
      const bool same1 = pend0 && key1 == key[0];
#pragma unroll
      for (int q = 0; q < 5; ++q) if ((kLaneWords >> q) & 1u) { const uint32_t o = qperm<0xB1>(W[0][q]); W[0][q] += (same1 && !(lane & 1u)) ? o : 0u; }
      if (same1 && (lane & 1u)) pend0 = false;
      const uint32_t key2 = pend0 ? key[0] : kFreeSlot;
      const uint32_t key3 = qperm<0x4E>(key2);                                  // lane ^ 2 (lanes 0 and 2 of a quad: never given away above)
      const bool same2 = pend0 && key3 == key[0];
#pragma unroll
      for (int q = 0; q < 5; ++q) if ((kLaneWords >> q) & 1u) { const uint32_t o = qperm<0x4E>(W[0][q]); W[0][q] += (same2 && !(lane & 2u)) ? o : 0u; }
      if (same2 && (lane & 2u)) pend0 = false;
    }
    for (;;) {
      // 2. a slot each, leader by leader: compare-and-swap "free -> my cell".  What the swap RETURNS says whose the slot is — it was
      // free (now mine) or held my cell already —: no second look at the slot;
      // 3. the leaders that have their slot add their sums to it, three 64-bit LDS adds; the others stay pending
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        auto add = [&](uint32_t to) {
          typedef unsigned long long u64;
          if (kWords & 3u) atomicAdd(reinterpret_cast<u64*>(&T.xy[to]), (u64)W[j][0] | ((u64)W[j][1] << 32));
          if (kMode == 2u) atomicAdd(reinterpret_cast<u64*>(&T.zg[to]), (u64)W[j][2] | ((u64)(W[j][4] & 0xFFFFu) << 32));
          else if (kMode == 0u) atomicAdd(reinterpret_cast<uint32_t*>(&T.zg[to]), W[j][2]);                   // (the low half: z)
          else atomicAdd(reinterpret_cast<uint32_t*>(&T.zg[to]) + 1, W[j][4] & 0xFFFFu);                     // (the high half: G)
          if (kMode == 0u) atomicAdd(reinterpret_cast<uint32_t*>(&T.crb[to]), W[j][3]);                      // (no colours: the count)
          else atomicAdd(reinterpret_cast<u64*>(&T.crb[to]), (u64)W[j][3] | ((u64)(W[j][4] >> 16) << 40));
        };
        if (pend(j)) {                                        // (leaders only: the LDS's time goes with the lanes that take part)
          const uint32_t was = atomicCAS(&T.key[slot[j]], kFreeSlot, key[j]);
          if (was == kFreeSlot || was == key[j]) {
            add(slot[j]);
            pend(j) = false;
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      if (__ballot(pend0 || pend1 || pend2 || pend3) == 0) break;
      // 4. somebody's slot is taken by another cell: deliver what the table holds, and again
      stats_flush<kMode>(T, dst, tpatch, m, lane);
    }
  } while (__ballot(rem0 || rem1 || rem2 || rem3) != 0);
}

template <uint32_t kMode>
__device__ __forceinline__ QuadIn load_stat_input(const DevFrame& f, uint32_t q, uint32_t n) {
  QuadIn in{};
  if (q * 4u < n) {
    in.p = load_quad_xyz(f.out_xyz, q);
    if (kMode) load_quad_rgb(f.out_rgb, q, in.col);
    load_quad_patch(f.out_patch, q, in.patch);
  }
  return in;
}

}  // namespace

// A wave handles kStatChunks consecutive chunks — kSmoothListSpan points — with one table and one cell list.
constexpr uint32_t kStatChunks = kSmoothListSpan / 256u;
template <uint32_t kMode, bool kPow2>
__global__ __launch_bounds__(256) void k_smooth_stats(const DevFrame* __restrict__ frames, uint32_t first,
                                                      SmoothGrid sg, GridDims gd) {
  __shared__ StatTable s_tab[4];
  const DevFrame& f = frames[first + blockIdx.y];
  const uint32_t n = min(*gl(f.n_points), f.capacity);
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const uint32_t span = blockIdx.x * 4u + wave;                          // the wave's kSmoothListSpan points
  if (span * kSmoothListSpan >= n) return;
  StatTable& T = s_tab[wave];
  T.key[lane] = kFreeSlot;
  T.xy[lane] = T.zg[lane] = T.crb[lane] = 0ull;
  StatDst dst;
  dst.grid = sg.cells(blockIdx.y);
  dst.cgrid = sg.color_cells(blockIdx.y);
  dst.list = sg.lists(blockIdx.y) + (size_t)span * kSmoothListLen;
  uint32_t m = 0, tpatch = 0;
  // A chunk's points are fetched right before it is worked on.  (Fetching the next chunk during the current one is no faster:
  // gfx9 counts loads and stores with ONE in-order counter, so a wait for a load issued while the table's atomics are in
  // flight would be a wait for those atomics.  Fetching all four chunks up front is no faster per wave either and holds 44
  // registers: 128 instead of 96, four waves per SIMD instead of five, 0.450 instead of 0.442 ms.)
#pragma unroll
  for (uint32_t c = 0; c < kStatChunks; ++c) {
    const uint32_t chunk = span * kStatChunks + c;
    if (chunk * 256u >= n) break;
    const QuadIn in = load_stat_input<kMode>(f, chunk * 64u + lane, n);
    const uint32_t i0 = chunk * 256u + 4u * lane;
    const uint32_t nvalid = i0 >= n ? 0u : min(n - i0, 4u);
    stats_chunk<kMode, kPow2>(in, nvalid, gd, dst, T, tpatch, m);
  }
  stats_flush<kMode>(T, dst, tpatch, m, lane);
  if (lane == 0) sg.list_counts(blockIdx.y)[span] = m;
}

namespace {
// "the cell holds points of more than one patch" from its sums (vpcc_device.hpp)
__device__ __forceinline__ bool sums_mixed(uint32_t count, uint64_t sp2, uint32_t sp) { return (uint64_t)count * sp2 != (uint64_t)sp * sp; }

// A point's 2x2x2 neighbourhood is named by its lower corner s in [-1, w - 1]^3, stored shifted by one: S = s + 1 in
// [0, w]^3, flag index (Sz (w + 1) + Sy)(w + 1) + Sx.  A cell c belongs to the eight neighbourhoods S = c + {0, 1}^3.
// Beside them, a byte per CELL: "a mixed cell among the 3x3x3 around this one" — which is what all eight neighbourhoods
// of the cell's points can see: a list of cells without any such byte set means none of its points has anything to do.
__device__ __forceinline__ void paint_flags(const SmoothGrid& sg, uint32_t frame, uint32_t key, uint32_t w, unsigned char value) {
  unsigned char* flags = sg.flags(frame);
  unsigned char* near = sg.near(frame);
  const uint32_t cx = key % w, cy = (key / w) % w, cz = key / (w * w), w1 = w + 1u;
#pragma unroll
  for (uint32_t d = 0; d < 8; ++d)
    flags[((size_t)(cz + (d >> 2)) * w1 + (cy + ((d >> 1) & 1u))) * w1 + (cx + (d & 1u))] = value;
  for (uint32_t z = cz ? cz - 1u : 0u; z <= min(cz + 1u, w - 1u); ++z)
    for (uint32_t y = cy ? cy - 1u : 0u; y <= min(cy + 1u, w - 1u); ++y)
      for (uint32_t x = cx ? cx - 1u : 0u; x <= min(cx + 1u, w - 1u); ++x)
        near[((size_t)z * w + y) * w + x] = value;
}

// A wave per list: lane t sees to entries t and 64 + t of the list at once (these passes are chains of dependent loads —
// entry, cell, colour cell — so they want as many of them in flight as there are entries: a list holds 50 cells on
// average, and with one entry per lane and trip the waves of the longer lists went through the chain twice, the second
// time for a handful of lanes: mark / clear 0.080 / 0.122 ms per 128 frames instead of 0.077 / 0.120; sixteen threads per
// list took 0.18 / 0.16, 34 us per wave; a wave per four lists, entry t of each per lane and trip, was no faster).
// body(key[2], span, j[2], valid[2], second): `second` = the list has entries in the second half of this trip.
template <class F>
__device__ __forceinline__ void for_listed_cells(const SmoothGrid& sg, uint32_t frame, uint32_t e, uint32_t n, F body) {
  const uint32_t span = e >> 6, t = e & 63u;
  if (span * kSmoothListSpan >= n) return;
  const uint32_t cnt = sg.list_counts(frame)[span];
  const uint32_t* list = sg.lists(frame) + (size_t)span * kSmoothListLen;
  for (uint32_t j0 = 0; j0 < cnt; j0 += 128u) {             // (the trip count is the wave's: the body may use ballots)
    const uint32_t j[2] = {j0 + t, j0 + 64u + t};
    const bool valid[2] = {j[0] < cnt, j[1] < cnt};
    const uint32_t key[2] = {valid[0] ? gl(list)[j[0]] : 0u, valid[1] ? gl(list)[j[1]] : 0u};
    body(key, span, j, valid, j0 + 64u < cnt);
  }
}
}  // namespace

// A filter changes a point only if one of the 2x2x2 cells around it holds points of more than one patch ("mixed") —
// one cell in a hundred.  This pass reads every listed cell, gives a mixed one its bit (above the count: the filters
// read the first 16 bytes of a cell only) and paints its flags: a byte on each of the eight neighbourhoods the cell
// belongs to, so that a filter decides with ONE byte load whether a point needs its eight cells at all, and a byte on
// each of the 27 cells around it, so that a filter decides from a list of cells whether any of its points does (round 3
// painted the 27 cells only and looked a point's own cell up through a cell index stored per point).  Nothing is
// written for the other ninety-nine.
__global__ __launch_bounds__(256) void k_smooth_mark(const DevFrame* __restrict__ frames, uint32_t first, SmoothGrid sg,
                                                     uint32_t w) {
  const DevFrame& f = frames[first + blockIdx.y];
  const uint32_t n = min(*gl(f.n_points), f.capacity);
  if (blockIdx.x * (4u * kSmoothListSpan) >= n) return;     // 256 threads = 4 lists
  for_listed_cells(sg, blockIdx.y, blockIdx.x * 256u + threadIdx.x, n, [&](const uint32_t (&key)[2], uint32_t span, const uint32_t (&j)[2],
                                                                          const bool (&valid)[2], bool second) {
    SmoothCell c2[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) c2[u] = valid[u] ? gload(sg.cells(blockIdx.y) + key[u]) : SmoothCell{};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      bool painted = false;
      if (valid[u]) {
        SmoothCell* cell = sg.cells(blockIdx.y) + key[u];
        const SmoothCell& c = c2[u];
        if (!(c.mixed & kSmoothMixed)) {                    // else: another entry of the same cell has been here
          const uint32_t count = c.count & kCountMask;      // (a second visitor at the same moment may see the first one's bit)
          // The cells' sums are 32 bits wide, two to a 64-bit atomic add: beyond this many points in ONE cell a sum of 16-bit
          // values could carry into its neighbour (the specification's u32 sums would wrap instead) — reported, not smoothed over.
          if (count > kSmoothCellMaxPoints) atomicOr(f.error_flag, kErrorSmoothCellOverflow);
          if (sums_mixed(count, c.sp2, c.sp)) {
            painted = true;
            paint_flags(sg, blockIdx.y, key[u], w, 1);
            cell->count = count | kSmoothCountMixed;        // (no atomic touches the cell during this kernel)
            cell->mixed = kSmoothMixed | kSmoothPainted;
            if (sg.color_offset) {                          // both filters: the colour filter reads the colour cells only
              SmoothColorCell* cc = sg.color_cells(blockIdx.y) + key[u];
              cc->count = count | kColorCellMixed;
            }
          }
        }
      }
      // which entries had their flags painted from here: k_smooth_clear un-paints those, and reads no cell for it
      const uint64_t mask = __ballot(painted);
      if ((threadIdx.x & 63u) == 0 && (u == 0 || second)) sg.painted(blockIdx.y)[(size_t)span * (kSmoothListLen / 64u) + (j[u] >> 6)] = mask;
    }
  });
}

namespace {
// Restores the all-zero state of one cell — nothing is read: a cell that several lists hold is zeroed several times.
__device__ __forceinline__ void zero_cell(const SmoothGrid& sg, uint32_t frame, uint32_t key, bool both) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  VPCC_GLOBAL u32x4* c = (VPCC_GLOBAL u32x4*)(sg.cells(frame) + key);
  c[0] = u32x4{0u, 0u, 0u, 0u}; c[1] = u32x4{0u, 0u, 0u, 0u};
  if (both) *(VPCC_GLOBAL u32x4*)(sg.color_cells(frame) + key) = u32x4{0u, 0u, 0u, 0u};
}
}  // namespace

// Restores the all-zero state: every listed cell is zeroed, and un-painted if k_smooth_mark painted flags for that entry
// (round 4; before, this pass read every cell to find out: 0.165 ms per 128 frames).  both: the colour cells too, and what
// the moved points touched.
template <bool kPow2>
__global__ __launch_bounds__(256) void k_smooth_clear(const DevFrame* __restrict__ frames, uint32_t first, SmoothGrid sg,
                                                      GridDims gd, bool both) {
  const DevFrame& f = frames[first + blockIdx.y];
  const uint32_t n = min(*gl(f.n_points), f.capacity);
  if (blockIdx.x * (4u * kSmoothListSpan) >= n) return;
  const uint32_t e = blockIdx.x * 256u + threadIdx.x;
  for_listed_cells(sg, blockIdx.y, e, n, [&](const uint32_t (&key)[2], uint32_t span, const uint32_t (&j)[2], const bool (&valid)[2], bool) {
    uint64_t painted[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) painted[u] = valid[u] ? gl(sg.painted(blockIdx.y))[(size_t)span * (kSmoothListLen / 64u) + (j[u] >> 6)] : 0ull;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (!valid[u]) continue;
      zero_cell(sg, blockIdx.y, key[u], both);
      if ((painted[u] >> (j[u] & 63u)) & 1ull) paint_flags(sg, blockIdx.y, key[u], gd.w, 0);
    }
  });
  if (both) {
    // the cells moved points went INTO are in no list, and flags k_smooth_moved_mark painted are in no list's bits:
    // sixteen of a list's threads see to the moved points of one wave of 64 points each
    static_assert(kSmoothListSpan == 16u * 64u, "sixteen words of moved bits per list");
    const uint32_t word = (e >> 6) * 16u + (e & 63u);
    if ((e & 63u) < 16u && word * 64u < n) {
      const uint64_t moved = sg.moved(blockIdx.y)[word];
      if (moved) {
        const uint64_t pf = sg.moved_painted(blockIdx.y)[2u * word], pt_ = sg.moved_painted(blockIdx.y)[2u * word + 1u];
        for (uint64_t mask = moved; mask; mask &= mask - 1ull) {
          const uint32_t b = (uint32_t)__builtin_ctzll(mask), i = word * 64u + b;
          const vpcc_point3 pt = gload(f.out_xyz + i);
          const uint32_t to = cell_key<kPow2>(pt.x, pt.y, pt.z, gd);
          zero_cell(sg, blockIdx.y, to, true);
          if ((pt_ >> b) & 1ull) paint_flags(sg, blockIdx.y, to, gd.w, 0);
          if ((pf >> b) & 1ull) paint_flags(sg, blockIdx.y, sg.old_keys(blockIdx.y)[i], gd.w, 0);
        }
      }
    }
  }
}

namespace {
// One axis of a point's neighbourhood: the shifted lower corner S (= s + 1) and the weights of its two cells.
template <bool kPow2>
__device__ __forceinline__ uint32_t corner_coord(uint32_t p, const GridDims& d) {
  const uint32_t q = cell_coord<kPow2>(p, d);
  return (p - q * d.G < d.G / 2u) ? q : q + 1u;
}
__device__ __forceinline__ void axis_weights(uint32_t p, uint32_t S, uint32_t G, int64_t wt[2]) {
  const int32_t t = 2 * ((int32_t)p - (((int32_t)S - 1) * (int32_t)G + (int32_t)(G / 2u))) + 1;
  wt[0] = 2 * (int64_t)G - t;
  wt[1] = t;
}
template <bool kPow2>
__device__ __forceinline__ uint32_t flag_index(uint32_t x, uint32_t y, uint32_t z, const GridDims& d) {
  const uint32_t w1 = d.w + 1u;
  return (corner_coord<kPow2>(z, d) * w1 + corner_coord<kPow2>(y, d)) * w1 + corner_coord<kPow2>(x, d);
}

// The 2x2x2 cells around a point: what a filter needs of a cell is its first 16 bytes, {count | mixed bit, three sums}
// (SmoothCell for a single filter, SmoothColorCell for the colour filter of a pass that serves both).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct Hood { u32x4 c[8]; };
__device__ __forceinline__ bool load_hood(const unsigned char* cells, uint32_t cell_bytes, const uint32_t (&S)[3], uint32_t w, Hood& h) {
  uint32_t any = 0;
#pragma unroll
  for (int d = 0; d < 8; ++d) {
    const uint32_t cx = S[0] - 1u + (d & 1), cy = S[1] - 1u + ((d >> 1) & 1), cz = S[2] - 1u + (d >> 2);   // (-1 wraps to 2^32 - 1: outside)
    h.c[d] = u32x4{0u, 0u, 0u, 0u};
    if (cx < w && cy < w && cz < w)
      h.c[d] = *(const VPCC_GLOBAL u32x4*)(cells + (((size_t)cz * w + cy) * w + cx) * cell_bytes);
    any |= h.c[d].x;
  }
  return (any & kSmoothCountMixed) != 0;
}

// A moved point that is in another cell now takes its count, colour and patch sums from the cell it was counted in to
// the cell it is in (both filters in one pass: the colour filter's cells are those of the smoothed positions).  The
// geometry filter is still reading the first 16 bytes of the cells — {count, coordinate sums} — so those stay as they
// are (nobody needs them afterwards): the colour cells' count is the one that moves.
__device__ __forceinline__ void move_sums(const SmoothGrid& sg, uint32_t frame, uint32_t from, uint32_t to, uint32_t patch,
                                          const vpcc_color3& col) {
  SmoothCell* a = sg.cells(frame) + from;
  SmoothCell* b = sg.cells(frame) + to;
  SmoothColorCell* ca = sg.color_cells(frame) + from;
  SmoothColorCell* cb = sg.color_cells(frame) + to;
  atomicSub(&ca->count, 1u); atomicAdd(&cb->count, 1u);     // (the mixed bit above the count is not touched: count >= 1)
  atomicSub(&a->sp, patch); atomicAdd(&b->sp, patch);
  const unsigned long long q = (unsigned long long)patch * patch;
  atomicAdd(reinterpret_cast<unsigned long long*>(&a->sp2), 0ull - q);
  atomicAdd(reinterpret_cast<unsigned long long*>(&b->sp2), q);
  const uint32_t rgb[3] = {col.r, col.g, col.b};
  for (int c = 0; c < 3; ++c) {
    atomicSub(&ca->s[c], rgb[c]);
    atomicAdd(&cb->s[c], rgb[c]);
  }
}

// Quotients of the filters without the 64-bit integer division (some 150 instructions each): for 0 <= a < 2^52 and
// 0 < b < 2^52 the product a * fl(1 / b) is within one of a / b, so its integer part is the quotient or a neighbour,
// and the remainder says which.  Anything else (weights of a point outside the grid can be negative) divides in integers.
__device__ __forceinline__ bool small_operands(int64_t a, int64_t b) {
  return (uint64_t)a < (1ull << 52) && (uint64_t)b < (1ull << 52);       // (b > 0 is the caller's)
}
__device__ __forceinline__ int64_t div_small(int64_t a, int64_t b, double inv_b) {
  int64_t q = (int64_t)((double)a * inv_b);
  const int64_t r = a - q * b;
  q += r < 0 ? -1 : r >= b ? 1 : 0;
  return q;
}
// floor(a / b) for a < 2^24 and a / b < 2^8 (a sum of up to 65 537 bytes over their number), inv_b = fl(1 / b).
// For b < 8 190 without a correction: (a + 0.5) / b has the integer part of a / b and is at least 0.5 / b > 6.1e-5 away from
// every integer, and fl((a + 0.5) fl(1 / b)) — a + 0.5 < 2^21 + 0.5 is exact, one rounding in the product — is within
// 255.5 x 2^-23 = 3.1e-5 of it.  (A cell of grid size 8 holds a few hundred points.)
__device__ __forceinline__ uint32_t mean_small(uint32_t a, uint32_t b, float inv_b) {
  if (b < 8190u) return (uint32_t)__builtin_fmaf((float)a, inv_b, 0.5f * inv_b);
  uint32_t q = (uint32_t)((float)a * inv_b);
  const int32_t r = (int32_t)(a - q * b);
  q += r < 0 ? -1 : (uint32_t)r >= b ? 1 : 0;
  return q;
}

// The weights of a point inside the grid are 1 .. 2 G - 1 per axis: for grids up to 512 their products stay below 2^30 and a
// weighted sum is one 32 x 32 + 64-bit multiply-add per cell and value.  (Weights of a point outside the grid can be negative
// or large: 64-bit arithmetic as the specification writes it.)
__device__ __forceinline__ bool small_weights(const int64_t (&wt)[3][2]) {
  return (uint64_t)(wt[0][0] | wt[0][1] | wt[1][0] | wt[1][1] | wt[2][0] | wt[2][1]) < 1024u;
}
__device__ __forceinline__ void hood_weights32(const int64_t (&wt)[3][2], uint32_t (&W)[8]) {
  const uint32_t xy[4] = {(uint32_t)wt[0][0] * (uint32_t)wt[1][0], (uint32_t)wt[0][1] * (uint32_t)wt[1][0],
                          (uint32_t)wt[0][0] * (uint32_t)wt[1][1], (uint32_t)wt[0][1] * (uint32_t)wt[1][1]};
#pragma unroll
  for (int d = 0; d < 8; ++d) W[d] = xy[d & 3] * (uint32_t)wt[2][d >> 2];
}

template <bool kPow2>
__device__ __forceinline__ void smooth_apply_geometry_point(const DevFrame& f, uint32_t frame, uint32_t i, const SmoothGrid& sg,
                                                            const GridDims& gd, uint32_t T, bool both, const uint32_t (&p)[3]) {
  const uint32_t S[3] = {corner_coord<kPow2>(p[0], gd), corner_coord<kPow2>(p[1], gd), corner_coord<kPow2>(p[2], gd)};
  Hood h;
  if (!load_hood(reinterpret_cast<const unsigned char*>(sg.cells(frame)), sizeof(SmoothCell), S, gd.w, h)) return;
  int64_t wt[3][2];
#pragma unroll
  for (int a = 0; a < 3; ++a) axis_weights(p[a], S[a], gd.G, wt[a]);
  int64_t num[3] = {0, 0, 0}, den = 0;
  if (small_weights(wt)) {
    uint32_t W[8];
    hood_weights32(wt, W);
    uint64_t n0 = 0, n1 = 0, n2 = 0, dn = 0;
#pragma unroll
    for (int d = 0; d < 8; ++d) {                           // (an empty or absent cell adds nothing: its sums are zero)
      const u32x4 c = h.c[d];
      n0 += (uint64_t)W[d] * c.y; n1 += (uint64_t)W[d] * c.z; n2 += (uint64_t)W[d] * c.w;
      dn += (uint64_t)W[d] * (c.x & kCountMask);
    }
    num[0] = (int64_t)n0; num[1] = (int64_t)n1; num[2] = (int64_t)n2; den = (int64_t)dn;
  } else {
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      const u32x4 c = h.c[d];
      const uint32_t count = c.x & kCountMask;
      if (!count) continue;
      const int64_t W = wt[0][d & 1] * wt[1][(d >> 1) & 1] * wt[2][d >> 2];
      num[0] += W * c.y; num[1] += W * c.z; num[2] += W * c.w;
      den += W * count;
    }
  }
  if (den <= 0) return;
  int64_t C[3], d2 = 0;
  const int64_t A[3] = {16 * num[0] + den / 2, 16 * num[1] + den / 2, 16 * num[2] + den / 2};
  if (small_operands(A[0] | A[1] | A[2], den)) {
    const double inv = 1.0 / (double)den;
#pragma unroll
    for (int a = 0; a < 3; ++a) C[a] = div_small(A[a], den, inv);
  } else {
#pragma unroll
    for (int a = 0; a < 3; ++a) C[a] = A[a] / den;
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int64_t d = 16 * (int64_t)p[a] - C[a];
    d2 += d * d;
  }
  if (d2 > 256 * (int64_t)T * T) {
    vpcc_point3 o;
    const int64_t x = (C[0] + 8) >> 4, y = (C[1] + 8) >> 4, z = (C[2] + 8) >> 4;
    o.x = (uint16_t)(x > 65535 ? 65535 : x); o.y = (uint16_t)(y > 65535 ? 65535 : y); o.z = (uint16_t)(z > 65535 ? 65535 : z);
    gstore(f.out_xyz + i, o);
    // Both filters in one pass: which points moved is noted, one bit per point (a word per wave of 64 points, all-zero
    // before: vpcc_gof_smooth) with the cell each was counted in, for k_smooth_moved_mark and k_smooth_clear.  One point
    // in a few thousand moves: non-returning atomics.
    if (both) {
      const uint32_t from = cell_key<kPow2>(p[0], p[1], p[2], gd), to = cell_key<kPow2>(o.x, o.y, o.z, gd);
      sg.old_keys(frame)[i] = from;
      atomicOr(reinterpret_cast<uint32_t*>(sg.moved(frame) + (i >> 6)) + ((i >> 5) & 1u), 1u << (i & 31u));
      if (to != from) move_sums(sg, frame, from, to, gl(f.out_patch)[i], gload(f.out_rgb + i));
    }
  }
}
}  // namespace

// Both filters in one pass, after the geometry filter — a thread per wave of 64 points, which walks the wave's moved
// bits: once all moved points have taken their sums along (k_smooth_apply_geometry), both cells of every point that
// changed cell are re-examined: the mixed bit is set or cleared, and flags are
// painted for a cell that has become mixed (flags of a cell that no longer is stay — they only make a point look at
// its neighbourhood in vain — and are un-painted with the rest by k_smooth_clear, which the painted bit tells).
template <bool kPow2>
__global__ __launch_bounds__(256) void k_smooth_moved_mark(const DevFrame* __restrict__ frames, uint32_t first, SmoothGrid sg,
                                                           GridDims gd) {
  const DevFrame& f = frames[first + blockIdx.y];
  const uint32_t n = min(*gl(f.n_points), f.capacity);
  const uint32_t wave = blockIdx.x * 256u + threadIdx.x;
  if (wave * 64u >= n) return;
  const uint64_t moved = sg.moved(blockIdx.y)[wave];
  if (!moved) return;
  uint64_t painted[2] = {0ull, 0ull};                        // flags painted from here, around the cell a point left / entered
  for (uint64_t mask = moved; mask; mask &= mask - 1ull) {
    const uint32_t b = (uint32_t)__builtin_ctzll(mask), i = wave * 64u + b;
    const vpcc_point3 pt = gload(f.out_xyz + i);
    const uint32_t was = sg.old_keys(blockIdx.y)[i], now = cell_key<kPow2>(pt.x, pt.y, pt.z, gd);
    if (now == was) continue;
    // (both cells' loads before either's stores)
    SmoothCell c2[2];
    SmoothColorCell v2[2];
#pragma unroll
    for (int side = 0; side < 2; ++side) {
      c2[side] = gload(sg.cells(blockIdx.y) + (side ? now : was));
      v2[side] = gload(sg.color_cells(blockIdx.y) + (side ? now : was));
    }
#pragma unroll
    for (int side = 0; side < 2; ++side) {
      const uint32_t key = side ? now : was;
      SmoothCell* cell = sg.cells(blockIdx.y) + key;
      const SmoothCell& c = c2[side];
      const SmoothColorCell& v = v2[side];
      const uint32_t count = v.count & kCountMask;           // (the colour cell's count is the one that moved)
      const bool mixed = sums_mixed(count, c.sp2, c.sp);
      // several moved points may share a cell: all compute the same from the same (final) sums
      if (mixed && !(c.mixed & kSmoothPainted)) {
        paint_flags(sg, blockIdx.y, key, gd.w, 1);
        painted[side] |= 1ull << b;
        *sg.frame_dirty(blockIdx.y) = 1u;                       // the spans with something to do have to be found again
      }
      const uint32_t bits = (mixed ? kSmoothMixed : 0u) | ((mixed || (c.mixed & kSmoothPainted)) ? kSmoothPainted : 0u);
      if (bits != c.mixed) cell->mixed = bits;
      const uint32_t cnt_bits = count | (mixed ? kColorCellMixed : 0u);
      if (cnt_bits != v.count) sg.color_cells(blockIdx.y)[key].count = cnt_bits;
    }
  }
  sg.moved_painted(blockIdx.y)[2u * wave] = painted[0];
  sg.moved_painted(blockIdx.y)[2u * wave + 1u] = painted[1];
}

namespace {
template <bool kPow2>
__device__ __forceinline__ void smooth_apply_color_point(const DevFrame& f, uint32_t frame, uint32_t i, const SmoothGrid& sg,
                                                         const GridDims& gd, uint32_t Ts, uint32_t Td, bool both,
                                                         const uint32_t (&p)[3]) {
  const uint32_t S[3] = {corner_coord<kPow2>(p[0], gd), corner_coord<kPow2>(p[1], gd), corner_coord<kPow2>(p[2], gd)};
  Hood h;
  const bool any = both ? load_hood(reinterpret_cast<const unsigned char*>(sg.color_cells(frame)), sizeof(SmoothColorCell), S, gd.w, h)
                        : load_hood(reinterpret_cast<const unsigned char*>(sg.cells(frame)), sizeof(SmoothCell), S, gd.w, h);
  if (!any) return;
  int64_t wt[3][2];
#pragma unroll
  for (int a = 0; a < 3; ++a) axis_weights(p[a], S[a], gd.G, wt[a]);
  const vpcc_color3 col = gload(f.out_rgb + i);
  const int32_t cl[3] = {col.r, col.g, col.b};
  // the point's own cell is one of the eight: index of (q - s) per axis
  const uint32_t qx = cell_coord<kPow2>(p[0], gd), qy = cell_coord<kPow2>(p[1], gd), qz = cell_coord<kPow2>(p[2], gd);
  const int own = (int)((qx + 1u - S[0]) | ((qy + 1u - S[1]) << 1) | ((qz + 1u - S[2]) << 2));
  u32x4 cc = h.c[0];
#pragma unroll
  for (int d = 1; d < 8; ++d) if (d == own) cc = h.c[d];
  // the cells' means, floor(sum / count) per channel (oracle/vpcc_smoothing_spec.h): sums of bytes, below 2^24 as long as a
  // cell holds no more than 65 537 points (beyond: VPCC_ERR_UNSUPPORTED, k_smooth_mark)
  const uint32_t own_count = cc.x & kCountMask;
  const float own_inv = 1.0f / (float)own_count;
  const int32_t mc[3] = {(int32_t)mean_small(cc.y, own_count, own_inv), (int32_t)mean_small(cc.z, own_count, own_inv),
                         (int32_t)mean_small(cc.w, own_count, own_inv)};
  int64_t num[3] = {0, 0, 0}, den = 0;
  bool mixed = false;
  if (small_weights(wt)) {
    // (32-bit weights, one multiply-add per cell and value, see the geometry filter; no branches: a cell that does not take part —
    // empty, or its mean too far from the own cell's — has the weight 0)
    uint32_t W32[8];
    hood_weights32(wt, W32);
    uint64_t un[4] = {0, 0, 0, 0};
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      const u32x4 c = h.c[d];
      const uint32_t count = c.x & kCountMask;
      bool use = count != 0u;
      if (d != own) {
        const uint32_t cnt1 = count ? count : 1u;
        const float inv = 1.0f / (float)cnt1;
        const int32_t m0 = (int32_t)mean_small(c.y, cnt1, inv) - mc[0], m1 = (int32_t)mean_small(c.z, cnt1, inv) - mc[1],
                      m2 = (int32_t)mean_small(c.w, cnt1, inv) - mc[2];
        const uint32_t diff = (uint32_t)((m0 < 0 ? -m0 : m0) + (m1 < 0 ? -m1 : m1) + (m2 < 0 ? -m2 : m2));
        use = use && diff <= Td;
      }
      const uint32_t W = use ? W32[d] : 0u;
      un[0] += (uint64_t)W * c.y; un[1] += (uint64_t)W * c.z; un[2] += (uint64_t)W * c.w;
      un[3] += (uint64_t)W * count;
      mixed |= use && (c.x & kSmoothCountMixed) != 0;
    }
    num[0] = (int64_t)un[0]; num[1] = (int64_t)un[1]; num[2] = (int64_t)un[2]; den = (int64_t)un[3];
  } else {
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      const u32x4 c = h.c[d];
      const uint32_t count = c.x & kCountMask;
      if (!count) continue;
      if (d != own) {
        const float inv = 1.0f / (float)count;
        const int32_t m0 = (int32_t)mean_small(c.y, count, inv) - mc[0], m1 = (int32_t)mean_small(c.z, count, inv) - mc[1],
                      m2 = (int32_t)mean_small(c.w, count, inv) - mc[2];
        const uint32_t diff = (uint32_t)((m0 < 0 ? -m0 : m0) + (m1 < 0 ? -m1 : m1) + (m2 < 0 ? -m2 : m2));
        if (diff > Td) continue;
      }
      const int64_t W = wt[0][d & 1] * wt[1][(d >> 1) & 1] * wt[2][d >> 2];
      num[0] += W * c.y; num[1] += W * c.z; num[2] += W * c.w;
      den += W * count;
      mixed |= (c.x & kSmoothCountMixed) != 0;
    }
  }
  if (!mixed || den <= 0) return;
  int64_t m[3], dist = 0;
  const int64_t A[3] = {2 * num[0] + den, 2 * num[1] + den, 2 * num[2] + den}, den2 = 2 * den;
  if (small_operands(A[0] | A[1] | A[2], den2)) {
    const double inv = 1.0 / (double)den2;
#pragma unroll
    for (int a = 0; a < 3; ++a) m[a] = div_small(A[a], den2, inv);
  } else {
#pragma unroll
    for (int a = 0; a < 3; ++a) m[a] = A[a] / den2;
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int64_t d = cl[a] - m[a];
    dist += d < 0 ? -d : d;
  }
  if (dist >= (int64_t)Ts) {
    vpcc_color3 o;
    o.r = (uint8_t)m[0]; o.g = (uint8_t)m[1]; o.b = (uint8_t)m[2];
    gstore(f.out_rgb + i, o);
  }
}
}  // namespace

// Which spans of kSmoothListSpan points have anything to do: those whose list holds a cell with a mixed cell among the
// 27 around it — one span in six.  A wave sees to four lists, their loads issued together.
__global__ __launch_bounds__(256) void k_smooth_spans(const DevFrame* __restrict__ frames, uint32_t first, SmoothGrid sg) {
  // (the frame's "a cell has become mixed behind the geometry filter" word starts at zero: k_smooth_moved_mark, k_smooth_apply)
  if (blockIdx.x == 0 && threadIdx.x == 0) *sg.frame_dirty(blockIdx.y) = 0u;
  const DevFrame& f = frames[first + blockIdx.y];
  const uint32_t n = min(*gl(f.n_points), f.capacity);
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t s0 = (blockIdx.x * 4u + (threadIdx.x >> 6)) * 4u;
  if (s0 * kSmoothListSpan >= n) return;
  const uint32_t* counts = sg.list_counts(blockIdx.y);
  const unsigned char* near = sg.near(blockIdx.y);
  uint32_t cnt[4], key[4];
#pragma unroll
  for (uint32_t k = 0; k < 4; ++k) cnt[k] = (s0 + k) * kSmoothListSpan < n ? gl(counts)[s0 + k] : 0u;
#pragma unroll
  for (uint32_t k = 0; k < 4; ++k) key[k] = lane < cnt[k] ? gl(sg.lists(blockIdx.y))[(size_t)(s0 + k) * kSmoothListLen + lane] : 0u;
  bool any[4];
#pragma unroll
  for (uint32_t k = 0; k < 4; ++k) any[k] = lane < cnt[k] && gl(near)[key[k]] != 0;
#pragma unroll
  for (uint32_t k = 0; k < 4; ++k) {
    for (uint32_t j = lane + 64u; j < cnt[k]; j += 64u)      // (a list of more than 64 cells)
      any[k] = any[k] || gl(near)[gl(sg.lists(blockIdx.y))[(size_t)(s0 + k) * kSmoothListLen + j]] != 0;
    const uint64_t mask = __ballot(any[k]);
    if (lane == k && (s0 + k) * kSmoothListSpan < n) sg.span_flags(blockIdx.y)[s0 + k] = mask ? 1u : 0u;
  }
}

// The filters.  A wave sees to 512 points of a span that has anything to do (k_smooth_spans; the others' waves leave at
// once), two quads per thread: their positions, then the neighbourhood flags of the eight points, are fetched together
// (the kernels are chains of dependent loads — positions, flags, cells — and want many of them in flight; a wave that
// went through several spans one after the other left the device empty: 0.26 / 0.37 ms instead of 0.16 / 0.24).
// Flagged points come in runs (patch boundaries), four to a lane — so the wave queues them in LDS and works on them
// lane <-> point, 64 at a time: the arithmetic of a flagged point (eight cells, 64-bit weights, three 64-bit divisions)
// is several hundred instructions, which a lane with one flagged point among idle neighbours would make the whole wave
// wait for.
constexpr uint32_t kApplyQuads = 2, kApplyUnit = kApplyQuads * 256u;     // points per wave
template <bool kPow2, bool kColor>
__global__ __launch_bounds__(256) void k_smooth_apply(const DevFrame* __restrict__ frames, uint32_t first, SmoothGrid sg,
                                                      GridDims gd, uint32_t T0, uint32_t T1, bool both) {
  __shared__ uint2 s_queue[4][kApplyUnit];
  const DevFrame& f = frames[first + blockIdx.y];
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const uint32_t base = (blockIdx.x * 4u + wave) * kApplyUnit;
  if (base >= f.capacity) return;
  const uint32_t active = gl(sg.span_flags(blockIdx.y))[base / kSmoothListSpan];     // (a span beyond the frame's points: whatever)
  // the colour filter of a pass that serves both: cells may have become mixed behind the geometry filter (k_smooth_moved_mark
  // says so per frame, rarely) — then the span's list is looked through again here, as k_smooth_spans did
  const uint32_t dirty = kColor && both ? *gl(sg.frame_dirty(blockIdx.y)) : 0u;
  const uint32_t n = min(*gl(f.n_points), f.capacity);
  if (base >= n) return;
  if (!active) {
    if (!dirty) return;
    const uint32_t span = base / kSmoothListSpan, cnt = gl(sg.list_counts(blockIdx.y))[span];
    const uint32_t* list = sg.lists(blockIdx.y) + (size_t)span * kSmoothListLen;
    const unsigned char* near = sg.near(blockIdx.y);
    bool any = false;
    for (uint32_t j = lane; j < cnt; j += 64u) any = any || gl(near)[gl(list)[j]] != 0;
    if (__ballot(any) == 0) return;
  }
  const unsigned char* flags = sg.flags(blockIdx.y);
  QuadXyz raw[kApplyQuads];
#pragma unroll
  for (uint32_t h = 0; h < kApplyQuads; ++h) {
    const uint32_t i0 = base + h * 256u + lane * 4u;
    raw[h] = QuadXyz{};
    if (i0 < n) raw[h] = load_quad_xyz(f.out_xyz, i0 >> 2);
  }
  uint32_t x[kApplyQuads][4], y[kApplyQuads][4], z[kApplyQuads][4];
  unsigned char flag[kApplyQuads][4];
#pragma unroll
  for (uint32_t h = 0; h < kApplyQuads; ++h) {
    unpack_xyz(raw[h], x[h], y[h], z[h]);
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j)
      flag[h][j] = base + h * 256u + lane * 4u + j < n ? gl(flags)[flag_index<kPow2>(x[h][j], y[h][j], z[h][j], gd)] : (unsigned char)0;
  }
  uint32_t total = 0;
#pragma unroll
  for (uint32_t h = 0; h < kApplyQuads; ++h)
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
      const uint64_t mask = __ballot(flag[h][j] != 0);
      if (flag[h][j]) s_queue[wave][total + mbcnt(mask)] = make_uint2(x[h][j] | (y[h][j] << 16), z[h][j] | ((h * 256u + lane * 4u + j) << 16));
      total += (uint32_t)__builtin_popcountll(mask);
    }
  __builtin_amdgcn_wave_barrier();                           // (LDS operations of one wave execute in order)
  for (uint32_t e = lane; e < total; e += 64u) {
    const uint2 ent = s_queue[wave][e];
    const uint32_t p[3] = {ent.x & 0xFFFFu, ent.x >> 16, ent.y & 0xFFFFu};
    const uint32_t i = base + (ent.y >> 16);
    if (kColor) smooth_apply_color_point<kPow2>(f, blockIdx.y, i, sg, gd, T0, T1, both, p);
    else smooth_apply_geometry_point<kPow2>(f, blockIdx.y, i, sg, gd, T0, both, p);
  }
}

// ---- launches ------------------------------------------------------------------------------------------------------
namespace {
GridDims grid_dims(uint32_t w, uint32_t G) {
  GridDims d{G, w, 0u};
  while ((1u << d.sh) < G) ++d.sh;
  return d;
}
bool pow2(uint32_t G) { return (G & (G - 1u)) == 0u; }
}  // namespace

void launch_smooth_stats(const DevFrame* d_frames, uint32_t first, uint32_t count, uint32_t max_points, SmoothGrid sg,
                         uint32_t w, uint32_t G, uint32_t mode, void* stream) {
  if (!count || !max_points) return;
  const dim3 grid((max_points + 4u * kSmoothListSpan - 1) / (4u * kSmoothListSpan), count), block(256);
  const GridDims gd = grid_dims(w, G);
  hipStream_t s = (hipStream_t)stream;
#define VPCC_STATS(M)                                                                                           \
  if (pow2(G)) hipLaunchKernelGGL((k_smooth_stats<M, true>), grid, block, 0, s, d_frames, first, sg, gd);       \
  else hipLaunchKernelGGL((k_smooth_stats<M, false>), grid, block, 0, s, d_frames, first, sg, gd);
  if (mode == 0u) { VPCC_STATS(0) } else if (mode == 1u) { VPCC_STATS(1) } else { VPCC_STATS(2) }
#undef VPCC_STATS
}
void launch_smooth_apply_geometry(const DevFrame* d_frames, uint32_t first, uint32_t count, uint32_t max_points,
                                  SmoothGrid sg, uint32_t w, uint32_t G, uint32_t T, bool both, void* stream) {
  if (!count || !max_points) return;
  const uint32_t per_block = 4u * kApplyUnit;
  const dim3 grid((max_points + per_block - 1) / per_block, count), block(256);
  const GridDims gd = grid_dims(w, G);
  if (pow2(G)) hipLaunchKernelGGL((k_smooth_apply<true, false>), grid, block, 0, (hipStream_t)stream, d_frames, first, sg, gd, T, 0u, both);
  else hipLaunchKernelGGL((k_smooth_apply<false, false>), grid, block, 0, (hipStream_t)stream, d_frames, first, sg, gd, T, 0u, both);
}
void launch_smooth_apply_color(const DevFrame* d_frames, uint32_t first, uint32_t count, uint32_t max_points,
                               SmoothGrid sg, uint32_t w, uint32_t G, uint32_t Ts, uint32_t Td, bool both, void* stream) {
  if (!count || !max_points) return;
  const uint32_t per_block = 4u * kApplyUnit;
  const dim3 grid((max_points + per_block - 1) / per_block, count), block(256);
  const GridDims gd = grid_dims(w, G);
  if (pow2(G)) hipLaunchKernelGGL((k_smooth_apply<true, true>), grid, block, 0, (hipStream_t)stream, d_frames, first, sg, gd, Ts, Td, both);
  else hipLaunchKernelGGL((k_smooth_apply<false, true>), grid, block, 0, (hipStream_t)stream, d_frames, first, sg, gd, Ts, Td, both);
}
void launch_smooth_moved(const DevFrame* d_frames, uint32_t first, uint32_t count, uint32_t max_points, SmoothGrid sg,
                         uint32_t w, uint32_t G, void* stream) {
  if (!count || !max_points) return;
  const dim3 grid(((max_points + 63) / 64 + 255) / 256, count), block(256);
  const GridDims gd = grid_dims(w, G);
  if (pow2(G)) hipLaunchKernelGGL(k_smooth_moved_mark<true>, grid, block, 0, (hipStream_t)stream, d_frames, first, sg, gd);
  else hipLaunchKernelGGL(k_smooth_moved_mark<false>, grid, block, 0, (hipStream_t)stream, d_frames, first, sg, gd);
}
void launch_smooth_spans(const DevFrame* d_frames, uint32_t first, uint32_t count, uint32_t max_points, SmoothGrid sg, void* stream) {
  if (!count || !max_points) return;
  const uint32_t per_block = 16u * kSmoothListSpan;
  hipLaunchKernelGGL(k_smooth_spans, dim3((max_points + per_block - 1) / per_block, count), dim3(256), 0, (hipStream_t)stream, d_frames,
                     first, sg);
}
void launch_smooth_mark(const DevFrame* d_frames, uint32_t first, uint32_t count, uint32_t max_points, SmoothGrid sg,
                        uint32_t w, void* stream) {
  if (!count || !max_points) return;
  const uint32_t per_block = 4u * kSmoothListSpan;
  hipLaunchKernelGGL(k_smooth_mark, dim3((max_points + per_block - 1) / per_block, count), dim3(256), 0, (hipStream_t)stream, d_frames,
                     first, sg, w);
}
void launch_smooth_clear(const DevFrame* d_frames, uint32_t first, uint32_t count, uint32_t max_points, SmoothGrid sg,
                         uint32_t w, uint32_t G, bool both, void* stream) {
  if (!count || !max_points) return;
  const uint32_t per_block = 4u * kSmoothListSpan;
  const dim3 grid((max_points + per_block - 1) / per_block, count), block(256);
  const GridDims gd = grid_dims(w, G);
  if (pow2(G)) hipLaunchKernelGGL(k_smooth_clear<true>, grid, block, 0, (hipStream_t)stream, d_frames, first, sg, gd, both);
  else hipLaunchKernelGGL(k_smooth_clear<false>, grid, block, 0, (hipStream_t)stream, d_frames, first, sg, gd, both);
}

}  // namespace vpcc
