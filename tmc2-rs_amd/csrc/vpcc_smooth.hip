// vpcc_smooth.hip — grid-based geometry and colour smoothing on gfx950.
//
// The reference implements neither (every hook is unimplemented!(): src/decoder.rs:291-299, 630-658,
// src/codec.rs:498-500); the behaviour is this repository's own integer specification "gs1"/"cs1",
// written down in oracle/vpcc_smoothing_spec.h and tested bit for bit against its CPU form.
//
// Four kernels per filter, one thread per point:
//   k_smooth_stats : per occupied grid cell {count, 3 sums, sum of patch indices, sum of their squares} — every field
//                    a sum, one atomic instruction per (wave, cell) — into a dense w^3 grid that is all-zero between
//                    launches; keeps every point's cell index;
//   k_smooth_mark  : paints a byte flag around every cell that mixes patches;
//   k_smooth_apply : one flag load; only flagged points read their 2x2x2 cells -> integer trilinear weights
//                    -> centroid / mean, thresholded replacement in place (a thread reads only its own point
//                    and the grid);
//   k_smooth_clear : un-paints the flags and zeroes exactly the touched cells again.
#include <hip/hip_runtime.h>

#include "vpcc_device.hpp"
#include "vpcc_devfn.hpp"

namespace vpcc {

namespace {

// min(p / G, w - 1) for a 16-bit coordinate without an integer division: (p + 0.5) * fl(1/G) is within
// 2^-24 * p/G of (p + 0.5)/G, which is at least 0.5/G away from every integer — more than that error as
// long as p * G < 2^23 — so truncation yields floor(p / G) exactly; larger grids take the division.
__device__ __forceinline__ uint32_t cell_coord(uint32_t p, uint32_t G, uint32_t w) {
  uint32_t q;
  if (G < 128u) {                                        // p < 2^16  =>  p * G < 2^23
    const float r = 1.0f / (float)G;
    q = (uint32_t)__builtin_fmaf((float)p, r, 0.5f * r);
  } else {
    q = p / G;
  }
  return q < w ? q : w - 1u;
}

__device__ __forceinline__ void axis_setup(uint32_t p, uint32_t G, uint32_t w, int32_t& s, int64_t wt[2]) {
  const int32_t q = (int32_t)cell_coord(p, G, w), r = (int32_t)p - q * (int32_t)G, h = (int32_t)(G / 2u);
  s = (r < h) ? q - 1 : q;
  const int32_t t = 2 * ((int32_t)p - (s * (int32_t)G + h)) + 1;
  wt[0] = 2 * (int64_t)G - t;
  wt[1] = t;
}

}  // namespace

// A point / a colour with ONE 8-byte load (4-byte aligned) instead of three 2-byte / 1-byte loads: element i starts
// 0 or 2 (points), 0-3 (colours) bytes into the aligned pair of dwords, which therefore reaches up to 2 / 5 bytes
// past the element — into the next element, or into the padding every output array ends with (vpcc_runtime.hip).
__device__ __forceinline__ vpcc_point3 load_point(const vpcc_point3* base, uint32_t i) {
  typedef uint32_t v2 __attribute__((ext_vector_type(2)));
  const uint32_t off = i * 6u;
  const v2 d = *(const VPCC_GLOBAL v2*)((const VPCC_GLOBAL unsigned char*)base + (off & ~3u));
  const uint64_t v = (((uint64_t)d.y << 32) | d.x) >> ((off & 2u) * 8u);
  vpcc_point3 p;
  p.x = (uint16_t)v; p.y = (uint16_t)(v >> 16); p.z = (uint16_t)(v >> 32);
  return p;
}
__device__ __forceinline__ vpcc_color3 load_color(const vpcc_color3* base, uint32_t i) {
  typedef uint32_t v2 __attribute__((ext_vector_type(2)));
  const uint32_t off = i * 3u;
  const v2 d = *(const VPCC_GLOBAL v2*)((const VPCC_GLOBAL unsigned char*)base + (off & ~3u));
  const uint64_t v = (((uint64_t)d.y << 32) | d.x) >> ((off & 3u) * 8u);
  vpcc_color3 c;
  c.r = (uint8_t)v; c.g = (uint8_t)(v >> 8); c.b = (uint8_t)(v >> 16);
  return c;
}
typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
// value of lane - kShift within the lane's row of 16 (DPP row_shr); `outside` where there is no such lane
// value of the lane's partner inside its quad of four lanes (DPP quad_perm)
template <int kCtrl>
__device__ __forceinline__ uint32_t qperm(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, kCtrl, 0xF, 0xF, false);
}
template <int kShift>
__device__ __forceinline__ uint32_t row_shr(uint32_t v, uint32_t outside) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)outside, (int)v, 0x110 + kShift, 0xF, 0xF, false);
}

// Wave-wide sum / packed-16-bit max with DPP row shifts and row broadcasts (no LDS); the result is valid in lane 63.
template <int kCtrl, int kRowMask, bool kPkMax>
__device__ __forceinline__ uint32_t dpp_step(uint32_t v) {
  const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, kCtrl, kRowMask, 0xF, kRowMask == 0xF);
  if (kPkMax)
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, o), __builtin_bit_cast(u16x2, v)));
  return v + o;
}
template <bool kPkMax>
__device__ __forceinline__ uint32_t wave_reduce(uint32_t v) {
  v = dpp_step<0x111, 0xF, kPkMax>(v);
  v = dpp_step<0x112, 0xF, kPkMax>(v);
  v = dpp_step<0x114, 0xF, kPkMax>(v);
  v = dpp_step<0x118, 0xF, kPkMax>(v);
  v = dpp_step<0x142, 0xA, kPkMax>(v);
  v = dpp_step<0x143, 0xC, kPkMax>(v);
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// mode 0: sums of coordinates (geometry); mode 1: sums of R,G,B (colour).
// Points arrive in emission order, so the 64 points of a wave fall into a handful of cells (a block row of
// 16 pixels spans two cells of size 8).  The wave reduces its points per distinct cell first and issues ONE
// atomic instruction per cell (four 64-bit adds, nothing returned: every field of a cell is a sum).  The grids are all-zero between
// launches: every point's cell index is kept, and k_smooth_clear zeroes exactly those cells afterwards — no
// dense memset (50 MB per frame at w = 128) per launch.
// Measured alternatives, per 32 S-longdress frames (this form: 0.31 ms): one set of atomics per point 17.5 ms; a
// list of touched cells fed by RETURNING atomics and one list counter per frame 8.9 ms; runs of equal cells
// reduced by a segmented DPP scan inside rows of 16 lanes with six atomics per run 2.7 ms (30 M atomics: the L2
// retires ~12 per ns); the same merged through an LDS table with ds_cmpst/ds_add per workgroup 0.62-0.70 ms;
// every point added with LDS atomics into an 8x8x8-cell window of the grid held in LDS, occupied cells flushed
// with six global atomics each 1.19 ms (lanes of one instruction that hit the same LDS word are served one at a
// time, ~4 cycles each).  With {max(65535 - patch), max(patch)} as two 32-bit max operations next to the adds (two
// atomic instructions per cell) 0.29-0.30 ms, with all-sum cells (one instruction) 0.245; collecting up to 16 cells of a
// wave into ONE atomic instruction changed nothing further, nor did removing reductions from the loop.  Where the
// 0.245 ms go (ablations): loads + key store 0.06 (at the memory rate), the de-duplication loop 0.10, the atomics 0.08.
// Every workgroup handles kSmoothChunks chunks of 256 consecutive points: with one chunk per workgroup these
// kernels were bound by the rate at which workgroups can be launched (1.3-1.7 resident waves per SIMD on average,
// VALU 23 % busy — tools/pmc_smooth.sh), not by anything they do.
constexpr uint32_t kSmoothChunks = 8;       // (statistics for both filters: 1 / 2 / 4 / 8 / 16 / 32 chunks 1.03 / 0.95 / 0.876 / 0.865 / 0.869 / 0.897 ms)
constexpr uint32_t kApplyChunks = 4;    // points per thread of the apply kernels (2: 2.41 ms for the step, 8: 2.25, 16: 3.77; 4: 2.24)
// What a thread reads of its point: fetched one chunk ahead of its use (k_smooth_stats).
struct StatInput { vpcc_point3 p; vpcc_color3 col; uint32_t patch; };
__device__ __forceinline__ StatInput load_stat_input(const DevFrame& f, uint32_t i, uint32_t n, uint32_t mode) {
  StatInput in{};
  if (i < n) {
    in.p = load_point(f.out_xyz, i);
    if (mode) in.col = load_color(f.out_rgb, i);
    in.patch = gl(f.out_patch)[i];
  }
  return in;
}
__device__ __forceinline__ void smooth_stats_chunk(const DevFrame& f, uint32_t frame, uint32_t chunk, uint32_t n,
                                                   SmoothGrid sg, uint32_t w, uint32_t G, uint32_t mode, const StatInput& in) {
  const uint32_t i = chunk * 256u + threadIdx.x;
  if (chunk * 256u >= n) return;
  const bool active = i < n;
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t key = 0xFFFFFFFFu, v[3] = {0, 0, 0}, patch = 0, cslot = 0, c01 = 0, c2 = 0;
  if (active) {
    const vpcc_point3 p = in.p;
    const uint32_t cx = cell_coord(p.x, G, w), cy = cell_coord(p.y, G, w), cz = cell_coord(p.z, G, w);
    key = (cz * w + cy) * w + cx;
    cslot = (cx & 3u) | ((cy & 3u) << 2) | ((cz & 3u) << 4);
    v[0] = p.x; v[1] = p.y; v[2] = p.z;
    if (mode) {
      const vpcc_color3 col = in.col;
      if (mode == 1u) { v[0] = col.r; v[1] = col.g; v[2] = col.b; }
      else { c01 = col.r | ((uint32_t)col.g << 16); c2 = col.b; }     // mode 2: colour sums next to the coordinate sums
    }
    patch = in.patch;
    sg.keys(frame)[i] = key;                                 // for the apply kernels (one dword per point: cheaper than the point)
  }
  uint32_t* list = sg.lists(frame) + (size_t)(i >> 6) * kSmoothListLen;   // this wave's cell list
  SmoothCell* grid = sg.cells(frame);
  SmoothColorCell* cgrid = sg.color_cells(frame);           // (mode 2 only)
  uint64_t todo = __ballot(active);
  // Two of the three sums share one reduction when no value of the wave exceeds 1023 (64 x 1023 < 2^16): always for
  // colours, and for coordinates of 10-bit content.
  const bool pack = __ballot((v[0] | v[1]) > 1023u) == 0;
  const uint32_t v01 = v[0] | (v[1] << 16);
  // Fast path: no loop at all.  The cells of 64 consecutive points are neighbours in the grid, so the low two bits of
  // each cell coordinate give every cell of the wave a slot of its own among 64 (two cells collide only if they are a
  // multiple of four cells apart in every axis).  The first lane of every run of equal cell indices writes the index
  // (and its patch, and its lane number) to its slot and every lane reads the slot back: if every lane finds its own
  // index and patch, the slots are collision-free and every cell's points share one patch; the lane whose number
  // stayed in a slot is elected for it.  Then every lane adds its values to its slot with LDS atomics (lanes of one
  // slot are served one after the other, slots side by side; the point count rides in the upper bits of the third
  // sum), the elected lanes line their slots up in a list, and lane 4 j + t carries word t of the j-th cell: the four
  // words of a cell leave as ONE 32-byte atomic request.  A wave with a collision, or with points of two patches in
  // one cell (rare), takes the loop below.  Per 128 S-longdress frames: geometry 0.93 -> 0.58 ms, colour 1.02 -> 0.68
  // (profiles/r03/ab_smooth_stats.txt; with four separate atomic instructions issued by the elected lanes 1.73 ms —
  // it is the number of atomic REQUESTS that counts; numbering the cells in a loop first and adding into numbered
  // slots 0.74 ms).
  {
    struct StatTable { uint32_t key[64], pl[64], lid[64], acc[5][64]; };
    __shared__ StatTable s_tab[4];
    StatTable& T = s_tab[threadIdx.x >> 6];
    const uint32_t slot = cslot;
    const bool head = active && row_shr<1>(key, ~key) != key;
    if (head) {
      T.key[slot] = key; T.pl[slot] = patch; T.lid[slot] = lane;
      T.acc[0][slot] = 0u; T.acc[1][slot] = 0u; T.acc[2][slot] = 0u;
      if (mode == 2u) { T.acc[3][slot] = 0u; T.acc[4][slot] = 0u; }
    }
    __builtin_amdgcn_wave_barrier();                         // (LDS operations of one wave execute in order)
    const bool clash = active && (T.key[slot] != key || T.pl[slot] != patch);
    if (__ballot(clash) == 0) {
      // Lanes of one slot are served one after the other by the LDS (a CU has ONE; ten lanes per slot and five adds made
      // it the busiest unit of the kernel), so neighbours with the same cell first add up in registers: lane pairs, then
      // pairs of pairs (two DPP steps inside every quad of lanes; a third step across quads cost more than it saved);
      // the lanes that were added to a neighbour stay out.  Statistics for both filters 0.97 -> 0.875 ms.
      uint32_t a[5] = {pack ? v01 : v[0], pack ? 0u : v[1], v[2] + (1u << 22), c01, c2};   // 64 x 65535 < 2^22: the count above the sum
      bool live = active;
      {
        const bool same = qperm<0xB1>(key) == key;               // lane ^ 1
#pragma unroll
        for (int q = 0; q < 5; ++q) { const uint32_t o = qperm<0xB1>(a[q]); if (same && !(lane & 1u)) a[q] += o; }
        if (same && (lane & 1u)) live = false;
        const bool same2 = qperm<0x4E>(key) == key;              // lane ^ 2 (lanes 0 and 2 of a quad: never given away above)
#pragma unroll
        for (int q = 0; q < 5; ++q) { const uint32_t o = qperm<0x4E>(a[q]); if (same2 && !(lane & 3u)) a[q] += o; }
        if (same2 && (lane & 3u) == 2u) live = false;
      }
      if (live) {
        atomicAdd(&T.acc[0][slot], a[0]);
        if (!pack) atomicAdd(&T.acc[1][slot], a[1]);
        atomicAdd(&T.acc[2][slot], a[2]);
        if (mode == 2u) { atomicAdd(&T.acc[3][slot], a[3]); atomicAdd(&T.acc[4][slot], a[4]); }   // 64 x 255 < 2^16
      }
      __builtin_amdgcn_wave_barrier();
      const bool elected = head && T.lid[slot] == lane;
      const uint64_t em = __ballot(elected);
      __builtin_amdgcn_wave_barrier();
      const uint32_t ncell = (uint32_t)__builtin_popcountll(em);
      const uint32_t rank = (uint32_t)__builtin_popcountll(em & ((1ull << lane) - 1ull));
      if (elected) T.lid[rank] = slot;                        // (the election is over: reuse)
      // the cells this wave touched, for the passes that follow (mark, clear): a list of 16, or — more cells than
      // that — "see the cell index of every point"
      if (ncell <= kSmoothListLen) {
        if (elected) list[rank] = key;
        if (lane >= ncell && lane < kSmoothListLen) list[lane] = kSmoothNoCell;
      } else if (lane == 0) {
        list[0] = kSmoothListOverflow;
      }
      __builtin_amdgcn_wave_barrier();
      for (uint32_t base = 0; base < ncell; base += 16u) {
        const uint32_t j = base + (lane >> 2), t = lane & 3u;
        if (j < ncell) {
          const uint32_t sl = T.lid[j];
          const uint32_t k = T.key[sl], pl = T.pl[sl], a0 = T.acc[0][sl], a2 = T.acc[2][sl];
          const uint32_t cnt = a2 >> 22, s2 = a2 & 0x3FFFFFu;
          const uint32_t s0 = pack ? a0 & 0xFFFFu : a0, s1 = pack ? a0 >> 16 : T.acc[1][sl];
          const uint32_t sp = cnt * pl;
          const uint64_t val = t == 0 ? (uint64_t)cnt | ((uint64_t)s0 << 32)
                             : t == 1 ? (uint64_t)s1 | ((uint64_t)s2 << 32)
                             : t == 2 ? (uint64_t)sp * pl : (uint64_t)sp;
          atomicAdd(reinterpret_cast<unsigned long long*>(grid + k) + t, (unsigned long long)val);
        }
      }
      if (mode == 2u)                                          // the colour cells: lane 2 j + t carries word t of the j-th
        for (uint32_t base = 0; base < ncell; base += 32u) {
          const uint32_t j = base + (lane >> 1), t = lane & 1u;
          if (j < ncell) {
            const uint32_t sl = T.lid[j];
            const uint32_t rg = T.acc[3][sl];
            const uint64_t val = t == 0 ? (uint64_t)(T.acc[2][sl] >> 22) | ((uint64_t)(rg & 0xFFFFu) << 32)        // {count, R}
                                        : (uint64_t)(rg >> 16) | ((uint64_t)T.acc[4][sl] << 32);                    // {G, B}
            atomicAdd(reinterpret_cast<unsigned long long*>(cgrid + T.key[sl]) + t, (unsigned long long)val);
          }
        }
      return;
    }
  }
  if (lane == 0) list[0] = kSmoothListOverflow;             // (rare path: mark and clear take the cells from the key array)
  while (todo) {                                           // one trip per distinct cell of the wave
    const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)key, (int)__builtin_ctzll(todo));
    const bool in = active && key == k;
    const uint64_t mask = __ballot(in);
    uint32_t s0, s1;
    if (pack) {
      const uint32_t s01 = wave_reduce<false>(in ? v01 : 0u);
      s0 = s01 & 0xFFFFu; s1 = s01 >> 16;
    } else {
      s0 = wave_reduce<false>(in ? v[0] : 0u); s1 = wave_reduce<false>(in ? v[1] : 0u);
    }
    const uint32_t s2 = wave_reduce<false>(in ? v[2] : 0u);
    // sums of the patch indices and of their squares: the points a wave has in one cell nearly always belong to ONE
    // patch — then both follow from the count; else two more reductions
    const uint32_t cnt = (uint32_t)__builtin_popcountll(mask);
    const uint32_t pl = (uint32_t)__builtin_amdgcn_readlane((int)patch, (int)__builtin_ctzll(mask));
    uint32_t sp = cnt * pl;
    uint64_t sp2 = (uint64_t)sp * pl;
    if (__ballot(in && patch != pl) != 0) {
      sp = wave_reduce<false>(in ? patch : 0u);
      // squares of 16-bit indices: 64 of them fit 2^38 — low and high halves of p^2 are summed apart
      const uint32_t q = patch * patch;
      sp2 = (uint64_t)wave_reduce<false>(in ? q & 0xFFFFu : 0u) + ((uint64_t)wave_reduce<false>(in ? q >> 16 : 0u) << 16);
    }
    if (lane < 4u) {
      const uint64_t val = lane == 0 ? (uint64_t)cnt | ((uint64_t)s0 << 32)
                         : lane == 1 ? (uint64_t)s1 | ((uint64_t)s2 << 32)
                         : lane == 2 ? sp2 : (uint64_t)sp;
      atomicAdd(reinterpret_cast<unsigned long long*>(grid + k) + lane, (unsigned long long)val);
    }
    if (mode == 2u) {
      const uint32_t rg = wave_reduce<false>(in ? c01 : 0u), bb = wave_reduce<false>(in ? c2 : 0u);
      if (lane < 2u)
        atomicAdd(reinterpret_cast<unsigned long long*>(cgrid + k) + lane,
                  (unsigned long long)(lane == 0 ? (uint64_t)cnt | ((uint64_t)(rg & 0xFFFFu) << 32) : (uint64_t)(rg >> 16) | ((uint64_t)bb << 32)));
    }
    todo &= ~mask;
  }
}

__global__ __launch_bounds__(256) void k_smooth_stats(const DevFrame* __restrict__ frames, uint32_t first,
                                                      SmoothGrid sg, uint32_t w, uint32_t G, uint32_t mode) {
  const DevFrame& f = frames[first + blockIdx.y];
  const uint32_t n = min(*gl(f.n_points), f.capacity);
  StatInput cur = load_stat_input(f, blockIdx.x * kSmoothChunks * 256u + threadIdx.x, n, mode);
#pragma unroll 1
  for (uint32_t c = 0; c < kSmoothChunks; ++c) {
    const uint32_t chunk = blockIdx.x * kSmoothChunks + c;
    const StatInput nxt = c + 1u < kSmoothChunks ? load_stat_input(f, (chunk + 1u) * 256u + threadIdx.x, n, mode) : StatInput{};
    smooth_stats_chunk(f, blockIdx.y, chunk, n, sg, w, G, mode, cur);
    cur = nxt;
  }
}

namespace {
__device__ __forceinline__ bool cell_mixed(const SmoothCell& c) { return (uint64_t)c.count * c.sp2 != (uint64_t)c.sp * c.sp; }

// Writes `value` into the flag of every cell of the 3x3x3 block around cell `key` (clipped to the grid).
__device__ __forceinline__ void paint_flags(unsigned char* flags, uint32_t key, uint32_t w, unsigned char value) {
  const int32_t cx = (int32_t)(key % w), cy = (int32_t)((key / w) % w), cz = (int32_t)(key / (w * w));
  for (int32_t z = max(cz - 1, 0); z <= min(cz + 1, (int32_t)w - 1); ++z)
    for (int32_t y = max(cy - 1, 0); y <= min(cy + 1, (int32_t)w - 1); ++y)
      for (int32_t x = max(cx - 1, 0); x <= min(cx + 1, (int32_t)w - 1); ++x)
        flags[((size_t)z * w + y) * w + x] = value;
}
}  // namespace

// A filter changes a point only if one of the 2x2x2 cells around it holds points of more than one patch
// ("mixed"), and those eight cells always lie in the 3x3x3 block around the point's own cell.  This pass
// paints a byte flag on the 3x3x3 block around every mixed cell, so that the filter kernels decide with ONE
// byte load whether a point needs its eight 24-byte cells at all (few do: patch boundaries).
// Four consecutive points per thread (one 16-byte load of their cell indices): a quarter of the workgroups of a
// thread-per-point launch, which for these light passes was bound by workgroup dispatch (0.08 -> 0.0x ms).
// leader bit j: key j starts a run of equal cell indices (differs from the key before it; the first key of a row
// of 16 lanes always leads).  Keys beyond n read as 0xFFFFFFFF and lead nothing.
__device__ __forceinline__ uint32_t load_keys4(const uint32_t* keys, uint32_t i4, uint32_t n, uint32_t k[4]) {
  if (i4 + 3u < n) {
    const uint4 v = *reinterpret_cast<const uint4*>(keys + i4);
    k[0] = v.x; k[1] = v.y; k[2] = v.z; k[3] = v.w;
  } else {
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) k[j] = i4 + j < n ? keys[i4 + j] : 0xFFFFFFFFu;
  }
  const uint32_t before = row_shr<1>(k[3], ~k[0]);           // the previous thread's last key
  uint32_t lead = before != k[0] ? 1u : 0u;
#pragma unroll
  for (uint32_t j = 1; j < 4; ++j) lead |= (k[j] != k[j - 1u] ? 1u : 0u) << j;
#pragma unroll
  for (uint32_t j = 0; j < 4; ++j) if (k[j] == 0xFFFFFFFFu) lead &= ~(1u << j);
  return lead;
}

// The cells thread e of a frame's launch is responsible for: entry e of the frame's cell lists, or — where a wave's
// list says "overflow" — the cells of four of that wave's points, taken from the key array (leader bits as above).
// Returns the number of cells in k[] (0, 1 or up to 4).
__device__ __forceinline__ uint32_t listed_cells(const SmoothGrid& sg, uint32_t frame, uint32_t e, uint32_t n, uint32_t k[4]) {
  const uint32_t wave = e / kSmoothListLen, j = e % kSmoothListLen;
  if (wave * 64u >= n) return 0;
  const uint32_t* list = sg.lists(frame) + (size_t)wave * kSmoothListLen;
  if (list[0] != kSmoothListOverflow) {
    k[0] = list[j];
    return k[0] != kSmoothNoCell ? 1u : 0u;
  }
  static_assert(kSmoothListLen * 4u == 64u, "four points per thread in an overflow wave");
  uint32_t all[4];
  const uint32_t lead = load_keys4(sg.keys(frame), wave * 64u + j * 4u, n, all);
  uint32_t m = 0;
#pragma unroll
  for (uint32_t q = 0; q < 4; ++q) if ((lead >> q) & 1u) k[m++] = all[q];
  return m;
}

__global__ __launch_bounds__(256) void k_smooth_mark(const DevFrame* __restrict__ frames, uint32_t first, SmoothGrid sg,
                                                     uint32_t w) {
  const DevFrame& f = frames[first + blockIdx.y];
  const uint32_t n = min(*gl(f.n_points), f.capacity);
  if (blockIdx.x * 1024u >= n) return;                      // 256 entries = 16 waves = 1 024 points per workgroup
  uint32_t k[4];
  const uint32_t m = listed_cells(sg, blockIdx.y, blockIdx.x * 256u + threadIdx.x, n, k);
  for (uint32_t q = 0; q < m; ++q) {
    const SmoothCell c = gload(sg.cells(blockIdx.y) + k[q]);
    // The cells' sums are 32 bits wide, two to a 64-bit atomic add: beyond this many points in ONE cell a sum of 16-bit
    // values could carry into its neighbour (the specification's u32 sums would wrap instead) — reported, not smoothed over.
    if (c.count > kSmoothCellMaxPoints) atomicOr(f.error_flag, kErrorSmoothCellOverflow);
    if (cell_mixed(c)) {
      paint_flags(sg.flags(blockIdx.y), k[q], w, 1);
      (sg.cells(blockIdx.y) + k[q])->mixed = kSmoothMixed | kSmoothPainted;   // for the apply kernels: the 64-bit test once per cell, not per point
      if (sg.color_offset) {                                                   // both filters: the colour filter reads the colour cells only
        SmoothColorCell* cc = sg.color_cells(blockIdx.y) + k[q];
        cc->count = cc->count | kColorCellMixed;
      }
    }
  }
}

// Restores the all-zero state: every listed cell is un-painted if flags were painted around it, and zeroed (several
// waves list the same cell; the first to read it still sees the bit).  both: the colour cell too, and the cells that
// points were moved INTO (no wave listed those).
__device__ __forceinline__ void clear_cell(const SmoothGrid& sg, uint32_t frame, uint32_t key, uint32_t w, bool both) {
  SmoothCell* cell = sg.cells(frame) + key;
  const SmoothCell seen = gload(cell);
  if (!(seen.count | seen.s[0] | seen.s[1] | seen.s[2] | (uint32_t)seen.sp2 | (uint32_t)(seen.sp2 >> 32) | seen.sp | seen.mixed))
    return;                                                 // another entry of the same cell has cleared it
  if (both) {                                               // (a colour cell holds nothing its cell does not: count, sums of those points)
    uint4* cc = reinterpret_cast<uint4*>(sg.color_cells(frame) + key);
    *cc = make_uint4(0u, 0u, 0u, 0u);
  }
  if (seen.mixed & kSmoothPainted) paint_flags(sg.flags(frame), key, w, 0);
  uint4* c = reinterpret_cast<uint4*>(cell);
  c[0] = make_uint4(0u, 0u, 0u, 0u); c[1] = make_uint4(0u, 0u, 0u, 0u);
}
__global__ __launch_bounds__(256) void k_smooth_clear(const DevFrame* __restrict__ frames, uint32_t first, SmoothGrid sg,
                                                      uint32_t w, bool both) {
  const DevFrame& f = frames[first + blockIdx.y];
  const uint32_t n = min(*gl(f.n_points), f.capacity);
  if (blockIdx.x * 1024u >= n) return;
  uint32_t k[4];
  const uint32_t m = listed_cells(sg, blockIdx.y, blockIdx.x * 256u + threadIdx.x, n, k);
  for (uint32_t q = 0; q < m; ++q) clear_cell(sg, blockIdx.y, k[q], w, both);
  if (both) {
    // the cells moved points went INTO are in no list, and a wave whose list overflowed finds its cells through the
    // key array, where a moved point's entry has changed: the first of a wave's sixteen threads sees to its moved points
    const uint32_t e = blockIdx.x * 256u + threadIdx.x, wave = e / kSmoothListLen;
    if (e % kSmoothListLen == 0 && wave * 64u < n)
      for (uint64_t mask = sg.moved(blockIdx.y)[wave]; mask; mask &= mask - 1ull) {
        const uint32_t i = wave * 64u + (uint32_t)__builtin_ctzll(mask);
        const uint32_t now = sg.keys(blockIdx.y)[i], was = sg.old_keys(blockIdx.y)[i];
        clear_cell(sg, blockIdx.y, now, w, true);
        if (was != now) clear_cell(sg, blockIdx.y, was, w, true);
      }
  }
}

namespace {
struct Hood {                  // the 2x2x2 cells around a point
  SmoothCell c[8];
  bool inside[8];
};
// Loads the neighbourhood and tells whether any of its cells holds points of more than one patch: only then
// does a filter do anything (most points of a frame are far from a patch boundary and stop here).
__device__ __forceinline__ bool load_hood(const SmoothCell* grid, const int32_t s[3], uint32_t w, Hood& h) {
  bool mixed = false;
#pragma unroll
  for (int d = 0; d < 8; ++d) {
    const int32_t cx = s[0] + (d & 1), cy = s[1] + ((d >> 1) & 1), cz = s[2] + (d >> 2);
    h.inside[d] = !(cx < 0 || cy < 0 || cz < 0 || cx >= (int32_t)w || cy >= (int32_t)w || cz >= (int32_t)w);
    h.c[d] = SmoothCell{};
    if (h.inside[d]) h.c[d] = gload(grid + ((size_t)cz * w + cy) * w + cx);
    mixed |= (h.c[d].mixed & kSmoothMixed) != 0;
  }
  return mixed;
}
}  // namespace

__device__ __forceinline__ void smooth_apply_geometry_point(const DevFrame& f, uint32_t frame, uint32_t i, uint32_t n,
                                                            SmoothGrid sg, uint32_t w, uint32_t G, uint32_t T, bool both,
                                                            const vpcc_point3& pt) {
  const uint32_t p[3] = {pt.x, pt.y, pt.z};
  int32_t s[3];
  int64_t wt[3][2];
#pragma unroll
  for (int a = 0; a < 3; ++a) axis_setup(p[a], G, w, s[a], wt[a]);
  Hood h;
  if (!load_hood(sg.cells(frame), s, w, h)) return;
  int64_t num[3] = {0, 0, 0}, den = 0;
#pragma unroll
  for (int d = 0; d < 8; ++d) {
    const SmoothCell& c = h.c[d];
    if (!c.count) continue;
    const int64_t W = wt[0][d & 1] * wt[1][(d >> 1) & 1] * wt[2][d >> 2];
    num[0] += W * c.s[0]; num[1] += W * c.s[1]; num[2] += W * c.s[2];
    den += W * c.count;
  }
  if (den <= 0) return;
  int64_t C[3], d2 = 0;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    C[a] = (16 * num[a] + den / 2) / den;
    const int64_t d = 16 * (int64_t)p[a] - C[a];
    d2 += d * d;
  }
  if (d2 > 256 * (int64_t)T * T) {
    vpcc_point3 o;
    const int64_t x = (C[0] + 8) >> 4, y = (C[1] + 8) >> 4, z = (C[2] + 8) >> 4;
    o.x = (uint16_t)(x > 65535 ? 65535 : x); o.y = (uint16_t)(y > 65535 ? 65535 : y); o.z = (uint16_t)(z > 65535 ? 65535 : z);
    gstore(f.out_xyz + i, o);
    // The colour filter's cells are those of the smoothed positions: which points moved is noted, one bit per point
    // (a word per wave of 64 points, all-zero before: vpcc_gof_smooth), and k_smooth_moved_* take a moved point's sums
    // to its new cell if it has one.  One point in a few thousand moves: a non-returning atomic OR each (a list fed by
    // returning atomics on one counter per frame cost this kernel 0.15 ms; the wave's ballot stored by its first lane
    // 0.06-0.09, a reconvergence point in a kernel of early exits).
    if (both) atomicOr(reinterpret_cast<uint32_t*>(sg.moved(frame) + (i >> 6)) + ((i >> 5) & 1u), 1u << (i & 31u));
  }
}

// Both filters in one pass, after the geometry filter — a thread per wave of 64 points, which walks the wave's moved
// bits (one point in a few thousand moves): (1) a moved point that is in another cell now takes its count, colour and
// patch sums from the cell it left to the cell it is in (coordinate sums are not needed any more) and has both cells
// noted (key array: the new one; old-key array: the one it left); (2) once all have moved, both cells of every such
// point are re-examined: the mixed bit is set or cleared, and flags are painted around a cell that has become mixed
// (flags around a cell that no longer is stay — they only make a point look at its neighbourhood in vain — and are
// un-painted with the rest by k_smooth_clear, which the painted bit tells).
__global__ __launch_bounds__(256) void k_smooth_moved_sums(const DevFrame* __restrict__ frames, uint32_t first, SmoothGrid sg,
                                                           uint32_t w, uint32_t G) {
  const DevFrame& f = frames[first + blockIdx.y];
  const uint32_t n = min(*gl(f.n_points), f.capacity);
  const uint32_t wave = blockIdx.x * 256u + threadIdx.x;
  if (wave * 64u >= n) return;
  for (uint64_t mask = sg.moved(blockIdx.y)[wave]; mask; mask &= mask - 1ull) {
    const uint32_t i = wave * 64u + (uint32_t)__builtin_ctzll(mask);
    const vpcc_point3 pt = gload(f.out_xyz + i);
    const uint32_t from = sg.keys(blockIdx.y)[i];
    const uint32_t to = (cell_coord(pt.z, G, w) * w + cell_coord(pt.y, G, w)) * w + cell_coord(pt.x, G, w);
    sg.old_keys(blockIdx.y)[i] = from;
    if (to == from) continue;                                 // moved inside its cell
    sg.keys(blockIdx.y)[i] = to;                              // the point's cell from here on
    const uint32_t patch = gl(f.out_patch)[i];
    const vpcc_color3 col = gload(f.out_rgb + i);
    const uint32_t rgb[3] = {col.r, col.g, col.b};
    SmoothCell* a = sg.cells(blockIdx.y) + from;
    SmoothCell* b = sg.cells(blockIdx.y) + to;
    SmoothColorCell* ca = sg.color_cells(blockIdx.y) + from;
    SmoothColorCell* cb = sg.color_cells(blockIdx.y) + to;
    atomicSub(&a->count, 1u); atomicAdd(&b->count, 1u);
    atomicSub(&ca->count, 1u); atomicAdd(&cb->count, 1u);       // (the mixed bit above the count is not touched: count >= 1)
    atomicSub(&a->sp, patch); atomicAdd(&b->sp, patch);
    const unsigned long long q = (unsigned long long)patch * patch;
    atomicAdd(reinterpret_cast<unsigned long long*>(&a->sp2), 0ull - q);
    atomicAdd(reinterpret_cast<unsigned long long*>(&b->sp2), q);
    for (int c = 0; c < 3; ++c) {
      atomicSub(&ca->s[c], rgb[c]);
      atomicAdd(&cb->s[c], rgb[c]);
    }
  }
}
__global__ __launch_bounds__(256) void k_smooth_moved_mark(const DevFrame* __restrict__ frames, uint32_t first, SmoothGrid sg,
                                                           uint32_t w) {
  const DevFrame& f = frames[first + blockIdx.y];
  const uint32_t n = min(*gl(f.n_points), f.capacity);
  const uint32_t wave = blockIdx.x * 256u + threadIdx.x;
  if (wave * 64u >= n) return;
  for (uint64_t mask = sg.moved(blockIdx.y)[wave]; mask; mask &= mask - 1ull) {
    const uint32_t i = wave * 64u + (uint32_t)__builtin_ctzll(mask);
    const uint32_t now = sg.keys(blockIdx.y)[i], was = sg.old_keys(blockIdx.y)[i];
    if (now == was) continue;
    for (int side = 0; side < 2; ++side) {
      const uint32_t key = side ? now : was;
      SmoothCell* cell = sg.cells(blockIdx.y) + key;
      const SmoothCell c = gload(cell);
      const bool mixed = cell_mixed(c);
      // several moved points may share a cell: all compute the same bits from the same (final) sums
      if (mixed && !(c.mixed & kSmoothPainted)) paint_flags(sg.flags(blockIdx.y), key, w, 1);
      const uint32_t bits = (mixed ? kSmoothMixed : 0u) | ((mixed || (c.mixed & kSmoothPainted)) ? kSmoothPainted : 0u);
      if (bits != c.mixed) cell->mixed = bits;
      sg.color_cells(blockIdx.y)[key].count = c.count | (mixed ? kColorCellMixed : 0u);
    }
  }
}

__global__ __launch_bounds__(256) void k_smooth_apply_geometry(const DevFrame* __restrict__ frames, uint32_t first,
                                                               SmoothGrid sg, uint32_t w, uint32_t G, uint32_t T, bool both) {
  const DevFrame& f = frames[first + blockIdx.y];
  const uint32_t n = min(*gl(f.n_points), f.capacity);
  // ONE flag load (through the point's cell index) decides whether a point needs its 2x2x2 cells at all (k_smooth_mark);
  // the indices and flags of the thread's points are fetched together: two round trips instead of two per point
  uint32_t key[kApplyChunks];
  bool flagged[kApplyChunks];
#pragma unroll
  for (uint32_t c = 0; c < kApplyChunks; ++c) {
    const uint32_t i = (blockIdx.x * kApplyChunks + c) * 256u + threadIdx.x;
    key[c] = i < n ? sg.keys(blockIdx.y)[i] : 0xFFFFFFFFu;
  }
#pragma unroll
  for (uint32_t c = 0; c < kApplyChunks; ++c) flagged[c] = key[c] != 0xFFFFFFFFu && sg.flags(blockIdx.y)[key[c]] != 0;
  vpcc_point3 pt[kApplyChunks];                             // ... and the flagged points themselves, before the first is worked on
#pragma unroll
  for (uint32_t c = 0; c < kApplyChunks; ++c)
    if (flagged[c]) pt[c] = gload(f.out_xyz + (blockIdx.x * kApplyChunks + c) * 256u + threadIdx.x);
#pragma unroll
  for (uint32_t c = 0; c < kApplyChunks; ++c)
    if (flagged[c])
      smooth_apply_geometry_point(f, blockIdx.y, (blockIdx.x * kApplyChunks + c) * 256u + threadIdx.x, n, sg, w, G, T, both, pt[c]);
}

__device__ __forceinline__ void smooth_apply_color_point(const DevFrame& f, uint32_t frame, uint32_t i, uint32_t n,
                                                         SmoothGrid sg, uint32_t w, uint32_t G, uint32_t Ts, uint32_t Td, bool both,
                                                         const vpcc_point3& pt) {
  const uint32_t p[3] = {pt.x, pt.y, pt.z};
  int32_t s[3];
  int64_t wt[3][2];
#pragma unroll
  for (int a = 0; a < 3; ++a) axis_setup(p[a], G, w, s[a], wt[a]);
  Hood h;
  if (both) {                                               // both filters in one pass: the 16-byte colour cells say it all
    const SmoothColorCell* cg = sg.color_cells(frame);
    bool any = false;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      const int32_t cx = s[0] + (d & 1), cy = s[1] + ((d >> 1) & 1), cz = s[2] + (d >> 2);
      h.inside[d] = !(cx < 0 || cy < 0 || cz < 0 || cx >= (int32_t)w || cy >= (int32_t)w || cz >= (int32_t)w);
      h.c[d] = SmoothCell{};
      if (h.inside[d]) {
        const SmoothColorCell cc = gload(cg + ((size_t)cz * w + cy) * w + cx);
        h.c[d].count = cc.count & ~kColorCellMixed;
        h.c[d].s[0] = cc.s[0]; h.c[d].s[1] = cc.s[1]; h.c[d].s[2] = cc.s[2];
        h.c[d].mixed = (cc.count & kColorCellMixed) ? kSmoothMixed : 0u;
      }
      any |= h.c[d].mixed != 0;
    }
    if (!any) return;
  } else if (!load_hood(sg.cells(frame), s, w, h)) {
    return;
  }
  const vpcc_color3 col = gload(f.out_rgb + i);
  const int64_t cl[3] = {col.r, col.g, col.b};
  // the point's own cell is one of the eight: index of (q - s) per axis
  const int32_t qx = (int32_t)cell_coord(p[0], G, w), qy = (int32_t)cell_coord(p[1], G, w), qz = (int32_t)cell_coord(p[2], G, w);
  const int own = (qx - s[0]) | ((qy - s[1]) << 1) | ((qz - s[2]) << 2);
  SmoothCell cc = h.c[0];
#pragma unroll
  for (int d = 1; d < 8; ++d) if (d == own) cc = h.c[d];
  const int64_t mc[3] = {cc.s[0] / cc.count, cc.s[1] / cc.count, cc.s[2] / cc.count};
  int64_t num[3] = {0, 0, 0}, den = 0;
  bool mixed = false;
#pragma unroll
  for (int d = 0; d < 8; ++d) {
    const SmoothCell& c = h.c[d];
    if (!c.count) continue;
    if (d != own) {
      int64_t diff = 0;
#pragma unroll
      for (int a = 0; a < 3; ++a) { const int64_t m = (int64_t)(c.s[a] / c.count) - mc[a]; diff += m < 0 ? -m : m; }
      if (diff > (int64_t)Td) continue;
    }
    const int64_t W = wt[0][d & 1] * wt[1][(d >> 1) & 1] * wt[2][d >> 2];
    num[0] += W * c.s[0]; num[1] += W * c.s[1]; num[2] += W * c.s[2];
    den += W * c.count;
    mixed |= (c.mixed & kSmoothMixed) != 0;
  }
  if (!mixed || den <= 0) return;
  int64_t m[3], dist = 0;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    m[a] = (2 * num[a] + den) / (2 * den);
    const int64_t d = cl[a] - m[a];
    dist += d < 0 ? -d : d;
  }
  if (dist >= (int64_t)Ts) {
    vpcc_color3 o;
    o.r = (uint8_t)m[0]; o.g = (uint8_t)m[1]; o.b = (uint8_t)m[2];
    gstore(f.out_rgb + i, o);
  }
}

__global__ __launch_bounds__(256) void k_smooth_apply_color(const DevFrame* __restrict__ frames, uint32_t first,
                                                            SmoothGrid sg, uint32_t w, uint32_t G, uint32_t Ts, uint32_t Td, bool both) {
  const DevFrame& f = frames[first + blockIdx.y];
  const uint32_t n = min(*gl(f.n_points), f.capacity);
  uint32_t key[kApplyChunks];                              // (as in k_smooth_apply_geometry)
  bool flagged[kApplyChunks];
#pragma unroll
  for (uint32_t c = 0; c < kApplyChunks; ++c) {
    const uint32_t i = (blockIdx.x * kApplyChunks + c) * 256u + threadIdx.x;
    key[c] = i < n ? sg.keys(blockIdx.y)[i] : 0xFFFFFFFFu;
  }
#pragma unroll
  for (uint32_t c = 0; c < kApplyChunks; ++c) flagged[c] = key[c] != 0xFFFFFFFFu && sg.flags(blockIdx.y)[key[c]] != 0;
#pragma unroll 1                                            // (unrolled, with the points fetched ahead as in the geometry kernel: 86 VGPRs, 0.495 vs 0.458 ms)
  for (uint32_t c = 0; c < kApplyChunks; ++c)
    if (flagged[c]) {
      const uint32_t i = (blockIdx.x * kApplyChunks + c) * 256u + threadIdx.x;
      smooth_apply_color_point(f, blockIdx.y, i, n, sg, w, G, Ts, Td, both, gload(f.out_xyz + i));
    }
}


void launch_smooth_stats(const DevFrame* d_frames, uint32_t first, uint32_t count, uint32_t max_points, SmoothGrid sg,
                         uint32_t w, uint32_t G, uint32_t mode, void* stream) {
  if (!count || !max_points) return;
  hipLaunchKernelGGL(k_smooth_stats, dim3((max_points + 256 * kSmoothChunks - 1) / (256 * kSmoothChunks), count), dim3(256), 0,
                     (hipStream_t)stream, d_frames, first, sg, w, G, mode);
}
void launch_smooth_apply_geometry(const DevFrame* d_frames, uint32_t first, uint32_t count, uint32_t max_points,
                                  SmoothGrid sg, uint32_t w, uint32_t G, uint32_t T, bool both, void* stream) {
  if (!count || !max_points) return;
  hipLaunchKernelGGL(k_smooth_apply_geometry, dim3((max_points + 256 * kApplyChunks - 1) / (256 * kApplyChunks), count), dim3(256),
                     0, (hipStream_t)stream, d_frames, first, sg, w, G, T, both);
}
void launch_smooth_apply_color(const DevFrame* d_frames, uint32_t first, uint32_t count, uint32_t max_points,
                               SmoothGrid sg, uint32_t w, uint32_t G, uint32_t Ts, uint32_t Td, bool both, void* stream) {
  if (!count || !max_points) return;
  hipLaunchKernelGGL(k_smooth_apply_color, dim3((max_points + 256 * kApplyChunks - 1) / (256 * kApplyChunks), count), dim3(256), 0,
                     (hipStream_t)stream, d_frames, first, sg, w, G, Ts, Td, both);
}
void launch_smooth_moved(const DevFrame* d_frames, uint32_t first, uint32_t count, uint32_t max_points, SmoothGrid sg,
                         uint32_t w, uint32_t G, void* stream) {
  if (!count || !max_points) return;
  const dim3 grid(((max_points + 63) / 64 + 255) / 256, count);
  hipLaunchKernelGGL(k_smooth_moved_sums, grid, dim3(256), 0, (hipStream_t)stream, d_frames, first, sg, w, G);
  hipLaunchKernelGGL(k_smooth_moved_mark, grid, dim3(256), 0, (hipStream_t)stream, d_frames, first, sg, w);
}
void launch_smooth_mark(const DevFrame* d_frames, uint32_t first, uint32_t count, uint32_t max_points, SmoothGrid sg,
                        uint32_t w, void* stream) {
  if (!count || !max_points) return;
  hipLaunchKernelGGL(k_smooth_mark, dim3((max_points + 1023) / 1024, count), dim3(256), 0, (hipStream_t)stream, d_frames,
                     first, sg, w);
}
void launch_smooth_clear(const DevFrame* d_frames, uint32_t first, uint32_t count, uint32_t max_points, SmoothGrid sg,
                         uint32_t w, bool both, void* stream) {
  if (!count || !max_points) return;
  hipLaunchKernelGGL(k_smooth_clear, dim3((max_points + 1023) / 1024, count), dim3(256), 0, (hipStream_t)stream, d_frames,
                     first, sg, w, both);
}

}  // namespace vpcc
