// vpcc_kernels.hip — gfx950 (CDNA4, wave64) kernels of the V-PCC reconstruction path.
//
// GENERAL kernel sequence (any orientation, any block size):
//   k_block_owner  : block -> patch index      (reference src/codec.rs:205-250)
//   k_general      : ONE pass over the virtual blocks in emission order (src/codec.rs:352-480): every pixel is evaluated once
//                    (src/codec.rs:517-565, src/decoder.rs:871-888), counted, ranked by a decoupled look-back over units of up to
//                    256 pixels, and emitted — fused with the attribute gather (src/codec.rs:569-658) and YUV->RGB
//                    (src/codec.rs:661-687) — through LDS as whole 16-byte pieces of the output arrays
//   k_general_blocks : the same pass for frames whose units are chunks of ONE virtual block (block side 16 ... 256): what a block
//                    decides is scalar work, and a block that belongs to another patch costs nothing
// The W x H occupancy map of src/codec.rs:288-301, point_to_pixel and colors16bit are never
// materialised: occupancy is read through the low-resolution plane, and colour is fetched by the
// thread that emits the point.
//
// Everything is integer/byte work bound by HBM; there is no contraction, hence no MFMA.
// The only floating point is the reference's f64 colour matrix, compiled without contraction
// (-ffp-contract=off) so that it is bit-identical to the Rust code.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "vpcc_device.hpp"
#include "vpcc_devfn.hpp"

namespace vpcc {

// ------------------------------------------------------------ k_block_owner
// non_zero_pixel > 0  <=>  any occupancy sample under the block's R*R mapped pixels is non-zero; ascending-patch overwrite ==
// max over the patches that write.  patch_to_canvas (src/decoder.rs:841-867) maps the block's R x R pixels onto an axis-aligned
// R x R square of the canvas (the coefficients are a signed permutation), so the samples under them are a RECTANGLE of the
// occupancy plane between the images of two opposite corners: (R / precision + 1)^2 samples at most — 16-25 for 16 x 16
// blocks at precision 4, where walking the 256 pixels read every sample sixteen times (0.28 ms per 128 S-longdress frames;
// now 0.02).  kLanes = 1: a thread per virtual block (small rectangles); 64: a wave per virtual block.
template <uint32_t kLanes>
__global__ __launch_bounds__(256) void k_block_owner(const DevFrame* __restrict__ frames, uint32_t first) {
  const DevFrame& f = frames[first + blockIdx.y];
  const uint32_t vb = kLanes == 1u ? blockIdx.x * 256u + threadIdx.x : blockIdx.x * 4u + (threadIdx.x >> 6);
  if (vb >= f.n_vblocks) return;
  const VBlock b = gload(f.vblocks + vb);
  const int32_t e = (int32_t)f.R - 1;                                       // the block's pixels (0, 0) and (R - 1, R - 1)
  const int32_t cux = (int32_t)(b.coef & 3u) - 1, cvx = (int32_t)((b.coef >> 2) & 3u) - 1;
  const int32_t cuy = (int32_t)((b.coef >> 4) & 3u) - 1, cvy = (int32_t)(b.coef >> 6) - 1;
  const int32_t xa = (int32_t)b.x0, xb = xa + (cux + cvx) * e, ya = (int32_t)b.y0, yb = ya + (cuy + cvy) * e;
  const uint32_t sx0 = (uint32_t)min(xa, xb) / f.prec, sx1 = (uint32_t)max(xa, xb) / f.prec;     // (inside the canvas: validate_frame)
  const uint32_t sy0 = (uint32_t)min(ya, yb) / f.prec, sy1 = (uint32_t)max(ya, yb) / f.prec;
  const uint32_t nx = sx1 - sx0 + 1u, total = nx * (sy1 - sy0 + 1u);
  uint32_t any = 0;
  for (uint32_t i = kLanes == 1u ? 0u : lane_id(); i < total; i += kLanes) {
    const uint32_t sy = sy0 + i / nx, sx = sx0 + i % nx;
    any |= gl(f.occ)[(size_t)sy * f.occ_stride + sx];
  }
  const bool hit = kLanes == 1u ? any != 0u : (__ballot(any != 0u) != 0ull && lane_id() == 0);
  if (hit) atomicMax(f.block_to_patch + b.canvas_block, (uint32_t)b.patch + 1u);
}

// ------------------------------------------------------------------ planning
// generate_block_to_patch_from_occupancy_map_video (src/codec.rs:205-250) and the tile kernel's work list, built where the
// occupancy plane lies, by every launch (the planes of a gof that borrows them may have changed since the last one).  For
// Default/Swap patches and 16x16 blocks the pixels of a virtual block are the pixels of its canvas block, so the reference's
// block_to_patch — ascending patches, each overwriting the blocks under its occupied virtual blocks — is "the highest patch
// that covers the block, if the block holds any occupancy".
//
// The host writes O(patches) per frame (vb_base, one item template per patch); the virtual blocks are derived here.
__device__ __forceinline__ bool block_occupied(const DevFrame& f, uint32_t cb) {
  const uint32_t spb = f.prec >= 16u ? 1u : 16u / f.prec;                // occupancy samples per block side
  const uint32_t sx = (cb % f.bw) * 16u / f.prec, sy = (cb / f.bw) * 16u / f.prec;
  // A block's samples lie inside the plane (validate_frame: the plane covers the canvas, the blocks lie inside the canvas).
  // Rows of 4, 8 or 16 samples whose addresses are multiples of their length come as whole words.
  const bool words = spb >= 4u && (((uint32_t)(uintptr_t)f.occ | f.occ_stride) & (spb - 1u)) == 0u;
  uint32_t any = 0;
  if (words) {
    const VPCC_GLOBAL uint8_t* row = gl(f.occ) + (size_t)sy * f.occ_stride + sx;
    for (uint32_t y = 0; y < spb; ++y, row += f.occ_stride)
      for (uint32_t x = 0; x < spb; x += 4u) any |= *(const VPCC_GLOBAL uint32_t*)(row + x);
    return any != 0;
  }
  for (uint32_t y = sy; y < sy + spb && y < f.occ_h; ++y)
    for (uint32_t x = sx; x < sx + spb && x < f.occ_w; ++x) any |= gl(f.occ)[y * f.occ_stride + x];
  return any != 0;
}

// k_plan_tiles: ONE kernel, a workgroup per frame, the frame's block_to_patch and patch table in LDS.
//   1. a thread per canvas block: does the block hold any occupancy?  LDS word = 0 (yes) or kPlanEmpty (no) — eight blocks
//      per thread with all their loads issued together (the occupancy plane comes from HBM: one round trip for a frame of up
//      to 8 192 blocks, not one per block); a thread per patch: {vb_base, origin, size_u0 | swap} into LDS;
//   2. a thread per virtual block (binary search of vb_base in LDS): LDS max(word, patch + 1) where the block is occupied;
//   3. ordered compaction of the virtual blocks that own their canvas block: every thread takes a CONTIGUOUS run of them
//      (emission order = thread order), counts its owners, ONE workgroup-wide exclusive scan of the counts, then every
//      thread completes its owners' items from their patches' templates — no barrier per step of the walk;
//      block_to_patch goes to global memory for whoever asks (vpcc_gof_block_to_patch).
constexpr uint32_t kPlanEmpty = 0x80000000u;
constexpr uint32_t kPlanThreads = 1024;
#ifdef VPCC_PLAN_STAMPS                                   // tools/exp_plan_stamps.py: where k_plan_tiles' time goes (never in the product)
__device__ unsigned long long g_plan_stamps[32];       // [0, 16): workgroup 0, [16, 32): the launch's last workgroup; thread 0 of each
#define VPCC_STAMP(k) do { if ((blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) && threadIdx.x == 0) g_plan_stamps[(blockIdx.x ? 16 : 0) + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define VPCC_STAMP(k) do { } while (0)
#endif
// floor(a / d) for a * d < 2^32 by one multiply-high with m = ceil(2^32 / d) (d >= 2; d == 1: the caller's business): the
// kernel is ONE workgroup per frame, and the ~40 instructions of a 32-bit division per block and per virtual block were most
// of its time.  Here a < 32768 (canvas blocks of a frame that fits the LDS) and d <= 2048.
__device__ __forceinline__ uint32_t magic_of(uint32_t d) { return d > 1u ? 0xFFFFFFFFu / d + 1u : 0u; }      // ceil(2^32 / d), in 32 bits
__device__ __forceinline__ uint32_t div_by(uint32_t a, uint32_t d, uint32_t magic) { return d > 1u ? __umulhi(a, magic) : a; }

__global__ __launch_bounds__(kPlanThreads) void k_plan_tiles(DevFrame* frames_rw, uint32_t first, uint32_t lds_bytes, uint32_t write_b2p) {
  // The descriptor is read through the constant address space, once: through a plain reference every field is a generic
  // load that may alias the LDS and the items the kernel stores — reloaded (a round trip to the L2 each) after every store.
  VPCC_STAMP(0);
  DevFrame f;
  __builtin_memcpy(&f, (const __attribute__((address_space(4))) void*)(frames_rw + first + blockIdx.x), sizeof f);
  if (!f.patch_items) return;                                              // (a frame of the general sequence)
  extern __shared__ __attribute__((aligned(16))) uint32_t plan_lds[];
  __shared__ uint32_t wave_total[kPlanThreads / 64];
  const uint32_t bw = f.bw, nb = bw * f.bh, P = f.n_patches, n = f.n_vblocks;
  VPCC_STAMP(9);
  const uint32_t bw_magic = magic_of(bw);
  uint32_t* const b2p = plan_lds;                                          // [nb]
  uint32_t* const pbase = plan_lds + nb;                                   // [P + 1]
  uint32_t* const porg = pbase + P + 1;                                    // [P]: u0 | v0 << 16 (blocks)
  uint32_t* const pdim = porg + P;                                         // [P]: size_u0 | swap << 16
  // the item templates too, where the launch's LDS has room for them (32 B per patch): the compaction then completes its
  // items without a round trip to memory per owner
  TileItem* const ptmpl = (TileItem*)(plan_lds + ((nb + 3u * P + 1u + 3u) & ~3u));
  const bool tmpl_in_lds = plan_tiles_lds_bytes(nb, P, true) <= lds_bytes;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  // ONE round trip to memory for everything the frame's planning reads: the occupancy of the canvas blocks (rows of 4 samples
  // — precision 4 — whose addresses are multiples of four come as whole words, eight blocks per thread in flight; anything
  // else block by block, below) and the patch table, all loads issued before the first of them is waited for.
  const uint32_t spb = f.prec >= 16u ? 1u : 16u / f.prec;                  // occupancy samples per block side
  const bool words = spb == 4u && (((uint32_t)(uintptr_t)f.occ | f.occ_stride) & 3u) == 0u;
  const uint32_t stride = f.occ_stride;
  // ... four blocks of a block row per thread where rows are multiples of 16 bytes: four 16-byte loads for four blocks.
  // (The four-byte loads are 448 wave-instructions through ONE CU's address unit per frame: 4.4 of the kernel's 18 us went
  // into ISSUING them, profiles/r05/plan_stamps.txt.)
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const bool quads = words && (bw & 3u) == 0u && (((uint32_t)(uintptr_t)f.occ | stride) & 15u) == 0u;
  const uint32_t bw4 = bw >> 2, nq = nb >> 2, bw4_magic = magic_of(bw4);
  u32x4 v4[2] = {};
  if (quads) {
#pragma unroll
    for (uint32_t k = 0; k < 2u; ++k) {
      const uint32_t qd = k * kPlanThreads + tid;
      if (qd < nq) {
        const uint32_t by = div_by(qd, bw4, bw4_magic), qx = qd - by * bw4;
        const VPCC_GLOBAL uint8_t* row = gl(f.occ) + (by * 4u) * stride + qx * 16u;
        v4[k] = *(const VPCC_GLOBAL u32x4*)row | *(const VPCC_GLOBAL u32x4*)(row + stride) |
                *(const VPCC_GLOBAL u32x4*)(row + 2u * stride) | *(const VPCC_GLOBAL u32x4*)(row + 3u * stride);
      }
    }
  }
  uint32_t v[8];
  if (words && !quads) {
#pragma unroll
    for (uint32_t k = 0; k < 8u; ++k) {
      const uint32_t cb = k * kPlanThreads + tid;
      v[k] = 0;
      if (cb < nb) {
        const uint32_t by = div_by(cb, bw, bw_magic), bx = cb - by * bw;
        const VPCC_GLOBAL uint8_t* row = gl(f.occ) + (by * 4u) * stride + bx * 4u;          // (the tile path's planes: 32-bit offsets)
        v[k] = *(const VPCC_GLOBAL uint32_t*)row | *(const VPCC_GLOBAL uint32_t*)(row + stride) |
               *(const VPCC_GLOBAL uint32_t*)(row + 2u * stride) | *(const VPCC_GLOBAL uint32_t*)(row + 3u * stride);
      }
    }
  }
  VPCC_STAMP(10);
  for (uint32_t p = tid; p <= P; p += kPlanThreads) {
    pbase[p] = gl(f.vb_base)[p];
    if (p < P) {
      const TileItem t = gload(f.patch_items + p);
      porg[p] = (uint32_t)t.x0 | ((uint32_t)t.y0 << 16);
      pdim[p] = (uint32_t)t.patch | ((uint32_t)(t.flags & kTileSwap) << 16);
      if (tmpl_in_lds) ptmpl[p] = t;
    }
  }
  VPCC_STAMP(11);
  if (quads) {
    auto flags_of = [](u32x4 w) { return u32x4{w.x ? 0u : kPlanEmpty, w.y ? 0u : kPlanEmpty, w.z ? 0u : kPlanEmpty, w.w ? 0u : kPlanEmpty}; };
#pragma unroll
    for (uint32_t k = 0; k < 2u; ++k) {
      const uint32_t qd = k * kPlanThreads + tid;
      if (qd < nq) *(u32x4*)(b2p + 4u * qd) = flags_of(v4[k]);               // (block row by, blocks 4 qx .. + 3: by * bw + 4 qx = 4 qd)
    }
    for (uint32_t q0 = 2u * kPlanThreads; q0 < nq; q0 += 2u * kPlanThreads) {   // (frames beyond 8 192 blocks)
#pragma unroll
      for (uint32_t k = 0; k < 2u; ++k) {
        const uint32_t qd = q0 + k * kPlanThreads + tid;
        v4[k] = u32x4{0u, 0u, 0u, 0u};
        if (qd < nq) {
          const uint32_t by = div_by(qd, bw4, bw4_magic), qx = qd - by * bw4;
          const VPCC_GLOBAL uint8_t* row = gl(f.occ) + (by * 4u) * stride + qx * 16u;
          v4[k] = *(const VPCC_GLOBAL u32x4*)row | *(const VPCC_GLOBAL u32x4*)(row + stride) |
                  *(const VPCC_GLOBAL u32x4*)(row + 2u * stride) | *(const VPCC_GLOBAL u32x4*)(row + 3u * stride);
        }
      }
#pragma unroll
      for (uint32_t k = 0; k < 2u; ++k) {
        const uint32_t qd = q0 + k * kPlanThreads + tid;
        if (qd < nq) *(u32x4*)(b2p + 4u * qd) = flags_of(v4[k]);
      }
    }
  } else if (words) {
#pragma unroll
    for (uint32_t k = 0; k < 8u; ++k) {
      const uint32_t cb = k * kPlanThreads + tid;
      if (cb < nb) b2p[cb] = v[k] ? 0u : kPlanEmpty;
    }
    for (uint32_t c0 = 8u * kPlanThreads; c0 < nb; c0 += 8u * kPlanThreads) {          // (frames beyond 8 192 blocks)
#pragma unroll
      for (uint32_t k = 0; k < 8u; ++k) {
        const uint32_t cb = c0 + k * kPlanThreads + tid;
        v[k] = 0;
        if (cb < nb) {
          const uint32_t by = div_by(cb, bw, bw_magic), bx = cb - by * bw;
          const VPCC_GLOBAL uint8_t* row = gl(f.occ) + (by * 4u) * stride + bx * 4u;
          v[k] = *(const VPCC_GLOBAL uint32_t*)row | *(const VPCC_GLOBAL uint32_t*)(row + stride) |
                 *(const VPCC_GLOBAL uint32_t*)(row + 2u * stride) | *(const VPCC_GLOBAL uint32_t*)(row + 3u * stride);
        }
      }
#pragma unroll
      for (uint32_t k = 0; k < 8u; ++k) {
        const uint32_t cb = c0 + k * kPlanThreads + tid;
        if (cb < nb) b2p[cb] = v[k] ? 0u : kPlanEmpty;
      }
    }
  } else {
    for (uint32_t cb = tid; cb < nb; cb += kPlanThreads) b2p[cb] = block_occupied(f, cb) ? 0u : kPlanEmpty;
  }
  VPCC_STAMP(1);
  __syncthreads();
  VPCC_STAMP(2);
  // A thread walks a CONTIGUOUS run of the emission order: located once (binary search of vb_base, one division), then
  // stepped: u0, v0, the patch.
  struct Walk {
    uint32_t p, u0, v0, su, org, next;       // patch, block of it, its size_u0, origin u0 | v0 << 16, first virtual block of the patch after it
    bool swap;
  };
  auto enter = [&](Walk& w, uint32_t p) {      // patch p (with blocks), at its first block
    w.p = p; w.u0 = 0; w.v0 = 0;
    w.su = pdim[p] & 0xFFFFu; w.swap = (pdim[p] >> 16) != 0; w.org = porg[p]; w.next = pbase[p + 1];
  };
  auto locate = [&](Walk& w, uint32_t vb) {
    enter(w, patch_of_vblock(pbase, P, vb));
    const uint32_t r = vb - pbase[w.p];
    w.v0 = w.su > 1u ? r / w.su : r;           // (once per thread and round)
    w.u0 = r - w.v0 * w.su;
  };
  auto step = [&](Walk& w, uint32_t vb_next) { // to virtual block vb_next = the current one + 1 (< n)
    if (vb_next == w.next) {
      uint32_t p = w.p + 1u;
      while (pbase[p + 1] == vb_next) ++p;     // patches without blocks
      enter(w, p);
    } else if (++w.u0 == w.su) { w.u0 = 0; ++w.v0; }
  };
  auto canvas_block = [&](const Walk& w) {     // src/decoder.rs:853-867: Default (u0 + u, v0 + v), Swap (u0 + v, v0 + u)
    const uint32_t bx = (w.org & 0xFFFFu) + (w.swap ? w.v0 : w.u0), by = (w.org >> 16) + (w.swap ? w.u0 : w.v0);
    return by * bw + bx;
  };
  // The virtual blocks in ROUNDS of up to 8 192: a thread's run in a round is at most eight blocks long, so that what the
  // compaction learns of them fits its registers.
  constexpr uint32_t kRun = 8u, kRound = kRun * kPlanThreads;
  // 2. cover: LDS max(word, patch + 1) where the block is occupied
  for (uint32_t r0 = 0; r0 < n; r0 += kRound) {
    const uint32_t in_round = min(n - r0, kRound), per = (in_round + kPlanThreads - 1u) / kPlanThreads;
    const uint32_t v_lo = r0 + min(tid * per, in_round), v_hi = r0 + min(tid * per + per, in_round);
    if (v_lo >= v_hi) continue;
    Walk w;
    locate(w, v_lo);
    for (uint32_t vb = v_lo; vb < v_hi; ++vb) {
      const uint32_t cb = canvas_block(w);
      if (b2p[cb] != kPlanEmpty) atomicMax(&b2p[cb], w.p + 1u);           // (an empty block's word never changes)
      if (vb + 1u < v_hi) step(w, vb + 1u);
    }
  }
  VPCC_STAMP(3);
  __syncthreads();
  VPCC_STAMP(4);
  // 3. ordered compaction, round by round: the owners of a thread's run follow those of all threads before it
  uint32_t base_items = 0;
  for (uint32_t r0 = 0; r0 < n; r0 += kRound) {
    const uint32_t in_round = min(n - r0, kRound), per = (in_round + kPlanThreads - 1u) / kPlanThreads;
    const uint32_t v_lo = r0 + min(tid * per, in_round), v_hi = r0 + min(tid * per + per, in_round);
    uint32_t own_pu[kRun], own_cv[kRun], mine = 0;                         // the run's owners: patch | u0 << 16, canvas block | v0 << 16
    if (v_lo < v_hi) {
      Walk w;
      locate(w, v_lo);
#pragma unroll
      for (uint32_t k = 0; k < kRun; ++k) {
        const uint32_t vb = v_lo + k;
        if (vb < v_hi) {
          const uint32_t cb = canvas_block(w);
          if (b2p[cb] == w.p + 1u) {
            // (a compile-time index: the owners are kept in order by shifting the two short arrays, not by indexing them)
#pragma unroll
            for (uint32_t q = 0; q < kRun; ++q) if (q == mine) { own_pu[q] = w.p | (w.u0 << 16); own_cv[q] = cb | (w.v0 << 16); }
            ++mine;
          }
          if (vb + 1u < v_hi) step(w, vb + 1u);
        }
      }
    }
    uint32_t incl = mine;                                                  // inclusive scan inside the wave, then over the waves
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(incl, off, 64);
      if ((int)lane >= off) incl += t;
    }
    if (r0) __syncthreads();                                               // (wave_total is read by the round before)
    if (lane == 63u) wave_total[wave] = incl;
    VPCC_STAMP(5);
    __syncthreads();
    VPCC_STAMP(6);
    uint32_t at = base_items + incl - mine;
    for (uint32_t w = 0; w < kPlanThreads / 64; ++w) {
      const uint32_t t = wave_total[w];
      if (w < wave) at += t;
      base_items += t;
    }
#pragma unroll
    for (uint32_t q = 0; q < kRun; ++q) {
      if (q < mine) {
        const uint32_t pp = own_pu[q] & 0xFFFFu, u0 = own_pu[q] >> 16, cb = own_cv[q] & 0xFFFFu, v0 = own_cv[q] >> 16;
        TileItem t = tmpl_in_lds ? ptmpl[pp] : gload(f.patch_items + pp);
        const uint32_t by = div_by(cb, bw, bw_magic), bx = cb - by * bw;
        t.x0 = (uint16_t)(bx * 16u);
        t.y0 = (uint16_t)(by * 16u);
        t.patch = (uint16_t)pp;
        t.tb += u0 * 16u * t.lod_x;
        t.bb += v0 * 16u * t.lod_y;
        gstore(f.tiles + at + q, t);
      }
    }
  }
  VPCC_STAMP(7);
  if (tid == 0) *glw(&frames_rw[first + blockIdx.x].n_tiles) = base_items;
  // block_to_patch leaves the LDS for whoever asked (vpcc_gof_block_to_patch plans once more for it)
  if (write_b2p) {
    VPCC_GLOBAL uint32_t* const out_b2p = glw(f.block_to_patch);
    for (uint32_t cb = tid; cb < nb; cb += kPlanThreads) out_b2p[cb] = b2p[cb] & ~kPlanEmpty;
  }
  VPCC_STAMP(8);
}
#ifdef VPCC_PLAN_STAMPS
extern "C" int vpcc_debug_plan_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_plan_stamps), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : 1;
}
#endif

// Frames beyond k_plan_tiles' LDS, and the general sequence: the virtual blocks written out once per gof ...
__global__ __launch_bounds__(256) void k_plan_vblocks(DevFrame* __restrict__ frames, uint32_t first) {
  const DevFrame& f = frames[first + blockIdx.y];
  const uint32_t vb = blockIdx.x * 256u + threadIdx.x;
  if (!f.patches || vb >= f.n_vblocks) return;
  const uint32_t p = patch_of_vblock(f.vb_base, f.n_patches, vb);
  gstore(f.vblocks + vb, vblock_of(gload(f.patches + p), p, vb, f.bw, f.R));
}
// ... and, per launch, over block_to_patch zeroed in global memory:
//   k_plan_cover: a thread per virtual block: atomicMax(block_to_patch[canvas block], patch + 1) if the block is occupied;
//   k_plan_items: a workgroup per frame walks the virtual blocks in emission order, keeps those that own their canvas
//                 block (ordered compaction: ballot + wave scan + a running base), and completes each item from its
//                 patch's template.
__global__ __launch_bounds__(256) void k_plan_cover(const DevFrame* __restrict__ frames, uint32_t first) {
  const DevFrame& f = frames[first + blockIdx.y];
  const uint32_t vb = blockIdx.x * 256u + threadIdx.x;
  if (!f.patch_items || vb >= f.n_vblocks) return;
  const VBlock b = gload(f.vblocks + vb);
  if (block_occupied(f, b.canvas_block)) atomicMax(f.block_to_patch + b.canvas_block, (uint32_t)b.patch + 1u);
}
__global__ __launch_bounds__(1024) void k_plan_items(DevFrame* __restrict__ frames, uint32_t first) {
  DevFrame& f = frames[first + blockIdx.x];
  if (!f.patch_items) return;                                              // (a frame of the general sequence)
  __shared__ uint32_t wave_total[16];
  __shared__ uint32_t base_s;
  if (threadIdx.x == 0) base_s = 0;
  __syncthreads();
  const uint32_t n = f.n_vblocks, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  for (uint32_t v0 = 0; v0 < n; v0 += 1024u) {
    const uint32_t vb = v0 + threadIdx.x;
    VBlock b{};
    bool own = false;
    if (vb < n) {
      b = gload(f.vblocks + vb);
      own = gl(f.block_to_patch)[b.canvas_block] == (uint32_t)b.patch + 1u;
    }
    const uint64_t m = __ballot(own);
    if (lane == 0) wave_total[wave] = (uint32_t)__builtin_popcountll(m);
    __syncthreads();
    uint32_t before = base_s;
    for (uint32_t w = 0; w < wave; ++w) before += wave_total[w];
    if (own) {
      TileItem t = gload(f.patch_items + b.patch);
      t.x0 = (uint16_t)((b.canvas_block % f.bw) * 16u);
      t.y0 = (uint16_t)((b.canvas_block / f.bw) * 16u);
      t.patch = b.patch;
      t.tb += (uint32_t)b.u0 * 16u * t.lod_x;
      t.bb += (uint32_t)b.v0 * 16u * t.lod_y;
      gstore(f.tiles + before + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull)), t);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t tot = 0;
      for (uint32_t w = 0; w < 16u; ++w) tot += wave_total[w];
      base_s += tot;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) f.n_tiles = base_s;
}
void launch_plan_tiles(DevFrame* d_frames, uint32_t first, uint32_t count, size_t lds_bytes, bool write_block_to_patch, void* stream) {
  if (!count) return;
  // (a frame beyond 64 KB of LDS: the function's limit on the calling thread's device is raised first — a microsecond)
  if (lds_bytes > (size_t(60) << 10))
    (void)hipFuncSetAttribute((const void*)k_plan_tiles, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)kPlanLdsMax);
  hipLaunchKernelGGL(k_plan_tiles, dim3(count), dim3(kPlanThreads), lds_bytes, (hipStream_t)stream, d_frames, first, (uint32_t)lds_bytes,
                     write_block_to_patch ? 1u : 0u);
}
void launch_plan_tiles_global(DevFrame* d_frames, uint32_t first, uint32_t count, uint32_t max_vb, uint32_t* d_b2p, size_t b2p_words,
                              void* stream) {
  if (!count) return;
  if (b2p_words) (void)hipMemsetAsync(d_b2p, 0, b2p_words * sizeof(uint32_t), (hipStream_t)stream);
  if (max_vb) hipLaunchKernelGGL(k_plan_cover, dim3((max_vb + 255u) / 256u, count), dim3(256), 0, (hipStream_t)stream, d_frames, first);
  hipLaunchKernelGGL(k_plan_items, dim3(count), dim3(1024), 0, (hipStream_t)stream, d_frames, first);
}
void launch_plan_vblocks(DevFrame* d_frames, uint32_t first, uint32_t count, uint32_t max_vb, void* stream) {
  if (!count || !max_vb) return;
  hipLaunchKernelGGL(k_plan_vblocks, dim3((max_vb + 255u) / 256u, count), dim3(256), 0, (hipStream_t)stream, d_frames, first);
}

// ---------------------------------------------------------------- k_general
// The general sequence's single pass.  A UNIT is up to 256 consecutive pixels of the emission order: a raster chunk of one
// virtual block (R*R >= 256), or floor(256 / (R*R)) whole virtual blocks; a workgroup takes a GROUP of kGenUnitsPerGroup
// consecutive units, a pixel of each per thread (with one unit per workgroup the chain of dependent round trips — virtual
// block, ownership, patch, occupancy, depths, look-back — was paid per 256 pixels: 2.5 ms per 128 S-longdress frames):
//   1. evaluate the pixel (ownership of its block, occupancy, both depths: at most two points) — ONCE: the points stay in
//      registers (rounds 1-4 counted in one kernel and evaluated everything again in another);
//   2. rank inside the unit (ballots + wave totals); the unit's total is published as {generation, AGGREGATE, total};
//   3. attribute gather + colour conversion while the predecessors finish;
//   4. decoupled look-back (wave 0, 64 predecessors per step) -> the unit's first point index; published as PREFIX;
//   5. the unit's points are laid out in LDS exactly as they will lie in memory — at the output address modulo 16 — and leave
//      as whole aligned 16-byte pieces (non-temporal), the ragged first and last piece byte by byte: ~190 + 95 store
//      instructions per unit of ~450 points where k_emit issued six 2-byte / 1-byte stores per point.
// A frame's groups are dispatched in ascending order (gen_work_of, vpcc_device.hpp), so a group only ever waits for groups that
// were dispatched before it.  Status words carry the launch generation: nothing is cleared between launches.
namespace {
constexpr uint32_t kGenThreads = 256;
constexpr uint64_t kGenStatusShift = 32, kGenGenShift = 34;
constexpr uint64_t kGenAggregate = 1ull << kGenStatusShift, kGenPrefix = 2ull << kGenStatusShift;
constexpr uint32_t kGenSpinLimit = 1u << 24;
__device__ __forceinline__ uint32_t gen_status(uint64_t s, uint32_t gen) { return (uint32_t)(s >> kGenGenShift) == gen ? (uint32_t)(s >> kGenStatusShift) & 3u : 0u; }

// Workgroup barrier that orders LDS only: the waves of a workgroup hand each other nothing through global memory here, and
// __syncthreads() — a fence over ALL address spaces — makes every wave wait for its outstanding output stores (s_waitcnt
// vmcnt(0): a round trip to HBM) at each of a group's up to nineteen barriers.
__device__ __forceinline__ void gen_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// `n` elements of `esize` bytes staged in `lds` (at byte offset shift = address of element `first` modulo 16) -> out + first * esize:
// the whole 16-byte pieces by threads in turn, the ragged piece in front and the one behind a byte per thread (threads 0-15 /
// 16-31) — no loop over bytes: with one, every wave that held an edge piece walked sixteen iterations per array and unit, and
// that was most of the kernel's vector instructions.
__device__ __forceinline__ void copy_out(const unsigned char* lds, unsigned char* out, size_t first, uint32_t n, uint32_t esize) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  if (!n) return;
  VPCC_GLOBAL unsigned char* dst = (VPCC_GLOBAL unsigned char*)out + first * esize;
  const uint32_t shift = (uint32_t)((uintptr_t)dst & 15u), end = shift + n * esize;
  VPCC_GLOBAL unsigned char* base = dst - shift;                            // 16-byte aligned
  const uint32_t full_lo = (shift + 15u) & ~15u, full_hi = end & ~15u;      // the whole pieces: [full_lo, full_hi)
  for (uint32_t c = full_lo + threadIdx.x * 16u; c < full_hi; c += kGenThreads * 16u)
    __builtin_nontemporal_store(*(const u32x4*)(lds + c), (VPCC_GLOBAL u32x4*)(base + c));
  if (threadIdx.x < 16u) {                                                  // in front: [shift, min(full_lo, end))
    const uint32_t k = full_lo - 16u + threadIdx.x;
    if ((shift & 15u) && k >= shift && k < min(full_lo, end)) base[k] = lds[k];
  } else if (threadIdx.x < 32u) {                                           // behind: [full_hi, end), unless the piece in front holds it
    const uint32_t k = full_hi + (threadIdx.x - 16u);
    if (full_hi >= full_lo && k < end) base[k] = lds[k];
  }
}
}  // namespace

// Which (frame, group) a workgroup of the general sequence's pass takes.  Workgroups go to the eight XCDs in turn (workgroup L to
// XCD L % 8), and each XCD has its own L2: with the frame = blockIdx.y the groups of one frame were spread over all eight, and a
// 128-byte line of a plane — the rows of four neighbouring blocks — was fetched from memory by up to four L2s (TCC_EA0_RDREQ:
// 2.2 GB per 128 S-longdress frames where the tile kernel, whose frames stay on one XCD, reads 1.3; now 0.94).  Here XCD x takes
// frames x, x + 8, ..., kGenInterleave of them at a time with their groups in turn, every frame's groups in ascending order: a
// group's predecessors still precede it in dispatch order.  Why several frames at a time: a group cannot finish before ALL its
// predecessors have published their totals, so one slow group holds up every later group of its frame that is resident — with one
// frame per XCD that is the whole XCD (1.30 ms per 128 S-longdress frames; two frames 1.07, four 0.84, eight 0.77, sixteen 0.78).
#ifndef VPCC_GEN_INTERLEAVE
#define VPCC_GEN_INTERLEAVE 8
#endif
constexpr uint32_t kGenInterleave = VPCC_GEN_INTERLEAVE;
__device__ __forceinline__ GenWork gen_work(uint32_t first, uint32_t count, uint32_t groups_per_frame, uint32_t interleave, uint32_t lanes) {
  GenWork w = gen_work_of(blockIdx.x, count, groups_per_frame, interleave, lanes);   // vpcc_device.hpp (tests/fuzz_plan.cpp checks it on the CPU)
  w.frame += first;
  return w;
}

#ifndef VPCC_GEN_WAVES
#define VPCC_GEN_WAVES 7
#endif
__global__ __launch_bounds__(kGenThreads) __attribute__((amdgpu_waves_per_eu(VPCC_GEN_WAVES, 8))) void k_general(const DevFrame* __restrict__ frames, uint32_t first, uint32_t count, uint32_t groups_per_frame, uint32_t interleave, uint32_t lanes, uint32_t gen) {
  const GenWork work = gen_work(first, count, groups_per_frame, interleave, lanes);
  if (!work.any) return;
  const DevFrame& f = frames[work.frame];
  const uint32_t R = f.R, RR = R * R, n_vb = f.n_vblocks;
  const uint32_t per = RR >= kGenThreads ? 0u : kGenThreads / RR;           // whole virtual blocks per unit (small blocks)
  const uint32_t chunks = per ? 1u : (RR + kGenThreads - 1u) / kGenThreads; // units per virtual block (large blocks)
  const uint32_t n_units = per ? (n_vb + per - 1u) / per : n_vb * chunks;
  const uint32_t n_groups = (n_units + kGenUnitsPerGroup - 1u) / kGenUnitsPerGroup;
  constexpr uint32_t kU = kGenUnitsPerGroup;
  __shared__ uint32_t wave_sum[kU][4];
  __shared__ uint32_t prefix_s;
  __shared__ __attribute__((aligned(16))) unsigned char stage_xyz[2 * kGenThreads * 6 + 32];
  __shared__ __attribute__((aligned(16))) unsigned char stage_rgb[2 * kGenThreads * 3 + 32];
  __shared__ __attribute__((aligned(16))) unsigned char stage_idx[2 * kGenThreads * 2 + 32];
  const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
  const uint32_t log2R = (R & (R - 1u)) == 0u ? 31u - (uint32_t)__builtin_clz(R) : 0xFFu;
  // One group per workgroup (group g waits for groups before it only: they were dispatched before it, and publish their totals
  // before they wait for anything).
  const uint32_t group = work.group;
  if (group >= n_groups) return;
  // 1. the thread's pixel of each of the group's units: all of them evaluated before anything is waited for
  // (packed: a unit costs a thread 7 registers until it is emitted — x | y << 16 of each point, z0 | z1 << 16, the canvas pixel,
  // patch | points << 16 — and two for its colours)
  uint32_t pxy[kU][2], pz[kU], cxy[kU], patch_n[kU];
#pragma unroll
  for (uint32_t j = 0; j < kU; ++j) {
    const uint32_t unit = group * kU + j;
    uint32_t vb, i;
    bool active;
    if (per) {
      const uint32_t q = tid / RR;
      vb = unit * per + q; i = tid - q * RR; active = q < per && vb < n_vb;
    } else {
      vb = chunks == 1u ? unit : unit / chunks; i = (unit - vb * chunks) * kGenThreads + tid; active = i < RR && vb < n_vb;
    }
    pxy[j][0] = pxy[j][1] = pz[j] = cxy[j] = patch_n[j] = 0;
    if (active && unit < n_units) {
      // ONE record says where the block's pixels lie and how they become points; the ownership of the block (src/codec.rs:379) and
      // the pixel's samples are then requested together — the samples of a block that turns out to be somebody else's are not used
      const VBlock b = gload(f.vblocks + vb);
      patch_n[j] = b.patch;
      const uint32_t pv = log2R != 0xFFu ? i >> log2R : i / R, pu = i - pv * R;
      const uint32_t owner = gl(f.block_to_patch)[b.canvas_block];
      const PixelOut o = eval_pixel(f, b, pu, pv);
      if (owner == (uint32_t)b.patch + 1u) {
        if (o.n) {
          pxy[j][0] = (uint32_t)o.p0.c[0] | ((uint32_t)o.p0.c[1] << 16);
          pz[j] = o.p0.c[2];
          cxy[j] = o.x | (o.y << 16);                                       // (canvas sides <= 32768)
          if (o.n > 1u) { pxy[j][1] = (uint32_t)o.p1.c[0] | ((uint32_t)o.p1.c[1] << 16); pz[j] |= (uint32_t)o.p1.c[2] << 16; }
          patch_n[j] |= o.n << 16;
        }
      }
    }
  }
  // 2. ranks inside the units
  uint32_t before[kU];
#pragma unroll
  for (uint32_t j = 0; j < kU; ++j) {
    const uint64_t m1 = __ballot((patch_n[j] >> 16) >= 1u), m2 = __ballot((patch_n[j] >> 16) == 2u);
    before[j] = mbcnt(m1) + mbcnt(m2);
    if (lane == 0) wave_sum[j][wave] = (uint32_t)__popcll(m1) + (uint32_t)__popcll(m2);
  }
  gen_sync_lds();
  uint32_t ubase[kU], utot[kU], total = 0;                                   // a unit's first rank in the group, its points
#pragma unroll
  for (uint32_t j = 0; j < kU; ++j) {
    uint32_t wb = 0, t = 0;
#pragma unroll
    for (uint32_t w = 0; w < 4; ++w) {
      const uint32_t x = wave_sum[j][w];
      if (w < wave) wb += x;
      t += x;
    }
    ubase[j] = total;
    utot[j] = t;
    before[j] += wb;
    total += t;
  }
  uint64_t* const state = reinterpret_cast<uint64_t*>(f.vb_count);          // one {generation | status | value} word per group
  if (tid == 0)
    __hip_atomic_store(glw(state) + group, ((uint64_t)gen << kGenGenShift) | (group ? kGenAggregate : kGenPrefix) | total, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  // 3. colours of the thread's points (color_point_cloud, src/codec.rs:626-644: layer l of the attribute video for point l),
  // r | g << 8 | b << 16
  // (the two layers written out, not a loop over a layer index: indexing the descriptor's arrays by a per-lane loop
  // counter made the compiler fetch their fields with scalar loads inside a waterfall loop, per unit and layer)
  uint32_t col[kU][2];
  const uint16_t* const ay0 = f.attr_y[0]; const uint16_t* const au0 = f.attr_u[0]; const uint16_t* const av0 = f.attr_v[0];
  const uint16_t* const ay1 = f.attr_y[1]; const uint16_t* const au1 = f.attr_u[1]; const uint16_t* const av1 = f.attr_v[1];
  const uint32_t as0 = f.attr_stride[0], as1 = f.attr_stride[1], ac0 = f.attr_cstride[0], ac1 = f.attr_cstride[1];
  const bool has_attr = f.has_attr != 0;
#pragma unroll
  for (uint32_t j = 0; j < kU; ++j) {
    col[j][0] = col[j][1] = 0;
    const uint32_t np = patch_n[j] >> 16, x = cxy[j] & 0xFFFFu, y = cxy[j] >> 16;
    if (has_attr && np >= 1u) {
      const uint32_t cidx = (y >> 1) * ac0 + (x >> 1);                                            // chroma nearest neighbour
      const vpcc_color3 c = yuv10_to_rgb8_fast(gl(ay0)[y * as0 + x], gl(au0)[cidx], gl(av0)[cidx]);
      col[j][0] = (uint32_t)c.r | ((uint32_t)c.g << 8) | ((uint32_t)c.b << 16);
    }
    if (has_attr && np == 2u) {
      const uint32_t cidx = (y >> 1) * ac1 + (x >> 1);
      const vpcc_color3 c = yuv10_to_rgb8_fast(gl(ay1)[y * as1 + x], gl(au1)[cidx], gl(av1)[cidx]);
      col[j][1] = (uint32_t)c.r | ((uint32_t)c.g << 8) | ((uint32_t)c.b << 16);
    }
  }
  // 4. look-back
  if (wave == 0 && group) {
    uint32_t excl = 0, spins = 0;
    for (int32_t hi = (int32_t)group - 1; hi >= 0;) {                       // predecessors hi, hi - 1, ..., 64 at a time
      const int32_t u = hi - (int32_t)lane;
      const uint64_t s = u >= 0 ? __hip_atomic_load(gl(state) + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (((uint64_t)gen << kGenGenShift) | kGenPrefix);
      const uint32_t st = gen_status(s, gen);
      const uint64_t pending = __ballot(st == 0u), prefix = __ballot(st == 2u);
      const uint32_t first_prefix = prefix ? (uint32_t)__builtin_ctzll(prefix) : 64u;
      // everything in front of the nearest PREFIX must have arrived
      const uint64_t needed = first_prefix >= 63u ? ~0ull : ((2ull << first_prefix) - 1ull);
      if (pending & needed) {
        if (++spins > kGenSpinLimit) { if (lane == 0) atomicOr(f.error_flag, kErrorSpinLimit); break; }      // never in a healthy run
        __builtin_amdgcn_s_sleep(1);
        continue;
      }
      uint32_t v = lane <= first_prefix ? (uint32_t)s : 0u;
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
      excl += v;
      if (prefix) break;
      hi -= 64;
    }
    if (lane == 0) {
      prefix_s = excl;
      __hip_atomic_store(glw(state) + group, ((uint64_t)gen << kGenGenShift) | kGenPrefix | (uint64_t)(excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else if (tid == 0 && group == 0) {
    prefix_s = 0;
  }
  gen_sync_lds();
  const uint32_t base = prefix_s;
  if (group == n_groups - 1u && tid == 0) *glw(f.n_points) = base + total;
  // 5. unit by unit: its points [first, first + utot), clipped to the caller's capacity, laid out in LDS as they will lie in memory
#pragma unroll
  for (uint32_t j = 0; j < kU; ++j) {
    if (!utot[j]) continue;                                                 // (uniform: nothing to stage, no barrier)
    const uint32_t u_first = base + ubase[j];
    const uint32_t lo = min(u_first, f.capacity), n_out = min(u_first + utot[j], f.capacity) - lo;
    const uint32_t sx = (uint32_t)((uintptr_t)((const unsigned char*)f.out_xyz + (size_t)lo * 6u) & 15u);
    const uint32_t sc = (uint32_t)((uintptr_t)((const unsigned char*)f.out_rgb + (size_t)lo * 3u) & 15u);
    const uint32_t si = (uint32_t)((uintptr_t)((const unsigned char*)f.out_patch + (size_t)lo * 2u) & 15u);
    const uint32_t k0 = before[j], np = patch_n[j] >> 16;                   // rank inside the unit
    auto put = [&](uint32_t k, uint32_t xy, uint32_t z, uint32_t c) {
      uint16_t* q = (uint16_t*)(stage_xyz + sx + 6u * k);                   // (2-byte aligned: arrays are 256-byte aligned, elements 6 bytes)
      q[0] = (uint16_t)xy; q[1] = (uint16_t)(xy >> 16); q[2] = (uint16_t)z;
      if (has_attr) { unsigned char* o = stage_rgb + sc + 3u * k; o[0] = (unsigned char)c; o[1] = (unsigned char)(c >> 8); o[2] = (unsigned char)(c >> 16); }
      if (f.out_patch) *(uint16_t*)(stage_idx + si + 2u * k) = (uint16_t)patch_n[j];                 // partition, codec.rs:452
    };
    if (np >= 1u && k0 < n_out) put(k0, pxy[j][0], pz[j], col[j][0]);      // (n_out: the caller's capacity)
    if (np == 2u && k0 + 1u < n_out) put(k0 + 1u, pxy[j][1], pz[j] >> 16, col[j][1]);
    gen_sync_lds();
    copy_out(stage_xyz, (unsigned char*)f.out_xyz, lo, n_out, 6u);
    if (f.has_attr) copy_out(stage_rgb, (unsigned char*)f.out_rgb, lo, n_out, 3u);
    if (f.out_patch) copy_out(stage_idx, (unsigned char*)f.out_patch, lo, n_out, 2u);
    gen_sync_lds();                                                        // (the stage is the next unit's)
  }
}

// ---------------------------------------------------------- k_general_blocks
// k_general for the frames every stream has (FrameShape::block_units: block side a power of two in [16, 256], occupancy precision a
// power of two, strides <= 65536, every patch's three axes distinct): a unit is a 256-pixel chunk of ONE virtual block, so
//   - the block's record and the ownership test (src/codec.rs:379) are SCALAR loads, and a unit whose block belongs to another
//     patch costs its lanes nothing (k_general asked for the samples of every pixel of every virtual block: a third of them
//     lie in blocks that are somebody else's);
//   - the samples of all the group's owned units are requested in one straight stretch of code (k_general's per-lane branches
//     made the compiler wait for one unit's samples before it asked for the next unit's record: fifteen dependent round trips
//     per group of five units);
//   - addresses are {scalar base, 32-bit lane offset} built with 24-bit multiply-adds (32-bit multiplies and 64-bit address
//     arithmetic run at a quarter of the rate, and were a third of k_general's vector time);
//   - a point is kept as (normal coordinate of each layer, canvas pixel): the tangent and bitangent coordinates are the same for
//     both layers and recomputed when the point is staged, where the axis assignment (src/decoder.rs:871-888) is three LDS
//     addresses instead of nine selects per point; two points of a pixel differ iff their normal coordinates do.
// Steps 2, 4 and 5 (ranks, look-back, staging) are k_general's.
#ifdef VPCC_GEN_STAMPS                                    // tools/exp_general_stamps.py: where a group's time goes (never in the product)
constexpr uint32_t kGenStampEvery = 397, kGenStampSlots = 512;
__device__ unsigned long long g_gen_stamps[kGenStampSlots][16];
#define VPCC_GSTAMP(k) do { __builtin_amdgcn_sched_barrier(0); if (blockIdx.x % kGenStampEvery == 0 && blockIdx.x / kGenStampEvery < kGenStampSlots && threadIdx.x == 0) \
    g_gen_stamps[blockIdx.x / kGenStampEvery][(k)] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define VPCC_GSTAMP(k) do { } while (0)
#endif
#ifndef VPCC_GENB_WAVES
#define VPCC_GENB_WAVES 8
#endif
__global__ __launch_bounds__(kGenThreads) __attribute__((amdgpu_waves_per_eu(VPCC_GENB_WAVES, 8))) void k_general_blocks(const DevFrame* __restrict__ frames, uint32_t first, uint32_t count, uint32_t groups_per_frame, uint32_t interleave, uint32_t lanes, uint32_t gen) {
  const GenWork work = gen_work(first, count, groups_per_frame, interleave, lanes);
  if (!work.any) return;
  const VPCC_CONST DevFrame& f = *(const VPCC_CONST DevFrame*)(frames + work.frame);   // (host-written: scalar loads)
  constexpr uint32_t kU = kGenBlockUnits;
  const uint32_t log2R = 31u - (uint32_t)__builtin_clz(f.R), log2C = 2u * log2R - 8u;      // R * R / 256 chunks per virtual block
  const uint32_t n_units = f.n_vblocks << log2C, n_groups = (n_units + kU - 1u) / kU;
  __shared__ uint32_t wave_sum[kU][4];
  __shared__ uint32_t prefix_s;
  constexpr uint32_t kS = kGenBlockStage;                                   // units staged together
  __shared__ __attribute__((aligned(16))) unsigned char stage_xyz[kS * 2 * kGenThreads * 6 + 32];
  __shared__ __attribute__((aligned(16))) unsigned char stage_rgb[kS * 2 * kGenThreads * 3 + 32];
  __shared__ __attribute__((aligned(16))) unsigned char stage_idx[kS * 2 * kGenThreads * 2 + 32];
  const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
  const uint32_t ps = f.prec_shift;
  // the thread's pixel of a block's chunk c: (pu0, pv0 + c * (256 / R)) — R <= 256: a chunk is whole rows
  const uint32_t pu0 = tid & (f.R - 1u), pv0 = tid >> log2R;
  const ColourKeys keys = vpcc_colour_keys();
  const bool two = f.map_count > 1u, absolute = f.absolute_d1 != 0u, has_attr = f.has_attr != 0u;
  uint64_t* const state = reinterpret_cast<uint64_t*>(f.vb_count);          // one {generation | status | value} word per group
  // One group per workgroup, no loop: around a loop the compiler must assume that a register it is about to write may still be the
  // target of a load of the previous trip, and waits for ALL outstanding loads — in the middle of step 1b.
  const uint32_t group = work.group;
  if (group >= n_groups) return;
  VPCC_GSTAMP(0);
  // 1a. the units' blocks, and which of them own their canvas block (uniform; no branch: all records, then all owners, are
  // requested together — a unit past the frame's last reads block 0 and is nobody's)
  VBlock b[kU];
  bool own[kU];
#pragma unroll
  for (uint32_t j = 0; j < kU; ++j) {
    const uint32_t unit = group * kU + j;
    own[j] = unit < n_units;
    b[j] = cload(f.vblocks + (own[j] ? unit >> log2C : 0u));
  }
#pragma unroll
  for (uint32_t j = 0; j < kU; ++j) own[j] = own[j] & (cload(f.block_to_patch + b[j].canvas_block) == (uint32_t)b[j].patch + 1u);
  VPCC_GSTAMP(1);
  // 1b. the thread's pixel of every owned unit: occupancy and both depths (the pixel lies inside the canvas and the planes
  // cover it: validate_frame; one map: geo[1] is the descriptor's alias of geo[0])
  uint32_t cxy[kU], s_occ[kU], s_d0[kU], s_d1[kU];
#pragma unroll
  for (uint32_t j = 0; j < kU; ++j) {
    cxy[j] = s_occ[j] = s_d0[j] = s_d1[j] = 0;
    if (own[j]) {
      const int32_t pu = (int32_t)pu0, pv = (int32_t)(pv0 + ((((group * kU + j) & ((1u << log2C) - 1u)) << 8) >> log2R));
      const int32_t cux = (int32_t)(b[j].coef & 3u) - 1, cvx = (int32_t)((b[j].coef >> 2) & 3u) - 1;   // patch_to_canvas, src/decoder.rs:841-867
      const int32_t cuy = (int32_t)((b[j].coef >> 4) & 3u) - 1, cvy = (int32_t)(b[j].coef >> 6) - 1;
      const uint32_t x = (uint32_t)((int32_t)b[j].x0 + __mul24(cux, pu) + __mul24(cvx, pv));
      const uint32_t y = (uint32_t)((int32_t)b[j].y0 + __mul24(cuy, pu) + __mul24(cvy, pv));
      cxy[j] = x | (y << 16);                                                                   // (canvas sides <= 32768)
      s_occ[j] = ld32(f.occ, __umul24(y >> ps, f.occ_stride) + (x >> ps));                     // src/codec.rs:288-301, 393
      s_d0[j] = ld32(f.geo[0], (__umul24(y, f.geo_stride[0]) + x) * 2u);
      s_d1[j] = ld32(f.geo[1], (__umul24(y, f.geo_stride[1]) + x) * 2u);
    }
  }
  VPCC_GSTAMP(2);
  // (Opaque to the compiler: it folds the first operation on a sample into the branch that loads it, and with it the wait for
  // the sample — one round trip per unit instead of one per group.)
#pragma unroll
  for (uint32_t j = 0; j < kU; ++j) asm volatile("" : "+v"(s_occ[j]), "+v"(s_d0[j]), "+v"(s_d1[j]));
  VPCC_GSTAMP(3);
  // 1c. the normal coordinate of the pixel's points (Patch::generate_point, src/decoder.rs:871-888; generate_points,
  // src/codec.rs:517-565) and how many there are (codec.rs:422-427)
  uint32_t nn[kU], np[kU];
#pragma unroll
  for (uint32_t j = 0; j < kU; ++j) {
    nn[j] = np[j] = 0;
    if (own[j]) {
      const bool mode0 = (b[j].axes_mode >> 6) == 0u;
      const uint32_t pd1 = b[j].d1, d0 = s_d0[j] >> 2, d1 = s_d1[j] >> 2;                       // depth / 4, codec.rs:534
      const uint32_t n0 = (mode0 ? d0 + pd1 : (pd1 > d0 ? pd1 : d0) - d0) & 0xFFFFu;            // `as u16`
      uint32_t n1 = n0;
      if (two) n1 = (absolute ? (mode0 ? d1 + pd1 : (pd1 > d1 ? pd1 : d1) - d1) : (mode0 ? n0 + d1 : n0 - d1)) & 0xFFFFu;
      nn[j] = n0 | (n1 << 16);
      np[j] = s_occ[j] ? (n1 != n0 ? 2u : 1u) : 0u;
    }
  }
  // 2. ranks inside the units
  uint32_t before[kU];
#pragma unroll
  for (uint32_t j = 0; j < kU; ++j) {
    before[j] = 0;
    if (own[j]) {
      const uint64_t m1 = __ballot(np[j] >= 1u), m2 = __ballot(np[j] == 2u);
      before[j] = mbcnt(m1) + mbcnt(m2);
      if (lane == 0) wave_sum[j][wave] = (uint32_t)__popcll(m1) + (uint32_t)__popcll(m2);
    }
  }
  gen_sync_lds();
  VPCC_GSTAMP(4);
  uint32_t ubase[kU], utot[kU], total = 0;                                   // a unit's first rank in the group, its points
#pragma unroll
  for (uint32_t j = 0; j < kU; ++j) {
    uint32_t wb = 0, t = 0;
    if (own[j]) {
#pragma unroll
      for (uint32_t w = 0; w < 4; ++w) {
        const uint32_t x = wave_sum[j][w];
        if (w < wave) wb += x;
        t += x;
      }
    }
    ubase[j] = total;
    utot[j] = t;
    before[j] = (before[j] + wb) | (np[j] << 16);                          // rank in the unit (<= 512) | points
    total += t;
  }
  if (tid == 0)
    __hip_atomic_store(glw(state) + group, ((uint64_t)gen << kGenGenShift) | (group ? kGenAggregate : kGenPrefix) | total, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  // 3. colours of the thread's points (color_point_cloud, src/codec.rs:626-644: layer l of the attribute video for point l),
  // r | g << 8 | b << 16: a layer's samples of all units are requested together
  uint32_t col[kU][2];
#pragma unroll
  for (uint32_t l = 0; l < 2; ++l) {
    const uint16_t* const ay = f.attr_y[l]; const uint16_t* const au = f.attr_u[l]; const uint16_t* const av = f.attr_v[l];
    const uint32_t as = f.attr_stride[l], ac = f.attr_cstride[l];
    uint32_t sy[kU], su[kU], sv[kU];
#pragma unroll
    for (uint32_t j = 0; j < kU; ++j) {
      sy[j] = su[j] = sv[j] = 0;
      if (has_attr && (before[j] >> 16) > l) {
        const uint32_t x = cxy[j] & 0xFFFFu, y = cxy[j] >> 16;
        const uint32_t cidx = (__umul24(y >> 1, ac) + (x >> 1)) * 2u;                           // chroma nearest neighbour
        sy[j] = ld32(ay, (__umul24(y, as) + x) * 2u);
        su[j] = ld32(au, cidx);
        sv[j] = ld32(av, cidx);
      }
    }
#pragma unroll
    for (uint32_t j = 0; j < kU; ++j) asm volatile("" : "+v"(sy[j]), "+v"(su[j]), "+v"(sv[j]));
#pragma unroll
    for (uint32_t j = 0; j < kU; ++j) {
      col[j][l] = 0;
      if (has_attr && (before[j] >> 16) > l) {
        col[j][l] = yuv10_to_rgb8_packed(sy[j], su[j], sv[j], keys);
      }
    }
    VPCC_GSTAMP(5 + l);
  }
  // 4. look-back
  if (wave == 0 && group) {
    uint32_t excl = 0, spins = 0;
    for (int32_t hi = (int32_t)group - 1; hi >= 0;) {                       // predecessors hi, hi - 1, ..., 64 at a time
      const int32_t u = hi - (int32_t)lane;
      const uint64_t s = u >= 0 ? __hip_atomic_load(gl(state) + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (((uint64_t)gen << kGenGenShift) | kGenPrefix);
      const uint32_t st = gen_status(s, gen);
      const uint64_t pending = __ballot(st == 0u), prefix = __ballot(st == 2u);
      const uint32_t first_prefix = prefix ? (uint32_t)__builtin_ctzll(prefix) : 64u;
      const uint64_t needed = first_prefix >= 63u ? ~0ull : ((2ull << first_prefix) - 1ull);   // everything in front of the nearest PREFIX
      if (pending & needed) {
        if (++spins > kGenSpinLimit) { if (lane == 0) atomicOr(f.error_flag, kErrorSpinLimit); break; }      // never in a healthy run
        __builtin_amdgcn_s_sleep(1);
        continue;
      }
      uint32_t v = lane <= first_prefix ? (uint32_t)s : 0u;
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
      excl += v;
      if (prefix) break;
      hi -= 64;
    }
    if (lane == 0) {
      prefix_s = excl;
      __hip_atomic_store(glw(state) + group, ((uint64_t)gen << kGenGenShift) | kGenPrefix | (uint64_t)(excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else if (tid == 0 && group == 0) {
    prefix_s = 0;
  }
  VPCC_GSTAMP(7);
  gen_sync_lds();
  VPCC_GSTAMP(8);
  const uint32_t base = prefix_s;
  if (group == n_groups - 1u && tid == 0) *glw(f.n_points) = base + total;
  // 5. kS units at a time: their points [first, first + pts) — one stretch of the output, the units follow each other —, clipped
  // to the caller's capacity, laid out in LDS as they will lie in memory (a unit at a time: two barriers per unit, eight per group)
#pragma unroll
  for (uint32_t j0 = 0; j0 < kU; j0 += kS) {
    uint32_t pts = 0;
#pragma unroll
    for (uint32_t j = j0; j < j0 + kS && j < kU; ++j) pts += utot[j];
    if (!pts) continue;                                                     // (uniform: nothing to stage, no barrier)
    const uint32_t p_first = base + ubase[j0];
    const uint32_t lo = min(p_first, f.capacity), n_out = min(p_first + pts, f.capacity) - lo;
    const uint32_t sx = (uint32_t)((uintptr_t)((const unsigned char*)f.out_xyz + (size_t)lo * 6u) & 15u);
    const uint32_t sc = (uint32_t)((uintptr_t)((const unsigned char*)f.out_rgb + (size_t)lo * 3u) & 15u);
    const uint32_t si = (uint32_t)((uintptr_t)((const unsigned char*)f.out_patch + (size_t)lo * 2u) & 15u);
#pragma unroll
    for (uint32_t j = j0; j < j0 + kS && j < kU; ++j) {
      if (!utot[j]) continue;
      const uint32_t k0 = ubase[j] - ubase[j0] + (before[j] & 0xFFFFu), npj = before[j] >> 16;      // rank among the staged points
      // tangent and bitangent coordinate of the pixel (src/decoder.rs:875-876), and where the three coordinates go
      const uint32_t pv = pv0 + ((((group * kU + j) & ((1u << log2C) - 1u)) << 8) >> log2R);
      const uint32_t t = __umul24(pu0, b[j].lod_x) + b[j].t0, bt = __umul24(pv, b[j].lod_y) + b[j].b0;
      const uint32_t at_n = sx + 2u * (b[j].axes_mode & 3u), at_t = sx + 2u * ((b[j].axes_mode >> 2) & 3u), at_b = sx + 2u * ((b[j].axes_mode >> 4) & 3u);
      auto put = [&](uint32_t k, uint32_t n, uint32_t c) {
        unsigned char* q = stage_xyz + 6u * k;                              // (2-byte aligned: arrays are 256-byte aligned, elements 6 bytes)
        *(uint16_t*)(q + at_n) = (uint16_t)n; *(uint16_t*)(q + at_t) = (uint16_t)t; *(uint16_t*)(q + at_b) = (uint16_t)bt;
        if (has_attr) { unsigned char* o = stage_rgb + sc + 3u * k; o[0] = (unsigned char)c; o[1] = (unsigned char)(c >> 8); o[2] = (unsigned char)(c >> 16); }
        if (f.out_patch) *(uint16_t*)(stage_idx + si + 2u * k) = b[j].patch;                   // partition, codec.rs:452
      };
      if (npj >= 1u && k0 < n_out) put(k0, nn[j], col[j][0]);              // (n_out: the caller's capacity)
      if (npj == 2u && k0 + 1u < n_out) put(k0 + 1u, nn[j] >> 16, col[j][1]);
    }
    gen_sync_lds();
    copy_out(stage_xyz, (unsigned char*)f.out_xyz, lo, n_out, 6u);
    if (has_attr) copy_out(stage_rgb, (unsigned char*)f.out_rgb, lo, n_out, 3u);
    if (f.out_patch) copy_out(stage_idx, (unsigned char*)f.out_patch, lo, n_out, 2u);
    if (j0 + kS < kU) gen_sync_lds();                                      // (the stage is the next units')
    VPCC_GSTAMP(9 + j0 / kS);
  }
  VPCC_GSTAMP(15);
}
#ifdef VPCC_GEN_STAMPS
extern "C" int vpcc_debug_general_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gen_stamps), sizeof(unsigned long long) * kGenStampSlots * 16) == hipSuccess ? 0 : 1;
}
#endif

// ---------------------------------------------------- k_upsample_occupancy
// tile.occupancy_map, src/codec.rs:288-301 (kept for API completeness; the main path never builds it).
__global__ __launch_bounds__(256) void k_upsample_occupancy(const DevFrame* __restrict__ frames, uint32_t frame,
                                                            uint8_t* __restrict__ out) {
  const DevFrame& f = frames[frame];
  const uint32_t x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x < f.width) out[(size_t)y * f.width + x] = gl(f.occ)[(y / f.prec) * f.occ_stride + (x / f.prec)];
}

// ----------------------------------------------------------------- launchers
void launch_block_owner(const DevFrame* d_frames, uint32_t first, uint32_t count, uint32_t max_vb, uint32_t max_samples, void* stream) {
  if (!count || !max_vb) return;
  if (max_samples <= 32u)
    hipLaunchKernelGGL(k_block_owner<1>, dim3((max_vb + 255u) / 256u, count), dim3(256), 0, (hipStream_t)stream, d_frames, first);
  else
    hipLaunchKernelGGL(k_block_owner<64>, dim3((max_vb + 3) / 4, count), dim3(256), 0, (hipStream_t)stream, d_frames, first);
}
void launch_general(const DevFrame* d_frames, uint32_t first, uint32_t count, uint32_t max_units, uint32_t gen, bool block_units, void* stream) {
  if (!count || !max_units) return;
  if (block_units) {
    const uint32_t groups = (max_units + kGenBlockUnits - 1u) / kGenBlockUnits;
    const GenShape shape = gen_shape(count, groups, kGenInterleave);
    hipLaunchKernelGGL(k_general_blocks, dim3(shape.grid), dim3(kGenThreads), 0, (hipStream_t)stream, d_frames, first, count, groups, shape.interleave, shape.lanes, gen);
    return;
  }
  const uint32_t groups = (max_units + kGenUnitsPerGroup - 1u) / kGenUnitsPerGroup;
  const GenShape shape = gen_shape(count, groups, kGenInterleave);
  hipLaunchKernelGGL(k_general, dim3(shape.grid), dim3(kGenThreads), 0, (hipStream_t)stream, d_frames, first, count, groups, shape.interleave, shape.lanes, gen);
}
// Plane ingest by the GPU itself: every workgroup pulls 64-KB pieces of page-locked HOST memory over PCIe (zero-copy
// reads, 16 B per lane, coalesced) and stores them in HBM.  One launch moves all planes of a gof: 57 GB/s with 64
// workgroups and more — the rate of ONE big hipMemcpyAsync, where the 1 280 plane-sized copies (0.1-3.6 MB) of a
// 128-frame unit reach 34 GB/s on the copy engines (tools/micro/zero_copy.hip, profiles/r04/zero_copy.txt).
// A piece's source and destination are congruent modulo 16 (the runtime places the destination so); up to 15 bytes in
// front of and behind the aligned body go byte by byte — single-byte reads over PCIe, which is why only a plane's first
// and last piece have any (with eight in front of and behind EVERY piece: 2 060 instead of 2 190 frames/s end to end).
// Measured and not kept: 256-KB pieces with eight loads in flight per lane 2 110; the next piece's descriptor fetched
// beside the current piece's data 2 130-2 150.
__global__ __launch_bounds__(256) void k_ingest_planes(const IngestPiece* __restrict__ pieces, uint32_t n) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  for (uint32_t c = blockIdx.x; c < n; c += gridDim.x) {
    const IngestPiece k = pieces[c];
    const VPCC_GLOBAL unsigned char* src = (const VPCC_GLOBAL unsigned char*)k.src;
    VPCC_GLOBAL unsigned char* dst = (VPCC_GLOBAL unsigned char*)k.dst;
    const uint32_t head = min((uint32_t)((16u - ((uintptr_t)k.src & 15u)) & 15u), k.bytes);
    const uint32_t n16 = (k.bytes - head) >> 4, tail = (k.bytes - head) & 15u;
    if (threadIdx.x < head) dst[threadIdx.x] = src[threadIdx.x];
    if (threadIdx.x < tail) dst[head + 16u * n16 + threadIdx.x] = src[head + 16u * n16 + threadIdx.x];
    const VPCC_GLOBAL u32x4* s16 = (const VPCC_GLOBAL u32x4*)(src + head);
    VPCC_GLOBAL u32x4* d16 = (VPCC_GLOBAL u32x4*)(dst + head);
    for (uint32_t i = threadIdx.x; i < n16; i += 1024u) {
      u32x4 v[4];
#pragma unroll
      for (uint32_t q = 0; q < 4; ++q) if (i + 256u * q < n16) v[q] = __builtin_nontemporal_load(s16 + i + 256u * q);
#pragma unroll
      for (uint32_t q = 0; q < 4; ++q) if (i + 256u * q < n16) d16[i + 256u * q] = v[q];
    }
  }
}
// Sixteen workgroups pull at the link's rate already (55 GB/s alone; 8 / 16 / 64 / 256 workgroups in the Decoder: 2 467 /
// 2 648 / 2 686 / 2 635 frames/s without downloads — profiles/r04/zero_copy.txt, e2e.txt).
constexpr uint32_t kIngestWorkgroups = 16;
void launch_ingest_planes(const IngestPiece* d_pieces, uint32_t n, void* stream) {
  if (!n) return;
  hipLaunchKernelGGL(k_ingest_planes, dim3(std::min<uint32_t>(n, kIngestWorkgroups)), dim3(256), 0, (hipStream_t)stream, d_pieces, n);
}

// The other direction: up to three arrays of one frame's result pushed into page-locked host memory by a kernel (the
// pieces by value: nothing to upload).  Beside the ingest kernel a hipMemcpyAsync device-to-host gets 17 GB/s, this 38.
struct PushList { IngestPiece p[3]; };
__global__ __launch_bounds__(256) void k_push_results(const PushList list) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
  for (uint32_t a = 0; a < 3; ++a) {
    const IngestPiece k = list.p[a];
    if (!k.bytes && !k.pad) continue;
    const size_t bytes = ((size_t)k.pad << 32) | k.bytes;                     // (an array may exceed 4 GB: pad holds the high half)
    const VPCC_GLOBAL unsigned char* src = (const VPCC_GLOBAL unsigned char*)k.src;
    VPCC_GLOBAL unsigned char* dst = (VPCC_GLOBAL unsigned char*)k.dst;
    // device arrays are 256-byte aligned; the host side decides: aligned body only when dst is 16-byte aligned too
    const bool wide = (((uintptr_t)k.src | (uintptr_t)k.dst) & 15u) == 0;
    const size_t n16 = wide ? bytes >> 4 : 0;
    const VPCC_GLOBAL u32x4* s16 = (const VPCC_GLOBAL u32x4*)src;
    VPCC_GLOBAL u32x4* d16 = (VPCC_GLOBAL u32x4*)dst;
    for (size_t i = (size_t)blockIdx.x * 1024u + threadIdx.x; i < n16; i += (size_t)gridDim.x * 1024u) {
      u32x4 v[4];
#pragma unroll
      for (uint32_t q = 0; q < 4; ++q) if (i + 256u * q < n16) v[q] = s16[i + 256u * q];
#pragma unroll
      for (uint32_t q = 0; q < 4; ++q) if (i + 256u * q < n16) __builtin_nontemporal_store(v[q], d16 + i + 256u * q);
    }
    for (size_t i = 16u * n16 + (size_t)blockIdx.x * 256u + threadIdx.x; i < bytes; i += (size_t)gridDim.x * 256u) dst[i] = src[i];
  }
}
void launch_push_results(const IngestPiece pieces[3], void* stream) {
  PushList l;
  for (int a = 0; a < 3; ++a) l.p[a] = pieces[a];
  hipLaunchKernelGGL(k_push_results, dim3(kIngestWorkgroups), dim3(256), 0, (hipStream_t)stream, l);
}

// The reconstruction kernel's output pattern, alone: 4 096 waves, each writing runs of 304 points — 1 824 B of positions into
// `xyz`, 912 B of colours into `rgb` — with the kernel's store instructions (12 B per lane non-temporal, colours from the
// even lanes, runs of neighbouring waves adjacent).  Its rate tells whether the two arrays lie in one kind of VRAM region
// or in two (vpcc_ctx_reserve); the bytes it writes mean nothing.
__global__ __launch_bounds__(256) void k_probe_outputs(unsigned char* __restrict__ xyz, unsigned char* __restrict__ rgb, uint32_t items) {
  typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const uint32_t groups = (items + 15u) / 16u;
  for (uint32_t g = blockIdx.x; g < groups; g += gridDim.x)
    for (uint32_t t = 0; t < 4; ++t) {
      const uint32_t i = g * 16u + t * 4u + wave;
      if (i >= items) continue;
      VPCC_GLOBAL unsigned char* px = (VPCC_GLOBAL unsigned char*)xyz + (size_t)i * 1824u;
      VPCC_GLOBAL unsigned char* pc = (VPCC_GLOBAL unsigned char*)rgb + (size_t)i * 912u;
      const int shift = (int)(((size_t)i * 304u) & 127u);
      for (int kk = 2 * (int)lane; kk < 304 + shift; kk += 128) {
        const int k = kk - shift;
        if (k < 0 || k + 1 >= 304) continue;
        const u32x3 v = {(uint32_t)k, lane, i};
        __builtin_nontemporal_store(v, (VPCC_GLOBAL u32x3*)(px + k * 6));
        if (!(lane & 1u) && k + 3 < 304) __builtin_nontemporal_store(v, (VPCC_GLOBAL u32x3*)(pc + k * 3));
      }
    }
}
void launch_probe_outputs(unsigned char* xyz, unsigned char* rgb, uint32_t items, void* stream) {
  hipLaunchKernelGGL(k_probe_outputs, dim3(1024), dim3(256), 0, (hipStream_t)stream, xyz, rgb, items);
}

// An empty kernel per translation unit with kernels of the per-frame path: the runtime loads a unit's code object at its
// first launch (milliseconds) — vpcc_ctx_create pays that, not the first gof.
__global__ void k_warm_kernels() {}
void launch_warm_kernels(void* stream) { hipLaunchKernelGGL(k_warm_kernels, dim3(1), dim3(64), 0, (hipStream_t)stream); }

void launch_upsample_occupancy(const DevFrame* d_frames, uint32_t frame, uint8_t* d_out, uint32_t width,
                               uint32_t height, void* stream) {
  if (!width || !height) return;
  hipLaunchKernelGGL(k_upsample_occupancy, dim3((width + 255) / 256, height), dim3(256), 0, (hipStream_t)stream,
                     d_frames, frame, d_out);
}

}  // namespace vpcc
