// vpcc_tiles.hip — single-pass, wave-per-tile reconstruction kernel for gfx950 (CDNA4, wave64).
//
// This is the production path for the common configuration (block size R = 16, Default/Swap
// patches, 8-byte aligned luma rows).  One launch per batch of frames; every plane is read once,
// every output byte written once.
//
// Work decomposition
//   item   = one virtual block that owns its canvas block (host-filtered, see vpcc_host.cpp), in
//            the reference's emission order (src/codec.rs:352-385);
//   wave   = 4 consecutive items, processed one after the other; lane l of the wave owns the 4
//            pixels u1 = 4*(l&3)..+3 of row v1 = l>>2 of the 16x16 block, in PATCH-LOCAL
//            coordinates, so "lane order, then pixel order inside the lane" IS the reference's
//            emission order for both orientations.  Default tiles read 8 B per lane per plane
//            (canvas rows); Swap tiles read the transposed pixels with four 2-B loads per plane;
//   group  = one 256-thread workgroup = 16 consecutive items = one ticket and one look-back word.
//
// Per group: (1) every wave loads occupancy + both geometry layers of its 4 items, then — only
// where occupied — both attribute layers (these loads fly during the look-back); (2) points are
// counted per item (D1 is dropped when it equals D0, src/codec.rs:422-427) and the group total is
// published; (3) wave 0 obtains the group's output offset by decoupled look-back over the earlier
// groups of the frame; (4) each wave compacts its items' points into LDS slots at their rank,
// colour converted on the way (src/codec.rs:626-644, 661-687), and streams the slots out with
// contiguous unaligned dwordx3 stores (2 points / 4 colours per lane).
//
// Cross-workgroup ordering is placement-independent (cdna_hip_programming.md §6 Guideline 16):
// groups are drawn from a per-frame TICKET counter, so a look-back only waits for tickets that
// running workgroups hold; state words are 8-byte {status,value} granules moved with relaxed
// agent-scope atomics.  XCD-aware blockIdx mapping is used for L2 locality only.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "vpcc_device.hpp"
#include "vpcc_devfn.hpp"

namespace vpcc {

namespace {

constexpr uint64_t kStatusShift = 62;
constexpr uint64_t kAggregate = 1ull << kStatusShift;
constexpr uint64_t kPrefix = 2ull << kStatusShift;
constexpr uint32_t kSpinLimit = 1u << 22;

__device__ __forceinline__ uint64_t st_load(const uint64_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_store(uint64_t* p, uint64_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Exclusive prefix of group `g` within its frame.  One full wave; same result in every lane.
__device__ uint32_t look_back_groups(const DevFrame& f, uint32_t g) {
  uint32_t excl = 0;
  int32_t idx = (int32_t)g - 1;
  const uint32_t lane = lane_id();
  while (idx >= 0) {
    const int32_t my = idx - (int32_t)lane;
    uint64_t s = kPrefix;
    if (my >= 0) {
      uint32_t spins = 0;
      s = st_load(f.scan_state + my);
      while ((s >> kStatusShift) == 0) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > kSpinLimit) {
          atomicOr(f.error_flag, 1u);
          s = kPrefix;
          break;
        }
        s = st_load(f.scan_state + my);
      }
    }
    const uint64_t pm = __ballot((s >> kStatusShift) == 2);
    const uint32_t firstp = pm ? (uint32_t)__builtin_ctzll(pm) : 64u;
    uint32_t v = lane <= firstp ? (uint32_t)s : 0u;
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    excl += v;
    if (pm) break;
    idx -= 64;
  }
  return excl;
}

struct Px4 { uint32_t lo, hi; };   // four u16 samples, pixel j in bits 16*(j&1) of (j<2 ? lo : hi)

template <int J>
__device__ __forceinline__ uint32_t px(const Px4& v) {
  return J == 0 ? (v.lo & 0xFFFFu) : J == 1 ? (v.lo >> 16) : J == 2 ? (v.hi & 0xFFFFu) : (v.hi >> 16);
}

struct TileRegs {
  Px4 g0, g1;          // geometry D0 / D1 samples of the lane's 4 pixels
  Px4 y0, y1;          // attribute luma, layer 0 / 1
  uint32_t u0, v0, u1, v1;   // chroma: sample for pixels 0,1 in the low half, for pixels 2,3 in the high half
  uint32_t occ;        // bit j: pixel j occupied
  uint32_t dup;        // bit j: D1 point equals D0 point (or single map): one point only
  uint32_t cnt;        // points this lane emits
};

__device__ __forceinline__ Px4 load4_row(const uint16_t* p) {       // 8-B aligned by construction
  const uint2 v = *reinterpret_cast<const uint2*>(p);
  return Px4{v.x, v.y};
}
__device__ __forceinline__ Px4 load4_col(const uint16_t* p, uint32_t stride) {
  const uint32_t a = p[0], b = p[stride], c = p[2 * stride], d = p[3 * stride];
  return Px4{a | (b << 16), c | (d << 16)};
}

// Coordinates of one point as the reference builds them (src/decoder.rs:871-888): assignment order
// normal, tangent, bitangent; `as u16` truncation.  Returns {x | y << 16, z}.
__device__ __forceinline__ uint2 pack_point(const TileItem& it, uint32_t n, uint32_t t, uint32_t b) {
  const uint32_t na = it.axes & 3u, ta = (it.axes >> 2) & 3u, ba = (it.axes >> 4) & 3u;
  uint32_t c[3];
#pragma unroll
  for (uint32_t a = 0; a < 3; ++a) {
    uint32_t v = 0;
    if (na == a) v = n;
    if (ta == a) v = t;
    if (ba == a) v = b;
    c[a] = v & 0xFFFFu;
  }
  return make_uint2(c[0] | (c[1] << 16), c[2]);
}

__device__ __forceinline__ uint32_t normal_of(const TileItem& it, uint32_t depth) {
  return (it.flags & kTileMode1) ? (it.d1 > depth ? it.d1 : depth) - depth : depth + it.d1;
}

// D1 point in relative mode (src/codec.rs:551-559): point0 with +-d1 on coordinate index normal_axis.
__device__ __forceinline__ uint2 relative_point(const TileItem& it, uint2 p0, uint32_t d1) {
  const uint32_t na = it.axes & 3u;
  uint32_t c[3] = {p0.x & 0xFFFFu, p0.x >> 16, p0.y & 0xFFFFu};
#pragma unroll
  for (uint32_t a = 0; a < 3; ++a)
    if (na == a) c[a] = ((it.flags & kTileMode1) ? c[a] - d1 : c[a] + d1) & 0xFFFFu;
  return make_uint2(c[0] | (c[1] << 16), c[2]);
}

__device__ __forceinline__ uint32_t pack_rgb(vpcc_color3 c) {
  return (uint32_t)c.r | ((uint32_t)c.g << 8) | ((uint32_t)c.b << 16);
}

// ---- phase 1a: occupancy + geometry of one item ------------------------------------------------
__device__ __forceinline__ void load_geometry(const DevFrame& f, const TileItem& it, bool valid, uint32_t lane,
                                              TileRegs& t) {
  t.occ = 0; t.dup = 0; t.cnt = 0;
  t.g0 = Px4{0, 0}; t.g1 = Px4{0, 0};
  if (!valid) return;
  const uint32_t q = lane & 3u, r = lane >> 2;
  const bool swap = it.flags & kTileSwap;
  // first pixel of the lane and the canvas step between its 4 pixels
  const uint32_t px0 = swap ? it.x0 + r : it.x0 + 4u * q;
  const uint32_t py0 = swap ? it.y0 + 4u * q : it.y0 + r;
  if (f.prec >= 4) {                                  // the 4 pixels share one occupancy sample
    t.occ = f.occ[(py0 / f.prec) * f.occ_stride + px0 / f.prec] ? 0xFu : 0u;     // src/codec.rs:288-301, 393
  } else {
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
      const uint32_t x = swap ? px0 : px0 + j, y = swap ? py0 + j : py0;
      t.occ |= (f.occ[(y / f.prec) * f.occ_stride + x / f.prec] ? 1u : 0u) << j;
    }
  }
  if (t.occ == 0) return;
  if (!swap) {
    t.g0 = load4_row(f.geo[0] + py0 * f.geo_stride[0] + px0);
    if (f.map_count > 1) t.g1 = load4_row(f.geo[1] + py0 * f.geo_stride[1] + px0);
  } else {
    t.g0 = load4_col(f.geo[0] + py0 * f.geo_stride[0] + px0, f.geo_stride[0]);
    if (f.map_count > 1) t.g1 = load4_col(f.geo[1] + py0 * f.geo_stride[1] + px0, f.geo_stride[1]);
  }
}

// ---- phase 1b: attribute samples of the occupied lanes ------------------------------------------
__device__ __forceinline__ void load_attributes(const DevFrame& f, const TileItem& it, uint32_t lane, TileRegs& t) {
  t.y0 = Px4{0, 0}; t.y1 = Px4{0, 0};
  t.u0 = t.v0 = t.u1 = t.v1 = 0;
  if (t.occ == 0 || !f.has_attr) return;
  const uint32_t q = lane & 3u, r = lane >> 2;
  const bool swap = it.flags & kTileSwap;
  const uint32_t px0 = swap ? it.x0 + r : it.x0 + 4u * q;
  const uint32_t py0 = swap ? it.y0 + 4u * q : it.y0 + r;
  if (!swap) {
    // chroma nearest neighbour (src/decoder.rs:977): pixels 0,1 -> sample (px0/2), pixels 2,3 -> next
    const uint32_t c0 = (py0 >> 1) * f.attr_cstride[0] + (px0 >> 1);
    t.y0 = load4_row(f.attr_y[0] + py0 * f.attr_stride[0] + px0);
    t.u0 = *reinterpret_cast<const uint32_t*>(f.attr_u[0] + c0);
    t.v0 = *reinterpret_cast<const uint32_t*>(f.attr_v[0] + c0);
    if (f.map_count > 1) {
      const uint32_t c1 = (py0 >> 1) * f.attr_cstride[1] + (px0 >> 1);
      t.y1 = load4_row(f.attr_y[1] + py0 * f.attr_stride[1] + px0);
      t.u1 = *reinterpret_cast<const uint32_t*>(f.attr_u[1] + c1);
      t.v1 = *reinterpret_cast<const uint32_t*>(f.attr_v[1] + c1);
    }
  } else {
    // pixels run down a canvas column: 0,1 share chroma row py0/2, pixels 2,3 the next one
    const uint32_t c0 = (py0 >> 1) * f.attr_cstride[0] + (px0 >> 1);
    t.y0 = load4_col(f.attr_y[0] + py0 * f.attr_stride[0] + px0, f.attr_stride[0]);
    t.u0 = (uint32_t)f.attr_u[0][c0] | ((uint32_t)f.attr_u[0][c0 + f.attr_cstride[0]] << 16);
    t.v0 = (uint32_t)f.attr_v[0][c0] | ((uint32_t)f.attr_v[0][c0 + f.attr_cstride[0]] << 16);
    if (f.map_count > 1) {
      const uint32_t c1 = (py0 >> 1) * f.attr_cstride[1] + (px0 >> 1);
      t.y1 = load4_col(f.attr_y[1] + py0 * f.attr_stride[1] + px0, f.attr_stride[1]);
      t.u1 = (uint32_t)f.attr_u[1][c1] | ((uint32_t)f.attr_u[1][c1 + f.attr_cstride[1]] << 16);
      t.v1 = (uint32_t)f.attr_v[1][c1] | ((uint32_t)f.attr_v[1][c1 + f.attr_cstride[1]] << 16);
    }
  }
}

// ---- phase 2: which D1 points are duplicates, and the lane's point count -----------------------
template <int J>
__device__ __forceinline__ void classify_pixel(const DevFrame& f, const TileItem& it, bool normal_visible,
                                               TileRegs& t) {
  if (!((t.occ >> J) & 1u)) return;
  bool dup = true;                                     // single map: D0 only
  if (f.map_count > 1) {
    const uint32_t d0 = px<J>(t.g0) >> 2, d1 = px<J>(t.g1) >> 2;      // depth / 4, src/codec.rs:534, 548
    if (f.absolute_d1)
      dup = !normal_visible || ((normal_of(it, d0) ^ normal_of(it, d1)) & 0xFFFFu) == 0;
    else
      dup = d1 == 0;                                   // u16 += / -= d1 leaves the point unchanged only for 0
  }
  t.dup |= (dup ? 1u : 0u) << J;
  t.cnt += dup ? 1u : 2u;
}

// ---- phase 4: emit the lane's points of pixel J into the staging slots -------------------------
template <int J>
__device__ __forceinline__ void emit_pixel(const DevFrame& f, const TileItem& it, const TileRegs& t, uint32_t lane,
                                           uint32_t& rank, uint2* sx, uint32_t* sc) {
  if (!((t.occ >> J) & 1u)) return;
  const uint32_t du = 4u * (lane & 3u) + J, dv = lane >> 2;            // patch-local offsets inside the block
  const uint32_t tg = it.tb + du * it.lod_x, bt = it.bb + dv * it.lod_y;
  const uint32_t d0 = px<J>(t.g0) >> 2;
  const uint2 p0 = pack_point(it, normal_of(it, d0), tg, bt);
  sx[rank] = p0;
  if (f.has_attr) {
    const uint32_t u = J < 2 ? (t.u0 & 0xFFFFu) : (t.u0 >> 16), v = J < 2 ? (t.v0 & 0xFFFFu) : (t.v0 >> 16);
    sc[rank] = pack_rgb(yuv10_to_rgb8_fast((uint16_t)px<J>(t.y0), (uint16_t)u, (uint16_t)v));
  }
  ++rank;
  if (!((t.dup >> J) & 1u)) {
    const uint32_t d1 = px<J>(t.g1) >> 2;
    sx[rank] = f.absolute_d1 ? pack_point(it, normal_of(it, d1), tg, bt) : relative_point(it, p0, d1);
    if (f.has_attr) {
      const uint32_t u = J < 2 ? (t.u1 & 0xFFFFu) : (t.u1 >> 16), v = J < 2 ? (t.v1 & 0xFFFFu) : (t.v1 >> 16);
      sc[rank] = pack_rgb(yuv10_to_rgb8_fast((uint16_t)px<J>(t.y1), (uint16_t)u, (uint16_t)v));
    }
    ++rank;
  }
}

struct __attribute__((packed)) U3 { uint32_t a, b, c; };

// Streams `n` staged points to out_xyz/out_rgb[base ...): 2 points (12 B) resp. 4 colours (12 B) per lane.
__device__ __forceinline__ void flush_item(const DevFrame& f, const TileItem& it, uint32_t base, uint32_t n,
                                           uint32_t lane, const uint2* sx, const uint32_t* sc) {
  unsigned char* gx = reinterpret_cast<unsigned char*>(f.out_xyz) + (size_t)base * 6u;
  for (uint32_t pi = lane; 2u * pi < n; pi += 64u) {
    const uint4 a = *reinterpret_cast<const uint4*>(sx + 2u * pi);     // two 8-B slots
    if (2u * pi + 1u < n) {
      const U3 o{a.x, (a.y & 0xFFFFu) | (a.z << 16), (a.z >> 16) | (a.w << 16)};
      __builtin_memcpy(gx + 12u * pi, &o, 12);
    } else {
      const uint32_t x = a.x;
      const uint16_t z = (uint16_t)a.y;
      __builtin_memcpy(gx + 12u * pi, &x, 4);
      __builtin_memcpy(gx + 12u * pi + 4, &z, 2);
    }
    if (f.out_patch) {                                                 // partition, src/codec.rs:452
      f.out_patch[base + 2u * pi] = it.patch;
      if (2u * pi + 1u < n) f.out_patch[base + 2u * pi + 1u] = it.patch;
    }
  }
  if (!f.has_attr) return;
  unsigned char* gc = reinterpret_cast<unsigned char*>(f.out_rgb) + (size_t)base * 3u;
  for (uint32_t qi = lane; 4u * qi < n; qi += 64u) {
    const uint4 c = *reinterpret_cast<const uint4*>(sc + 4u * qi);     // four 4-B slots (r,g,b,-)
    const uint32_t left = n - 4u * qi;
    if (left >= 4u) {
      const U3 o{(c.x & 0xFFFFFFu) | (c.y << 24), ((c.y >> 8) & 0xFFFFu) | (c.z << 16), ((c.z >> 16) & 0xFFu) | (c.w << 8)};
      __builtin_memcpy(gc + 12u * qi, &o, 12);
    } else {
      const uint32_t cc[3] = {c.x, c.y, c.z};
      for (uint32_t k = 0; k < left; ++k) {
        gc[12u * qi + 3u * k] = (unsigned char)cc[k];
        gc[12u * qi + 3u * k + 1] = (unsigned char)(cc[k] >> 8);
        gc[12u * qi + 3u * k + 2] = (unsigned char)(cc[k] >> 16);
      }
    }
  }
}

}  // namespace

// `variant`: 0 in production; timing-only ablation bits (VPCC_TILES_VARIANT): 1 skip look-back wait,
// 8 skip colour conversion, 16 skip global stores.
__global__ __launch_bounds__(256) void k_recon_tiles(const DevFrame* __restrict__ frames, uint32_t first,
                                                     uint32_t count, uint32_t groups_stride, uint32_t variant) {
  // XCD-aware placement (speed only): ids equal mod 8 share an XCD/L2; a frame stays on one label.
  const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
  const uint32_t fi = xcd + 8u * (slot / groups_stride);
  if (fi >= count) return;
  const DevFrame& f = frames[first + fi];

  __shared__ uint32_t s_group;
  __shared__ uint32_t s_base;
  __shared__ uint32_t s_tot[16];
  __shared__ __attribute__((aligned(16))) uint2 s_xyz[4][512];
  __shared__ __attribute__((aligned(16))) uint32_t s_rgb[4][512];

  if (threadIdx.x == 0) s_group = atomicAdd(f.ticket, 1u);
  __syncthreads();
  const uint32_t g = __builtin_amdgcn_readfirstlane(s_group);
  const uint32_t n_groups = (f.n_tiles + kTileItemsPerGroup - 1u) / kTileItemsPerGroup;
  if (g >= n_groups) return;                            // surplus workgroups of this frame

  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = lane_id();
  const uint32_t item0 = g * kTileItemsPerGroup + wave * 4u;

  // ---- phase 1: loads -------------------------------------------------------------------------
  TileItem it[4];
  TileRegs t[4];
  bool valid[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    valid[i] = item0 + i < f.n_tiles;
    it[i] = f.tiles[valid[i] ? item0 + i : 0u];
    load_geometry(f, it[i], valid[i], lane, t[i]);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) load_attributes(f, it[i], lane, t[i]);

  // ---- phase 2: count -------------------------------------------------------------------------
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t na = it[i].axes & 3u;
    const bool normal_visible = na != ((it[i].axes >> 2) & 3u) && na != ((it[i].axes >> 4) & 3u);
    classify_pixel<0>(f, it[i], normal_visible, t[i]);
    classify_pixel<1>(f, it[i], normal_visible, t[i]);
    classify_pixel<2>(f, it[i], normal_visible, t[i]);
    classify_pixel<3>(f, it[i], normal_visible, t[i]);
    uint32_t s = t[i].cnt;
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) s_tot[wave * 4u + i] = s;
  }
  __syncthreads();

  // ---- phase 3: publish the group total, look back ---------------------------------------------
  if (wave == 0) {
    uint32_t total = lane < 16u ? s_tot[lane] : 0u;
    for (int off = 8; off > 0; off >>= 1) total += __shfl_xor(total, off, 64);
    total = __shfl(total, 0, 64);
    if (lane == 0) st_store(f.scan_state + g, (g == 0 ? kPrefix : kAggregate) | total);
    uint32_t excl = 0;
    if (g != 0 && !(variant & 1u)) {
      excl = look_back_groups(f, g);
      if (lane == 0) st_store(f.scan_state + g, kPrefix | (uint64_t)(excl + total));
    }
    if (lane == 0) {
      s_base = excl;
      if (g + 1u == n_groups) *f.n_points = excl + total;              // tile.total_number_of_regular_points
    }
  }
  __syncthreads();
  uint32_t base = s_base;
  for (uint32_t k = 0; k < wave * 4u; ++k) base += s_tot[k];

  // ---- phase 4: compact through LDS, convert colour, stream out --------------------------------
  uint2* sx = s_xyz[wave];
  uint32_t* sc = s_rgb[wave];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t n = s_tot[wave * 4u + i];
    if (n != 0) {                                       // wave-uniform
      uint32_t incl = t[i].cnt;                         // exclusive scan of the lane counts = lane's first rank
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off, 64);
        if ((int)lane >= off) incl += v;
      }
      uint32_t rank = incl - t[i].cnt;
      if (variant & 8u) {
        for (uint32_t k = 0; k < t[i].cnt; ++k) { sx[rank + k] = make_uint2(rank, k); sc[rank + k] = k; }
      } else {
        emit_pixel<0>(f, it[i], t[i], lane, rank, sx, sc);
        emit_pixel<1>(f, it[i], t[i], lane, rank, sx, sc);
        emit_pixel<2>(f, it[i], t[i], lane, rank, sx, sc);
        emit_pixel<3>(f, it[i], t[i], lane, rank, sx, sc);
      }
    }
    __syncthreads();
    if (n != 0 && !(variant & 16u)) {
      const uint32_t room = base < f.capacity ? f.capacity - base : 0u;   // never write past the caller's arrays
      flush_item(f, it[i], base, n < room ? n : room, lane, sx, sc);
    }
    base += n;
    __syncthreads();
  }
}

void launch_tiles(const DevFrame* d_frames, uint32_t first, uint32_t count, uint32_t max_groups, void* stream) {
  if (!count || !max_groups) return;
  static const uint32_t variant = [] {
    const char* e = getenv("VPCC_TILES_VARIANT");
    return e ? (uint32_t)atoi(e) : 0u;
  }();
  const uint32_t frame_groups = (count + 7u) / 8u;
  const uint32_t grid = 8u * frame_groups * max_groups;
  hipLaunchKernelGGL(k_recon_tiles, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_frames, first, count, max_groups,
                     variant);
}

}  // namespace vpcc
