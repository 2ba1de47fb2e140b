// vpcc_kernels.hip — gfx950 (CDNA4, wave64) kernels of the V-PCC reconstruction path.
//
// GENERAL kernel sequence (any orientation, any block size):
//   k_block_owner  : block -> patch index      (reference src/codec.rs:205-250)
//   k_count        : points per virtual block  (enumeration of src/codec.rs:352-480, counting only)
//   k_scan         : exclusive prefix per frame
//   k_emit         : ordered emission of xyz (src/codec.rs:517-565, src/decoder.rs:871-888) fused with
//                    the attribute gather (src/codec.rs:569-658) and YUV->RGB (src/codec.rs:661-687)
// The W x H occupancy map of src/codec.rs:288-301, point_to_pixel and colors16bit are never
// materialised: occupancy is read through the low-resolution plane, and colour is fetched by the
// thread that emits the point.
//
// Everything is integer/byte work bound by HBM; there is no contraction, hence no MFMA.
// The only floating point is the reference's f64 colour matrix, compiled without contraction
// (-ffp-contract=off) so that it is bit-identical to the Rust code.
#include <hip/hip_runtime.h>

#include "vpcc_device.hpp"

namespace vpcc {

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
  const uint8_t occ = f.occ[(o.y / f.prec) * f.occ_stride + (o.x / f.prec)];   // src/codec.rs:288-301, 393
  if (occ == 0) return o;
  const uint32_t d0 = (uint32_t)(f.geo[0][o.y * f.geo_stride[0] + o.x] >> 2);  // depth / 4, codec.rs:534
  o.p0 = make_point(p, u, v, d0);
  o.n = 1;
  if (f.map_count > 1) {
    const uint32_t d1 = (uint32_t)(f.geo[1][o.y * f.geo_stride[1] + o.x] >> 2);
    o.p1 = make_point1(f, p, u, v, o.p0, d1);
    if (!same_point(o.p0, o.p1)) o.n = 2;                                      // codec.rs:422-427
  }
  return o;
}

// ------------------------------------------------------------ k_block_owner
// One wave per virtual block.  non_zero_pixel > 0  <=>  any occupancy sample under the block's
// R*R mapped pixels is non-zero; ascending-patch overwrite == max over the patches that write.
__global__ __launch_bounds__(256) void k_block_owner(const DevFrame* __restrict__ frames, uint32_t first) {
  const DevFrame& f = frames[first + blockIdx.y];
  const uint32_t vb = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (vb >= f.n_vblocks) return;
  const VBlock b = f.vblocks[vb];
  const DevPatch& p = f.patches[b.patch];
  const uint32_t R = f.R, RR = R * R;
  bool any = false;
  for (uint32_t i = lane_id(); i < RR; i += 64) {
    const uint32_t u = b.u0 * R + (i % R), v = b.v0 * R + (i / R);
    const int32_t x = p.ax_u * (int32_t)u + p.ax_v * (int32_t)v + p.cx;
    const int32_t y = p.ay_u * (int32_t)u + p.ay_v * (int32_t)v + p.cy;
    any |= f.occ[((uint32_t)y / f.prec) * f.occ_stride + ((uint32_t)x / f.prec)] != 0;
  }
  if (__ballot(any) != 0ull && lane_id() == 0) atomicMax(&f.block_to_patch[b.canvas_block], (uint32_t)b.patch + 1u);
}

// ------------------------------------------------------------------ k_count
__global__ __launch_bounds__(256) void k_count(const DevFrame* __restrict__ frames, uint32_t first) {
  const DevFrame& f = frames[first + blockIdx.y];
  const uint32_t vb = blockIdx.x;
  if (vb >= f.n_vblocks) return;
  const VBlock b = f.vblocks[vb];
  __shared__ uint32_t wave_sum[4];
  uint32_t total = 0;
  if (f.block_to_patch[b.canvas_block] == (uint32_t)b.patch + 1u) {           // codec.rs:379
    const DevPatch p = f.patches[b.patch];
    const uint32_t R = f.R, RR = R * R;
    uint32_t mine = 0;
    for (uint32_t i = threadIdx.x; i < RR; i += 256) {
      const PixelOut o = eval_pixel(f, p, b.u0 * R + (i % R), b.v0 * R + (i / R));
      mine += o.n;
    }
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off, 64);
    if (lane_id() == 0) wave_sum[threadIdx.x >> 6] = mine;
    __syncthreads();
    total = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
  }
  if (threadIdx.x == 0) f.vb_count[vb] = total;
}

// ------------------------------------------------------------------- k_scan
// One workgroup per frame: exclusive prefix of vb_count (<= a few 10^4 entries).
__global__ __launch_bounds__(1024) void k_scan(const DevFrame* __restrict__ frames, uint32_t first) {
  const DevFrame& f = frames[first + blockIdx.x];
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  const uint32_t n = f.n_vblocks;
  for (uint32_t base = 0; base < n; base += 1024) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t c = i < n ? f.vb_count[i] : 0u;
    uint32_t incl = c;
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(incl, off, 64);
      if ((int)lane_id() >= off) incl += t;
    }
    if (lane_id() == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t wbase = 0;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) wbase += wsum[w];
    const uint32_t carry = carry_s;
    if (i < n) f.vb_offset[i] = carry + wbase + incl - c;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = carry + wbase + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *f.n_points = carry_s;
}

// ------------------------------------------------------------------- k_emit
__global__ __launch_bounds__(256) void k_emit(const DevFrame* __restrict__ frames, uint32_t first) {
  const DevFrame& f = frames[first + blockIdx.y];
  const uint32_t vb = blockIdx.x;
  if (vb >= f.n_vblocks) return;
  if (f.vb_count[vb] == 0) return;
  const VBlock b = f.vblocks[vb];
  const DevPatch p = f.patches[b.patch];
  __shared__ uint32_t wave_sum[4];
  uint32_t base = f.vb_offset[vb];
  const uint32_t R = f.R, RR = R * R;
  const uint32_t wave = threadIdx.x >> 6;
  for (uint32_t i0 = 0; i0 < RR; i0 += 256) {                                  // raster chunks keep the order
    const uint32_t i = i0 + threadIdx.x;
    PixelOut o;
    o.n = 0;
    const uint32_t u = b.u0 * R + (i % R), v = b.v0 * R + (i / R);
    if (i < RR) o = eval_pixel(f, p, u, v);
    const uint64_t m1 = __ballot(o.n >= 1), m2 = __ballot(o.n == 2);
    const uint32_t before = mbcnt(m1) + mbcnt(m2);
    if (lane_id() == 0) wave_sum[wave] = (uint32_t)__popcll(m1) + (uint32_t)__popcll(m2);
    __syncthreads();
    uint32_t wbase = 0, tot = 0;
#pragma unroll
    for (uint32_t w = 0; w < 4; ++w) {
      const uint32_t s = wave_sum[w];
      if (w < wave) wbase += s;
      tot += s;
    }
    uint32_t k = base + wbase + before;
    for (uint32_t j = 0; j < o.n; ++j, ++k) {
      if (k >= f.capacity) break;                                               // never write past the caller's arrays
      const Pt& pt = j == 0 ? o.p0 : o.p1;
      vpcc_point3 q;
      q.x = pt.c[0]; q.y = pt.c[1]; q.z = pt.c[2];
      f.out_xyz[k] = q;
      if (f.out_patch) f.out_patch[k] = b.patch;                                // partition, codec.rs:452
      if (f.has_attr) {                                                         // color_point_cloud, codec.rs:626-644
        const uint32_t cidx = (o.y >> 1) * f.attr_cstride[j] + (o.x >> 1);     // chroma nearest neighbour
        const uint16_t Y = f.attr_y[j][o.y * f.attr_stride[j] + o.x];
        const uint16_t U = f.attr_u[j][cidx];
        const uint16_t V = f.attr_v[j][cidx];
        f.out_rgb[k] = yuv10_to_rgb8(Y, U, V);
      }
    }
    base += tot;
    __syncthreads();
  }
}

// ---------------------------------------------------- k_upsample_occupancy
// tile.occupancy_map, src/codec.rs:288-301 (kept for API completeness; the main path never builds it).
__global__ __launch_bounds__(256) void k_upsample_occupancy(const DevFrame* __restrict__ frames, uint32_t frame,
                                                            uint8_t* __restrict__ out) {
  const DevFrame& f = frames[frame];
  const uint32_t x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x < f.width) out[(size_t)y * f.width + x] = f.occ[(y / f.prec) * f.occ_stride + (x / f.prec)];
}

// ----------------------------------------------------------------- launchers
void launch_block_owner(const DevFrame* d_frames, uint32_t first, uint32_t count, uint32_t max_vb, void* stream) {
  if (!count || !max_vb) return;
  hipLaunchKernelGGL(k_block_owner, dim3((max_vb + 3) / 4, count), dim3(256), 0, (hipStream_t)stream, d_frames, first);
}
void launch_count(const DevFrame* d_frames, uint32_t first, uint32_t count, uint32_t max_vb, void* stream) {
  if (!count || !max_vb) return;
  hipLaunchKernelGGL(k_count, dim3(max_vb, count), dim3(256), 0, (hipStream_t)stream, d_frames, first);
}
void launch_scan(const DevFrame* d_frames, uint32_t first, uint32_t count, void* stream) {
  if (!count) return;
  hipLaunchKernelGGL(k_scan, dim3(count), dim3(1024), 0, (hipStream_t)stream, d_frames, first);
}
void launch_emit(const DevFrame* d_frames, uint32_t first, uint32_t count, uint32_t max_vb, void* stream) {
  if (!count || !max_vb) return;
  hipLaunchKernelGGL(k_emit, dim3(max_vb, count), dim3(256), 0, (hipStream_t)stream, d_frames, first);
}
void launch_upsample_occupancy(const DevFrame* d_frames, uint32_t frame, uint8_t* d_out, uint32_t width,
                               uint32_t height, void* stream) {
  if (!width || !height) return;
  hipLaunchKernelGGL(k_upsample_occupancy, dim3((width + 255) / 256, height), dim3(256), 0, (hipStream_t)stream,
                     d_frames, frame, d_out);
}

}  // namespace vpcc
