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
#include "vpcc_devfn.hpp"

namespace vpcc {

// ------------------------------------------------------------ k_block_owner
// One wave per virtual block.  non_zero_pixel > 0  <=>  any occupancy sample under the block's
// R*R mapped pixels is non-zero; ascending-patch overwrite == max over the patches that write.
__global__ __launch_bounds__(256) void k_block_owner(const DevFrame* __restrict__ frames, uint32_t first) {
  const DevFrame& f = frames[first + blockIdx.y];
  const uint32_t vb = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (vb >= f.n_vblocks) return;
  const VBlock b = gload(f.vblocks + vb);
  const DevPatch p = gload(f.patches + b.patch);
  const uint32_t R = f.R, RR = R * R;
  bool any = false;
  for (uint32_t i = lane_id(); i < RR; i += 64) {
    const uint32_t u = b.u0 * R + (i % R), v = b.v0 * R + (i / R);
    const int32_t x = p.ax_u * (int32_t)u + p.ax_v * (int32_t)v + p.cx;
    const int32_t y = p.ay_u * (int32_t)u + p.ay_v * (int32_t)v + p.cy;
    any |= gl(f.occ)[((uint32_t)y / f.prec) * f.occ_stride + ((uint32_t)x / f.prec)] != 0;
  }
  if (__ballot(any) != 0ull && lane_id() == 0) atomicMax(f.block_to_patch + b.canvas_block, (uint32_t)b.patch + 1u);
}

// ------------------------------------------------------------------ k_count
__global__ __launch_bounds__(256) void k_count(const DevFrame* __restrict__ frames, uint32_t first) {
  const DevFrame& f = frames[first + blockIdx.y];
  const uint32_t vb = blockIdx.x;
  if (vb >= f.n_vblocks) return;
  const VBlock b = gload(f.vblocks + vb);
  __shared__ uint32_t wave_sum[4];
  uint32_t total = 0;
  if (gl(f.block_to_patch)[b.canvas_block] == (uint32_t)b.patch + 1u) {       // codec.rs:379
    const DevPatch p = gload(f.patches + b.patch);
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
  if (threadIdx.x == 0) glw(f.vb_count)[vb] = total;
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
    const uint32_t c = i < n ? gl(f.vb_count)[i] : 0u;
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
    if (i < n) glw(f.vb_offset)[i] = carry + wbase + incl - c;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = carry + wbase + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *glw(f.n_points) = carry_s;
}

// ------------------------------------------------------------------- k_emit
__global__ __launch_bounds__(256) void k_emit(const DevFrame* __restrict__ frames, uint32_t first) {
  const DevFrame& f = frames[first + blockIdx.y];
  const uint32_t vb = blockIdx.x;
  if (vb >= f.n_vblocks) return;
  if (gl(f.vb_count)[vb] == 0) return;
  const VBlock b = gload(f.vblocks + vb);
  const DevPatch p = gload(f.patches + b.patch);
  __shared__ uint32_t wave_sum[4];
  uint32_t base = gl(f.vb_offset)[vb];
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
      {   // non-temporal: the output stream must not evict the geometry k_count just pulled into the L2
        VPCC_GLOBAL uint16_t* o = (VPCC_GLOBAL uint16_t*)(f.out_xyz + k);
        __builtin_nontemporal_store(q.x, o);
        __builtin_nontemporal_store(q.y, o + 1);
        __builtin_nontemporal_store(q.z, o + 2);
      }
      if (f.out_patch) glw(f.out_patch)[k] = b.patch;                           // partition, codec.rs:452
      if (f.has_attr) {                                                         // color_point_cloud, codec.rs:626-644
        const uint32_t cidx = (o.y >> 1) * f.attr_cstride[j] + (o.x >> 1);     // chroma nearest neighbour
        const uint16_t Y = gl(f.attr_y[j])[o.y * f.attr_stride[j] + o.x];
        const uint16_t U = gl(f.attr_u[j])[cidx];
        const uint16_t V = gl(f.attr_v[j])[cidx];
        const vpcc_color3 c = yuv10_to_rgb8_fast(Y, U, V);
        VPCC_GLOBAL uint8_t* oc = (VPCC_GLOBAL uint8_t*)(f.out_rgb + k);
        __builtin_nontemporal_store(c.r, oc);
        __builtin_nontemporal_store(c.g, oc + 1);
        __builtin_nontemporal_store(c.b, oc + 2);
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
  if (x < f.width) out[(size_t)y * f.width + x] = gl(f.occ)[(y / f.prec) * f.occ_stride + (x / f.prec)];
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
// Raster planes -> block-by-block planes (DevFrame::tiled), a wave per 16x16 block: lane l moves the block's pixels
// 4 (l & 3) .. + 3 of row l >> 2 of every luma plane (8 bytes) to bytes 8 l .. of the block's 512; lanes 0-15 move the
// 8x8 chroma block the same way (row l >> 1, samples 4 (l & 1) .. + 3).  Every block of the canvas, once per gof.
__global__ __launch_bounds__(256) void k_tile_planes(const DevFrame* __restrict__ raster, const DevFrame* __restrict__ tiled,
                                                     uint32_t first) {
  const DevFrame& r = raster[first + blockIdx.y];
  const DevFrame& t = tiled[first + blockIdx.y];
  const uint32_t blk = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
  if (blk >= r.bw * r.bh) return;
  const uint32_t x0 = (blk % r.bw) * 16u, y0 = (blk / r.bw) * 16u;
  typedef uint32_t v2 __attribute__((ext_vector_type(2)));
  const uint32_t px = x0 + 4u * (lane & 3u), py = y0 + (lane >> 2);
  for (uint32_t m = 0; m < r.map_count; ++m) {
    *(VPCC_GLOBAL v2*)((VPCC_GLOBAL unsigned char*)t.geo[m] + blk * 512u + 8u * lane) =
        *(const VPCC_GLOBAL v2*)((const VPCC_GLOBAL unsigned char*)r.geo[m] + ((size_t)py * r.geo_stride[m] + px) * 2u);
    if (r.has_attr) {
      *(VPCC_GLOBAL v2*)((VPCC_GLOBAL unsigned char*)t.attr_y[m] + blk * 512u + 8u * lane) =
          *(const VPCC_GLOBAL v2*)((const VPCC_GLOBAL unsigned char*)r.attr_y[m] + ((size_t)py * r.attr_stride[m] + px) * 2u);
      if (lane < 16u) {
        const size_t c = ((size_t)(y0 / 2u + (lane >> 1)) * r.attr_cstride[m] + x0 / 2u + 4u * (lane & 1u)) * 2u;
        *(VPCC_GLOBAL v2*)((VPCC_GLOBAL unsigned char*)t.attr_u[m] + blk * 128u + 8u * lane) =
            *(const VPCC_GLOBAL v2*)((const VPCC_GLOBAL unsigned char*)r.attr_u[m] + c);
        *(VPCC_GLOBAL v2*)((VPCC_GLOBAL unsigned char*)t.attr_v[m] + blk * 128u + 8u * lane) =
            *(const VPCC_GLOBAL v2*)((const VPCC_GLOBAL unsigned char*)r.attr_v[m] + c);
      }
    }
  }
}
void launch_tile_planes(const DevFrame* raster, const DevFrame* tiled, uint32_t first, uint32_t count, uint32_t max_blocks,
                        void* stream) {
  if (!count || !max_blocks) return;
  hipLaunchKernelGGL(k_tile_planes, dim3((max_blocks + 3) / 4, count), dim3(256), 0, (hipStream_t)stream, raster, tiled, first);
}

void launch_upsample_occupancy(const DevFrame* d_frames, uint32_t frame, uint8_t* d_out, uint32_t width,
                               uint32_t height, void* stream) {
  if (!width || !height) return;
  hipLaunchKernelGGL(k_upsample_occupancy, dim3((width + 255) / 256, height), dim3(256), 0, (hipStream_t)stream,
                     d_frames, frame, d_out);
}

}  // namespace vpcc
