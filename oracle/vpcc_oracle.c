/*
 * vpcc_oracle.c — CPU ORACLE.  TEST INFRASTRUCTURE ONLY (see vpcc_oracle.h).
 *
 * Single-threaded plain-C restatement of the reconstruction hot path of
 * benclmnt/tmc2-rs.  PARITY UNPINNED by the reference's own tests (it has none
 * for this path); pinned by tests/test_oracle_kat.py.
 *
 * Every function cites the reference lines it follows.  Rust semantics kept:
 *   - `usize` arithmetic of the release profile wraps (README.md:14 builds
 *     --release); asserts stay active in release and become status codes here;
 *   - `as u16` truncates;
 *   - f64 colour maths is IEEE, evaluated in source order, never contracted
 *     (this file is compiled with -ffp-contract=off).
 */
#define _POSIX_C_SOURCE 199309L
#include "vpcc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ---- Image<T>::get_helper / get, src/decoder.rs:971-1021 ------------------
 * luma: v*width+u ; chroma (Yuv420): (v/2)*(width/2)+(u/2); bounds assert on
 * (u < width && v < height) of the LUMA size for every channel.  The stride
 * fields stand in for `width` / `width/2` (the reference ignores linesize). */
static int occ_get(const vpcc_image_u8* im, uint64_t u, uint64_t v, uint8_t* out) {
  if (!(u < im->width && v < im->height)) return VPCC_ERR_PATCH_OUT_OF_CANVAS;
  *out = im->y[v * (uint64_t)im->stride + u];
  return VPCC_OK;
}

static int img16_get(const vpcc_image_u16* im, int channel, uint64_t u, uint64_t v, uint16_t* out) {
  if (!(u < im->width && v < im->height)) return VPCC_ERR_PATCH_OUT_OF_CANVAS;
  if (channel == 0) {
    *out = im->y[v * (uint64_t)im->stride + u];
  } else {
    const uint16_t* pl = channel == 1 ? im->u : im->v;
    *out = pl[(v / 2) * (uint64_t)im->cstride + (u / 2)];
  }
  return VPCC_OK;
}

/* ---- Patch::patch_to_canvas_helper, src/decoder.rs:853-867 ---------------- */
void vpcc_oracle_patch_to_canvas(const vpcc_patch* p, uint64_t u, uint64_t v, uint64_t resolution,
                                 uint64_t* x, uint64_t* y) {
  const uint64_t u0 = (uint64_t)p->u0 * resolution, v0 = (uint64_t)p->v0 * resolution;
  const uint64_t size_u0 = p->size_u0, size_v0 = p->size_v0; /* in BLOCKS at every resolution: reference quirk */
  switch (p->orientation) {
    case VPCC_ORIENT_DEFAULT: *x = u + u0;               *y = v + v0;               break;
    case VPCC_ORIENT_ROT90:   *x = size_v0 - 1 - v + u0; *y = u + v0;               break;
    case VPCC_ORIENT_ROT180:  *x = size_u0 - 1 - u + u0; *y = size_v0 - 1 - v + v0; break;
    case VPCC_ORIENT_ROT270:  *x = v + u0;               *y = size_u0 - 1 - u + v0; break;
    case VPCC_ORIENT_MIRROR:  *x = size_u0 - 1 - u + u0; *y = v + v0;               break;
    case VPCC_ORIENT_MROT90:  *x = size_v0 - 1 - v + u0; *y = size_u0 - 1 - u + v0; break;
    case VPCC_ORIENT_MROT180: *x = u + u0;               *y = size_v0 - 1 - v + v0; break;
    case VPCC_ORIENT_MROT270: *x = v + u0;               *y = u + v0;               break;
    case VPCC_ORIENT_SWAP:    *x = v + u0;               *y = u + v0;               break;
    default:                  *x = UINT64_MAX;           *y = UINT64_MAX;           break;
  }
}

/* ---- Patch::generate_normal_coordinate / generate_point, decoder.rs:871-888 */
vpcc_point3 vpcc_oracle_generate_point(const vpcc_patch* p, uint64_t u, uint64_t v, uint16_t depth) {
  uint16_t pt[3] = {0, 0, 0};
  const uint64_t d = depth;
  uint64_t n;
  if (p->projection_mode == 0) n = d + (uint64_t)p->d1;
  else n = ((uint64_t)p->d1 > d ? (uint64_t)p->d1 : d) - d;
  pt[p->normal_axis]    = (uint16_t)n;
  pt[p->tangent_axis]   = (uint16_t)(u * (uint64_t)p->lod_x + (uint64_t)p->u1);
  pt[p->bitangent_axis] = (uint16_t)(v * (uint64_t)p->lod_y + (uint64_t)p->v1);
  vpcc_point3 r = {pt[0], pt[1], pt[2]};
  return r;
}

/* ---- convert_yuv10_to_rgb8, src/codec.rs:661-687 -------------------------- */
static uint8_t clamp_u8(double x) {
  if (x < 0.) return 0;
  if (x > 255.) return 255;
  return (uint8_t)x;
}

vpcc_color3 vpcc_oracle_yuv10_to_rgb8(uint16_t y16, uint16_t u16, uint16_t v16) {
  const double offset = 512., scale = 1023.;
  const double y = (double)y16, u = (double)u16, v = (double)v16;
  const double r = y + 1.57480 * (v - offset);
  const double g = y - 0.18733 * (u - offset) - (0.46813 * (v - offset));
  const double b = y + 1.85563 * (u - offset);
  vpcc_color3 c;
  c.r = clamp_u8(floor(r / scale * 255.));
  c.g = clamp_u8(floor(g / scale * 255.));
  c.b = clamp_u8(floor(b / scale * 255.));
  return c;
}

/* ---- envelope checks: the unimplemented!()/assert sites the frame can hit -- */
static int check_envelope(const vpcc_frame_desc* f) {
  if (!f) return VPCC_ERR_INVALID_ARG;
  if (f->width == 0 || f->height == 0 || f->occupancy_resolution == 0 || f->occupancy_precision == 0)
    return VPCC_ERR_INVALID_ARG;
  if (f->map_count < 1 || f->map_count > 2) return VPCC_ERR_UNSUPPORTED;
  if (f->attribute_count > 1) return VPCC_ERR_UNSUPPORTED;         /* src/decoder.rs:133 */
  if (f->flags & VPCC_FRAME_RGB444) return VPCC_ERR_UNSUPPORTED;
  if (f->patch_count && !f->patches) return VPCC_ERR_INVALID_ARG;
  if (!f->occupancy.y) return VPCC_ERR_INVALID_ARG;
  for (uint32_t i = 0; i < f->patch_count; ++i) {
    const vpcc_patch* p = &f->patches[i];
    if (p->axis_of_additional_plane != 0) return VPCC_ERR_UNSUPPORTED; /* src/codec.rs:437 */
    if (p->normal_axis > 2 || p->tangent_axis > 2 || p->bitangent_axis > 2) return VPCC_ERR_INVALID_ARG;
    if (p->projection_mode > 1) return VPCC_ERR_INVALID_ARG;        /* unreachable!() decoder.rs:886 */
    if (p->orientation > VPCC_ORIENT_MROT270) return VPCC_ERR_INVALID_ARG;
  }
  return VPCC_OK;
}

/* ---- generate_block_to_patch_from_occupancy_map_video, codec.rs:205-250 --- */
int vpcc_oracle_block_to_patch(const vpcc_frame_desc* f, uint64_t* block_to_patch) {
  int st = check_envelope(f);
  if (st) return st;
  const uint64_t R = f->occupancy_resolution, prec = f->occupancy_precision;
  const uint64_t bw = f->width / R, bh = f->height / R;
  memset(block_to_patch, 0, sizeof(uint64_t) * bw * bh);
  for (uint64_t patch_index = 0; patch_index < f->patch_count; ++patch_index) {
    const vpcc_patch* patch = &f->patches[patch_index];
    for (uint64_t v0 = 0; v0 < patch->size_v0; ++v0) {
      for (uint64_t u0 = 0; u0 < patch->size_u0; ++u0) {
        uint64_t bx, by;
        vpcc_oracle_patch_to_canvas(patch, u0, v0, 1, &bx, &by);      /* patch_block_to_canvas_block */
        if (!(bx < bw && by < bh)) return VPCC_ERR_PATCH_OUT_OF_CANVAS;  /* assert decoder.rs:835 */
        const uint64_t block_index = by * bw + bx;
        uint64_t non_zero_pixel = 0;
        for (uint64_t v1 = 0; v1 < R; ++v1) {
          const uint64_t v = v0 * R + v1;
          for (uint64_t u1 = 0; u1 < R; ++u1) {
            const uint64_t u = u0 * R + u1;
            uint64_t x, y;
            vpcc_oracle_patch_to_canvas(patch, u, v, R, &x, &y);
            if (!(x < f->width && y < f->height)) return VPCC_ERR_PATCH_OUT_OF_CANVAS; /* decoder.rs:848 */
            uint8_t o;
            st = occ_get(&f->occupancy, x / prec, y / prec, &o);       /* left_top_in_frame == (0,0) */
            if (st) return st;
            non_zero_pixel += o;
          }
        }
        if (non_zero_pixel > 0) block_to_patch[block_index] = patch_index + 1;
      }
    }
  }
  return VPCC_OK;
}

void vpcc_oracle_frame_free(vpcc_oracle_frame* fr) {
  if (!fr) return;
  free(fr->occupancy_map); free(fr->block_to_patch); free(fr->positions); free(fr->colors16);
  free(fr->colors); free(fr->partition); free(fr->point_to_pixel);
  memset(fr, 0, sizeof(*fr));
}

static int grow_one(void** p, size_t bytes) {
  void* a = realloc(*p, bytes);
  if (!a) return -1;
  *p = a;
  return 0;
}

static int grow(vpcc_oracle_frame* fr) {
  const size_t nc = fr->cap_points ? fr->cap_points * 2 : 4096;
  if (grow_one((void**)&fr->positions, nc * sizeof(vpcc_point3))) return -1;
  if (grow_one((void**)&fr->colors16, nc * 3 * sizeof(uint16_t))) return -1;
  if (grow_one((void**)&fr->colors, nc * sizeof(vpcc_color3))) return -1;
  if (grow_one((void**)&fr->partition, nc * sizeof(uint64_t))) return -1;
  if (grow_one((void**)&fr->point_to_pixel, nc * sizeof(vpcc_oracle_p2p))) return -1;
  fr->cap_points = nc;
  return 0;
}

/* ---- generate_points, src/codec.rs:517-565 --------------------------------
 * n_created = 2 normally; 1 when map_count == 1 (the reference then unwraps a
 * None and panics, src/codec.rs:432 — here D0 only is emitted, a documented
 * extension outside the parity envelope). */
static int generate_points(const vpcc_frame_desc* f, const vpcc_patch* patch, uint64_t u, uint64_t v,
                           uint64_t x, uint64_t y, vpcc_point3 pts[2], int* n_created) {
  uint16_t s0;
  int st = img16_get(&f->geometry[0], 0, x, y, &s0);
  if (st) return st;
  pts[0] = vpcc_oracle_generate_point(patch, u, v, (uint16_t)(s0 / 4));   /* depth /4: codec.rs:532-534 */
  *n_created = 1;
  if (f->map_count > 1) {
    uint16_t s1;
    st = img16_get(&f->geometry[1], 0, x, y, &s1);
    if (st) return st;
    const uint16_t d1 = (uint16_t)(s1 / 4);
    if (f->absolute_d1) {
      pts[1] = vpcc_oracle_generate_point(patch, u, v, d1);
    } else {
      uint16_t c[3] = {pts[0].x, pts[0].y, pts[0].z};
      if (patch->projection_mode == 0) c[patch->normal_axis] = (uint16_t)(c[patch->normal_axis] + d1);
      else c[patch->normal_axis] = (uint16_t)(c[patch->normal_axis] - d1);
      pts[1].x = c[0]; pts[1].y = c[1]; pts[1].z = c[2];
    }
    *n_created = 2;
  }
  return VPCC_OK;
}

/* ---- per-frame loop body, src/decoder.rs:249-305 --------------------------- */
int vpcc_oracle_reconstruct_frame(const vpcc_frame_desc* f, vpcc_oracle_frame* out) {
  memset(out, 0, sizeof(*out));
  int st = check_envelope(f);
  if (st) return st;
  const uint64_t R = f->occupancy_resolution, prec = f->occupancy_precision;
  const uint64_t W = f->width, H = f->height;
  const uint64_t bw = W / R, bh = H / R;
  out->n_blocks = bw * bh;
  out->n_pixels = W * H;

  /* decoder.rs:249-255 */
  out->block_to_patch = (uint64_t*)calloc(out->n_blocks ? out->n_blocks : 1, sizeof(uint64_t));
  if (!out->block_to_patch) return VPCC_ERR_INVALID_ARG;
  st = vpcc_oracle_block_to_patch(f, out->block_to_patch);
  if (st) { vpcc_oracle_frame_free(out); return st; }

  /* generate_point_cloud, codec.rs:256-514 */
  /* occupancy upsample, codec.rs:288-301 */
  out->occupancy_map = (uint8_t*)calloc(out->n_pixels, 1);
  if (!out->occupancy_map) { vpcc_oracle_frame_free(out); return VPCC_ERR_INVALID_ARG; }
  for (uint64_t v = 0; v < H; ++v)
    for (uint64_t u = 0; u < W; ++u) {
      uint8_t o;
      st = occ_get(&f->occupancy, u / prec, v / prec, &o);
      if (st) { vpcc_oracle_frame_free(out); return st; }
      out->occupancy_map[v * W + u] = o;
    }

  /* codec.rs:317-321: geometry video must hold frames f*map_count .. +map_count */
  if (!f->geometry[0].y || (f->map_count > 1 && !f->geometry[1].y)) {
    vpcc_oracle_frame_free(out);
    return VPCC_ERR_SHORT_VIDEO;
  }

  /* enumeration, codec.rs:352-480 */
  for (uint64_t patch_index = 0; patch_index < f->patch_count; ++patch_index) {
    const vpcc_patch* patch = &f->patches[patch_index];
    for (uint64_t v0 = 0; v0 < patch->size_v0; ++v0) {
      for (uint64_t u0 = 0; u0 < patch->size_u0; ++u0) {
        uint64_t bx, by;
        vpcc_oracle_patch_to_canvas(patch, u0, v0, 1, &bx, &by);
        if (!(bx < bw && by < bh)) { vpcc_oracle_frame_free(out); return VPCC_ERR_PATCH_OUT_OF_CANVAS; }
        if (out->block_to_patch[by * bw + bx] != patch_index + 1) continue;
        for (uint64_t v1 = 0; v1 < R; ++v1) {
          const uint64_t v = v0 * R + v1;
          for (uint64_t u1 = 0; u1 < R; ++u1) {
            const uint64_t u = u0 * R + u1;
            uint64_t x, y;
            vpcc_oracle_patch_to_canvas(patch, u, v, R, &x, &y);
            if (!(x < W && y < H)) { vpcc_oracle_frame_free(out); return VPCC_ERR_PATCH_OUT_OF_CANVAS; }
            if (out->occupancy_map[y * W + x] == 0) continue;
            vpcc_point3 created[2];
            int n_created = 0;
            st = generate_points(f, patch, u, v, x, y, created, &n_created);
            if (st) { vpcc_oracle_frame_free(out); return st; }
            for (int i = 0; i < n_created; ++i) {
              /* duplicate removal is unconditional: codec.rs:422-427 */
              if (i != 0 && created[i].x == created[0].x && created[i].y == created[0].y &&
                  created[i].z == created[0].z)
                continue;
              if (out->n_points == out->cap_points && grow(out)) {
                vpcc_oracle_frame_free(out);
                return VPCC_ERR_INVALID_ARG;
              }
              const size_t k = out->n_points++;
              out->positions[k] = created[i];                      /* add_point, codec.rs:45-53 */
              out->colors[k].r = out->colors[k].g = out->colors[k].b = 127;
              out->colors16[3 * k] = out->colors16[3 * k + 1] = out->colors16[3 * k + 2] = 0;
              out->partition[k] = patch_index;                     /* codec.rs:452 */
              out->point_to_pixel[k].x = (uint32_t)x;              /* codec.rs:463-472 */
              out->point_to_pixel[k].y = (uint32_t)y;
              out->point_to_pixel[k].z = (uint32_t)i;
            }
          }
        }
      }
    }
  }

  /* color_point_cloud, codec.rs:569-658 */
  if (f->attribute_count > 0 && out->n_points > 0) {
    /* video.get(0).unwrap(), video.get(1).unwrap(): codec.rs:589-590 */
    if (!f->attribute[0].y || !f->attribute[0].u || !f->attribute[0].v ||
        (f->map_count > 1 && (!f->attribute[1].y || !f->attribute[1].u || !f->attribute[1].v))) {
      vpcc_oracle_frame_free(out);
      return VPCC_ERR_SHORT_VIDEO;
    }
    for (size_t i = 0; i < out->n_points; ++i) {
      const vpcc_oracle_p2p loc = out->point_to_pixel[i];
      const vpcc_image_u16* frame = &f->attribute[loc.z];          /* z + frame_index*map_count */
      for (int c = 0; c < 3; ++c) {
        st = img16_get(frame, c, loc.x, loc.y, &out->colors16[3 * i + c]);
        if (st) { vpcc_oracle_frame_free(out); return st; }
      }
    }
    /* convert_yuv16_to_rgb8, codec.rs:88-94 (decoder.rs:301-305) */
    for (size_t i = 0; i < out->n_points; ++i)
      out->colors[i] = vpcc_oracle_yuv10_to_rgb8(out->colors16[3 * i], out->colors16[3 * i + 1],
                                                 out->colors16[3 * i + 2]);
  }
  return VPCC_OK;
}

double vpcc_oracle_time_frames(const vpcc_frame_desc* frames, uint32_t n, uint32_t reps,
                               uint64_t* points_out, int* status_out) {
  double best = 1e300;
  uint64_t pts = 0;
  int status = VPCC_OK;
  for (uint32_t r = 0; r < reps && status == VPCC_OK; ++r) {
    struct timespec t0, t1;
    pts = 0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (uint32_t i = 0; i < n; ++i) {
      vpcc_oracle_frame fr;
      status = vpcc_oracle_reconstruct_frame(&frames[i], &fr);
      if (status) break;
      pts += fr.n_points;
      vpcc_oracle_frame_free(&fr);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const double s = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    if (s < best) best = s;
  }
  if (points_out) *points_out = pts;
  if (status_out) *status_out = status;
  return best;
}
