/*
 * vpcc_oracle.h — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A single-threaded plain-C restatement of the reference's (benclmnt/tmc2-rs)
 * reconstruction hot path, following the reference's loop structure and its
 * intermediate arrays literally.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this; the product path
 * (tmc2-rs_amd/) never links, imports or calls it.
 *
 * PARITY UNPINNED BY THE REFERENCE: the reference ships no golden vectors,
 * fixtures or tests for this path (its only tests cover the bit reader,
 * src/bitstream.rs:345-438) and it cannot be built here (Rust + libavcodec,
 * neither present).  The oracle is pinned instead by the hand-derived
 * known-answer tests in tests/test_oracle_kat.py, each computed from the
 * reference formulas cited next to the function it exercises.
 *
 * Descriptors are the public PODs of include/vpcc_recon.h.
 */
#ifndef VPCC_ORACLE_H
#define VPCC_ORACLE_H

#include "../include/vpcc_recon.h"

#ifdef __cplusplus
extern "C" {
#endif

/* point_to_pixel entry: (x, y, layer) — src/codec.rs:463-472 */
typedef struct vpcc_oracle_p2p { uint32_t x, y, z; } vpcc_oracle_p2p;

/* Result of one frame, with every intermediate the reference materialises. */
typedef struct vpcc_oracle_frame {
  uint8_t*   occupancy_map;     /* W*H          tile.occupancy_map, src/codec.rs:288-301 */
  uint64_t*  block_to_patch;    /* (W/R)*(H/R)  src/codec.rs:205-250 (usize)             */
  vpcc_point3* positions;       /* N            PointSet3.positions                      */
  uint16_t*  colors16;          /* 3*N          PointSet3.colors16bit (Y,U,V)            */
  vpcc_color3* colors;          /* N            PointSet3.colors                         */
  uint64_t*  partition;         /* N            patch index per point, src/codec.rs:452  */
  vpcc_oracle_p2p* point_to_pixel; /* N                                                  */
  size_t n_points;
  size_t cap_points;
  size_t n_blocks;
  size_t n_pixels;
} vpcc_oracle_frame;

/* Patch::patch_to_canvas_helper, src/decoder.rs:853-867, in release-profile
 * (wrapping) arithmetic.  Returns the 64-bit wrapped coordinates. */
void vpcc_oracle_patch_to_canvas(const vpcc_patch* p, uint64_t u, uint64_t v, uint64_t resolution,
                                 uint64_t* x, uint64_t* y);

/* Patch::generate_point, src/decoder.rs:871-888. */
vpcc_point3 vpcc_oracle_generate_point(const vpcc_patch* p, uint64_t u, uint64_t v, uint16_t depth);

/* convert_yuv10_to_rgb8, src/codec.rs:661-687. */
vpcc_color3 vpcc_oracle_yuv10_to_rgb8(uint16_t y, uint16_t u, uint16_t v);

/* generate_block_to_patch_from_occupancy_map_video, src/codec.rs:205-250.
 * out has (W/R)*(H/R) entries.  Returns a vpcc_status (non-zero where the
 * reference would panic). */
int vpcc_oracle_block_to_patch(const vpcc_frame_desc* f, uint64_t* out);

/* The body of the per-frame loop, src/decoder.rs:249-305: block_to_patch,
 * generate_point_cloud (occupancy upsample, enumeration, generate_points,
 * color_point_cloud), convert_yuv16_to_rgb8.  Allocates the result;
 * free it with vpcc_oracle_frame_free. */
int vpcc_oracle_reconstruct_frame(const vpcc_frame_desc* f, vpcc_oracle_frame* out);
void vpcc_oracle_frame_free(vpcc_oracle_frame* fr);

/* Timing helper for bench.py's cpu_baseline leg: runs
 * vpcc_oracle_reconstruct_frame over n frames `reps` times, single-threaded
 * like the reference (README.md:7), and returns the seconds of the fastest
 * repetition; *points_out = points of one repetition. */
double vpcc_oracle_time_frames(const vpcc_frame_desc* frames, uint32_t n, uint32_t reps,
                               uint64_t* points_out, int* status_out);

#ifdef __cplusplus
}
#endif
#endif
