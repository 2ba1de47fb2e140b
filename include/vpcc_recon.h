/*
 * vpcc_recon.h — C ABI of the MI355X-native V-PCC point-cloud reconstruction path.
 *
 * This is the drop-in boundary for the per-frame reconstruction hot path of the
 * Rust crate benclmnt/tmc2-rs (reference).  The reference has no FFI layer; the
 * entry points below are what a `extern "C"` binding placed inside the
 * reference's worker thread (src/decoder.rs:188-314) would call instead of
 *
 *   codec::generate_block_to_patch_from_occupancy_map_video   src/codec.rs:205-250
 *   codec::generate_point_cloud (+ generate_points,           src/codec.rs:256-514, 517-565
 *          color_point_cloud)                                  src/codec.rs:569-658
 *   PointSet3::convert_yuv16_to_rgb8 / convert_yuv10_to_rgb8   src/codec.rs:88-94, 661-687
 *
 * The Rust side of the binding is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C, no C++/torch/HIP types in signatures; HIP streams travel as void*.
 *   - every function returns a vpcc_status (0 = OK).  The reference has no error
 *     channel on this path: it panics (assert!/unwrap/unimplemented!) and the
 *     consumer sees end-of-stream (src/lib.rs:113-145).  Each non-zero status
 *     names the reference panic it stands for; the Rust shim turns it back into
 *     a panic to keep that behaviour.
 *   - all arithmetic that the reference does in `usize` and truncates with
 *     `as u16` is reproduced modulo 2^16, so 32-bit descriptor fields give
 *     identical results.
 */
#ifndef VPCC_RECON_H
#define VPCC_RECON_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VPCC_ABI_VERSION 5   /* 5: vpcc_ctx_reserve_within, vpcc_ctx_pool_alloc / _free, vpcc_release_kept_pools; block_to_patch and the work lists are built by every launch.  4: vpcc_ctx_reserve / vpcc_ctx_pool_info replace VPCC_GOF_TUNE_PLACEMENT / vpcc_gof_placement; planes stay raster.  3,  2: vpcc_ctx_bind_thread, vpcc_decoder_stats, vpcc_gof_profile_interval / _kernel_time_means; portable page-locking */

/* ------------------------------------------------------------------ status */
typedef enum vpcc_status {
  VPCC_OK = 0,
  VPCC_ERR_INVALID_ARG = 1,      /* null pointer / zero size / inconsistent descriptor          */
  VPCC_ERR_UNSUPPORTED = 2,      /* reference: unimplemented!() (src/codec.rs:285-287,429-440…) */
  VPCC_ERR_PATCH_OUT_OF_CANVAS = 3, /* reference: assert!(x < canvas_stride && y < canvas_height)
                                      src/decoder.rs:835,848, or Image::get bounds assert :974    */
  VPCC_ERR_SHORT_VIDEO = 4,      /* reference: generate_point_cloud -> None -> unwrap panic
                                      (src/codec.rs:318-320, src/decoder.rs:271) or
                                      video.get(1).unwrap() (src/codec.rs:589-590)               */
  VPCC_ERR_CAPACITY = 5,         /* caller-provided output capacity too small (no reference analogue) */
  VPCC_ERR_DEVICE = 6,           /* HIP runtime error; see vpcc_last_error()                      */
  VPCC_ERR_NO_DEVICE = 7,        /* no gfx950 device / HIP extension unusable: the product path
                                      has NO CPU fallback and fails loudly                        */
  VPCC_ERR_STATE = 8             /* call order violated (e.g. results read before reconstruct)    */
} vpcc_status;

/* ------------------------------------------------------------- output types */
/* Layout-compatible with cgmath::Vector3<u16> / Vector3<u8> (#[repr(C)]),
 * i.e. with Vec<Point3D> / Vec<Color3B> storage (src/codec.rs:13-14, 23-24). */
typedef struct vpcc_point3 { uint16_t x, y, z; } vpcc_point3;
typedef struct vpcc_color3 { uint8_t r, g, b; } vpcc_color3;

/* ------------------------------------------------------------ patch record */
/* Exactly the `Patch` fields the hot path reads (src/decoder.rs:719-755), as
 * produced by create_patch_frame (src/decoder.rs:415-486). */
typedef enum vpcc_orientation {   /* PatchOrientation, src/decoder.rs:694-707 */
  VPCC_ORIENT_DEFAULT = 0, VPCC_ORIENT_SWAP = 1, VPCC_ORIENT_ROT90 = 2,
  VPCC_ORIENT_ROT180 = 3, VPCC_ORIENT_ROT270 = 4, VPCC_ORIENT_MIRROR = 5,
  VPCC_ORIENT_MROT90 = 6, VPCC_ORIENT_MROT180 = 7, VPCC_ORIENT_MROT270 = 8
} vpcc_orientation;

typedef struct vpcc_patch {
  uint32_t u0, v0;            /* uv0: location in the atlas, in blocks                */
  uint32_t size_u0, size_v0;  /* size_uv0: size in blocks                             */
  uint32_t u1, v1;            /* uv1: tangential / bitangential 3-D shift             */
  uint32_t d1;                /* depth shift (already mode-adjusted, decoder.rs:468-473) */
  uint32_t lod_x, lod_y;      /* level_of_detail, (1,1) in the supported envelope     */
  uint8_t  normal_axis, tangent_axis, bitangent_axis;   /* axes, each in 0..2         */
  uint8_t  projection_mode;   /* 0: min-depth related, 1: max-depth related           */
  uint8_t  orientation;       /* vpcc_orientation                                      */
  uint8_t  axis_of_additional_plane; /* must be 0 (src/codec.rs:429-440)              */
  uint8_t  reserved[2];
} vpcc_patch;

/* ------------------------------------------------------------ video planes */
/* Decoded video frames as the reference holds them in Image<T>
 * (src/decoder.rs:961-1021): YUV420, luma index v*stride+u, chroma index
 * (v/2)*cstride+(u/2) — nearest-neighbour chroma.  Strides are in ELEMENTS.
 * The reference ignores libav's linesize and uses stride == width (luma) and
 * width/2 (chroma); pass those to be bit-identical with it. */
typedef struct vpcc_image_u8 {      /* occupancy video frame; only luma is read */
  const uint8_t* y;
  uint32_t width, height, stride;
} vpcc_image_u8;

typedef struct vpcc_image_u16 {     /* geometry / attribute frame, 10-bit in u16 (YUV420P10LE) */
  const uint16_t* y;
  const uint16_t* u;                /* may be NULL for geometry (never read)     */
  const uint16_t* v;
  uint32_t width, height;           /* luma size                                  */
  uint32_t stride, cstride;         /* luma / chroma stride in elements           */
} vpcc_image_u16;

/* One atlas frame = one tile (the reference supports exactly one tile per
 * frame, src/decoder.rs:200-206) with the video frames it consumes:
 * occ_frames[f], geo_frames[0][f*map_count + m], attr_frames[0][f*map_count + m]
 * (src/codec.rs:294, 317, 330, 545, 620-637). */
typedef struct vpcc_frame_desc {
  uint32_t width, height;            /* tile.width/height == ASPS frame size          */
  uint32_t occupancy_resolution;     /* R = 1 << log2_patch_packing_block_size        */
  uint32_t occupancy_precision;      /* vps.frame_width / occ video width (decoder.rs:194) */
  uint32_t map_count;                /* map_count_minus1 + 1; parity is defined for 2  */
  uint32_t absolute_d1;              /* GeneratePointCloudParams::absolute_d1          */
  uint32_t attribute_count;          /* 0 or 1 (src/decoder.rs:133)                    */
  uint32_t flags;                    /* VPCC_FRAME_* below                             */
  vpcc_image_u8  occupancy;
  vpcc_image_u16 geometry[2];        /* D0, D1                                          */
  vpcc_image_u16 attribute[2];       /* attribute frame of layer 0 / 1                  */
  const vpcc_patch* patches;         /* tile.patches, ascending patch index             */
  uint32_t patch_count;
  uint32_t reserved;
} vpcc_frame_desc;

#define VPCC_FRAME_RGB444 0x1u  /* ColorFormat::_Rgb444: copy_rgb16_to_rgb8 instead of
                                   convert_yuv16_to_rgb8 (src/decoder.rs:301-305); unreachable in
                                   the reference (format is always Yuv420) — rejected as UNSUPPORTED */

/* Where the plane pointers of a vpcc_frame_desc live. */
typedef enum vpcc_memory_kind {
  VPCC_MEM_HOST = 0,    /* host pointers: the library stages them into HBM (H2D)   */
  VPCC_MEM_DEVICE = 1   /* device pointers on the context's GPU: borrowed, zero-copy */
} vpcc_memory_kind;

/* ---------------------------------------------------------------- context */
typedef struct vpcc_ctx vpcc_ctx;   /* one per GPU / per worker thread */
typedef struct vpcc_gof vpcc_gof;   /* a batch of independent frames resident in HBM */

int  vpcc_abi_version(void);
const char* vpcc_status_string(int status);

/* Creates a context on HIP device `device_id`.  Fails with VPCC_ERR_NO_DEVICE
 * when no usable GPU exists — there is no CPU fallback. */
int  vpcc_ctx_create(int device_id, vpcc_ctx** out);
void vpcc_ctx_destroy(vpcc_ctx* ctx);
const char* vpcc_last_error(const vpcc_ctx* ctx);
/* The context's compute stream (a hipStream_t): the stream kernels are launched on when a call is given a null
 * stream.  For callers that order their own work behind the reconstruction or time it with HIP events. */
void* vpcc_ctx_stream(const vpcc_ctx* ctx);

/* Binds the CALLING thread to the CPUs of the NUMA node the context's GPU hangs off (PCI bus id -> sysfs
 * numa_node / cpulist -> sched_setaffinity), so that the thread's staging buffers, its HIP calls and the copy
 * engines' host side stay on the socket next to the device.  Meant for the one worker thread that drives a
 * context (the reference's decode thread, src/lib.rs:113-137).  Returns VPCC_OK and the node in *node_out (-1: the
 * platform reports none — nothing was changed); VPCC_ERR_DEVICE when the affinity call fails. */
int  vpcc_ctx_bind_thread(vpcc_ctx* ctx, int* node_out);

/* Two homes.  VRAM consists of KINDS of regions — 32 GB of physical address space each, alternating — and the memory
 * system serves a launch whose traffic stays inside one kind about a tenth slower than one that spreads it evenly over
 * two: the same 128-frame reconstruction takes 0.45 or 0.52 ms depending on where hipMalloc happened to put the gof's
 * planes and outputs.  Nothing but a measurement tells the kinds apart (virtual addresses do not).
 * vpcc_ctx_reserve makes ONE allocation of `bytes` (rounded up to whole GiB), classifies its GiB granules once — the
 * reconstruction kernel's output pattern, positions in granule 0, colours in granule j, is slow (3.7 TB/s against 5.3)
 * exactly when j is of granule 0's kind; under a millisecond per granule — and from then on every gof of the context
 * keeps its big blocks (ingested planes, output arrays) there: eight frames (one per XCD) in the home of one kind, the
 * next eight in the other, so that any launch moves the same bytes in both.  No cost per gof, nothing moves, nothing is
 * measured again; blocks go back to the pool when their gof is destroyed.  A gof that does not fit takes the other home,
 * then plain allocations.  When the allocation turns out to lie in one kind only (on some GPUs the first 60 GB of VRAM
 * are alike), a second one of half its size is looked for further away — behind 16-GiB spacers that are freed again,
 * up to four times, while a third of the device's memory stays free — and kept as the other home: the pool then holds
 * 1.5 x `bytes`.  Cost: 30-50 ms for 32 GiB on a GPU whose memory is clean; the median of 321 reservations on this pool of
 * machines was 218 ms and the slowest took 4.4 s (the driver wipes what another process freed before it hands it out:
 * profiles/r04/pool.txt) — reserve it beside the first work, not in front of it.  The pool of a destroyed context stays
 * with the process and is taken over,
 * classification included, by the next context that reserves one on the device (memory given back to the driver is
 * wiped before it is handed out again, and every allocation of the process waits for that).  Call it once, before or
 * beside the context's first gofs (it may run on a thread of its own; gofs created meanwhile allocate as without it).  VPCC_ERR_STATE: the context has
 * a pool; VPCC_ERR_DEVICE: no memory for it (the context works as before). */
typedef struct vpcc_pool_info {
  uint64_t bytes;              /* size of the pool (0: none)                                                    */
  uint32_t granules;           /* GiB granules classified                                                       */
  uint32_t kinds;              /* 2: two kinds found; 1: every pairing ran alike — the homes are one            */
  uint64_t bytes_of_kind[2];   /* [0]: granule 0's kind                                                         */
  uint64_t in_use[2];          /* bytes handed out, by kind                                                     */
  float    probe_gbps_same;    /* the probe's rate in the slowest / fastest pairing, GB/s                       */
  float    probe_gbps_other;
  float    ms_spent;           /* allocation + classification, wall clock                                       */
  uint32_t other_home;         /* blocks that had to go to the other home                                       */
  uint32_t fallbacks;          /* blocks that did not fit the pool at all (allocations of their own)            */
  uint32_t reused;             /* 1: the pool of an earlier context of this process, taken over as it was       */
  uint32_t reserved0;
} vpcc_pool_info;
int  vpcc_ctx_reserve(vpcc_ctx* ctx, uint64_t bytes, vpcc_pool_info* out /* may be NULL */);
/* The same within a budget of wall-clock time: once `budget_ms` have passed (the allocation itself can take seconds right
 * after another process has freed tens of GB: the driver wipes memory before it hands it out again) no second home is
 * looked for — the pool is what the first allocation turned out to be, of one kind if need be.  0: no limit. */
int  vpcc_ctx_reserve_within(vpcc_ctx* ctx, uint64_t bytes, float budget_ms, vpcc_pool_info* out /* may be NULL */);
int  vpcc_ctx_pool_info(vpcc_ctx* ctx, vpcc_pool_info* out);
/* Device memory of the pool for a PRODUCER that leaves decoded planes in HBM — a GPU video decoder's frame pool, handed to
 * vpcc_gof_create as VPCC_MEM_DEVICE planes.  Planes take part in a launch's traffic like the outputs do: ask for the
 * frames' planes as the gof places its outputs — frames 0-7 of a launch from home 0, 8-15 from home 1, and so on (`home` =
 * (frame / 8) % 2).  Without a pool (or when it is full) the memory is a plain allocation.  The pointer is 256-byte
 * aligned and stays valid until vpcc_ctx_pool_free or the end of the context. */
int  vpcc_ctx_pool_alloc(vpcc_ctx* ctx, int home, size_t bytes, void** out);
int  vpcc_ctx_pool_free(vpcc_ctx* ctx, void* ptr);
/* Pools of destroyed contexts stay with the process (see above) until a context of the device takes them over, an
 * allocation of the library fails, or this is called: gives every kept pool of `device` back to the driver — for a
 * process that goes on to use the GPU by other means.  Returns 1 if there was one, else 0. */
int  vpcc_release_kept_pools(int device);

/* Plane ingest (stand-in for LibavcodecDecoder::decode, src/decoder.rs:1089-1156, whose Vec<u8> planes
 * are the H2D source): page-locks a host range so that uploads from it are true asynchronous DMA.
 * Planes of a VPCC_GOF_ASYNC_UPLOAD gof that lie next to each other (up to 256 KB apart) inside ONE region page-locked
 * here — a decoder's frame pool, a decoded-GOF container — are copied as whole stretches, whatever lies between them
 * included, and keep their arrangement on the device: one copy per eight frames runs the link at its full rate in both
 * directions at once (57 + 53 GB/s on an MI355X), which neither a copy per plane (34 GB/s) nor a kernel reading the host
 * memory in place (46 GB/s beside the results going back) does.  Page-locked planes that do not lie like that are pulled
 * by kernel, planes with padded rows are copied one by one. */
int  vpcc_host_pin(vpcc_ctx* ctx, const void* ptr, size_t bytes);
int  vpcc_host_unpin(vpcc_ctx* ctx, const void* ptr);
/* Page-locked host buffers for results (D2H straight into the memory the consumer keeps). */
int  vpcc_host_alloc(vpcc_ctx* ctx, size_t bytes, void** out);
int  vpcc_host_free(vpcc_ctx* ctx, void* ptr);   /* ctx may be NULL once the context is gone */

/* Validates a frame descriptor on the host exactly as far as the reference's
 * asserts would fire while walking it (patch extents, plane sizes, supported
 * envelope).  Pure host function, no GPU needed. */
int  vpcc_frame_validate(const vpcc_frame_desc* frame);

/* Upper bound on the number of points one frame can produce
 * (map_count × width × height). */
uint64_t vpcc_frame_capacity_bound(const vpcc_frame_desc* frame);

/* ------------------------------------------ one-shot seam replacements (sync) */
/* Replaces codec::generate_block_to_patch_from_occupancy_map_video
 * (src/codec.rs:205-250).  block_to_patch_out has (width/R)*(height/R) entries;
 * 0 = unowned, else patch_index+1 (the reference stores usize; widen in the shim). */
int vpcc_generate_block_to_patch(vpcc_ctx* ctx, const vpcc_frame_desc* frame,
                                 vpcc_memory_kind planes, uint32_t* block_to_patch_out);

/* Replaces the nearest-neighbour occupancy upsample (src/codec.rs:288-301);
 * occupancy_map_out has width*height bytes (tile.occupancy_map). */
int vpcc_upsample_occupancy(vpcc_ctx* ctx, const vpcc_frame_desc* frame,
                            vpcc_memory_kind planes, uint8_t* occupancy_map_out);

/* Replaces, for one frame, the body of the per-frame loop src/decoder.rs:249-305:
 * block->patch, generate_point_cloud (incl. color_point_cloud) and
 * convert_yuv16_to_rgb8.  xyz_out/rgb_out are host arrays of `capacity` entries
 * (rgb_out may be NULL when attribute_count == 0); patch_index_out (optional,
 * may be NULL) receives the `partition` vector (src/codec.rs:452) as u16.
 * *n_points receives the point count even on VPCC_ERR_CAPACITY. */
int vpcc_reconstruct_frame(vpcc_ctx* ctx, const vpcc_frame_desc* frame, vpcc_memory_kind planes,
                           vpcc_point3* xyz_out, vpcc_color3* rgb_out, uint16_t* patch_index_out,
                           size_t capacity, size_t* n_points);

/* ------------------------------------------------- batched GOF path (async) */
/* Frames of a group-of-frames are independent (src/decoder.rs:186, 403-407).
 * A vpcc_gof keeps n_frames frames resident in HBM and reconstructs them in
 * one batched launch sequence.  With VPCC_MEM_HOST the planes are copied to
 * HBM at creation; with VPCC_MEM_DEVICE they are borrowed and must outlive the
 * gof: nothing reads them before vpcc_gof_reconstruct, which reads them on ITS
 * stream (a decoder that writes them on that stream needs no other ordering),
 * and vpcc_gof_create returns without waiting for the device — the host writes
 * O(patches) per frame, the per-block work is the launches'.
 * capacity_points is the per-frame output capacity (0 = the safe bound
 * vpcc_frame_capacity_bound()).
 * Planes stay in the raster layout a video decoder hands over, whoever owns them: the reconstruction kernel
 * reads them where they lie, nothing is re-arranged in HBM.  Frames whose capacity exceeds 715 827 880 points
 * (32-bit byte offsets into the positions) are reconstructed by the general kernel sequence. */
int  vpcc_gof_create(vpcc_ctx* ctx, const vpcc_frame_desc* frames, uint32_t n_frames,
                     vpcc_memory_kind planes, uint64_t capacity_points, uint32_t gof_flags,
                     vpcc_gof** out);
void vpcc_gof_destroy(vpcc_gof* gof);

#define VPCC_GOF_WANT_PATCH_INDEX 0x1u  /* also emit per-point patch index (partition)          */
#define VPCC_GOF_FORCE_GENERAL    0x2u  /* force the general (all-orientation) kernel sequence  */
#define VPCC_GOF_PROFILE          0x4u  /* record per-kernel HIP-event timings                  */
#define VPCC_GOF_ASYNC_UPLOAD     0x8u  /* VPCC_MEM_HOST only: vpcc_gof_create returns while the H2D copies
                                           are still in flight on the context's copy stream (plane ingest
                                           overlapping the previous GOF's kernels).  The planes must be pinned
                                           (vpcc_host_pin) and stay valid until the first vpcc_gof_sync /
                                           vpcc_gof_point_counts / vpcc_gof_download of this gof.          */

#define VPCC_GOF_COPY_PLANES      0x20u /* VPCC_MEM_DEVICE only: the planes are copied (device to device, on the context's
                                           copy stream; they must be complete when vpcc_gof_create is called and may go
                                           when it returns) into memory of the gof's own — with a reserved pool
                                           (vpcc_ctx_reserve) into its two homes, like ingested host planes.          */

/* Enqueues the reconstruction of frames [first, first+count) on `hip_stream`
 * (a hipStream_t passed as void*; NULL = the context's own stream).  Returns
 * immediately; results are valid after vpcc_gof_sync() or a stream sync.
 * Launches on one gof are serialised by the library (a launch on a different stream first waits for the
 * previous launch's kernels): they share the gof's output arrays and control words.  A device-side error
 * (VPCC_ERR_DEVICE from the counts/download calls: a look-back wait that gave up) is sticky for the gof.
 * Every vpcc_* call that touches the GPU makes the context's device the calling thread's current HIP
 * device and leaves it so (vpcc_host_free does not). */
int vpcc_gof_reconstruct(vpcc_gof* gof, uint32_t first, uint32_t count, void* hip_stream);
int vpcc_gof_sync(vpcc_gof* gof);

/* Per-frame point counts of the last reconstruct (synchronises). */
int vpcc_gof_point_counts(vpcc_gof* gof, uint32_t* counts_out /* n_frames */);

/* block_to_patch of frame `frame` (src/codec.rs:205-250: 0 = unowned, else patch index + 1; (width / R) x (height / R)
 * entries) and the number of work items the single-pass kernel has for the frame (0 for a gof of the general sequence).
 * Every vpcc_gof_reconstruct builds both on the device, on its stream, from the occupancy plane as it is then — a gof
 * that borrows the caller's device planes may be launched again after new frames have been decoded into them.  A launch
 * of the single-pass kernel keeps the map in its planning kernel's LDS (k_plan_tiles); this call plans the frame once
 * more, from the planes as they are NOW, with the map written out (general sequence: the map of the last launch that
 * covered the frame).  VPCC_ERR_STATE before the first launch.  Synchronises; either out-pointer may be NULL. */
int vpcc_gof_block_to_patch(vpcc_gof* gof, uint32_t frame, uint32_t* block_to_patch_out, uint32_t* work_items_out);

/* Device pointers of frame `frame`'s outputs (vpcc_point3[capacity],
 * vpcc_color3[capacity], uint16_t[capacity] or NULL) and of its device-side
 * point counter (uint32_t).  Any of the out-pointers may be NULL. */
int vpcc_gof_device_outputs(vpcc_gof* gof, uint32_t frame, void** d_xyz, void** d_rgb,
                            void** d_patch_index, void** d_count);

/* Copies one frame's result to host arrays (synchronises). */
int vpcc_gof_download(vpcc_gof* gof, uint32_t frame, vpcc_point3* xyz_out, vpcc_color3* rgb_out,
                      uint16_t* patch_index_out, size_t capacity, size_t* n_points);

/* The same without the wait: the copies are enqueued on the context's download stream (the first call waits for the
 * frame's point count), `*n_points` is final on return, and the arrays are complete once vpcc_gof_download_wait(gof,
 * frame) has returned — which, unlike every other call on a gof, may come from another thread than the one that drives
 * the context (it touches nothing but the frame's completion event).  Many small synchronous downloads leave the link
 * idle between them (16 GB/s for 7-MB frames); a window of asynchronous ones keeps the copy engine fed.  Page-locked
 * destinations (vpcc_host_alloc / vpcc_host_pin), or the copies are not asynchronous. */
int vpcc_gof_download_async(vpcc_gof* gof, uint32_t frame, vpcc_point3* xyz_out, vpcc_color3* rgb_out,
                            uint16_t* patch_index_out, size_t capacity, size_t* n_points);
int vpcc_gof_download_wait(vpcc_gof* gof, uint32_t frame);

/* Per-frame status of the last reconstruct: VPCC_OK or VPCC_ERR_CAPACITY. */
int vpcc_gof_frame_status(vpcc_gof* gof, uint32_t frame);

/* Profile mode (VPCC_GOF_PROFILE): names and milliseconds of the kernels of the
 * last vpcc_gof_reconstruct, measured with HIP events on the launch stream.
 * Returns the number of kernels (<= max). */
int vpcc_gof_kernel_times(vpcc_gof* gof, const char** names_out, float* ms_out, int max);

/* Profile mode keeps the event pairs of the last 512 launches (one per vpcc_gof_reconstruct; a following
 * vpcc_gof_smooth adds its kernels to the same launch).  Mean duration per kernel name over the last
 * `last_n` launches (0 = all kept); *launches_out = launches averaged.  Lets a caller time a long
 * back-to-back region and read the per-launch kernel durations of exactly those launches.
 * Returns the number of distinct kernels (<= max). */
/* Profile mode: time only every `every`-th vpcc_gof_reconstruct (default 1 = all).  An event pair costs a few
 * microseconds of stream time per launch; sampling keeps a back-to-back timed region undisturbed while its
 * own launches are still the ones measured. */
int vpcc_gof_profile_interval(vpcc_gof* gof, uint32_t every);
int vpcc_gof_kernel_time_means(vpcc_gof* gof, uint32_t last_n, const char** names_out, float* mean_ms_out,
                               uint32_t* launches_out, int max);

/* Algorithmic bytes of frame `frame` as SURVEY.md §8(d) defines them
 * (occupancy + geometry luma + attribute Y/U/V planes read once, 9 B/point
 * written once), using the measured point count of the last reconstruct. */
int vpcc_gof_algorithmic_bytes(vpcc_gof* gof, uint32_t frame, uint64_t* bytes_out);

/* ------------------------------------------------------- smoothing */
/* Grid-based geometry / colour smoothing of reconstructed frames, in place in HBM.  The reference has
 * only unimplemented!() hooks here (src/decoder.rs:291-299, 630-658; src/codec.rs:498-500) and parses
 * just the SEI syntax (src/bitstream/reader.rs:1452-1505), so there is no reference result: the
 * behaviour is this library's own integer specification "gs1"/"cs1" (oracle/vpcc_smoothing_spec.h),
 * against which the kernels are tested bit for bit. */
typedef struct vpcc_smoothing_params {
  uint32_t geometry_bitdepth_3d;        /* gi.geometry_3d_coordinates_bitdepth_minus1 + 1               */
  uint32_t flags;                       /* VPCC_SMOOTH_GEOMETRY | VPCC_SMOOTH_COLOR                       */
  uint32_t grid_size;                   /* SeiGeometrySmoothing::grid_size_minus_2 + 2                    */
  uint32_t threshold;                   /* SeiGeometrySmoothing::threshold                                */
  uint32_t color_grid_size;             /* ColorSmoothingParams::_cgrid_size   (src/codec.rs:180-186)     */
  uint32_t color_threshold_smoothing;   /* ColorSmoothingParams::_threshold_color_smoothing               */
  uint32_t color_threshold_difference;  /* ColorSmoothingParams::_threshold_color_difference              */
  uint32_t reserved;
} vpcc_smoothing_params;
#define VPCC_SMOOTH_GEOMETRY 0x1u
#define VPCC_SMOOTH_COLOR    0x2u
/* Smooths frames [first, first+count) of the last reconstruct.  The gof must have been created with
 * VPCC_GOF_WANT_PATCH_INDEX (the filters need each point's patch).  Geometry first, then colour.
 * Device memory: on first use the gof allocates, and keeps until vpcc_gof_destroy, a scratch of dense grids —
 * 34 bytes per cell, w = ceil(2^bitdepth / grid_size), w^3 cells (+ (w + 1)^3 bytes) per frame slot: 71 MB at 10 bits and
 * grid 8, 570 MB at 11 bits; one slot per frame of the range up to ~16 GiB in all (larger ranges are smoothed in chunks
 * of frames) — plus 4.3 bytes per point of capacity and frame for the cell lists (reserved; a few per cent of it are
 * touched).  It is zeroed once, when allocated.  When both filters are asked for with the same grid size, one pass over
 * the points serves both (50 bytes per cell, 4.4 more bytes per point); grids of 2^32 cells and more are not supported.
 * Bound of the all-sum cells, CHECKED on the device: at most 65 537 points of a frame in one grid cell (65 537 x 65 535
 * < 2^32: no 32-bit sum of coordinates, colours or patch indices can overflow; the specification's u32 sums would wrap
 * where the kernels' packed 64-bit adds would carry).  A frame that exceeds it — hundreds of points per position —
 * makes the gof's next vpcc_gof_point_counts / _download / _frame_status return VPCC_ERR_UNSUPPORTED (sticky). */
int vpcc_gof_smooth(vpcc_gof* gof, uint32_t first, uint32_t count, const vpcc_smoothing_params* params,
                    void* hip_stream);

/* ------------------------------------------------- host mirror of the library API */
/* C view of the C++ class tmc2rs::Decoder (tmc2-rs_amd/csrc/decoder.hpp), which mirrors the
 * reference's public API: Decoder::new (src/lib.rs:71-78), start() (:97-138), recv_frame() (:143-145).
 * Input: a decoded-GOF container (.vpccgof) — patch tables + decoded planes, the state after the
 * reference's three decompress() calls.  Frames are sharded over `devices` and delivered in
 * presentation order through a capacity-1 channel. */
typedef struct vpcc_decoder vpcc_decoder;
int  vpcc_decoder_open(const char* path, const int* devices, int n_devices, vpcc_decoder** out);
/* Same decoder on a V3C sample stream (.bin) parsed by the library (vpcc_v3c_*), with the three video
 * sub-bitstreams decoded by an EXTERNAL HEVC decoder into raw planar 4:2:0 files in the decoder's native
 * format, all GOFs back to back: occupancy 8-bit, geometry / attribute 16-bit little endian — the bytes the
 * reference copies out of libav's frames (src/decoder.rs:1131-1141).  `attribute_yuv` may be null;
 * `occupancy_precision` = frame width / occupancy video width (src/decoder.rs:194).  A stream the parser
 * rejects makes vpcc_decoder_start fail with the text in vpcc_decoder_error. */
int vpcc_decoder_open_v3c(const char* bin_path, const char* occupancy_yuv, const char* geometry_yuv,
                          const char* attribute_yuv, uint32_t occupancy_precision, const int* devices,
                          int n_devices, vpcc_decoder** out);
/* VPCC_ERR_STATE when called twice (the reference panics: "can only be started once"). */
/* The reference's post-processing switches (Params::apply_geo_smoothing_type / apply_attr_smoothing_type, src/lib.rs:45-46:
 * private and always false there, with unimplemented!() behind them, src/decoder.rs:291-299).  Between open and start.
 * Geometry smoothing runs for a GOF when its switch is on AND the GOF carries a geometry-smoothing SEI
 * (src/decoder.rs:291, 630-637; grid size and threshold are the SEI's) — or, for inputs without syntax (a .vpccgof
 * container), with `params`' geometry_bitdepth_3d / grid_size / threshold when grid_size >= 2.  Colour smoothing runs
 * when its switch is on, with `params`' color_* fields (the reference parses no attribute-smoothing SEI).  The filters are
 * this library's own specification (vpcc_gof_smooth); frames are delivered smoothed.  `params` may be NULL. */
int  vpcc_decoder_set_smoothing(vpcc_decoder* dec, int apply_geo_smoothing, int apply_attr_smoothing,
                                const vpcc_smoothing_params* params);
int  vpcc_decoder_start(vpcc_decoder* dec);
/* 1 and the next frame (pointers valid until the next call), or 0 at end of stream — also after a
 * failure in the worker, like the reference's consumer sees None after a worker panic. */
int  vpcc_decoder_recv_frame(vpcc_decoder* dec, size_t* n_points, const vpcc_point3** xyz, const vpcc_color3** rgb);
const char* vpcc_decoder_error(vpcc_decoder* dec);
/* Consumes the rest of the stream inside the library (no per-frame copies into the caller) and reports
 * frames, points and wall seconds since the call: the end-to-end, PCIe-inclusive rate of the ingest ->
 * reconstruct -> D2H pipeline. */
int  vpcc_decoder_drain(vpcc_decoder* dec, uint64_t* frames, uint64_t* points, double* seconds);
/* Seconds from the start of the last vpcc_decoder_drain to its first frame (contexts, page-locking the input,
 * first GOF): the start-up latency a long stream amortises. */
double vpcc_decoder_first_frame_seconds(const vpcc_decoder* d);
/* What the decoder's worker did so far (complete after end of stream).  The worker reconstructs every GOF that
 * is resident on a device in ONE launch: the first launch covers one GOF (start-up latency), later ones up to
 * four (GOFs and their frames are independent: src/lib.rs:113-137, src/decoder.rs:186). */
typedef struct vpcc_decoder_stats_t {
  uint64_t launches;               /* reconstruction launches, all lanes */
  uint64_t frames;                 /* frames those launches covered */
  uint32_t max_frames_per_launch;  /* largest launch on one lane */
  uint32_t lanes;                  /* devices in use */
  double kernel_seconds;           /* HIP-event time of the reconstruction kernels, summed over launches and lanes */
  double launch_seconds;           /* host time of planning + upload enqueue + launch, slowest lane per unit, summed */
  int32_t numa_node[8];            /* NUMA node lane i was bound to (-1: none reported), first 8 lanes */
} vpcc_decoder_stats_t;
int  vpcc_decoder_stats(const vpcc_decoder* dec, vpcc_decoder_stats_t* out);
void vpcc_decoder_close(vpcc_decoder* dec);

/* writer::PlyWriter::write, ASCII (src/writer.rs:25-74); rgb may be NULL (no colour properties). */
int  vpcc_write_ply(const char* path, const vpcc_point3* xyz, const vpcc_color3* rgb, size_t n_points);
/* Same with `binary` != 0: "format binary_little_endian 1.0" (the variant the reference's writer has commented
 * out, src/writer.rs:10-11, 39-44) with the same property list — 3 x uint32 (+ 3 x uchar) per vertex. */
int  vpcc_write_ply_format(const char* path, const vpcc_point3* xyz, const vpcc_color3* rgb, size_t n, int binary);

/* -------------------------------------------- syntax side (host only) */
/* V3C bit reader (src/bitstream.rs:53-190): read/peek MSB-first, Exp-Golomb, byte_align, copy_from. */
typedef struct vpcc_bitstream vpcc_bitstream;
vpcc_bitstream* vpcc_bs_new(const uint8_t* data, size_t n);
void  vpcc_bs_free(vpcc_bitstream* bs);
int   vpcc_bs_read(vpcc_bitstream* bs, unsigned bits, uint32_t* out);
int   vpcc_bs_peek(vpcc_bitstream* bs, unsigned bits, uint32_t* out);
int   vpcc_bs_read_uvlc(vpcc_bitstream* bs, uint32_t* out);
int   vpcc_bs_read_svlc(vpcc_bitstream* bs, int32_t* out);
int   vpcc_bs_byte_align(vpcc_bitstream* bs);
void  vpcc_bs_reset(vpcc_bitstream* bs);
int   vpcc_bs_copy_from(vpcc_bitstream* dst, vpcc_bitstream* src, size_t start_byte, size_t size);
size_t vpcc_bs_data(const vpcc_bitstream* bs, const uint8_t** data);
void  vpcc_bs_position(const vpcc_bitstream* bs, size_t* bytes, unsigned* bits);
/* SampleStreamV3CUnit::from_bitstream (src/bitstream/reader.rs:623-670): unit types (data[0] >> 3),
 * payload offsets and sizes of a V3C sample stream. */
int   vpcc_v3c_split(const uint8_t* data, size_t n, uint32_t max_units, uint8_t* types, size_t* offsets,
                     size_t* sizes, uint32_t* n_units, size_t* header_size);
/* VideoBitstream::sample_stream_to_bytestream (src/bitstream.rs:216-289): 4-byte NAL length prefixes ->
 * Annex-B start codes (codec_id 0 H264, 1 H265, 2 H266), for an external video decoder. */
int   vpcc_sample_stream_to_bytestream(const uint8_t* data, size_t n, int codec_id, uint8_t* out,
                                       size_t out_capacity, size_t* out_size);
/* Patch-table builder: one Intra patch data unit -> vpcc_patch (create_patch_frame, src/decoder.rs:415-486). */
typedef struct vpcc_patch_frame_params {
  uint32_t log2_patch_packing_block_size;   /* asps                                               */
  uint32_t geometry_3d_bitdepth;            /* asps.geometry_3d_bitdepth_minus1 + 1               */
  uint32_t pos_min_d_quantizer;             /* ath                                                */
  uint32_t patch_size_quantizer_present_flag, patch_size_info_quantizer_x, patch_size_info_quantizer_y;
  uint32_t plr_enabled_flag;
  uint32_t reserved;
} vpcc_patch_frame_params;
typedef struct vpcc_intra_pdu {
  uint32_t pos_2d_x, pos_2d_y, size_2d_x_minus1, size_2d_y_minus1;
  uint32_t pos_3d_offset_u, pos_3d_offset_v, pos_3d_offset_d, pos_3d_range_d;
  uint32_t projection_id, orientation_index, lod_enabled_flag, reserved;
} vpcc_intra_pdu;
int   vpcc_patch_from_intra_pdu(const vpcc_patch_frame_params* fp, const vpcc_intra_pdu* pdu, vpcc_patch* out);

/* V3C / V-PCC high-level syntax (SampleStreamV3CUnit::decode + V3CUnit::decode, src/bitstream/reader.rs:23-160,
 * 257-2037) and the per-GOF patch frames / reconstruction parameters (Decoder::create_patch_frame and
 * new_generate_point_cloud_params, src/decoder.rs:320-517, 590-627).  One GOF = the V3C units up to the next
 * V3C parameter set (src/lib.rs:119-133).  Features the reference rejects (assert!/unimplemented!) return
 * VPCC_ERR_UNSUPPORTED, malformed streams VPCC_ERR_INVALID_ARG; after an error the stream yields no more
 * GOFs, like the reference's worker thread, which dies on the panic. */
typedef struct vpcc_v3c_stream vpcc_v3c_stream;
typedef struct vpcc_v3c_gof_info {
  uint32_t frame_count;                      /* atlas tile layers == frames of the GOF                         */
  uint32_t frame_width, frame_height;        /* vps (what occupancy_precision is derived from, decoder.rs:194) */
  uint32_t atlas_frame_width, atlas_frame_height;   /* asps                                                    */
  uint32_t map_count;                        /* vps.map_count_minus1 + 1                                       */
  uint32_t absolute_d1;                      /* map_count_minus1 == 0 || map_absolute_coding_enable_flag[1]    */
  uint32_t occupancy_resolution;             /* 1 << asps.log2_patch_packing_block_size                        */
  uint32_t geometry_3d_bitdepth;             /* gi.geometry_3d_coordinates_bitdepth_minus1 + 1                 */
  uint32_t atlas_geometry_3d_bitdepth;       /* asps.geometry_3d_bitdepth_minus1 + 1 (patch d1, decoder.rs:471) */
  uint32_t geometry_2d_bitdepth, occupancy_2d_bitdepth, attribute_2d_bitdepth;
  uint32_t attribute_count;
  uint32_t occupancy_codec_id, geometry_codec_id, attribute_codec_id;
  uint32_t profile_codec_group_idc, profile_toolset_idc, profile_reconstruction_idc, level_idc;
  uint32_t use_eight_orientations_flag, remove_duplicate_point_enabled_flag;
  uint32_t geometry_smoothing_sei;           /* a prefix geometry-smoothing SEI precedes the first tile layer  */
  uint32_t smoothing_grid_size, smoothing_threshold;   /* grid_size_minus_2 + 2, threshold of its method-1 instance */
  uint32_t reserved;
  size_t   video_bytes[3];                   /* occupancy, geometry, attribute video sub-bitstream sizes       */
} vpcc_v3c_gof_info;
int   vpcc_v3c_open(const uint8_t* data, size_t n, vpcc_v3c_stream** out);
void  vpcc_v3c_close(vpcc_v3c_stream* s);
const char* vpcc_v3c_error(const vpcc_v3c_stream* s);
uint32_t vpcc_v3c_unit_count(const vpcc_v3c_stream* s);
/* Parses the next GOF; *have_gof = 0 at the end of the stream.  `info` may be null. */
int   vpcc_v3c_next_gof(vpcc_v3c_stream* s, int* have_gof, vpcc_v3c_gof_info* info);
/* Patches of frame `frame` of the current GOF in atlas order; `out` may be null to query the count.
 * *frame_index receives the atlas frame order count (as the reference's u8 frame_index). */
int   vpcc_v3c_frame_patches(const vpcc_v3c_stream* s, uint32_t frame, vpcc_patch* out, uint32_t capacity,
                             uint32_t* n_patches, uint32_t* frame_index);
/* Video sub-bitstream of the current GOF (0 occupancy, 1 geometry, 2 attribute), sample-stream framed: pass it
 * to vpcc_sample_stream_to_bytestream for an external video decoder.  The pointer lives until the next call
 * of vpcc_v3c_next_gof / vpcc_v3c_close. */
int   vpcc_v3c_video(const vpcc_v3c_stream* s, int video, const uint8_t** data, size_t* n);

#ifdef __cplusplus
}
#endif
#endif /* VPCC_RECON_H */
