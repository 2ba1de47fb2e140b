"""ctypes mirror of include/vpcc_recon.h and loader of libvpcc_recon.so.

The library is the product: hand-written HIP kernels + C++ host runtime behind
a C ABI.  There is NO Python/CPU fallback — if the shared object is missing the
import of the wrapper fails loudly (RuntimeError), and if no GPU is present
vpcc_ctx_create returns VPCC_ERR_NO_DEVICE.
"""
import ctypes as C
import os

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
PROJECT_DIR = os.path.dirname(PKG_DIR)          # tmc2-rs_amd/
REPO_ROOT = os.path.dirname(PROJECT_DIR)
LIB_PATH = os.path.join(PROJECT_DIR, "libvpcc_recon.so")
# tools/ only: the diagnostic build of the same sources (`make diag`, run-time ablation switches of the
# tile kernel).  tests/, bench.py and __graft_entry__ never set this.
if os.environ.get("VPCC_DIAG_LIB") == "1":
    LIB_PATH = os.path.join(PROJECT_DIR, "libvpcc_recon_diag.so")
elif os.environ.get("VPCC_DIAG_LIB"):                       # tools/ab_multi.sh: any other build of the library, by file name
    LIB_PATH = os.path.join(PROJECT_DIR, os.path.basename(os.environ["VPCC_DIAG_LIB"]))

VPCC_OK = 0
VPCC_ERR_INVALID_ARG = 1
VPCC_ERR_UNSUPPORTED = 2
VPCC_ERR_PATCH_OUT_OF_CANVAS = 3
VPCC_ERR_SHORT_VIDEO = 4
VPCC_ERR_CAPACITY = 5
VPCC_ERR_DEVICE = 6
VPCC_ERR_NO_DEVICE = 7
VPCC_ERR_STATE = 8

VPCC_MEM_HOST = 0
VPCC_MEM_DEVICE = 1

VPCC_GOF_WANT_PATCH_INDEX = 0x1
VPCC_GOF_FORCE_GENERAL = 0x2
VPCC_GOF_PROFILE = 0x4
VPCC_GOF_ASYNC_UPLOAD = 0x8
VPCC_GOF_COPY_PLANES = 0x20

ORIENT_DEFAULT, ORIENT_SWAP, ORIENT_ROT90, ORIENT_ROT180, ORIENT_ROT270 = 0, 1, 2, 3, 4
ORIENT_MIRROR, ORIENT_MROT90, ORIENT_MROT180, ORIENT_MROT270 = 5, 6, 7, 8


class Patch(C.Structure):
    _fields_ = [
        ("u0", C.c_uint32), ("v0", C.c_uint32),
        ("size_u0", C.c_uint32), ("size_v0", C.c_uint32),
        ("u1", C.c_uint32), ("v1", C.c_uint32),
        ("d1", C.c_uint32),
        ("lod_x", C.c_uint32), ("lod_y", C.c_uint32),
        ("normal_axis", C.c_uint8), ("tangent_axis", C.c_uint8), ("bitangent_axis", C.c_uint8),
        ("projection_mode", C.c_uint8),
        ("orientation", C.c_uint8),
        ("axis_of_additional_plane", C.c_uint8),
        ("reserved", C.c_uint8 * 2),
    ]


# numpy view of the same record (for vectorised patch-table construction)
PATCH_DTYPE = np.dtype([
    ("u0", "<u4"), ("v0", "<u4"), ("size_u0", "<u4"), ("size_v0", "<u4"),
    ("u1", "<u4"), ("v1", "<u4"), ("d1", "<u4"), ("lod_x", "<u4"), ("lod_y", "<u4"),
    ("normal_axis", "u1"), ("tangent_axis", "u1"), ("bitangent_axis", "u1"),
    ("projection_mode", "u1"), ("orientation", "u1"), ("axis_of_additional_plane", "u1"),
    ("reserved", "u1", (2,)),
])
assert PATCH_DTYPE.itemsize == C.sizeof(Patch) == 44


class ImageU8(C.Structure):
    _fields_ = [("y", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32), ("stride", C.c_uint32)]


class ImageU16(C.Structure):
    _fields_ = [("y", C.c_void_p), ("u", C.c_void_p), ("v", C.c_void_p),
                ("width", C.c_uint32), ("height", C.c_uint32),
                ("stride", C.c_uint32), ("cstride", C.c_uint32)]


class FrameDesc(C.Structure):
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32),
        ("occupancy_resolution", C.c_uint32), ("occupancy_precision", C.c_uint32),
        ("map_count", C.c_uint32), ("absolute_d1", C.c_uint32),
        ("attribute_count", C.c_uint32), ("flags", C.c_uint32),
        ("occupancy", ImageU8),
        ("geometry", ImageU16 * 2),
        ("attribute", ImageU16 * 2),
        ("patches", C.c_void_p),
        ("patch_count", C.c_uint32), ("reserved", C.c_uint32),
    ]


class SmoothingParams(C.Structure):
    _fields_ = [("geometry_bitdepth_3d", C.c_uint32), ("flags", C.c_uint32), ("grid_size", C.c_uint32),
                ("threshold", C.c_uint32), ("color_grid_size", C.c_uint32),
                ("color_threshold_smoothing", C.c_uint32), ("color_threshold_difference", C.c_uint32),
                ("reserved", C.c_uint32)]


VPCC_SMOOTH_GEOMETRY = 0x1
VPCC_SMOOTH_COLOR = 0x2

POINT3_DTYPE = np.dtype([("x", "<u2"), ("y", "<u2"), ("z", "<u2")])
COLOR3_DTYPE = np.dtype([("r", "u1"), ("g", "u1"), ("b", "u1")])


def _ptr(a):
    return None if a is None else a.ctypes.data


def _plane(a, dtype):
    """(keepalive, pointer, width, height, stride in elements) of a 2-D plane.  A row-strided
    view (e.g. padded[:, :w]) is passed as is, with its stride; anything else is made contiguous."""
    a = np.asarray(a)
    if a.dtype != dtype:
        a = a.astype(dtype)
    item = a.dtype.itemsize
    if not (a.ndim == 2 and a.strides[1] == item and a.strides[0] % item == 0 and a.strides[0] >= a.shape[1] * item):
        a = np.ascontiguousarray(a)
    return a, a.ctypes.data, a.shape[1], a.shape[0], a.strides[0] // item


def host_frame_desc(frame):
    """Builds a FrameDesc over the numpy arrays of a synthetic/ingested frame
    dict (see synth.make_frame).  Returns (desc, keepalive)."""
    d = FrameDesc()
    d.width, d.height = int(frame["width"]), int(frame["height"])
    d.occupancy_resolution = int(frame["occupancy_resolution"])
    d.occupancy_precision = int(frame["occupancy_precision"])
    d.map_count = int(frame.get("map_count", 2))
    d.absolute_d1 = int(frame.get("absolute_d1", 1))
    d.attribute_count = int(frame.get("attribute_count", 1))
    d.flags = int(frame.get("flags", 0))
    keep = []
    occ, d.occupancy.y, d.occupancy.width, d.occupancy.height, d.occupancy.stride = _plane(frame["occupancy"], np.uint8)
    keep.append(occ)
    for m in range(2):
        g = frame["geometry"][m] if m < len(frame["geometry"]) else None
        if g is not None:
            G = d.geometry[m]
            g, G.y, G.width, G.height, G.stride = _plane(g, np.uint16)
            keep.append(g)
            G.cstride = G.width // 2
        a = frame["attribute"][m] if m < len(frame["attribute"]) else None
        if a is not None:
            A = d.attribute[m]
            y, A.y, A.width, A.height, A.stride = _plane(a[0], np.uint16)
            u, A.u, _, _, A.cstride = _plane(a[1], np.uint16)
            v, A.v, _, _, cs2 = _plane(a[2], np.uint16)
            assert cs2 == A.cstride, "U and V planes must share one stride"
            keep += [y, u, v]
    patches = np.ascontiguousarray(frame["patches"], dtype=PATCH_DTYPE)
    keep.append(patches)
    d.patches = _ptr(patches) if len(patches) else None
    d.patch_count = len(patches)
    return d, keep


_lib = None


class PoolInfo(C.Structure):              # vpcc_pool_info
    _fields_ = [("bytes", C.c_uint64), ("granules", C.c_uint32), ("kinds", C.c_uint32),
                ("bytes_of_kind", C.c_uint64 * 2), ("in_use", C.c_uint64 * 2),
                ("probe_gbps_same", C.c_float), ("probe_gbps_other", C.c_float), ("ms_spent", C.c_float),
                ("other_home", C.c_uint32), ("fallbacks", C.c_uint32), ("reused", C.c_uint32), ("reserved0", C.c_uint32)]


class DecoderStats(C.Structure):          # vpcc_decoder_stats_t
    _fields_ = [("launches", C.c_uint64), ("frames", C.c_uint64), ("max_frames_per_launch", C.c_uint32),
                ("lanes", C.c_uint32), ("kernel_seconds", C.c_double), ("launch_seconds", C.c_double),
                ("numa_node", C.c_int32 * 8)]


class V3cGofInfo(C.Structure):
    """vpcc_v3c_gof_info (include/vpcc_recon.h)."""
    _fields_ = [(n, C.c_uint32) for n in (
        "frame_count", "frame_width", "frame_height", "atlas_frame_width", "atlas_frame_height", "map_count",
        "absolute_d1", "occupancy_resolution", "geometry_3d_bitdepth", "atlas_geometry_3d_bitdepth",
        "geometry_2d_bitdepth", "occupancy_2d_bitdepth", "attribute_2d_bitdepth", "attribute_count",
        "occupancy_codec_id", "geometry_codec_id", "attribute_codec_id", "profile_codec_group_idc",
        "profile_toolset_idc", "profile_reconstruction_idc", "level_idc", "use_eight_orientations_flag",
        "remove_duplicate_point_enabled_flag", "geometry_smoothing_sei", "smoothing_grid_size",
        "smoothing_threshold", "reserved")] + [("video_bytes", C.c_size_t * 3)]


def load_library():
    """Loads libvpcc_recon.so (built in-tree by `make` / __graft_entry__.build()).
    Raises — never falls back — when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"HIP extension {LIB_PATH} is missing: build it with `make -C {REPO_ROOT}` "
            "(there is no CPU fallback for the reconstruction path)")
    lib = C.CDLL(LIB_PATH)
    vp, u32, u64, sz = C.c_void_p, C.c_uint32, C.c_uint64, C.c_size_t
    FD = C.POINTER(FrameDesc)
    lib.vpcc_abi_version.restype = C.c_int
    lib.vpcc_status_string.restype = C.c_char_p
    lib.vpcc_status_string.argtypes = [C.c_int]
    lib.vpcc_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.vpcc_ctx_destroy.argtypes = [vp]
    lib.vpcc_ctx_stream.argtypes = [vp]
    lib.vpcc_ctx_stream.restype = vp
    lib.vpcc_ctx_destroy.restype = None
    lib.vpcc_last_error.argtypes = [vp]
    lib.vpcc_last_error.restype = C.c_char_p
    lib.vpcc_frame_validate.argtypes = [FD]
    lib.vpcc_frame_capacity_bound.argtypes = [FD]
    lib.vpcc_frame_capacity_bound.restype = u64
    lib.vpcc_generate_block_to_patch.argtypes = [vp, FD, C.c_int, vp]
    lib.vpcc_upsample_occupancy.argtypes = [vp, FD, C.c_int, vp]
    lib.vpcc_reconstruct_frame.argtypes = [vp, FD, C.c_int, vp, vp, vp, sz, C.POINTER(sz)]
    lib.vpcc_gof_create.argtypes = [vp, FD, u32, C.c_int, u64, u32, C.POINTER(vp)]
    lib.vpcc_gof_destroy.argtypes = [vp]
    lib.vpcc_gof_destroy.restype = None
    lib.vpcc_gof_reconstruct.argtypes = [vp, u32, u32, vp]
    lib.vpcc_gof_sync.argtypes = [vp]
    lib.vpcc_gof_point_counts.argtypes = [vp, vp]
    lib.vpcc_gof_device_outputs.argtypes = [vp, u32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    lib.vpcc_gof_download.argtypes = [vp, u32, vp, vp, vp, sz, C.POINTER(sz)]
    lib.vpcc_gof_block_to_patch.argtypes = [vp, u32, vp, C.POINTER(u32)]
    lib.vpcc_gof_download_async.argtypes = [vp, u32, vp, vp, vp, sz, C.POINTER(sz)]
    lib.vpcc_gof_download_wait.argtypes = [vp, u32]
    lib.vpcc_gof_frame_status.argtypes = [vp, u32]
    lib.vpcc_gof_kernel_times.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.c_int]
    lib.vpcc_gof_profile_interval.argtypes = [vp, u32]
    lib.vpcc_decoder_set_smoothing.argtypes = [vp, C.c_int, C.c_int, C.POINTER(SmoothingParams)]
    lib.vpcc_ctx_reserve.argtypes = [vp, C.c_uint64, C.POINTER(PoolInfo)]
    lib.vpcc_ctx_reserve_within.argtypes = [vp, C.c_uint64, C.c_float, C.POINTER(PoolInfo)]
    lib.vpcc_release_kept_pools.argtypes = [C.c_int]
    lib.vpcc_ctx_pool_alloc.argtypes = [vp, C.c_int, sz, C.POINTER(vp)]
    lib.vpcc_ctx_pool_free.argtypes = [vp, vp]
    lib.vpcc_ctx_pool_info.argtypes = [vp, C.POINTER(PoolInfo)]
    lib.vpcc_gof_kernel_time_means.argtypes = [vp, u32, C.POINTER(C.c_char_p), C.POINTER(C.c_float),
                                               C.POINTER(u32), C.c_int]
    lib.vpcc_gof_algorithmic_bytes.argtypes = [vp, u32, C.POINTER(u64)]
    lib.vpcc_gof_smooth.argtypes = [vp, u32, u32, C.POINTER(SmoothingParams), vp]
    lib.vpcc_decoder_open.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
    lib.vpcc_decoder_open_v3c.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, u32, C.POINTER(C.c_int), C.c_int,
                                          C.POINTER(vp)]
    lib.vpcc_decoder_start.argtypes = [vp]
    lib.vpcc_decoder_first_frame_seconds.argtypes = [vp]
    lib.vpcc_decoder_first_frame_seconds.restype = C.c_double
    lib.vpcc_decoder_recv_frame.argtypes = [vp, C.POINTER(sz), C.POINTER(vp), C.POINTER(vp)]
    lib.vpcc_decoder_error.argtypes = [vp]
    lib.vpcc_decoder_error.restype = C.c_char_p
    lib.vpcc_decoder_drain.argtypes = [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(C.c_double)]
    lib.vpcc_host_pin.argtypes = [vp, vp, sz]
    lib.vpcc_host_unpin.argtypes = [vp, vp]
    lib.vpcc_decoder_stats.argtypes = [vp, C.POINTER(DecoderStats)]
    lib.vpcc_ctx_bind_thread.argtypes = [vp, C.POINTER(C.c_int)]
    lib.vpcc_decoder_close.argtypes = [vp]
    lib.vpcc_decoder_close.restype = None
    lib.vpcc_write_ply.argtypes = [C.c_char_p, vp, vp, sz]
    lib.vpcc_write_ply_format.argtypes = [C.c_char_p, vp, vp, sz, C.c_int]
    lib.vpcc_v3c_open.argtypes = [C.c_char_p, sz, C.POINTER(vp)]
    lib.vpcc_v3c_close.argtypes = [vp]
    lib.vpcc_v3c_close.restype = None
    lib.vpcc_v3c_error.argtypes = [vp]
    lib.vpcc_v3c_error.restype = C.c_char_p
    lib.vpcc_v3c_unit_count.argtypes = [vp]
    lib.vpcc_v3c_unit_count.restype = u32
    lib.vpcc_v3c_next_gof.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(V3cGofInfo)]
    lib.vpcc_v3c_frame_patches.argtypes = [vp, u32, C.POINTER(Patch), u32, C.POINTER(u32), C.POINTER(u32)]
    lib.vpcc_v3c_video.argtypes = [vp, C.c_int, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(sz)]
    _lib = lib
    return lib
