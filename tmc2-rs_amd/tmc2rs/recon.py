"""Thin Python binding of the C ABI (include/vpcc_recon.h) — used by tests/ and bench.py.

Everything here calls into libvpcc_recon.so (HIP kernels + C++ host runtime).
There is no Python implementation of the reconstruction: a missing library or a
missing GPU raises.
"""
import ctypes as C
import weakref

import numpy as np

from . import _abi
from ._abi import (COLOR3_DTYPE, POINT3_DTYPE, VPCC_GOF_FORCE_GENERAL, VPCC_GOF_PROFILE,
                   VPCC_GOF_WANT_PATCH_INDEX, VPCC_MEM_DEVICE, VPCC_MEM_HOST, FrameDesc, host_frame_desc)


class VpccError(RuntimeError):
    def __init__(self, status, where, detail=""):
        self.status = status
        lib = _abi.load_library()
        msg = lib.vpcc_status_string(status).decode()
        super().__init__(f"{where}: status {status} ({msg}) {detail}")


def validate_frame(frame):
    """vpcc_frame_validate on a frame dict — pure host code, no GPU needed."""
    lib = _abi.load_library()
    desc, keep = host_frame_desc(frame)
    return lib.vpcc_frame_validate(C.byref(desc))


class Context:
    """vpcc_ctx: one per GPU / worker thread."""

    def __init__(self, device=0):
        self.lib = _abi.load_library()
        self.h = C.c_void_p()
        st = self.lib.vpcc_ctx_create(int(device), C.byref(self.h))
        if st:
            raise VpccError(st, "vpcc_ctx_create")
        self.device = device
        self._gofs = weakref.WeakSet()          # a vpcc_gof must not outlive its context

    def close(self):
        if self.h:
            for g in list(self._gofs):
                g.close()
            self.lib.vpcc_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stream(self):
        """The compute stream's hipStream_t as an integer (e.g. for torch.cuda.ExternalStream)."""
        return self.lib.vpcc_ctx_stream(self.h)

    @staticmethod
    def _pool_dict(p):
        return {"GiB": p.bytes >> 30, "granules": int(p.granules), "kinds": int(p.kinds),
                "GiB_of_kind": [int(p.bytes_of_kind[0] >> 30), int(p.bytes_of_kind[1] >> 30)],
                "in_use_MB": [int(p.in_use[0] >> 20), int(p.in_use[1] >> 20)],
                "probe_GBps_same_kind": round(p.probe_gbps_same, 0), "probe_GBps_two_kinds": round(p.probe_gbps_other, 0),
                "ms_spent": round(p.ms_spent, 1), "blocks_in_other_home": int(p.other_home), "blocks_outside_pool": int(p.fallbacks),
                "taken_over_from_an_earlier_context": bool(p.reused)}

    def reserve(self, gib, budget_ms=0.0):
        """vpcc_ctx_reserve(_within): one allocation of `gib` GiB, classified by kind of VRAM region; every later gof of
        this context keeps its big blocks in its two homes.  budget_ms > 0: no search for a second home once that much
        wall-clock time has passed."""
        p = _abi.PoolInfo()
        self._check(self.lib.vpcc_ctx_reserve_within(self.h, int(gib) << 30, float(budget_ms), C.byref(p)), "vpcc_ctx_reserve")
        return self._pool_dict(p)

    def pool_alloc(self, home, nbytes):
        """vpcc_ctx_pool_alloc: device memory of the pool's home `home` for a producer of device planes; returns the pointer."""
        p = C.c_void_p()
        self._check(self.lib.vpcc_ctx_pool_alloc(self.h, int(home), int(nbytes), C.byref(p)), "vpcc_ctx_pool_alloc")
        return p.value

    def pool_free(self, ptr):
        self._check(self.lib.vpcc_ctx_pool_free(self.h, C.c_void_p(ptr)), "vpcc_ctx_pool_free")

    def pool_info(self):
        p = _abi.PoolInfo()
        self._check(self.lib.vpcc_ctx_pool_info(self.h, C.byref(p)), "vpcc_ctx_pool_info")
        return self._pool_dict(p)

    def _check(self, st, where):
        if st:
            raise VpccError(st, where, self.lib.vpcc_last_error(self.h).decode())

    # ---- one-shot seam replacements -------------------------------------
    def generate_block_to_patch(self, frame):
        desc, keep = host_frame_desc(frame)
        R = desc.occupancy_resolution
        out = np.zeros((desc.width // R) * (desc.height // R), dtype=np.uint32)
        self._check(self.lib.vpcc_generate_block_to_patch(self.h, C.byref(desc), VPCC_MEM_HOST, out.ctypes.data),
                    "vpcc_generate_block_to_patch")
        return out

    def upsample_occupancy(self, frame):
        desc, keep = host_frame_desc(frame)
        out = np.zeros((desc.height, desc.width), dtype=np.uint8)
        self._check(self.lib.vpcc_upsample_occupancy(self.h, C.byref(desc), VPCC_MEM_HOST, out.ctypes.data),
                    "vpcc_upsample_occupancy")
        return out

    def reconstruct_frame(self, frame, capacity=None, want_patch_index=False):
        desc, keep = host_frame_desc(frame)
        cap = int(capacity if capacity is not None else self.lib.vpcc_frame_capacity_bound(C.byref(desc)))
        xyz = np.zeros(max(cap, 1), dtype=POINT3_DTYPE)
        rgb = np.zeros(max(cap, 1), dtype=COLOR3_DTYPE)
        pidx = np.zeros(max(cap, 1), dtype=np.uint16) if want_patch_index else None
        n = C.c_size_t(0)
        st = self.lib.vpcc_reconstruct_frame(self.h, C.byref(desc), VPCC_MEM_HOST, xyz.ctypes.data, rgb.ctypes.data,
                                             pidx.ctypes.data if pidx is not None else None, cap, C.byref(n))
        self._check(st, "vpcc_reconstruct_frame")
        k = n.value
        res = {"n": k, "xyz": _xyz(xyz[:k]), "rgb": _rgb(rgb[:k])}
        if pidx is not None:
            res["patch_index"] = pidx[:k].copy()
        return res

    def gof(self, frames, capacity=0, flags=0, memory=VPCC_MEM_HOST, descs=None):
        return Gof(self, frames, capacity, flags, memory, descs)


def _xyz(a):
    return np.stack([a["x"], a["y"], a["z"]], axis=1) if len(a) else np.zeros((0, 3), np.uint16)


def _rgb(a):
    return np.stack([a["r"], a["g"], a["b"]], axis=1) if len(a) else np.zeros((0, 3), np.uint8)


class Gof:
    """vpcc_gof: a batch of independent frames resident in HBM."""

    def __init__(self, ctx, frames, capacity=0, flags=0, memory=VPCC_MEM_HOST, descs=None):
        self.ctx, self.lib = ctx, ctx.lib
        self.n_frames = len(frames) if descs is None else len(descs)
        self._keep = []
        if descs is None:
            arr = (FrameDesc * self.n_frames)()
            for i, f in enumerate(frames):
                d, keep = host_frame_desc(f)
                arr[i] = d
                self._keep.append(keep)
        elif isinstance(descs, C.Array):                 # a prebuilt array (bench.py's fresh_gof leg: no conversion per gof)
            arr = descs
        else:
            arr = (FrameDesc * self.n_frames)(*descs)
        self._descs = arr
        self.h = C.c_void_p()
        st = self.lib.vpcc_gof_create(ctx.h, arr, self.n_frames, memory, int(capacity), int(flags), C.byref(self.h))
        ctx._check(st, "vpcc_gof_create")
        self.flags = flags
        ctx._gofs.add(self)

    def close(self):
        if self.h:
            self.lib.vpcc_gof_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reconstruct(self, first=0, count=None, stream=None):
        count = self.n_frames - first if count is None else count
        self.ctx._check(self.lib.vpcc_gof_reconstruct(self.h, first, count, C.c_void_p(stream) if stream else None),
                        "vpcc_gof_reconstruct")

    def sync(self):
        self.ctx._check(self.lib.vpcc_gof_sync(self.h), "vpcc_gof_sync")

    def point_counts(self):
        out = np.zeros(self.n_frames, dtype=np.uint32)
        self.ctx._check(self.lib.vpcc_gof_point_counts(self.h, out.ctypes.data), "vpcc_gof_point_counts")
        return out

    def frame_status(self, frame):
        return self.lib.vpcc_gof_frame_status(self.h, frame)

    def download(self, frame, want_patch_index=False):
        n = int(self.point_counts()[frame])
        xyz = np.zeros(max(n, 1), dtype=POINT3_DTYPE)
        rgb = np.zeros(max(n, 1), dtype=COLOR3_DTYPE)
        pidx = np.zeros(max(n, 1), dtype=np.uint16) if want_patch_index else None
        k = C.c_size_t(0)
        st = self.lib.vpcc_gof_download(self.h, frame, xyz.ctypes.data, rgb.ctypes.data,
                                        pidx.ctypes.data if pidx is not None else None, max(n, 1), C.byref(k))
        self.ctx._check(st, "vpcc_gof_download")
        res = {"n": k.value, "xyz": _xyz(xyz[:k.value]), "rgb": _rgb(rgb[:k.value])}
        if pidx is not None:
            res["patch_index"] = pidx[:k.value].copy()
        return res

    def smooth(self, bitdepth, grid_size=0, threshold=0, color_grid_size=0, color_threshold_smoothing=0,
               color_threshold_difference=0, first=0, count=None, stream=None):
        """vpcc_gof_smooth: geometry smoothing if grid_size > 0, colour smoothing if color_grid_size > 0."""
        p = _abi.SmoothingParams()
        p.geometry_bitdepth_3d = bitdepth
        p.flags = (_abi.VPCC_SMOOTH_GEOMETRY if grid_size else 0) | (_abi.VPCC_SMOOTH_COLOR if color_grid_size else 0)
        p.grid_size, p.threshold = grid_size, threshold
        p.color_grid_size = color_grid_size
        p.color_threshold_smoothing, p.color_threshold_difference = color_threshold_smoothing, color_threshold_difference
        count = self.n_frames - first if count is None else count
        self.ctx._check(self.lib.vpcc_gof_smooth(self.h, first, count, C.byref(p), C.c_void_p(stream) if stream else None),
                        "vpcc_gof_smooth")

    def block_to_patch(self, frame, n_blocks):
        """vpcc_gof_block_to_patch: (block_to_patch[n_blocks], work items of the single-pass kernel)."""
        out = np.zeros(n_blocks, dtype=np.uint32)
        items = C.c_uint32(0)
        self.ctx._check(self.lib.vpcc_gof_block_to_patch(self.h, frame, out.ctypes.data, C.byref(items)), "vpcc_gof_block_to_patch")
        return out, int(items.value)

    def device_outputs(self, frame):
        p = [C.c_void_p() for _ in range(4)]
        self.ctx._check(self.lib.vpcc_gof_device_outputs(self.h, frame, *[C.byref(x) for x in p]),
                        "vpcc_gof_device_outputs")
        return tuple(x.value for x in p)

    def kernel_times(self):
        names = (C.c_char_p * 16)()
        ms = (C.c_float * 16)()
        n = self.lib.vpcc_gof_kernel_times(self.h, names, ms, 16)
        return [(names[i].decode(), float(ms[i])) for i in range(n)]

    def profile_interval(self, every):
        """Profile mode: time only every `every`-th reconstruct."""
        self.ctx._check(self.lib.vpcc_gof_profile_interval(self.h, int(every)), "vpcc_gof_profile_interval")

    def kernel_time_means(self, last_n=0):
        """({kernel name: mean ms over the last `last_n` profiled launches}, launches averaged)."""
        names = (C.c_char_p * 16)()
        ms = (C.c_float * 16)()
        launches = C.c_uint32(0)
        n = self.lib.vpcc_gof_kernel_time_means(self.h, int(last_n), names, ms, C.byref(launches), 16)
        return {names[i].decode(): float(ms[i]) for i in range(n)}, launches.value

    def algorithmic_bytes(self, frame):
        b = C.c_uint64(0)
        self.ctx._check(self.lib.vpcc_gof_algorithmic_bytes(self.h, frame, C.byref(b)), "vpcc_gof_algorithmic_bytes")
        return b.value


class Decoder:
    """Binding of the C++ tmc2rs::Decoder (mirror of the reference's Decoder::new / start / recv_frame /
    Iterator, src/lib.rs:70-154).  Iterating yields dicts {n, xyz, rgb} in presentation order."""

    def __init__(self, path, devices=(0,), occupancy_yuv=None, geometry_yuv=None, attribute_yuv=None,
                 occupancy_precision=4):
        """`path`: a .vpccgof container, or — with the raw decoded videos given — a V3C sample stream (.bin)."""
        self.lib = _abi.load_library()
        self.h = C.c_void_p()
        dev = (C.c_int * len(devices))(*devices)
        if occupancy_yuv is None:
            st = self.lib.vpcc_decoder_open(str(path).encode(), dev, len(devices), C.byref(self.h))
        else:
            st = self.lib.vpcc_decoder_open_v3c(str(path).encode(), str(occupancy_yuv).encode(), str(geometry_yuv).encode(),
                                                str(attribute_yuv).encode() if attribute_yuv else None,
                                                occupancy_precision, dev, len(devices), C.byref(self.h))
        if st:
            raise VpccError(st, "vpcc_decoder_open")

    def set_smoothing(self, geometry=False, color=False, bitdepth=10, grid_size=0, threshold=0, color_grid_size=0,
                      color_threshold_smoothing=0, color_threshold_difference=0):
        """vpcc_decoder_set_smoothing: the reference's apply_geo_smoothing_type / apply_attr_smoothing_type switches;
        grid_size / threshold are used for inputs without a geometry-smoothing SEI only."""
        p = _abi.SmoothingParams()
        p.geometry_bitdepth_3d = bitdepth
        p.grid_size, p.threshold = grid_size, threshold
        p.color_grid_size = color_grid_size
        p.color_threshold_smoothing, p.color_threshold_difference = color_threshold_smoothing, color_threshold_difference
        st = self.lib.vpcc_decoder_set_smoothing(self.h, int(geometry), int(color), C.byref(p))
        if st:
            raise VpccError(st, "vpcc_decoder_set_smoothing", self.error())

    def start(self):
        st = self.lib.vpcc_decoder_start(self.h)
        if st:
            raise VpccError(st, "vpcc_decoder_start", self.error())

    def error(self):
        return self.lib.vpcc_decoder_error(self.h).decode()

    def recv_frame(self):
        n, px, pc = C.c_size_t(0), C.c_void_p(), C.c_void_p()
        if not self.lib.vpcc_decoder_recv_frame(self.h, C.byref(n), C.byref(px), C.byref(pc)):
            return None
        k = n.value
        xyz = np.zeros((k, 3), np.uint16)
        rgb = np.zeros((k, 3), np.uint8)
        if k:
            C.memmove(xyz.ctypes.data, px.value, k * 6)
            if pc.value:
                C.memmove(rgb.ctypes.data, pc.value, k * 3)
        return {"n": k, "xyz": xyz, "rgb": rgb if pc.value else None}

    def drain(self):
        """Consumes the rest of the stream inside the library; returns (frames, points, seconds)."""
        nf, npts, sec = C.c_uint64(0), C.c_uint64(0), C.c_double(0)
        st = self.lib.vpcc_decoder_drain(self.h, C.byref(nf), C.byref(npts), C.byref(sec))
        if st:
            raise VpccError(st, "vpcc_decoder_drain", self.error())
        return nf.value, npts.value, sec.value

    def first_frame_seconds(self):
        return self.lib.vpcc_decoder_first_frame_seconds(self.h)

    def stats(self):
        """Launches, frames per launch, kernel seconds, lane NUMA nodes (complete after end of stream)."""
        st = _abi.DecoderStats()
        rc = self.lib.vpcc_decoder_stats(self.h, C.byref(st))
        if rc:
            raise VpccError(rc, "vpcc_decoder_stats")
        return {"launches": st.launches, "frames": st.frames, "max_frames_per_launch": st.max_frames_per_launch,
                "lanes": st.lanes, "kernel_seconds": st.kernel_seconds, "launch_seconds": st.launch_seconds,
                "numa_node": list(st.numa_node)[:st.lanes]}

    def __iter__(self):
        return self

    def __next__(self):
        f = self.recv_frame()
        if f is None:
            raise StopIteration
        return f

    def close(self):
        if self.h:
            self.lib.vpcc_decoder_close(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def write_ply(path, xyz, rgb=None, binary=False):
    """writer::PlyWriter (ASCII, or binary little endian), through the C++ mirror."""
    lib = _abi.load_library()
    x = np.ascontiguousarray(xyz, dtype=np.uint16)
    c = np.ascontiguousarray(rgb, dtype=np.uint8) if rgb is not None else None
    st = lib.vpcc_write_ply_format(str(path).encode(), x.ctypes.data, c.ctypes.data if c is not None else None, len(x),
                                   1 if binary else 0)
    if st:
        raise VpccError(st, "vpcc_write_ply")


class V3cStream:
    """Binding of the V3C syntax parser (vpcc_v3c_*): iterate GOFs, read patch frames and video sub-bitstreams.
    Pure host code — no GPU needed."""

    def __init__(self, data):
        self.lib = _abi.load_library()
        self.h = C.c_void_p()
        self._data = bytes(data)
        st = self.lib.vpcc_v3c_open(self._data, len(self._data), C.byref(self.h))
        if st:
            raise VpccError(st, "vpcc_v3c_open")

    def unit_count(self):
        return self.lib.vpcc_v3c_unit_count(self.h)

    def next_gof(self):
        """Returns the vpcc_v3c_gof_info fields as a dict, or None at the end of the stream."""
        have, info = C.c_int(0), _abi.V3cGofInfo()
        st = self.lib.vpcc_v3c_next_gof(self.h, C.byref(have), C.byref(info))
        if st:
            raise VpccError(st, "vpcc_v3c_next_gof", self.lib.vpcc_v3c_error(self.h).decode())
        if not have.value:
            return None
        d = {n: getattr(info, n) for n, _ in _abi.V3cGofInfo._fields_ if n not in ("reserved", "video_bytes")}
        d["video_bytes"] = list(info.video_bytes)
        return d

    def frame_patches(self, frame):
        n, fi = C.c_uint32(0), C.c_uint32(0)
        st = self.lib.vpcc_v3c_frame_patches(self.h, frame, None, 0, C.byref(n), C.byref(fi))
        if st:
            raise VpccError(st, "vpcc_v3c_frame_patches")
        arr = (_abi.Patch * max(n.value, 1))()
        st = self.lib.vpcc_v3c_frame_patches(self.h, frame, arr, n.value, C.byref(n), C.byref(fi))
        if st:
            raise VpccError(st, "vpcc_v3c_frame_patches")
        return fi.value, [arr[i] for i in range(n.value)]

    def video(self, kind):
        p, n = C.POINTER(C.c_uint8)(), C.c_size_t(0)
        st = self.lib.vpcc_v3c_video(self.h, kind, C.byref(p), C.byref(n))
        if st:
            raise VpccError(st, "vpcc_v3c_video")
        return bytes(p[:n.value]) if n.value else b""

    def close(self):
        if self.h:
            self.lib.vpcc_v3c_close(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
