"""Hand-built tiny frames and the named parity cases shared by the oracle tests
(CPU) and the HIP parity tests (GPU)."""
import numpy as np

from tmc2rs import synth
from tmc2rs._abi import PATCH_DTYPE


def _tiny_frame(patches, occ, W=32, H=32, R=16, prec=4, geo0=None, geo1=None, seed=1):
    rng = np.random.RandomState(seed)
    g0 = geo0 if geo0 is not None else (rng.randint(0, 800, size=(H, W))).astype(np.uint16)
    g1 = geo1 if geo1 is not None else (g0 + 4 * rng.randint(0, 4, size=(H, W))).astype(np.uint16)
    attr = []
    for _ in range(2):
        attr.append((rng.randint(64, 941, size=(H, W)).astype(np.uint16),
                     rng.randint(64, 961, size=(H // 2, W // 2)).astype(np.uint16),
                     rng.randint(64, 961, size=(H // 2, W // 2)).astype(np.uint16)))
    return {"width": W, "height": H, "occupancy_resolution": R, "occupancy_precision": prec,
            "map_count": 2, "absolute_d1": 1, "attribute_count": 1, "flags": 0,
            "patches": np.array(patches, dtype=PATCH_DTYPE), "occupancy": occ.astype(np.uint8),
            "geometry": [g0, g1], "attribute": attr}


def _patch(u0, v0, su, sv, view=0, orient=0, u1=0, v1=0, d1=0):
    n, t, b, mode = synth.VIEW_AXES[view]
    r = np.zeros((), dtype=PATCH_DTYPE)
    r["u0"], r["v0"], r["size_u0"], r["size_v0"] = u0, v0, su, sv
    r["u1"], r["v1"], r["d1"], r["lod_x"], r["lod_y"] = u1, v1, d1, 1, 1
    r["normal_axis"], r["tangent_axis"], r["bitangent_axis"], r["projection_mode"] = n, t, b, mode
    r["orientation"] = orient
    return r



def exotic_frame():
    """Orientations other than Default/Swap (reference quirk: size_uv0 stays in blocks at pixel
    resolution), placed so that every mapped pixel stays inside the canvas."""
    occ = (np.arange(16 * 16).reshape(16, 16) % 3 != 0).astype(np.uint8)
    patches = [_patch(0, 0, 2, 1, orient=0, view=0, u1=3, v1=1, d1=7),
               _patch(0, 1, 2, 1, orient=8, view=1, u1=5, v1=2, d1=9),     # MRot270 == Swap
               _patch(2, 3, 2, 1, orient=7, view=2, u1=7, v1=3, d1=11),    # MRot180: y = sv-1-v+v0*R
               _patch(2, 1, 2, 1, orient=3, view=4, u1=9, v1=4, d1=600),   # Rot180
               _patch(1, 2, 1, 1, orient=1, view=5, u1=0, v1=0, d1=700)]
    return _tiny_frame(patches, occ, W=64, H=64)


def relative_d1_frame():
    occ = np.ones((8, 8), np.uint8)
    g0 = np.full((32, 32), 40, np.uint16)
    g1 = np.full((32, 32), 12, np.uint16)
    f = _tiny_frame([_patch(0, 0, 1, 1, view=0, d1=5), _patch(1, 0, 1, 1, view=3, d1=12)], occ, geo0=g0, geo1=g1)
    f["absolute_d1"] = 0
    return f


def overlap_frame():
    occ = np.ones((8, 8), np.uint8)
    return _tiny_frame([_patch(0, 0, 2, 2, view=0), _patch(1, 1, 1, 1, view=1), _patch(0, 1, 1, 1, view=4)], occ)


def truncation_frame():
    """u16 truncation of coordinates and degenerate (coinciding) axes."""
    occ = np.ones((8, 8), np.uint8)
    a = _patch(0, 0, 1, 1, view=0, u1=65530, v1=70000, d1=65536 + 9)
    b = _patch(1, 0, 1, 1, view=1, u1=1, v1=2, d1=3)
    b["normal_axis"], b["tangent_axis"], b["bitangent_axis"] = 1, 1, 0     # tangent overwrites normal
    c = _patch(0, 1, 1, 1, view=5, u1=4, v1=5, d1=2)                        # mode 1, d1 < depth -> 0
    return _tiny_frame([a, b, c], occ)


def block8_frame():
    """Block size 8, precision 2, canvas not a multiple of the block size (ragged)."""
    return synth.make_frame(72, 40, 2, 8, seed=77, max_side=3, cover_target=0.7, size_skew=1.0)


def block32_frame():
    """Block size 32 (> 256 pixels per block: several raster chunks per virtual block), precision 1."""
    return synth.make_frame(128, 96, 1, 32, seed=78, max_side=2, cover_target=0.9, size_skew=1.0,
                            occupancy_values="random")


def single_map_frame():
    f = synth.small_frame(5)
    f["map_count"] = 1
    f["geometry"] = [f["geometry"][0]]
    f["attribute"] = [f["attribute"][0]]
    return f


def no_attribute_frame():
    f = synth.small_frame(6)
    f["attribute_count"] = 0
    f["attribute"] = []
    return f


def strided_frame():
    """Planes with stride > width (explicit strides; the reference would need stride == width)."""
    f = synth.small_frame(7, width=96, height=80)

    def pad(a, extra):
        buf = np.full((a.shape[0], a.shape[1] + extra), 0xEE if a.dtype == np.uint8 else 0xEEEE, dtype=a.dtype)
        buf[:, :a.shape[1]] = a
        return buf[:, :a.shape[1]]

    f["occupancy"] = pad(f["occupancy"], 5)
    f["geometry"] = [pad(g, 7) for g in f["geometry"]]
    f["attribute"] = [(pad(y, 3), pad(u, 9), pad(v, 9)) for (y, u, v) in f["attribute"]]
    return f


def medium_frame(index=0, **kw):
    return synth.make_frame(320, 256, 4, 16, seed=0x3ED10000 + index, max_side=6, cover_target=0.6,
                            size_skew=2.0, overlap_prob=0.3, **kw)


def gray_boundary_frame():
    """Grey pixels (U = V = 512): c*255/1023 is an exact integer whenever luma is a multiple of 341,
    which is where the IEEE rounding of the reference's division decides the floor — this forces the
    exact-division branch of the fast colour path (rare on random data)."""
    f = medium_frame(3)
    h, w = f["geometry"][0].shape
    idx = np.arange(h * w, dtype=np.int64).reshape(h, w)
    y0 = (idx % 1024).astype(np.uint16)
    y1 = (np.array([0, 341, 682, 1023, 340, 342, 681, 683], dtype=np.uint16))[idx % 8]
    c = np.full((h // 2, w // 2), 512, dtype=np.uint16)
    f["attribute"] = [(y0, c, c.copy()), (y1, c.copy(), c.copy())]
    return f


def overlapping_3d_frame(index=0, base=None):
    """Patches whose 3-D footprints overlap (all placed around (100..300)^3): many grid cells hold
    points of several patches, which is what the smoothing filters act on."""
    f = base if base is not None else medium_frame(20 + index)
    p = f["patches"].copy()
    p["u1"] = 100 + (np.arange(len(p)) % 5) * 3
    p["v1"] = 100 + (np.arange(len(p)) % 7) * 2
    p["d1"] = np.where(p["projection_mode"] == 0, 100, 300)
    f["patches"] = p
    return f


def wide_samples_frame():
    """Samples that use all 16 bits: depths up to 16383 after the /4, attribute samples above 10 bits (the
    colour conversion then runs the reference formula itself), coordinates that wrap `as u16`."""
    f = medium_frame(7)
    rng = np.random.default_rng(99)
    f["geometry"] = [rng.integers(0, 65536, g.shape, dtype=np.uint16) for g in f["geometry"]]
    f["attribute"] = [tuple(rng.integers(0, 65536, pl.shape, dtype=np.uint16) for pl in layer) for layer in f["attribute"]]
    return f


def precision_frame(prec, seed):
    """Block size 16 (tile kernel) with occupancy precision 1, 2, 8 or 16."""
    return synth.make_frame(160, 128, prec, 16, seed=seed, max_side=4, cover_target=0.7, size_skew=1.5,
                            occupancy_values="random")


PARITY_CASES = {
    "small0": lambda: synth.small_frame(0),
    "small1_randocc": lambda: synth.small_frame(1, occupancy_values="random"),
    "small2_wide": lambda: synth.small_frame(2, width=128, height=96),
    "exotic_orientations": exotic_frame,
    "relative_d1": relative_d1_frame,
    "overlap": overlap_frame,
    "truncation_degenerate_axes": truncation_frame,
    "block8_ragged": block8_frame,
    "block32_multichunk": block32_frame,
    "single_map_extension": single_map_frame,
    "no_attribute": no_attribute_frame,
    "strided_planes": strided_frame,
    "gray_exact_boundaries": gray_boundary_frame,
    "medium0": lambda: medium_frame(0),
    "medium1_randocc": lambda: medium_frame(1, occupancy_values="random"),
    "wide_samples": wide_samples_frame,
    "precision1_block16": lambda: precision_frame(1, 201),
    "precision2_block16": lambda: precision_frame(2, 202),
    "precision8_block16": lambda: precision_frame(8, 203),
    "precision16_block16": lambda: precision_frame(16, 204),
    "empty_no_patches": lambda: _tiny_frame([], np.ones((8, 8), np.uint8)),
    "empty_no_occupancy": lambda: _tiny_frame([_patch(0, 0, 2, 2)], np.zeros((8, 8), np.uint8)),
}


def random_sweep_frames(n=40):
    """Seeded sweep over canvas sizes, precisions, occupancy value styles, patch statistics and map counts (block size
    16): the frames of test_parity_gpu.py::test_random_sweep_against_oracle and of the oracle-vs-pyref CPU test."""
    rng = np.random.default_rng(20261003)
    frames = []
    for i in range(n):
        prec = int(rng.choice([1, 2, 4, 4, 4, 8]))
        w = 16 * int(rng.integers(2, 26)) * (2 if prec == 8 else 1)
        h = 16 * int(rng.integers(2, 20)) * (2 if prec == 8 else 1)
        f = synth.make_frame(w, h, prec, 16, seed=0xABC000 + i, max_side=int(rng.integers(2, 9)),
                             cover_target=float(rng.uniform(0.2, 0.95)), size_skew=float(rng.uniform(0.7, 4.0)),
                             swap_prob=float(rng.uniform(0, 1)), overlap_prob=float(rng.uniform(0, 0.5)),
                             dup_prob=float(rng.uniform(0, 0.6)), ellipse_scale=float(rng.uniform(0.5, 1.3)),
                             occupancy_values="random" if i % 3 == 0 else "one")
        if i % 7 == 3:                                     # full-range samples
            f["attribute"] = [tuple(rng.integers(0, 65536, pl.shape, dtype=np.uint16) for pl in layer) for layer in f["attribute"]]
        if i % 11 == 5:
            f["absolute_d1"] = 0
        frames.append(f)
    return frames
