"""Known-answer tests that PIN the CPU oracle (oracle/vpcc_oracle.c).

The reference (benclmnt/tmc2-rs) holds no golden vectors for the reconstruction
path ("parity unpinned"), so each expected value below is derived
by hand from the reference formula cited beside it, and the whole-frame results
are cross-checked against a second, independently written pure-Python
restatement (tests/pyref.py).
"""
import numpy as np
import pytest

import oracle_binding as ob
import pyref
from cases import _patch, _tiny_frame
from tmc2rs import synth
from tmc2rs._abi import PATCH_DTYPE

# ---- colour: convert_yuv10_to_rgb8, src/codec.rs:661-687 --------------------
COLOR_KATS = [
    ((512, 512, 512), (127, 127, 127)),
    ((0, 512, 512), (0, 0, 0)),
    ((1023, 512, 512), (255, 255, 255)),
    ((64, 512, 512), (15, 15, 15)),
    ((940, 512, 512), (234, 234, 234)),
    ((4, 512, 512), (0, 0, 0)),          # 4/1023*255 = 0.997 -> 0
    ((5, 512, 512), (1, 1, 1)),          # 5/1023*255 = 1.246 -> 1
    ((512, 512, 1023), (255, 67, 127)),
    ((512, 1023, 512), (127, 103, 255)),
    ((512, 0, 0), (0, 211, 0)),
    ((700, 300, 800), (255, 150, 76)),
    ((1023, 1023, 1023), (255, 171, 255)),
    ((0, 0, 0), (0, 83, 0)),
    # y*255/1023 is an exact integer: the IEEE rounding of y/1023 then *255 decides the floor
    ((341, 512, 512), (85, 85, 85)),
    ((682, 512, 512), (170, 170, 170)),
    # out-of-10-bit inputs (u16 domain): clamp paths
    ((65535, 512, 512), (255, 255, 255)),
    ((0, 65535, 65535), (255, 0, 255)),
]


@pytest.mark.parametrize("yuv,rgb", COLOR_KATS)
def test_color_kat(yuv, rgb):
    assert ob.yuv_to_rgb(*yuv) == rgb
    assert pyref.yuv10_to_rgb8(*yuv) == rgb


def test_color_exhaustive_gray_and_sampled_grid_vs_python():
    # every gray level, and a coarse (Y,U,V) lattice, against the Python f64 restatement
    for y in range(1024):
        assert ob.yuv_to_rgb(y, 512, 512) == pyref.yuv10_to_rgb8(y, 512, 512)
    for y in range(0, 1024, 93):
        for u in range(0, 1024, 89):
            for v in range(0, 1024, 97):
                assert ob.yuv_to_rgb(y, u, v) == pyref.yuv10_to_rgb8(y, u, v)


# ---- back-projection: Patch::generate_point, src/decoder.rs:871-888 ----------
def test_generate_point_mode0():
    p = ob.make_patch(normal_axis=0, tangent_axis=2, bitangent_axis=1, projection_mode=0, u1=10, v1=20, d1=100)
    # raw geo sample 28 -> depth 7 (codec.rs:534); x = 7+100, z = 3+10, y = 5+20
    assert ob.generate_point(p, 3, 5, 28 // 4) == (107, 25, 13)


def test_generate_point_mode1():
    p = ob.make_patch(normal_axis=0, tangent_axis=2, bitangent_axis=1, projection_mode=1, u1=10, v1=20, d1=1024)
    assert ob.generate_point(p, 3, 5, 7) == (1017, 25, 13)
    p.d1 = 3                                 # max(3,7)-7 = 0
    assert ob.generate_point(p, 3, 5, 7) == (0, 25, 13)


def test_generate_point_axes_table():
    # set_view_id, src/decoder.rs:790-796
    for (n, t, b, mode) in synth.VIEW_AXES:
        p = ob.make_patch(normal_axis=n, tangent_axis=t, bitangent_axis=b, projection_mode=mode, u1=1, v1=2, d1=50)
        pt = ob.generate_point(p, 10, 20, 5)
        assert pt[t] == 11 and pt[b] == 22
        assert pt[n] == (55 if mode == 0 else 45)


def test_generate_point_u16_truncation():
    # `as u16` truncation of usize sums (decoder.rs:874-876)
    p = ob.make_patch(normal_axis=0, tangent_axis=2, bitangent_axis=1, projection_mode=0,
                      u1=65530, v1=70000, d1=65536 + 9)
    assert ob.generate_point(p, 10, 1, 1) == ((65536 + 9 + 1) & 0xFFFF, (70000 + 1) & 0xFFFF, (65530 + 10) & 0xFFFF)


# ---- orientation: patch_to_canvas_helper, src/decoder.rs:853-867 -------------
def test_patch_to_canvas_all_orientations():
    su, sv, U0, V0, R = 3, 5, 7, 2, 16
    for o in range(9):
        p = ob.make_patch(u0=U0, v0=V0, size_u0=su, size_v0=sv, orientation=o)
        rec = np.zeros((), dtype=PATCH_DTYPE)
        rec["u0"], rec["v0"], rec["size_u0"], rec["size_v0"], rec["orientation"] = U0, V0, su, sv, o
        for res in (1, R):
            for (u, v) in [(0, 0), (1, 0), (0, 1), (2, 4), (su * res - 1, sv * res - 1)]:
                assert ob.patch_to_canvas(p, u, v, res) == pyref.patch_to_canvas_helper(rec, u, v, res)
    # spot values: Default and Swap at pixel resolution
    p = ob.make_patch(u0=2, v0=3, size_u0=4, size_v0=5, orientation=0)
    assert ob.patch_to_canvas(p, 5, 7, 16) == (5 + 32, 7 + 48)
    p.orientation = 1
    assert ob.patch_to_canvas(p, 5, 7, 16) == (7 + 32, 5 + 48)
    # Rot180 keeps size_uv0 in BLOCKS even at pixel resolution (reference quirk)
    p.orientation = 3
    assert ob.patch_to_canvas(p, 1, 2, 16) == (4 - 1 - 1 + 32, 5 - 1 - 2 + 48)
    # wrapping (release profile): negative -> huge, which the caller's assert then rejects
    assert ob.patch_to_canvas(ob.make_patch(u0=0, v0=0, size_u0=1, size_v0=1, orientation=3), 5, 0, 16)[0] == (1 - 1 - 5) % (1 << 64)


# ---- tiny frames built by hand (tests/cases.py) ------------------------------


def test_duplicate_removal_and_depth_div4():
    # geo D0 sample 28 and D1 sample 31 both give depth 7 -> ONE point (codec.rs:534,548,422-427)
    occ = np.zeros((8, 8), np.uint8)
    occ[0, 0] = 1                                     # occupies canvas pixels (0..3, 0..3)
    g0 = np.full((32, 32), 28, np.uint16)
    g1 = np.full((32, 32), 31, np.uint16)
    g1[1, 2] = 32                                     # depth 8 at pixel (x=2,y=1) -> two points there
    f = _tiny_frame([_patch(0, 0, 1, 1, view=0, u1=10, v1=20, d1=100)], occ, geo0=g0, geo1=g1)
    st, r = ob.reconstruct(f)
    assert st == 0
    assert r["n"] == 16 + 1
    xyz = ob.xyz_array(r)
    # order: v1 outer, u1 inner; pixel (u=2,v=1) is the 7th visited -> indices 6 (D0), 7 (D1)
    assert tuple(xyz[0]) == (107, 20, 10)
    assert tuple(xyz[6]) == (107, 21, 12) and tuple(xyz[7]) == (108, 21, 12)
    assert list(r["point_to_pixel"][6]) == [2, 1, 0] and list(r["point_to_pixel"][7]) == [2, 1, 1]


def test_ownership_later_patch_wins():
    # two patches whose bounding boxes overlap on an occupied block: higher index owns it (codec.rs:242-244)
    occ = np.ones((8, 8), np.uint8)
    f = _tiny_frame([_patch(0, 0, 2, 2, view=0), _patch(1, 1, 1, 1, view=1)], occ)
    st, b2p = ob.block_to_patch(f)
    assert st == 0
    assert list(b2p) == [1, 1, 1, 2]
    st, r = ob.reconstruct(f)
    # patch 0 emits blocks 0,1,2 ; patch 1 emits block 3; the partition vector is sorted by patch
    assert list(np.unique(r["partition"])) == [0, 1]
    assert np.all(np.diff(r["partition"].astype(np.int64)) >= 0)
    assert np.all(r["point_to_pixel"][r["partition"] == 1][:, :2] >= 16)


def test_ownership_ignores_whose_pixels():
    # a patch claims a block if ANY occupancy lies in it (codec.rs:226-244): unoccupied block stays 0
    occ = np.zeros((8, 8), np.uint8)
    occ[5, 6] = 200                                    # only block (1,1) has occupancy; raw value kept
    f = _tiny_frame([_patch(0, 0, 2, 2)], occ)
    st, b2p = ob.block_to_patch(f)
    assert list(b2p) == [0, 0, 0, 1]
    st, r = ob.reconstruct(f)
    assert r["occupancy_map"][20, 24] == 200 and r["occupancy_map"][19, 24] == 0


def test_chroma_nearest_neighbour():
    # pixel (x,y)=(5,7) reads U/V at (2,3) of the half-size plane (decoder.rs:977)
    occ = np.zeros((8, 8), np.uint8)
    occ[1, 1] = 1                                      # canvas pixels x 4..7, y 4..7
    f = _tiny_frame([_patch(0, 0, 1, 1)], occ)
    st, r = ob.reconstruct(f)
    p2p = r["point_to_pixel"]
    k = [i for i in range(r["n"]) if tuple(p2p[i][:2]) == (5, 7)]
    assert k
    for i in k:
        layer = p2p[i][2]
        y, u, v = f["attribute"][layer]
        assert list(r["colors16"][i]) == [y[7, 5], u[3, 2], v[3, 2]]


def test_emission_order_default_and_swap():
    occ = np.ones((8, 16), np.uint8)
    f = _tiny_frame([_patch(0, 0, 2, 1, orient=0), _patch(2, 0, 1, 2, orient=1)], occ, W=64, H=32)
    st, r = ob.reconstruct(f)
    assert st == 0
    p2p = r["point_to_pixel"]
    d0 = p2p[p2p[:, 2] == 0]
    # Default patch: block (0,0) row-major, then block (1,0)
    assert [tuple(t[:2]) for t in d0[:3]] == [(0, 0), (1, 0), (2, 0)]
    assert tuple(d0[16][:2]) == (0, 1) and tuple(d0[256][:2]) == (16, 0)
    # Swap patch (size_u0=1,size_v0=2): u runs along canvas y; blocks (v0=0) then (v0=1) sit side by side in x
    s = d0[512:]
    assert [tuple(t[:2]) for t in s[:3]] == [(32, 0), (32, 1), (32, 2)]
    assert tuple(s[16][:2]) == (33, 0) and tuple(s[256][:2]) == (48, 0)


def test_relative_d1_wrapping():
    occ = np.ones((8, 8), np.uint8)
    g0 = np.full((32, 32), 40, np.uint16)              # depth 10
    g1 = np.full((32, 32), 12, np.uint16)              # d1 = 3
    f = _tiny_frame([_patch(0, 0, 1, 1, view=0, d1=5), _patch(1, 0, 1, 1, view=3, d1=12)], occ, geo0=g0, geo1=g1)
    f["absolute_d1"] = 0
    st, r = ob.reconstruct(f)
    xyz = ob.xyz_array(r)
    part = r["partition"]
    assert tuple(xyz[part == 0][0]) == (15, 0, 0) and tuple(xyz[part == 0][1]) == (18, 0, 0)   # += d1 (mode 0)
    # mode 1: point0 = max(12,10)-10 = 2; point1 = 2 - 3 wraps to 65535 (u16, release profile)
    assert tuple(xyz[part == 1][0]) == (2, 0, 0) and tuple(xyz[part == 1][1]) == (65535, 0, 0)


def test_panics_become_status_codes():
    occ = np.ones((8, 8), np.uint8)
    # patch sticks out of the canvas -> assert at decoder.rs:835
    st, _ = ob.reconstruct(_tiny_frame([_patch(1, 1, 2, 1)], occ))
    assert st == 3
    # Rot180 at pixel resolution underflows (size in blocks) -> assert at decoder.rs:848
    st, _ = ob.reconstruct(_tiny_frame([_patch(0, 0, 1, 1, orient=3)], occ))
    assert st == 3
    # axis_of_additional_plane != 0 -> unimplemented!() codec.rs:437
    p = _patch(0, 0, 1, 1)
    p["axis_of_additional_plane"] = 1
    st, _ = ob.reconstruct(_tiny_frame([p], occ))
    assert st == 2
    # missing D1 geometry frame -> generate_point_cloud returns None (codec.rs:318-320)
    f = _tiny_frame([_patch(0, 0, 1, 1)], occ)
    f["geometry"] = [f["geometry"][0]]
    st, _ = ob.reconstruct(f)
    assert st == 4


def test_empty_inputs():
    # no patches / no occupancy: zero points, block_to_patch all zero
    st, r = ob.reconstruct(_tiny_frame([], np.ones((8, 8), np.uint8)))
    assert st == 0 and r["n"] == 0 and not r["block_to_patch"].any()
    st, r = ob.reconstruct(_tiny_frame([_patch(0, 0, 2, 2)], np.zeros((8, 8), np.uint8)))
    assert st == 0 and r["n"] == 0 and not r["block_to_patch"].any()


def test_occupancy_upsample_is_nearest_neighbour():
    f = synth.small_frame(3, occupancy_values="random")
    st, r = ob.reconstruct(f)
    assert st == 0
    up = np.repeat(np.repeat(f["occupancy"], 4, axis=0), 4, axis=1)
    assert np.array_equal(r["occupancy_map"], up)


# ---- whole frames: C oracle vs the independent pure-Python restatement -------
def _compare_with_pyref(f):
    st, r = ob.reconstruct(f)
    assert st == 0
    pr = pyref.reconstruct(f)
    assert r["n"] == len(pr["positions"])
    assert [tuple(t) for t in ob.xyz_array(r)] == pr["positions"]
    assert [tuple(t) for t in ob.rgb_array(r)] == pr["colors"]
    assert [tuple(t) for t in r["colors16"]] == pr["colors16"]
    assert list(r["partition"]) == pr["partition"]
    assert [tuple(t) for t in r["point_to_pixel"]] == pr["point_to_pixel"]
    assert list(r["block_to_patch"]) == pr["block_to_patch"]
    return r


@pytest.mark.parametrize("index", [0, 1, 2])
def test_small_synthetic_frames_vs_pyref(index):
    r = _compare_with_pyref(synth.small_frame(index, occupancy_values="random" if index else "one"))
    assert r["n"] > 0


def test_ragged_frame_vs_pyref():
    # canvas not a multiple of the block size, precision 2, block size 8
    f = synth.make_frame(72, 40, 2, 8, seed=77, max_side=3, cover_target=0.7)
    _compare_with_pyref(f)


def test_exotic_orientations_vs_pyref():
    # orientations other than Default/Swap use size_uv0 in blocks at pixel resolution (reference quirk);
    # with a generous origin they stay inside the canvas and both restatements must agree on the result
    occ = (np.arange(16 * 16).reshape(16, 16) % 3 != 0).astype(np.uint8)
    patches = [_patch(0, 0, 2, 1, orient=0, view=0, u1=3, v1=1, d1=7),
               _patch(0, 1, 2, 1, orient=8, view=1, u1=5, v1=2, d1=9),     # MRot270 == Swap
               _patch(2, 3, 2, 1, orient=7, view=2, u1=7, v1=3, d1=11),    # MRot180: y = sv-1-v+v0*R
               _patch(2, 1, 2, 1, orient=3, view=4, u1=9, v1=4, d1=600),   # Rot180
               _patch(1, 2, 1, 1, orient=1, view=5, u1=0, v1=0, d1=700)]
    f = _tiny_frame(patches, occ, W=64, H=64)
    _compare_with_pyref(f)
