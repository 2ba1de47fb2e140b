"""Smoothing.  The reference has no implementation (unimplemented!()), so the behaviour is
this repository's own integer specification (oracle/vpcc_smoothing_spec.h).  CPU tests pin the spec on
hand-made cases; GPU tests require the HIP kernels to reproduce the spec bit for bit."""
import numpy as np
import pytest

import cases
import oracle_binding as ob
from tmc2rs import _abi, synth


# ---------------------------------------------------------------- the spec itself (CPU)
def test_single_patch_cells_are_left_alone():
    rng = np.random.RandomState(1)
    xyz = rng.randint(0, 1024, size=(5000, 3)).astype(np.uint16)
    patch = np.zeros(5000, np.uint16)                      # one patch everywhere: no cell is "mixed"
    assert np.array_equal(ob.spec_smooth_geometry(xyz, patch, 10, 8, 0), xyz)
    rgb = rng.randint(0, 256, size=(5000, 3)).astype(np.uint8)
    assert np.array_equal(ob.spec_smooth_color(xyz, rgb, patch, 10, 8, 0, 255 * 3), rgb)


def test_geometry_hand_case():
    # two points of different patches in ONE cell (grid 8, cell [8,16)^3); a far point in another cell
    xyz = np.array([[9, 9, 9], [15, 15, 15], [500, 500, 500]], np.uint16)
    patch = np.array([0, 1, 0], np.uint16)
    out = ob.spec_smooth_geometry(xyz, patch, 10, 8, 0)
    # point 0: r = 1 < 4 -> lower cell s = 0 (empty), upper cell = its own with weight t = 2*(9-4)+1 = 11 per axis
    # centroid = (9+15)/2 = 12 exactly, whatever the weight; moved since d2 > 0
    assert tuple(out[0]) == (12, 12, 12) and tuple(out[1]) == (12, 12, 12)
    assert tuple(out[2]) == (500, 500, 500)                 # its cell holds one patch only
    # threshold gates the move: |p - C| = 3*sqrt(3) = 5.2 per point -> T = 6 keeps, T = 5 moves
    assert np.array_equal(ob.spec_smooth_geometry(xyz, patch, 10, 8, 6), xyz)
    assert tuple(ob.spec_smooth_geometry(xyz, patch, 10, 8, 5)[0]) == (12, 12, 12)


def test_geometry_trilinear_weights_hand_case():
    # point at x = 17 (cell 2, r = 1 < 4): x-pair = cells 1 and 2, t = 2*(17 - (8+4)) + 1 = 11, weights (5, 11).
    # y, z = 12 (r = 4): pair = cells 1 and 2 with t = 2*(12-12)+1 = 1 -> weights (15, 1); cells 2 in y/z are empty.
    xyz = np.array([[17, 12, 12], [10, 12, 12], [18, 12, 12]], np.uint16)      # cells (2,1,1), (1,1,1), (2,1,1)
    patch = np.array([0, 1, 1], np.uint16)                                      # cell (2,1,1) is mixed
    out = ob.spec_smooth_geometry(xyz, patch, 10, 8, 0)
    # for point 0: num_x = 5*15*15*10 + 11*15*15*(17+18), den = 5*225*1 + 11*225*2 -> C = (50 + 385)/27 = 16.11
    c16 = (16 * (5 * 10 + 11 * 35) * 225 + (27 * 225) // 2) // (27 * 225)
    assert out[0][0] == (c16 + 8) >> 4 == 16
    assert out[0][1] == 12 and out[0][2] == 12


def test_color_hand_case():
    xyz = np.array([[9, 9, 9], [15, 15, 15], [10, 10, 10]], np.uint16)
    patch = np.array([0, 1, 0], np.uint16)
    rgb = np.array([[10, 20, 30], [40, 50, 60], [70, 80, 90]], np.uint8)
    out = ob.spec_smooth_color(xyz, rgb, patch, 10, 8, 1, 0)
    assert all(tuple(c) == (40, 50, 60) for c in out)       # mean of the single mixed cell, rounded
    assert np.array_equal(ob.spec_smooth_color(xyz, rgb, patch, 10, 8, 1000, 0), rgb)   # Ts gate


def test_spec_on_a_reconstructed_frame_moves_only_patch_boundaries():
    f = cases.overlapping_3d_frame(0)
    st, ref = ob.reconstruct(f)
    xyz, part = ob.xyz_array(ref), ref["partition"].astype(np.uint16)
    out = ob.spec_smooth_geometry(xyz, part, 10, 8, 2)
    moved = np.any(out != xyz, axis=1)
    assert 0 < moved.sum() < len(xyz) // 2


# ---------------------------------------------------------------- HIP kernels vs the spec (GPU)
@pytest.mark.gpu
@pytest.mark.parametrize("frames,params", [
    ("medium", dict(grid_size=8, threshold=2, color_grid_size=8, color_threshold_smoothing=10, color_threshold_difference=60)),
    ("medium", dict(grid_size=4, threshold=0, color_grid_size=2, color_threshold_smoothing=0, color_threshold_difference=765)),
    ("medium", dict(grid_size=16, threshold=5)),
    ("longdress", dict(grid_size=8, threshold=3, color_grid_size=8, color_threshold_smoothing=20, color_threshold_difference=100)),
    # depths drawn at random per pixel: the 64 consecutive points of a wave scatter over dozens of cells — more than a
    # wave's cell list holds, and cells a multiple of four apart in every axis (the statistics kernel's slot collisions)
    ("scattered", dict(grid_size=8, threshold=1, color_grid_size=8, color_threshold_smoothing=5, color_threshold_difference=200)),
])
def test_hip_smoothing_matches_spec(frames, params):
    from tmc2rs import recon
    fr = ([cases.overlapping_3d_frame(i) for i in range(3)] if frames in ("medium", "scattered")
          else [cases.overlapping_3d_frame(0, base=synth.longdress_frame(1))])
    if frames == "scattered":
        rng = np.random.RandomState(7)
        for f in fr:
            d0 = rng.randint(0, 800, size=f["geometry"][0].shape).astype(np.uint16)
            f["geometry"] = [d0, (d0 + rng.randint(0, 64, size=d0.shape)).astype(np.uint16)]
    ctx = recon.Context(0)
    g = ctx.gof(fr, flags=_abi.VPCC_GOF_WANT_PATCH_INDEX)
    g.reconstruct()
    before = [g.download(i, want_patch_index=True) for i in range(len(fr))]
    g.smooth(10, **params)
    for i, b in enumerate(before):
        after = g.download(i)
        exp_xyz = ob.spec_smooth_geometry(b["xyz"], b["patch_index"], 10, params["grid_size"], params["threshold"])
        assert np.array_equal(after["xyz"], exp_xyz)
        assert np.any(exp_xyz != b["xyz"])                  # the case really exercises the filter
        if params.get("color_grid_size"):
            exp_rgb = ob.spec_smooth_color(exp_xyz, b["rgb"], b["patch_index"], 10, params["color_grid_size"],
                                           params["color_threshold_smoothing"], params["color_threshold_difference"])
            assert np.array_equal(after["rgb"], exp_rgb)
        else:
            assert np.array_equal(after["rgb"], b["rgb"])
    g.close()
    ctx.close()


def _check_against_spec(g, before, bitdepth, params):
    for i, b in enumerate(before):
        after = g.download(i)
        exp_xyz = b["xyz"]
        if params.get("grid_size"):
            exp_xyz = ob.spec_smooth_geometry(b["xyz"], b["patch_index"], bitdepth, params["grid_size"], params["threshold"])
        assert np.array_equal(after["xyz"], exp_xyz), i
        exp_rgb = b["rgb"]
        if params.get("color_grid_size"):
            exp_rgb = ob.spec_smooth_color(exp_xyz, b["rgb"], b["patch_index"], bitdepth, params["color_grid_size"],
                                           params["color_threshold_smoothing"], params["color_threshold_difference"])
        assert np.array_equal(after["rgb"], exp_rgb), i
    return exp_xyz, exp_rgb


@pytest.mark.gpu
@pytest.mark.parametrize("bitdepth,params", [
    # grid sizes that are no power of two: the cell coordinate by the float reciprocal (G < 128) ...
    (10, dict(grid_size=6, threshold=1, color_grid_size=10, color_threshold_smoothing=4, color_threshold_difference=90)),
    (10, dict(grid_size=12, threshold=2, color_grid_size=12, color_threshold_smoothing=6, color_threshold_difference=120)),   # one pass for both
    # ... and by the integer division (G >= 128): seven cells per axis, thousands of points per cell
    (10, dict(grid_size=160, threshold=0, color_grid_size=160, color_threshold_smoothing=0, color_threshold_difference=765)),
    # one filter at a time
    (10, dict(grid_size=8, threshold=2)),
    (10, dict(color_grid_size=8, color_threshold_smoothing=3, color_threshold_difference=150)),
    (10, dict(color_grid_size=5, color_threshold_smoothing=3, color_threshold_difference=150)),
    # a finer grid than the content needs (11 bits: 256 cells per axis, 0.8 GB of cells per frame)
    (11, dict(grid_size=8, threshold=1, color_grid_size=8, color_threshold_smoothing=5, color_threshold_difference=100)),
])
def test_hip_smoothing_grids_and_modes(bitdepth, params):
    from tmc2rs import recon
    fr = [cases.overlapping_3d_frame(i) for i in range(2)]
    ctx = recon.Context(0)
    g = ctx.gof(fr, flags=_abi.VPCC_GOF_WANT_PATCH_INDEX)
    g.reconstruct()
    before = [g.download(i, want_patch_index=True) for i in range(len(fr))]
    g.smooth(bitdepth, **params)
    xyz, rgb = _check_against_spec(g, before, bitdepth, params)
    assert np.any(xyz != before[-1]["xyz"]) or np.any(rgb != before[-1]["rgb"])          # the case exercises a filter
    g.close()
    ctx.close()


def _frame_of_k_points(k):
    """k points in one grid cell: occupancy precision 1, the first ceil(k / 2) pixels of the collapsed frame's blocks
    occupied (two points per pixel; an odd k: the last pixel's two layers are equal, and the second is dropped as a
    duplicate, codec.rs:432-440)."""
    f = _collapsed_frame(4, prec=1)
    occ = np.zeros_like(f["occupancy"])
    g0, g1 = f["geometry"][0].copy(), f["geometry"][1].copy()
    bw = 320 // 16
    for t in range((k + 1) // 2):
        blk, within = divmod(t, 256)
        y, x = (blk // bw) * 16 + within // 16, (blk % bw) * 16 + within % 16
        occ[y, x] = 1
        if k % 2 and t == (k + 1) // 2 - 1:
            g1[y, x] = g0[y, x]
    f["occupancy"], f["geometry"] = occ, [g0, g1]
    return f


@pytest.mark.gpu
def test_hip_smoothing_frames_of_a_few_points():
    """The kernels read four points per lane: frames of 0 ... 9 points, and of 255 ... 257, 511 ... 513 (a wave's chunk of
    256 points, a filter wave's 512) and 1 023 ... 1 025 points (a cell list's span); beyond 512 points a second patch
    shares the cell."""
    from tmc2rs import recon
    ctx = recon.Context(0)
    params = dict(grid_size=8, threshold=0, color_grid_size=8, color_threshold_smoothing=0, color_threshold_difference=765)
    wanted = [0, 1, 2, 3, 4, 5, 7, 9, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025]
    fr = [_frame_of_k_points(k) for k in wanted]
    g = ctx.gof(fr, flags=_abi.VPCC_GOF_WANT_PATCH_INDEX)
    g.reconstruct()
    before = [g.download(i, want_patch_index=True) for i in range(len(fr))]
    assert [b["n"] for b in before] == wanted
    g.smooth(10, **params)
    xyz, _ = _check_against_spec(g, before, 10, params)
    assert np.any(xyz != before[-1]["xyz"])
    g.close()
    ctx.close()


@pytest.mark.gpu
def test_hip_smoothing_many_patches_in_a_wave():
    """Patches of one block with a few points each: a wave's 256 points hold dozens of patches (the statistics kernel works
    on a chunk patch by patch), all in the same handful of cells (every cell mixes patches)."""
    from tmc2rs import recon
    f = _collapsed_frame(120)
    occ = np.zeros_like(f["occupancy"])
    occ[::4, ::4] = 1                                  # one occupancy sample (32 points) per block
    f["occupancy"] = occ
    p = f["patches"].copy()
    p["u1"] = 101 + (np.arange(len(p)) % 3) * 5        # two neighbouring cells, shared by dozens of patches each
    f["patches"] = p
    ctx = recon.Context(0)
    params = dict(grid_size=8, threshold=0, color_grid_size=8, color_threshold_smoothing=0, color_threshold_difference=765)
    g = ctx.gof([f], flags=_abi.VPCC_GOF_WANT_PATCH_INDEX)
    g.reconstruct()
    before = [g.download(0, want_patch_index=True)]
    assert before[0]["n"] == 120 * 32 and len(np.unique(before[0]["patch_index"][:256])) == 8
    g.smooth(10, **params)
    xyz, _ = _check_against_spec(g, before, 10, params)
    assert np.any(xyz != before[0]["xyz"])
    g.close()
    ctx.close()


@pytest.mark.gpu
def test_smoothing_in_chunks_of_frames(monkeypatch):
    """A gof whose grids exceed the scratch limit (16 GiB; 11-bit content with grid 8 needs 0.8 GB per frame) is smoothed
    in chunks of frames that reuse the same scratch slots.  With both filters in one pass the moved-point bits of a slot
    must not survive into the next chunk: a big frame in which half of all points move, then smaller ones in its slot."""
    from tmc2rs import recon
    rng = np.random.RandomState(11)
    fr = [cases.overlapping_3d_frame(0, base=synth.longdress_frame(1)), cases.overlapping_3d_frame(1), cases.overlapping_3d_frame(2),
          cases.overlapping_3d_frame(0, base=synth.longdress_frame(2)), cases.overlapping_3d_frame(3)]
    for f in fr:
        d0 = rng.randint(0, 800, size=f["geometry"][0].shape).astype(np.uint16)
        f["geometry"] = [d0, (d0 + rng.randint(0, 64, size=d0.shape)).astype(np.uint16)]
    params = dict(grid_size=8, threshold=1, color_grid_size=8, color_threshold_smoothing=5, color_threshold_difference=200)
    ctx = recon.Context(0)
    for limit_mb, frames_per_chunk in ((250, 2), (120, 1)):           # a slot is 98 MB (grid 8, 10 bits, both filters)
        monkeypatch.setenv("VPCC_SMOOTH_SCRATCH_LIMIT_MB", str(limit_mb))
        g = ctx.gof(fr, flags=_abi.VPCC_GOF_WANT_PATCH_INDEX)
        g.reconstruct()
        before = [g.download(i, want_patch_index=True) for i in range(len(fr))]
        assert before[0]["n"] > 2 * before[1]["n"]
        g.smooth(10, **params)
        for i, b in enumerate(before):
            after = g.download(i)
            exp_xyz = ob.spec_smooth_geometry(b["xyz"], b["patch_index"], 10, 8, 1)
            exp_rgb = ob.spec_smooth_color(exp_xyz, b["rgb"], b["patch_index"], 10, 8, 5, 200)
            assert np.any(exp_xyz != b["xyz"])
            assert np.array_equal(after["xyz"], exp_xyz), (frames_per_chunk, i)
            assert np.array_equal(after["rgb"], exp_rgb), (frames_per_chunk, i)
        g.close()
    ctx.close()


@pytest.mark.gpu
def test_smoothing_makes_do_with_the_memory_there_is(monkeypatch):
    """The grids of a range of frames are one allocation; when the device has not got that much the range is smoothed in
    halves, quarters ... (the test hook makes allocations above a size fail) — and when not even one frame's grid fits, the
    call says so instead of leaving a stale device error behind."""
    from tmc2rs import recon
    fr = [cases.overlapping_3d_frame(i) for i in range(5)]
    params = dict(grid_size=8, threshold=1, color_grid_size=8, color_threshold_smoothing=5, color_threshold_difference=200)
    ctx = recon.Context(0)
    g = ctx.gof(fr, flags=_abi.VPCC_GOF_WANT_PATCH_INDEX)
    g.reconstruct()
    before = [g.download(i, want_patch_index=True) for i in range(len(fr))]
    monkeypatch.setenv("VPCC_SMOOTH_ALLOC_FAIL_ABOVE_MB", "250")       # a slot is 98 MB: 5 frames fail, 3 fail, 2 fit
    g.smooth(10, **params)
    _check_against_spec(g, before, 10, params)
    g.close()
    g = ctx.gof(fr, flags=_abi.VPCC_GOF_WANT_PATCH_INDEX)
    g.reconstruct()
    monkeypatch.setenv("VPCC_SMOOTH_ALLOC_FAIL_ABOVE_MB", "50")
    with pytest.raises(recon.VpccError) as e:
        g.smooth(10, **params)
    assert e.value.status == _abi.VPCC_ERR_DEVICE and "no device memory for the grid of one frame" in str(e.value)
    monkeypatch.delenv("VPCC_SMOOTH_ALLOC_FAIL_ABOVE_MB")
    g.smooth(10, **params)                                             # the gof is none the worse for it
    _check_against_spec(g, before, 10, params)
    g.close()
    ctx.close()


def _collapsed_frame(n_blocks, prec=4):
    """Every patch is one fully occupied block whose 512 points (lod 0: all pixels share the tangent and bitangent
    coordinate; two depths) fall into ONE grid cell."""
    f = synth.make_frame(320, 256, prec, 16, seed=5, max_side=2, cover_target=0.5)
    bw = 320 // 16
    p = np.zeros(n_blocks, dtype=_abi.PATCH_DTYPE)
    p["u0"] = np.arange(n_blocks) % bw
    p["v0"] = np.arange(n_blocks) // bw
    p["size_u0"] = p["size_v0"] = 1
    p["u1"], p["v1"], p["d1"] = 101, 102, 96
    p["lod_x"] = p["lod_y"] = 0
    p["normal_axis"], p["tangent_axis"], p["bitangent_axis"] = 0, 2, 1
    f["patches"] = p
    f["occupancy"] = np.ones_like(f["occupancy"])
    f["geometry"] = [np.full_like(f["geometry"][0], 8), np.full_like(f["geometry"][1], 12)]    # depths 2 and 3
    return f


@pytest.mark.gpu
def test_smoothing_cell_bound_is_checked():
    """The cells' 32-bit sums hold 65 537 points of 16-bit values: at the bound the result is the specification's, beyond
    it the gof reports VPCC_ERR_UNSUPPORTED instead of a silently different output."""
    from tmc2rs import recon
    ctx = recon.Context(0)
    params = dict(grid_size=8, threshold=0, color_grid_size=8, color_threshold_smoothing=0, color_threshold_difference=765)
    g = ctx.gof([_collapsed_frame(128), cases.overlapping_3d_frame(0)], flags=_abi.VPCC_GOF_WANT_PATCH_INDEX)   # 128 x 512 = 65 536 points in one cell
    g.reconstruct()
    b = g.download(0, want_patch_index=True)
    assert b["n"] == 65536 and len(np.unique(b["xyz"] // 8, axis=0)) == 1
    g.smooth(10, **params)
    after = g.download(0)
    exp_xyz = ob.spec_smooth_geometry(b["xyz"], b["patch_index"], 10, 8, 0)
    assert np.array_equal(after["xyz"], exp_xyz)
    assert np.array_equal(after["rgb"], ob.spec_smooth_color(exp_xyz, b["rgb"], b["patch_index"], 10, 8, 0, 765))
    g.close()
    g = ctx.gof([_collapsed_frame(200)], flags=_abi.VPCC_GOF_WANT_PATCH_INDEX)                                    # 102 400 points in one cell
    g.reconstruct()
    assert g.download(0)["n"] == 102400
    g.smooth(10, **params)
    with pytest.raises(recon.VpccError) as e:
        g.download(0)
    assert e.value.status == _abi.VPCC_ERR_UNSUPPORTED and "one grid cell" in str(e.value)
    g.close()
    ctx.close()


@pytest.mark.gpu
def test_smoothing_needs_patch_index():
    from tmc2rs import recon
    ctx = recon.Context(0)
    g = ctx.gof([cases.medium_frame(0)])
    g.reconstruct()
    with pytest.raises(recon.VpccError) as e:
        g.smooth(10, grid_size=8, threshold=1)
    assert e.value.status == _abi.VPCC_ERR_STATE
    g.close()
    ctx.close()


@pytest.mark.gpu
def test_repeated_smoothing_launches_leave_the_grid_clean():
    """The dense grids are cleared cell by cell from the touched list, not by a memset: a second reconstruct +
    smooth on the same gof — with another grid size, then the first again — must give the spec's result again."""
    from tmc2rs import recon
    fr = [cases.overlapping_3d_frame(i) for i in range(2)]
    ctx = recon.Context(0)
    g = ctx.gof(fr, flags=_abi.VPCC_GOF_WANT_PATCH_INDEX)
    for grid, cgrid in ((8, 8), (4, 16), (8, 8), (8, 8)):
        g.reconstruct()
        before = [g.download(i, want_patch_index=True) for i in range(len(fr))]
        g.smooth(10, grid_size=grid, threshold=1, color_grid_size=cgrid, color_threshold_smoothing=5,
                 color_threshold_difference=80)
        for i, b in enumerate(before):
            after = g.download(i)
            exp_xyz = ob.spec_smooth_geometry(b["xyz"], b["patch_index"], 10, grid, 1)
            exp_rgb = ob.spec_smooth_color(exp_xyz, b["rgb"], b["patch_index"], 10, cgrid, 5, 80)
            assert np.array_equal(after["xyz"], exp_xyz) and np.array_equal(after["rgb"], exp_rgb)
    g.close()
    ctx.close()
