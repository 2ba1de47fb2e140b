"""Parity tests proper: the HIP path, called through the C ABI, against the CPU oracle on the
same seeded inputs.  Bar: bit-exact for xyz, occupancy, block->patch, patch index AND for the
8-bit colour (the north-star tolerance is +-1 LSB; the f64 colour maths is compiled without
contraction so exact equality is asserted)."""
import ctypes as C

import os

import numpy as np
import pytest

import cases
import oracle_binding as ob
from tmc2rs import _abi, recon, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = recon.Context(0)
    yield c
    c.close()


def _check(res, ref, patch_index=True, colour=True):
    assert res["n"] == ref["n"]
    assert np.array_equal(res["xyz"], ob.xyz_array(ref)), "integer geometry must be bit-exact"
    if colour:
        assert np.array_equal(res["rgb"], ob.rgb_array(ref)), "8-bit colour must be bit-exact"
    if patch_index and "patch_index" in res:
        assert np.array_equal(res["patch_index"].astype(np.uint64), ref["partition"])


@pytest.mark.parametrize("name", sorted(cases.PARITY_CASES))
def test_reconstruct_frame_matches_oracle(ctx, name):
    f = cases.PARITY_CASES[name]()
    st, ref = ob.reconstruct(f)
    assert st == 0
    res = ctx.reconstruct_frame(f, want_patch_index=True)
    # without an attribute the reference's PointSet3 carries no colours at all (codec.rs:47-50, 274-276)
    _check(res, ref, colour=f.get("attribute_count", 1) > 0)


@pytest.mark.parametrize("name", ["small2_wide", "exotic_orientations", "overlap", "block8_ragged",
                                  "block32_multichunk", "medium1_randocc", "empty_no_occupancy"])
def test_block_to_patch_and_occupancy_match_oracle(ctx, name):
    f = cases.PARITY_CASES[name]()
    st, ref = ob.reconstruct(f)
    assert st == 0
    b2p = ctx.generate_block_to_patch(f)
    assert np.array_equal(b2p.astype(np.uint64), ref["block_to_patch"])
    occ = ctx.upsample_occupancy(f)
    assert np.array_equal(occ, ref["occupancy_map"])


def test_gof_batch_of_mixed_frames(ctx):
    frames = [cases.medium_frame(i, occupancy_values="random" if i % 2 else "one") for i in range(5)]
    frames += [synth.small_frame(0), cases.exotic_frame(), cases.block8_frame()]
    g = ctx.gof(frames, flags=_abi.VPCC_GOF_WANT_PATCH_INDEX)
    g.reconstruct()
    counts = g.point_counts()
    for i, f in enumerate(frames):
        st, ref = ob.reconstruct(f)
        assert counts[i] == ref["n"]
        _check(g.download(i, want_patch_index=True), ref)
    # sub-range relaunch leaves the other frames' results intact and reproduces its own
    g.reconstruct(first=2, count=3)
    for i in (0, 3, 7):
        st, ref = ob.reconstruct(frames[i])
        _check(g.download(i, want_patch_index=True), ref)
    g.close()


def test_general_and_fast_paths_agree(ctx):
    frames = [cases.medium_frame(i) for i in range(3)]
    a = ctx.gof(frames, flags=_abi.VPCC_GOF_FORCE_GENERAL)
    b = ctx.gof(frames)
    a.reconstruct()
    b.reconstruct()
    for i in range(3):
        ra, rb = a.download(i), b.download(i)
        assert ra["n"] == rb["n"] and np.array_equal(ra["xyz"], rb["xyz"]) and np.array_equal(ra["rgb"], rb["rgb"])
    a.close()
    b.close()


def test_both_kernels_of_the_general_sequence(ctx):
    """The general sequence's pass has two kernels: k_general_blocks for frames whose block side is a power of two in [16, 256]
    with an occupancy precision that is a power of two and patches whose three axes differ, k_general for any frame.  The named
    cases (all orientations, relative D1, one map, no attribute, strided planes, blocks of 32) and a batch of medium frames go
    through each — VPCC_GENERAL_ANY_FRAME sends eligible frames to k_general too — and must equal the oracle; a patch whose
    tangent axis is its normal axis (the reference never builds one; the interface takes it) sends its gof to k_general."""
    named = [cases.exotic_frame(), cases.relative_d1_frame(), cases.overlap_frame(), cases.block32_frame(),
             cases.single_map_frame(), cases.no_attribute_frame(), cases.strided_frame(), cases.wide_samples_frame()]
    batch = [cases.medium_frame(i) for i in range(11)]                     # (not a multiple of the eight XCDs)
    odd = cases.medium_frame(3)
    odd["patches"] = odd["patches"].copy()
    odd["patches"]["tangent_axis"][::3] = odd["patches"]["normal_axis"][::3]
    flags = _abi.VPCC_GOF_FORCE_GENERAL | _abi.VPCC_GOF_PROFILE | _abi.VPCC_GOF_WANT_PATCH_INDEX
    try:
        for env, kernel in ((None, "k_general_blocks"), ("1", "k_general")):
            if env:
                os.environ["VPCC_GENERAL_ANY_FRAME"] = env
            for frames in ([f] for f in named):
                g = ctx.gof(frames, flags=flags)
                g.reconstruct()
                assert [k for k, _ in g.kernel_times()] == ["k_block_owner", kernel], g.kernel_times()
                st, ref = ob.reconstruct(frames[0])
                assert st == 0
                # (without an attribute the reference's PointSet3 carries no colours at all: codec.rs:47-50, 274-276)
                _check(g.download(0, want_patch_index=True), ref, colour=frames[0].get("attribute_count", 1) > 0)
                g.close()
            g = ctx.gof(batch, flags=flags)
            for first, count in ((0, 11), (3, 5), (10, 1)):
                g.reconstruct(first, count)
                assert [k for k, _ in g.kernel_times()] == ["k_block_owner", kernel]
            for i, f in enumerate(batch):
                st, ref = ob.reconstruct(f)
                assert st == 0
                _check(g.download(i, want_patch_index=True), ref)
            g.close()
    finally:
        os.environ.pop("VPCC_GENERAL_ANY_FRAME", None)
    mixed = [batch[0], odd, batch[1], cases.truncation_frame(), cases.block8_frame()]      # (coinciding axes; blocks of 8)
    g = ctx.gof(mixed, flags=flags)
    g.reconstruct()
    assert [k for k, _ in g.kernel_times()] == ["k_block_owner", "k_general"]
    for i, f in enumerate(mixed):
        st, ref = ob.reconstruct(f)
        assert st == 0
        _check(g.download(i, want_patch_index=True), ref)
    g.close()


def test_every_block_size_and_occupancy_precision(ctx):
    """occupancy_resolution 1 .. 128 (log2_patch_packing_block_size 0 .. 7) x occupancy_precision 1, 2, 4, 8 — blocks smaller than an
    occupancy sample included — on the path gof creation chooses (the tile kernel for blocks of 16) and on the general sequence:
    points, colours, partition and block ownership against the oracle."""
    import itertools
    for R, prec in itertools.product((1, 2, 4, 8, 16, 32, 64, 128), (1, 2, 4, 8)):
        W, H = max(256, 4 * R), max(192, 3 * R)
        f = synth.make_frame(W, H, prec, R, seed=0x7E570000 + R * 16 + prec, max_side=max(2, 64 // R), cover_target=0.6, size_skew=1.5)
        st, ref = ob.reconstruct(f)
        assert st == 0 and ref["n"] > 20000
        for flags in (0, _abi.VPCC_GOF_FORCE_GENERAL):
            g = ctx.gof([f], flags=flags | _abi.VPCC_GOF_WANT_PATCH_INDEX)
            g.reconstruct()
            _check(g.download(0, want_patch_index=True), ref)
            b2p, items = g.block_to_patch(0, (W // R) * (H // R))
            assert np.array_equal(b2p.astype(np.uint64), ref["block_to_patch"]), (R, prec, flags)
            assert (items > 0) == (R == 16 and flags == 0), (R, prec, flags, items)
            g.close()


def test_the_largest_patch_table(ctx):
    """65 535 patches (the ABI's and the partition's limit: patch indices are 16 bits wide) of one block each on a 4096 x 4096
    canvas, every orientation and view, the last block left to nobody; plus overlapping patches among the first ones (the later one
    owns the block).  Points, colours, partition and block ownership against the oracle, on both kernel paths."""
    W = H = 4096
    n = 65535
    k = np.arange(n)
    p = np.zeros(n, dtype=_abi.PATCH_DTYPE)
    p["u0"], p["v0"], p["size_u0"], p["size_v0"] = k % 256, k // 256, 1, 1
    p["u0"][5], p["v0"][5] = p["u0"][4], p["v0"][4]               # patch 5 on top of patch 4
    views = np.array([synth.VIEW_AXES[v] for v in range(6)])
    v = k % 6
    p["normal_axis"], p["tangent_axis"], p["bitangent_axis"], p["projection_mode"] = views[v, 0], views[v, 1], views[v, 2], views[v, 3]
    p["orientation"] = np.where(k % 5 == 0, 1, 0)                  # Swap every fifth (one block: every orientation stays inside)
    p["u1"], p["v1"], p["d1"] = (k * 7) % 60000, (k * 13) % 60000, (k * 3) % 900
    p["lod_x"] = p["lod_y"] = 1
    rng = np.random.RandomState(65535)
    occ = (rng.randint(0, 4, size=(H // 4, W // 4)) != 0).astype(np.uint8)
    g0 = rng.randint(0, 1000, size=(H, W)).astype(np.uint16)
    g1 = (g0 + 4 * rng.randint(0, 3, size=(H, W))).astype(np.uint16)
    attr = [(rng.randint(64, 941, size=(H, W)).astype(np.uint16), rng.randint(64, 961, size=(H // 2, W // 2)).astype(np.uint16),
             rng.randint(64, 961, size=(H // 2, W // 2)).astype(np.uint16)) for _ in range(2)]
    f = {"width": W, "height": H, "occupancy_resolution": 16, "occupancy_precision": 4, "map_count": 2, "absolute_d1": 1,
         "attribute_count": 1, "flags": 0, "patches": p, "occupancy": occ, "geometry": [g0, g1], "attribute": attr}
    st, ref = ob.reconstruct(f)
    assert st == 0 and ref["n"] > 20_000_000
    for flags in (0, _abi.VPCC_GOF_FORCE_GENERAL):
        g = ctx.gof([f], flags=flags | _abi.VPCC_GOF_WANT_PATCH_INDEX)
        g.reconstruct()
        got = g.download(0, want_patch_index=True)
        assert got["n"] == ref["n"]
        assert np.array_equal(got["xyz"], ob.xyz_array(ref)) and np.array_equal(got["rgb"], ob.rgb_array(ref))
        assert np.array_equal(got["patch_index"].astype(np.uint64), ref["partition"]) and int(got["patch_index"].max()) == n - 1
        b2p, _ = g.block_to_patch(0, 65536)
        assert np.array_equal(b2p.astype(np.uint64), ref["block_to_patch"]) and b2p[4] == 6 and b2p[5] == 0 and b2p[-1] == 0
        g.close()


def test_capacity_beyond_32_bit_byte_offsets_takes_the_general_sequence(ctx):
    """The tile kernel's store loop addresses a frame's positions with 32-bit byte offsets: a gof whose frames may hold more
    than 715 827 880 points is reconstructed by the general sequence (64-bit indices; `tools/exp_max_canvas.py` runs a
    32768 x 32768 frame of 811 M points through it).  Here: a small frame with such a capacity (4.3 + 2.1 GB of outputs)."""
    f = cases.medium_frame(3)
    st, ref = ob.reconstruct(f)
    assert st == 0
    for cap, tiles in ((715_827_880, True), (715_827_881, False)):
        g = ctx.gof([f], capacity=cap, flags=_abi.VPCC_GOF_PROFILE)
        g.reconstruct()
        got = g.download(0)
        names = [n for n, _ in g.kernel_times()]
        assert any("k_recon_tiles" in n for n in names) == tiles, names
        assert got["n"] == ref["n"] and np.array_equal(got["xyz"], ob.xyz_array(ref)) and np.array_equal(got["rgb"], ob.rgb_array(ref))
        g.close()


def test_capacity_too_small_is_reported_not_overrun(ctx):
    f = cases.medium_frame(0)
    st, ref = ob.reconstruct(f)
    n = ref["n"]
    g = ctx.gof([f], capacity=n - 100)
    g.reconstruct()
    assert g.point_counts()[0] == n                       # the count is still the true count
    assert g.frame_status(0) == _abi.VPCC_ERR_CAPACITY
    with pytest.raises(recon.VpccError) as e:
        g.download(0)
    assert e.value.status == _abi.VPCC_ERR_CAPACITY
    g.close()
    g = ctx.gof([f], capacity=n)                           # exactly enough
    g.reconstruct()
    _check(g.download(0), ref)
    g.close()


def test_full_size_longdress_frame(ctx):
    f = synth.longdress_frame(0)
    st, ref = ob.reconstruct(f)
    assert st == 0 and 700_000 < ref["n"] < 900_000
    res = ctx.reconstruct_frame(f, want_patch_index=True)
    _check(res, ref)
    # size-independent properties
    assert np.all(np.diff(res["patch_index"].astype(np.int64)) >= 0)     # emission is patch-ordered
    assert res["xyz"].max() < 1024                                        # 10-bit coordinates


def test_full_size_owlii_frame(ctx):
    f = synth.owlii_frame(0)
    st, ref = ob.reconstruct(f)
    assert st == 0 and ref["n"] > 1_800_000
    _check(ctx.reconstruct_frame(f, want_patch_index=True), ref)


def test_gof_of_very_unequal_frames_on_the_tile_path(ctx):
    """Frames of one launch get shares of the resident workgroups in proportion to their tile counts
    (TileLaunchMap, plan_tile_launch): a GOF that mixes one full-size frame, medium and tiny frames and frames
    without any patch — over full, partial and single-frame launch ranges, repeated (every launch re-arms the ticket
    counters with the shares of ITS range)."""
    empty = dict(synth.small_frame(3))
    empty["patches"] = empty["patches"][:0]
    makers = [lambda i: synth.longdress_frame(i), lambda i: cases.medium_frame(i), lambda i: synth.small_frame(i),
              lambda i: dict(empty), lambda i: cases.medium_frame(100 + i, occupancy_values="random"),
              lambda i: synth.small_frame(50 + i)]
    order = [0, 1, 2, 3, 4, 5, 1, 2, 2, 3, 1, 4, 5, 5, 2, 1, 3, 2, 4, 1, 2, 5, 1, 2]       # one big frame, 24 in all
    frames = [makers[k](i) for i, k in enumerate(order)]
    refs = [ob.reconstruct(f)[1] for f in frames]
    g = ctx.gof(frames, flags=_abi.VPCC_GOF_PROFILE | _abi.VPCC_GOF_WANT_PATCH_INDEX)
    for first, count in [(0, None), (0, None), (3, 10), (0, 1), (23, 1), (5, 19), (0, None)]:
        g.reconstruct(first=first, count=count)
        assert [k for k, _ in g.kernel_times()] == ["k_plan_tiles", "k_recon_tiles"]
        counts = g.point_counts()
        for i in range(first, len(frames) if count is None else first + count):
            assert counts[i] == refs[i]["n"], (first, count, i)
            _check(g.download(i, want_patch_index=True), refs[i])
    g.close()


def test_large_launch_runs_in_rounds(ctx):
    """Launches of more than 128 frames work through each XCD label's frames sixteen at a time, a workgroup moving on
    to the next frame of its team when its frame has no ticket left (k_recon_tiles, kFramesInFlight): 150 frames of
    different sizes (a partial second round, labels with 18 and 19 frames), launched twice and over a sub-range that
    is itself more than one round."""
    makers = [lambda i: cases.medium_frame(i % 12), lambda i: synth.small_frame(i % 20),
              lambda i: cases.medium_frame(200 + i % 10, occupancy_values="random")]
    frames = [makers[(i * 7 + i // 5) % 3](i) for i in range(150)]
    cache = {}
    refs = []
    for i, f in enumerate(frames):
        key = ((i * 7 + i // 5) % 3, i % 12 if (i * 7 + i // 5) % 3 == 0 else i % 20 if (i * 7 + i // 5) % 3 == 1 else i % 10)
        if key not in cache:
            cache[key] = ob.reconstruct(f)[1]
        refs.append(cache[key])
    g = ctx.gof(frames, flags=_abi.VPCC_GOF_PROFILE | _abi.VPCC_GOF_WANT_PATCH_INDEX)
    for first, count in [(0, None), (0, None), (7, 137), (0, None)]:
        g.reconstruct(first=first, count=count)
        assert [k for k, _ in g.kernel_times()] == ["k_plan_tiles", "k_recon_tiles"]
        counts = g.point_counts()
        for i in range(first, len(frames) if count is None else first + count):
            assert counts[i] == refs[i]["n"], (first, count, i)
            _check(g.download(i, want_patch_index=True), refs[i])
    g.close()


@pytest.mark.parametrize("n_frames", [1, 7, 8, 9, 31, 33, 63, 64, 65, 71, 127, 128, 129, 137, 200, 260])
def test_launch_shapes(ctx, n_frames):
    """Every shape of a tile-kernel launch — fewer frames than XCDs, labels of unequal length, proportional shares up
    to 128 frames, rounds above — on GOFs cycled from a few small and medium frames, full range and a sub-range."""
    pool = [synth.small_frame(i) for i in range(6)] + [cases.medium_frame(i) for i in range(3)]
    refs_pool = [ob.reconstruct(f)[1] for f in pool]
    pick = [(i * 5 + i // 3) % len(pool) for i in range(n_frames)]
    g = ctx.gof([pool[k] for k in pick], flags=_abi.VPCC_GOF_PROFILE | _abi.VPCC_GOF_WANT_PATCH_INDEX)
    ranges = [(0, n_frames)]
    if n_frames > 2:
        ranges += [(n_frames // 3, n_frames - n_frames // 3 - 1), (0, n_frames)]
    for first, count in ranges:
        g.reconstruct(first=first, count=count)
        assert [k for k, _ in g.kernel_times()] == ["k_plan_tiles", "k_recon_tiles"]
        counts = g.point_counts()
        for i in range(first, first + count):
            assert counts[i] == refs_pool[pick[i]]["n"], (n_frames, first, count, i)
        for i in sorted({first, first + count // 2, first + count - 1}):
            _check(g.download(i, want_patch_index=True), refs_pool[pick[i]])
    g.close()


def test_ticket_counters_survive_skipped_launches(ctx):
    """Ticket counters carry the launch generation instead of being cleared or re-armed: frames that sit out some
    launches of their gof (disjoint sub-ranges, repeated) must be found fresh when their turn comes again, and frames
    helped by workgroups of other frames must come out the same."""
    pool = [cases.medium_frame(i) for i in range(4)] + [synth.small_frame(i) for i in range(4)]
    refs_pool = [ob.reconstruct(f)[1] for f in pool]
    n = 40
    pick = [(3 * i + i // 7) % len(pool) for i in range(n)]
    g = ctx.gof([pool[k] for k in pick], flags=_abi.VPCC_GOF_WANT_PATCH_INDEX)
    for first, count in [(0, 9), (9, 20), (9, 20), (29, 11), (9, 20), (0, 9), (0, 40), (17, 3), (0, 40)]:
        g.reconstruct(first=first, count=count)
        counts = g.point_counts()
        for i in range(first, first + count):
            assert counts[i] == refs_pool[pick[i]]["n"], (first, count, i)
        for i in sorted({first, first + count // 2, first + count - 1}):
            _check(g.download(i, want_patch_index=True), refs_pool[pick[i]])
    g.close()


def test_pool_homes_hold_the_blocks_not_the_results():
    """vpcc_ctx_reserve: the context classifies the granules of ONE allocation by kind of VRAM region and every gof keeps
    its planes and outputs there, eight frames in one home, the next eight in the other.  Results are those of a gof
    without a pool and of the oracle; blocks return to the pool when their gof goes; a gof that does not fit falls back."""
    c = recon.Context(0)
    info = c.reserve(36)                                          # a context of its own: the pool lives as long as it does
    # (36 GiB; 54 when the second home lay further away; more when an earlier context of this process left its pool behind)
    assert info["GiB"] >= 36 and info["granules"] == info["GiB"] and info["kinds"] in (1, 2), info
    assert sum(info["GiB_of_kind"]) == info["GiB"] and info["in_use_MB"] == [0, 0]
    assert info["probe_GBps_same_kind"] > 500 or info["taken_over_from_an_earlier_context"], info
    distinct = [synth.longdress_frame(i) for i in range(5)]
    refs5 = [ob.reconstruct(f)[1] for f in distinct]
    n = 20                                                        # frames 0-7 and 16-19: part 0, frames 8-15: part 1
    frames = [distinct[i % 5] for i in range(n)]
    refs = [refs5[i % 5] for i in range(n)]
    g = c.gof(frames, capacity=1_000_000, flags=_abi.VPCC_GOF_WANT_PATCH_INDEX)
    used = c.pool_info()
    assert sum(used["in_use_MB"]) > 300 and used["blocks_outside_pool"] == 0, used
    if used["kinds"] == 2 and min(used["GiB_of_kind"]) >= 2:
        assert min(used["in_use_MB"]) > 100, used               # both homes hold a part
    g.reconstruct(first=6, count=5)
    for i in range(6, 11):
        _check(g.download(i, want_patch_index=True), refs[i])
    g.reconstruct()
    for i in range(n):
        _check(g.download(i, want_patch_index=True), refs[i])
    g.smooth(10, grid_size=8, threshold=4)
    g.sync()
    g2 = c.gof(frames[:9], capacity=1_000_000)                      # a second gof beside it
    g2.reconstruct()
    for i in (0, 8):
        _check(g2.download(i), refs[i])
    g.close()
    g2.close()
    assert c.pool_info()["in_use_MB"] == [0, 0]
    with pytest.raises(recon.VpccError):
        c.reserve(4)                                              # one pool per context
    c.close()


def test_pool_lends_memory_to_a_producer_of_device_planes():
    """vpcc_ctx_pool_alloc / _free: a GPU video decoder's frame pool in the context's pool (frames 0-7 of a launch from home 0,
    8-15 from home 1), handed to vpcc_gof_create as VPCC_MEM_DEVICE planes — what bench.py's `fresh_gof` leg does; the pool's
    accounts follow, a pointer that is not the pool's is refused, memory still lent goes with the context.  And
    vpcc_ctx_reserve_within (a budget for the search for a second home), vpcc_release_kept_pools."""
    hip = C.CDLL("libamdhip64.so.7")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    c = recon.Context(0)
    info = c.reserve(4, budget_ms=1.0)                              # (the budget is spent by the first allocation: no second home is looked for)
    assert info["GiB"] >= 4 and info["kinds"] in (1, 2), info
    frames = [cases.medium_frame(i) for i in range(10)]
    refs = [ob.reconstruct(f)[1] for f in frames]
    held, descs, keep = [], [], []
    for r0 in (0, 8):
        run = frames[r0:r0 + 8]
        planes = [[f["occupancy"]] + [f["geometry"][m] for m in range(2)] + [pl for m in range(2) for pl in f["attribute"][m]] for f in run]
        total = sum((pl.nbytes + 255) // 256 * 256 for fr in planes for pl in fr)
        base = c.pool_alloc((r0 // 8) % 2, total)
        assert base and base % 256 == 0
        held.append(base)
        at = base
        for f, pls in zip(run, planes):
            d, k = _abi.host_frame_desc(f)
            keep.append(k)
            ptrs = []
            for pl in pls:
                a = np.ascontiguousarray(pl)
                assert hip.hipMemcpy(at, a.ctypes.data, a.nbytes, 1) == 0
                ptrs.append(at)
                at += (a.nbytes + 255) // 256 * 256
            d.occupancy.y, d.occupancy.stride = ptrs[0], d.occupancy.width
            for m in range(2):
                d.geometry[m].y = ptrs[1 + m]
                d.attribute[m].y, d.attribute[m].u, d.attribute[m].v = ptrs[3 + 3 * m], ptrs[4 + 3 * m], ptrs[5 + 3 * m]
            descs.append(d)
    lent = sum(c.pool_info()["in_use_MB"])
    assert lent >= 2                                                 # (blocks are whole 2-MB pieces)
    g = c.gof(None, capacity=200_000, memory=_abi.VPCC_MEM_DEVICE, descs=descs, flags=_abi.VPCC_GOF_PROFILE)
    g.reconstruct()
    assert [k for k, _ in g.kernel_times()] == ["k_plan_tiles", "k_recon_tiles"]
    for i in (0, 7, 8, 9):
        _check(g.download(i), refs[i])
    assert sum(c.pool_info()["in_use_MB"]) > lent                    # the gof's outputs lie in the pool too
    g.close()
    assert sum(c.pool_info()["in_use_MB"]) == lent
    with pytest.raises(recon.VpccError):
        c.pool_free(held[0] + 256)                                   # not a pointer of vpcc_ctx_pool_alloc
    c.pool_free(held[0])
    assert 0 < sum(c.pool_info()["in_use_MB"]) < lent
    c.close()                                                        # (held[1] goes with the context; the pool is whole again and kept)
    assert c.lib.vpcc_release_kept_pools(0) == 1                     # ... until somebody gives it back to the driver
    assert c.lib.vpcc_release_kept_pools(0) == 0


def test_pool_too_small_falls_back():
    c = recon.Context(0)
    gib = c.reserve(2)["GiB"]                                       # (2; up to 6 with a second slab or a pool taken over)
    frames = [synth.longdress_frame(i % 3) for i in range(128)]     # 2.3 GB of planes + more output capacity than the pool holds
    g = c.gof(frames, capacity=int(gib * 2**30 / (128 * 9)) + 1_000_000)
    assert c.pool_info()["blocks_outside_pool"] >= 1
    g.reconstruct()
    for i in (0, 64, 127):
        _check(g.download(i), ob.reconstruct(frames[i])[1])
    g.close()
    c.close()


def test_device_outputs_stay_valid(ctx):
    """Pointers handed out by vpcc_gof_device_outputs stay valid across launches (nothing ever moves a gof's blocks)."""
    frames = [synth.longdress_frame(i) for i in range(4)]
    g = ctx.gof(frames, capacity=1_000_000)
    before = [g.device_outputs(i) for i in range(4)]
    g.reconstruct()
    g.reconstruct()
    assert [g.device_outputs(i) for i in range(4)] == before
    _check(g.download(3), ob.reconstruct(frames[3])[1])
    g.close()


def test_work_lists_are_planned_on_the_device(ctx):
    """generate_block_to_patch_from_occupancy_map_video (src/codec.rs:205-250) on the production path: block_to_patch and the
    single-pass kernel's work list are built by EVERY launch (k_plan_tiles: a workgroup per frame, block_to_patch and the patch
    table in LDS; the host writes O(patches) per frame, the virtual blocks are derived on the device), from the occupancy
    plane where it lies — host planes and the caller's device planes alike.  Against the oracle's block_to_patch; the work
    list holds exactly the owned blocks.  The same through the planning kernels that work in global memory (frames beyond
    k_plan_tiles' LDS take them: VPCC_NO_LDS_PLANNING=1 sends every frame there)."""
    import torch
    frames = [synth.longdress_frame(3), cases.medium_frame(5, occupancy_values="random"), cases.precision_frame(2, 9),
              cases.precision_frame(16, 4), cases.precision_frame(1, 2), cases.precision_frame(8, 6), synth.small_frame(1)]
    refs = [ob.reconstruct(f)[1] for f in frames]
    for env in (None, "1"):
        if env:
            os.environ["VPCC_NO_LDS_PLANNING"] = env
        try:
            for f, ref in zip(frames, refs):
                g = ctx.gof([f], flags=_abi.VPCC_GOF_PROFILE)
                nb = (f["width"] // 16) * (f["height"] // 16)
                with pytest.raises(recon.VpccError) as e:               # nothing is planned before a launch
                    g.block_to_patch(0, nb)
                assert e.value.status == _abi.VPCC_ERR_STATE
                g.reconstruct()
                assert [k for k, _ in g.kernel_times()] == (["k_plan_cover+items", "k_recon_tiles"] if env else ["k_plan_tiles", "k_recon_tiles"])
                b2p, items = g.block_to_patch(0, nb)
                assert np.array_equal(b2p.astype(np.uint64), ref["block_to_patch"].astype(np.uint64))
                assert items == int(np.count_nonzero(ref["block_to_patch"])) > 0
                _check(g.download(0), ref)
                g.close()
        finally:
            os.environ.pop("VPCC_NO_LDS_PLANNING", None)
    # the caller's device planes: the same list (round 3 kept every covered block there: the host could not see the occupancy)
    f, ref = frames[0], refs[0]
    dev = torch.device("cuda:0")
    d, keep = _abi.host_frame_desc(f)
    t = []

    def up(a):
        t.append(torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).to(dev))
        return t[-1].data_ptr()

    d.occupancy.y = up(f["occupancy"])
    d.occupancy.stride = d.occupancy.width
    for m in range(2):
        d.geometry[m].y = up(f["geometry"][m])
        d.attribute[m].y, d.attribute[m].u, d.attribute[m].v = (up(p) for p in f["attribute"][m])
    torch.cuda.synchronize()
    g = ctx.gof(None, capacity=1_000_000, memory=_abi.VPCC_MEM_DEVICE, descs=[d])
    g.reconstruct()
    b2p, items = g.block_to_patch(0, (f["width"] // 16) * (f["height"] // 16))
    assert np.array_equal(b2p.astype(np.uint64), ref["block_to_patch"].astype(np.uint64))
    assert items == int(np.count_nonzero(ref["block_to_patch"]))
    _check(g.download(0), ref)
    g.close()


def test_planning_in_rounds_and_beyond_the_lds(ctx):
    """k_plan_tiles compacts a frame's virtual blocks in rounds of 8 192 (a thread's run in a round fits its registers): a 2048 x 2048
    frame whose patch table is repeated three times — 846 patches, 23 397 virtual blocks over 16 384 canvas blocks, three rounds;
    every block owned by a patch of the LAST copy — through the planning kernel in LDS and through the kernels that work in
    global memory (VPCC_NO_LDS_PLANNING); and a table of 2 049 patches, one more than the LDS form takes, which must choose the
    other form by itself.  Points, colours, partition, block_to_patch and the number of work items against the oracle."""
    base = synth.owlii_frame(1)
    f = dict(base)
    f["patches"] = np.concatenate([base["patches"]] * 3)
    n_vb = int((f["patches"]["size_u0"].astype(np.int64) * f["patches"]["size_v0"]).sum())
    assert n_vb > 2 * 8192 and len(f["patches"]) <= 2048
    st, ref = ob.reconstruct(f)
    assert st == 0 and ref["n"] > 1_500_000
    assert int(ref["partition"].min()) >= 2 * len(base["patches"])          # the last copy owns everything
    nb = (f["width"] // 16) * (f["height"] // 16)
    for env, kernels in ((None, ["k_plan_tiles", "k_recon_tiles"]), ("1", ["k_plan_cover+items", "k_recon_tiles"])):
        if env:
            os.environ["VPCC_NO_LDS_PLANNING"] = env
        try:
            g = ctx.gof([f], capacity=2_400_000, flags=_abi.VPCC_GOF_PROFILE | _abi.VPCC_GOF_WANT_PATCH_INDEX)
            g.reconstruct()
            assert [k for k, _ in g.kernel_times()] == kernels
            _check(g.download(0, want_patch_index=True), ref)
            b2p, items = g.block_to_patch(0, nb)
            assert np.array_equal(b2p.astype(np.uint64), ref["block_to_patch"].astype(np.uint64))
            assert items == int(np.count_nonzero(ref["block_to_patch"]))
            g.reconstruct()                                                  # (and again, behind the re-planning of the query)
            _check(g.download(0, want_patch_index=True), ref)
            g.close()
        finally:
            os.environ.pop("VPCC_NO_LDS_PLANNING", None)
    # 2 049 patches: beyond the LDS form's table
    small = cases.medium_frame(7)
    many = dict(small)
    reps = 2049 // len(small["patches"]) + 1
    many["patches"] = np.concatenate([small["patches"]] * reps)[:2049]
    st, ref2 = ob.reconstruct(many)
    assert st == 0
    g = ctx.gof([many], flags=_abi.VPCC_GOF_PROFILE | _abi.VPCC_GOF_WANT_PATCH_INDEX)
    g.reconstruct()
    assert [k for k, _ in g.kernel_times()] == ["k_plan_cover+items", "k_recon_tiles"]
    _check(g.download(0, want_patch_index=True), ref2)
    g.close()


def test_borrowed_planes_may_change_between_launches(ctx):
    """A gof that borrows the caller's device planes (VPCC_MEM_DEVICE) reads them at vpcc_gof_reconstruct, on the launch's
    stream, and nowhere else: block_to_patch and the work lists are planned by every launch from the occupancy as it is
    THEN.  A decoder's frame pool is refilled between two launches of the same gof — on the launch stream, without any
    synchronisation with the host in between — with a frame whose occupancy covers blocks the first one left empty
    (round 4 planned once, when the gof was created: the second launch would have dropped those blocks' points)."""
    import torch
    a, b = synth.longdress_frame(1), synth.longdress_frame(2)
    refs = [ob.reconstruct(f)[1] for f in (a, b)]
    assert not np.array_equal(refs[0]["block_to_patch"] != 0, refs[1]["block_to_patch"] != 0)
    dev = torch.device("cuda:0")
    d, keep = _abi.host_frame_desc(a)
    slots, staged = [], {0: [], 1: []}

    def plane(k, arr_a, arr_b):
        ta = torch.from_numpy(np.ascontiguousarray(arr_a).view(np.uint8).reshape(-1)).to(dev)
        tb = torch.from_numpy(np.ascontiguousarray(arr_b).view(np.uint8).reshape(-1)).to(dev)
        slots.append(torch.empty_like(ta))
        staged[0].append(ta)
        staged[1].append(tb)
        return slots[-1].data_ptr()

    d.occupancy.y = plane(0, a["occupancy"], b["occupancy"])
    d.occupancy.stride = d.occupancy.width
    for m in range(2):
        d.geometry[m].y = plane(0, a["geometry"][m], b["geometry"][m])
        d.attribute[m].y, d.attribute[m].u, d.attribute[m].v = (plane(0, pa, pb) for pa, pb in zip(a["attribute"][m], b["attribute"][m]))
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    # the patch table is the gof's (host side, read at creation): both frames are reconstructed with frame a's patches, so the
    # reference for the second launch is the oracle on "a's patch table over b's planes"
    mixed = dict(a)
    mixed["occupancy"], mixed["geometry"], mixed["attribute"] = b["occupancy"], b["geometry"], b["attribute"]
    ref_mixed = ob.reconstruct(mixed)[1]
    g = ctx.gof(None, capacity=1_200_000, memory=_abi.VPCC_MEM_DEVICE, descs=[d])     # created BEFORE the planes hold anything
    with torch.cuda.stream(stream):
        for s_, t_ in zip(slots, staged[0]):
            s_.copy_(t_, non_blocking=True)
        g.reconstruct(stream=stream.cuda_stream)
        first = None
    first = g.download(0)
    _check(first, refs[0])
    with torch.cuda.stream(stream):
        for s_, t_ in zip(slots, staged[1]):
            s_.copy_(t_, non_blocking=True)
        g.reconstruct(stream=stream.cuda_stream)
    _check(g.download(0), ref_mixed)
    b2p, items = g.block_to_patch(0, (a["width"] // 16) * (a["height"] // 16))
    assert np.array_equal(b2p.astype(np.uint64), ref_mixed["block_to_patch"].astype(np.uint64))
    assert items == int(np.count_nonzero(ref_mixed["block_to_patch"]))
    g.close()


def test_gof_is_deterministic_and_idempotent(ctx):
    frames = [synth.longdress_frame(i) for i in range(4)]
    g = ctx.gof(frames, capacity=1_000_000)
    g.reconstruct()
    first = [g.download(i) for i in range(4)]
    g.reconstruct()
    for i in range(4):
        again = g.download(i)
        assert again["n"] == first[i]["n"]
        assert np.array_equal(again["xyz"], first[i]["xyz"]) and np.array_equal(again["rgb"], first[i]["rgb"])
    st, ref = ob.reconstruct(frames[3])
    _check(first[3], ref)
    # algorithmic bytes follow SURVEY §8(d): planes + 9 B/point
    assert g.algorithmic_bytes(3) == 112_640 + 7_208_960 + 10_813_440 + 9 * ref["n"]
    g.close()


def test_device_resident_planes(ctx):
    """VPCC_MEM_DEVICE: planes already in HBM (torch is only the allocator here)."""
    import torch
    f = cases.medium_frame(2)
    st, ref = ob.reconstruct(f)
    dev = torch.device("cuda:0")
    d, keep = _abi.host_frame_desc(f)
    t = {}

    def up(a):
        x = torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).to(dev)
        t[len(t)] = x
        return x.data_ptr()

    d.occupancy.y = up(f["occupancy"])
    d.occupancy.stride = d.occupancy.width
    for m in range(2):
        d.geometry[m].y = up(f["geometry"][m])
        d.attribute[m].y, d.attribute[m].u, d.attribute[m].v = (up(p) for p in f["attribute"][m])
    torch.cuda.synchronize()
    g = ctx.gof(None, memory=_abi.VPCC_MEM_DEVICE, descs=[d])
    g.reconstruct(stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    _check(g.download(0), ref)
    g.close()


def test_copied_device_planes():
    """VPCC_GOF_COPY_PLANES: the gof takes a copy of the caller's device planes (into the pool's homes when the context
    has one): the caller's planes may go as soon as the gof exists."""
    import torch
    c = recon.Context(0)
    c.reserve(4)
    distinct = [synth.longdress_frame(i) for i in range(3)]
    refs3 = [ob.reconstruct(f)[1] for f in distinct]
    dev = torch.device("cuda:0")
    keepalive, descs = [], []

    def up(a):
        x = torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).to(dev)
        keepalive.append(x)
        return x.data_ptr()

    for i in range(12):
        f = distinct[i % 3]
        d, keep = _abi.host_frame_desc(f)
        keepalive.append(keep)
        d.occupancy.y = up(f["occupancy"])
        d.occupancy.stride = d.occupancy.width
        for m in range(2):
            d.geometry[m].y = up(f["geometry"][m])
            d.attribute[m].y, d.attribute[m].u, d.attribute[m].v = (up(p) for p in f["attribute"][m])
        descs.append(d)
    torch.cuda.synchronize()
    g = c.gof(None, capacity=1_000_000, memory=_abi.VPCC_MEM_DEVICE, descs=descs)      # borrowed planes
    g.reconstruct()
    for i in (0, 7, 8, 11):
        _check(g.download(i), refs3[i % 3])
    g.close()
    g = c.gof(None, capacity=1_000_000, memory=_abi.VPCC_MEM_DEVICE, descs=descs, flags=_abi.VPCC_GOF_COPY_PLANES)
    for x in keepalive:
        if isinstance(x, torch.Tensor):
            x.zero_()
    torch.cuda.synchronize()
    g.reconstruct()
    for i in (0, 7, 8, 11):
        _check(g.download(i), refs3[i % 3])
    g.close()
    c.close()
def test_random_sweep_against_oracle(ctx):
    """Seeded sweep over canvas sizes, precisions, occupancy value styles, patch statistics and map counts;
    every frame runs through the single-pass kernel (block size 16) in ONE batch and must equal the oracle."""
    frames = cases.random_sweep_frames()
    g = ctx.gof(frames, flags=_abi.VPCC_GOF_PROFILE | _abi.VPCC_GOF_WANT_PATCH_INDEX)
    g.reconstruct()
    assert [k for k, _ in g.kernel_times()] == ["k_plan_tiles", "k_recon_tiles"]
    counts = g.point_counts()
    total = 0
    for i, f in enumerate(frames):
        st, ref = ob.reconstruct(f)
        assert st == 0 and counts[i] == ref["n"], i
        _check(g.download(i, want_patch_index=True), ref)
        total += ref["n"]
    assert total > 500_000
    g.close()
