"""The C++ Decoder mirror end to end on the GPU: container -> start() -> frames in presentation order."""
import numpy as np
import pytest

import cases
import oracle_binding as ob
from tmc2rs import container, recon, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("devices", [(0,), (0, 0)])     # (0, 0): two contexts -> the sharding / re-sequencing path
def test_decoder_streams_frames_in_order(tmp_path, devices):
    gofs = [[cases.medium_frame(i) for i in range(3)],
            [cases.medium_frame(10 + i, occupancy_values="random") for i in range(2)] + [synth.small_frame(0)]]
    path = tmp_path / "s.vpccgof"
    container.write_container(path, gofs)
    d = recon.Decoder(path, devices=devices)
    d.start()
    frames = list(d)
    assert d.error() == ""
    expected = [f for g in gofs for f in g]
    assert len(frames) == len(expected)
    for got, f in zip(frames, expected):
        st, ref = ob.reconstruct(f)
        assert st == 0 and got["n"] == ref["n"]
        assert np.array_equal(got["xyz"], ob.xyz_array(ref)) and np.array_equal(got["rgb"], ob.rgb_array(ref))
    assert d.recv_frame() is None                       # None forever after the last frame
    d.close()


def test_decoder_error_ends_stream_early(tmp_path):
    bad = cases.medium_frame(0)
    bad["patches"] = bad["patches"].copy()
    bad["patches"]["u0"][0] = 10_000                    # patch outside the canvas: assert in the reference
    path = tmp_path / "bad.vpccgof"
    container.write_container(path, [[cases.medium_frame(1)], [bad]])
    d = recon.Decoder(path)
    d.start()
    frames = list(d)
    assert len(frames) == 1                             # first GOF delivered, then the stream just ends
    assert "canvas" in d.error()
    d.close()


def test_decoder_on_a_v3c_stream_with_raw_decoded_video(tmp_path):
    """.bin parsed by the C++ syntax parser + externally 'decoded' planar videos -> the same points as the
    oracle on the frames the stream was written from (tests/v3c_writer.py)."""
    import v3c_writer as W
    gofs = [[cases.medium_frame(i) for i in range(3)], [cases.medium_frame(40 + i, occupancy_values="random") for i in range(2)]]
    paths = W.write_sequence(tmp_path, gofs)
    d = recon.Decoder(paths["bin"], occupancy_yuv=paths["occ"], geometry_yuv=paths["geo"], attribute_yuv=paths["attr"])
    d.start()
    frames = list(d)
    assert d.error() == ""
    expected = [f for g in gofs for f in g]
    assert len(frames) == len(expected)
    for got, f in zip(frames, expected):
        st, ref = ob.reconstruct(f)
        assert st == 0 and got["n"] == ref["n"]
        assert np.array_equal(got["xyz"], ob.xyz_array(ref)) and np.array_equal(got["rgb"], ob.rgb_array(ref))
    d.close()


def test_decoder_applies_the_smoothing_filters(tmp_path):
    """The reference's post-processing switches (apply_geo_smoothing_type / apply_attr_smoothing_type, src/lib.rs:45-46):
    off, the stream is delivered as reconstructed whatever SEI it carries; on, a GOF WITH a geometry-smoothing SEI is
    filtered with the SEI's grid size and threshold (src/decoder.rs:291-299) and one without is not; colour smoothing
    follows its own switch.  Expected output: oracle reconstruction + oracle/vpcc_smoothing_spec.c."""
    import v3c_writer as W
    gofs = [[cases.overlapping_3d_frame(i) for i in range(2)], [cases.overlapping_3d_frame(2 + i) for i in range(2)],
            [cases.overlapping_3d_frame(4)]]
    seis = [(8, 2), None, (4, 1)]
    paths = W.write_sequence(tmp_path, gofs, seis=seis)
    kw = dict(occupancy_yuv=paths["occ"], geometry_yuv=paths["geo"], attribute_yuv=paths["attr"])
    expected = [(f, sei) for g, sei in zip(gofs, seis) for f in g]

    def run(geometry, color):
        d = recon.Decoder(paths["bin"], **kw)
        if geometry or color:
            # (grid_size / threshold are for inputs WITHOUT syntax: a V3C GOF that carries no SEI — the second one — must stay
            # unfiltered although they are given)
            d.set_smoothing(geometry=geometry, color=color, grid_size=6, threshold=3, color_grid_size=8, color_threshold_smoothing=10,
                            color_threshold_difference=100)
        d.start()
        frames = list(d)
        assert d.error() == "" and len(frames) == len(expected)
        d.close()
        return frames

    moved = 0
    for geometry, color in ((False, False), (True, False), (True, True), (False, True)):
        for got, (f, sei) in zip(run(geometry, color), expected):
            st, ref = ob.reconstruct(f)
            xyz, rgb, part = ob.xyz_array(ref), ob.rgb_array(ref), ref["partition"].astype(np.uint16)
            if geometry and sei:
                xs = ob.spec_smooth_geometry(xyz, part, 10, sei[0], sei[1])
                moved += int(np.any(xs != xyz))
                xyz = xs
            if color:
                rgb = ob.spec_smooth_color(xyz, rgb, part, 10, 8, 10, 100)
            assert got["n"] == ref["n"] and np.array_equal(got["xyz"], xyz) and np.array_equal(got["rgb"], rgb), (geometry, color, sei)
    assert moved >= 4                                   # the SEI GOFs really were filtered

    # a container has no syntax: geometry smoothing with the parameters given
    path = tmp_path / "c.vpccgof"
    container.write_container(path, gofs[:1])
    d = recon.Decoder(path)
    d.set_smoothing(geometry=True, grid_size=8, threshold=2)
    d.start()
    for got, f in zip(list(d), gofs[0]):
        st, ref = ob.reconstruct(f)
        part = ref["partition"].astype(np.uint16)
        assert np.array_equal(got["xyz"], ob.spec_smooth_geometry(ob.xyz_array(ref), part, 10, 8, 2))
        assert np.array_equal(got["rgb"], ob.rgb_array(ref))
    with pytest.raises(recon.VpccError):
        d.set_smoothing(geometry=False)                 # after start
    d.close()


def test_decoder_v3c_short_video_and_unsupported_stream(tmp_path):
    import v3c_writer as W
    gofs = [[cases.medium_frame(i) for i in range(2)]]
    paths = W.write_sequence(tmp_path, gofs)
    with open(paths["geo"], "r+b") as f:
        f.truncate(1000)
    d = recon.Decoder(paths["bin"], occupancy_yuv=paths["occ"], geometry_yuv=paths["geo"], attribute_yuv=paths["attr"])
    with pytest.raises(recon.VpccError) as e:
        d.start()
    assert "geometry video shorter" in str(e.value)
    d.close()


def test_long_multi_sequence_stream_in_order(tmp_path):
    """BASELINE configs 2/3 in miniature: sequences that cycle a set of distinct frames, several sequences back to
    back, GOF after GOF through the sharded decoder — every frame must arrive, in presentation order, with the
    oracle's content (checksum per frame and a checksum of the checksums)."""
    import zlib
    distinct = [cases.medium_frame(60 + i, occupancy_values="random" if i % 2 else "one") for i in range(8)]
    ref = []
    for f in distinct:
        st, r = ob.reconstruct(f)
        assert st == 0
        ref.append(zlib.crc32(ob.rgb_array(r).tobytes(), zlib.crc32(ob.xyz_array(r).tobytes())))
    order = [(s * 3 + i) % 8 for s in range(3) for i in range(60)]          # three 60-frame sequences, different phases
    gofs = [[distinct[k] for k in order[g:g + 12]] for g in range(0, len(order), 12)]
    path = tmp_path / "long.vpccgof"
    container.write_container(path, gofs)
    d = recon.Decoder(path, devices=(0, 0))
    d.start()
    got = [zlib.crc32(fr["rgb"].tobytes(), zlib.crc32(fr["xyz"].tobytes())) for fr in d]
    assert d.error() == ""
    assert got == [ref[k] for k in order]
    assert zlib.crc32(np.array(got, np.uint32).tobytes()) == zlib.crc32(np.array([ref[k] for k in order], np.uint32).tobytes())
    d.close()


def test_ctx_bind_thread_keeps_the_thread_inside_its_allowed_cpus():
    """vpcc_ctx_bind_thread: the calling thread ends up on the CPUs of the GPU's NUMA node (or stays where it is when
    the platform reports none) — never outside what the process may use; run on a thread of its own, like a lane."""
    import ctypes as C
    import os
    import threading
    from tmc2rs import _abi
    lib = _abi.load_library()
    out = {}

    def lane():
        before = os.sched_getaffinity(0)
        ctx = C.c_void_p()
        assert lib.vpcc_ctx_create(0, C.byref(ctx)) == 0
        node = C.c_int(-7)
        out["status"] = lib.vpcc_ctx_bind_thread(ctx, C.byref(node))
        out["node"] = node.value
        out["before"], out["after"] = before, os.sched_getaffinity(0)
        lib.vpcc_ctx_destroy(ctx)

    t = threading.Thread(target=lane)
    t.start()
    t.join()
    assert out["status"] == 0 and out["node"] >= -1
    assert out["after"] and out["after"] <= out["before"]
    if out["node"] == -1:
        assert out["after"] == out["before"]
