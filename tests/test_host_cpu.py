"""Host logic that needs no GPU: the C++ Decoder mirror's API contract, the PLY writer, the container,
and the multi-GPU bookkeeping over gloo (world size 2)."""
import os

import numpy as np
import pytest

import cases
from tmc2rs import _abi, container, recon, sharding


def _have_gpu():
    import torch
    return torch.cuda.device_count() > 0


def test_ply_writer_matches_reference_format(tmp_path):
    # src/writer.rs:32-74: header lines and "x y z r g b" body, ASCII
    xyz = np.array([[1, 2, 3], [65535, 0, 7]], np.uint16)
    rgb = np.array([[255, 0, 9], [1, 2, 3]], np.uint8)
    p = tmp_path / "0000.ply"
    recon.write_ply(p, xyz, rgb)
    expected = ("ply\nformat ascii 1.0\nelement vertex 2\nproperty uint x\nproperty uint y\nproperty uint z\n"
                "property uchar red\nproperty uchar green\nproperty uchar blue\nelement face 0\n"
                "property list uint8 int32 vertex_index\nend_header\n1 2 3 255 0 9\n65535 0 7 1 2 3\n")
    assert p.read_text() == expected
    recon.write_ply(p, xyz, None)       # with_colors == false: no colour properties, no colour columns
    assert p.read_text() == ("ply\nformat ascii 1.0\nelement vertex 2\nproperty uint x\nproperty uint y\nproperty uint z\n"
                             "element face 0\nproperty list uint8 int32 vertex_index\nend_header\n1 2 3\n65535 0 7\n")


def test_ply_writer_ascii_every_value(tmp_path):
    """The ASCII writer looks every number up in a table of digits: every u16 as a coordinate, every u8 as a colour, in all
    six columns, and the empty cloud."""
    v = np.arange(65536, dtype=np.uint16)
    xyz = np.stack([v, v[::-1], np.roll(v, 12345)], axis=1)
    rgb = np.stack([(v & 255), (v >> 8), 255 - (v & 255)], axis=1).astype(np.uint8)
    p = tmp_path / "all.ply"
    recon.write_ply(p, xyz, rgb)
    body = p.read_bytes().split(b"end_header\n", 1)[1]
    assert body == "".join(f"{a[0]} {a[1]} {a[2]} {c[0]} {c[1]} {c[2]}\n" for a, c in zip(xyz.tolist(), rgb.tolist())).encode()
    recon.write_ply(p, xyz, None)
    assert p.read_bytes().split(b"end_header\n", 1)[1] == "".join(f"{a[0]} {a[1]} {a[2]}\n" for a in xyz.tolist()).encode()
    recon.write_ply(p, xyz[:0], rgb[:0])
    assert p.read_bytes().endswith(b"end_header\n") and b"element vertex 0\n" in p.read_bytes()


def test_ply_writer_binary_little_endian(tmp_path):
    # the variant the reference's writer keeps commented out (src/writer.rs:10-11, 39-44): same properties
    rng = np.random.default_rng(5)
    xyz = rng.integers(0, 65536, (1000, 3), dtype=np.uint16)
    rgb = rng.integers(0, 256, (1000, 3), dtype=np.uint8)
    p = tmp_path / "b.ply"
    recon.write_ply(p, xyz, rgb, binary=True)
    raw = p.read_bytes()
    head, body = raw.split(b"end_header\n", 1)
    assert head == (b"ply\nformat binary_little_endian 1.0\nelement vertex 1000\nproperty uint x\nproperty uint y\n"
                    b"property uint z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nelement face 0\n"
                    b"property list uint8 int32 vertex_index\n")
    rec = np.frombuffer(body, dtype=np.dtype([("p", "<u4", 3), ("c", "u1", 3)]))
    assert len(rec) == 1000 and np.array_equal(rec["p"], xyz) and np.array_equal(rec["c"], rgb)
    recon.write_ply(p, xyz, None, binary=True)
    body = p.read_bytes().split(b"end_header\n", 1)[1]
    assert np.array_equal(np.frombuffer(body, dtype="<u4").reshape(-1, 3), xyz)


def test_decoder_api_contract_without_gpu(tmp_path):
    path = tmp_path / "a.vpccgof"
    container.write_container(path, [[cases.medium_frame(0)]])
    d = recon.Decoder(path)
    d.start()
    with pytest.raises(recon.VpccError) as e:      # the reference panics: "can only be started once"
        d.start()
    assert e.value.status == _abi.VPCC_ERR_STATE
    if not _have_gpu():
        # no GPU: the worker fails, and — like a worker panic in the reference — the consumer just sees
        # the end of the stream; the product never falls back to a CPU path
        assert d.recv_frame() is None and d.recv_frame() is None
        assert "gfx950" in d.error()
    d.close()


def test_decoder_rejects_garbage(tmp_path):
    p = tmp_path / "bad.vpccgof"
    p.write_bytes(b"not a container at all")
    d = recon.Decoder(p)
    with pytest.raises(recon.VpccError):
        d.start()
    d.close()
    d = recon.Decoder(tmp_path / "missing.vpccgof")
    with pytest.raises(recon.VpccError):           # the reference unwraps the io error on the caller's thread
        d.start()
    d.close()


def test_frame_dealing():
    assert sharding.frames_of_rank(10, 0, 4) == [0, 4, 8]
    assert sharding.frames_of_rank(10, 3, 4) == [3, 7]
    seen = sorted(f for r in range(3) for f in sharding.frames_of_rank(32, r, 3))
    assert seen == list(range(32))
    for f in range(32):
        r, i = sharding.owner_of_frame(f, 3)
        assert sharding.frames_of_rank(32, r, 3)[i] == f


def _gloo_worker(rank, world, port, n_frames, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = sharding.frames_of_rank(n_frames, rank, world)
    local_counts = [1000 * f + 7 for f in mine]             # stand-in for the per-frame point counts of this rank
    counts = sharding.presentation_order_counts(dist, local_counts, n_frames)
    elapsed, points = sharding.job_totals(dist, 0.5 + rank, sum(local_counts))
    # the timed region bench.py runs on every rank (same function, fake step): rank 1 is the slow one
    import time
    ran = []
    reg = sharding.timed_region(step=lambda: (ran.append(1), time.sleep(0.002 * (1 + rank))), sync=lambda: None,
                                steps=3, warmup=2, points_per_step=100 + rank, dist=dist, min_seconds=0.05)
    q.put((rank, counts, elapsed, points, reg, len(ran)))
    dist.destroy_process_group()


def test_multi_rank_bookkeeping_over_gloo():
    import torch.multiprocessing as mp
    world, n_frames = 2, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gloo_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    regs = {}
    for rank, counts, elapsed, points, reg, ran in res:
        assert counts == [1000 * f + 7 for f in range(n_frames)]       # presentation order on every rank
        assert elapsed == 1.5                                           # MAX over ranks
        assert points == sum(1000 * f + 7 for f in range(n_frames))    # SUM over ranks
        assert reg["world"] == 2 and reg["points_total_per_step"] == 201
        assert ran == 2 + reg["steps_effective"]                        # warm-up + exactly K timed steps
        regs[rank] = reg
    # both ranks ran the same K (raised from 3 to cover min_seconds at the FAST rank's pace) and report the
    # slow rank's time
    assert regs[0]["steps_effective"] == regs[1]["steps_effective"] >= 10
    assert regs[0]["elapsed_s"] == regs[1]["elapsed_s"] >= 0.004 * regs[0]["steps_effective"]


def test_timed_region_single_rank():
    import time
    n = []
    reg = sharding.timed_region(step=lambda: (n.append(1), time.sleep(0.001)), sync=lambda: None, steps=5, warmup=1,
                                points_per_step=42)
    assert reg["steps_effective"] == 5 and len(n) == 6 and reg["points_total_per_step"] == 42 and reg["world"] == 1
    assert reg["elapsed_s"] >= 0.005


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2 ...` outside any launcher (the shape of the driver's 1-GPU command at N = 2): bench.py starts
    its ranks itself — torch.distributed.run as a CHILD process, before torch or the GPU are touched — and passes rank 0's
    ONE JSON line and the exit status through.  --dry-run: process group (gloo), the timed region's barriers, K agreed by
    all ranks, MAX of the time, SUM of the points, MIN of the verification flag, with a step that only sleeps."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["VPCC_BENCH_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--dry-run"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dry_run"] and d["verified"] and d["steps"] == 4 and d["warmup"] == 2
    assert d["points_total_per_step"] == 3000                       # SUM over the two ranks (1000 + 2000)
    assert d["ms_per_step"] >= 2.0                                   # MAX over ranks: rank 1 sleeps 2 ms per step
    # a rank count that does not match the launcher's is refused, not silently run
    env2 = dict(env, WORLD_SIZE="3", RANK="0")
    out = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True,
                         timeout=120, env=env2)
    assert out.returncode == 2
