#!/usr/bin/env python3
"""bench.py — reconstruction throughput of the MI355X-native V-PCC hot path.

A "step" is one pass of the hot path over one batch: ONE launch over four GOFs of 32 distinct synthetic
S-longdress frames each (128 frames; 1280x1408 geometry+attribute, 320x352 occupancy, ~800 k points/frame —
BASELINE.json configs[1]), all decoded planes — in the RASTER layout a video decoder hands over — and
patch tables already resident in HBM when the timed region starts; every kernel between those planes and the points is
inside the region (`roofline.all_kernels_ms`): k_plan_tiles — block_to_patch and the work list, the device port of
generate_block_to_patch_from_occupancy_map_video, src/codec.rs:205-250, built by EVERY launch from the occupancy as it lies —
and k_recon_tiles; nothing is re-arranged.  `roofline.frac` is the dominant kernel's, `roofline.path_frac` the same bytes over
the sum of all kernels of the step.  The gof's blocks are allocated from the context's pool (vpcc_ctx_reserve, `config.pool`):
the policy for callers whose planes are in HBM already — the streaming Decoder, whose kernels are under 1 % of its wall
time, does without it by default.  Frames are independent, so with N GPUs every rank reconstructs its own batch (weak
scaling, no data-path collective); `value` is the whole-job Mpoints/s.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python bench.py --gpus N --steps K --warmup W        # starts its N ranks itself (torch.distributed.run, one per GPU, RCCL)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W     # ... or is started as one of them

The timed region is tmc2rs.sharding.timed_region (the function the world-size-2 gloo test runs): W warm-up
steps, then K steps between barrier + synchronisation pairs; K is raised to `steps_effective` until the
region lasts >= --min-seconds (a step is ~0.1 ms).  Every 16th launch of the region carries a HIP-event pair
on the launch stream (profile mode of the library; a pair on EVERY launch would add ~10 us of stream time per
step), so `roofline.kernel_ms` is the mean duration of launches `ms_per_step` was measured on.

One JSON line is printed by rank 0, with the contract fields plus
  roofline        : HBM roofline of the dominant kernel.  `achieved`/`frac` use the ALGORITHMIC bytes of
                    SURVEY §8d (whole planes once + 9 B/point) — a rate of bytes the path stands for, NOT a
                    physical HBM rate; the physical one is `frac_traffic` (memory-side bytes of the committed
                    rocprofv3 PMC passes / kernel time; `traffic_stale` says whether those passes measured
                    the kernel sources this run was built from).  Next to it: `necessary_bytes` (only the
                    16x16 blocks a patch owns and that hold occupancy, + occupancy plane + 9 B/point),
                    `line_floor_bytes` (distinct 128-B lines of the raster planes holding needed samples:
                    the floor of any kernel reading this layout).
  library         : the shared object this run loaded (name, sha16) and the sha of the kernel sources + flags.
  verified_frames : EVERY entry of the timed batch is downloaded after the region and compared (count, xyz,
                    rgb by CRC) with the CPU oracle's output for its frame — the run fails if one differs.
  other_configs   : BASELINE configs 5 (S-owlii, 2048x2048) and 4 (S-longdress + grid smoothing, own spec), N=1
                    only, each a short timed loop over 128-frame launches with its own verification.
  fresh_gof       : what the product pays ONCE PER GOF, none of which the headline's relaunches contain: three distinct
                    128-frame batches of raster planes resident in HBM (VPCC_MEM_DEVICE, in the pool's homes), every step =
                    vpcc_gof_create + vpcc_gof_reconstruct on the next batch + vpcc_gof_destroy of the one before last —
                    validation, the host's O(patches) records, descriptors, planning and reconstruction kernels.  N=1 only.
  end_to_end      : host-buffer (PCIe-inclusive) rate through the C++ Decoder (pinned container -> H2D ->
                    kernels -> D2H -> consumer); never `value`.  N=1 only.  Measured COLD: it is the first thing this
                    process does on the GPU (no context, no pool, no warm allocator); `end_to_end.warm` is a second
                    Decoder at the end of the run.
  cpu_baseline    : the CPU oracle (a port of the reference's algorithm; the Rust crate cannot be built
                    here) timed single-threaded on this box, rank 0 at N=1 only.
"""
import argparse
import json
import os
import sys
import time
import zlib

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "tmc2-rs_amd"))

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy rate


def stale(traffic, family):
    """Was the committed traffic figure measured on other kernel sources than the ones this run was built from?  The
    family's own sources count where the file records them (a smoothing change does not age the tile kernel's figure)."""
    from tmc2rs import provenance
    if traffic.get("family_source_sha16"):
        return traffic["family_source_sha16"] != provenance.kernel_source_sha16(family)
    return traffic.get("kernel_source_sha16") != provenance.kernel_source_sha16()


def measured_traffic(kernel, workload, frames):
    """Memory-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (latest profiles/rNN/traffic*.json whose kernel / workload / frame count match), or None."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r*", "traffic*.json"))):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if kernel.split("<")[0] in d.get("kernel", "") and workload in d.get("workload", "") \
                and f"{frames} frames" in d.get("workload", ""):
            best = d
    return best


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--min-seconds", type=float, default=0.5,
                    help="the timed region runs max(--steps, enough steps to last this long)")
    ap.add_argument("--ramp-ms", type=float, default=150.0,
                    help="untimed launches before the warmup steps until this much wall time has passed: a step is ~0.1 ms, "
                         "far shorter than the GPU's clock ramp from idle")
    ap.add_argument("--frames", type=int, default=32, help="distinct synthetic frames (per rank): one GOF of the sequence")
    ap.add_argument("--cycles", type=int, default=4,
                    help="a step reconstructs the --frames distinct frames this many times over, each copy in device buffers "
                         "of its own, in ONE launch (default 4 x 32 = 128 frames, 3.7 GB: a 300-frame job is two or three such "
                         "launches).  Larger launches amortise the kernel's ramp and tail: 4.4 us per frame at 32 frames per "
                         "launch, 3.9 at 128")
    ap.add_argument("--gofs", type=int, default=1,
                    help="resident copies of the step's batch that consecutive steps rotate over.  A launch must not find the "
                         "previous launch's planes in the 256 MB Infinity Cache (SURVEY 8d): the default batch reads 1.2 GB per "
                         "launch, so one copy is enough; a 32-frame batch (--cycles 1) reads 0.29 GB and needs --gofs 3")
    ap.add_argument("--workload", default="longdress", choices=["longdress", "owlii"])
    ap.add_argument("--general", action="store_true", help="force the general kernel sequence")
    ap.add_argument("--smooth", action="store_true",
                    help="BASELINE config 4: grid geometry + colour smoothing after reconstruction (own spec, see DESIGN.md)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gpu-state", action="store_true", help="skip the rocm-smi reading of clocks and power under load")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the S-owlii and smoothing legs (`other_configs`)")
    ap.add_argument("--pool-gib", type=int, default=32,
                    help="vpcc_ctx_reserve: the context's pool (two homes in the two kinds of VRAM regions), as the streaming "
                         "Decoder's lanes take it (VPCC_DECODER_POOL_GIB, default 32).  0: every gof block is a hipMalloc of its own")
    ap.add_argument("--diag", action="store_true",
                    help="allow the diagnostic library (VPCC_DIAG_LIB=1, tools/ only): its timings are not the product's")
    ap.add_argument("--no-compare", action="store_true",
                    help="skip the `launches_of_one_gof` leg (32-frame launches, the step of round 1): its ~1 000 short "
                         "launches of the same kernel would blur a rocprofv3 average of the run")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle check of the timed GOF's output (tools/ only)")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the PCIe-inclusive Decoder run")
    ap.add_argument("--no-warm-e2e", action="store_true", help="skip the second (warm) Decoder run at the end")
    ap.add_argument("--e2e-gofs", type=int, default=17, help="GOFs in the end-to-end container (launches of 1 + 4 + 4 + 4 + 2 + 1 + 1 GOFs: the stream's last unit is dealt out in halves)")
    ap.add_argument("--profile-every", type=int, default=16,
                    help="an event pair around every n-th launch of the timed region (each costs a few us of stream time)")
    ap.add_argument("--profile-steps", type=int, default=0, help=argparse.SUPPRESS)   # accepted for old tool scripts
    ap.add_argument("--no-fresh-gof", action="store_true", help="skip the `fresh_gof` leg (create + reconstruct + destroy per step)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: the launch of the ranks, the process group (VPCC_BENCH_BACKEND, default gloo here), the timed "
                         "region's barriers and its MAX / SUM / MIN reductions with a step that only sleeps — what `--gpus N` "
                         "does around the kernels, for the CPU test of that path")
    return ap.parse_args()


def smoothing_algorithmic_bytes(gof, n_frames, bitdepth, grid, cgrid):
    """Algorithmic bytes of one smoothing pass pair over the GOF: per point read
    xyz 6 + rgb 3 + patch index 2 B and write 9 B; per occupied grid cell 24 B written and read once,
    for the geometry grid and for the colour grid (the verdict's figure; the kernels' cells are 32 B)."""
    import numpy as np
    total = 0
    for i in range(n_frames):
        r = gof.download(i)
        n = r["n"]
        total += n * (6 + 3 + 2 + 9)
        for g in (grid, cgrid):
            if not g:
                continue
            w = ((1 << bitdepth) + g - 1) // g
            c = r["xyz"].astype(np.int64) // g
            cells = np.unique((c[:, 0] * w + c[:, 1]) * w + c[:, 2]).size
            total += cells * 24 * 2
    return total


def end_to_end_leg(frames, args, device):
    """Host buffers in, host buffers out: the C++ Decoder on a container of --e2e-gofs copies of the GOF (written to /dev/shm)."""
    import tempfile
    from tmc2rs import container, recon
    d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    path = os.path.join(d, "bench_e2e.vpccgof")
    try:
        container.write_container(path, [frames] * args.e2e_gofs)
        size = os.path.getsize(path)
        dec = recon.Decoder(path, devices=(device,))
        t_call = time.perf_counter()
        dec.start()                       # reads the container on the caller's thread (like the reference, src/lib.rs:98); the
        t_call = time.perf_counter() - t_call   # lanes' contexts — the runtime's initialisation in a cold process — are made beside it
        nf, npts, sec = dec.drain()
        t_first = dec.first_frame_seconds()
        dstats = dec.stats()
        dec.close()
        # Two bounds of the steady rate: the whole run includes the start-up (contexts, page-locking, first GOF);
        # the rate after the first frame profits from the uploads of the next units that were already running
        # during the start-up (two units of look-ahead).
        after = (sec - t_first) / max(nf - 1, 1)
        return {"_frames": nf, "_points": npts,
                "whole_run_frames_per_s": round(nf / sec, 1), "after_first_frame_frames_per_s": round(1.0 / after, 1),
                "whole_run_Mpoints_per_s": round(npts / sec / 1e6, 1),
                "h2d_GBps_whole_run": round(size / sec / 1e9, 2), "d2h_GBps_whole_run": round(npts * 9 / sec / 1e9, 2),
                "startup_s": round(t_first, 3),
                "start_call_s": round(t_call, 3),
                # the product's own launches: GOF 0 alone, then every resident run of up to four GOFs in ONE launch
                "launches": dstats["launches"], "max_frames_per_launch": dstats["max_frames_per_launch"],
                "kernel_seconds": round(dstats["kernel_seconds"], 6),
                "kernels_share_of_wall": round(dstats["kernel_seconds"] / sec, 4),
                # host work of a unit on its lane's thread — validation, the O(patches) records, descriptors, the enqueue of
                # ingest and launch — summed over units, per frame: it runs while the previous unit's transfers do
                "host_plan_and_enqueue_us_per_frame": round(dstats["launch_seconds"] / max(nf, 1) * 1e6, 1),
                "lane_numa_nodes": dstats["numa_node"],
                "pool": os.environ.get("VPCC_DECODER_POOL_GIB", "none (the Decoder's default: its kernels are under 1 % of its wall time)"),
                "sample": f"{args.e2e_gofs} GOFs x {args.frames} frames through tmc2rs::Decoder (pinned container -> "
                          f"H2D -> kernels -> D2H -> consumer); the rate after the first frame still contains the stream's end, "
                          f"where the last GOF's results travel back alone"}
    finally:
        if os.path.exists(path):
            os.remove(path)
        os.rmdir(d)


def launch_ranks(n):
    """`python bench.py --gpus N` outside a launcher: start the N ranks — `python -m torch.distributed.run`, one process per
    GPU on this node, rendezvous on 127.0.0.1 at a free port — as a CHILD of this process, which has touched neither torch
    nor the GPU (never a re-exec), pass rank 0's JSON line through and return the launcher's exit status."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # RCCL between processes needs dmabuf IPC on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def dry_run(args, rank, world):
    """The rank path without a GPU: process group, timed region (barriers, K identical on every rank, MAX of the time, SUM
    of the points), the MIN reduction of the verification flag, one JSON line from rank 0."""
    import torch
    from tmc2rs import sharding
    backend = os.environ.get("VPCC_BENCH_BACKEND", "gloo")
    dist = None
    if world > 1 or os.environ.get("VPCC_BENCH_FORCE_DIST"):
        import torch.distributed as dist
        dist.init_process_group(backend)
    points = 1000 * (rank + 1)
    reg = sharding.timed_region(lambda: time.sleep(0.001 * (1 + rank)), lambda: None, args.steps, args.warmup, points, dist=dist,
                                device="cpu", min_seconds=min(args.min_seconds, 0.05))
    ok = 1
    if dist is not None:
        v = torch.tensor([ok], dtype=torch.int64)
        dist.all_reduce(v, op=dist.ReduceOp.MIN)
        ok = int(v.item())
    if rank == 0:
        print(json.dumps({"metric": "dry run: no kernel was launched", "value": round(reg["points_total_per_step"] * reg["steps_effective"] / reg["elapsed_s"] / 1e6, 3),
                          "unit": "Mpoints/s", "n_gpus": world, "steps": args.steps, "steps_effective": reg["steps_effective"], "warmup": args.warmup,
                          "ms_per_step": round(reg["elapsed_s"] / reg["steps_effective"] * 1e3, 4), "points_total_per_step": reg["points_total_per_step"],
                          "verified": bool(ok), "dry_run": True, "backend": backend}))
    if dist is not None:
        dist.destroy_process_group()
    return 0 if ok else 3


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    if args.dry_run:
        sys.exit(dry_run(args, rank, world))

    import numpy as np
    import torch
    from tmc2rs import _abi, provenance, recon, sharding, synth, traffic
    lib_info = provenance.library()
    if lib_info["diagnostic"] and not args.diag:
        print("bench.py: the diagnostic library is loaded (VPCC_DIAG_LIB=1); pass --diag to time it anyway", file=sys.stderr)
        sys.exit(2)

    # VPCC_BENCH_BACKEND=gloo is a REHEARSAL switch for boxes with fewer GPUs than ranks (ranks then share
    # devices and the scalar reductions run on the CPU); measurements use the default, RCCL.
    backend = os.environ.get("VPCC_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    red_dev = "cuda" if backend == "nccl" else "cpu"
    torch.cuda.set_device(local_rank)
    dist = None
    # (VPCC_BENCH_FORCE_DIST=1: a one-rank job joins a process group too — the RCCL bookkeeping of the N > 1 path, its
    # barriers and reductions, rehearsed on a box with one GPU)
    if world > 1 or os.environ.get("VPCC_BENCH_FORCE_DIST"):
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend)

    # ---- synthetic GOF of this rank (32 distinct frames, ~0.8 GB of planes and outputs) -----------------------
    make = synth.longdress_frame if args.workload == "longdress" else synth.owlii_frame
    kw = {}
    if os.environ.get("VPCC_BENCH_SWAP_PROB"):          # diagnostic only: share of Swap-oriented patches
        kw["swap_prob"] = float(os.environ["VPCC_BENCH_SWAP_PROB"])
    if os.environ.get("VPCC_BENCH_ALIGN"):              # diagnostic only: patch x positions / widths in multiples of N blocks
        kw["align"] = int(os.environ["VPCC_BENCH_ALIGN"])
    frames = [make(rank * args.frames + i, **kw) for i in range(args.frames)]
    cap = 1_000_000 if args.workload == "longdress" else 2_400_000

    # ---- end to end FIRST, cold: the first thing this process does on the GPU (no context, no pool to take over, no warm
    # allocator) — what a user who starts a Decoder gets.  Checked against the point counts further down.
    e2e = None
    if rank == 0 and world == 1 and not args.no_end_to_end and not args.smooth and not args.general:
        e2e = end_to_end_leg(frames, args, local_rank)
        e2e["cold"] = True

    ctx = recon.Context(local_rank)
    # the pool: the allocation policy for callers whose planes are in HBM (this bench's legs; the Decoder does without)
    pool_info = ctx.reserve(args.pool_gib) if args.pool_gib >= 2 else None
    flags = (_abi.VPCC_GOF_FORCE_GENERAL if args.general else 0) | _abi.VPCC_GOF_PROFILE
    if args.smooth:
        flags |= _abi.VPCC_GOF_WANT_PATCH_INDEX
    bitdepth = 10 if args.workload == "longdress" else 11
    smooth_kw = dict(grid_size=8, threshold=4, color_grid_size=8, color_threshold_smoothing=10,
                     color_threshold_difference=100)
    # H2D happens here, outside the timed region.  The steps ROTATE over --gofs resident copies of the GOF (2.4 GB for
    # three S-longdress GOFs): with one GOF reconstructed again and again a launch finds part of the planes the
    # previous launch read (290 MB, about the size of the Infinity Cache) still on chip and runs 15-18 % faster than
    # any real stream of GOFs would (tools/two_gofs.py) — that rate is reported as `repeat_one_gof`, never as `value`.
    batch = frames * max(args.cycles, 1)                      # the same host planes, uploaded once per copy
    n_batch = len(batch)
    gofs = [ctx.gof(batch, capacity=cap, flags=flags) for _ in range(max(args.gofs, 1))]
    for g_ in gofs:
        g_.profile_interval(args.profile_every)
    gof = gofs[0]
    turn = [0]

    def sync_all():
        for g_ in gofs:
            g_.sync()

    def step():
        g_ = gofs[turn[0] % len(gofs)]
        turn[0] += 1
        g_.reconstruct()
        if args.smooth:
            g_.smooth(bitdepth, **smooth_kw)

    gof.reconstruct()
    counts = gof.point_counts().astype(np.int64)
    assert all(gof.frame_status(i) == 0 for i in range(n_batch)), "capacity too small"
    points_per_step = int(counts.sum())
    alg_bytes = sum(gof.algorithmic_bytes(i) for i in range(n_batch))
    smooth_bytes = smoothing_algorithmic_bytes(gof, args.frames, bitdepth, smooth_kw["grid_size"],
                                               smooth_kw["color_grid_size"]) * max(args.cycles, 1) if args.smooth and rank == 0 else 0

    t_ramp = time.perf_counter()
    while (time.perf_counter() - t_ramp) * 1e3 < args.ramp_ms:     # untimed: brings the clocks up from idle
        for _ in range(16):
            step()
        sync_all()

    # HIP events on the stream the kernels are launched on, bracketing the timed region as a whole
    # (torch.cuda.Event on torch's current stream would not see this stream)
    ext = torch.cuda.ExternalStream(ctx.stream(), device=torch.device("cuda", local_rank))
    ev = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]
    n_steps = [0]

    def timed_step():
        if n_steps[0] == args.warmup:
            ev[0].record(ext)                                       # first timed step
        n_steps[0] += 1
        step()

    reg = sharding.timed_region(timed_step, sync_all, args.steps, args.warmup, points_per_step, dist=dist,
                                device=red_dev, min_seconds=args.min_seconds, device_sync=torch.cuda.synchronize)
    ev[1].record(ext)
    ev[1].synchronize()
    steps_eff = reg["steps_effective"]
    elapsed = reg["elapsed_s"]
    region_ms_per_step = ev[0].elapsed_time(ev[1]) / steps_eff
    # the timed region's own launches: every --profile-every-th of them carries an event pair
    kernels, launches_averaged = {}, 0
    for g_ in gofs:
        k_, n_ = g_.kernel_time_means(min(max(steps_eff // len(gofs) // args.profile_every, 1), 512))
        for name, ms in k_.items():
            kernels[name] = kernels.get(name, 0.0) + ms * n_
        launches_averaged += n_
    kernels = {name: v / max(launches_averaged, 1) for name, v in kernels.items()}

    # Clocks and power WHILE the kernel runs (rank 0, best effort): boxes and runs of one build differ by up to 10 %
    # in `ms_per_step`; this says which state a run was measured in.  A short untimed burst of the same launches keeps
    # the GPU busy while rocm-smi reads its sensors.
    gpu_state = None
    if rank == 0 and not args.no_gpu_state:
        try:
            import re
            import subprocess
            for _ in range(1500):
                step()
            txt = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showbus", "--showtemp",
                                  "--showcomputepartition", "--showmemorypartition"], capture_output=True, text=True, timeout=20).stdout
            sync_all()
            gpu_state = {}
            for name in ("sclk", "mclk", "fclk"):
                m = re.search(r"GPU\[%d\]\s*:\s*%s clock level: \S+ \((\d+)Mhz\)" % (0, name), txt)
                if m:
                    gpu_state[name + "_MHz"] = int(m.group(1))
            m = re.search(r"Power \(W\):\s*([0-9.]+)", txt)
            if m:
                gpu_state["socket_power_W"] = float(m.group(1))
            for key, pat in (("pci_bus", r"PCI Bus:\s*(\S+)"), ("compute_partition", r"Compute Partition:\s*(\S+)"),
                             ("memory_partition", r"Memory Partition:\s*(\S+)")):
                m = re.search(pat, txt)
                if m:
                    gpu_state[key] = m.group(1)
            for key, pat in (("junction_C", r"Temperature \(Sensor junction\) \(C\):\s*([0-9.]+)"),
                             ("memory_C", r"Temperature \(Sensor (?:memory|HBM \d)\) \(C\):\s*([0-9.]+)")):
                m = re.search(pat, txt)
                if m:
                    gpu_state[key] = float(m.group(1))
            gpu_state["note"] = "rocm-smi while an untimed burst of the same launches runs, right after the timed region"
            # what this GPU's memory system gives a plain copy (1 GiB read + 1 GiB written per copy): a reference for
            # the box, next to which the kernel's rate can be read (GPUs of this pool differ by 10 % here as well)
            dev_ = torch.device("cuda", local_rank)
            a_ = torch.empty(1 << 30, dtype=torch.uint8, device=dev_)
            b_ = torch.empty_like(a_)
            e0_, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(3):
                b_.copy_(a_)
            e0_.record()
            for _ in range(10):
                b_.copy_(a_)
            e1_.record()
            e1_.synchronize()
            gpu_state["copy_1GiB_GBps_read_plus_write"] = round(10 * 2 * (1 << 30) / (e0_.elapsed_time(e1_) * 1e-3) / 1e9, 0)
            del a_, b_
        except Exception as e:                                   # sensors are evidence, not a dependency
            gpu_state = {"error": str(e)[:200]}
            sync_all()

    # Launches of ONE 32-frame GOF, the step of round 1, for comparison (rank 0, never `value`): over three copies in
    # rotation (an HBM rate) and on one copy again and again (how round 1 was timed; part of its planes stays in the
    # Infinity Cache).
    one_gof = None
    if rank == 0 and world == 1 and not args.smooth and not args.no_compare and max(args.cycles, 1) > 1:
        small = [ctx.gof(frames, capacity=cap, flags=flags & ~_abi.VPCC_GOF_PROFILE) for _ in range(3)]

        def leg(seq, launches=384):
            for k_ in range(48):
                seq[k_ % len(seq)].reconstruct()
            for g_ in seq:
                g_.sync()
            t_ = time.perf_counter()
            for k_ in range(launches):
                seq[k_ % len(seq)].reconstruct()
            for g_ in seq:
                g_.sync()
            return (time.perf_counter() - t_) / launches

        pts32 = int(counts[:args.frames].sum())
        alg32 = sum(gof.algorithmic_bytes(i) for i in range(args.frames))
        t_rot, t_one = leg(small), leg(small[:1])
        one_gof = {"frames_per_launch": args.frames,
                   "rotating_over_3": {"ms_per_launch": round(t_rot * 1e3, 4), "Mpoints_per_s": round(pts32 / t_rot / 1e6, 1),
                                       "frac": round(alg32 / t_rot / 1e9 / HBM_PEAK_GBPS, 4)},
                   "one_copy_repeated": {"ms_per_launch": round(t_one * 1e3, 4), "Mpoints_per_s": round(pts32 / t_one / 1e6, 1),
                                         "frac_if_it_were_hbm": round(alg32 / t_one / 1e9 / HBM_PEAK_GBPS, 4),
                                         "note": "how round 1 was timed (0.1467 ms, frac 0.677); not an HBM rate: part of the "
                                                 "0.29 GB a launch reads stays in the 256 MB Infinity Cache"}}
        for g_ in small:
            g_.close()

    # the same launches on ONE batch over and over, when the steps rotate over several (--gofs > 1)
    repeat = None
    if len(gofs) > 1 and rank == 0:
        for _ in range(32):
            gofs[0].reconstruct()
            if args.smooth:
                gofs[0].smooth(bitdepth, **smooth_kw)
        gofs[0].sync()
        t_rep, n_rep = time.perf_counter(), 0
        while n_rep < 256:                                  # short: these launches also end up in a rocprof average of the run
            for _ in range(32):
                gofs[0].reconstruct()
                if args.smooth:
                    gofs[0].smooth(bitdepth, **smooth_kw)
            gofs[0].sync()
            n_rep += 32
        t_rep = time.perf_counter() - t_rep
        repeat = {"ms_per_step": round(t_rep / n_rep * 1e3, 4), "Mpoints_per_s": round(points_per_step * n_rep / t_rep / 1e6, 1),
                  "steps": n_rep,
                  "frac_if_it_were_hbm": None if args.smooth else round(alg_bytes / (t_rep / n_rep) / 1e9 / HBM_PEAK_GBPS, 4),
                  "note": "one batch reconstructed again and again instead of --gofs batches in rotation"}

    # ---- the timed batch's output against the CPU oracle (checker only) ---------------------------
    # EVERY entry of the batch (and of every batch in rotation) is downloaded and compared — count and the CRCs of
    # xyz and rgb — with the oracle's output for its frame (entry i is a copy of frame i % --frames).
    verified, ok = [], 1
    if not args.no_verify and not args.smooth:
        sys.path.insert(0, os.path.join(REPO, "tests"))
        import oracle_binding as ob            # checker/baseline only — never on the product path
        ok = 1
        for g_ in gofs:
            ok &= int(np.array_equal(g_.point_counts().astype(np.int64), counts))
        ref_crc = []
        for i in range(args.frames):
            st, ref = ob.reconstruct(frames[i])
            ref_crc.append((st, ref["n"], zlib.crc32(ob.xyz_array(ref).tobytes()), zlib.crc32(ob.rgb_array(ref).tobytes())))
        bad = []
        for gi, g_ in enumerate(gofs):                          # every batch of the rotation was written by timed launches
            for i in range(n_batch):
                res = g_.download(i)
                st, n_ref, cx, cc = ref_crc[i % args.frames]
                good = st == 0 and res["n"] == n_ref and zlib.crc32(res["xyz"].tobytes()) == cx and zlib.crc32(res["rgb"].tobytes()) == cc
                ok &= int(good)
                if not good:
                    bad.append((gi, i))
                if i in (0, n_batch // 2, n_batch - 1) or not good:
                    verified.append({"frame": rank * args.frames + i % args.frames, "batch_entry": i, "gof": gi, "points": int(res["n"]),
                                     "xyz_crc32": zlib.crc32(res["xyz"].tobytes()), "rgb_crc32": zlib.crc32(res["rgb"].tobytes()),
                                     "equals_oracle": bool(good)})
        verified.append({"entries_checked": n_batch * len(gofs), "entries_equal_oracle": n_batch * len(gofs) - len(bad)})
    if args.smooth and not args.no_verify:
        sys.path.insert(0, os.path.join(REPO, "tests"))
        import oracle_binding as ob
        for i in (0, n_batch - 1):
            st, ref = ob.reconstruct(frames[i % args.frames])
            xyz_r, rgb_r, part = ob.xyz_array(ref), ob.rgb_array(ref), ref["partition"].astype(np.uint16)
            xs = ob.spec_smooth_geometry(xyz_r, part, bitdepth, smooth_kw["grid_size"], smooth_kw["threshold"])
            cs = ob.spec_smooth_color(xs, rgb_r, part, bitdepth, smooth_kw["color_grid_size"],
                                      smooth_kw["color_threshold_smoothing"], smooth_kw["color_threshold_difference"])
            got = gof.download(i)
            good = st == 0 and got["n"] == ref["n"] and np.array_equal(got["xyz"], xs) and np.array_equal(got["rgb"], cs)
            ok &= int(good)
            verified.append({"frame": rank * args.frames + i % args.frames, "batch_entry": i, "points": int(got["n"]),
                             "moved_points": int(np.any(xs != xyz_r, axis=1).sum()), "equals_spec": bool(good)})
    if dist is not None:
        v = torch.tensor([ok], dtype=torch.int64, device=red_dev)
        dist.all_reduce(v, op=dist.ReduceOp.MIN)
        ok = int(v.item())
    if not ok:
        print(f"bench.py: rank {rank}: output of the timed batch differs from the CPU oracle / the smoothing spec: {verified}", file=sys.stderr)
        sys.exit(3)

    # ---- roofline of the dominant kernel ---------------------------------------------------------
    roofline = None
    if rank == 0:
        rd = {k: v * max(args.cycles, 1) for k, v in traffic.gof_read_bytes(frames).items() if isinstance(v, (int, float))}
        out_bytes = 9 * points_per_step
        necessary = rd["block_bytes"] + rd["occupancy_plane"] + out_bytes
        line_floor = rd["seg128"] + rd["occupancy_plane"] + out_bytes
        dom = max(kernels, key=kernels.get)
        dom_ms = kernels[dom]
        dom_bytes = alg_bytes
        if args.smooth:                        # --smooth: the line is about the filters — all their launches, their algorithmic bytes
            dom = "k_smooth_*"
            dom_bytes = smooth_bytes
            dom_ms = sum(v for k, v in kernels.items() if k.startswith(("k_smooth", "smooth_")))
        if dom_ms <= 0:                        # (a counter pass of three steps: none of them carried an event pair)
            dom_ms = region_ms_per_step
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
        tr = measured_traffic(dom, args.workload, n_batch) if not args.smooth else None
        if args.smooth:
            tr = measured_traffic("k_smooth", args.workload, n_batch)
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                    "achieved_is": "algorithmic bytes (SURVEY 8d: whole planes once + 9 B/point) / kernel time — a rate of the bytes the path stands for; the physical HBM rate is frac_traffic",
                    "traffic": tr["hbm_bytes_per_launch"] if tr else None,
                    "traffic_stale": stale(tr, "k_smooth" if args.smooth else "k_recon_tiles") if tr else None,
                    "traffic_measured_on": {"kernel_source_sha16": tr.get("kernel_source_sha16"), "library_sha16": (tr.get("library") or {}).get("sha16")} if tr else None,
                    "algorithmic_bytes_per_launch": dom_bytes, "kernel_ms": round(dom_ms, 4),
                    "kernel_ms_launches_averaged": launches_averaged,
                    "necessary_bytes": necessary, "line_floor_bytes": line_floor,
                    "frac_necessary": round(necessary / (kernels.get("k_recon_tiles", dom_ms) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                    "frac_line_floor": round(line_floor / (kernels.get("k_recon_tiles", dom_ms) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                    "frac_traffic": round(tr["hbm_bytes_per_launch"] / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if tr else None,
                    "traffic_source": tr.get("source") if tr else None,
                    "timed_region_ms_per_step": round(region_ms_per_step, 4),
                    "all_kernels_ms": {k: round(v, 4) for k, v in kernels.items()},
                    # the same algorithmic bytes over EVERY kernel of the step — planning (block_to_patch + work list,
                    # src/codec.rs:205-250) and reconstruction: the fraction the per-frame path as a whole reaches
                    "path_kernels_ms": round(sum(kernels.values()), 4),
                    "path_frac": None if args.smooth else round(alg_bytes / (sum(kernels.values()) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
        if args.smooth:
            roofline["smoothing_algorithmic_bytes_per_launch"] = smooth_bytes
            roofline["reconstruction_algorithmic_bytes_per_launch"] = alg_bytes

    # ---- BASELINE configs 5 and 4 beside the headline (N=1, rank 0): short legs with their own verification -----
    def spec_check(g_, fr_list, entries):
        """Smoothed output of batch entries against oracle reconstruction + oracle/vpcc_smoothing_spec.c (checker only)."""
        sys.path.insert(0, os.path.join(REPO, "tests"))
        import oracle_binding as ob
        res_all = []
        for i in entries:
            st, ref = ob.reconstruct(fr_list[i % len(fr_list)])
            xyz, rgb, part = ob.xyz_array(ref), ob.rgb_array(ref), ref["partition"].astype(np.uint16)
            xs = ob.spec_smooth_geometry(xyz, part, bitdepth, smooth_kw["grid_size"], smooth_kw["threshold"])
            cs = ob.spec_smooth_color(xs, rgb, part, bitdepth, smooth_kw["color_grid_size"],
                                      smooth_kw["color_threshold_smoothing"], smooth_kw["color_threshold_difference"])
            got = g_.download(i)
            good = st == 0 and got["n"] == ref["n"] and np.array_equal(got["xyz"], xs) and np.array_equal(got["rgb"], cs)
            res_all.append({"batch_entry": i, "points": int(got["n"]), "moved_points": int(np.any(xs != xyz, axis=1).sum()),
                            "recoloured_points": int(np.any(cs != rgb, axis=1).sum()), "equals_spec": bool(good)})
        return res_all

    def secondary(workload, smooth, n_distinct, cycles_):
        mk = synth.longdress_frame if workload == "longdress" else synth.owlii_frame
        fr2 = frames[:n_distinct] if workload == args.workload else [mk(i) for i in range(n_distinct)]
        cap2 = 1_000_000 if workload == "longdress" else 2_400_000
        fl = _abi.VPCC_GOF_PROFILE | (_abi.VPCC_GOF_WANT_PATCH_INDEX if smooth else 0)
        g2 = ctx.gof(fr2 * cycles_, capacity=cap2, flags=fl)
        g2.profile_interval(4)
        nb = n_distinct * cycles_

        def st2():
            g2.reconstruct()
            if smooth:
                g2.smooth(10 if workload == "longdress" else 11, **smooth_kw)
        for _ in range(8):
            st2()
        g2.sync()
        c2 = g2.point_counts().astype(np.int64)
        t0, n2 = time.perf_counter(), 0
        while n2 < 16 or time.perf_counter() - t0 < 0.3:
            for _ in range(8):
                st2()
            g2.sync()
            n2 += 8
        dt = (time.perf_counter() - t0) / n2
        k2, nl = g2.kernel_time_means(min(max(n2 // 4, 1), 512))
        alg2 = sum(g2.algorithmic_bytes(i) for i in range(nb))
        out2 = {"workload": f"S-{workload}, {nb} frames per launch ({n_distinct} distinct)" + (" + geometry and colour smoothing" if smooth else ""),
                "ms_per_step": round(dt * 1e3, 4), "Mpoints_per_s": round(int(c2.sum()) / dt / 1e6, 1),
                "all_kernels_ms": {k: round(v, 4) for k, v in k2.items()}, "kernel_ms_launches_averaged": nl}
        if smooth:
            sm = sum(v for k, v in k2.items() if k.startswith("k_smooth"))
            sb = smoothing_algorithmic_bytes(g2, n_distinct, 10, smooth_kw["grid_size"], smooth_kw["color_grid_size"]) * cycles_
            trs = measured_traffic("k_smooth", workload, nb)
            out2.update({"smoothing_kernels_ms": round(sm, 4), "smoothing_algorithmic_bytes_per_launch": sb,
                         "frac": round(sb / (sm * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                         "traffic": trs["hbm_bytes_per_launch"] if trs else None,
                         "frac_traffic": round(trs["hbm_bytes_per_launch"] / (sm * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if trs else None,
                         "traffic_stale": stale(trs, "k_smooth") if trs else None,
                         "verified": spec_check(g2, fr2, [0, nb - 1])})
            out2["equals_spec"] = all(v["equals_spec"] for v in out2["verified"])
        else:
            km = k2.get("k_recon_tiles", dt * 1e3)
            tro = measured_traffic("k_recon_tiles", workload, nb)
            sys.path.insert(0, os.path.join(REPO, "tests"))
            import oracle_binding as ob
            eq = True
            for i in range(n_distinct):
                st_, ref = ob.reconstruct(fr2[i])
                for e in (i, nb - n_distinct + i):                     # its first and its last copy in the batch
                    got = g2.download(e)
                    eq = eq and st_ == 0 and got["n"] == ref["n"] and np.array_equal(got["xyz"], ob.xyz_array(ref)) and \
                        np.array_equal(got["rgb"], ob.rgb_array(ref))
            out2.update({"kernel_ms": round(km, 4), "algorithmic_bytes_per_launch": alg2,
                         "frac": round(alg2 / (km * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                         "traffic": tro["hbm_bytes_per_launch"] if tro else None,
                         "frac_traffic": round(tro["hbm_bytes_per_launch"] / (km * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if tro else None,
                         "traffic_stale": stale(tro, "k_recon_tiles") if tro else None,
                         "entries_checked": 2 * n_distinct, "equals_oracle": bool(eq)})
        g2.close()
        return out2

    pool_after = ctx.pool_info() if pool_info else None          # (with the timed batches alive)
    # ---- fresh_gof: what the product pays once per gof (N=1, rank 0) -----------------------------------------------
    def fresh_gof_leg(n_batches=3):
        """Three distinct 128-frame batches of raster planes RESIDENT IN HBM — a GPU video decoder's frame pool, in memory of
        the context's pool (vpcc_ctx_pool_alloc: frames 0-7 in home 0, 8-15 in home 1, ...) — borrowed by the library
        (VPCC_MEM_DEVICE).  A step = vpcc_gof_create on the next batch (validation, the host's O(patches) records, one
        descriptor copy) + vpcc_gof_reconstruct (k_plan_tiles + k_recon_tiles) + vpcc_gof_destroy of the gof before last:
        every gof is launched ONCE, as the streaming product launches it.  Timed like the headline."""
        import ctypes as C
        hip = C.CDLL("libamdhip64.so.7")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        lib = ctx.lib
        held, keep, arrays = [], [], []
        for b in range(n_batches):
            order = [(i * (2 * b + 1) + 5 * b) % len(batch) for i in range(len(batch))]      # every batch in an order of its own
            arr = (_abi.FrameDesc * len(batch))()
            for r0 in range(0, len(batch), 8):
                run = [batch[order[i]] for i in range(r0, min(r0 + 8, len(batch)))]
                planes = []
                for f in run:
                    planes.append([f["occupancy"]] + [f["geometry"][m] for m in range(2)] + [pl for m in range(2) for pl in f["attribute"][m]])
                total = sum((pl.nbytes + 255) // 256 * 256 for fr_ in planes for pl in fr_)
                base = ctx.pool_alloc((r0 // 8) % 2, total)
                held.append(base)
                at = base
                for j, f in enumerate(run):
                    d, k_ = _abi.host_frame_desc(f)
                    keep.append(k_)
                    ptrs = []
                    for pl in planes[j]:
                        a_ = np.ascontiguousarray(pl)
                        assert hip.hipMemcpy(at, a_.ctypes.data, a_.nbytes, 1) == 0
                        ptrs.append(at)
                        at += (a_.nbytes + 255) // 256 * 256
                    d.occupancy.y = ptrs[0]
                    d.occupancy.stride = d.occupancy.width
                    for m in range(2):
                        d.geometry[m].y = ptrs[1 + m]
                        d.attribute[m].y, d.attribute[m].u, d.attribute[m].v = ptrs[3 + 3 * m], ptrs[4 + 3 * m], ptrs[5 + 3 * m]
                    arr[r0 + j] = d
            arrays.append((arr, order))
        live, host_s, wait_s, k_sum, k_n, turn_ = [], [0.0], [0.0], {}, [0], [0]

        def read_times(h_):
            names = (C.c_char_p * 8)()
            msv = (C.c_float * 8)()
            n_ = lib.vpcc_gof_kernel_times(h_, names, msv, 8)
            for q in range(n_):
                k_sum[names[q].decode()] = k_sum.get(names[q].decode(), 0.0) + float(msv[q])
            k_n[0] += 1

        def step_():
            t_ = time.perf_counter()
            k = turn_[0]
            turn_[0] += 1
            h_ = C.c_void_p()
            prof = _abi.VPCC_GOF_PROFILE if k % 16 == 0 else 0
            st_ = lib.vpcc_gof_create(ctx.h, arrays[k % n_batches][0], len(batch), _abi.VPCC_MEM_DEVICE, cap, prof, C.byref(h_))
            ctx._check(st_, "vpcc_gof_create")
            ctx._check(lib.vpcc_gof_reconstruct(h_, 0, len(batch), None), "vpcc_gof_reconstruct")
            live.append((h_, k % n_batches, bool(prof)))
            host_s[0] += time.perf_counter() - t_
            if len(live) > 2:
                old, _, was_prof = live.pop(0)
                t_ = time.perf_counter()
                ctx._check(lib.vpcc_gof_sync(old), "vpcc_gof_sync")          # the caller's thread runs ahead of the GPU: it waits HERE
                wait_s[0] += time.perf_counter() - t_
                if was_prof:
                    read_times(old)
                t_ = time.perf_counter()
                lib.vpcc_gof_destroy(old)
                host_s[0] += time.perf_counter() - t_

        def sync_():
            for h_, _, _ in live:
                ctx._check(lib.vpcc_gof_sync(h_), "vpcc_gof_sync")

        for _ in range(64):
            step_()
        sync_()
        host_s[0], wait_s[0], turn0 = 0.0, 0.0, turn_[0]
        k_sum.clear()
        k_n[0] = 0
        reg_ = sharding.timed_region(step_, sync_, args.steps, args.warmup, points_per_step, min_seconds=args.min_seconds,
                                     device_sync=torch.cuda.synchronize)
        steps_run = turn_[0] - turn0
        host_us = host_s[0] / max(steps_run, 1) * 1e6
        # the last gofs' output against the oracle (entries of each batch lie in that batch's order)
        sys.path.insert(0, os.path.join(REPO, "tests"))
        import oracle_binding as ob
        good, checked = True, 0
        for h_, bi, _ in live:
            order = arrays[bi][1]
            cts = np.zeros(len(batch), dtype=np.uint32)
            ctx._check(lib.vpcc_gof_point_counts(h_, cts.ctypes.data), "vpcc_gof_point_counts")
            for e in (0, len(batch) // 2 + 3, len(batch) - 1):
                st_, ref = ob.reconstruct(batch[order[e]])
                n_ = int(cts[e])
                xyz = np.zeros(max(n_, 1), dtype=_abi.POINT3_DTYPE)
                rgb = np.zeros(max(n_, 1), dtype=_abi.COLOR3_DTYPE)
                kk = C.c_size_t(0)
                ctx._check(lib.vpcc_gof_download(h_, e, xyz.ctypes.data, rgb.ctypes.data, None, max(n_, 1), C.byref(kk)), "vpcc_gof_download")
                good = good and st_ == 0 and kk.value == ref["n"] and zlib.crc32(xyz[:kk.value].tobytes()) == zlib.crc32(ob.xyz_array(ref).tobytes()) \
                    and zlib.crc32(rgb[:kk.value].tobytes()) == zlib.crc32(ob.rgb_array(ref).tobytes())
                checked += 1
        for h_, _, was_prof in live:
            if was_prof:
                read_times(h_)
            lib.vpcc_gof_destroy(h_)
        for p_ in held:
            ctx.pool_free(p_)
        ms_ = reg_["elapsed_s"] / reg_["steps_effective"] * 1e3
        km = {k_: round(v / max(k_n[0], 1), 4) for k_, v in k_sum.items()}
        ksum = sum(km.values())
        return {"step": f"vpcc_gof_create + vpcc_gof_reconstruct on the next of {n_batches} resident {len(batch)}-frame batches (raster planes in HBM, "
                        f"VPCC_MEM_DEVICE, frame pool in the context's pool) + vpcc_gof_destroy of the gof before last: every gof launched once",
                "ms_per_step": round(ms_, 4), "steps_effective": reg_["steps_effective"],
                "frames_per_s": round(len(batch) / (ms_ * 1e-3), 1), "Mpoints_per_s": round(points_per_step / (ms_ * 1e-3) / 1e6, 1),
                "host_us_per_step": round(host_us, 1), "host_us_per_frame": round(host_us / len(batch), 3),
                "host_is": "wall time of vpcc_gof_create + vpcc_gof_reconstruct + vpcc_gof_destroy on the caller's thread (validation, "
                           "O(patches) records, layout, one staged copy, memsets, launches, release); the thread runs ahead of the GPU and "
                           "waits for the gof before last in a vpcc_gof_sync of its own, which is not in it",
                "host_waits_for_gpu_us_per_step": round(wait_s[0] / max(steps_run, 1) * 1e6, 1),
                "all_kernels_ms": km, "kernel_launches_averaged": k_n[0],
                "frac": round(alg_bytes / (ksum * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if ksum else None,
                "frac_of_the_step": round(alg_bytes / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                "entries_checked": checked, "equals_oracle": bool(good)}

    fresh = None
    if rank == 0 and world == 1 and not args.no_fresh_gof and not args.smooth and not args.general:
        for g_ in gofs:                                          # their 5 GB are not needed any more
            g_.close()
        gofs = []
        fresh = fresh_gof_leg()
        if not fresh["equals_oracle"]:
            print(f"bench.py: fresh_gof failed verification: {fresh}", file=sys.stderr)
            sys.exit(3)

    other = None
    if rank == 0 and world == 1 and not args.no_other_configs and not args.smooth and not args.general and args.workload == "longdress":
        for g_ in gofs:                                          # their 5 GB are not needed any more
            g_.close()
        gofs = []
        other = {"owlii": secondary("owlii", False, 8, 16), "smooth": secondary("longdress", True, min(args.frames, 32), 4)}
        if not (other["owlii"]["equals_oracle"] and other["smooth"]["equals_spec"]):
            print(f"bench.py: other_configs failed verification: {other}", file=sys.stderr)
            sys.exit(3)

    # ---- end to end once more, WARM: a second Decoder in a process whose allocator, page-locked buffers and clocks are up
    for g_ in gofs:
        g_.close()
    gofs = []
    ctx.close()
    if e2e is not None:
        points_per_gof = int(counts[:args.frames].sum())
        assert e2e.pop("_frames") == args.frames * args.e2e_gofs and e2e.pop("_points") == points_per_gof * args.e2e_gofs, "Decoder output differs"
        if not args.no_warm_e2e:
            warm = end_to_end_leg(frames, args, local_rank)
            assert warm.pop("_frames") == args.frames * args.e2e_gofs and warm.pop("_points") == points_per_gof * args.e2e_gofs, "Decoder output differs"
            e2e["warm"] = {k: warm[k] for k in ("whole_run_frames_per_s", "after_first_frame_frames_per_s", "startup_s", "start_call_s", "h2d_GBps_whole_run",
                                                "host_plan_and_enqueue_us_per_frame")}
            e2e["warm"]["note"] = "a second Decoder at the end of the run (contexts, caches and clocks warm): secondary, never the figure quoted"

    # ---- CPU baseline: the oracle (port of the reference algorithm), single thread, rank 0, N=1 ---
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import ctypes as C
        sys.path.insert(0, os.path.join(REPO, "tests"))
        import oracle_binding as ob            # checker/baseline only — never on the product path
        n_cpu = min(args.frames, 32)
        descs = (_abi.FrameDesc * n_cpu)()
        keep = []
        for i in range(n_cpu):
            d, k = _abi.host_frame_desc(frames[i])
            descs[i] = d
            keep.append(k)
        pts, st = C.c_uint64(0), C.c_int(0)
        reps = 4
        secs = ob.lib().vpcc_oracle_time_frames(descs, n_cpu, reps, C.byref(pts), C.byref(st))
        assert st.value == 0
        assert pts.value == int(counts[:n_cpu].sum()), "oracle and HIP path disagree on the point count"
        cpu = {"value": round(pts.value / secs / 1e6, 3), "unit": "Mpoints/s", "cores": 1, "kind": "port",
               "frames_per_s": round(n_cpu / secs, 2),
               "sample": f"{n_cpu} S-{args.workload} frames x {reps} reps (fastest rep), CPU oracle single-threaded"}
        # extra information: the same oracle frames-parallel on the host cores this process may use
        import threading
        n_thr = max(1, min(len(os.sched_getaffinity(0)), n_cpu))
        per = [(i * n_cpu // n_thr, (i + 1) * n_cpu // n_thr) for i in range(n_thr)]
        done = [0] * n_thr

        def work(t):
            lo, hi = per[t]
            sub = (_abi.FrameDesc * (hi - lo))(*[descs[i] for i in range(lo, hi)])
            p_, s_ = C.c_uint64(0), C.c_int(0)
            ob.lib().vpcc_oracle_time_frames(sub, hi - lo, 1, C.byref(p_), C.byref(s_))   # ctypes releases the GIL
            done[t] = p_.value

        t_par = time.perf_counter()
        th = [threading.Thread(target=work, args=(t,)) for t in range(n_thr)]
        [t.start() for t in th]
        [t.join() for t in th]
        t_par = time.perf_counter() - t_par
        cpu["all_cores"] = {"value": round(sum(done) / t_par / 1e6, 3), "unit": "Mpoints/s", "cores": n_thr,
                            "frames_per_s": round(n_cpu / t_par, 2),
                            "sample": f"{n_cpu} frames dealt to {n_thr} threads, one pass"}

    if rank == 0:
        ms_per_step = elapsed / steps_eff * 1e3
        out = {
            "metric": "V-PCC reconstruction throughput (points/s; frames/s alongside)",
            "value": round(reg["points_total_per_step"] * steps_eff / elapsed / 1e6, 2),
            "unit": "Mpoints/s",
            "frames_per_s": round(n_batch * world * steps_eff / elapsed, 1),
            "us_per_frame": round(elapsed / steps_eff / n_batch * 1e6, 3),
            "n_gpus": world, "steps": args.steps, "steps_effective": steps_eff, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "ms_per_32_frames": round(ms_per_step * 32 / n_batch, 4),       # for comparison with round 1's 32-frame step
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u16", "data": "synthetic",
            "config": {"workload": f"S-{args.workload}: {frames[0]['width']}x{frames[0]['height']} geometry+attribute, "
                                   f"occupancy /{frames[0]['occupancy_precision']}, {args.frames} distinct frames per GOF "
                                   f"per GPU, ~{int(counts.mean())} points/frame (BASELINE configs[1] shape); a step = ONE launch "
                                   f"over {max(args.cycles, 1)} such GOFs ({n_batch} frames, every copy in device buffers of its own)",
                       "frames_per_step_per_gpu": n_batch, "points_per_step_per_gpu": points_per_step,
                       "distinct_frames": args.frames, "gofs_per_launch": max(args.cycles, 1),
                       "batches_in_rotation": max(args.gofs, 1),
                       "pool": pool_after if pool_after else "none: every block of a gof is a hipMalloc of its own",
                       "kernel_path": "general" if args.general else "default",
                       "smoothing": smooth_kw if args.smooth else None,
                       "parallelism": f"frame-sharded x{world}, no collective on the data path"},
            "roofline": roofline,
            "library": lib_info,
            "gpu_state_under_load": gpu_state,
            "fresh_gof": fresh,
            "other_configs": other,
            "launches_of_one_gof": one_gof,
            "repeat_one_batch": repeat,
            "verified_frames": verified,
            "end_to_end": e2e,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    for g_ in gofs:
        g_.close()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
