#!/usr/bin/env python3
"""bench.py — reconstruction throughput of the MI355X-native V-PCC hot path.

A "step" is one pass of the hot path over one GOF (32 distinct synthetic S-longdress frames:
1280x1408 geometry+attribute, 320x352 occupancy, ~800 k points/frame — BASELINE.json configs[1],
SURVEY.md §8d), all decoded planes and patch tables already resident in HBM when the timed region
starts.  Frames are independent, so with N GPUs every rank reconstructs its own GOF (weak scaling,
no data-path collective); `value` is the whole-job Mpoints/s.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

One JSON line is printed by rank 0, with the contract fields plus
  roofline     : HBM roofline of the dominant kernel (algorithmic bytes of SURVEY §8d per launch /
                 its mean launch duration, measured with HIP events on the launch stream)
  cpu_baseline : the CPU oracle (a port of the reference's algorithm; the Rust crate cannot be
                 built here) timed single-threaded on this box, rank 0 at N=1 only.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "tmc2-rs_amd"))

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy rate


def measured_traffic(kernel, workload, frames):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/rNN/traffic.json: FETCH_SIZE x2 + WRITE_SIZE, calibrated there), or None."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r*", "traffic.json"))):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if kernel.split("<")[0] in d.get("kernel", "") and workload in d.get("workload", "") \
                and f"{frames} frames" in d.get("workload", ""):
            best = d
    return best["hbm_bytes_per_launch"] if best else None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--ramp-ms", type=float, default=150.0,
                    help="untimed launches before the warmup steps until this much wall time has passed: a step is ~0.15 ms, "
                         "far shorter than the GPU's clock ramp from idle")
    ap.add_argument("--frames", type=int, default=32, help="frames per GOF (per rank)")
    ap.add_argument("--workload", default="longdress", choices=["longdress", "owlii"])
    ap.add_argument("--general", action="store_true", help="force the general kernel sequence")
    ap.add_argument("--smooth", action="store_true",
                    help="BASELINE config 4: grid geometry + colour smoothing after reconstruction (own spec, see DESIGN.md)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=30)
    return ap.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run",
                  file=sys.stderr)
        sys.exit(2)

    import numpy as np
    import torch
    from tmc2rs import _abi, recon, synth

    # VPCC_BENCH_BACKEND=gloo is a REHEARSAL switch for boxes with fewer GPUs than ranks (ranks then share
    # devices and the two scalar reductions run on the CPU); measurements use the default, RCCL.
    backend = os.environ.get("VPCC_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    red_dev = "cuda" if backend == "nccl" else "cpu"
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend)

    # ---- synthetic GOF of this rank (distinct frames: the working set, ~0.8 GB, is >> the 256 MB
    # Infinity Cache, so planes stream from HBM) -------------------------------------------------
    make = synth.longdress_frame if args.workload == "longdress" else synth.owlii_frame
    kw = {}
    if os.environ.get("VPCC_BENCH_SWAP_PROB"):          # diagnostic only: share of Swap-oriented patches
        kw["swap_prob"] = float(os.environ["VPCC_BENCH_SWAP_PROB"])
    frames = [make(rank * args.frames + i, **kw) for i in range(args.frames)]
    cap = 1_000_000 if args.workload == "longdress" else 2_400_000

    ctx = recon.Context(local_rank)
    flags = _abi.VPCC_GOF_FORCE_GENERAL if args.general else 0
    if args.smooth:
        flags |= _abi.VPCC_GOF_WANT_PATCH_INDEX
    bitdepth = 10 if args.workload == "longdress" else 11
    smooth_kw = dict(grid_size=8, threshold=4, color_grid_size=8, color_threshold_smoothing=10,
                     color_threshold_difference=100)
    gof = ctx.gof(frames, capacity=cap, flags=flags)          # H2D happens here, outside the timed region

    def step(g):
        g.reconstruct()
        if args.smooth:
            g.smooth(bitdepth, **smooth_kw)

    gof.reconstruct()
    counts = gof.point_counts().astype(np.int64)
    assert all(gof.frame_status(i) == 0 for i in range(args.frames)), "capacity too small"
    points_per_step = int(counts.sum())
    alg_bytes = sum(gof.algorithmic_bytes(i) for i in range(args.frames))

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        gof.sync()

    t_ramp = time.perf_counter()
    while (time.perf_counter() - t_ramp) * 1e3 < args.ramp_ms:     # untimed: brings the clocks up from idle
        for _ in range(16):
            step(gof)
        gof.sync()
    for _ in range(args.warmup):
        step(gof)
    barrier()
    # HIP events on the stream the kernels are launched on, bracketing exactly the timed region
    # (torch.cuda.Event on torch's current stream would not see this stream)
    ext = torch.cuda.ExternalStream(ctx.stream(), device=torch.device("cuda", local_rank))
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(ext)
    for _ in range(args.steps):
        step(gof)
    ev1.record(ext)
    gof.sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ev1.synchronize()
    region_ms_per_step = ev0.elapsed_time(ev1) / args.steps
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        p = torch.tensor([points_per_step], dtype=torch.int64, device=red_dev)
        dist.all_reduce(p, op=dist.ReduceOp.SUM)
        total_points_per_step = int(p.item())
    else:
        total_points_per_step = points_per_step
    barrier()

    # ---- roofline of the dominant kernel: HIP events on the launch stream (profile-mode GOF) ------
    roofline = None
    kernels = {}
    if rank == 0:
        pg = ctx.gof(frames, capacity=cap, flags=flags | _abi.VPCC_GOF_PROFILE)
        for _ in range(3):                      # warm the profile GOF's own buffers
            pg.reconstruct()
        pg.sync()
        acc = {}
        for _ in range(max(args.profile_steps, 1)):
            pg.reconstruct()
            for name, ms in pg.kernel_times():
                acc.setdefault(name, []).append(ms)
        pg.close()
        kernels = {k: float(np.mean(v)) for k, v in acc.items()}
        dom = max(kernels, key=kernels.get)
        dom_ms = kernels[dom]
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                    "traffic": measured_traffic(dom, args.workload, args.frames),
                    "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": round(dom_ms, 4),
                    "timed_region_ms_per_step": round(region_ms_per_step, 4),
                    "all_kernels_ms": {k: round(v, 4) for k, v in kernels.items()},
                    "pipeline_achieved": round(alg_bytes / (sum(kernels.values()) * 1e-3) / 1e9, 1)}

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
        ms_per_step = elapsed / args.steps * 1e3
        out = {
            "metric": "V-PCC reconstruction throughput (points/s; frames/s alongside)",
            "value": round(total_points_per_step * args.steps / elapsed / 1e6, 2),
            "unit": "Mpoints/s",
            "frames_per_s": round(args.frames * world * args.steps / elapsed, 1),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u16", "data": "synthetic",
            "config": {"workload": f"S-{args.workload}: {frames[0]['width']}x{frames[0]['height']} geometry+attribute, "
                                   f"occupancy /{frames[0]['occupancy_precision']}, {args.frames} distinct frames per GOF "
                                   f"per GPU, ~{int(counts.mean())} points/frame (BASELINE configs[1] shape)",
                       "frames_per_step_per_gpu": args.frames, "points_per_step_per_gpu": points_per_step,
                       "kernel_path": "general" if args.general else "default",
                       "smoothing": smooth_kw if args.smooth else None,
                       "parallelism": f"frame-sharded x{world}, no collective on the data path"},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    gof.close()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
