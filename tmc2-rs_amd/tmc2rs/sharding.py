"""Host logic of the multi-GPU path: frames of a GOF are independent (src/decoder.rs:186), so they are
dealt round-robin over the ranks and no data-path collective exists.  The only communication is the
bookkeeping below (timing / point counts / in-order merge), which works on any torch.distributed
backend (RCCL on the GPUs, gloo in the CPU tests)."""


def frames_of_rank(n_frames, rank, world):
    """Frame f is reconstructed by rank f % world (same dealing as tmc2rs::Decoder::worker)."""
    return list(range(rank, n_frames, world))


def owner_of_frame(frame, world):
    return frame % world, frame // world          # (rank, index in that rank's local list)


def job_totals(dist, elapsed_s, points, device="cpu"):
    """Whole-job elapsed time (MAX over ranks) and points (SUM over ranks)."""
    import torch
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    p = torch.tensor([points], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(p, op=dist.ReduceOp.SUM)
    return float(t.item()), int(p.item())


def presentation_order_counts(dist, local_counts, n_frames, device="cpu"):
    """Per-frame point counts of the whole GOF in presentation order, from every rank's local counts."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    per_rank = (n_frames + world - 1) // world
    mine = torch.full((per_rank,), -1, dtype=torch.int64, device=device)
    mine[:len(local_counts)] = torch.as_tensor(list(local_counts), dtype=torch.int64, device=device)
    allc = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allc, mine)
    out = []
    for f in range(n_frames):
        r, i = owner_of_frame(f, world)
        out.append(int(allc[r][i].item()))
    return out


def timed_region(step, sync, steps, warmup, points_per_step, dist=None, device="cpu", min_seconds=0.0,
                 device_sync=None):
    """The timed region of bench.py, as the driver contract words it: `warmup` untimed steps, then K steps
    bracketed by a barrier + device synchronisation on both sides, MAX of the elapsed time over ranks, SUM of
    the points.  K = `steps`, raised (identically on every rank) until the region lasts about `min_seconds`:
    a step of ~0.1 ms would otherwise give a region far shorter than any clock or power ramp.

    step()        enqueues one pass of the hot path (asynchronous)
    sync()        waits for everything this rank has enqueued
    device_sync() optional extra device-wide synchronisation (torch.cuda.synchronize on the GPU)
    Returns {"elapsed_s", "steps_effective", "points_total_per_step", "world"}."""
    import math
    import time

    def full_sync():
        sync()
        if device_sync is not None:
            device_sync()

    def barrier():
        full_sync()
        if dist is not None:
            dist.barrier()
        full_sync()

    t_est = 0.0
    if warmup > 0:
        full_sync()
        t0 = time.perf_counter()
        for _ in range(warmup):
            step()
        full_sync()
        t_est = (time.perf_counter() - t0) / warmup
    k = int(steps)
    if min_seconds > 0.0 and t_est > 0.0:
        k = max(k, int(math.ceil(min_seconds / t_est)))
    if dist is not None:                       # every rank must run the same number of steps
        import torch
        kk = torch.tensor([k], dtype=torch.int64, device=device)
        dist.all_reduce(kk, op=dist.ReduceOp.MAX)
        k = int(kk.item())
    barrier()
    t0 = time.perf_counter()
    for _ in range(k):
        step()
    full_sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        elapsed, total = job_totals(dist, elapsed, points_per_step, device)
        dist.barrier()
        world = dist.get_world_size()
    else:
        total, world = int(points_per_step), 1
    return {"elapsed_s": elapsed, "steps_effective": k, "points_total_per_step": total, "world": world}
