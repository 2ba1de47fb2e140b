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
