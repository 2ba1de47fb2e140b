#!/usr/bin/env python3
"""Differential soak of the HIP path against the CPU oracle: thousands of seeded random frames — canvas sizes, block sizes,
precisions, occupancy styles, patch statistics, orientations, relative D1, full-range samples, padded rows — in gofs of random
sizes, planes in host memory or in the caller's device memory (borrowed or copied, some not aligned), on the path gof creation chooses, points, colours and partition compared for every frame.
Usage: tools/soak_parity.py [frames = 2000] [first seed = 0]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tmc2-rs_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import torch                                                     # (the allocator of the caller's device planes; before the library's own HIP start-up)
torch.cuda.init()
from tmc2rs import _abi, recon, synth
import oracle_binding as ob
n_total = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(0x50A6 + seed0)

def frame(i, R):
    prec = int(rng.choice([1, 2, 4, 4, 4, 8]))
    unit = max(R, prec, 2) * (2 if prec == 8 else 1)
    span = 1400 if rng.random() < 0.08 else 420              # now and then a canvas of the size of a real one
    w = unit * int(rng.integers(2, max(3, span // unit))); h = unit * int(rng.integers(2, max(3, (span * 3 // 4) // unit)))
    f = synth.make_frame(w, h, prec, R, seed=0x50A60000 + seed0 * 100003 + i, max_side=int(rng.integers(1, max(2, 128 // R + 1))),
                         cover_target=float(rng.uniform(0.1, 0.98)), size_skew=float(rng.uniform(0.7, 4.0)),
                         swap_prob=float(rng.uniform(0, 1)), overlap_prob=float(rng.uniform(0, 0.6)),
                         dup_prob=float(rng.uniform(0, 0.7)), ellipse_scale=float(rng.uniform(0.4, 1.5)),
                         occupancy_values="random" if i % 3 == 0 else "one", coord_bits=int(rng.choice([8, 10, 11, 16])))
    if i % 7 == 3:
        f["attribute"] = [tuple(rng.integers(0, 65536, pl.shape, dtype=np.uint16) for pl in layer) for layer in f["attribute"]]
        f["geometry"] = [rng.integers(0, 65536, g.shape, dtype=np.uint16) for g in f["geometry"]]
    if i % 11 == 5:
        f["absolute_d1"] = 0
    if i % 13 == 7:
        f["map_count"] = 1
    if i % 17 == 9:
        f["attribute_count"] = 0
    return f

ctx = recon.Context(0)
done = bad = 0
t0 = time.time()
paths = {"tiles": 0, "general": 0, "general, block units": 0}
modes = [0, 0, 0]
while done < n_total:
    k = int(rng.integers(1, 41))
    R = int(rng.choice([16] * 7 + [8, 32, 4, 64]))                  # one block size per gof: gofs of 16s take the tile kernel
    frames = [frame(done + j, R) for j in range(k)]
    refs = [ob.reconstruct(f) for f in frames]
    mode = int(rng.choice([0, 0, 1, 2]))                          # planes: the host's / the caller's device planes, borrowed / ... copied
    force = _abi.VPCC_GOF_FORCE_GENERAL if rng.random() < 0.25 else 0   # a quarter of the gofs of 16s through the general sequence too
    if mode == 0:
        g = ctx.gof(frames, flags=_abi.VPCC_GOF_WANT_PATCH_INDEX | _abi.VPCC_GOF_PROFILE | force)
    else:
        keep, descs = [], []
        def up(a):
            raw = np.ascontiguousarray(a).view(np.uint8).reshape(-1)
            off = int(rng.choice([0, 0, 0, 2, 8, 16, 30, 256]))  # now and then a plane that does not start where an allocation does
            x = torch.empty(len(raw) + off + 64, dtype=torch.uint8, device="cuda:0")
            x[off:off + len(raw)] = torch.from_numpy(raw).to("cuda:0")
            keep.append(x)
            return x.data_ptr() + off
        for f in frames:
            d, k0 = _abi.host_frame_desc(f)
            keep.append(k0)
            d.occupancy.y = up(f["occupancy"]); d.occupancy.stride = d.occupancy.width
            for m in range(f["map_count"]):
                d.geometry[m].y = up(f["geometry"][m]); d.geometry[m].stride = d.geometry[m].width
                if f["attribute_count"]:
                    d.attribute[m].y, d.attribute[m].u, d.attribute[m].v = (up(pl) for pl in f["attribute"][m])
                    d.attribute[m].stride, d.attribute[m].cstride = d.attribute[m].width, d.attribute[m].width // 2
            descs.append(d)
        torch.cuda.synchronize()
        g = ctx.gof(None, memory=_abi.VPCC_MEM_DEVICE, descs=descs,
                    flags=_abi.VPCC_GOF_WANT_PATCH_INDEX | _abi.VPCC_GOF_PROFILE | force | (_abi.VPCC_GOF_COPY_PLANES if mode == 2 else 0))
        if mode == 2:
            del keep[:]                                          # copied: the caller's planes may go
            torch.cuda.synchronize()
    modes[mode] += k
    g.reconstruct()
    names = [n for n, _ in g.kernel_times()]
    paths["tiles" if any("k_recon_tiles" in n for n in names) else "general, block units" if "k_general_blocks" in names else "general"] += k
    for j, (st, ref) in enumerate(refs):
        got = g.download(j, want_patch_index=True)
        colours = frames[j]["attribute_count"] > 0               # (a frame without attribute: with_colors == false, nothing is written)
        ok = st == 0 and got["n"] == ref["n"] and np.array_equal(got["xyz"], ob.xyz_array(ref)) and \
            (not colours or np.array_equal(got["rgb"], ob.rgb_array(ref))) and np.array_equal(got["patch_index"].astype(np.uint64), ref["partition"])
        if not ok:
            bad += 1
            what = [] if st else [w for w, e in (("xyz", np.array_equal(got["xyz"], ob.xyz_array(ref)) if got["n"] == ref["n"] else False),
                                                   ("rgb", not colours or (got["n"] == ref["n"] and np.array_equal(got["rgb"], ob.rgb_array(ref)))),
                                                   ("partition", got["n"] == ref["n"] and np.array_equal(got["patch_index"].astype(np.uint64), ref["partition"]))) if not e]
            print(f"MISMATCH at frame {done + j} (gof of {k}, kernels {names}): oracle status {st}, points {ref['n'] if st == 0 else '-'} vs {got['n']}, differs in {what}, "
                  f"attribute_count {frames[j]['attribute_count']} map_count {frames[j]['map_count']} rgb {None if got['rgb'] is None else got['rgb'].shape}", flush=True)
    g.close()
    done += k
    if (done // 200) != ((done - k) // 200):
        print(f"{done} frames, {bad} mismatches, {time.time() - t0:.0f} s, frames by path {paths}, by plane memory (host, device, device copied) {modes}", flush=True)
print(f"soak: {done} frames, {bad} mismatches, frames by path {paths}, by plane memory (host, device, device copied) {modes}")
sys.exit(1 if bad else 0)
