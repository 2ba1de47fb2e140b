#!/usr/bin/env python3
"""Soak of the streaming Decoder: random streams — 1 .. 14 GOFs of 1 .. 40 frames drawn from a pool of random frames of several canvas
sizes —, 1 .. 4 lanes (on one GPU), random ingest switches (stretches / kernel / copy engine, staged descriptors or not, the input
page-locked in chunks, with or without the tail split, with or without a pool), now and then a consumer that stops early, three streams in ten with the smoothing filters switched on (random parameters); every
frame that arrives is compared with the oracle's (and the smoothing specification's), in presentation order; of a stream that differs the
tool prints which frames, points and values, and decodes the same file three more times.  Usage: tools/soak_decoder.py [streams = 150] [first seed = 0]
Environment: VPCC_SOAK_SHORT_STREAMS=1 — 1-3 GOFs of 1-8 frames (the start of a stream, over and over: seven streams a second);
VPCC_SOAK_FORCE_SWITCHES=A,B — these switches on in every stream (a hunt in one configuration)."""
import os, sys, tempfile, time, zlib
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tmc2-rs_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
from tmc2rs import container, recon, synth
import oracle_binding as ob
n_streams = int(sys.argv[1]) if len(sys.argv) > 1 else 150
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(0xDEC0 + seed0)
def crc(x, c): return zlib.crc32(np.ascontiguousarray(c).tobytes(), zlib.crc32(np.ascontiguousarray(x).tobytes()))
pool = []
for i in range(36):
    big = i % 6 == 0
    w = 16 * int(rng.integers(30, 81) if big else rng.integers(2, 26)); h = 16 * int(rng.integers(30, 70) if big else rng.integers(2, 20))
    f = synth.make_frame(w, h, int(rng.choice([1, 2, 4, 4])), 16, seed=0xDEC00000 + seed0 * 1009 + i, max_side=int(rng.integers(2, 20 if big else 9)),
                         cover_target=float(rng.uniform(0.2, 0.9)), swap_prob=float(rng.uniform(0, 1)), dup_prob=float(rng.uniform(0, 0.6)))
    if i % 3 != 0:                                             # patches that overlap in 3-D: cells that mix patches, work for the filters
        p = f["patches"].copy()
        p["u1"] = 100 + (np.arange(len(p)) % 5) * 3; p["v1"] = 100 + (np.arange(len(p)) % 7) * 2
        p["d1"] = np.where(p["projection_mode"] == 0, 100, 300)
        f["patches"] = p
    st, r = ob.reconstruct(f)
    assert st == 0
    pool.append((f, (r["n"], crc(ob.xyz_array(r), ob.rgb_array(r))), (ob.xyz_array(r), ob.rgb_array(r), r["partition"].astype(np.uint16))))
smoothed = {}
changed = [0]
def expected(k, sm):
    """(points, CRC) of pool frame k behind the filters `sm` = (geometry, colour, grid, threshold, colour grid, Ts, Td) — the specification's."""
    if sm is None:
        return pool[k][1]
    if (k, sm) not in smoothed:
        xyz, rgb, part = pool[k][2]
        geo, col, G, T, CG, Ts, Td = sm
        if geo:
            xyz = ob.spec_smooth_geometry(xyz, part, 10, G, T)
        if col:
            rgb = ob.spec_smooth_color(xyz, rgb, part, 10, CG, Ts, Td)
        smoothed[(k, sm)] = (len(xyz), crc(xyz, rgb))
        changed[0] += int(smoothed[(k, sm)] != pool[k][1])
    return smoothed[(k, sm)]
switches = ["VPCC_NO_EXTENT_INGEST", "VPCC_DECODER_PIN_AT_ONCE", "VPCC_NO_PULL_INGEST", "VPCC_NO_PUSH_DOWNLOAD", "VPCC_DECODER_NO_TAIL_SPLIT",
            "VPCC_DECODER_NO_HUGEPAGES"]
bad = frames = 0
t0 = time.time()
d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
path = os.path.join(d, "soak.vpccgof")
try:
    for si in range(n_streams):
        for s in switches + ["VPCC_DECODER_PIN_CHUNK_MB", "VPCC_DECODER_POOL_GIB"]:
            os.environ.pop(s, None)
        on = [s for s in switches if rng.random() < 0.2]
        on += [s for s in os.environ.get("VPCC_SOAK_FORCE_SWITCHES", "").split(",") if s and s not in on]     # (a hunt in one configuration)
        for s in on:
            os.environ[s] = "1"
        if rng.random() < 0.25:
            os.environ["VPCC_DECODER_PIN_CHUNK_MB"] = str(int(rng.choice([1, 3, 16])))
        if rng.random() < 0.7:
            os.environ["VPCC_DECODER_POOL_GIB"] = str(int(rng.choice([0, 0, 2, 4])))
        sm = None
        if rng.random() < 0.3:                                 # the post-processing switches (src/lib.rs:45-46), parameters from Params
            geo, col = [(True, False), (False, True), (True, True)][int(rng.integers(0, 3))]
            G = int(rng.choice([4, 8, 8, 16]))
            sm = (geo, col, G, int(rng.choice([0, 2, 4])), G if rng.random() < 0.6 else int(rng.choice([4, 8, 16])), int(rng.choice([0, 5, 10])), int(rng.choice([50, 100, 765])))
        gofs, expect, picked = [], [], []
        short = os.environ.get("VPCC_SOAK_SHORT_STREAMS") is not None     # many short streams: the start of a stream, over and over
        for g in range(int(rng.integers(1, 4 if short else 15))):
            idx = [int(k) for k in rng.integers(0, len(pool), size=int(rng.integers(1, 9 if short else 41)))]
            gofs.append([pool[k][0] for k in idx]); expect += [expected(k, sm) for k in idx]; picked += idx
        lanes = int(rng.choice([1, 1, 2, 3, 4]))
        stop_at = int(rng.integers(0, len(expect))) if rng.random() < 0.15 else None
        container.write_container(path, gofs)
        dec = recon.Decoder(path, devices=(0,) * lanes)
        if sm is not None:
            dec.set_smoothing(geometry=sm[0], color=sm[1], bitdepth=10, grid_size=sm[2], threshold=sm[3], color_grid_size=sm[4],
                              color_threshold_smoothing=sm[5], color_threshold_difference=sm[6])
        dec.start()
        got, detail = [], []
        for fr in dec:
            got.append((fr["n"], crc(fr["xyz"], fr["rgb"])))
            if sm is None and got[-1] != expect[len(got) - 1]:     # what differs, while the frame is at hand
                ex, ec, _ = pool[picked[len(got) - 1]][2]
                gx, gc = np.asarray(fr["xyz"]), np.asarray(fr["rgb"])
                m = min(len(ex), len(gx))
                dx = np.nonzero((gx[:m] != ex[:m]).any(axis=1))[0]; dc = np.nonzero((gc[:m] != ec[:m]).any(axis=1))[0]
                detail.append(f"frame {len(got) - 1}: points {len(gx)} vs {len(ex)}; positions differ at {len(dx)} points (first {dx[:6].tolist()}, last {dx[-3:].tolist()}), "
                              f"colours at {len(dc)} (first {dc[:6].tolist()}, last {dc[-3:].tolist()}); "
                              f"got xyz {gx[dx[:2]].tolist() if len(dx) else []} want {ex[dx[:2]].tolist() if len(dx) else []}; got rgb {gc[dc[:2]].tolist() if len(dc) else []} want {ec[dc[:2]].tolist() if len(dc) else []}")
            if stop_at is not None and len(got) > stop_at:
                break
        err = dec.error()
        dec.close()
        ok = err == "" and got == (expect if stop_at is None else expect[:stop_at + 1])
        frames += len(got)
        if not ok:
            bad += 1
            first = next((i for i, (a, b) in enumerate(zip(got, expect)) if a != b), None)
            differ = [i for i, (a, b) in enumerate(zip(got, expect)) if a != b]
            print(f"MISMATCH stream {si}: {len(gofs)} GOFs of {[len(g) for g in gofs]} frames, {lanes} lanes, {on}, "
                  f"{ {k: os.environ[k] for k in ('VPCC_DECODER_PIN_CHUNK_MB', 'VPCC_DECODER_POOL_GIB') if k in os.environ} }, smoothing {sm}, stop_at {stop_at}: "
                  f"error {err!r}, {len(got)} of {len(expect)} frames, first difference at {first}", flush=True)
            for line in detail[:8]:
                print("  " + line, flush=True)
            # which frames, what they are, and whether the same stream decoded again differs again
            print("  frames that differ:", [(i, picked[i], pool[picked[i]][0]["width"], pool[picked[i]][0]["height"], pool[picked[i]][0]["occupancy_precision"],
                                            "points %d vs %d" % (got[i][0], expect[i][0])) for i in differ[:12]], flush=True)
            print("  the stream's frames (pool index, width, height, precision):", [(k, pool[k][0]["width"], pool[k][0]["height"], pool[k][0]["occupancy_precision"]) for k in picked[:48]], flush=True)
            for again in range(3):
                dec2 = recon.Decoder(path, devices=(0,) * lanes)
                if sm is not None:
                    dec2.set_smoothing(geometry=sm[0], color=sm[1], bitdepth=10, grid_size=sm[2], threshold=sm[3], color_grid_size=sm[4],
                                       color_threshold_smoothing=sm[5], color_threshold_difference=sm[6])
                dec2.start()
                got2 = [(fr["n"], crc(fr["xyz"], fr["rgb"])) for fr in dec2]
                dec2.close()
                print(f"  decoded again ({again}): differs at {[i for i, (a, b) in enumerate(zip(got2, expect)) if a != b][:12]}", flush=True)
        if si % 10 == 9:
            print(f"{si + 1} streams, {frames} frames, {bad} bad, {time.time() - t0:.0f} s", flush=True)
finally:
    if os.path.exists(path):
        os.remove(path)
    os.rmdir(d)
print(f"soak: {n_streams} streams, {frames} frames, {bad} bad; {len(smoothed)} (frame, filter parameters) pairs, {changed[0]} of them changed by the filters")
sys.exit(1 if bad else 0)
