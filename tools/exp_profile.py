import os, sys, time
sys.path.insert(0, "/root/repo/tmc2-rs_amd")
from tmc2rs import recon, synth, _abi
ctx = recon.Context(0)
frames = [synth.longdress_frame(i) for i in range(32)]
def run(flags, every, label, nsync=100):
    g = ctx.gof(frames * 4, capacity=1_000_000, flags=flags)
    if every: g.profile_interval(every)
    g.reconstruct(); g.sync()
    out = []
    for _ in range(12):
        t0 = time.perf_counter()
        for _ in range(nsync):
            g.reconstruct()
        g.sync()
        out.append((time.perf_counter() - t0) / nsync * 1e3)
    print(label, " ".join("%.3f" % x for x in out))
    g.close()
run(0, 0, "no profile        ")
run(_abi.VPCC_GOF_PROFILE, 16, "profile every 16  ")
run(_abi.VPCC_GOF_PROFILE, 1, "profile every 1   ")
run(0, 0, "no profile again  ")
run(0, 0, "no profile, sync/1000", 1000)
