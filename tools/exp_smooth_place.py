#!/usr/bin/env python3
"""Diagnostic: do the smoothing kernels care where their scratch (cell grids, cell-index array) lies?  A gof of 128
S-longdress frames per spacer size: `spacer` GB are allocated (and kept) before the first vpcc_gof_smooth allocates the
scratch, then the kernels of reconstruct + smooth are timed."""
import ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tmc2-rs_amd"))
from tmc2rs import recon, synth, _abi
hip = C.CDLL("libamdhip64.so")
frames = [synth.longdress_frame(i) for i in range(32)] * 4
ctx = recon.Context(0)
kw = dict(grid_size=8, threshold=4, color_grid_size=8, color_threshold_smoothing=10, color_threshold_difference=100)
for spacer in [int(x) for x in sys.argv[1:]] or [0, 8, 16, 24, 32, 48, 64]:
    g = ctx.gof(frames, capacity=1_000_000, flags=_abi.VPCC_GOF_PROFILE | _abi.VPCC_GOF_WANT_PATCH_INDEX | _abi.VPCC_GOF_TUNE_PLACEMENT)
    g.reconstruct(); g.sync()
    held = []
    for _ in range(spacer):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), C.c_size_t(1000 << 20)) == 0
        held.append(p)
    for _ in range(6):
        g.reconstruct(); g.smooth(10, **kw)
    g.sync()
    k, n = g.kernel_time_means(4)
    tot = sum(v for name, v in k.items() if name.startswith("k_smooth"))
    print("spacer %3d GB: tiles %.3f  smoothing %.3f ms  " % (spacer, k.get("k_recon_tiles", 0), tot) +
          " ".join("%s %.3f" % (name.replace("k_smooth_", ""), v) for name, v in k.items() if name.startswith("k_smooth")), flush=True)
    g.close()
    for p in held: hip.hipFree(p)
ctx.close()
