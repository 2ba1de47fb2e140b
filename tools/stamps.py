#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the tile kernel (VPCC_TILES_VARIANT=64 build path).
Shares only — the stamped run serialises loads and must not be used for timing."""
import ctypes as C, os, sys
os.environ["VPCC_DIAG_LIB"] = "1"          # libvpcc_recon_diag.so (`make diag`): stamps exist in the diagnostic build only
os.environ["VPCC_TILES_VARIANT"] = os.environ.get("VPCC_TILES_VARIANT", "64")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tmc2-rs_amd"))
from tmc2rs import recon, synth, _abi
ctx = recon.Context(0)
owlii = os.environ.get("VPCC_WORKLOAD") == "owlii"
frames = [(synth.owlii_frame if owlii else synth.longdress_frame)(i) for i in range(32)]
g = ctx.gof(frames, capacity=2_400_000 if owlii else 1_000_000)
lib = _abi.load_library()
buf = (C.c_uint64 * 16)()
g.reconstruct(); g.sync()
lib.vpcc_debug_read_stamps(buf, 1)
for _ in range(3):
    g.reconstruct()
g.sync()
lib.vpcc_debug_read_stamps(buf, 1)
names = ["ticket+barrier", "count(occ+geo x4)", "barrier#1", "publish next", "look-back+1st loads", "4 items (all)", "drain", "  item: ranks+colour+records", "  item: wait for next item's loads", "  item: store loop"]
n = buf[15] or 1
tot = sum(buf[i] for i in range(7)) or 1
for i, nm in enumerate(names):
    print(f"{nm:16s} {buf[i]/n:10.0f} cycles/wave  {100.0*buf[i]/tot:5.1f}%")
print("waves sampled", n, " total cycles/wave", tot / n, "(s_memtime ticks; 100 MHz? see below)")
