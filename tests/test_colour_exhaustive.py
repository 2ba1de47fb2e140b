"""The tile kernel converts colour without a division: f64 FMAs on a 2^-20 grid, integer floor/clamp, and the
reference formula itself only for the few fraction patterns that could hide an exact integer
(tmc2-rs_amd/csrc/vpcc_colour.h).  tests/colour_exhaustive.c compiles that SAME header for the CPU and checks
it against the reference formula (src/codec.rs:661-687) on ALL 2^30 10-bit (Y,U,V) triplets — about 15 s on
one core with hardware FMA (IEEE fma is the same function on the CPU and on gfx950)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "tmc2-rs_amd", "csrc")


def _has_fma():
    try:
        with open("/proc/cpuinfo") as f:
            return " fma " in f.read()
    except OSError:
        return False


def test_fast_colour_path_is_exact_on_the_whole_10bit_cube(tmp_path):
    exe = tmp_path / "colour_exhaustive"
    flags = ["-O2", "-ffp-contract=off", "-I", CSRC] + (["-mfma"] if _has_fma() else [])
    subprocess.check_call(["gcc"] + flags + ["-o", str(exe), os.path.join(HERE, "colour_exhaustive.c"), "-lm"])
    out = subprocess.check_output([str(exe)], timeout=1200).decode()
    assert "total 1073741824 mismatches 0 " in out, out
    assert "ambiguous(slow path) 25601 exact-unflagged 0" in out, out
