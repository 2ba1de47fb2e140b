"""The tile kernel converts colour with INTEGER arithmetic (vpcc_tiles.hip: yuv10_to_rgb8_int) and
falls back to the IEEE f64 formula only when the integer quotient is exact.  tests/colour_exhaustive.c
checks the identity that makes this bit-exact against the reference formula (src/codec.rs:661-687) on
ALL 2^30 10-bit (Y,U,V) triplets — about 10 s on one core."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_integer_colour_path_is_exact_on_the_whole_10bit_cube(tmp_path):
    exe = tmp_path / "colour_exhaustive"
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", str(exe), os.path.join(HERE, "colour_exhaustive.c"), "-lm"])
    out = subprocess.check_output([str(exe)], timeout=600).decode()
    assert "total 1073741824 mismatches 0 " in out, out
    assert "exact-multiple(slow path) 10364" in out
