"""Mutation fuzzing of the V3C syntax parser under AddressSanitizer + UBSan (CPU build of the host sources
only; GPU sanitizers are not available on the pool): malformed streams must end in a SyntaxError, never in
an out-of-bounds access, undefined arithmetic or an uncaught exception."""
import os
import subprocess

import v3c_writer as W
from test_v3c_syntax import BASE, patches_for

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(REPO, "tmc2-rs_amd", "csrc")


def test_parser_survives_mutated_streams(tmp_path):
    exe = tmp_path / "fuzz_v3c"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I", os.path.join(REPO, "include"), "-I", CSRC, "-o", str(exe),
                           os.path.join(HERE, "fuzz_v3c.cpp"), os.path.join(CSRC, "v3c_syntax.cpp"),
                           os.path.join(CSRC, "bitstream.cpp")])
    p = dict(BASE, use_eight_orientations=1)
    def nals(*payloads):                                # a video sub-bitstream: 4-byte length prefixes + HEVC-like NALs
        return b"".join(len(x).to_bytes(4, "big") + x for x in payloads)

    videos = (nals(b"\x40\x01vps", b"\x42\x01sps", b"\x26\x01slice-data"), nals(b"\x40\x01", b"\x02\x01abc"), nals(b"\x26\x01x" * 3))
    data = W.sample_stream(W.gof_units(p, [patches_for(f, n=6, eight=True) for f in range(3)], sei=(8, 64), videos=videos) +
                           W.gof_units(dict(BASE), [patches_for(f, n=2) for f in range(2)], videos=videos))
    seed = tmp_path / "seed.bin"
    seed.write_bytes(data)
    out = subprocess.run([str(exe), str(seed), "20000"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "iterations 20000" in out.stdout and "rejected" in out.stdout, out.stdout
