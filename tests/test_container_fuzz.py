"""Mutation fuzzing of the decoded-GOF container reader under AddressSanitizer + UBSan (CPU build of the host
sources only): a damaged container is rejected, or every plane it describes lies inside the file."""
import os
import subprocess

import cases
from tmc2rs import container

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(REPO, "tmc2-rs_amd", "csrc")


def test_container_reader_survives_mutated_files(tmp_path):
    exe = tmp_path / "fuzz_container"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I", os.path.join(REPO, "include"), "-I", CSRC, "-o", str(exe),
                           os.path.join(HERE, "fuzz_container.cpp"), os.path.join(CSRC, "decoder_input.cpp"),
                           os.path.join(CSRC, "v3c_syntax.cpp"), os.path.join(CSRC, "bitstream.cpp"),
                           os.path.join(CSRC, "vpcc_host.cpp")])
    seed = tmp_path / "seed.vpccgof"
    container.write_container(seed, [[cases.PARITY_CASES["small0"](), cases.PARITY_CASES["block8_ragged"]()],
                                     [cases.PARITY_CASES["no_attribute"]()]])
    out = subprocess.run([str(exe), str(seed), "20000"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "iterations 20000" in out.stdout and "rejected" in out.stdout, out.stdout
