"""Host-side frame validation + planning (tmc2-rs_amd/csrc/vpcc_host.cpp) under AddressSanitizer + UBSan with random
and adversarial patch tables: rejected, or planned into work lists that stay inside the canvas; and the shares of the
resident workgroups per frame of a launch (plan_tile_launch) keep their invariants for random frame sizes; and the free-space
book-keeping of a context's pool (PoolExtents, behind vpcc_ctx_reserve) under random takes and returns."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(REPO, "tmc2-rs_amd", "csrc")


def test_validate_and_plan_survive_random_patch_tables(tmp_path):
    exe = tmp_path / "fuzz_plan"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I", os.path.join(REPO, "include"), "-I", CSRC, "-o", str(exe),
                           os.path.join(HERE, "fuzz_plan.cpp"), os.path.join(CSRC, "vpcc_host.cpp")])
    out = subprocess.run([str(exe), "4000"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "iterations 4000" in out.stdout, out.stdout
    maps = [l for l in out.stdout.splitlines() if l.startswith("launch maps")]
    assert maps and int(maps[0].split()[2].rstrip(",")) > 1000, out.stdout        # the share tables were exercised
