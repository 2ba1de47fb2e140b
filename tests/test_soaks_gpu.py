"""Short runs of the differential soaks under tools/ (the long ones are recorded in profiles/r04/soak.txt): random frames
against the oracle, random filter parameters against the smoothing specification, random streams through the Decoder, random
V3C streams through writer, parser and Decoder.  Each runs in a process of its own, as the tools do."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
@pytest.mark.parametrize("tool,args", [("soak_parity.py", ["600", "11"]), ("soak_smoothing.py", ["40", "11"]),
                                       ("soak_decoder.py", ["24", "11"]), ("soak_v3c.py", ["60", "11"])])
def test_soak(tool, args):
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", tool)] + args, capture_output=True, text=True, timeout=580)
    tail = "\n".join((r.stdout + r.stderr).strip().splitlines()[-6:])
    assert r.returncode == 0 and "MISMATCH" not in r.stdout, tail
