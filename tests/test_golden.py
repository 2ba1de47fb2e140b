"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py): the oracle must keep
reproducing them (CPU), and the HIP path must reproduce them bit for bit (GPU)."""
import glob
import os

import numpy as np
import pytest

import oracle_binding as ob
from golden.make_golden import GOLDEN, unpack

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))


def test_every_fixture_is_present():
    assert sorted(os.path.basename(f)[:-4] for f in FILES) == sorted(GOLDEN)


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_oracle_reproduces_golden(path):
    z = np.load(path)
    st, ref = ob.reconstruct(unpack(z))
    assert st == 0
    assert np.array_equal(ob.xyz_array(ref), z["out_xyz"])
    assert np.array_equal(ob.rgb_array(ref), z["out_rgb"])
    assert np.array_equal(ref["partition"].astype(np.uint32), z["out_partition"])
    assert np.array_equal(ref["block_to_patch"].astype(np.uint32), z["out_block_to_patch"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_hip_reproduces_golden(path):
    from tmc2rs import recon
    z = np.load(path)
    f = unpack(z)
    ctx = recon.Context(0)
    res = ctx.reconstruct_frame(f, want_patch_index=True)
    assert res["n"] == len(z["out_xyz"])
    assert np.array_equal(res["xyz"], z["out_xyz"])
    if f["attribute_count"]:
        assert np.array_equal(res["rgb"], z["out_rgb"])
    assert np.array_equal(res["patch_index"].astype(np.uint32), z["out_partition"])
    assert np.array_equal(ctx.generate_block_to_patch(f), z["out_block_to_patch"])
    ctx.close()
