"""The only independent check of the oracle that this environment allows (the reference cannot be built and holds
no fixture for the path — "parity unpinned"): the C restatement oracle/vpcc_oracle.c against
the separately written pure-Python restatement tests/pyref.py, on the 40-frame random sweep the GPU parity test
uses and on one FULL-SIZE S-longdress frame (1280x1408, ~800 k points).  Every array the reference materialises
is compared: positions, colours, 16-bit colours, partition, point_to_pixel, block_to_patch."""
import numpy as np

import cases
import oracle_binding as ob
import pyref
from tmc2rs import synth


def _compare(f):
    st, r = ob.reconstruct(f)
    assert st == 0
    pr = pyref.reconstruct(f)
    n = len(pr["positions"])
    assert r["n"] == n
    assert np.array_equal(ob.xyz_array(r), np.asarray(pr["positions"], np.uint16).reshape(n, 3))
    assert np.array_equal(np.asarray(r["partition"], np.int64), np.asarray(pr["partition"], np.int64))
    assert np.array_equal(np.asarray(r["point_to_pixel"], np.int64).reshape(n, 3), np.asarray(pr["point_to_pixel"], np.int64).reshape(n, 3))
    assert np.array_equal(np.asarray(r["block_to_patch"], np.int64), np.asarray(pr["block_to_patch"], np.int64))
    if f.get("attribute_count", 1) > 0:
        assert np.array_equal(ob.rgb_array(r), np.asarray(pr["colors"], np.uint8).reshape(n, 3))
        assert np.array_equal(np.asarray(r["colors16"], np.int64).reshape(n, 3), np.asarray(pr["colors16"], np.int64).reshape(n, 3))
    return n


def test_random_sweep_oracle_equals_pyref():
    total = sum(_compare(f) for f in cases.random_sweep_frames())
    assert total > 500_000


def test_full_size_longdress_frame_oracle_equals_pyref():
    assert _compare(synth.longdress_frame(5)) > 700_000


def test_full_size_owlii_frame_oracle_equals_pyref():
    """BASELINE config 5's shape (2048x2048, 11-bit coordinates, ~2 M points) through both restatements."""
    assert _compare(synth.owlii_frame(2)) > 1_800_000
