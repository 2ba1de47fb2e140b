"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/vpcc_recon.h declares, and its host-only entry points behave (no compute calls here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import cases
import oracle_binding as ob
from tmc2rs import _abi, recon, synth

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(REPO, "include", "vpcc_recon.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vpcc_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _abi.load_library()
    names = _declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/vpcc_recon.h but not exported"
    assert lib.vpcc_abi_version() == 5


def test_struct_layout_matches_header():
    # sizes the Rust #[repr(C)] mirror in INTEGRATION.md relies on
    assert C.sizeof(_abi.Patch) == 44
    assert C.sizeof(_abi.ImageU8) == 24
    assert C.sizeof(_abi.ImageU16) == 40
    assert C.sizeof(_abi.FrameDesc) == 32 + 24 + 2 * 40 + 2 * 40 + 16
    assert _abi.POINT3_DTYPE.itemsize == 6 and _abi.COLOR3_DTYPE.itemsize == 3


def test_status_strings():
    lib = _abi.load_library()
    for s in range(9):
        assert lib.vpcc_status_string(s)
    assert b"CPU fallback" in lib.vpcc_status_string(_abi.VPCC_ERR_NO_DEVICE)


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(recon.VpccError) as e:
        recon.Context(0)
    assert e.value.status == _abi.VPCC_ERR_NO_DEVICE


@pytest.mark.parametrize("name", sorted(cases.PARITY_CASES))
def test_validate_accepts_every_parity_case(name):
    assert recon.validate_frame(cases.PARITY_CASES[name]()) == 0


def test_validate_mirrors_reference_panics():
    occ = np.ones((8, 8), np.uint8)
    P, T = cases._patch, cases._tiny_frame
    bad = [
        (T([P(1, 1, 2, 1)], occ), 3),                    # block outside canvas: assert decoder.rs:835
        (T([P(0, 0, 1, 1, orient=3)], occ), 3),          # Rot180 underflow at pixel level: decoder.rs:848
        (T([P(0, 0, 1, 1, orient=2)], occ), 3),          # Rot90 likewise
    ]
    p = P(0, 0, 1, 1)
    p["axis_of_additional_plane"] = 1
    bad.append((T([p], occ), 2))                          # unimplemented!() codec.rs:437
    f = T([P(0, 0, 1, 1)], occ)
    f["geometry"] = [f["geometry"][0]]
    bad.append((f, 4))                                    # geometry video too short: codec.rs:318-320
    f = T([P(0, 0, 1, 1)], occ)
    f["map_count"] = 3
    bad.append((f, 2))
    f = T([P(0, 0, 1, 1)], np.ones((7, 8), np.uint8))     # occupancy plane too small for the upsample
    bad.append((f, 3))
    for frame, status in bad:
        assert recon.validate_frame(frame) == status
        st, _ = ob.reconstruct(frame)
        assert st == status                               # the oracle agrees on which panic it is


def test_capacity_bound():
    lib = _abi.load_library()
    d, keep = _abi.host_frame_desc(synth.small_frame(0))
    assert lib.vpcc_frame_capacity_bound(C.byref(d)) == 2 * 64 * 64


def test_product_library_has_no_diagnostic_switches():
    """The timing ablations / in-kernel stamps of the tile kernel exist in libvpcc_recon_diag.so only
    (make diag, -DVPCC_DIAGNOSTIC): the product neither exports the diagnostic entry point nor knows the
    environment variables that select an ablation."""
    lib = _abi.load_library()
    assert not hasattr(lib, "vpcc_debug_read_stamps")
    blob = open(_abi.LIB_PATH, "rb").read()
    for name in (b"VPCC_TILES_VARIANT", b"VPCC_TILES_DEPTH"):
        assert name not in blob, name
