import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tmc2-rs_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    # counting devices does not initialise the GPU on this image
    try:
        import torch
        return torch.cuda.device_count() > 0
    except Exception:
        return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """The oracle is test infrastructure: build it if missing (gcc, < 1 s)."""
    so = os.path.join(REPO, "oracle", "libvpcc_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", REPO, "oracle"])
