"""What produced a measurement: the shared object that was loaded and the kernel sources + compiler flags it was
built from.  bench.py prints both; tools/traffic_json.py records them next to the counter bytes, so that a bench
line can tell whether the committed traffic figure belongs to the kernel it timed (`traffic_stale`)."""
import hashlib
import os
import re

from . import _abi

# every file the tile / smoothing kernels are compiled from — and the host code that decides what they are given: the
# work lists (vpcc_kernels.hip: k_plan_*; vpcc_host.cpp), where the planes and outputs lie and how launches are shaped
# (vpcc_gof_create.hip, vpcc_gof.hip, vpcc_pool.hip) change a kernel's memory traffic as surely as its own source does
KERNEL_SOURCES = ("vpcc_tiles.hip", "vpcc_smooth.hip", "vpcc_kernels.hip", "vpcc_device.hpp", "vpcc_devfn.hpp",
                  "vpcc_colour.h", "vpcc_gof_create.hip", "vpcc_gof.hip", "vpcc_gof_smooth.hip", "vpcc_pool.hip", "vpcc_host.cpp")
# ... and per kernel family (a traffic figure of the tile kernel does not go stale when a smoothing kernel changes)
FAMILY_SOURCES = {"k_recon_tiles": ("vpcc_tiles.hip", "vpcc_kernels.hip", "vpcc_device.hpp", "vpcc_devfn.hpp", "vpcc_colour.h",
                                    "vpcc_gof_create.hip", "vpcc_gof.hip", "vpcc_pool.hip", "vpcc_host.cpp"),
                  "k_smooth": ("vpcc_smooth.hip", "vpcc_device.hpp", "vpcc_devfn.hpp", "vpcc_gof_smooth.hip")}


def _sha16(chunks):
    h = hashlib.sha256()
    for c in chunks:
        h.update(c)
    return h.hexdigest()[:16]


def hipflags():
    """The HIPFLAGS line of the Makefile (the flags `make` builds the product library with)."""
    lines = open(os.path.join(_abi.REPO_ROOT, "Makefile")).read().split("\n")
    for i, l in enumerate(lines):
        if re.match(r"HIPFLAGS\s*:=", l):
            text = l.split(":=", 1)[1]
            while text.rstrip().endswith("\\") and i + 1 < len(lines):
                i += 1
                text = text.rstrip()[:-1] + " " + lines[i]
            return " ".join(text.split())
    return ""


def kernel_source_sha16(family=None):
    """sha of the kernel sources + HIPFLAGS: of all kernels, or of one family ("k_recon_tiles", "k_smooth")."""
    csrc = os.path.join(_abi.PROJECT_DIR, "csrc")
    chunks = [open(os.path.join(csrc, n), "rb").read() for n in (FAMILY_SOURCES[family] if family else KERNEL_SOURCES)]
    chunks.append(hipflags().encode())
    return _sha16(chunks)


def library():
    """{name, sha16, diagnostic} of the shared object this process loads."""
    path = _abi.LIB_PATH
    return {"name": os.path.relpath(path, _abi.REPO_ROOT), "sha16": _sha16([open(path, "rb").read()]),
            "diagnostic": os.path.basename(path) != "libvpcc_recon.so", "kernel_source_sha16": kernel_source_sha16()}
