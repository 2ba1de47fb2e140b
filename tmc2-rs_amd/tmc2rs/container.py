"""Decoded-GOF container (.vpccgof): patch tables + decoded planes of every frame — the state the
reference holds after its three `decompress()` calls (src/decoder.rs:82-171).  It stands in for the
V3C bitstream + HEVC decoder, which are outside this round's scope; the C++ `tmc2rs::Decoder`
(tmc2-rs_amd/csrc/decoder.cpp) reads it.  Layout: see decoder.cpp (every section padded to 8 bytes)."""
import struct

import numpy as np

from ._abi import PATCH_DTYPE

MAGIC = b"VPCCGOF1"


def _pad(b):
    return b + b"\0" * ((-len(b)) % 8)


def serialize_frame(fr):
    """One frame record of the container (header, patch table, planes; every section padded to 8 bytes)."""
    occ = np.ascontiguousarray(fr["occupancy"], dtype=np.uint8)
    geo = [np.ascontiguousarray(g, dtype=np.uint16) for g in fr["geometry"]]
    m = int(fr.get("map_count", 2))
    ac = int(fr.get("attribute_count", 1))
    attr = [tuple(np.ascontiguousarray(p, dtype=np.uint16) for p in a) for a in fr["attribute"]] if ac else []
    aw, ah = (attr[0][0].shape[1], attr[0][0].shape[0]) if ac else (0, 0)
    patches = np.ascontiguousarray(fr["patches"], dtype=PATCH_DTYPE)
    parts = [struct.pack("<16I", fr["width"], fr["height"], fr["occupancy_resolution"],
                         fr["occupancy_precision"], m, int(fr.get("absolute_d1", 1)), ac,
                         int(fr.get("flags", 0)), occ.shape[1], occ.shape[0], geo[0].shape[1],
                         geo[0].shape[0], aw, ah, len(patches), 0),
             _pad(patches.tobytes()), _pad(occ.tobytes())]
    for i in range(m):
        parts.append(_pad(geo[i].tobytes()))
    for i in range(m if ac else 0):
        for p in attr[i]:
            parts.append(_pad(p.tobytes()))
    return b"".join(parts)


def write_container(path, gofs):
    """gofs: list of GOFs, each a list of frame dicts (see synth.make_frame).  A frame object that occurs
    several times (long streams cycle over a few distinct frames) is serialised once."""
    cache = {}
    with open(path, "wb") as f:
        f.write(_pad(MAGIC + struct.pack("<II", 1, len(gofs))))
        for frames in gofs:
            f.write(_pad(struct.pack("<I", len(frames))))
            for fr in frames:
                b = cache.get(id(fr))
                if b is None:
                    b = cache[id(fr)] = serialize_frame(fr)
                f.write(b)
