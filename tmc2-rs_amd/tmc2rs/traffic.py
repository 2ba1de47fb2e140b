"""Byte accounting of a frame's reconstruction (host side, numpy; used by bench.py and tools/).

Three figures for the plane READS of one frame, all derived from the patch table and the occupancy plane:
  block_bytes  : whole 16x16 blocks that are owned by a patch and hold any occupancy, x 10 B per pixel
                 (2 geometry + 2 attribute-luma layers x 2 B, chroma 2 layers x 2 planes x 2 B / 4) —
                 the "necessary bytes" of VERDICT r01;
  pixel_bytes  : only the samples of occupied pixels of owned blocks;
  seg{32,64,128}: distinct 32/64/128-byte segments of the raster planes that hold such samples x segment
                 size — HBM is read in whole segments, so this is the floor for ANY kernel that reads the
                 planes in the decoder's raster layout (row pitch = width, planes 128-B aligned).
Outputs are 9 B per point; the occupancy plane is read whole.
"""
import numpy as np

from .synth import canvas_bbox_blocks


def cover_map(fr):
    """cover[by, bx] = highest patch index whose bounding box covers the block, -1 if none
    (as k_plan_tiles finds them: Default/Swap patches)."""
    R = fr["occupancy_resolution"]
    bw, bh = fr["width"] // R, fr["height"] // R
    cover = np.full((bh, bw), -1, dtype=np.int32)
    for i, p in enumerate(fr["patches"]):
        x, y, w, h = canvas_bbox_blocks(p)
        cover[y:y + h, x:x + w] = np.maximum(cover[y:y + h, x:x + w], i)
    return cover


def frame_read_bytes(fr):
    R, prec = fr["occupancy_resolution"], fr["occupancy_precision"]
    W, H, M = fr["width"], fr["height"], fr["map_count"]
    has_attr = int(fr.get("attribute_count", 1)) > 0
    spb = R // prec                                               # occupancy samples per block side
    cover = cover_map(fr)
    occ = fr["occupancy"][:H // prec, :W // prec] != 0
    need = occ & np.kron(cover >= 0, np.ones((spb, spb), dtype=bool))    # at occupancy-sample resolution
    blocks_any = need.reshape(H // R, spb, W // R, spb).any(axis=(1, 3))
    per_px = 2 * M + (2 * M + M if has_attr else 0)
    out = {"items": int((cover >= 0).sum()), "blocks_occupied": int(blocks_any.sum()),
           "pixels": int(need.sum()) * prec * prec,
           "block_bytes": int(blocks_any.sum()) * R * R * per_px,
           "occupancy_plane": int(fr["occupancy"].size)}
    # a pixel row is `prec` identical copies of its occupancy row; chroma rows pair two pixel rows
    rows_l = prec                                                 # pixel rows per occupancy row
    chroma_rows = max(prec // 2, 1) if prec >= 2 else 1           # chroma rows per occupancy row (prec >= 2)
    out["pixel_bytes"] = out["pixels"] * (2 * M + (2 * M if has_attr else 0)) + \
        (out["pixels"] // 4) * (4 * M if has_attr else 0)
    for seg in (32, 64, 128):
        lp = seg // 2                                             # luma pixels per segment
        so = max(lp // prec, 1)                                   # occupancy samples per luma segment
        luma = need.reshape(need.shape[0], -1, so).any(axis=2).sum() * rows_l
        sc = max(2 * lp // prec, 1)                               # occupancy samples per chroma segment
        wpad = (-need.shape[1]) % sc
        nc = np.pad(need, ((0, 0), (0, wpad)))
        chroma = nc.reshape(nc.shape[0], -1, sc).any(axis=2).sum() * chroma_rows
        out["seg%d" % seg] = int(luma) * seg * (M + (M if has_attr else 0)) + (int(chroma) * seg * 2 * M if has_attr else 0)
    return out


def gof_read_bytes(frames):
    tot = {}
    for fr in frames:
        for k, v in frame_read_bytes(fr).items():
            tot[k] = tot.get(k, 0) + v
    return tot
