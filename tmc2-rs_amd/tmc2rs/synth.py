"""Deterministic synthetic decoded-plane frames.

No 8iVFB/Owlii bitstreams, no HEVC decoder and no network exist in this
environment, so every BASELINE config is realised as synthetic *decoded*
planes of the same shape: what the reference holds after its three
`decompress` calls (src/decoder.rs:82-171) plus the patch table that
`create_patch_frame` (src/decoder.rs:415-486) would have produced.

All randomness is a counter-based splitmix64 hash of (seed, stream, index), so
frames are bit-reproducible across numpy versions and trivially re-creatable in
any language.  Seed convention: 0x5EED0000 + frame index.
"""
import numpy as np

from ._abi import PATCH_DTYPE, ORIENT_DEFAULT, ORIENT_SWAP

_M = np.uint64(0xFFFFFFFFFFFFFFFF)

# set_view_id, src/decoder.rs:788-796: projection_id -> (normal, tangent, bitangent, mode)
VIEW_AXES = [(0, 2, 1, 0), (1, 2, 0, 0), (2, 0, 1, 0), (0, 2, 1, 1), (1, 2, 0, 1), (2, 0, 1, 1)]


def _mix(z):
    with np.errstate(over="ignore"):
        z = (z + np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def rand_u64(seed, stream, n):
    """n 64-bit values of stream `stream` of generator `seed` (counter-based)."""
    with np.errstate(over="ignore"):
        base = _mix(np.uint64(seed) ^ _mix(np.uint64(stream) * np.uint64(0xD1342543DE82EF95)))
        return _mix(base + np.arange(n, dtype=np.uint64))


def rand_below(seed, stream, n, bound):
    return (rand_u64(seed, stream, n) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53)) * bound


class _Scalar:
    """Sequential scalar draws from one stream (for the packer)."""

    def __init__(self, seed, stream):
        self.v = rand_u64(seed, stream, 1 << 14)
        self.i = 0

    def u(self):
        x = float(self.v[self.i] >> np.uint64(11)) / float(1 << 53)
        self.i += 1
        return x

    def below(self, n):
        return int(self.u() * n)


def _new_patch(rs, x, y, cw, ch, swap, R, max_coord, depth_span):
    view = rs.below(6)
    n_ax, t_ax, b_ax, mode = VIEW_AXES[view]
    su, sv = (ch, cw) if swap else (cw, ch)            # canvas extent of a Swap patch is (sv, su)
    p = np.zeros((), dtype=PATCH_DTYPE)
    p["u0"], p["v0"], p["size_u0"], p["size_v0"] = x, y, su, sv
    p["u1"] = rs.below(max_coord - su * R)
    p["v1"] = rs.below(max_coord - sv * R)
    d = rs.below(max_coord - depth_span - 48)
    p["d1"] = d if mode == 0 else max_coord - d
    p["lod_x"] = p["lod_y"] = 1
    p["normal_axis"], p["tangent_axis"], p["bitangent_axis"], p["projection_mode"] = n_ax, t_ax, b_ax, mode
    p["orientation"] = ORIENT_SWAP if swap else ORIENT_DEFAULT
    return p


def _pack_patches(seed, bw, bh, R, coord_bits, cover_target, max_side, max_patches, swap_prob,
                  overlap_prob, size_skew=8.0, align=1):
    """TMC2-style packer in block units: bounding boxes sorted by area (largest
    first = lowest patch index), each placed at the first raster position where
    it collides with nothing.  Returns a PATCH_DTYPE array."""
    rs = _Scalar(seed, 1)
    max_coord = 1 << coord_bits
    depth_span = 208  # geometry depth after /4 is < 208 (see make_frame)
    sizes, covered = [], 0
    while len(sizes) < max_patches and covered < cover_target * bw * bh:
        cw = min(2 + int((max_side - 1) * rs.u() ** size_skew), bw)
        ch = min(2 + int((max_side - 1) * rs.u() ** size_skew), bh)
        if align > 1:                            # diagnostic layouts only: widths and x positions in whole cache lines
            cw = min(-(-cw // align) * align, bw // align * align)
        sizes.append((cw, ch))
        covered += cw * ch
    sizes.sort(key=lambda s: -(s[0] * s[1]))
    used = np.zeros((bh, bw), dtype=np.int32)
    patches = []
    for (cw, ch) in sizes:
        sat = np.zeros((bh + 1, bw + 1), dtype=np.int32)
        sat[1:, 1:] = used.cumsum(0).cumsum(1)
        win = sat[ch:, cw:] - sat[:-ch, cw:] - sat[ch:, :-cw] + sat[:-ch, :-cw]   # (bh-ch+1, bw-cw+1)
        ok = win == 0
        if align > 1:
            ok[:, np.arange(ok.shape[1]) % align != 0] = False
        free = np.flatnonzero(ok.reshape(-1))
        if len(free) == 0:
            continue
        y, x = divmod(int(free[0]), win.shape[1])
        used[y:y + ch, x:x + cw] = 1
        patches.append(_new_patch(rs, x, y, cw, ch, rs.u() < swap_prob, R, max_coord, depth_span))
        # an overlapping small patch in a corner of the parent's bounding box: exercises the
        # "later patch wins the block" rule (src/codec.rs:242-244) in both index orders
        if rs.u() < overlap_prob and cw >= 4 and ch >= 4:
            cs = 1 + rs.below(2)
            corner = rs.below(4)
            cx = x if corner in (0, 2) else x + cw - cs
            cy = y if corner in (0, 1) else y + ch - cs
            c = _new_patch(rs, cx, cy, cs, cs, rs.u() < swap_prob, R, max_coord, depth_span)
            if rs.u() < 0.5:
                patches.append(c)
            else:
                patches.insert(len(patches) - 1, c)
    return np.array(patches, dtype=PATCH_DTYPE) if patches else np.zeros(0, dtype=PATCH_DTYPE)


def canvas_bbox_blocks(p):
    """Canvas bounding box (x, y, w, h) in blocks of a Default/Swap patch."""
    if int(p["orientation"]) == ORIENT_SWAP:
        return int(p["u0"]), int(p["v0"]), int(p["size_v0"]), int(p["size_u0"])
    return int(p["u0"]), int(p["v0"]), int(p["size_u0"]), int(p["size_v0"])


def make_frame(width=1280, height=1408, precision=4, resolution=16, seed=0x5EED0000, coord_bits=10,
               cover_target=0.42, ellipse_scale=0.86, max_side=24, max_patches=400, swap_prob=0.3,
               overlap_prob=0.12, occupancy_values="one", dup_prob=0.15, patches=None, size_skew=8.0, align=1):
    """One synthetic atlas frame with its decoded planes.

    Returns a dict: width, height, occupancy_resolution, occupancy_precision,
    map_count=2, absolute_d1=1, attribute_count=1, patches (PATCH_DTYPE[]),
    occupancy (u8 [H/prec, W/prec]), geometry [D0, D1] (u16 [H, W]),
    attribute [(Y, U, V), (Y, U, V)] (u16, chroma half-size).
    """
    R, prec = resolution, precision
    bw, bh = width // R, height // R
    ow, oh = width // prec, height // prec
    if patches is None:
        patches = _pack_patches(seed, bw, bh, R, coord_bits, cover_target, max_side, max_patches,
                                swap_prob, overlap_prob, size_skew, align)

    # occupancy: one ellipse per patch bounding box, at occupancy-sample granularity
    occ = np.zeros((oh, ow), dtype=np.uint8)
    spb = R // prec if R >= prec else 1          # occupancy samples per block side
    if occupancy_values == "one":
        vals = None
    else:
        vals = (1 + (rand_u64(seed, 2, oh * ow) % np.uint64(255))).astype(np.uint8).reshape(oh, ow)
    for p in patches:
        bx, by, cw, ch = canvas_bbox_blocks(p)
        sx0, sy0, sw, sh = bx * spb, by * spb, cw * spb, ch * spb
        yy, xx = np.mgrid[0:sh, 0:sw]
        cx, cy = (sw - 1) / 2.0, (sh - 1) / 2.0
        rx, ry = max(sw * 0.5 * ellipse_scale, 0.75), max(sh * 0.5 * ellipse_scale, 0.75)
        m = ((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2 <= 1.0
        sub = occ[sy0:sy0 + sh, sx0:sx0 + sw]
        if vals is None:
            sub[m[:sub.shape[0], :sub.shape[1]]] = 1
        else:
            mm = m[:sub.shape[0], :sub.shape[1]]
            sub[mm] = vals[sy0:sy0 + sh, sx0:sx0 + sw][mm]

    # geometry: D0 = 4*smooth(0..200) + low-bit noise 0..3 (exercises the /4 of codec.rs:534),
    # D1 = D0 + 4*delta, delta in 0..4 with P(0) = dup_prob (exercises duplicate removal)
    n = width * height
    ph = rand_below(seed, 3, 4, 2 * np.pi)
    fx = 2 * np.pi / (180.0 + 60.0 * rand_below(seed, 4, 1, 1.0)[0])
    fy = 2 * np.pi / (140.0 + 80.0 * rand_below(seed, 4, 2, 1.0)[1])
    xs = np.arange(width, dtype=np.float64)[None, :]
    ys = np.arange(height, dtype=np.float64)[:, None]
    smooth = 100.0 + 60.0 * np.sin(fx * xs + ph[0]) * np.cos(fy * ys + ph[1]) + 39.0 * np.sin(
        0.5 * fy * ys + 0.7 * fx * xs + ph[2])
    smooth = np.clip(np.floor(smooth), 0, 200).astype(np.uint16)
    noise = (rand_u64(seed, 5, n) & np.uint64(3)).astype(np.uint16).reshape(height, width)
    d0 = (smooth * np.uint16(4) + noise).astype(np.uint16)
    r = rand_u64(seed, 6, n)
    is_dup = ((r >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))) < dup_prob
    delta = (1 + ((r >> np.uint64(3)) & np.uint64(3))).astype(np.uint16)
    delta[is_dup] = 0
    d1 = (d0 + np.uint16(4) * delta.reshape(height, width)).astype(np.uint16)

    # attribute: Y in [64,940], U,V in [64,960], independent per layer
    attr = []
    cw2, ch2 = width // 2, height // 2
    for layer in range(2):
        y = (64 + rand_u64(seed, 10 + layer, n) % np.uint64(940 - 64 + 1)).astype(np.uint16).reshape(height, width)
        u = (64 + rand_u64(seed, 20 + layer, cw2 * ch2) % np.uint64(960 - 64 + 1)).astype(np.uint16).reshape(ch2, cw2)
        v = (64 + rand_u64(seed, 30 + layer, cw2 * ch2) % np.uint64(960 - 64 + 1)).astype(np.uint16).reshape(ch2, cw2)
        attr.append((y, u, v))

    return {
        "width": width, "height": height,
        "occupancy_resolution": R, "occupancy_precision": prec,
        "map_count": 2, "absolute_d1": 1, "attribute_count": 1, "flags": 0,
        "patches": patches, "occupancy": occ, "geometry": [d0, d1], "attribute": attr,
        "seed": seed,
    }


def longdress_frame(index, **kw):
    """S-longdress: 1280x1408, occupancy 320x352 (precision 4), R=16, ~800 k points."""
    return make_frame(1280, 1408, 4, 16, seed=0x5EED0000 + index, coord_bits=10, **kw)


def owlii_frame(index, **kw):
    """S-owlii: 2048x2048, occupancy 512x512, 11-bit coordinates, ~2 M points."""
    kw.setdefault("cover_target", 0.455)
    kw.setdefault("max_side", 40)
    return make_frame(2048, 2048, 4, 16, seed=0x0E110000 + index, coord_bits=11, **kw)


def small_frame(index, width=64, height=64, precision=4, resolution=16, **kw):
    """Tiny frames for golden fixtures and edge cases."""
    kw.setdefault("max_side", 3)
    kw.setdefault("size_skew", 1.0)
    kw.setdefault("cover_target", 0.8)
    kw.setdefault("overlap_prob", 0.4)
    return make_frame(width, height, precision, resolution, seed=0x51A11000 + index, coord_bits=10, **kw)
