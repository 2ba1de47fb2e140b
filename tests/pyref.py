"""Second, independent restatement of the reference hot path in pure Python
loops — for SMALL cases only.  It exists to cross-check the C oracle
(oracle/vpcc_oracle.c): two restatements written separately from the Rust text
must agree on every array.  TEST INFRASTRUCTURE ONLY.

Follows: src/codec.rs:205-250, 256-514, 517-565, 569-658, 661-687 and
src/decoder.rs:788-888, 971-1021 of benclmnt/tmc2-rs.
"""
import math

M64 = (1 << 64) - 1


class RefPanic(Exception):
    """Where the Rust reference would panic (assert!/unwrap/unimplemented!)."""


def patch_to_canvas_helper(p, u, v, resolution):
    # src/decoder.rs:853-867, release-profile wrapping usize arithmetic
    u0, v0 = int(p["u0"]) * resolution, int(p["v0"]) * resolution
    su, sv = int(p["size_u0"]), int(p["size_v0"])
    o = int(p["orientation"])
    if o == 0:
        x, y = u + u0, v + v0
    elif o == 2:   # Rot90
        x, y = sv - 1 - v + u0, u + v0
    elif o == 3:   # Rot180
        x, y = su - 1 - u + u0, sv - 1 - v + v0
    elif o == 4:   # Rot270
        x, y = v + u0, su - 1 - u + v0
    elif o == 5:   # Mirror
        x, y = su - 1 - u + u0, v + v0
    elif o == 6:   # MRot90
        x, y = sv - 1 - v + u0, su - 1 - u + v0
    elif o == 7:   # MRot180
        x, y = u + u0, sv - 1 - v + v0
    elif o in (8, 1):  # MRot270, Swap
        x, y = v + u0, u + v0
    else:
        raise RefPanic("orientation")
    return x & M64, y & M64


def generate_point(p, u, v, depth):
    # src/decoder.rs:871-888
    pt = [0, 0, 0]
    d1 = int(p["d1"])
    if int(p["projection_mode"]) == 0:
        n = depth + d1
    else:
        n = max(d1, depth) - depth
    pt[int(p["normal_axis"])] = n & 0xFFFF
    pt[int(p["tangent_axis"])] = (u * int(p["lod_x"]) + int(p["u1"])) & 0xFFFF
    pt[int(p["bitangent_axis"])] = (v * int(p["lod_y"]) + int(p["v1"])) & 0xFFFF
    return tuple(pt)


def yuv10_to_rgb8(y, u, v):
    # src/codec.rs:661-687 (Python floats are IEEE f64, no contraction)
    def clamp(x):
        if x < 0.:
            return 0
        if x > 255.:
            return 255
        return int(x)
    y, u, v = float(y), float(u), float(v)
    r = y + 1.57480 * (v - 512.)
    g = y - 0.18733 * (u - 512.) - (0.46813 * (v - 512.))
    b = y + 1.85563 * (u - 512.)
    return (clamp(math.floor(r / 1023. * 255.)), clamp(math.floor(g / 1023. * 255.)),
            clamp(math.floor(b / 1023. * 255.)))


def _occ_get(occ, u, v):
    h, w = occ.shape
    if not (u < w and v < h):
        raise RefPanic("occupancy bounds")
    return int(occ[v, u])


def block_to_patch(frame):
    # src/codec.rs:205-250
    W, H = frame["width"], frame["height"]
    R, prec = frame["occupancy_resolution"], frame["occupancy_precision"]
    bw, bh = W // R, H // R
    b2p = [0] * (bw * bh)
    for pi, p in enumerate(frame["patches"]):
        for v0 in range(int(p["size_v0"])):
            for u0 in range(int(p["size_u0"])):
                bx, by = patch_to_canvas_helper(p, u0, v0, 1)
                if not (bx < bw and by < bh):
                    raise RefPanic("block out of canvas")
                nz = 0
                for v1 in range(R):
                    v = v0 * R + v1
                    for u1 in range(R):
                        u = u0 * R + u1
                        x, y = patch_to_canvas_helper(p, u, v, R)
                        if not (x < W and y < H):
                            raise RefPanic("pixel out of canvas")
                        nz += _occ_get(frame["occupancy"], x // prec, y // prec)
                if nz > 0:
                    b2p[by * bw + bx] = pi + 1
    return b2p


def reconstruct(frame):
    """Returns dict(positions, colors, colors16, partition, point_to_pixel, block_to_patch, occupancy_map)."""
    W, H = frame["width"], frame["height"]
    R, prec = frame["occupancy_resolution"], frame["occupancy_precision"]
    bw = W // R
    map_count = frame.get("map_count", 2)
    absolute_d1 = frame.get("absolute_d1", 1)
    b2p = block_to_patch(frame)
    occ = frame["occupancy"]
    occupancy_map = [[_occ_get(occ, u // prec, v // prec) for u in range(W)] for v in range(H)]
    g0 = frame["geometry"][0]
    g1 = frame["geometry"][1] if map_count > 1 else None
    positions, partition, p2p = [], [], []
    for pi, p in enumerate(frame["patches"]):
        for v0 in range(int(p["size_v0"])):
            for u0 in range(int(p["size_u0"])):
                bx, by = patch_to_canvas_helper(p, u0, v0, 1)
                if b2p[by * bw + bx] != pi + 1:
                    continue
                for v1 in range(R):
                    v = v0 * R + v1
                    for u1 in range(R):
                        u = u0 * R + u1
                        x, y = patch_to_canvas_helper(p, u, v, R)
                        if not (x < W and y < H):
                            raise RefPanic("pixel out of canvas")
                        if occupancy_map[y][x] == 0:
                            continue
                        if not (x < g0.shape[1] and y < g0.shape[0]):
                            raise RefPanic("geometry bounds")
                        pt0 = generate_point(p, u, v, int(g0[y, x]) // 4)
                        created = [pt0]
                        if map_count > 1:
                            d1 = int(g1[y, x]) // 4
                            if absolute_d1:
                                pt1 = generate_point(p, u, v, d1)
                            else:
                                l = list(pt0)
                                ax = int(p["normal_axis"])
                                l[ax] = (l[ax] + d1) & 0xFFFF if int(p["projection_mode"]) == 0 else (l[ax] - d1) & 0xFFFF
                                pt1 = tuple(l)
                            created.append(pt1)
                        for i, c in enumerate(created):
                            if i != 0 and c == created[0]:
                                continue
                            positions.append(c)
                            partition.append(pi)
                            p2p.append((x, y, i))
    colors16, colors = [], []
    if frame.get("attribute_count", 1) > 0:
        for (x, y, z) in p2p:
            ay, au, av = frame["attribute"][z]
            if not (x < ay.shape[1] and y < ay.shape[0]):
                raise RefPanic("attribute bounds")
            cw = ay.shape[1] // 2
            yy = int(ay[y, x])
            uu = int(au.reshape(-1)[(y // 2) * cw + (x // 2)])
            vv = int(av.reshape(-1)[(y // 2) * cw + (x // 2)])
            colors16.append((yy, uu, vv))
            colors.append(yuv10_to_rgb8(yy, uu, vv))
    return {"positions": positions, "colors": colors, "colors16": colors16, "partition": partition,
            "point_to_pixel": p2p, "block_to_patch": b2p, "occupancy_map": occupancy_map}
