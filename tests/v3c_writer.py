"""Test infrastructure: writes V3C sample streams in the syntax the reference's reader consumes
(src/bitstream/reader.rs — element order and widths as cited in tmc2-rs_amd/csrc/v3c_syntax.cpp), so that the
C++ parser can be exercised without real bitstreams (none ship with the reference).
Nothing under tmc2-rs_amd/ imports this module."""


class BitWriter:
    def __init__(self):
        self.bits = []

    def u(self, value, n):
        assert 0 <= value < (1 << n) or n == 0, (value, n)
        self.bits += [(value >> (n - 1 - i)) & 1 for i in range(n)]
        return self

    def flag(self, b):
        return self.u(1 if b else 0, 1)

    def ue(self, v):                       # 0-th order Exp-Golomb, src/bitstream.rs:166-175
        code = v + 1
        n = code.bit_length()
        return self.u(0, n - 1).u(code, n)

    def se(self, v):                       # inverse of read_svlc, src/bitstream.rs:178-185
        return self.ue(2 * v - 1 if v > 0 else -2 * v)

    def byte_align(self):                  # the reader takes one bit, then skips to the byte boundary
        self.bits.append(1)
        while len(self.bits) % 8:
            self.bits.append(0)
        return self

    def bytes(self):
        assert len(self.bits) % 8 == 0, "not byte aligned"
        return bytes(int("".join(map(str, self.bits[i:i + 8])), 2) for i in range(0, len(self.bits), 8))


def vps_payload(p):
    w = BitWriter()
    w.u(0, 5).u(0, 27)                                                   # V3C unit header: VPS
    w.flag(0).u(p.get("profile_codec_group_idc", 1), 7).u(p.get("profile_toolset_idc", 0), 8)
    w.u(p.get("profile_reconstruction_idc", 0), 8).u(0, 32).u(p.get("level_idc", 30), 8)
    w.u(p.get("num_sub_profiles", 0), 6).flag(0).flag(p.get("tool_constraints_present", 0))
    w.u(p.get("vps_id", 0), 4).u(0, 8).u(p.get("atlas_count_minus1", 0), 6).u(0, 6)
    w.ue(p["frame_width"]).ue(p["frame_height"])
    mc = p.get("map_count_minus1", 1)
    w.u(mc, 4)
    if mc > 0:
        w.flag(p.get("multiple_map_streams", 0))
    w.flag(p.get("auxiliary_video_present", 0)).flag(1).flag(1).flag(p.get("attribute_count", 1) > 0 or p.get("force_ai", 0))
    w.u(p.get("occupancy_codec_id", 1), 8).u(p.get("lossy_threshold", 0), 8).u(p.get("occupancy_2d_bitdepth_minus1", 7), 5).flag(0)
    w.u(p.get("geometry_codec_id", 1), 8).u(p.get("geometry_2d_bitdepth_minus1", 7), 5).flag(0)
    w.u(p.get("geometry_3d_bitdepth_minus1", 9), 5)
    if p.get("attribute_count", 1) > 0 or p.get("force_ai", 0):
        w.u(p.get("attribute_count", 1), 7)
        for _ in range(p.get("attribute_count", 1)):
            w.u(0, 4).u(p.get("attribute_codec_id", 1), 8)
            if mc > 0:
                w.flag(1)
            dim = p.get("attribute_dimension_minus1", 2)
            w.u(dim, 6)
            if dim > 0:
                w.u(0, 6)                                                # one partition
            w.u(p.get("attribute_2d_bitdepth_minus1", 7), 5).flag(0)
    w.flag(p.get("vps_extension", 0))
    return w.byte_align().bytes()


def ref_list(w, entries):
    w.ue(len(entries))
    for d, sign in entries:
        w.ue(d)
        if d > 0:
            w.flag(sign)


def asps_nal(p):
    w = BitWriter()
    w.ue(p.get("asps_id", 0)).ue(p["frame_width"]).ue(p["frame_height"])
    w.u(p.get("asps_geometry_3d_bitdepth_minus1", p.get("geometry_3d_bitdepth_minus1", 9)), 5)
    w.u(p.get("geometry_2d_bitdepth_minus1", 7), 5)
    w.ue(p.get("log2_max_afoc_lsb_minus4", 4)).ue(0).flag(0)
    lists = p.get("asps_ref_lists", [[(1, True)]])
    w.ue(len(lists))
    for l in lists:
        ref_list(w, l)
    w.flag(p.get("use_eight_orientations", 0)).flag(p.get("extended_projection", 0))
    w.flag(p.get("normal_axis_limits_quantization", 1)).flag(p.get("normal_axis_max_delta_value", 0))
    w.flag(0).u(p.get("log2_patch_packing_block_size", 4), 3).flag(p.get("patch_size_quantizer_present", 0))
    w.u(p.get("map_count_minus1", 1), 4).flag(p.get("pixel_deinterleaving", 0))
    w.flag(p.get("raw_patch_enabled", 0)).flag(p.get("eom_patch_enabled", 0))
    w.flag(p.get("plr_enabled", 0)).flag(p.get("vui_present", 0))
    ext = p.get("asps_vpcc_extension", True)
    w.flag(ext)
    if ext:
        w.flag(1).u(0, 7).flag(p.get("remove_duplicate_point_enabled", 1))
    return 36, w.byte_align().bytes()


def afps_nal(p):
    w = BitWriter()
    w.ue(p.get("afps_id", 0)).ue(p.get("asps_id", 0))
    w.flag(p.get("single_tile", 1)).flag(p.get("signalled_tile_id", 0))
    w.flag(p.get("output_flag_present", 0)).ue(0).ue(0).flag(p.get("lod_mode_enable", 0)).flag(0).flag(0)
    return 37, w.byte_align().bytes()


def sei_nal(grid_size, threshold, payload_type=66):
    w = BitWriter()
    w.u(payload_type, 8).u(6, 8)
    w.flag(1).flag(0).u(1, 8)                                            # persistence, reset, one instance
    w.u(0, 8).flag(0).ue(1).flag(0).u(grid_size - 2, 7).u(threshold, 8)
    w.byte_align().u(0x80, 8)
    return 45, w.bytes()


def atl_nal(p, frame, patches, nal_type=None, tile_type=1):
    """patches: list of dicts {pos_2d, size_2d_minus1, pos_3d_offset, pos_3d_offset_d, projection_id, orientation}."""
    w = BitWriter()
    t = nal_type if nal_type is not None else (23 if frame == 0 else 1)
    if 16 <= t <= 27:
        w.flag(0)
    w.ue(p.get("afps_id", 0)).ue(0).ue(tile_type)
    if p.get("output_flag_present", 0):
        w.flag(1)
    w.u(frame % (1 << (p.get("log2_max_afoc_lsb_minus4", 4) + 4)), p.get("log2_max_afoc_lsb_minus4", 4) + 4)
    lists = p.get("asps_ref_lists", [[(1, True)]])
    if len(lists) > 0:
        w.flag(1)                                                        # ref_atlas_frame_list_sps_flag
    if len(lists) > 1:
        w.u(0, max(1, (len(lists) - 1).bit_length()))
    q = p.get("pos_min_d_quantizer", 0)
    if tile_type != 2:
        if p.get("normal_axis_limits_quantization", 1):
            w.u(q, 5).u(p.get("pos_delta_max_d_quantizer", 0), 5)
        if tile_type == 0 and len(lists[0]) > 1:
            w.flag(0)
    w.byte_align()
    b3 = p.get("asps_geometry_3d_bitdepth_minus1", p.get("geometry_3d_bitdepth_minus1", 9))
    if tile_type != 2:
        for pa in patches:
            w.ue(pa.get("patch_mode", 0 if tile_type == 1 else 3))
            if pa.get("kind", "intra") == "intra":
                w.ue(pa["pos_2d"][0]).ue(pa["pos_2d"][1]).ue(pa["size_2d_minus1"][0]).ue(pa["size_2d_minus1"][1])
                w.u(pa["pos_3d_offset"][0], b3 + 1).u(pa["pos_3d_offset"][1], b3 + 1)
                w.u(pa["pos_3d_offset_d"], b3 - q + 1)
                if p.get("normal_axis_max_delta_value", 0):
                    w.u(pa.get("pos_3d_range_d", 0), min(p.get("geometry_2d_bitdepth_minus1", 7), b3) + 1 - p.get("pos_delta_max_d_quantizer", 0))
                w.u(pa["projection_id"], 3).u(pa["orientation"], 3 if p.get("use_eight_orientations", 0) else 1)
            elif pa["kind"] == "inter":
                w.se(pa.get("ref_patch_index", 0))
                for v in pa.get("values", [0] * 7):
                    w.se(v)
            elif pa["kind"] == "merge":
                w.flag(1)
                for v in pa.get("values", [0] * 4):
                    w.se(v)
        w.ue(14)
    w.byte_align()
    return t, w.bytes()


def nal_stream(nals, precision=2):
    out = bytearray([(precision - 1) << 5])
    for t, payload in nals:
        body = ((t << 9) | (0 << 3) | 1).to_bytes(2, "big") + payload
        out += len(body).to_bytes(precision, "big") + body
    return bytes(out)


def unit_header(unit_type):
    w = BitWriter()
    w.u(unit_type, 5).u(0, 4).u(0, 6).u(0, 17)
    return w.bytes()


def gof_units(p, frames, videos=(b"occ", b"geometry", b"attribute"), sei=None, extra_nals=()):
    """frames: list of patch lists (one I tile per frame).  Returns the list of V3C unit payloads of one GOF."""
    nals = [asps_nal(p), afps_nal(p)]
    if sei is not None:
        nals.append(sei_nal(*sei))
    nals += list(extra_nals)
    for i, patches in enumerate(frames):
        nals.append(atl_nal(p, i, patches))
    units = [vps_payload(p), unit_header(1) + nal_stream(nals)]
    units.append(unit_header(2) + videos[0])
    units.append(unit_header(3) + videos[1])
    if p.get("attribute_count", 1) > 0:
        units.append(unit_header(4) + videos[2])
    return units


def sample_stream(units, precision=4):
    out = bytearray([(precision - 1) << 5])
    for u in units:
        out += len(u).to_bytes(precision, "big") + u
    return bytes(out)


# ---- whole sequences: synthetic frame dicts (tmc2rs.synth) -> V3C sample stream + raw decoded videos -------
PROJECTION_ID = {(0, 2, 1, 0): 0, (1, 2, 0, 0): 1, (2, 0, 1, 0): 2, (0, 2, 1, 1): 3, (1, 2, 0, 1): 4, (2, 0, 1, 1): 5}


def pdus_of_frame(frame, bits3=10):
    """Intra patch data units that create_patch_frame (src/decoder.rs:415-486) turns back into frame['patches']."""
    out = []
    for q in frame["patches"]:
        pid = PROJECTION_ID[(int(q["normal_axis"]), int(q["tangent_axis"]), int(q["bitangent_axis"]), int(q["projection_mode"]))]
        d = int(q["d1"]) if int(q["projection_mode"]) == 0 else (1 << bits3) - int(q["d1"])
        out.append(dict(pos_2d=(int(q["u0"]), int(q["v0"])), size_2d_minus1=(int(q["size_u0"]) - 1, int(q["size_v0"]) - 1),
                        pos_3d_offset=(int(q["u1"]), int(q["v1"])), pos_3d_offset_d=d, projection_id=pid,
                        orientation=int(q["orientation"])))
    return out


def write_sequence(dirpath, gofs, bits3=10, seis=None):
    """gofs: list of lists of frame dicts of one size.  Writes s.bin, occ.yuv, geo.yuv, attr.yuv; returns the paths.
    seis: per GOF None or (grid_size, threshold) of a prefix geometry-smoothing SEI."""
    import numpy as np
    import os
    f0 = gofs[0][0]
    p = dict(frame_width=f0["width"], frame_height=f0["height"], geometry_3d_bitdepth_minus1=bits3 - 1,
             log2_patch_packing_block_size=int(f0["occupancy_resolution"]).bit_length() - 1,
             map_count_minus1=f0["map_count"] - 1, attribute_count=1 if f0["attribute_count"] else 0,
             use_eight_orientations=1 if any(int(q["orientation"]) > 1 for g in gofs for f in g for q in f["patches"]) else 0)
    units = []
    for k, g in enumerate(gofs):
        units += gof_units(p, [pdus_of_frame(f, bits3) for f in g], sei=seis[k] if seis else None)
    paths = {k: os.path.join(str(dirpath), n) for k, n in (("bin", "s.bin"), ("occ", "occ.yuv"), ("geo", "geo.yuv"), ("attr", "attr.yuv"))}
    with open(paths["bin"], "wb") as o:
        o.write(sample_stream(units))
    with open(paths["occ"], "wb") as o, open(paths["geo"], "wb") as gq, open(paths["attr"], "wb") as a:
        for g in gofs:
            for f in g:
                occ = np.ascontiguousarray(f["occupancy"], dtype=np.uint8)
                o.write(occ.tobytes())
                o.write(bytes(2 * ((occ.shape[1] + 1) // 2) * ((occ.shape[0] + 1) // 2)))       # chroma, never read
                for m in range(f["map_count"]):
                    y = np.ascontiguousarray(f["geometry"][m], dtype="<u2")
                    gq.write(y.tobytes())
                    gq.write(bytes(2 * 2 * (y.shape[1] // 2) * (y.shape[0] // 2)))
                    if f["attribute_count"]:
                        for pl in f["attribute"][m]:
                            a.write(np.ascontiguousarray(pl, dtype="<u2").tobytes())
    if not f0["attribute_count"]:
        paths["attr"] = None
    return paths
