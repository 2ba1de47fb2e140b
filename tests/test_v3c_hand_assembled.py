"""A complete V3C sample stream assembled BY HAND, bit field by bit field, from the order in which the reference's
reader consumes them (src/bitstream/reader.rs; line numbers per field below) — independent of tests/v3c_writer.py.
The bit strings are written out literally; their concatenation must equal the committed hex bytes, and the C++
parser (v3c_syntax.cpp) must read the same values back: VPS (+PTL/OI/GI/AI), NAL sample stream, ASPS with a
reference list, AFPS with its tile information, an IDR_N_LP tile header, two intra patch data units and the
patch derivation of src/decoder.rs:415-486.  ue(v) codes follow src/bitstream.rs:170-179."""
from tmc2rs import recon


def ue(v):                                  # Exp-Golomb: (leading zeros) 1 (suffix), src/bitstream.rs:170-179
    b = bin(v + 1)[2:]
    return "0" * (len(b) - 1) + b


def u(v, n):
    return format(v, "0%db" % n) if n else ""


def pack(bits, trailing=True):
    if trailing:                            # byte_align(): reads ONE bit, then skips to the byte boundary (bitstream.rs:112-119)
        bits += "1"
    bits += "0" * (-len(bits) % 8)
    return bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8))


# ---------------------------------------------------------------- V3C parameter set (reader.rs:35-80, 257-340, 525-570)
VPS_BITS = "".join([
    u(0, 5), u(0, 27),                      # vuh_unit_type = V3C_VPS, 27 reserved bits           :36, :69
    u(0, 1), u(1, 7), u(0, 8), u(0, 8),     # ptl: tier, codec group 1 (HEVC Main10), toolset, reconstruction  :527-530
    u(0, 32),                               # four bytes skipped (reserved)                        :534-537
    u(30, 8), u(0, 6), u(0, 1), u(0, 1),    # level 30, num_sub_profiles 0, extended flag, tool constraints   :539-562
    u(0, 4), u(0, 8), u(0, 6), u(0, 6),     # vps id, 8 zero bits, atlas_count_minus1, atlas id    :263-272
    ue(64), ue(32),                         # frame_width 64, frame_height 32                      :273-274
    u(1, 4), u(0, 1),                       # map_count_minus1 1, multiple_map_streams_present 0   :275-278
    u(0, 1), u(1, 1), u(1, 1), u(1, 1),     # auxiliary 0, occupancy / geometry / attribute video present :297-300
    u(0, 8), u(0, 8), u(7, 5), u(0, 1),     # oi: codec 0, lossy threshold 0, 2-D bit depth 8, msb align 0   :364-371
    u(0, 8), u(7, 5), u(0, 1), u(9, 5),     # gi: codec 0, 2-D bit depth 8, msb align 0, 3-D bit depth 10    :397-410
    u(1, 7), u(0, 4), u(0, 8), u(1, 1),     # ai: one attribute, type texture, codec 0, persistence flag (map_count_minus1 > 0) :449-459
    u(2, 6), u(0, 6), u(7, 5), u(0, 1),     #     dimension_minus1 2, partitions_minus1 0, 2-D bit depth 8, msb align 0        :460-478
    u(0, 1),                                # extension_present_flag                               :324
])
VPS_HEX = "00000000010000000000001e000000000208211380001c0072408004200e40"

# ---------------------------------------------------------------- atlas data unit
AD_HEADER = pack(u(1, 5) + u(0, 4) + u(0, 6) + u(0, 17), trailing=False)      # V3C_AD, vps id 0, atlas 0, 17 reserved  :36-66
NAL_STREAM_HEADER = pack(u(3, 3) + u(0, 5), trailing=False)                   # 4-byte NAL sizes          :745-749


def nal_header(nal_type):                   # forbidden zero bit, type, layer id, temporal id + 1   :772-778
    return pack(u(0, 1) + u(nal_type, 6) + u(0, 6) + u(1, 3), trailing=False)


ASPS_BITS = "".join([
    ue(0), ue(64), ue(32),                  # asps id 0, frame 64 x 32                             :1024-1026
    u(9, 5), u(7, 5),                       # geometry_3d_bitdepth_minus1 9, geometry_2d_bitdepth_minus1 7  :1027-1028
    ue(0), ue(0), u(0, 1),                  # log2_max_afoc_lsb_minus4 0, max_dec_buffering_minus1 0, long_term 0  :1029-1031
    ue(1),                                  # num_ref_atlas_frame_lists_in_asps 1                  :1032
    ue(1), ue(1), u(1, 1),                  #   ref list: one entry, abs_delta_afoc_st 1, sign 1   :1135-1156
    u(1, 1), u(0, 1),                       # use_eight_orientations 1, extended_projection 0      :1041-1042
    u(1, 1), u(0, 1), u(0, 1),              # normal_axis_limits_quantization 1, max_delta 0, precedence order 0  :1047-1049
    u(4, 3), u(0, 1), u(1, 4), u(0, 1),     # log2 block size 4, size quantizer 0, map_count_minus1 1, deinterleaving 0  :1050-1054
    u(0, 1), u(0, 1), u(0, 1), u(0, 1),     # raw, eom, plr, vui                                   :1065-1090
    u(1, 1), u(1, 1), u(0, 7), u(1, 1),     # extension, vpcc extension, 7 extension bits, remove_duplicate_point 1  :1096-1104
])
ASPS_HEX = "810410a4f925a4083018"

AFPS_BITS = "".join([
    ue(0), ue(0),                           # afps id 0, asps id 0                                 :1194-1195
    u(1, 1), u(0, 1),                       # afti: single tile 1, signalled tile id 0             :1261, :1279
    u(0, 1), ue(0), ue(0),                  # output_flag_present 0, num_ref_idx_default_active_minus1 0, additional_lt_afoc_lsb_len 0  :1200-1202
    u(0, 1), u(0, 1), u(0, 1),              # lod_mode_enable 0, raw_3d_offset explicit 0, extension 0  :1203-1205
])
AFPS_HEX = "e620"

TILE_BITS = "".join([                       # NAL type IDR_N_LP (23): inside [BLA_W_LP, GCRA]
    u(0, 1),                                # no_output_of_prior_atlas_frames_flag                 :1576
    ue(0), ue(0),                           # afps id 0, aaps id 0                                 :1586-1587
    ue(1),                                  # tile type I (no tile id: single tile)                :1603
    u(0, 4),                                # atlas_frame_order_cnt_lsb (4 bits), frame 0          :1610-1611
    u(1, 1),                                # ref_atlas_frame_list_sps_flag (one list in the ASPS) :1612-1613
    u(2, 5), u(0, 5),                       # pos_min_d_quantizer 2, pos_delta_max_d_quantizer 0   :1657-1661
])                                          # then byte_align()                                    :1682
TILE_HEX = "682204"

PDUS = [dict(pos_2d=(1, 2), size_2d_minus1=(2, 0), pos_3d_offset=(300, 17), pos_3d_offset_d=25, projection_id=4, orientation=1),
        dict(pos_2d=(0, 0), size_2d_minus1=(0, 1), pos_3d_offset=(5, 1000), pos_3d_offset_d=200, projection_id=2, orientation=6)]
DATA_BITS = "".join(
    "".join([ue(0),                                                        # patch mode I_INTRA           :1801
             ue(p["pos_2d"][0]), ue(p["pos_2d"][1]),                       # pos_2d                       :1881-1884
             ue(p["size_2d_minus1"][0]), ue(p["size_2d_minus1"][1]),       # size_2d_minus1               :1885-1888
             u(p["pos_3d_offset"][0], 10), u(p["pos_3d_offset"][1], 10),   # 10 = geometry_3d_bitdepth_minus1 + 1  :1889-1892
             u(p["pos_3d_offset_d"], 8),                                   # 8 = 9 - pos_min_d_quantizer + 1       :1893
             u(p["projection_id"], 3),                                     # ceil(log2(max_number_projections_minus1 + 1)) = 3 (default 5)  :1904-1906
             u(p["orientation"], 3)])                                      # three bits with eight orientations    :1908-1913
    for p in PDUS) + ue(14)                                                # patch mode I_END             :1801, :1851
DATA_HEX = "a6e96022330fa017e8c8587c"


def test_bit_strings_equal_the_committed_bytes():
    assert pack(VPS_BITS).hex() == VPS_HEX
    assert pack(ASPS_BITS).hex() == ASPS_HEX
    assert pack(AFPS_BITS).hex() == AFPS_HEX
    assert pack(TILE_BITS).hex() == TILE_HEX
    assert pack(DATA_BITS).hex() == DATA_HEX


def hand_stream():
    def sized(payload):
        return len(payload).to_bytes(4, "big") + payload
    vps = bytes.fromhex(VPS_HEX)
    atl = nal_header(23) + bytes.fromhex(TILE_HEX) + bytes.fromhex(DATA_HEX)
    ad = (AD_HEADER + NAL_STREAM_HEADER + sized(nal_header(36) + bytes.fromhex(ASPS_HEX)) +
          sized(nal_header(37) + bytes.fromhex(AFPS_HEX)) + sized(atl))
    ovd = pack(u(2, 5) + u(0, 4) + u(0, 6) + u(0, 17), trailing=False) + b"occupancy"     # V3C_OVD              :62-64
    gvd = pack(u(3, 5) + u(0, 4) + u(0, 6) + u(0, 4) + u(0, 1) + u(0, 12), trailing=False) + b"geo"    # map index, aux flag  :57-61
    avd = pack(u(4, 5) + u(0, 4) + u(0, 6) + u(0, 7) + u(0, 5) + u(0, 4) + u(0, 1), trailing=False) + b"attribute!"   # :51-56
    return pack(u(3, 3) + u(0, 5), trailing=False) + b"".join(sized(x) for x in (vps, ad, ovd, gvd, avd))   # ssvh :643-647


def test_parser_reads_the_hand_assembled_stream():
    s = recon.V3cStream(hand_stream())
    assert s.unit_count() == 5
    info = s.next_gof()
    assert info["frame_count"] == 1 and (info["frame_width"], info["frame_height"]) == (64, 32)
    assert (info["atlas_frame_width"], info["atlas_frame_height"]) == (64, 32)
    assert info["map_count"] == 2 and info["absolute_d1"] == 1            # single stream: D1 absolute (decoder.rs:605)
    assert info["occupancy_resolution"] == 16                            # 1 << log2_patch_packing_block_size
    assert info["geometry_3d_bitdepth"] == 10 and info["atlas_geometry_3d_bitdepth"] == 10
    assert info["geometry_2d_bitdepth"] == 8 and info["occupancy_2d_bitdepth"] == 8 and info["attribute_2d_bitdepth"] == 8
    assert info["attribute_count"] == 1 and info["profile_codec_group_idc"] == 1 and info["level_idc"] == 30
    assert info["use_eight_orientations_flag"] == 1 and info["remove_duplicate_point_enabled_flag"] == 1
    assert info["geometry_smoothing_sei"] == 0
    assert info["video_bytes"] == [9, 3, 10]
    assert s.video(0) == b"occupancy" and s.video(1) == b"geo" and s.video(2) == b"attribute!"
    fi, patches = s.frame_patches(0)
    assert fi == 0 and len(patches) == 2
    # create_patch_frame (decoder.rs:415-486) with minLevel = 1 << pos_min_d_quantizer = 4:
    #   projection 4 -> axes (1, 2, 0), mode 1 -> d1 = 1024 - 25 * 4;  projection 2 -> axes (2, 0, 1), mode 0 -> d1 = 200 * 4
    want = [dict(u0=1, v0=2, size_u0=3, size_v0=1, u1=300, v1=17, d1=1024 - 100, normal_axis=1, tangent_axis=2, bitangent_axis=0,
                 projection_mode=1, orientation=1, lod_x=1, lod_y=1, axis_of_additional_plane=0),
            dict(u0=0, v0=0, size_u0=1, size_v0=2, u1=5, v1=1000, d1=800, normal_axis=2, tangent_axis=0, bitangent_axis=1,
                 projection_mode=0, orientation=6, lod_x=1, lod_y=1, axis_of_additional_plane=0)]
    for got, w in zip(patches, want):
        for k, v in w.items():
            assert getattr(got, k) == v, (k, getattr(got, k), v)
    assert s.next_gof() is None
    s.close()


# ---------------------------------------------------------------- prefix SEI (geometry smoothing) and a second frame
SEI_BITS = "".join([
    u(66, 8),                               # payload type GeometrySmoothing = 66, one byte < 0xff     :1379-1386, enum :1362
    u(5, 8),                                # payload size (read and ignored)                          :1388-1394
    u(1, 1), u(0, 1), u(1, 8),              # persistence 1, reset 0, instances_updated 1              :1470-1473
    u(0, 8), u(0, 1),                       # instance index 0, cancel flag 0                          :1486-1488
    ue(1),                                  # method type 1 (grid smoothing)                           :1492
    u(0, 1), u(6, 7), u(64, 8),             # filter_eom_points 0, grid_size_minus_2 6, threshold 64   :1494-1496
])                                          # byte_align(), then ONE more byte is read                 :1410-1411
SEI_HEX = "4205804008190200"      # 42 05 | 1 0 00000001 00000000 0 010 0 0000110 01000000 | 1 + padding | the extra byte

TILE2_BITS = "".join([                      # NAL type TRAIL_R (1): no no_output_of_prior_atlas_frames_flag
    ue(0), ue(0), ue(1),                    # afps id, aaps id, tile type I                            :1586-1603
    u(1, 4),                                # atlas_frame_order_cnt_lsb = 1: the second frame          :1610-1611
    u(1, 1), u(2, 5), u(0, 5),              # list from the ASPS, quantizers as in the first tile      :1612-1661
])
DATA2_BITS = "".join([ue(0), ue(3), ue(0), ue(0), ue(0), u(1, 10), u(2, 10), u(3, 8), u(0, 3), u(0, 3)]) + ue(14)


def test_hand_assembled_sei_and_second_frame():
    assert pack(SEI_BITS).hex() + "00" == SEI_HEX            # + the extra byte the reference consumes after byte_align
    # 1 1 010 | 0001 | 1 | 00010 | 00000 | trailing 1 -> 1101 0000 1100 0100 0000 1000
    assert pack(TILE2_BITS).hex() == "d0c408"

    def sized(payload):
        return len(payload).to_bytes(4, "big") + payload
    atl1 = nal_header(23) + bytes.fromhex(TILE_HEX) + bytes.fromhex(DATA_HEX)
    atl2 = nal_header(1) + pack(TILE2_BITS) + pack(DATA2_BITS)
    ad = (AD_HEADER + NAL_STREAM_HEADER + sized(nal_header(36) + bytes.fromhex(ASPS_HEX)) +
          sized(nal_header(37) + bytes.fromhex(AFPS_HEX)) + sized(nal_header(43) + bytes.fromhex(SEI_HEX)) +     # PREFIX_NSEI = 43
          sized(atl1) + sized(atl2))
    ovd = pack(u(2, 5) + u(0, 4) + u(0, 6) + u(0, 17), trailing=False) + b"o"
    gvd = pack(u(3, 5) + u(0, 4) + u(0, 6) + u(0, 4) + u(0, 1) + u(0, 12), trailing=False) + b"g"
    avd = pack(u(4, 5) + u(0, 4) + u(0, 6) + u(0, 7) + u(0, 5) + u(0, 4) + u(0, 1), trailing=False) + b"a"
    data = pack(u(3, 3) + u(0, 5), trailing=False) + b"".join(sized(x) for x in (bytes.fromhex(VPS_HEX), ad, ovd, gvd, avd))
    s = recon.V3cStream(data)
    info = s.next_gof()
    assert info["frame_count"] == 2
    assert info["geometry_smoothing_sei"] == 1 and info["smoothing_grid_size"] == 8 and info["smoothing_threshold"] == 64
    fi0, p0 = s.frame_patches(0)
    fi1, p1 = s.frame_patches(1)
    assert (fi0, len(p0)) == (0, 2) and (fi1, len(p1)) == (1, 1)
    g = p1[0]                                                # projection 0: axes (0, 2, 1), mode 0, d1 = 3 * 4
    assert (g.u0, g.v0, g.size_u0, g.size_v0, g.u1, g.v1, g.d1) == (3, 0, 1, 1, 1, 2, 12)
    assert (g.normal_axis, g.tangent_axis, g.bitangent_axis, g.projection_mode, g.orientation) == (0, 2, 1, 0, 0)
    s.close()
