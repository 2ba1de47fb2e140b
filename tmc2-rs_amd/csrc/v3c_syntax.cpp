// v3c_syntax.cpp — see v3c_syntax.hpp.  Syntax element order and widths follow the reference's reader
// (benclmnt/tmc2-rs, src/bitstream/reader.rs), including its deviations from ISO/IEC 23090-5 where the
// two differ (noted at the element); everything the reference rejects is rejected here.
#include "v3c_syntax.hpp"

#include <algorithm>
#include <cstring>

namespace tmc2rs {

namespace {

[[noreturn]] void unsupported(const std::string& what) { throw SyntaxError(VPCC_ERR_UNSUPPORTED, what); }
[[noreturn]] void invalid(const std::string& what) { throw SyntaxError(VPCC_ERR_INVALID_ARG, what); }

// Bit reader with the reference's panics mapped to SyntaxError: reading past the end
// (index out of bounds) or more than 32 bits at once ("Bitstream::read: bits > 32", src/bitstream.rs:138).
class Reader {
 public:
  explicit Reader(Bitstream& bs) : bs_(bs) {}
  uint32_t u(unsigned bits) {
    if (bits > 32) invalid("read of more than 32 bits");
    try {
      return bs_.read(bits);
    } catch (const std::out_of_range&) {
      invalid("bitstream ends inside a syntax element");
    }
  }
  bool flag() { return u(1) != 0; }
  uint32_t ue() {
    try {
      return bs_.read_uvlc();
    } catch (const std::out_of_range&) {
      invalid("bitstream ends inside an Exp-Golomb code");
    }
  }
  int32_t se() {
    try {
      return bs_.read_svlc();
    } catch (const std::out_of_range&) {
      invalid("bitstream ends inside an Exp-Golomb code");
    }
  }
  void byte_align() {
    try {
      bs_.byte_align();
    } catch (const std::out_of_range&) {
      invalid("bitstream ends inside the byte alignment");
    }
  }
  Bitstream& bs() { return bs_; }

 private:
  Bitstream& bs_;
};

// The reference takes ceil / floor of fast_math::log2_raw (reader.rs:1597, 1632, 1671, 1905).  That
// approximation is exact at powers of two and off by < 0.01 elsewhere, so for the integer arguments that
// occur it equals the exact integer logarithms below.
uint32_t ceil_log2(uint32_t x) {
  uint32_t n = 0;
  while ((1ull << n) < x) ++n;
  return n;
}
uint32_t floor_log2(uint32_t x) {
  uint32_t n = 0;
  while ((2ull << n) <= x) ++n;
  return n;
}

// ---- 8.3.4.2 profile, tier, level (reader.rs:524-583) --------------------------------------------
ProfileTierLevel parse_ptl(Reader& r) {
  ProfileTierLevel p;
  p.tier_flag = r.flag();
  p.profile_codec_group_idc = (uint8_t)r.u(7);
  p.profile_toolset_idc = (uint8_t)r.u(8);
  p.profile_reconstruction_idc = (uint8_t)r.u(8);
  r.u(32);                                                   // reserved: four move_to_next_byte()
  p.level_idc = (uint8_t)r.u(8);
  if (r.u(6) != 0) unsupported("ptl sub-profiles");
  r.u(1);                                                    // extended_sub_profile_flag
  if (r.u(1) != 0) unsupported("ptl toolset constraints information");
  return p;
}

// ---- 8.3.4.5 attribute information (reader.rs:444-491) -------------------------------------------
AttributeInformation parse_ai(Reader& r, bool aux_present, uint8_t map_count_minus1) {
  AttributeInformation ai;
  const uint32_t count = r.u(7);
  ai.attributes.resize(count);
  for (auto& a : ai.attributes) {
    a.type_id = (uint8_t)r.u(4);
    a.codec_id = (uint8_t)r.u(8);
    if (aux_present) a.auxiliary_codec_id = (uint8_t)r.u(8);
    a.map_absolute_coding_persistence_flag = true;
    if (map_count_minus1 > 0) a.map_absolute_coding_persistence_flag = r.flag();
    a.dimension_minus1 = (uint8_t)r.u(6);
    if (a.dimension_minus1 > 0) {
      a.dimension_partitions_minus1 = (uint8_t)r.u(6);
      uint32_t remaining = a.dimension_minus1;
      const uint32_t k = a.dimension_partitions_minus1;
      for (uint32_t j = 0; j < k; ++j) {
        const uint32_t channels = (k - j == remaining) ? 0u : (r.ue() & 0xFFu);
        if (channels > remaining) invalid("attribute partition channels exceed the dimension");
        a.partition_channels_minus1.push_back((uint8_t)channels);
        remaining -= channels;
      }
      a.partition_channels_minus1.push_back((uint8_t)remaining);
    }
    a.bitdepth_2d_minus1 = (uint8_t)r.u(5);
    a.msb_align_flag = r.flag();
  }
  return ai;
}

// ---- 8.3.4.1 V3C parameter set (reader.rs:257-337) -----------------------------------------------
V3CParameterSet parse_vps(Reader& r) {
  V3CParameterSet s;
  s.ptl = parse_ptl(r);
  s.v3c_parameter_set_id = (uint8_t)r.u(4);
  r.u(8);
  s.atlas_count_minus1 = (uint8_t)r.u(6);
  if (s.atlas_count_minus1 != 0) unsupported("more than one atlas");
  s.atlas_id = (uint8_t)r.u(6);
  s.frame_width = (uint16_t)r.ue();
  s.frame_height = (uint16_t)r.ue();
  s.map_count_minus1 = (uint8_t)r.u(4);
  s.map_absolute_coding_enable_flag.assign(s.map_count_minus1 + 1u, true);
  s.map_predictor_index_diff.assign(s.map_count_minus1 + 1u, false);
  if (s.map_count_minus1 > 0) {
    s.multiple_map_streams_present_flag = r.flag();
    if (s.multiple_map_streams_present_flag) unsupported("multiple map streams");
  }
  for (uint32_t k = 1; k <= s.map_count_minus1; ++k) {
    if (s.multiple_map_streams_present_flag) s.map_absolute_coding_enable_flag[k] = r.flag();
    if (!s.map_absolute_coding_enable_flag[k]) s.map_predictor_index_diff[k] = r.ue() != 0;
  }
  s.auxiliary_video_present_flag = r.flag();
  s.occupancy_video_present_flag = r.flag();
  s.geometry_video_present_flag = r.flag();
  s.attribute_video_present_flag = r.flag();
  if (s.occupancy_video_present_flag) {                      // 8.3.4.3, reader.rs:364-372
    s.oi.codec_id = (uint8_t)r.u(8);
    s.oi.lossy_compression_threshold = (uint8_t)r.u(8);
    s.oi.bitdepth_2d_minus1 = (uint8_t)r.u(5);
    s.oi.msb_align_flag = r.flag();
  }
  if (s.geometry_video_present_flag) {                       // 8.3.4.4, reader.rs:397-410
    s.gi.codec_id = (uint8_t)r.u(8);
    s.gi.bitdepth_2d_minus1 = (uint8_t)r.u(5);
    s.gi.msb_align_flag = r.flag();
    s.gi.bitdepth_3d_coordinates_minus1 = (uint8_t)r.u(5);
    if (s.auxiliary_video_present_flag) s.gi.auxiliary_codec_id = (uint8_t)r.u(8);
  }
  if (s.attribute_video_present_flag) s.ai = parse_ai(r, s.auxiliary_video_present_flag, s.map_count_minus1);
  if (r.flag()) unsupported("vps extension");
  r.byte_align();
  return s;
}

// ---- 8.3.6.12 reference list structure (reader.rs:1129-1170) -------------------------------------
RefListStruct parse_ref_list(Reader& r, bool long_term, uint32_t log2_max_afoc) {
  RefListStruct l;
  l.num_ref_entries = (uint8_t)r.ue();
  for (uint32_t i = 0; i < l.num_ref_entries; ++i) {
    const bool st = long_term ? r.flag() : true;
    l.st_ref_atlas_frame_flag.push_back(st);
    if (st) {
      const uint8_t d = (uint8_t)r.ue();
      l.abs_delta_afoc_st.push_back(d);
      l.strpf_entry_sign_flag.push_back(d > 0 ? r.flag() : true);
    } else {
      l.afoc_lsb_lt.push_back((uint8_t)r.u(log2_max_afoc));
    }
  }
  return l;
}

// ---- 8.3.6.1 atlas sequence parameter set (reader.rs:1022-1113) ----------------------------------
AtlasSequenceParameterSet parse_asps(Reader& r) {
  AtlasSequenceParameterSet a;
  a.id = (uint8_t)r.ue();
  a.frame_width = r.ue();
  a.frame_height = r.ue();
  a.geometry_3d_bitdepth_minus1 = (uint8_t)r.u(5);
  a.geometry_2d_bitdepth_minus1 = (uint8_t)r.u(5);
  a.log2_max_atlas_frame_order_cnt_lsb_minus_4 = (uint8_t)r.ue();
  a.max_dec_atlas_frame_buffering_minus1 = (uint8_t)r.ue();
  a.long_term_ref_atlas_frames_flag = r.flag();
  a.num_ref_atlas_frame_lists_in_asps = (uint8_t)r.ue();
  for (uint32_t i = 0; i < a.num_ref_atlas_frame_lists_in_asps; ++i)
    a.ref_list_struct.push_back(
        parse_ref_list(r, a.long_term_ref_atlas_frames_flag, a.log2_max_atlas_frame_order_cnt_lsb_minus_4 + 4u));
  a.use_eight_orientations_flag = r.flag();
  a.extended_projection_enabled_flag = r.flag();
  if (a.extended_projection_enabled_flag) unsupported("extended projection");
  a.normal_axis_limits_quantization_enabled_flag = r.flag();
  a.normal_axis_max_delta_value_enabled_flag = r.flag();
  a.patch_precedence_order_flag = r.flag();
  a.log2_patch_packing_block_size = (uint8_t)r.u(3);
  a.patch_size_quantizer_present_flag = r.flag();
  if (a.patch_size_quantizer_present_flag) unsupported("patch size quantizer");
  a.map_count_minus1 = (uint8_t)r.u(4);
  a.pixel_deinterleaving_flag = r.flag();
  if (a.pixel_deinterleaving_flag) unsupported("pixel de-interleaving");
  a.raw_patch_enabled_flag = r.flag();
  a.eom_patch_enabled_flag = r.flag();
  if (a.raw_patch_enabled_flag) unsupported("RAW patches");
  if (a.eom_patch_enabled_flag) unsupported("EOM patches");
  a.plr_enabled_flag = r.flag();
  if (a.plr_enabled_flag) unsupported("point local reconstruction");
  a.vui_parameters_present_flag = r.flag();
  if (a.vui_parameters_present_flag) unsupported("VUI parameters");
  a.extension_flag = r.flag();
  if (a.extension_flag) {
    a.vpcc_extension_flag = r.flag();
    a.extension_7bits = (uint8_t)r.u(7);
  }
  if (a.vpcc_extension_flag) {
    a.remove_duplicate_point_enabled_flag = r.flag();
    // surface thickness is only present with de-interleaving or PLR, both rejected above
  }
  if (a.extension_7bits > 0) unsupported("asps extension bits");
  r.byte_align();
  return a;
}

// ---- 8.3.6.2 atlas frame parameter set + tile information (reader.rs:1192-1305) ------------------
AtlasFrameParameterSet parse_afps(Reader& r, const GofSyntax& g) {
  AtlasFrameParameterSet f;
  f.id = (uint8_t)r.ue();
  f.atlas_sequence_parameter_set_id = (uint8_t)r.ue();
  if (f.atlas_sequence_parameter_set_id >= g.asps.size()) invalid("afps refers to a missing asps");
  f.single_tile_in_atlas_frame_flag = r.flag();
  if (!f.single_tile_in_atlas_frame_flag) unsupported("atlas frame partitioning");
  // asps.auxiliary_video_enabled_flag is never set in the supported subset: no auxiliary tile rows
  f.signalled_tile_id_flag = r.flag();
  if (f.signalled_tile_id_flag) unsupported("signalled tile ids");
  f.output_flag_present_flag = r.flag();
  f.num_ref_idx_default_active_minus1 = (uint8_t)r.ue();
  f.additional_lt_afoc_lsb_len = (uint8_t)r.ue();
  f.lod_mode_enable_flag = r.flag();
  f.raw_3d_offset_bitcount_explicit_mode_flag = r.flag();
  f.extension_flag = r.flag();
  if (f.extension_flag) f.extension_8bits = (uint8_t)r.u(8);
  if (f.extension_8bits > 0) unsupported("afps extension bits");
  r.byte_align();
  return f;
}

// ---- SEI: general syntax + H.20.2.19 geometry smoothing (reader.rs:1371-1505) --------------------
SeiGeometrySmoothing parse_prefix_sei(Reader& r) {
  uint32_t payload_type = 0;
  for (;;) {
    const uint32_t b = r.u(8);
    payload_type += b;
    if (payload_type > 255) invalid("sei payload type overflows u8");          // u8 addition in the reference
    if (b != 0xFF) break;
  }
  for (;;)
    if (r.u(8) != 0xFF) break;                                                   // payload size, unused
  if (payload_type != 66) unsupported("SEI payload type " + std::to_string(payload_type) + " (only geometry smoothing)");
  SeiGeometrySmoothing s;
  s.persistence_flag = r.flag();
  s.reset_flag = r.flag();
  s.instances_updated = (uint8_t)r.u(8);
  const size_t n = s.instances_updated;
  s.instance_index.assign(n, 0); s.method_type.assign(n, 0); s.grid_size_minus_2.assign(n, 0); s.threshold.assign(n, 0);
  s.instance_cancel_flag.assign(n, false); s.filter_eom_points_flag.assign(n, false);
  for (size_t i = 0; i < n; ++i) {
    s.instance_index[i] = (uint8_t)r.u(8);
    const size_t k = s.instance_index[i];
    if (k >= n) invalid("geometry smoothing instance index out of range");      // index panic in the reference
    s.instance_cancel_flag[k] = r.flag();
    if (s.instance_cancel_flag[k]) continue;
    s.method_type[k] = (uint8_t)r.ue();
    if (s.method_type[k] == 1) {
      s.filter_eom_points_flag[k] = r.flag();
      s.grid_size_minus_2[k] = (uint8_t)r.u(7);
      s.threshold[k] = (uint8_t)r.u(8);
    }
  }
  r.byte_align();
  r.u(8);                               // the reference's stand-in for rbsp trailing bits (reader.rs:1414-1417)
  return s;
}

const RefListStruct& active_ref_list(const AtlasSequenceParameterSet& asps, const AtlasTileHeader& h) {
  if (!h.ref_atlas_frame_list_sps_flag) return h.ref_list_struct;
  if (h.ref_atlas_frame_list_idx >= asps.ref_list_struct.size()) invalid("reference list index out of range");
  return asps.ref_list_struct[h.ref_atlas_frame_list_idx];
}

// get_num_ref_idx_active (src/common/context.rs:234-259)
uint32_t num_ref_idx_active(const AtlasSequenceParameterSet& asps, const AtlasFrameParameterSet& afps,
                            const AtlasTileHeader& h) {
  if (h.tile_type == kTileI) return 0;
  if (h.num_ref_idx_active_override_flag) return h.num_ref_idx_active_minus1 + 1u;
  return std::min<uint32_t>(active_ref_list(asps, h).num_ref_entries, afps.num_ref_idx_default_active_minus1 + 1u);
}

// ---- 8.3.6.11 atlas tile header (reader.rs:1571-1690) --------------------------------------------
AtlasTileHeader parse_ath(Reader& r, const GofSyntax& g, uint8_t nal_type) {
  AtlasTileHeader h;
  if (nal_type >= 16 && nal_type <= 27) h.no_output_of_prior_atlas_frames_flag = r.flag();
  if (nal_type == 1) h.tile_nalu_type_info = 1;
  if (nal_type == 0) h.tile_nalu_type_info = 2;
  h.atlas_frame_parameter_set_id = (uint8_t)r.ue();
  h.atlas_adaptation_parameter_set_id = (uint8_t)r.ue();
  if (h.atlas_frame_parameter_set_id >= g.afps.size()) invalid("tile header refers to a missing afps");
  const AtlasFrameParameterSet& afps = g.afps[h.atlas_frame_parameter_set_id];
  const AtlasSequenceParameterSet& asps = g.asps[afps.atlas_sequence_parameter_set_id];
  h.id = 0;                                                  // single tile, no signalled ids
  const uint32_t tt = r.ue() & 0xFFu;
  h.tile_type = tt <= 2 ? (uint8_t)tt : (uint8_t)kTileP;    // TileType::from: unknown values fall to the default, P
  h.atlas_output_flag = afps.output_flag_present_flag ? r.flag() : false;
  h.atlas_frame_order_count_lsb = r.u(asps.log2_max_atlas_frame_order_cnt_lsb_minus_4 + 4u);
  h.ref_atlas_frame_list_sps_flag = asps.num_ref_atlas_frame_lists_in_asps > 0 ? r.flag() : false;
  h.ref_atlas_frame_list_idx = 0;
  if (!h.ref_atlas_frame_list_sps_flag)
    h.ref_list_struct =
        parse_ref_list(r, asps.long_term_ref_atlas_frames_flag, asps.log2_max_atlas_frame_order_cnt_lsb_minus_4 + 4u);
  else
    h.ref_list_struct = asps.ref_list_struct[0];
  if (asps.num_ref_atlas_frame_lists_in_asps > 1)
    h.ref_atlas_frame_list_idx = (uint8_t)r.u(ceil_log2(asps.num_ref_atlas_frame_lists_in_asps));
  const RefListStruct& list = active_ref_list(asps, h);
  uint32_t long_term_entries = 0;
  for (uint32_t i = 0; i < list.num_ref_entries; ++i)
    if (!list.st_ref_atlas_frame_flag[i]) ++long_term_entries;
  for (uint32_t j = 0; j < long_term_entries; ++j) {
    h.additional_afoc_lsb_present_flag.push_back(r.flag());
    if (h.additional_afoc_lsb_present_flag[j]) h.additional_afoc_lsb_val.push_back((uint8_t)r.u(afps.additional_lt_afoc_lsb_len));
  }
  if (h.tile_type != kTileSkip) {
    if (asps.normal_axis_limits_quantization_enabled_flag) {
      h.pos_min_d_quantizer = (uint8_t)r.u(5);
      // the reference reads the max-delta quantizer under the SAME flag (reader.rs:1656-1661), not under
      // asps_normal_axis_max_delta_value_enabled_flag as ISO/IEC 23090-5 does
      h.pos_delta_max_d_quantizer = (uint8_t)r.u(5);
    }
    if (afps.raw_3d_offset_bitcount_explicit_mode_flag)
      h.raw_3d_offset_axis_bitcount_minus1 = (uint8_t)r.u(floor_log2(asps.geometry_3d_bitdepth_minus1 + 1u));
    else   // u8 arithmetic of the reference, wrapping as in its release build; the value is never used
      h.raw_3d_offset_axis_bitcount_minus1 =
          (uint8_t)((uint8_t)(asps.geometry_3d_bitdepth_minus1 - asps.geometry_2d_bitdepth_minus1) - 1u);
    if (h.tile_type == kTileP && list.num_ref_entries > 1) {
      h.num_ref_idx_active_override_flag = r.flag();
      if (h.num_ref_idx_active_override_flag) h.num_ref_idx_active_minus1 = (uint8_t)r.ue();
    }
  }
  r.byte_align();
  return h;
}

// ---- 8.3.7.3 patch data unit (reader.rs:1873-1925) -----------------------------------------------
vpcc_intra_pdu parse_intra_pdu(Reader& r, const AtlasSequenceParameterSet& asps, const AtlasFrameParameterSet& afps,
                               const AtlasTileHeader& h) {
  vpcc_intra_pdu p{};
  const uint32_t bitcount_uv = asps.geometry_3d_bitdepth_minus1 + 1u;
  if (h.pos_min_d_quantizer > asps.geometry_3d_bitdepth_minus1 + 1u) invalid("pos_min_d_quantizer exceeds the 3-D bit depth");
  const uint32_t bitcount_d = asps.geometry_3d_bitdepth_minus1 - h.pos_min_d_quantizer + 1u;
  p.pos_2d_x = r.ue(); p.pos_2d_y = r.ue();
  p.size_2d_x_minus1 = r.ue(); p.size_2d_y_minus1 = r.ue();
  p.pos_3d_offset_u = r.u(bitcount_uv); p.pos_3d_offset_v = r.u(bitcount_uv);
  p.pos_3d_offset_d = r.u(bitcount_d);
  if (asps.normal_axis_max_delta_value_enabled_flag) {
    const uint32_t m = std::min<uint32_t>(asps.geometry_2d_bitdepth_minus1, asps.geometry_3d_bitdepth_minus1) + 1u;
    if (h.pos_delta_max_d_quantizer > m) invalid("pos_delta_max_d_quantizer exceeds the bit depth");
    p.pos_3d_range_d = r.u(m - h.pos_delta_max_d_quantizer);
  }
  p.projection_id = r.u(ceil_log2(asps.max_number_projections_minus1 + 1u));
  if (p.projection_id > 5) invalid("projection id above 5");
  p.orientation_index = r.u(asps.use_eight_orientations_flag ? 3 : 1);
  if (afps.lod_mode_enable_flag) unsupported("level-of-detail patches");
  if (asps.plr_enabled_flag) unsupported("point local reconstruction");
  return p;
}

// ---- 8.3.7.1 / 8.3.7.2 atlas tile data unit, patch information data (reader.rs:1712-1857) ---------
std::vector<PatchInformationData> parse_atdu(Reader& r, const GofSyntax& g, const AtlasTileHeader& h) {
  std::vector<PatchInformationData> out;
  if (h.tile_type == kTileSkip) return out;
  const AtlasFrameParameterSet& afps = g.afps[h.atlas_frame_parameter_set_id];
  const AtlasSequenceParameterSet& asps = g.asps[afps.atlas_sequence_parameter_set_id];
  for (;;) {
    PatchInformationData d;
    d.patch_mode = (uint8_t)r.ue();
    if (d.patch_mode == 14) break;                           // I_END / P_END
    if (h.tile_type == kTileI) {
      d.kind = kPatchIntra;                                  // PatchModeITile::from: unknown modes fall to Intra
      d.intra = parse_intra_pdu(r, asps, afps, h);
    } else if (d.patch_mode == 3) {
      d.kind = kPatchIntra;
      d.intra = parse_intra_pdu(r, asps, afps, h);
    } else if (d.patch_mode == 2) {                          // inter patch data unit (reader.rs:1942-1973)
      d.kind = kPatchInter;
      if (num_ref_idx_active(asps, afps, h) > 1) d.ref_index = r.ue();
      d.ref_patch_index = r.se();
      d.pos_2d_x = r.se(); d.pos_2d_y = r.se();
      d.delta_2d_size_x = r.se(); d.delta_2d_size_y = r.se();
      d.pos_3d_offset_u = r.se(); d.pos_3d_offset_v = r.se();
      d.pos_3d_offset_d = r.se();
      if (asps.normal_axis_max_delta_value_enabled_flag) unsupported("inter patch with max delta value");
    } else if (d.patch_mode == 1) {                          // merge patch data unit (reader.rs:1996-2036)
      d.kind = kPatchMerge;
      if (num_ref_idx_active(asps, afps, h) > 1) d.ref_index = r.ue();
      d.override_2d_params_flag = r.flag();
      if (d.override_2d_params_flag) {
        d.pos_2d_x = r.se(); d.pos_2d_y = r.se();
        d.delta_2d_size_x = r.se(); d.delta_2d_size_y = r.se();
      } else {
        d.override_3d_params_flag = r.flag();
        d.pos_3d_offset_u = r.se(); d.pos_3d_offset_v = r.se();     // read whatever the flag says, like the reference
        d.pos_3d_offset_d = r.se();
        if (asps.normal_axis_max_delta_value_enabled_flag) unsupported("merge patch with max delta value");
      }
    } else {
      d.kind = kPatchSkip;                                   // PatchModePTile::from: 0 and unknown modes
    }
    out.push_back(d);
  }
  return out;
}

// ---- sample-stream NAL units of one atlas sub-bitstream (reader.rs:721-816) -----------------------
void parse_atlas_data(Reader& r, GofSyntax& g) {
  const uint32_t precision = r.u(3) + 1u;
  r.u(5);
  int prefix_sei = -1;
  while (r.bs().more_data()) {
    const uint32_t size = r.u(8 * precision);                // > 4 bytes: "bits > 32" panic in the reference
    const size_t start = r.bs().position_bytes();
    if (size < 2 || start + size > r.bs().data().size()) invalid("NAL unit size runs past the atlas sub-bitstream");
    r.u(1);
    const uint8_t type = (uint8_t)r.u(6);
    r.u(6);                                                  // layer id
    r.u(3);                                                  // temporal id + 1
    if (type == 36) {
      g.asps.push_back(parse_asps(r));
    } else if (type == 37) {
      g.afps.push_back(parse_afps(r, g));
    } else if (type <= 11 || type == 23) {                   // TRAIL_N .. SKIP_R, IDR_N_LP
      AtlasTileLayer l;
      l.nal_unit_type = type;
      l.header = parse_ath(r, g, type);
      l.patches = parse_atdu(r, g, l.header);
      r.byte_align();
      l.prefix_sei = prefix_sei;
      g.atls.push_back(std::move(l));
    } else if (type == 43 || type == 45) {                   // prefix SEI
      g.seis.push_back(parse_prefix_sei(r));
      prefix_sei = (int)g.seis.size() - 1;
    } else if (type == 44 || type == 46) {
      unsupported("suffix SEI");
    } else {
      unsupported("NAL unit type " + std::to_string(type));  // unreachable!() in the reference
    }
    const size_t end = r.bs().position_bytes() + (r.bs().position_bits() ? 1 : 0);
    if (end > start + size) invalid("NAL unit payload longer than its size field");
    // The reference trusts the payload to end where the size field says; a shorter payload would
    // desynchronise it.  Re-synchronise on the size field instead.
    r.bs().seek(start + size);
  }
}

void parse_unit(const V3CUnit& unit, GofSyntax& g) {
  Bitstream bs(unit.payload);
  Reader r(bs);
  // ---- V3C unit header, 4 bytes (reader.rs:35-79)
  const uint32_t type = r.u(5);
  if (type > kAttributeVideo) unsupported("V3C unit type " + std::to_string(type));
  V3CUnitHeader h = g.unit_header[type];
  if (type != kVps) {
    h.sequence_parameter_set_id = (uint8_t)r.u(4);
    h.atlas_id = (uint8_t)r.u(6);
    if (h.atlas_id != 0) unsupported("more than one atlas");
  }
  if (type == kAttributeVideo) {
    h.attribute_index = (uint8_t)r.u(7);
    h.attribute_dimension_index = (uint8_t)r.u(5);
    h.map_index = (uint8_t)r.u(4);
    h.auxiliary_video_flag = r.flag();
  } else if (type == kGeometryVideo) {
    h.map_index = (uint8_t)r.u(4);
    h.auxiliary_video_flag = r.flag();
    r.u(12);
  } else if (type == kOccupancyVideo || type == kAtlasData) {
    r.u(17);
  } else {
    r.u(27);
  }
  if (h.auxiliary_video_flag) unsupported("auxiliary video");
  g.unit_header[type] = h;
  // ---- payload (reader.rs:82-158)
  auto video = [&]() {
    VideoSubstream v;
    v.unit_type = (uint8_t)type;
    v.data.assign(unit.payload.begin() + 4, unit.payload.end());     // unit size - 4 bytes
    g.videos.push_back(std::move(v));
  };
  switch (type) {
    case kVps:
      if (g.has_vps) invalid("second V3C parameter set inside one GOF");
      g.vps = parse_vps(r);
      g.has_vps = true;
      break;
    case kAtlasData:
      parse_atlas_data(r, g);
      break;
    case kOccupancyVideo:
      video();
      break;
    case kGeometryVideo:
      if (!g.has_vps) invalid("geometry video before the V3C parameter set");
      video();
      break;
    case kAttributeVideo:
      if (!g.has_vps) invalid("attribute video before the V3C parameter set");
      if (g.vps.ai.attributes.empty()) break;
      if (h.attribute_dimension_index != 0) unsupported("attribute dimension partitions");
      video();
      break;
  }
}

}  // namespace

const VideoSubstream* GofSyntax::video(uint8_t unit_type) const {
  for (const VideoSubstream& v : videos)
    if (v.unit_type == unit_type) return &v;
  return nullptr;
}

size_t parse_gof(const std::vector<V3CUnit>& units, size_t first, GofSyntax* out) {
  *out = GofSyntax{};
  size_t i = first;
  int n_vps = 0;
  for (; i < units.size(); ++i) {
    if (units[i].payload.size() < 4) invalid("V3C unit shorter than its header");
    if ((units[i].payload[0] >> 3) == kVps && ++n_vps > 1) break;      // the next GOF starts here (reader.rs:680-694)
    parse_unit(units[i], *out);
  }
  return i;
}

std::vector<PatchFrame> build_patch_frames(const GofSyntax& g) {
  std::vector<PatchFrame> frames;
  uint32_t prev_lsb = 0, prev_msb = 0, prev_val = 0;
  for (size_t i = 0; i < g.atls.size(); ++i) {
    const AtlasTileLayer& l = g.atls[i];
    const AtlasTileHeader& h = l.header;
    const AtlasFrameParameterSet& afps = g.afps[h.atlas_frame_parameter_set_id];
    const AtlasSequenceParameterSet& asps = g.asps[afps.atlas_sequence_parameter_set_id];
    // derive_afoc_val, src/common/context.rs:142-172 (u32, wrapping like the release build)
    uint32_t msb = 0;
    const uint32_t lsb = h.atlas_frame_order_count_lsb;
    if (i > 0) {
      const uint32_t max_lsb = 1u << (asps.log2_max_atlas_frame_order_cnt_lsb_minus_4 + 4u);
      if (lsb < prev_lsb && prev_lsb - lsb >= max_lsb / 2) msb = prev_msb + max_lsb;
      else if (lsb > prev_lsb && lsb - prev_lsb > max_lsb / 2) msb = prev_msb - max_lsb;
      else msb = prev_msb;
    }
    const uint32_t val = i == 0 ? lsb : msb + lsb;
    if (i > 0 && val == prev_val) unsupported("two tile layers of one atlas frame");   // unreachable!() decoder.rs:357
    PatchFrame f;
    f.atlas_frame_order_count_msb = msb;
    f.atlas_frame_order_count_val = val;
    f.frame_index = val & 0xFFu;                                                        // `as u8`
    f.width = asps.frame_width;
    f.height = asps.frame_height;
    if (f.frame_index > 0 && h.tile_type != kTileI) unsupported("only I tiles are supported");   // decoder.rs:403-407
    vpcc_patch_frame_params fp{};
    fp.log2_patch_packing_block_size = asps.log2_patch_packing_block_size;
    fp.geometry_3d_bitdepth = asps.geometry_3d_bitdepth_minus1 + 1u;
    fp.pos_min_d_quantizer = h.pos_min_d_quantizer;
    fp.patch_size_quantizer_present_flag = asps.patch_size_quantizer_present_flag;
    fp.patch_size_info_quantizer_x = h.patch_size_info_quantizer_x;
    fp.patch_size_info_quantizer_y = h.patch_size_info_quantizer_y;
    fp.plr_enabled_flag = asps.plr_enabled_flag;
    for (const PatchInformationData& d : l.patches) {
      // PatchType::from_tile_type_and_patch_mode, decoder.rs:676-691
      if (h.tile_type == kTileSkip) unsupported("skip patch");
      switch (d.kind) {
        case kPatchIntra: {
          vpcc_patch p;
          const int st = vpcc_patch_from_intra_pdu(&fp, &d.intra, &p);
          if (st != VPCC_OK) throw SyntaxError(st, "intra patch data unit not representable");
          // decoder.rs:474-478 asserts one of the three axis tables, i.e. projection ids 0..5 (checked at parse)
          f.patches.push_back(p);
          break;
        }
        case kPatchInter: unsupported("inter patches");
        case kPatchMerge: unsupported("merge patches");
        case kPatchSkip:
          if (d.patch_mode != 0) unsupported("unknown patch mode in a P tile");
          unsupported("skip patches");
      }
    }
    frames.push_back(std::move(f));
    prev_lsb = lsb; prev_msb = msb; prev_val = val;
  }
  return frames;
}

GofParams build_gof_params(const GofSyntax& g) {
  if (!g.has_vps) invalid("no V3C parameter set");
  if (g.asps.empty()) invalid("no atlas sequence parameter set");
  const AtlasSequenceParameterSet& asps = g.asps[0];                   // decoder.rs:598 uses set 0
  GofParams p;
  p.frame_width = g.vps.frame_width;
  p.frame_height = g.vps.frame_height;
  p.occupancy_resolution = 1u << asps.log2_patch_packing_block_size;
  p.map_count = g.vps.map_count_minus1 + 1u;
  p.absolute_d1 = g.vps.map_count_minus1 == 0 || g.vps.map_absolute_coding_enable_flag[1];
  p.multiple_streams = g.vps.multiple_map_streams_present_flag;
  p.enable_size_quantization = asps.patch_size_quantizer_present_flag;
  p.surface_thickness = asps.surface_thickness_minus1 + 1u;
  p.geometry_bitdepth_3d = g.vps.gi.bitdepth_3d_coordinates_minus1 + 1u;
  if (!g.atls.empty() && g.atls[0].prefix_sei >= 0) {
    const SeiGeometrySmoothing& s = g.seis[(size_t)g.atls[0].prefix_sei];
    p.geometry_smoothing_sei = true;
    for (size_t k = 0; k < s.method_type.size(); ++k)
      if (!s.instance_cancel_flag[k] && s.method_type[k] == 1) {
        p.smoothing_grid_size = s.grid_size_minus_2[k] + 2u;
        p.smoothing_threshold = s.threshold[k];
        break;
      }
  }
  return p;
}

}  // namespace tmc2rs

// ------------------------------------------------------------------ C ABI
struct vpcc_v3c_stream {
  std::vector<tmc2rs::V3CUnit> units;
  size_t next_unit = 0;
  size_t header_size = 0;
  tmc2rs::GofSyntax gof;
  std::vector<tmc2rs::PatchFrame> frames;
  tmc2rs::GofParams params;
  bool have_gof = false;
  std::string error;
};

extern "C" int vpcc_v3c_open(const uint8_t* data, size_t n, vpcc_v3c_stream** out) {
  if (!data || !n || !out) return VPCC_ERR_INVALID_ARG;
  *out = nullptr;
  auto* s = new vpcc_v3c_stream();
  try {
    tmc2rs::Bitstream bs(std::vector<uint8_t>(data, data + n));
    s->units = tmc2rs::split_sample_stream(bs, &s->header_size);
  } catch (const std::exception&) {
    delete s;
    return VPCC_ERR_INVALID_ARG;
  }
  *out = s;
  return VPCC_OK;
}

extern "C" void vpcc_v3c_close(vpcc_v3c_stream* s) { delete s; }

extern "C" const char* vpcc_v3c_error(const vpcc_v3c_stream* s) { return s ? s->error.c_str() : ""; }

extern "C" uint32_t vpcc_v3c_unit_count(const vpcc_v3c_stream* s) { return s ? (uint32_t)s->units.size() : 0; }

extern "C" int vpcc_v3c_next_gof(vpcc_v3c_stream* s, int* have_gof, vpcc_v3c_gof_info* info) {
  if (!s || !have_gof) return VPCC_ERR_INVALID_ARG;
  *have_gof = 0;
  s->have_gof = false;
  s->error.clear();
  if (s->next_unit >= s->units.size()) return VPCC_OK;
  try {
    const size_t next = tmc2rs::parse_gof(s->units, s->next_unit, &s->gof);
    s->next_unit = next;
    s->params = tmc2rs::build_gof_params(s->gof);
    s->frames = tmc2rs::build_patch_frames(s->gof);
  } catch (const tmc2rs::SyntaxError& e) {
    s->error = e.what();
    s->next_unit = s->units.size();           // the reference's worker dies here: no further GOFs
    return e.status;
  }
  s->have_gof = true;
  *have_gof = 1;
  if (info) {
    const tmc2rs::GofSyntax& g = s->gof;
    std::memset(info, 0, sizeof(*info));
    info->frame_count = (uint32_t)s->frames.size();
    info->frame_width = s->params.frame_width;
    info->frame_height = s->params.frame_height;
    info->atlas_frame_width = g.asps[0].frame_width;
    info->atlas_frame_height = g.asps[0].frame_height;
    info->map_count = s->params.map_count;
    info->absolute_d1 = s->params.absolute_d1;
    info->occupancy_resolution = s->params.occupancy_resolution;
    info->geometry_3d_bitdepth = s->params.geometry_bitdepth_3d;
    info->atlas_geometry_3d_bitdepth = g.asps[0].geometry_3d_bitdepth_minus1 + 1u;
    info->geometry_2d_bitdepth = g.vps.gi.bitdepth_2d_minus1 + 1u;
    info->occupancy_2d_bitdepth = g.vps.oi.bitdepth_2d_minus1 + 1u;
    info->attribute_count = (uint32_t)g.vps.ai.attributes.size();
    info->attribute_2d_bitdepth = g.vps.ai.attributes.empty() ? 0u : g.vps.ai.attributes[0].bitdepth_2d_minus1 + 1u;
    info->occupancy_codec_id = g.vps.oi.codec_id;
    info->geometry_codec_id = g.vps.gi.codec_id;
    info->attribute_codec_id = g.vps.ai.attributes.empty() ? 0u : g.vps.ai.attributes[0].codec_id;
    info->profile_codec_group_idc = g.vps.ptl.profile_codec_group_idc;
    info->profile_toolset_idc = g.vps.ptl.profile_toolset_idc;
    info->profile_reconstruction_idc = g.vps.ptl.profile_reconstruction_idc;
    info->level_idc = g.vps.ptl.level_idc;
    info->use_eight_orientations_flag = g.asps[0].use_eight_orientations_flag;
    info->remove_duplicate_point_enabled_flag = g.asps[0].remove_duplicate_point_enabled_flag;
    info->geometry_smoothing_sei = s->params.geometry_smoothing_sei;
    info->smoothing_grid_size = s->params.smoothing_grid_size;
    info->smoothing_threshold = s->params.smoothing_threshold;
    for (int t = 0; t < 3; ++t) {
      const tmc2rs::VideoSubstream* v = g.video((uint8_t)(tmc2rs::kOccupancyVideo + t));
      info->video_bytes[t] = v ? v->data.size() : 0;
    }
  }
  return VPCC_OK;
}

extern "C" int vpcc_v3c_frame_patches(const vpcc_v3c_stream* s, uint32_t frame, vpcc_patch* out, uint32_t capacity,
                                      uint32_t* n_patches, uint32_t* frame_index) {
  if (!s || !n_patches) return VPCC_ERR_INVALID_ARG;
  if (!s->have_gof) return VPCC_ERR_STATE;
  if (frame >= s->frames.size()) return VPCC_ERR_INVALID_ARG;
  const tmc2rs::PatchFrame& f = s->frames[frame];
  *n_patches = (uint32_t)f.patches.size();
  if (frame_index) *frame_index = f.frame_index;
  if (!out) return VPCC_OK;
  if (capacity < f.patches.size()) return VPCC_ERR_CAPACITY;
  std::copy(f.patches.begin(), f.patches.end(), out);
  return VPCC_OK;
}

extern "C" int vpcc_v3c_video(const vpcc_v3c_stream* s, int video, const uint8_t** data, size_t* n) {
  if (!s || !data || !n || video < 0 || video > 2) return VPCC_ERR_INVALID_ARG;
  if (!s->have_gof) return VPCC_ERR_STATE;
  const tmc2rs::VideoSubstream* v = s->gof.video((uint8_t)(tmc2rs::kOccupancyVideo + video));
  *data = v ? v->data.data() : nullptr;
  *n = v ? v->data.size() : 0;
  return VPCC_OK;
}
