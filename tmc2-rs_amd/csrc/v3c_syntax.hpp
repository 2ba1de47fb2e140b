// v3c_syntax.hpp — V3C / V-PCC high-level syntax parser and the per-GOF patch-frame
// builder (row 2), host side, C++.  Own implementation of the BEHAVIOUR of the reference's reader for the
// same supported subset; every parse function cites the reference lines it follows, and features the
// reference rejects with assert!/unimplemented!() raise SyntaxError(VPCC_ERR_UNSUPPORTED) here.
//
//   V3C unit header / payload dispatch          src/bitstream/reader.rs:23-160
//   V3C parameter set (+ PTL, OI, GI, AI)       src/bitstream/reader.rs:257-583
//   GOF = units up to the next VPS              src/bitstream/reader.rs:672-700, src/lib.rs:119-133
//   sample-stream NAL units, NAL header         src/bitstream/reader.rs:720-816
//   ASPS, ref list struct                       src/bitstream/reader.rs:1021-1172
//   AFPS, atlas frame tile information          src/bitstream/reader.rs:1191-1305
//   SEI (geometry smoothing only)               src/bitstream/reader.rs:1370-1505
//   atlas tile layer / header / data unit       src/bitstream/reader.rs:1525-1733
//   patch information data, intra/inter/merge   src/bitstream/reader.rs:1794-2037
//   atlas frame order count                     src/common/context.rs:142-172
//   patch frames + reconstruction parameters    src/decoder.rs:320-517, 590-627
#pragma once

#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "bitstream.hpp"
#include "vpcc_recon.h"

namespace tmc2rs {

struct SyntaxError : std::runtime_error {
  int status;   // VPCC_ERR_UNSUPPORTED (reference: unimplemented!/assert!) or VPCC_ERR_INVALID_ARG (malformed)
  SyntaxError(int st, const std::string& what) : std::runtime_error(what), status(st) {}
};

enum V3CUnitType : uint8_t { kVps = 0, kAtlasData = 1, kOccupancyVideo = 2, kGeometryVideo = 3, kAttributeVideo = 4 };

struct V3CUnitHeader {
  uint8_t sequence_parameter_set_id = 0, atlas_id = 0, attribute_index = 0, attribute_dimension_index = 0, map_index = 0;
  bool auxiliary_video_flag = false;
};

struct ProfileTierLevel {
  bool tier_flag = false;
  uint8_t profile_codec_group_idc = 0, profile_toolset_idc = 0, profile_reconstruction_idc = 0, level_idc = 0;
};

struct OccupancyInformation {
  uint8_t codec_id = 0, lossy_compression_threshold = 0, bitdepth_2d_minus1 = 10;
  bool msb_align_flag = false;
};

struct GeometryInformation {
  uint8_t codec_id = 0, auxiliary_codec_id = 0, bitdepth_2d_minus1 = 10, bitdepth_3d_coordinates_minus1 = 9;
  bool msb_align_flag = false;
};

struct AttributeInformation {
  struct Attribute {
    uint8_t type_id = 0, codec_id = 0, auxiliary_codec_id = 0;
    bool map_absolute_coding_persistence_flag = true;
    uint8_t dimension_minus1 = 0, dimension_partitions_minus1 = 0;
    std::vector<uint8_t> partition_channels_minus1;
    uint8_t bitdepth_2d_minus1 = 0;
    bool msb_align_flag = false;
  };
  std::vector<Attribute> attributes;
};

struct V3CParameterSet {
  ProfileTierLevel ptl;
  uint8_t v3c_parameter_set_id = 0, atlas_count_minus1 = 0, atlas_id = 0;
  uint16_t frame_width = 0, frame_height = 0;       // `as u16` like the reference
  uint8_t map_count_minus1 = 0;
  bool multiple_map_streams_present_flag = false;
  std::vector<bool> map_absolute_coding_enable_flag, map_predictor_index_diff;
  bool auxiliary_video_present_flag = false, occupancy_video_present_flag = false, geometry_video_present_flag = false,
       attribute_video_present_flag = false;
  OccupancyInformation oi;
  GeometryInformation gi;
  AttributeInformation ai;
};

struct RefListStruct {
  uint8_t num_ref_entries = 0;
  std::vector<bool> st_ref_atlas_frame_flag, strpf_entry_sign_flag;
  std::vector<uint8_t> abs_delta_afoc_st, afoc_lsb_lt;
};

struct AtlasSequenceParameterSet {
  uint8_t id = 0;
  uint32_t frame_width = 0, frame_height = 0;
  uint8_t geometry_2d_bitdepth_minus1 = 0, geometry_3d_bitdepth_minus1 = 0;
  uint8_t log2_max_atlas_frame_order_cnt_lsb_minus_4 = 4, max_dec_atlas_frame_buffering_minus1 = 0;
  bool long_term_ref_atlas_frames_flag = false;
  uint8_t num_ref_atlas_frame_lists_in_asps = 0;
  std::vector<RefListStruct> ref_list_struct;
  bool use_eight_orientations_flag = false, extended_projection_enabled_flag = false;
  uint32_t max_number_projections_minus1 = 5;
  bool normal_axis_limits_quantization_enabled_flag = true, normal_axis_max_delta_value_enabled_flag = false,
       patch_precedence_order_flag = false;
  uint8_t log2_patch_packing_block_size = 0;
  bool patch_size_quantizer_present_flag = false;
  uint8_t map_count_minus1 = 0;
  bool pixel_deinterleaving_flag = false, eom_patch_enabled_flag = false, raw_patch_enabled_flag = false,
       auxiliary_video_enabled_flag = false, plr_enabled_flag = false, vui_parameters_present_flag = false,
       extension_flag = false, vpcc_extension_flag = false;
  uint8_t eom_fix_bit_count_minus1 = 0, extension_7bits = 0;
  bool remove_duplicate_point_enabled_flag = false;    // vpcc extension
  uint8_t surface_thickness_minus1 = 0;
};

struct AtlasFrameParameterSet {
  uint8_t id = 0, atlas_sequence_parameter_set_id = 0;
  bool single_tile_in_atlas_frame_flag = true, signalled_tile_id_flag = false;
  uint32_t num_tiles_in_atlas_frame_minus1 = 0;
  uint8_t signalled_tile_id_length_minus1 = 0;
  bool output_flag_present_flag = false;
  uint8_t num_ref_idx_default_active_minus1 = 0, additional_lt_afoc_lsb_len = 0;
  bool lod_mode_enable_flag = false, raw_3d_offset_bitcount_explicit_mode_flag = false, extension_flag = false;
  uint8_t extension_8bits = 0;
};

struct SeiGeometrySmoothing {            // H.20.2.19; instance k of the arrays is addressed by instance_index
  bool persistence_flag = false, reset_flag = false;
  uint8_t instances_updated = 0;
  std::vector<uint8_t> instance_index, method_type, grid_size_minus_2, threshold;
  std::vector<bool> instance_cancel_flag, filter_eom_points_flag;
};

enum TileType : uint8_t { kTileP = 0, kTileI = 1, kTileSkip = 2 };

struct AtlasTileHeader {
  bool no_output_of_prior_atlas_frames_flag = false;
  uint8_t atlas_frame_parameter_set_id = 0, atlas_adaptation_parameter_set_id = 0;
  uint32_t id = 0;
  uint8_t tile_type = kTileP;
  bool atlas_output_flag = false;
  uint32_t atlas_frame_order_count_lsb = 0;
  bool ref_atlas_frame_list_sps_flag = false;
  uint8_t ref_atlas_frame_list_idx = 0;
  std::vector<bool> additional_afoc_lsb_present_flag;
  std::vector<uint8_t> additional_afoc_lsb_val;
  uint8_t pos_min_d_quantizer = 0, pos_delta_max_d_quantizer = 0;
  uint8_t patch_size_info_quantizer_x = 0, patch_size_info_quantizer_y = 0;
  uint8_t raw_3d_offset_axis_bitcount_minus1 = 0;
  bool num_ref_idx_active_override_flag = false;
  uint8_t num_ref_idx_active_minus1 = 0;
  RefListStruct ref_list_struct;
  uint8_t tile_nalu_type_info = 0;
};

enum PatchKind : uint8_t { kPatchIntra, kPatchInter, kPatchMerge, kPatchSkip };

struct PatchInformationData {
  uint8_t patch_mode = 0;
  PatchKind kind = kPatchIntra;
  vpcc_intra_pdu intra{};                                         // kPatchIntra
  uint32_t ref_index = 0;                                         // inter / merge
  int32_t ref_patch_index = 0;
  int32_t pos_2d_x = 0, pos_2d_y = 0, delta_2d_size_x = 0, delta_2d_size_y = 0;
  int32_t pos_3d_offset_u = 0, pos_3d_offset_v = 0, pos_3d_offset_d = 0;
  bool override_2d_params_flag = false, override_3d_params_flag = false;
};

struct AtlasTileLayer {
  uint8_t nal_unit_type = 0;
  AtlasTileHeader header;
  std::vector<PatchInformationData> patches;
  int prefix_sei = -1;                      // index into GofSyntax::seis of the prefix SEI in force, or -1
};

struct VideoSubstream {
  uint8_t unit_type = 0;                    // kOccupancyVideo / kGeometryVideo / kAttributeVideo
  std::vector<uint8_t> data;                // sample-stream framed video sub-bitstream (unit size - 4 bytes)
};

// Everything one GOF's V3C units carry: the reference's `Context` after SampleStreamV3CUnit::decode.
struct GofSyntax {
  bool has_vps = false;
  V3CParameterSet vps;
  V3CUnitHeader unit_header[5];
  std::vector<AtlasSequenceParameterSet> asps;      // looked up by position == id, like the reference
  std::vector<AtlasFrameParameterSet> afps;
  std::vector<SeiGeometrySmoothing> seis;
  std::vector<AtlasTileLayer> atls;
  std::vector<VideoSubstream> videos;
  const VideoSubstream* video(uint8_t unit_type) const;
};

// Parses the units [first, ...) of a split sample stream up to (not including) the second VPS.
// Returns the index of the first unit of the next GOF.
size_t parse_gof(const std::vector<V3CUnit>& units, size_t first, GofSyntax* out);

// One atlas frame of the GOF, ready for reconstruction (create_patch_frame, src/decoder.rs:320-506).
struct PatchFrame {
  uint32_t frame_index = 0;                 // afoc_val as u8
  uint32_t atlas_frame_order_count_val = 0, atlas_frame_order_count_msb = 0;
  uint32_t width = 0, height = 0;           // asps frame size
  std::vector<vpcc_patch> patches;
};

// new_generate_point_cloud_params (src/decoder.rs:590-627) + the per-frame constants of Decoder::decode.
struct GofParams {
  uint32_t frame_width = 0, frame_height = 0;        // vps
  uint32_t occupancy_resolution = 0;                 // 1 << asps[0].log2_patch_packing_block_size
  uint32_t map_count = 1;
  bool absolute_d1 = true, multiple_streams = false, enable_size_quantization = false;
  uint32_t surface_thickness = 1, geometry_bitdepth_3d = 10;
  bool geometry_smoothing_sei = false;               // a prefix geometry-smoothing SEI is attached to ATL 0
  uint32_t smoothing_grid_size = 0, smoothing_threshold = 0;
};

std::vector<PatchFrame> build_patch_frames(const GofSyntax& g);
GofParams build_gof_params(const GofSyntax& g);

}  // namespace tmc2rs
