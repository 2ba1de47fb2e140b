// decoder_input.cpp — the two input readers of the C++ Decoder (pure host code, no GPU calls): the decoded-GOF
// container (.vpccgof) and a V3C sample stream with externally decoded raw videos.  Both take untrusted bytes;
// tests/fuzz_container.cpp mutates containers under AddressSanitizer + UBSan.
#include "decoder.hpp"
#include "v3c_syntax.hpp"

#include <cstring>

namespace tmc2rs {

// ------------------------------------------------------------------ container (.vpccgof)
//   header : "VPCCGOF1" | u32 version (1) | u32 gof_count
//   gof    : u32 frame_count | frame*
//   frame  : 16 x u32 { width, height, occupancy_resolution, occupancy_precision, map_count, absolute_d1,
//                       attribute_count, flags, occ_w, occ_h, geo_w, geo_h, attr_w, attr_h, patch_count, 0 }
//            vpcc_patch[patch_count] | occupancy u8[occ_w*occ_h] | geometry Y u16[geo_w*geo_h] x map_count
//            | (Y u16[attr_w*attr_h], U, V u16[(attr_w/2)*(attr_h/2)]) x map_count   (if attribute_count; even dims)
//   every section is padded to a multiple of 8 bytes.
namespace {
struct Cursor {
  const unsigned char* p;
  size_t left;
  bool take(size_t n, const unsigned char** out) {
    if (n > left) return false;                     // before padding: (n + 7) may wrap
    const size_t padded = (n + 7) & ~size_t(7);
    if (padded > left) return false;
    *out = p;
    p += padded;
    left -= padded;
    return true;
  }
};
}  // namespace

bool parse_container(const std::vector<unsigned char>& buf, std::vector<DecodedGof>* gofs, std::string* err) {
  Cursor c{buf.data(), buf.size()};
  const unsigned char* h;
  if (!c.take(16, &h) || std::memcmp(h, "VPCCGOF1", 8) != 0) { *err = "not a .vpccgof container"; return false; }
  uint32_t version, gof_count;
  std::memcpy(&version, h + 8, 4);
  std::memcpy(&gof_count, h + 12, 4);
  if (version != 1) { *err = "unsupported container version"; return false; }
  gofs->clear();
  for (uint32_t g = 0; g < gof_count; ++g) {
    const unsigned char* q;
    if (!c.take(8, &q)) { *err = "truncated container"; return false; }
    uint32_t frame_count;
    std::memcpy(&frame_count, q, 4);
    DecodedGof gof;
    for (uint32_t f = 0; f < frame_count; ++f) {
      if (!c.take(64, &q)) { *err = "truncated container"; return false; }
      uint32_t w[16];
      std::memcpy(w, q, 64);
      vpcc_frame_desc d{};
      d.width = w[0]; d.height = w[1]; d.occupancy_resolution = w[2]; d.occupancy_precision = w[3];
      d.map_count = w[4]; d.absolute_d1 = w[5]; d.attribute_count = w[6]; d.flags = w[7];
      const uint32_t occ_w = w[8], occ_h = w[9], geo_w = w[10], geo_h = w[11], attr_w = w[12], attr_h = w[13];
      d.patch_count = w[14];
      if (d.map_count < 1 || d.map_count > 2) { *err = "map_count out of range"; return false; }
      // untrusted header fields: bound every dimension before it is multiplied (validate_frame bounds the canvas
      // the same way), and 4:2:0 planes have even dimensions — the chroma index (v/2)*(w/2)+(u/2) of an odd
      // plane would run past a (w/2) x (h/2) section
      const uint32_t kMaxDim = 32768;
      if (occ_w > kMaxDim || occ_h > kMaxDim || geo_w > kMaxDim || geo_h > kMaxDim || attr_w > kMaxDim || attr_h > kMaxDim ||
          d.patch_count > 65535) { *err = "plane dimensions or patch count out of range"; return false; }
      if (d.attribute_count && ((attr_w | attr_h) & 1u)) { *err = "4:2:0 attribute planes need even dimensions"; return false; }
      if (!c.take(sizeof(vpcc_patch) * (size_t)d.patch_count, &q)) { *err = "truncated container"; return false; }
      d.patches = d.patch_count ? reinterpret_cast<const vpcc_patch*>(q) : nullptr;
      if (!c.take((size_t)occ_w * occ_h, &q)) { *err = "truncated container"; return false; }
      d.occupancy = vpcc_image_u8{q, occ_w, occ_h, occ_w};
      for (uint32_t m = 0; m < d.map_count; ++m) {
        if (!c.take((size_t)geo_w * geo_h * 2, &q)) { *err = "truncated container"; return false; }
        d.geometry[m] = vpcc_image_u16{reinterpret_cast<const uint16_t*>(q), nullptr, nullptr, geo_w, geo_h, geo_w, geo_w / 2};
      }
      if (d.attribute_count) {
        const size_t cw = attr_w / 2, ch = attr_h / 2;
        for (uint32_t m = 0; m < d.map_count; ++m) {
          const unsigned char *y, *u, *v;
          if (!c.take((size_t)attr_w * attr_h * 2, &y) || !c.take(cw * ch * 2, &u) || !c.take(cw * ch * 2, &v)) {
            *err = "truncated container";
            return false;
          }
          d.attribute[m] = vpcc_image_u16{reinterpret_cast<const uint16_t*>(y), reinterpret_cast<const uint16_t*>(u),
                                          reinterpret_cast<const uint16_t*>(v), attr_w, attr_h, attr_w, (uint32_t)cw};
        }
      }
      gof.frames.push_back(d);
    }
    gofs->push_back(std::move(gof));
  }
  return true;
}

// ------------------------------------------------------------------ V3C sample stream + raw decoded video
// The per-GOF driver of the reference (src/decoder.rs:82-314) with the three decompress() calls replaced by
// raw planar files: frame f of a GOF uses occupancy frame f and geometry / attribute frames f*map_count + m
// (src/codec.rs:317, 589-590).
bool parse_v3c_with_raw_video(const std::vector<unsigned char>& bin, const unsigned char* occ, size_t occ_bytes,
                              const unsigned char* geo, size_t geo_bytes, const unsigned char* attr, size_t attr_bytes,
                              uint32_t occupancy_precision, std::vector<DecodedGof>* gofs, std::string* err, int* status) {
  *status = VPCC_ERR_INVALID_ARG;
  gofs->clear();
  if (occupancy_precision == 0) { *err = "occupancy_precision is zero"; return false; }
  std::vector<V3CUnit> units;
  try {
    Bitstream bs(std::vector<uint8_t>(bin.begin(), bin.end()));
    size_t header = 0;
    units = split_sample_stream(bs, &header);
  } catch (const std::exception& e) {
    *err = std::string("not a V3C sample stream: ") + e.what();
    return false;
  }
  size_t next = 0, occ_off = 0, geo_off = 0, attr_off = 0;
  while (next < units.size()) {                       // while ssvu.get_v3c_unit_count() > 0, src/lib.rs:118
    GofSyntax syn;
    std::vector<PatchFrame> frames;
    GofParams gp;
    try {
      next = parse_gof(units, next, &syn);
      gp = build_gof_params(syn);
      frames = build_patch_frames(syn);
    } catch (const SyntaxError& e) {
      *err = e.what();
      *status = e.status;
      return false;
    }
    const uint32_t W = gp.frame_width, H = gp.frame_height;
    if (W == 0 || H == 0 || W % occupancy_precision || H % occupancy_precision || (W & 1) || (H & 1)) {
      *err = "frame size not divisible by the occupancy precision / not even";
      return false;
    }
    if (gp.map_count > 2) { *err = "more than two maps"; *status = VPCC_ERR_UNSUPPORTED; return false; }
    const uint32_t ow = W / occupancy_precision, oh = H / occupancy_precision;
    const size_t occ_frame = (size_t)ow * oh + 2 * (size_t)((ow + 1) / 2) * ((oh + 1) / 2);
    const size_t luma = (size_t)W * H * 2, chroma = (size_t)(W / 2) * (H / 2) * 2;
    const size_t vid_frame = luma + 2 * chroma;
    const bool has_attr = !syn.vps.ai.attributes.empty();
    DecodedGof gof;
    gof.has_syntax = true;
    if (gp.geometry_smoothing_sei && gp.smoothing_grid_size >= 2) {      // SeiGeometrySmoothing, src/bitstream/reader.rs:1452-1505
      gof.sei_smoothing.flags = VPCC_SMOOTH_GEOMETRY;
      gof.sei_smoothing.geometry_bitdepth_3d = gp.geometry_bitdepth_3d;
      gof.sei_smoothing.grid_size = gp.smoothing_grid_size;
      gof.sei_smoothing.threshold = gp.smoothing_threshold;
    }
    gof.patch_store.reserve(frames.size());
    for (size_t f = 0; f < frames.size(); ++f) {
      vpcc_frame_desc d{};
      d.width = W; d.height = H;
      d.occupancy_resolution = gp.occupancy_resolution;
      d.occupancy_precision = occupancy_precision;
      d.map_count = gp.map_count;
      d.absolute_d1 = gp.absolute_d1 ? 1u : 0u;
      d.attribute_count = has_attr ? 1u : 0u;
      gof.patch_store.push_back(std::move(frames[f].patches));
      d.patch_count = (uint32_t)gof.patch_store.back().size();
      d.patches = d.patch_count ? gof.patch_store.back().data() : nullptr;
      if (occ_off + occ_frame > occ_bytes) { *err = "occupancy video shorter than the atlas"; *status = VPCC_ERR_SHORT_VIDEO; return false; }
      d.occupancy = vpcc_image_u8{occ + occ_off, ow, oh, ow};
      occ_off += occ_frame;
      for (uint32_t m = 0; m < gp.map_count; ++m) {
        if (geo_off + vid_frame > geo_bytes) { *err = "geometry video shorter than the atlas"; *status = VPCC_ERR_SHORT_VIDEO; return false; }
        d.geometry[m] = vpcc_image_u16{reinterpret_cast<const uint16_t*>(geo + geo_off), nullptr, nullptr, W, H, W, W / 2};
        geo_off += vid_frame;
        if (has_attr) {
          if (attr_off + vid_frame > attr_bytes) { *err = "attribute video shorter than the atlas"; *status = VPCC_ERR_SHORT_VIDEO; return false; }
          const unsigned char* a = attr + attr_off;
          d.attribute[m] = vpcc_image_u16{reinterpret_cast<const uint16_t*>(a), reinterpret_cast<const uint16_t*>(a + luma),
                                          reinterpret_cast<const uint16_t*>(a + luma + chroma), W, H, W, W / 2};
          attr_off += vid_frame;
        }
      }
      gof.frames.push_back(d);
    }
    gofs->push_back(std::move(gof));
  }
  *status = VPCC_OK;
  return true;
}

}  // namespace tmc2rs
