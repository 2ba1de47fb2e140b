// bitstream.cpp — see bitstream.hpp.  Host-only code (no GPU calls).
#include "bitstream.hpp"

#include <cmath>
#include <cstring>
#include <stdexcept>

namespace tmc2rs {

uint32_t Bitstream::read(unsigned bits) {
  if (bits > 32) throw std::invalid_argument("Bitstream::read: bits > 32");      // src/bitstream.rs:135-137
  uint32_t val = 0;
  for (unsigned i = 0; i < bits; ++i) {
    if (bytes_ >= data_.size()) throw std::out_of_range("Bitstream::read past the end");   // the reference's slice index panics
    val |= (uint32_t)((data_[bytes_] >> (7 - bits_)) & 1u) << (bits - i - 1);
    if (++bits_ == 8) { ++bytes_; bits_ = 0; }
  }
  return val;
}

uint32_t Bitstream::peek(unsigned bits) {
  const size_t b = bytes_;
  const unsigned s = bits_;
  const uint32_t v = read(bits);
  bytes_ = b;
  bits_ = s;
  return v;
}

uint32_t Bitstream::read_uvlc() {
  unsigned leading_zeros = 0;
  while (read(1) == 0) {
    // the reference's `1 << leading_zeros` / read(> 32 bits) panic on such a code (src/bitstream.rs:166-175)
    if (++leading_zeros >= 32) throw std::out_of_range("Exp-Golomb code longer than 32 bits");
  }
  if (leading_zeros == 0) return 0;
  return (1u << leading_zeros) - 1u + read(leading_zeros);
}

int32_t Bitstream::read_svlc() {
  const uint32_t x = read_uvlc();
  return (x & 1u) ? (int32_t)(x >> 1) + 1 : -(int32_t)(x >> 1);
}

void Bitstream::byte_align() {
  read(1);                                  // the reference keeps upstream's "read one bit first" wrinkle
  if (bits_ != 0) { bits_ = 0; ++bytes_; }
}

void Bitstream::copy_from(Bitstream& src, size_t start_byte, size_t size) {
  if (start_byte + size > src.data_.size()) throw std::out_of_range("Bitstream::copy_from source range");
  if (data_.size() < bytes_ + size) data_.resize(bytes_ + size, 0);
  if (size) std::memcpy(data_.data() + bytes_, src.data_.data() + start_byte, size);
  bytes_ += size;
  src.bytes_ += size;
}

std::vector<V3CUnit> split_sample_stream(Bitstream& bs, size_t* header_size) {
  // read_header: u(3) unit size precision minus 1, u(5) padding   (src/bitstream/reader.rs:644-648)
  const unsigned precision = bs.read(3) + 1;
  bs.read(5);
  size_t hs = 1;
  std::vector<V3CUnit> units;
  while (bs.more_data()) {
    V3CUnit u;
    const size_t size = bs.read(8 * precision);
    Bitstream payload;
    payload.copy_from(bs, bs.position_bytes(), size);
    u.payload = payload.data();
    u.unit_type = u.payload.empty() ? 0 : (uint8_t)(u.payload[0] >> 3);
    units.push_back(std::move(u));
    hs += precision;
  }
  if (header_size) *header_size = hs;
  return units;
}

std::vector<uint8_t> sample_stream_to_bytestream(const std::vector<uint8_t>& data, CodecId codec, size_t precision) {
  if (precision != 4) throw std::invalid_argument("precision must be 4");         // assert_eq!(precision, 4)
  size_t size_start_code = 4, start = 0;
  bool new_frame = true;
  std::vector<uint8_t> out;
  out.reserve(data.size());
  for (;;) {
    if (start + precision > data.size()) throw std::out_of_range("truncated sample stream");
    size_t nalu = 0;
    for (size_t i = 0; i < precision; ++i) nalu = (nalu << 8) + data[start + i];
    const size_t end = start + precision + nalu;
    if (end > data.size()) throw std::out_of_range("truncated sample stream");
    for (size_t i = 0; i + 1 < size_start_code; ++i) out.push_back(0);
    out.push_back(1);
    out.insert(out.end(), data.begin() + (long)(start + precision), data.begin() + (long)end);
    start = end;
    if (start + precision < data.size()) {
      bool long_code = true;
      new_frame = false;                    // as in the reference: cleared before the per-codec test
      if (codec == CodecId::H265) {
        const unsigned t = (data[start + precision] & 126u) >> 1;
        long_code = new_frame || (t >= 32 && t < 41);
        if (t < 12) new_frame = true;
      } else if (codec == CodecId::H266) {
        const unsigned t = (data[start + precision + 1] & 248u) >> 3;
        long_code = new_frame || (t >= 12 && t < 20);
        if (t < 12) new_frame = true;
      }
      size_start_code = long_code ? 4 : 3;
    }
    if (end >= data.size()) break;
  }
  return out;
}

}  // namespace tmc2rs

// ------------------------------------------------------------------ C ABI (host-only helpers)
struct vpcc_bitstream { tmc2rs::Bitstream bs; };

extern "C" vpcc_bitstream* vpcc_bs_new(const uint8_t* data, size_t n) {
  return new vpcc_bitstream{tmc2rs::Bitstream(std::vector<uint8_t>(data, data + n))};
}
extern "C" void vpcc_bs_free(vpcc_bitstream* b) { delete b; }
extern "C" int vpcc_bs_read(vpcc_bitstream* b, unsigned bits, uint32_t* out) {
  try { *out = b->bs.read(bits); } catch (...) { return VPCC_ERR_INVALID_ARG; }
  return VPCC_OK;
}
extern "C" int vpcc_bs_peek(vpcc_bitstream* b, unsigned bits, uint32_t* out) {
  try { *out = b->bs.peek(bits); } catch (...) { return VPCC_ERR_INVALID_ARG; }
  return VPCC_OK;
}
extern "C" int vpcc_bs_read_uvlc(vpcc_bitstream* b, uint32_t* out) {
  try { *out = b->bs.read_uvlc(); } catch (...) { return VPCC_ERR_INVALID_ARG; }
  return VPCC_OK;
}
extern "C" int vpcc_bs_read_svlc(vpcc_bitstream* b, int32_t* out) {
  try { *out = b->bs.read_svlc(); } catch (...) { return VPCC_ERR_INVALID_ARG; }
  return VPCC_OK;
}
extern "C" int vpcc_bs_byte_align(vpcc_bitstream* b) {
  try { b->bs.byte_align(); } catch (...) { return VPCC_ERR_INVALID_ARG; }
  return VPCC_OK;
}
extern "C" void vpcc_bs_reset(vpcc_bitstream* b) { b->bs.reset(); }
extern "C" int vpcc_bs_copy_from(vpcc_bitstream* dst, vpcc_bitstream* src, size_t start_byte, size_t size) {
  try { dst->bs.copy_from(src->bs, start_byte, size); } catch (...) { return VPCC_ERR_INVALID_ARG; }
  return VPCC_OK;
}
extern "C" size_t vpcc_bs_data(const vpcc_bitstream* b, const uint8_t** data) {
  if (data) *data = b->bs.data().data();
  return b->bs.data().size();
}
extern "C" void vpcc_bs_position(const vpcc_bitstream* b, size_t* bytes, unsigned* bits) {
  if (bytes) *bytes = b->bs.position_bytes();
  if (bits) *bits = b->bs.position_bits();
}

extern "C" int vpcc_v3c_split(const uint8_t* data, size_t n, uint32_t max_units, uint8_t* types, size_t* offsets,
                              size_t* sizes, uint32_t* n_units, size_t* header_size) {
  if (!data || !n_units) return VPCC_ERR_INVALID_ARG;
  try {
    tmc2rs::Bitstream bs(std::vector<uint8_t>(data, data + n));
    const unsigned precision = (unsigned)(data[0] >> 5) + 1;
    size_t hs = 0;
    const auto units = tmc2rs::split_sample_stream(bs, &hs);
    size_t off = 1;
    uint32_t k = 0;
    for (const auto& u : units) {
      off += precision;
      if (k < max_units) {
        if (types) types[k] = u.unit_type;
        if (offsets) offsets[k] = off;
        if (sizes) sizes[k] = u.payload.size();
      }
      off += u.payload.size();
      ++k;
    }
    *n_units = k;
    if (header_size) *header_size = hs;
  } catch (...) {
    return VPCC_ERR_INVALID_ARG;
  }
  return VPCC_OK;
}

extern "C" int vpcc_sample_stream_to_bytestream(const uint8_t* data, size_t n, int codec_id, uint8_t* out,
                                                size_t out_capacity, size_t* out_size) {
  if (!data || !out_size) return VPCC_ERR_INVALID_ARG;
  try {
    const auto r = tmc2rs::sample_stream_to_bytestream(std::vector<uint8_t>(data, data + n),
                                                       (tmc2rs::CodecId)codec_id, 4);
    *out_size = r.size();
    if (out) {
      if (r.size() > out_capacity) return VPCC_ERR_CAPACITY;
      std::memcpy(out, r.data(), r.size());
    }
  } catch (...) {
    return VPCC_ERR_INVALID_ARG;
  }
  return VPCC_OK;
}

// Intra PDU -> Patch, src/decoder.rs:415-486 (the fields the hot path reads; see vpcc_patch).
extern "C" int vpcc_patch_from_intra_pdu(const vpcc_patch_frame_params* fp, const vpcc_intra_pdu* pdu, vpcc_patch* out) {
  if (!fp || !pdu || !out) return VPCC_ERR_INVALID_ARG;
  if (pdu->lod_enabled_flag) return VPCC_ERR_UNSUPPORTED;                        // decoder.rs:432-433
  if (fp->plr_enabled_flag) return VPCC_ERR_UNSUPPORTED;                         // decoder.rs:482-484
  vpcc_patch p{};
  const uint32_t block = 1u << fp->log2_patch_packing_block_size;
  p.u0 = pdu->pos_2d_x; p.v0 = pdu->pos_2d_y;
  p.u1 = pdu->pos_3d_offset_u; p.v1 = pdu->pos_3d_offset_v;
  p.lod_x = p.lod_y = 1;
  if (fp->patch_size_quantizer_present_flag) {                                   // decoder.rs:442-452 (f64 ceil)
    p.size_u0 = (uint32_t)std::ceil((double)(pdu->size_2d_x_minus1 + 1) * (double)(1u << fp->patch_size_info_quantizer_x) / (double)block);
    p.size_v0 = (uint32_t)std::ceil((double)(pdu->size_2d_y_minus1 + 1) * (double)(1u << fp->patch_size_info_quantizer_y) / (double)block);
  } else {
    p.size_u0 = pdu->size_2d_x_minus1 + 1;
    p.size_v0 = pdu->size_2d_y_minus1 + 1;
  }
  p.orientation = (uint8_t)pdu->orientation_index;
  // set_view_id, decoder.rs:788-814: {additional plane, normal, tangent, bitangent, mode}
  static const uint8_t view[18][5] = {{0, 0, 2, 1, 0}, {0, 1, 2, 0, 0}, {0, 2, 0, 1, 0}, {0, 0, 2, 1, 1}, {0, 1, 2, 0, 1},
                                      {0, 2, 0, 1, 1}, {1, 0, 2, 1, 0}, {1, 2, 0, 1, 0}, {1, 0, 2, 1, 1}, {1, 2, 0, 1, 1},
                                      {2, 2, 0, 1, 0}, {2, 1, 2, 0, 0}, {2, 2, 0, 1, 1}, {2, 1, 2, 0, 1}, {3, 1, 2, 0, 0},
                                      {3, 0, 2, 1, 0}, {3, 1, 2, 0, 1}, {3, 0, 2, 1, 1}};
  if (pdu->projection_id > 17) return VPCC_ERR_INVALID_ARG;                      // unreachable!()
  const uint8_t* v = view[pdu->projection_id];
  p.axis_of_additional_plane = v[0];
  p.normal_axis = v[1]; p.tangent_axis = v[2]; p.bitangent_axis = v[3];
  p.projection_mode = v[4];
  const uint32_t min_level = 1u << fp->pos_min_d_quantizer;                      // decoder.rs:410
  if (p.projection_mode == 0) p.d1 = pdu->pos_3d_offset_d * min_level;           // decoder.rs:468-473
  else p.d1 = (1u << fp->geometry_3d_bitdepth) - pdu->pos_3d_offset_d * min_level;
  *out = p;
  return VPCC_OK;
}
