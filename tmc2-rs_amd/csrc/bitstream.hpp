// bitstream.hpp — C++ host counterpart of the reference's V3C bit reader and stream re-framing
// (first instalment).  Own implementation of the behaviour of:
//   Bitstream::read / peek / read_uvlc / read_svlc / byte_align / copy_from / more_data
//                                                       src/bitstream.rs:53-190
//   SampleStreamV3CUnit::from_bitstream / read_header / read_v3c_unit   src/bitstream/reader.rs:623-670
//   VideoBitstream::sample_stream_to_bytestream (NAL length prefixes -> Annex-B start codes)
//                                                       src/bitstream.rs:216-289
//   the Intra-PDU -> Patch mapping of create_patch_frame   src/decoder.rs:415-486
// Pinned by the reference's own five bit-reader tests (src/bitstream.rs:349-437), re-expressed in
// tests/test_bitstream.py with the same vectors.  The atlas syntax parser (VPS/ASPS/AFPS/SEI/ATL,
// src/bitstream/reader.rs) is v3c_syntax.{hpp,cpp}.
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

#include "vpcc_recon.h"

namespace tmc2rs {

class Bitstream {
 public:
  Bitstream() = default;
  explicit Bitstream(std::vector<uint8_t> data) : data_(std::move(data)) {}

  uint32_t read(unsigned bits);             // MSB first; bits <= 32; throws std::out_of_range past the end
  uint32_t peek(unsigned bits);
  uint32_t read_uvlc();                     // 0-th order Exp-Golomb
  int32_t read_svlc();
  void byte_align();                        // reads the stop bit, then skips to the next byte boundary
  void copy_from(Bitstream& src, size_t start_byte, size_t size);   // advances BOTH positions by `size`
  bool more_data() const { return bytes_ < data_.size(); }
  void reset() { bytes_ = 0; bits_ = 0; }
  void seek(size_t byte) { bytes_ = byte; bits_ = 0; }
  size_t position_bytes() const { return bytes_; }
  unsigned position_bits() const { return bits_; }
  const std::vector<uint8_t>& data() const { return data_; }

 private:
  std::vector<uint8_t> data_;
  size_t bytes_ = 0;
  unsigned bits_ = 0;
};

struct V3CUnit {
  uint8_t unit_type = 0;                    // data[0] >> 3
  std::vector<uint8_t> payload;
};

// Splits a V3C sample stream into its units; *header_size as the reference accounts it.
std::vector<V3CUnit> split_sample_stream(Bitstream& bs, size_t* header_size);

enum class CodecId { H264 = 0, H265 = 1, H266 = 2 };
std::vector<uint8_t> sample_stream_to_bytestream(const std::vector<uint8_t>& data, CodecId codec, size_t precision);

}  // namespace tmc2rs
