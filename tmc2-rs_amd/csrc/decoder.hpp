// decoder.hpp — C++ host mirror of the reference's public library API (src/lib.rs:15-154):
//   Params, Decoder::new / start() / recv_frame() / Iterator<Item = PointSet3>, and writer::PlyWriter
//   (src/writer.rs:8-74).  Same names, argument meaning and observable behaviour:
//   * start() reads the input on the CALLER's thread, then spawns ONE worker thread; calling it twice
//     throws (the reference panics: "library decoder can only be started once", src/lib.rs:108-111);
//   * frames arrive in presentation order through a capacity-1 channel (crossbeam bounded(1), src/lib.rs:72):
//     the producer is at most one frame ahead of the consumer;
//   * any failure in the worker ends the stream early — recv_frame() returns nullopt from then on, as the
//     reference's consumer sees None after a worker panic (src/lib.rs:113-145).
// The reconstruction itself is the GPU path behind include/vpcc_recon.h.  Input is a decoded-GOF
// container (.vpccgof: patch tables + decoded planes, i.e. the state after the reference's three
// decompress() calls, src/decoder.rs:82-171), or a V3C sample stream parsed by v3c_syntax.cpp together with
// its externally decoded raw videos; HEVC decoding itself is out of scope (no decoder in this image).
#pragma once

#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <optional>
#include <string>
#include <thread>
#include <vector>

#include "vpcc_recon.h"

namespace tmc2rs {

struct Params {                       // src/lib.rs:23-57
  std::string compressed_stream_path; // a .vpccgof decoded-GOF container, or a V3C sample stream (.bin)
  std::vector<int> devices{0};        // GPUs to shard frames over (extension; the reference is single-threaded)
  bool keep_intermediate_files = false;
  // V3C input only: the three video sub-bitstreams decoded by an EXTERNAL HEVC decoder to raw planar
  // 4:2:0 files in the decoder's native format, all GOFs concatenated — occupancy 8-bit, geometry and
  // attribute 16-bit little endian (10-bit content), i.e. exactly the bytes the reference copies out of
  // libav's AVFrame (src/decoder.rs:1131-1141).  The attribute path is empty for attribute-less streams.
  std::string occupancy_yuv_path, geometry_yuv_path, attribute_yuv_path;
  uint32_t occupancy_precision = 4;   // frame_width / occupancy video width (the reference derives it from
                                      // the decoded video, src/decoder.rs:194; a raw file carries no size)
  // Post-processing switches of the reference's Params (src/lib.rs:45-46: private and always false there, and
  // `unimplemented!()` behind them, src/decoder.rs:291-299).  Geometry smoothing runs when the switch is on AND the GOF
  // carries a geometry-smoothing SEI (src/decoder.rs:291, 630-637): grid size and threshold are the SEI's.  The
  // reference parses no attribute-smoothing SEI (src/bitstream/reader.rs:1370-1505 reads the geometry one only), so the
  // colour filter takes its grid size and thresholds from `attr_smoothing` (ColorSmoothingParams, src/codec.rs:180-186)
  // whenever its switch is on.  Both filters are this library's own specification (oracle/vpcc_smoothing_spec.h).
  bool apply_geo_smoothing_type = false;
  bool apply_attr_smoothing_type = false;
  vpcc_smoothing_params attr_smoothing{};   // color_grid_size, color_threshold_smoothing, color_threshold_difference
  // Inputs that carry no SEI (a .vpccgof container): geometry smoothing with these parameters when the switch is on
  // and grid_size >= 2 (geometry_bitdepth_3d, grid_size, threshold).
  vpcc_smoothing_params geo_smoothing_without_sei{};
  explicit Params(std::string path = {}) : compressed_stream_path(std::move(path)) {}
};

// Pool of page-locked host blocks: a frame's storage is DMA'd into directly and returns to the pool when
// the consumer drops the frame (the reference moves owned Vecs through its channel, src/decoder.rs:311).
class PinnedPool;
struct PinnedBlock {
  void* ptr = nullptr;
  size_t bytes = 0;
  std::shared_ptr<PinnedPool> pool;   // null: plain heap memory (no GPU context, tests)
};

// Contiguous array with Vec-like read access, backed by a pooled pinned block.
template <class T>
class PinnedVec {
 public:
  PinnedVec() = default;
  PinnedVec(PinnedVec&& o) noexcept { *this = std::move(o); }
  PinnedVec& operator=(PinnedVec&& o) noexcept {
    if (this != &o) { release(); blk_ = o.blk_; n_ = o.n_; o.blk_ = PinnedBlock{}; o.n_ = 0; }
    return *this;
  }
  PinnedVec(const PinnedVec&) = delete;
  PinnedVec& operator=(const PinnedVec&) = delete;
  ~PinnedVec() { release(); }
  void adopt(PinnedBlock b, size_t n) { release(); blk_ = std::move(b); n_ = n; }
  size_t size() const { return n_; }
  bool empty() const { return n_ == 0; }
  T* data() { return static_cast<T*>(blk_.ptr); }
  const T* data() const { return static_cast<const T*>(blk_.ptr); }
  const T& operator[](size_t i) const { return data()[i]; }
  T& operator[](size_t i) { return data()[i]; }
  const T* begin() const { return data(); }
  const T* end() const { return data() + n_; }

 private:
  void release();
  PinnedBlock blk_;
  size_t n_ = 0;
};

struct PointSet3 {                    // src/codec.rs:20-36 (public part)
  PinnedVec<vpcc_point3> positions;
  PinnedVec<vpcc_color3> colors;
  bool with_colors = false;
  size_t len() const { return positions.size(); }
};

class PinnedPool : public std::enable_shared_from_this<PinnedPool> {
 public:
  explicit PinnedPool(vpcc_ctx* ctx) : ctx_(ctx) {}
  ~PinnedPool();
  PinnedBlock get(size_t bytes);      // a block of at least `bytes` (empty ptr on failure)
  void put(void* ptr, size_t bytes);
  void detach();                      // the context is going away: free what is pooled, stop pooling

 private:
  std::mutex m_;
  vpcc_ctx* ctx_;
  std::vector<std::pair<void*, size_t>> free_;
};

template <class T>
void PinnedVec<T>::release() {
  if (blk_.ptr) {
    if (blk_.pool) blk_.pool->put(blk_.ptr, blk_.bytes);
    else ::operator delete(blk_.ptr);
  }
  blk_ = PinnedBlock{};
  n_ = 0;
}

// Capacity-bounded single-producer/single-consumer channel with close(), like crossbeam bounded(n).
template <class T>
class BoundedChannel {
 public:
  explicit BoundedChannel(size_t cap) : cap_(cap) {}
  bool send(T v) {                    // false when the receiver is gone
    std::unique_lock<std::mutex> lk(m_);
    cv_.wait(lk, [&] { return q_.size() < cap_ || rx_dropped_; });
    if (rx_dropped_) return false;
    q_.push_back(std::move(v));
    cv_.notify_all();
    return true;
  }
  std::optional<T> recv() {           // nullopt once the sender is closed and the queue is drained
    std::unique_lock<std::mutex> lk(m_);
    cv_.wait(lk, [&] { return !q_.empty() || tx_closed_; });
    if (q_.empty()) return std::nullopt;
    T v = std::move(q_.front());
    q_.pop_front();
    cv_.notify_all();
    return v;
  }
  void close_tx() { std::lock_guard<std::mutex> lk(m_); tx_closed_ = true; cv_.notify_all(); }
  void drop_rx() { std::lock_guard<std::mutex> lk(m_); rx_dropped_ = true; cv_.notify_all(); }

 private:
  size_t cap_;
  std::mutex m_;
  std::condition_variable cv_;
  std::deque<T> q_;
  bool tx_closed_ = false, rx_dropped_ = false;
};

// One decoded GOF: frame descriptors pointing into the container buffer.
struct DecodedGof {
  std::vector<vpcc_frame_desc> frames;
  // the GOF's geometry-smoothing SEI (V3C input; flags == 0: none): geometry_bitdepth_3d, grid_size, threshold
  vpcc_smoothing_params sei_smoothing{};
  bool has_syntax = false;                            // V3C input: a GOF without the SEI is NOT smoothed (src/decoder.rs:291); a
                                                      // container carries no syntax: Params' geometry parameters decide
  std::vector<std::vector<vpcc_patch>> patch_store;   // V3C input: patch tables built by the syntax parser
};

// Builds the GOFs of a V3C sample stream `bin` whose decoded videos lie back to back in `yuv` at the given
// offsets (see Params).  Returns false + message on error; *status receives the vpcc_status of a syntax error.
bool parse_v3c_with_raw_video(const std::vector<unsigned char>& bin, const unsigned char* occ, size_t occ_bytes,
                              const unsigned char* geo, size_t geo_bytes, const unsigned char* attr, size_t attr_bytes,
                              uint32_t occupancy_precision, std::vector<DecodedGof>* gofs, std::string* err, int* status);

// Parses a .vpccgof container held in `buf` (kept alive by the caller).  Returns false + message on error.
bool parse_container(const std::vector<unsigned char>& buf, std::vector<DecodedGof>* gofs, std::string* err);

class Decoder {
 public:
  explicit Decoder(Params params);
  ~Decoder();
  Decoder(const Decoder&) = delete;
  Decoder& operator=(const Decoder&) = delete;

  void start();                               // src/lib.rs:97-138
  std::optional<PointSet3> recv_frame();      // src/lib.rs:143-145
  const std::string& last_error() const { return error_; }   // extension: why the stream ended early
  using Stats = vpcc_decoder_stats_t;         // launches, frames per launch, kernel time, lane affinity (vpcc_recon.h)
  Stats stats() const { return stats_; }      // complete once recv_frame() has returned nullopt

  struct iterator {                           // impl Iterator for Decoder, src/lib.rs:148-154
    Decoder* d;
    std::optional<PointSet3> cur;
    PointSet3& operator*() { return *cur; }
    iterator& operator++() { cur = d->recv_frame(); return *this; }
    bool operator!=(const iterator&) const { return cur.has_value(); }
  };
  iterator begin() { iterator it{this, std::nullopt}; ++it; return it; }
  iterator end() { return iterator{this, std::nullopt}; }

 private:
  void worker();
  struct LaneSet;                           // one thread + vpcc_ctx per GPU (decoder.cpp)
  std::shared_ptr<LaneSet> lanes_;
  Params params_;
  BoundedChannel<PointSet3> chan_{1};
  std::vector<unsigned char> file_;
  std::vector<unsigned char> bin_;          // V3C input: the sample stream (file_ then holds the raw videos)
  std::vector<DecodedGof> gofs_;
  std::thread thread_;
  bool started_ = false;
  std::string error_;
  Stats stats_{};
};

enum class Format { Ascii, BinaryLittleEndian };   // src/writer.rs:8-12 (the binary variant is commented out there)

class PlyWriter {                             // src/writer.rs:14-74
 public:
  PlyWriter(const PointSet3& pc, Format format) : pc_(pc), format_(format) {}
  // plain arrays (C ABI entry point)
  static std::string to_string(const vpcc_point3* xyz, const vpcc_color3* rgb, size_t n, Format format = Format::Ascii);
  bool write(const std::string& path) const;
  std::string to_string() const;

 private:
  const PointSet3& pc_;
  Format format_;
};

}  // namespace tmc2rs
