// decoder.cpp — C++ host mirror of src/lib.rs + the per-GOF driver of src/decoder.rs:188-314, running the
// reconstruction on one or several MI355X through the C ABI (include/vpcc_recon.h).
#include "decoder.hpp"
#include "v3c_syntax.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>
#include <stdexcept>

namespace tmc2rs {

// ------------------------------------------------------------------ container (.vpccgof)
//   header : "VPCCGOF1" | u32 version (1) | u32 gof_count
//   gof    : u32 frame_count | frame*
//   frame  : 16 x u32 { width, height, occupancy_resolution, occupancy_precision, map_count, absolute_d1,
//                       attribute_count, flags, occ_w, occ_h, geo_w, geo_h, attr_w, attr_h, patch_count, 0 }
//            vpcc_patch[patch_count] | occupancy u8[occ_w*occ_h] | geometry Y u16[geo_w*geo_h] x map_count
//            | (Y u16[attr_w*attr_h], U, V u16[(attr_w/2)*((attr_h+1)/2)]) x map_count   (if attribute_count)
//   every section is padded to a multiple of 8 bytes.
namespace {
struct Cursor {
  const unsigned char* p;
  size_t left;
  bool take(size_t n, const unsigned char** out) {
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
      if (!c.take(sizeof(vpcc_patch) * (size_t)d.patch_count, &q)) { *err = "truncated container"; return false; }
      d.patches = d.patch_count ? reinterpret_cast<const vpcc_patch*>(q) : nullptr;
      if (!c.take((size_t)occ_w * occ_h, &q)) { *err = "truncated container"; return false; }
      d.occupancy = vpcc_image_u8{q, occ_w, occ_h, occ_w};
      for (uint32_t m = 0; m < d.map_count; ++m) {
        if (!c.take((size_t)geo_w * geo_h * 2, &q)) { *err = "truncated container"; return false; }
        d.geometry[m] = vpcc_image_u16{reinterpret_cast<const uint16_t*>(q), nullptr, nullptr, geo_w, geo_h, geo_w, geo_w / 2};
      }
      if (d.attribute_count) {
        const size_t cw = attr_w / 2, ch = (attr_h + 1) / 2;
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

// ------------------------------------------------------------------ PinnedPool
PinnedPool::~PinnedPool() { detach(); }

PinnedBlock PinnedPool::get(size_t bytes) {
  if (bytes == 0) bytes = 1;
  {
    std::lock_guard<std::mutex> lk(m_);
    for (size_t k = 0; k < free_.size(); ++k)
      if (free_[k].second >= bytes) {
        PinnedBlock b{free_[k].first, free_[k].second, shared_from_this()};
        free_.erase(free_.begin() + (long)k);
        return b;
      }
  }
  PinnedBlock b;
  const size_t cap = (bytes + (size_t(1) << 20) - 1) & ~((size_t(1) << 20) - 1);   // 1 MiB granules: blocks get reused
  if (ctx_ && vpcc_host_alloc(ctx_, cap, &b.ptr) == VPCC_OK) {
    b.bytes = cap;
    b.pool = shared_from_this();
  }
  return b;
}

void PinnedPool::put(void* ptr, size_t bytes) {
  std::lock_guard<std::mutex> lk(m_);
  if (ctx_ && free_.size() < 8) free_.emplace_back(ptr, bytes);
  else vpcc_host_free(ctx_, ptr);             // also after the context is gone (frames may outlive the decoder)
}

void PinnedPool::detach() {
  std::lock_guard<std::mutex> lk(m_);
  if (ctx_) for (auto& f : free_) vpcc_host_free(ctx_, f.first);
  free_.clear();
  ctx_ = nullptr;
}

// ------------------------------------------------------------------ Decoder
Decoder::Decoder(Params params) : params_(std::move(params)) {}

Decoder::~Decoder() {
  chan_.drop_rx();                            // receiver dropped: the worker's next send fails and it stops
  if (thread_.joinable()) thread_.join();
}

void Decoder::start() {
  if (started_) throw std::logic_error("library decoder can only be started once");   // src/lib.rs:108-111
  started_ = true;
  // Bitstream::from_file on the caller's thread (src/lib.rs:98); the reference unwraps the io error
  auto slurp = [](const std::string& path, std::vector<unsigned char>* out, size_t at) {
    std::ifstream in(path, std::ios::binary | std::ios::ate);
    if (!in) throw std::runtime_error("cannot open " + path);
    const size_t n = (size_t)in.tellg();
    out->resize(at + ((n + 15) & ~size_t(15)));        // next section starts 16-B aligned
    in.seekg(0);
    in.read(reinterpret_cast<char*>(out->data() + at), (std::streamsize)n);
    return n;
  };
  std::vector<unsigned char> head;
  head.resize(slurp(params_.compressed_stream_path, &head, 0));   // no padding behind the stream itself
  std::string err;
  if (head.size() >= 8 && std::memcmp(head.data(), "VPCCGOF1", 8) == 0) {
    file_ = std::move(head);
    if (!parse_container(file_, &gofs_, &err)) throw std::runtime_error(err);
  } else {
    if (params_.occupancy_yuv_path.empty() || params_.geometry_yuv_path.empty())
      throw std::runtime_error("V3C input needs the externally decoded occupancy and geometry videos "
                               "(HEVC decoding is not part of this library)");
    bin_ = std::move(head);
    const size_t o0 = 0, on = slurp(params_.occupancy_yuv_path, &file_, o0);
    const size_t g0 = file_.size(), gn = slurp(params_.geometry_yuv_path, &file_, g0);
    const size_t a0 = file_.size(), an = params_.attribute_yuv_path.empty() ? 0 : slurp(params_.attribute_yuv_path, &file_, a0);
    int status = 0;
    if (!parse_v3c_with_raw_video(bin_, file_.data() + o0, on, file_.data() + g0, gn, file_.data() + a0, an,
                                  params_.occupancy_precision, &gofs_, &err, &status))
      throw std::runtime_error(std::string(vpcc_status_string(status)) + ": " + err);
  }
  thread_ = std::thread([this] { worker(); });
}

std::optional<PointSet3> Decoder::recv_frame() { return chan_.recv(); }

namespace {
struct CtxDeleter { void operator()(vpcc_ctx* c) const { vpcc_ctx_destroy(c); } };
struct GofDeleter { void operator()(vpcc_gof* g) const { vpcc_gof_destroy(g); } };
}  // namespace

void Decoder::worker() {
  // one context per GPU; frames of a GOF are dealt round-robin (frame f -> device f % G) and delivered in
  // presentation order — they are independent (src/decoder.rs:186), so no data moves between GPUs.
  // Plane ingest: the container buffer is page-locked once, so every plane upload is an asynchronous
  // DMA on the context's copy stream, and GOF k+1 is uploaded while GOF k is reconstructed and drained.
  const size_t G = params_.devices.empty() ? 1 : params_.devices.size();
  std::vector<std::unique_ptr<vpcc_ctx, CtxDeleter>> ctxs(G);
  for (size_t d = 0; d < G; ++d) {
    vpcc_ctx* c = nullptr;
    const int st = vpcc_ctx_create(params_.devices.empty() ? 0 : params_.devices[d], &c);
    if (st) { error_ = std::string("vpcc_ctx_create: ") + vpcc_status_string(st); chan_.close_tx(); return; }
    ctxs[d].reset(c);
  }
  auto pool = std::make_shared<PinnedPool>(ctxs[0].get());
  struct Detach {
    std::shared_ptr<PinnedPool> p;
    ~Detach() { p->detach(); }              // frames the consumer still holds outlive the context safely
  } detach{pool};
  const bool pinned = !file_.empty() && vpcc_host_pin(ctxs[0].get(), file_.data(), file_.size()) == VPCC_OK;
  struct Unpin {
    vpcc_ctx* c; const void* p; bool on;
    ~Unpin() { if (on) vpcc_host_unpin(c, p); }
  } unpin{ctxs[0].get(), file_.data(), pinned};

  struct InFlight {                                   // one GOF, sharded over the devices
    std::vector<std::vector<vpcc_frame_desc>> part;
    std::vector<std::unique_ptr<vpcc_gof, GofDeleter>> dg;
    bool ok = false;
  };
  auto launch = [&](const DecodedGof& gof, InFlight* f) -> bool {
    f->part.assign(G, {});
    f->dg.clear();
    f->dg.resize(G);
    for (size_t i = 0; i < gof.frames.size(); ++i) f->part[i % G].push_back(gof.frames[i]);
    for (size_t d = 0; d < G; ++d) {
      if (f->part[d].empty()) continue;
      vpcc_gof* g = nullptr;
      int st = vpcc_gof_create(ctxs[d].get(), f->part[d].data(), (uint32_t)f->part[d].size(), VPCC_MEM_HOST, 0,
                               pinned ? VPCC_GOF_ASYNC_UPLOAD : 0u, &g);
      if (st == VPCC_OK) {
        f->dg[d].reset(g);
        st = vpcc_gof_reconstruct(g, 0, (uint32_t)f->part[d].size(), nullptr);   // asynchronous: all GPUs run concurrently
      }
      if (st) {
        error_ = std::string(vpcc_status_string(st)) + ": " + vpcc_last_error(ctxs[d].get());
        return false;                                 // reference: panic in the worker -> consumer sees end-of-stream
      }
    }
    f->ok = true;
    return true;
  };

  // VPCC_DECODER_TRACE=1: where the worker's wall time goes (stderr, one line at the end of the stream)
  const bool trace = std::getenv("VPCC_DECODER_TRACE") != nullptr;
  double t_launch = 0, t_counts = 0, t_download = 0, t_send = 0;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double>(b - a).count();
  };
  struct Report {
    const bool on; const double &a, &b, &c, &d;
    ~Report() {
      if (on) std::fprintf(stderr, "[vpcc decoder] launch(plan+enqueue) %.3f s, wait-for-counts %.3f s, download %.3f s, send %.3f s\n", a, b, c, d);
    }
  } report{trace, t_launch, t_counts, t_download, t_send};
  // Two GOFs are kept queued behind the one being drained: the copy engine then always has the next
  // upload waiting (with one GOF of look-ahead it idled while the worker downloaded the current GOF:
  // 20 ms per 32-frame GOF instead of the 13.5 ms the 0.58 GB upload takes).
  constexpr size_t kAhead = 2;
  std::deque<InFlight> inflight;
  size_t launched = 0, failed_at = gofs_.size();       // first GOF whose launch failed (none: past the end)
  auto launch_more = [&](size_t upto) {
    while (launched < gofs_.size() && launched <= upto && failed_at == gofs_.size()) {
      inflight.emplace_back();
      const auto t0 = now();
      if (!launch(gofs_[launched], &inflight.back())) {
        failed_at = launched;
        inflight.pop_back();
      } else {
        ++launched;
      }
      t_launch += secs(t0, now());
    }
  };
  for (size_t k = 0; k < gofs_.size(); ++k) {         // while ssvu.get_v3c_unit_count() > 0, src/lib.rs:118
    launch_more(k + kAhead);                          // their ingest overlaps this GOF's work
    if (k >= failed_at) break;                        // the stream ends where the failing GOF would have started
    const DecodedGof& gof = gofs_[k];
    InFlight& cur = inflight.front();
    const size_t n = gof.frames.size();
    std::vector<std::vector<uint32_t>> counts(G);
    for (size_t d = 0; d < G; ++d) {
      if (!cur.dg[d]) continue;
      counts[d].resize(cur.part[d].size());
      const auto t0 = now();
      const int st = vpcc_gof_point_counts(cur.dg[d].get(), counts[d].data());
      t_counts += secs(t0, now());
      if (st) { error_ = vpcc_last_error(ctxs[d].get()); chan_.close_tx(); return; }
    }
    for (size_t f = 0; f < n; ++f) {                  // presentation order, src/decoder.rs:188
      const size_t d = f % G, local = f / G;
      PointSet3 ps;
      ps.with_colors = gof.frames[f].attribute_count > 0;
      const size_t np = counts[d][local];
      PinnedBlock bx = pool->get(np * sizeof(vpcc_point3)), bc;
      if (ps.with_colors) bc = pool->get(np * sizeof(vpcc_color3));
      if (!bx.ptr || (ps.with_colors && !bc.ptr)) { error_ = "out of page-locked memory"; chan_.close_tx(); return; }
      ps.positions.adopt(std::move(bx), np);
      if (ps.with_colors) ps.colors.adopt(std::move(bc), np);
      size_t got = 0;
      const auto t0 = now();
      const int st = vpcc_gof_download(cur.dg[d].get(), (uint32_t)local, ps.positions.data(),
                                       ps.with_colors ? ps.colors.data() : nullptr, nullptr, np ? np : 1, &got);
      const auto t1 = now();
      t_download += secs(t0, t1);
      if (st || got != np) { error_ = vpcc_last_error(ctxs[d].get()); chan_.close_tx(); return; }
      const bool sent = chan_.send(std::move(ps));
      t_send += secs(t1, now());
      if (!sent) { chan_.close_tx(); return; }            // receiver dropped, src/decoder.rs:311-313
    }
    inflight.pop_front();
  }
  chan_.close_tx();                                   // drop(tx), src/lib.rs:136
}

// ------------------------------------------------------------------ PlyWriter (src/writer.rs:25-74)
std::string PlyWriter::to_string(const vpcc_point3* xyz, const vpcc_color3* rgb, size_t n, Format format) {
  std::string s;
  const bool binary = format == Format::BinaryLittleEndian;
  s.reserve(256 + n * (binary ? 15 : 24));
  s += "ply\n";
  s += binary ? "format binary_little_endian 1.0\n" : "format ascii 1.0\n";
  s += "element vertex " + std::to_string(n) + "\n";
  s += "property uint x\nproperty uint y\nproperty uint z\n";
  if (rgb) s += "property uchar red\nproperty uchar green\nproperty uchar blue\n";
  s += "element face 0\n";
  s += "property list uint8 int32 vertex_index\n";
  s += "end_header\n";
  if (binary) {                                   // same properties as the ASCII form: 3 x uint32 (+ 3 x uchar) per vertex
    const size_t stride = rgb ? 15 : 12, at = s.size();
    s.resize(at + n * stride);
    char* o = &s[at];
    for (size_t i = 0; i < n; ++i, o += stride) {
      const uint32_t v[3] = {xyz[i].x, xyz[i].y, xyz[i].z};     // little-endian host (x86-64)
      std::memcpy(o, v, 12);
      if (rgb) { o[12] = (char)rgb[i].r; o[13] = (char)rgb[i].g; o[14] = (char)rgb[i].b; }
    }
    return s;
  }
  char line[64];
  for (size_t i = 0; i < n; ++i) {
    const vpcc_point3& p = xyz[i];
    int k = std::snprintf(line, sizeof line, "%u %u %u", (unsigned)p.x, (unsigned)p.y, (unsigned)p.z);
    if (rgb) {
      const vpcc_color3& c = rgb[i];
      k += std::snprintf(line + k, sizeof line - k, " %u %u %u", (unsigned)c.r, (unsigned)c.g, (unsigned)c.b);
    }
    line[k++] = '\n';
    s.append(line, (size_t)k);
  }
  return s;
}

std::string PlyWriter::to_string() const {
  return to_string(pc_.positions.data(), pc_.with_colors ? pc_.colors.data() : nullptr, pc_.len(), format_);
}

bool PlyWriter::write(const std::string& path) const {
  std::ofstream out(path, std::ios::binary);
  if (!out) return false;
  const std::string s = to_string();
  out.write(s.data(), (std::streamsize)s.size());
  return (bool)out;
}

}  // namespace tmc2rs

// ------------------------------------------------------------------ C ABI of the host mirror
struct vpcc_decoder {
  tmc2rs::Decoder dec;
  std::optional<tmc2rs::PointSet3> cur;
  std::string err;
  double first_frame_seconds = 0;           // vpcc_decoder_drain: start-up latency (contexts, page-locking, first GOF)
  explicit vpcc_decoder(tmc2rs::Params p) : dec(std::move(p)) {}
};

extern "C" int vpcc_decoder_open(const char* path, const int* devices, int n_devices, vpcc_decoder** out) {
  if (!path || !out) return VPCC_ERR_INVALID_ARG;
  tmc2rs::Params p{std::string(path)};
  if (devices && n_devices > 0) p.devices.assign(devices, devices + n_devices);
  *out = new vpcc_decoder(std::move(p));
  return VPCC_OK;
}

extern "C" int vpcc_decoder_open_v3c(const char* bin_path, const char* occupancy_yuv, const char* geometry_yuv,
                                     const char* attribute_yuv, uint32_t occupancy_precision, const int* devices,
                                     int n_devices, vpcc_decoder** out) {
  if (!bin_path || !occupancy_yuv || !geometry_yuv || !out) return VPCC_ERR_INVALID_ARG;
  tmc2rs::Params p{std::string(bin_path)};
  p.occupancy_yuv_path = occupancy_yuv;
  p.geometry_yuv_path = geometry_yuv;
  if (attribute_yuv) p.attribute_yuv_path = attribute_yuv;
  p.occupancy_precision = occupancy_precision;
  if (devices && n_devices > 0) p.devices.assign(devices, devices + n_devices);
  *out = new vpcc_decoder(std::move(p));
  return VPCC_OK;
}

extern "C" int vpcc_decoder_start(vpcc_decoder* d) {
  if (!d) return VPCC_ERR_INVALID_ARG;
  try {
    d->dec.start();
  } catch (const std::logic_error& e) {
    d->err = e.what();
    return VPCC_ERR_STATE;
  } catch (const std::exception& e) {
    d->err = e.what();
    return VPCC_ERR_INVALID_ARG;
  }
  return VPCC_OK;
}

extern "C" int vpcc_decoder_recv_frame(vpcc_decoder* d, size_t* n_points, const vpcc_point3** xyz,
                                       const vpcc_color3** rgb) {
  if (!d || !n_points) return 0;
  d->cur = d->dec.recv_frame();
  if (!d->cur) return 0;                      // None: end of stream (or the worker failed: vpcc_decoder_error)
  *n_points = d->cur->len();
  if (xyz) *xyz = d->cur->positions.data();
  if (rgb) *rgb = d->cur->with_colors ? d->cur->colors.data() : nullptr;
  return 1;
}

extern "C" const char* vpcc_decoder_error(vpcc_decoder* d) {
  if (!d) return "";
  return d->dec.last_error().empty() ? d->err.c_str() : d->dec.last_error().c_str();   // the worker's error wins
}

extern "C" int vpcc_decoder_drain(vpcc_decoder* d, uint64_t* frames, uint64_t* points, double* seconds) {
  if (!d) return VPCC_ERR_INVALID_ARG;
  const auto t0 = std::chrono::steady_clock::now();
  uint64_t nf = 0, np = 0;
  while (auto fr = d->dec.recv_frame()) {
    if (nf == 0) d->first_frame_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    ++nf;
    np += fr->len();
  }
  const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (frames) *frames = nf;
  if (points) *points = np;
  if (seconds) *seconds = s;
  return d->dec.last_error().empty() ? VPCC_OK : VPCC_ERR_DEVICE;
}

extern "C" double vpcc_decoder_first_frame_seconds(const vpcc_decoder* d) { return d ? d->first_frame_seconds : 0.0; }

extern "C" void vpcc_decoder_close(vpcc_decoder* d) { delete d; }

extern "C" int vpcc_write_ply_format(const char* path, const vpcc_point3* xyz, const vpcc_color3* rgb, size_t n, int binary) {
  if (!path || (n && !xyz)) return VPCC_ERR_INVALID_ARG;
  std::ofstream out(path, std::ios::binary);
  if (!out) return VPCC_ERR_INVALID_ARG;
  const std::string s = tmc2rs::PlyWriter::to_string(xyz, rgb, n, binary ? tmc2rs::Format::BinaryLittleEndian : tmc2rs::Format::Ascii);
  out.write(s.data(), (std::streamsize)s.size());
  return out ? VPCC_OK : VPCC_ERR_INVALID_ARG;
}

extern "C" int vpcc_write_ply(const char* path, const vpcc_point3* xyz, const vpcc_color3* rgb, size_t n) {
  return vpcc_write_ply_format(path, xyz, rgb, n, 0);
}
