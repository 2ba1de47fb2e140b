// decoder.cpp — C++ host mirror of src/lib.rs + the per-GOF driver of src/decoder.rs:188-314, running the
// reconstruction on one or several MI355X through the C ABI (include/vpcc_recon.h).
#include "decoder.hpp"
#include "v3c_syntax.hpp"

#include <cctype>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <future>
#include <memory>
#include <sstream>
#include <stdexcept>

#include <hip/hip_runtime_api.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <unistd.h>

namespace tmc2rs {

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
  if (ctx_ && free_.size() < 64) free_.emplace_back(ptr, bytes);
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
  lanes_.reset();                             // (after the worker: contexts are destroyed on their lanes' threads)
}

namespace {
// The NUMA nodes the input buffer is to be spread over, as a bit mask: the nodes of the lanes' GPUs when they are more than
// one (sysfs: /sys/bus/pci/devices/<bus id>/numa_node) — every lane pulls 57 GB/s out of this buffer, and eight lanes out of
// ONE node's memory would be 450 GB/s from one socket; pages interleaved over the GPUs' nodes share the load between the
// sockets' memory controllers (a frame's planes are 18 MB = nine huge pages: every lane reads from every node).  0: leave the
// placement to the first touch (one GPU, GPUs of one node, a machine that reports no nodes).  VPCC_DECODER_INTERLEAVE_NODES=a,b,..
// names the nodes outright (tests; machines whose sysfs says nothing).
unsigned long input_node_mask(const std::vector<int>& devices) {
  unsigned long mask = 0;
  if (const char* e = std::getenv("VPCC_DECODER_INTERLEAVE_NODES")) {
    for (const char* c = e; *c;) {
      char* end = nullptr;
      const long v = std::strtol(c, &end, 10);
      if (end == c) break;
      if (v >= 0 && v < (long)(8 * sizeof mask)) mask |= 1ul << v;
      c = *end ? end + 1 : end;
    }
    return mask;
  }
  if (devices.size() < 2) return 0;                     // one lane: one node (and no HIP call on the caller's thread)
  for (int dev : devices) {
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    for (char* c = bus; *c; ++c) *c = (char)std::tolower((unsigned char)*c);      // sysfs names are lower case
    int node = -1;
    if (FILE* f = std::fopen((std::string("/sys/bus/pci/devices/") + bus + "/numa_node").c_str(), "r")) {
      if (std::fscanf(f, "%d", &node) != 1) node = -1;
      std::fclose(f);
    }
    if (node < 0 || node >= (int)(8 * sizeof mask)) return 0;
    mask |= 1ul << node;
  }
  return (mask & (mask - 1)) ? mask : 0;                // two nodes or more
}
}  // namespace

struct CtxDeleter { void operator()(vpcc_ctx* c) const { vpcc_ctx_destroy(c); } };

// One lane per GPU: a thread that owns the device's vpcc_ctx and runs every call on it, in FIFO order.  A
// vpcc_ctx is a single-thread object (include/vpcc_recon.h), and with one lane per device the host work of
// a GOF — frame planning, vpcc_gof_create with its H2D enqueues, the result downloads — runs in parallel
// across the devices instead of on one thread that would bound an 8-GPU node.  The thread binds itself to the
// CPUs of its GPU's NUMA node before it allocates anything: its page-locked result pool, its HIP calls and the
// host side of its copy engines stay on the socket next to the device.
class Lane {
 public:
  explicit Lane(int device) : device_(device), th_([this] { run(); }) {}
  ~Lane() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
    }
    cv_.notify_all();
    th_.join();
  }
  std::future<int> post(std::function<int(vpcc_ctx*)> f) {
    std::packaged_task<int(vpcc_ctx*)> t(std::move(f));
    std::future<int> r = t.get_future();
    {
      std::lock_guard<std::mutex> lk(m_);
      q_.push_back(std::move(t));
    }
    cv_.notify_all();
    return r;
  }
  // valid once a posted task has run
  int numa_node() const { return numa_node_; }
  const std::shared_ptr<PinnedPool>& pool() const { return pool_; }

 private:
  // One 16 x 16 frame through the whole path — create (allocations, page-locked staging, the descriptor copy), launch,
  // point counts, destroy: whatever the runtime does at the first call of a kind (memory pools, copy queues, events: 8 ms
  // in front of a cold process's first unit) happens here, while start() reads the input.
  static void warm_up(vpcc_ctx* c) {
    // (planes in page-locked memory, uploaded asynchronously as one stretch of a few MB: the copy engine's queue for such
    // copies is the slowest thing to come up — a stretch of a few hundred bytes goes another way and left 8 ms in front of a
    // cold process's first unit)
    constexpr uint32_t W = 1024, H = 512;
    constexpr size_t kOcc = (size_t)(W / 4) * (H / 4), kLuma = (size_t)W * H * 2, kBytes = 4096 + kOcc + kLuma;
    void* mem = nullptr;
    if (vpcc_host_alloc(c, kBytes, &mem) != VPCC_OK) return;
    std::memset(mem, 0, kBytes);
    uint8_t* occ = (uint8_t*)mem;
    occ[0] = 1;
    const uint16_t* plane = (const uint16_t*)((char*)mem + ((kOcc + 4095) & ~size_t(4095)));   // W x H samples, shared by every plane of the frame
    vpcc_patch p{};
    p.size_u0 = p.size_v0 = 1; p.lod_x = p.lod_y = 1; p.tangent_axis = 1; p.bitangent_axis = 2;
    vpcc_frame_desc f{};
    f.width = W; f.height = H; f.occupancy_resolution = 16; f.occupancy_precision = 4; f.map_count = 2; f.absolute_d1 = 1; f.attribute_count = 1;
    f.occupancy = vpcc_image_u8{occ, W / 4, H / 4, W / 4};
    for (int m = 0; m < 2; ++m) {
      f.geometry[m] = vpcc_image_u16{plane, nullptr, nullptr, W, H, W, W / 2};
      f.attribute[m] = vpcc_image_u16{plane, plane, plane, W, H, W, W / 2};
    }
    f.patches = &p; f.patch_count = 1;
    vpcc_gof* g = nullptr;
    if (vpcc_gof_create(c, &f, 1, VPCC_MEM_HOST, 0, VPCC_GOF_ASYNC_UPLOAD, &g) == VPCC_OK) {
      uint32_t n = 0;
      if (vpcc_gof_reconstruct(g, 0, 1, nullptr) == VPCC_OK) (void)vpcc_gof_point_counts(g, &n);
      vpcc_gof_destroy(g);
    }
    (void)vpcc_host_free(c, mem);
  }
  void run() {
    vpcc_ctx* c = nullptr;
    create_status_ = vpcc_ctx_create(device_, &c);
    ctx_.reset(c);
    if (c) {
      (void)vpcc_ctx_bind_thread(c, &numa_node_);
      pool_ = std::make_shared<PinnedPool>(c);        // blocks are allocated by this thread (PinnedPool::get in a task)
      warm_up(c);
    }
    for (;;) {
      std::packaged_task<int(vpcc_ctx*)> t;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return stop_ || !q_.empty(); });
        if (q_.empty()) break;                       // stop requested and everything posted has run
        t = std::move(q_.front());
        q_.pop_front();
      }
      t(ctx_.get());                                 // tasks see a null context when its creation failed
    }
    if (pool_) pool_->detach();                      // frames the consumer still holds outlive the context safely
    ctx_.reset();                                    // destroyed on the thread that used it
  }
  int device_;
  std::mutex m_;
  std::condition_variable cv_;
  std::deque<std::packaged_task<int(vpcc_ctx*)>> q_;
  bool stop_ = false;
  std::unique_ptr<vpcc_ctx, CtxDeleter> ctx_;
  std::shared_ptr<PinnedPool> pool_;
  int numa_node_ = -1;
  int create_status_ = 0;
  std::thread th_;                                   // last member: starts when everything above exists
};

// The lanes of a Decoder.  They are made at the very beginning of start(): a lane's first act is vpcc_ctx_create, and the
// first HIP call of a process initialises the runtime — 0.19-0.24 s on this pool of machines (profiles/r05/cold_start.txt)
// that now pass while start() reads the input on the caller's thread (src/lib.rs:98), instead of in front of the first frame.
struct Decoder::LaneSet {
  std::vector<std::unique_ptr<Lane>> lanes;
};

void Decoder::start() {
  if (started_) throw std::logic_error("library decoder can only be started once");   // src/lib.rs:108-111
  started_ = true;
  lanes_ = std::make_shared<LaneSet>();
  {
    const size_t G = params_.devices.empty() ? 1 : params_.devices.size();
    for (size_t d = 0; d < G; ++d) lanes_->lanes.push_back(std::make_unique<Lane>(params_.devices.empty() ? 0 : params_.devices[d]));
  }
  const unsigned long nodes = input_node_mask(params_.devices.empty() ? std::vector<int>{0} : params_.devices);
  if (nodes && std::getenv("VPCC_DECODER_TRACE")) std::fprintf(stderr, "[vpcc decoder] input pages interleaved over the NUMA nodes of mask 0x%lx\n", nodes);
  // Bitstream::from_file on the caller's thread (src/lib.rs:98); the reference unwraps the io error
  auto slurp = [nodes](const std::string& path, std::vector<unsigned char>* out, size_t at) {
    std::ifstream in(path, std::ios::binary | std::ios::ate);
    if (!in) throw std::runtime_error("cannot open " + path);
    const size_t n = (size_t)in.tellg();
    const size_t upto = at + ((n + 15) & ~size_t(15));     // next section starts 16-B aligned
    // The buffer is page-locked later and read by the copy engines: ask for huge pages BEFORE its first touch (resize
    // zero-fills) — page-locking then handles 2-MB pages instead of 4-KB ones.  Advice only; VPCC_DECODER_NO_HUGEPAGES=1: none.
    if (upto > out->capacity() && upto >= (size_t(64) << 20) && (nodes || !std::getenv("VPCC_DECODER_NO_HUGEPAGES"))) {
      out->reserve(upto);
      const uintptr_t lo = ((uintptr_t)out->data() + 4095) & ~uintptr_t(4095), hi = ((uintptr_t)out->data() + out->capacity()) & ~uintptr_t(4095);
      if (hi > lo && !std::getenv("VPCC_DECODER_NO_HUGEPAGES")) (void)madvise((void*)lo, (size_t)(hi - lo), MADV_HUGEPAGE);
      // (advice too: MPOL_INTERLEAVE = 3; a refusal leaves the pages where the first touch puts them)
      if (hi > lo && nodes) (void)syscall(SYS_mbind, (void*)lo, (unsigned long)(hi - lo), 3, &nodes, (unsigned long)(8 * sizeof nodes + 1), 0u);
    }
    out->resize(upto);
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

void Decoder::worker() {
  // One lane (thread + context) per GPU; the frames of a unit are dealt round-robin (frame f -> device f % G) and
  // delivered in presentation order — they are independent (src/decoder.rs:186), so no data moves between
  // GPUs.  Plane ingest: the container buffer is page-locked once (portable: every device DMAs from it), so
  // every plane upload is an asynchronous DMA on its context's copy stream, and the next units are uploaded
  // while the current one is reconstructed and drained.
  //
  // A UNIT is what one launch per lane reconstructs: the first GOF alone (the consumer gets its first frame after
  // one GOF's upload), then up to four consecutive GOFs per lane — GOFs are as independent of each other as
  // frames are (a fresh Context per GOF, src/lib.rs:120), and a launch over 128 frames costs 12 % less per frame
  // than four launches over 32.
  const size_t G = params_.devices.empty() ? 1 : params_.devices.size();
  const auto t_worker = std::chrono::steady_clock::now();
  const char* const trace_env = std::getenv("VPCC_DECODER_TRACE");
  const bool trace_steps = trace_env != nullptr && trace_env[0] == '2';
  auto step = [&](const char* what) {                   // VPCC_DECODER_TRACE=2: the start-up, step by step
    if (trace_steps)
      std::fprintf(stderr, "[vpcc decoder] +%.1f ms %s   (@%.1f)\n",
                   std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_worker).count(), what,
                   std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count());
  };
  std::vector<std::unique_ptr<Lane>>& lanes = lanes_->lanes;         // (made by start(), before it read the input)
  auto fail = [&](const std::string& why) { error_ = why; chan_.close_tx(); };
  for (size_t d = 0; d < G; ++d)
    if (lanes[d]->post([](vpcc_ctx* c) { return c ? 0 : 1; }).get()) { fail("vpcc_ctx_create failed: no usable gfx950 device (no CPU fallback)"); return; }
  step("contexts ready");
  // Two homes (vpcc_ctx_reserve): OFF by default here.  A lane's kernels are under 1 % of its wall time (the link sets the
  // pace: 44 ms of planes per 128-frame unit against 0.5 ms of kernels), the pool makes them a tenth faster — and costs
  // 30-50 ms on a GPU whose memory is clean, 0.2 s at the median of 321 recorded reservations on this pool of machines and
  // up to 4.4 s right after another process has freed tens of GB (profiles/r04/pool.txt): round 4 joined the reservation in
  // front of the first launch and a cold Decoder's first frame waited for it.  VPCC_DECODER_POOL_GIB=<n> asks for it: it is
  // then reserved on a thread of its own ONCE THE STREAM IS UNDER WAY (three units launched, the allocations of a steady
  // state made) and only for streams with at least eight units still to come; nothing ever waits for it — gofs created
  // meanwhile allocate as without it — and the search for a second home ends after 100 ms.  The pool is for callers whose
  // planes are in HBM already (bench.py's headline and `fresh_gof` leg: thousands of launches per second).
  struct Reservers {
    std::vector<std::thread> t;
    ~Reservers() { for (auto& x : t) if (x.joinable()) x.join(); }      // before the lanes (and their contexts) go
  } reservers;
  uint64_t pool_gib = 0;
  {
    const char* e = std::getenv("VPCC_DECODER_POOL_GIB");
    pool_gib = e ? (uint64_t)std::strtoull(e, nullptr, 10) : 0u;
  }
  auto reserve_pools = [&] {
    const bool tr = std::getenv("VPCC_DECODER_TRACE") != nullptr;
    for (size_t d = 0; d < G; ++d) {
      vpcc_ctx* c = nullptr;
      lanes[d]->post([&c](vpcc_ctx* x) { c = x; return 0; }).get();
      reservers.t.emplace_back([c, pool_gib, tr] {
        vpcc_pool_info pi{};
        const int st = vpcc_ctx_reserve_within(c, pool_gib << 30, 100.f, &pi);     // no memory for it: the gofs allocate as before
        if (tr)
          std::fprintf(stderr, "[vpcc decoder] pool of %llu GiB: %s, %u kind(s), %llu + %llu GiB, %.1f ms\n", (unsigned long long)pool_gib,
                       st ? vpcc_status_string(st) : "reserved", pi.kinds, (unsigned long long)(pi.bytes_of_kind[0] >> 30),
                       (unsigned long long)(pi.bytes_of_kind[1] >> 30), pi.ms_spent);
      });
    }
  };
  stats_.lanes = (uint32_t)G;
  for (size_t d = 0; d < G && d < 8; ++d) stats_.numa_node[d] = lanes[d]->numa_node();

  // The input is page-locked (portable: every device may DMA from it) before the first unit is launched: 16 ms for 10 GB of
  // huge pages (start() asks for them), 50-300 ms for 4-KB pages.
  // (Page-locking it chunk by chunk on a thread of its own, ahead of the launches, was tried: the first unit's kernels were done
  // after 58 ms instead of 110 — but page-locking holds a lock of the runtime's that every allocation and enqueue waits for, the
  // first FRAME came no sooner and the stream ran at 1 800 frames/s instead of 2 450.)  VPCC_DECODER_PIN_CHUNK_MB page-locks it in
  // chunks of that size instead of in one piece — stretches of planes and planes that cross from one page-locked region into the
  // next are copied piece by piece (vpcc_gof_create) —, for tests.
  struct InputPins {
    vpcc_ctx* ctx = nullptr;
    std::vector<const void*> locked;
    bool lock(const unsigned char* p, size_t bytes, size_t chunk) {
      const uintptr_t lo = (uintptr_t)p & ~uintptr_t(4095), hi = ((uintptr_t)p + bytes + 4095) & ~uintptr_t(4095);   // whole pages
      if (!chunk) chunk = hi - lo;
      const size_t before = locked.size();
      for (uintptr_t at = lo; at < hi; at += chunk) {
        if (vpcc_host_pin(ctx, (const void*)at, (size_t)(std::min<uintptr_t>(at + chunk, hi) - at)) != VPCC_OK) {
          // all of this call or nothing (a half page-locked range is of no use); what earlier calls locked stays: uploads may run from it
          while (locked.size() > before) { (void)vpcc_host_unpin(ctx, locked.back()); locked.pop_back(); }
          return false;
        }
        locked.push_back((const void*)at);
      }
      return true;
    }
    ~InputPins() { for (const void* q : locked) (void)vpcc_host_unpin(ctx, q); }
  } pins;
  lanes[0]->post([&pins](vpcc_ctx* c) { pins.ctx = c; return 0; }).get();
  // Round 5: only the HEAD of the input — the first GOF's planes, when they lie at its start (a container; the raw videos of a
  // V3C stream hold a GOF's planes in three places) — is page-locked in front of the first unit; the rest follows on the lane
  // right behind that unit's launch, while its planes travel: the first frame no longer waits for 16-20 ms of page-locking, and
  // the second unit's upload starts that much earlier (VPCC_DECODER_PIN_AT_ONCE=1: everything first, as before).
  bool pinned = false;
  const unsigned char* rest = nullptr;                // where the part of the input that is page-locked later begins (null: none)
  std::future<int> rest_locked;
  struct WaitRest {                                   // the task works on `pins`: it has run before this scope is left, on every path
    std::future<int>& f;
    ~WaitRest() { if (f.valid()) f.wait(); }
  } wait_rest{rest_locked};
  if (!file_.empty()) {
    const char* e = std::getenv("VPCC_DECODER_PIN_CHUNK_MB");
    const size_t chunk = e ? (size_t)std::strtoull(e, nullptr, 10) << 20 : 0;
    if (!chunk && gofs_.size() > 1 && !std::getenv("VPCC_DECODER_PIN_AT_ONCE")) {
      const unsigned char* end0 = file_.data();
      bool inside = true;
      auto plane = [&](const void* ptr, size_t bytes) {
        const unsigned char* q = (const unsigned char*)ptr;
        inside = inside && q >= file_.data() && q + bytes <= file_.data() + file_.size();
        end0 = std::max(end0, q + bytes);
      };
      for (const vpcc_frame_desc& fr : gofs_[0].frames) {
        plane(fr.occupancy.y, (size_t)fr.occupancy.stride * fr.occupancy.height);
        for (uint32_t m = 0; m < fr.map_count; ++m) {
          plane(fr.geometry[m].y, (size_t)fr.geometry[m].stride * fr.geometry[m].height * 2);
          if (fr.attribute_count) {
            plane(fr.attribute[m].y, (size_t)fr.attribute[m].stride * fr.attribute[m].height * 2);
            plane(fr.attribute[m].u, (size_t)fr.attribute[m].cstride * ((fr.attribute[m].height + 1) / 2) * 2);
            plane(fr.attribute[m].v, (size_t)fr.attribute[m].cstride * ((fr.attribute[m].height + 1) / 2) * 2);
          }
        }
      }
      const uintptr_t cut = ((uintptr_t)end0 + (size_t(2) << 20) - 1) & ~((uintptr_t(2) << 20) - 1);      // a huge-page boundary
      if (inside && cut < (uintptr_t)file_.data() + file_.size() / 4) rest = (const unsigned char*)cut;   // a small head: worth it
    }
    const size_t head = rest ? (size_t)(rest - file_.data()) : file_.size();
    pinned = lanes[0]->post([&, head, chunk](vpcc_ctx*) { return pins.lock(file_.data(), head, chunk) ? 0 : 1; }).get() == 0;
    if (!pinned) rest = nullptr;
  }
  step(rest ? "head of the input page-locked" : "input page-locked");
  struct Part {                                       // one device's share of one unit
    std::vector<vpcc_frame_desc> frames;
    vpcc_gof* g = nullptr;
    std::string err;
    std::future<int> launched;
    double launch_seconds = 0, kernel_seconds = 0;
  };
  struct InFlight {
    std::vector<const vpcc_frame_desc*> frames;       // the unit's frames in presentation order
    std::vector<Part> part;
    vpcc_smoothing_params smooth{};                   // flags != 0: the unit's frames are smoothed behind their reconstruction
  };
  // What the post-processing switches make of a GOF (src/decoder.rs:291-299): geometry smoothing needs the switch AND the
  // SEI (or, for inputs without syntax, the parameters given in Params); colour smoothing its switch and an attribute.
  auto smoothing_of = [&](const DecodedGof& g) {
    vpcc_smoothing_params sp{};
    const bool has_attr = !g.frames.empty() && g.frames[0].attribute_count > 0;
    if (params_.apply_geo_smoothing_type) {
      // (a GOF of a V3C stream without the SEI is left alone whatever Params holds: the reference smooths `if the SEI is present`)
      static const vpcc_smoothing_params none{};
      const vpcc_smoothing_params& src = g.sei_smoothing.flags ? g.sei_smoothing : g.has_syntax ? none : params_.geo_smoothing_without_sei;
      if (src.grid_size >= 2) {
        sp.flags |= VPCC_SMOOTH_GEOMETRY;
        sp.geometry_bitdepth_3d = src.geometry_bitdepth_3d;
        sp.grid_size = src.grid_size;
        sp.threshold = src.threshold;
      }
    }
    if (params_.apply_attr_smoothing_type && has_attr && params_.attr_smoothing.color_grid_size >= 2) {
      sp.flags |= VPCC_SMOOTH_COLOR;
      if (!sp.geometry_bitdepth_3d)
        sp.geometry_bitdepth_3d = params_.attr_smoothing.geometry_bitdepth_3d ? params_.attr_smoothing.geometry_bitdepth_3d
                                : g.sei_smoothing.geometry_bitdepth_3d ? g.sei_smoothing.geometry_bitdepth_3d : 10u;
      sp.color_grid_size = params_.attr_smoothing.color_grid_size;
      sp.color_threshold_smoothing = params_.attr_smoothing.color_threshold_smoothing;
      sp.color_threshold_difference = params_.attr_smoothing.color_threshold_difference;
    }
    return sp;
  };
  auto same_smoothing = [](const vpcc_smoothing_params& a, const vpcc_smoothing_params& b) {
    return std::memcmp(&a, &b, sizeof a) == 0;
  };
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double>(b - a).count();
  };
  // posts the planning + upload + launch of a unit to every lane; returns at once
  auto launch = [&](InFlight* f) {
    f->part.clear();
    f->part.resize(G);
    for (size_t i = 0; i < f->frames.size(); ++i) f->part[i % G].frames.push_back(*f->frames[i]);
    for (size_t d = 0; d < G; ++d) {
      Part* p = &f->part[d];
      if (p->frames.empty()) continue;
      const vpcc_smoothing_params sp = f->smooth;
      p->launched = lanes[d]->post([p, pinned, now, secs, sp](vpcc_ctx* c) {
        const auto t0 = now();
        int st = vpcc_gof_create(c, p->frames.data(), (uint32_t)p->frames.size(), VPCC_MEM_HOST, 0,
                                 (pinned ? VPCC_GOF_ASYNC_UPLOAD : 0u) | VPCC_GOF_PROFILE | (sp.flags ? VPCC_GOF_WANT_PATCH_INDEX : 0u), &p->g);
        if (st == VPCC_OK) st = vpcc_gof_reconstruct(p->g, 0, (uint32_t)p->frames.size(), nullptr);   // ONE launch, asynchronous
        if (st == VPCC_OK && sp.flags) st = vpcc_gof_smooth(p->g, 0, (uint32_t)p->frames.size(), &sp, nullptr);   // src/decoder.rs:291-299
        if (st) p->err = std::string(vpcc_status_string(st)) + ": " + vpcc_last_error(c);
        p->launch_seconds = secs(t0, now());
        return st;
      });
    }
  };
  auto destroy = [&](InFlight* f) {                   // on the lanes, behind whatever they still have queued for it
    for (size_t d = 0; d < G; ++d)
      if (f->part[d].launched.valid() || f->part[d].g) {
        Part* p = &f->part[d];
        if (p->launched.valid()) p->launched.wait();
        lanes[d]->post([p](vpcc_ctx*) { vpcc_gof_destroy(p->g); p->g = nullptr; return 0; }).get();
      }
  };

  // VPCC_DECODER_TRACE=1: where the wall time goes (stderr, one line at the end of the stream).  `launch` is
  // the longest lane's planning + enqueue time per unit, summed over units: it must not grow with G.
  const bool trace = std::getenv("VPCC_DECODER_TRACE") != nullptr;
  double t_counts = 0, t_download = 0, t_send = 0;
  struct Report {
    const bool on; const Stats& s; const double &b, &c, &d;
    ~Report() {
      if (!on) return;
      std::string nodes;
      for (uint32_t i = 0; i < s.lanes && i < 8; ++i) nodes += (i ? "," : "") + std::to_string(s.numa_node[i]);
      std::fprintf(stderr, "[vpcc decoder] %u device(s), lane NUMA nodes [%s]: %llu launch(es) over %llu frames (largest %u), "
                           "kernels %.4f s, launch(plan+enqueue, slowest lane) %.3f s, wait-for-counts %.3f s, "
                           "wait-for-downloads %.3f s, send %.3f s\n", s.lanes, nodes.c_str(), (unsigned long long)s.launches,
                   (unsigned long long)s.frames, s.max_frames_per_launch, s.kernel_seconds, s.launch_seconds, b, c, d);
    }
  } report{trace, stats_, t_counts, t_download, t_send};

  // The units of the stream: [GOF 0], then runs of up to kGofsPerLaunch GOFs PER LANE, each run below ~16 GiB of device memory
  // per lane: a lane's launch covers 128 frames however many lanes there are (with four GOFs per unit regardless, eight lanes would
  // launch over 16 frames each and this thread's work per unit — counts, downloads, hand-over — would come round every 5 ms).
  const size_t kGofsPerLaunch = 4 * G;
  std::vector<std::pair<size_t, size_t>> units;       // [first GOF, one past the last)
  for (size_t k = 0; k < gofs_.size();) {
    size_t end = k + 1;
    uint64_t bytes = 0;
    auto gof_bytes = [&](const DecodedGof& g) {
      uint64_t b = 0;
      for (const vpcc_frame_desc& fr : g.frames)
        b += (uint64_t)fr.map_count * fr.width * fr.height * (4 + 9 + (fr.attribute_count ? 3 : 0));   // planes + output capacity bound
      return b;
    };
    bytes = gof_bytes(gofs_[k]);
    while (k != 0 && end < gofs_.size() && end - k < kGofsPerLaunch) {
      if (!same_smoothing(smoothing_of(gofs_[k]), smoothing_of(gofs_[end]))) break;    // one set of filter parameters per launch
      const uint64_t more = gof_bytes(gofs_[end]);
      if ((bytes + more) / G > (uint64_t(16) << 30)) break;
      bytes += more;
      ++end;
    }
    units.emplace_back(k, end);
    k = end;
  }
  // The results of the stream's LAST unit travel back with nothing left to overlap them with (20 ms for 128 frames): it is
  // dealt out in halves — 4 GOFs as 2 + 1 + 1 — so that only a GOF's worth of downloads is left at the end (units of fewer than
  // 64 frames per lane are left alone: nothing to gain).
  auto frames_of = [&](const std::pair<size_t, size_t>& u) {
    size_t n = 0;
    for (size_t q = u.first; q < u.second; ++q) n += gofs_[q].frames.size();
    return n;
  };
  while (units.size() > 1 && units.back().second - units.back().first > 1 && frames_of(units.back()) >= 64 * G &&
         !std::getenv("VPCC_DECODER_NO_TAIL_SPLIT")) {
    const size_t a = units.back().first, b = units.back().second, mid = a + (b - a + 1) / 2;
    units.back().second = mid;
    units.emplace_back(mid, b);
  }

  // Two units are kept queued behind the one being drained: the copy engines then always have the next
  // upload waiting (with one unit of look-ahead they idled while the current one was downloaded).
  constexpr size_t kAhead = 2;
  std::deque<InFlight> inflight;
  struct Cleanup {                                    // early returns: nothing may outlive the lanes
    std::deque<InFlight>& q; decltype(destroy)& destroy_fn;
    ~Cleanup() { for (auto& f : q) destroy_fn(&f); }
  } cleanup{inflight, destroy};
  size_t launched = 0;
  // Units are posted up to kAhead ahead (their ingest overlaps the current unit's work) — but only once the current
  // unit's point counts are back and its first downloads are posted: a lane serves its queue in order, and with the
  // look-ahead posted first the first unit's counts and downloads waited behind the planning and allocation of the next
  // two units (first frame after 190 ms instead of 110, of which 50-130 are the page-locking of the input).
  auto launch_upto = [&](size_t last) {
    while (launched < units.size() && launched <= last) {
      if (launched == 1 && rest_locked.valid() && rest_locked.get() != 0) pinned = false;   // (the rest could not be page-locked: plain uploads)
      inflight.emplace_back();
      inflight.back().smooth = smoothing_of(gofs_[units[launched].first]);
      for (size_t q = units[launched].first; q < units[launched].second; ++q)
        for (const vpcc_frame_desc& fr : gofs_[q].frames) inflight.back().frames.push_back(&fr);
      launch(&inflight.back());
      ++launched;
    }
  };
  for (size_t k = 0; k < units.size(); ++k) {         // while ssvu.get_v3c_unit_count() > 0, src/lib.rs:118
    if (pool_gib >= 2 && reservers.t.empty() && k == 3 && units.size() - k >= 8) reserve_pools();   // (never waited for)
    launch_upto(k);
    if (k == 0 && rest) {                               // the rest of the input, on the lane, right behind the first unit's launch
      const unsigned char* from = rest;
      const size_t bytes = (size_t)(file_.data() + file_.size() - rest);
      rest_locked = lanes[0]->post([&pins, from, bytes](vpcc_ctx*) { return pins.lock(from, bytes, 0) ? 0 : 1; });
    }
    InFlight& cur = inflight.front();
    const size_t n = cur.frames.size();
    double slowest = 0;
    std::string first_err;
    for (size_t d = 0; d < G; ++d) {
      if (!cur.part[d].launched.valid()) continue;
      const int st = cur.part[d].launched.get();
      slowest = std::max(slowest, cur.part[d].launch_seconds);
      if (st && first_err.empty()) first_err = cur.part[d].err;
      if (!st) {
        stats_.launches += 1;
        stats_.frames += cur.part[d].frames.size();
        stats_.max_frames_per_launch = std::max<uint32_t>(stats_.max_frames_per_launch, (uint32_t)cur.part[d].frames.size());
      }
    }
    // reference: a panic in the worker -> the consumer sees end-of-stream where this GOF would have started
    if (!first_err.empty()) { fail(first_err); return; }
    if (k == 0) step("first unit planned, uploads and launch enqueued");
    stats_.launch_seconds += slowest;
    std::vector<std::vector<uint32_t>> counts(G);
    {
      const auto t0 = now();
      std::vector<std::future<int>> fc(G);
      std::vector<std::string> errs(G);
      for (size_t d = 0; d < G; ++d) {
        if (!cur.part[d].g) continue;
        counts[d].resize(cur.part[d].frames.size());
        Part* p = &cur.part[d];
        uint32_t* out = counts[d].data();
        std::string* e = &errs[d];
        fc[d] = lanes[d]->post([p, out, e](vpcc_ctx* c) {
          int st = vpcc_gof_point_counts(p->g, out);            // waits for the launch
          if (st) *e = vpcc_last_error(c);
          const char* names[8];
          float ms[8];
          const int nk = st ? 0 : vpcc_gof_kernel_times(p->g, names, ms, 8);
          for (int q = 0; q < nk; ++q) p->kernel_seconds += ms[q] * 1e-3;
          return st;
        });
      }
      // every lane's task writes into `counts` and `errs`: all of them have finished before either goes away
      int bad = -1;
      for (size_t d = 0; d < G; ++d)
        if (fc[d].valid() && fc[d].get() && bad < 0) bad = (int)d;
      if (bad >= 0) { fail(errs[(size_t)bad]); return; }
      for (size_t d = 0; d < G; ++d) stats_.kernel_seconds += cur.part[d].kernel_seconds;
      t_counts += secs(t0, now());
      if (k == 0) step("first unit reconstructed (point counts back)");
      if (trace_steps && k) { char b[64]; std::snprintf(b, sizeof b, "unit %zu (%zu frames): point counts back", k, n); step(b); }
    }
    // Downloads are posted a window of frames ahead of the hand-over — every lane works through its own
    // frames while earlier ones are delivered in presentation order (src/decoder.rs:188) — but not the whole
    // unit at once: the page-locked result blocks come from a small pool per lane (allocated by the lane's
    // thread, on its NUMA node) and are recycled as the consumer drops frames (allocating and freeing pinned
    // memory costs milliseconds).
    const size_t window = std::max<size_t>(8, 4 * G);
    std::vector<PointSet3> sets(n);
    std::vector<std::future<int>> done(n);
    std::vector<std::string> derr(n);
    size_t posted = 0;
    auto post_downloads = [&](size_t upto) {
      for (; posted < n && posted < upto; ++posted) {
        const size_t f = posted, d = f % G, local = f / G;
        PointSet3* ps = &sets[f];
        ps->with_colors = cur.frames[f]->attribute_count > 0;
        const size_t np = counts[d][local];
        Part* p = &cur.part[d];
        PinnedPool* pool = lanes[d]->pool().get();
        std::string* e = &derr[f];
        done[f] = lanes[d]->post([p, local, ps, np, pool, e](vpcc_ctx* c) {
          PinnedBlock bx = pool->get(np * sizeof(vpcc_point3)), bc;
          if (ps->with_colors) bc = pool->get(np * sizeof(vpcc_color3));
          if (!bx.ptr || (ps->with_colors && !bc.ptr)) { *e = "out of page-locked memory"; return (int)VPCC_ERR_DEVICE; }
          ps->positions.adopt(std::move(bx), np);
          if (ps->with_colors) ps->colors.adopt(std::move(bc), np);
          size_t got = 0;
          // asynchronous: the copies of a window of frames queue up on the download stream; the worker waits for a
          // frame's completion event itself (vpcc_gof_download_wait below), the lane goes on
          const int st = vpcc_gof_download_async(p->g, (uint32_t)local, ps->positions.data(), ps->with_colors ? ps->colors.data() : nullptr,
                                                 nullptr, np ? np : 1, &got);
          if (st || got != np) { *e = st ? vpcc_last_error(c) : "point count changed between calls"; return st ? st : (int)VPCC_ERR_DEVICE; }
          return 0;
        });
      }
    };
    auto settle = [&](size_t from) {                       // downloads in flight write into `sets`
      for (size_t q = from; q < posted; ++q)
        if (done[q].get() == VPCC_OK) (void)vpcc_gof_download_wait(cur.part[q % G].g, (uint32_t)(q / G));
    };
    post_downloads(window);                                // this unit's first frames before the lanes hear of the next units
    launch_upto(k + kAhead);
    for (size_t f = 0; f < n; ++f) {
      post_downloads(f + window);
      const auto t0 = now();
      int st = done[f].get();
      if (!st && vpcc_gof_download_wait(cur.part[f % G].g, (uint32_t)(f / G)) != VPCC_OK) { st = VPCC_ERR_DEVICE; derr[f] = "download failed"; }
      const auto t1 = now();
      t_download += secs(t0, t1);
      bool sent = false;
      if (!st) {
        sent = chan_.send(std::move(sets[f]));
        t_send += secs(t1, now());
        if (k == 0 && f == 0) step("first frame handed over");
        if (trace_steps && k && (f == 0 || f + 1 == n)) { char b[64]; std::snprintf(b, sizeof b, "unit %zu: frame %zu handed over", k, f); step(b); }
      }
      if (st || !sent) {                                   // device error, or receiver dropped (src/decoder.rs:311-313)
        settle(f + 1);
        if (st) error_ = derr[f];
        chan_.close_tx();
        return;
      }
    }
    destroy(&cur);
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
  // ASCII: every value is a u16 or a u8 — its decimal digits plus one separator are looked up as ONE 8-byte word (65 536 entries,
  // 512 KB; coordinates of a frame cluster in a few hundred of them) and stored unaligned; the last separator of a line becomes
  // the line feed.  An 800 000-point frame with colours: 17-24 ms with the file write (six snprintf conversions per point: 840).
  struct Digits {
    uint64_t word[65536];                         // low bytes: the digits and a blank; top byte: how many of them
    Digits() {
      for (uint32_t v = 0; v < 65536; ++v) {
        char t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const int k = std::snprintf(t, sizeof t, "%u ", (unsigned)v);      // at most "65535 "
        uint64_t w = 0;
        std::memcpy(&w, t, 8);
        word[v] = (w & 0x00FFFFFFFFFFFFFFull) | ((uint64_t)k << 56);
      }
    }
  };
  static const Digits* const digits = new Digits();                         // (never freed: the process's)
  const size_t at = s.size(), per_point = rgb ? 6 * 3 + 4 * 3 : 6 * 3;      // "65535 " x 3 (+ "255 " x 3)
  s.resize(at + n * per_point + 8);                                          // + 8: the last unaligned store
  char* o = &s[at];
  auto put = [&](uint32_t v) {
    const uint64_t w = digits->word[v];
    std::memcpy(o, &w, 8);
    o += w >> 56;
  };
  for (size_t i = 0; i < n; ++i) {
    put(xyz[i].x); put(xyz[i].y); put(xyz[i].z);
    if (rgb) { put(rgb[i].r); put(rgb[i].g); put(rgb[i].b); }
    o[-1] = '\n';
  }
  s.resize((size_t)(o - s.data()));
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
  tmc2rs::Params params;                    // may still change between open and start (vpcc_decoder_set_smoothing)
  std::unique_ptr<tmc2rs::Decoder> made;    // Decoder::new happens at vpcc_decoder_start
  bool started = false;
  std::optional<tmc2rs::PointSet3> cur;
  std::string err;
  double first_frame_seconds = 0;           // vpcc_decoder_drain: start-up latency (contexts, page-locking, first GOF)
  explicit vpcc_decoder(tmc2rs::Params p) : params(std::move(p)) {}
  tmc2rs::Decoder& dec_ref() {
    if (!made) made = std::make_unique<tmc2rs::Decoder>(params);
    return *made;
  }
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

extern "C" int vpcc_decoder_set_smoothing(vpcc_decoder* d, int apply_geo_smoothing, int apply_attr_smoothing,
                                          const vpcc_smoothing_params* params) {
  if (!d) return VPCC_ERR_INVALID_ARG;
  if (d->started) { d->err = "vpcc_decoder_set_smoothing after vpcc_decoder_start"; return VPCC_ERR_STATE; }
  d->params.apply_geo_smoothing_type = apply_geo_smoothing != 0;
  d->params.apply_attr_smoothing_type = apply_attr_smoothing != 0;
  if (params) {
    d->params.attr_smoothing = *params;
    d->params.geo_smoothing_without_sei = *params;
  }
  return VPCC_OK;
}

extern "C" int vpcc_decoder_start(vpcc_decoder* d) {
  if (!d) return VPCC_ERR_INVALID_ARG;
  try {
    d->started = true;
    d->dec_ref().start();
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
  d->cur = d->dec_ref().recv_frame();
  if (!d->cur) return 0;                      // None: end of stream (or the worker failed: vpcc_decoder_error)
  *n_points = d->cur->len();
  if (xyz) *xyz = d->cur->positions.data();
  if (rgb) *rgb = d->cur->with_colors ? d->cur->colors.data() : nullptr;
  return 1;
}

extern "C" const char* vpcc_decoder_error(vpcc_decoder* d) {
  if (!d) return "";
  return d->dec_ref().last_error().empty() ? d->err.c_str() : d->dec_ref().last_error().c_str();   // the worker's error wins
}

extern "C" int vpcc_decoder_drain(vpcc_decoder* d, uint64_t* frames, uint64_t* points, double* seconds) {
  if (!d) return VPCC_ERR_INVALID_ARG;
  const auto t0 = std::chrono::steady_clock::now();
  uint64_t nf = 0, np = 0;
  while (auto fr = d->dec_ref().recv_frame()) {
    if (nf == 0) d->first_frame_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    ++nf;
    np += fr->len();
  }
  const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (frames) *frames = nf;
  if (points) *points = np;
  if (seconds) *seconds = s;
  return d->dec_ref().last_error().empty() ? VPCC_OK : VPCC_ERR_DEVICE;
}

extern "C" double vpcc_decoder_first_frame_seconds(const vpcc_decoder* d) { return d ? d->first_frame_seconds : 0.0; }

extern "C" int vpcc_decoder_stats(const vpcc_decoder* d, vpcc_decoder_stats_t* out) {
  if (!d || !out) return VPCC_ERR_INVALID_ARG;
  *out = d->made ? d->made->stats() : vpcc_decoder_stats_t{};
  return VPCC_OK;
}

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
