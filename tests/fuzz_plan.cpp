// Robustness harness for the host-side frame validation and planning (vpcc_host.cpp), CPU only, built with
// -fsanitize=address,undefined by tests/test_plan_fuzz.py: random and adversarial patch tables must either be
// rejected by validate_frame or planned into work lists that stay inside the canvas.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "vpcc_host.hpp"

static uint64_t st = 0x243F6A8885A308D3ull;
static uint64_t rnd() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; }
static uint32_t below(uint32_t n) { return n ? (uint32_t)(rnd() % n) : 0; }

int main(int argc, char** argv) {
  const long iterations = argc > 1 ? std::atol(argv[1]) : 1000;
  static uint8_t dummy8[16];
  static uint16_t dummy16[16];
  long accepted = 0, rejected = 0, items = 0;
  for (long it = 0; it < iterations; ++it) {
    vpcc_frame_desc f{};
    const uint32_t R = 1u << (3 + below(3)), prec = 1u << below(3);
    f.occupancy_resolution = R;
    f.occupancy_precision = prec;
    f.width = R * (1 + below(40));
    f.height = R * (1 + below(40));
    if (below(50) == 0) f.width += below(R);                       // ragged canvas
    f.map_count = 1 + below(2);
    f.absolute_d1 = below(2);
    f.attribute_count = below(2);
    f.occupancy = vpcc_image_u8{dummy8, f.width / prec, f.height / prec, f.width / prec};
    for (uint32_t m = 0; m < f.map_count; ++m) {
      f.geometry[m] = vpcc_image_u16{dummy16, nullptr, nullptr, f.width, f.height, f.width, f.width / 2};
      f.attribute[m] = vpcc_image_u16{dummy16, dummy16, dummy16, f.width, f.height, f.width, f.width / 2};
    }
    std::vector<vpcc_patch> patches(below(40));
    const uint32_t bw = f.width / R, bh = f.height / R;
    const bool wild = below(4) == 0;
    for (vpcc_patch& p : patches) {
      p = vpcc_patch{};
      p.orientation = (uint8_t)below(wild ? 12 : 2);
      p.size_u0 = 1 + below(wild ? 70000 : 6);
      p.size_v0 = 1 + below(wild ? 70000 : 6);
      p.u0 = below(wild ? 0xFFFFFFFFu : bw + 1);
      p.v0 = below(wild ? 0xFFFFFFFFu : bh + 1);
      p.u1 = (uint32_t)rnd(); p.v1 = (uint32_t)rnd(); p.d1 = (uint32_t)rnd();
      p.lod_x = wild ? (uint32_t)rnd() : 1; p.lod_y = wild ? below(3) : 1;
      p.normal_axis = (uint8_t)below(wild ? 5 : 3); p.tangent_axis = (uint8_t)below(3); p.bitangent_axis = (uint8_t)below(3);
      p.projection_mode = (uint8_t)below(wild ? 3 : 2);
      p.axis_of_additional_plane = wild ? (uint8_t)below(2) : 0;
    }
    f.patches = patches.empty() ? nullptr : patches.data();
    f.patch_count = (uint32_t)patches.size();
    if (vpcc::validate_frame(&f) != VPCC_OK) { ++rejected; continue; }
    ++accepted;
    vpcc::FramePlan plan;
    vpcc::plan_frame(f, &plan);
    uint64_t expect_vb = 0;
    for (const vpcc_patch& p : patches) expect_vb += (uint64_t)p.size_u0 * p.size_v0;
    if (plan.vblocks.size() != expect_vb) { std::fprintf(stderr, "vblock count\n"); return 1; }
    for (const vpcc::VBlock& b : plan.vblocks)
      if (b.canvas_block >= (uint64_t)plan.bw * plan.bh) { std::fprintf(stderr, "canvas block out of range\n"); return 1; }
    // the tile kernel's items are completed on the device (k_plan_items) from one template per patch and the virtual
    // blocks checked above: the host's part is the templates and the bound that sizes the device arrays
    if (plan.tile_eligible) {
      if (plan.patch_items.size() != patches.size()) { std::fprintf(stderr, "item templates\n"); return 1; }
      if (plan.tile_bound > plan.vblocks.size() || plan.tile_bound > (uint64_t)plan.bw * plan.bh) { std::fprintf(stderr, "tile bound\n"); return 1; }
      for (const vpcc::VBlock& b : plan.vblocks)
        if ((b.canvas_block % plan.bw) * 16u + 16u > f.width || (b.canvas_block / plan.bw) * 16u + 16u > f.height) { std::fprintf(stderr, "tile outside the canvas\n"); return 1; }
      items += plan.tile_bound;
    }
  }
  // Shares of the resident workgroups per frame (plan_tile_launch): every frame with tiles gets at least one
  // workgroup, the table's per-slot share equals the number of slots the frame really has (the kernel re-arms a
  // frame's ticket counter after exactly that many workgroups have left it), no slot names a frame outside the launch.
  long maps = 0, equal_split = 0;
  for (long it = 0; it < iterations; ++it) {
    const uint32_t count = 1 + below(below(8) == 0 ? 2100 : 70);
    std::vector<uint32_t> tiles(count);
    const uint32_t scale = 1 + below(9000);
    for (uint32_t& t : tiles) t = below(6) == 0 ? 0 : below(scale) + (below(3) == 0 ? below(40) : 0);
    const uint32_t resident = below(10) == 0 ? below(300) : 128, depth = 1 + below(6);
    vpcc::TileLaunchMap map;
    vpcc::plan_tile_launch(tiles.data(), count, resident, depth, map);
    if (!map.slots) { ++equal_split; continue; }
    ++maps;
    if (map.slots > vpcc::kTileMapSlots) { std::fprintf(stderr, "slots\n"); return 1; }
    for (uint32_t x = 0; x < 8; ++x) {
      std::vector<uint32_t> have((count + 7) / 8, 0);
      for (uint32_t s = 0; s < vpcc::kTileMapSlots; ++s) {
        const uint32_t v = map.frame_of_slot[x][s];
        if (v == 0xFF) continue;
        if (s >= map.slots || x + 8u * v >= count) { std::fprintf(stderr, "slot names a frame outside the launch\n"); return 1; }
        ++have[v];
      }
      for (uint32_t s = 0; s < map.slots; ++s) {
        const uint32_t v = map.frame_of_slot[x][s];
        if (v != 0xFF && map.wgs_of_slot[x][s] != have[v]) { std::fprintf(stderr, "share != slots of the frame\n"); return 1; }
      }
      for (uint32_t i = x; i < count; i += 8)
        if ((tiles[i] != 0) != (have[i / 8] != 0)) { std::fprintf(stderr, "frame %u tiles %u workgroups %u\n", i, tiles[i], have[i / 8]); return 1; }
    }
  }
  // The free space of a context's pool (vpcc::PoolExtents, behind vpcc_ctx_reserve): random runs of two kinds, random takes and
  // returns.  Blocks never overlap, lie inside one run of the kind asked for, the byte accounts add up, and once everything is
  // back every run is ONE free extent again (coalescing never crosses a run).
  long pool_ops = 0;
  for (long it = 0; it < iterations / 8 + 4; ++it) {
    vpcc::PoolExtents E;
    static char arena[1];                                     // addresses only: nothing is dereferenced
    char* base = arena;
    const size_t G = 1u << 20;
    size_t at = 0, total[2] = {0, 0};
    const uint32_t n_runs = 1 + below(6);
    for (uint32_t r = 0; r < n_runs; ++r) {
      const size_t len = (1 + below(48)) * G;
      const int kind = (int)below(2);
      E.add_run(base + at, len, kind);
      total[kind] += len;
      at += len + (below(3) == 0 ? (1 + below(4)) * G : 0);    // sometimes a gap: another slab
    }
    if (E.in_use[0] != 0 || E.in_use[1] != 0) { std::fprintf(stderr, "pool: fresh runs in use\n"); return 1; }
    struct Held { char* p; size_t bytes; uint32_t run; };
    std::vector<Held> held;
    for (int op = 0; op < 200; ++op, ++pool_ops) {
      if (held.empty() || below(3) != 0) {
        const int kind = (int)below(2);
        const size_t bytes = (1 + below(below(4) == 0 ? 40 * 1024 : 4096)) * 1024;
        char* p = nullptr; uint32_t run = 0;
        if (!E.take(kind, bytes, &p, &run)) continue;
        const vpcc::PoolExtents::Run& R = E.runs[run];
        if (R.kind != kind || p < R.ptr || p + bytes > R.ptr + R.bytes) { std::fprintf(stderr, "pool: block outside a run of its kind\n"); return 1; }
        for (const Held& h : held)
          if (p < h.p + h.bytes && h.p < p + bytes) { std::fprintf(stderr, "pool: blocks overlap\n"); return 1; }
        held.push_back(Held{p, bytes, run});
      } else {
        const size_t k = below((uint32_t)held.size());
        E.give_back(held[k].run, held[k].p, held[k].bytes);
        held.erase(held.begin() + k);
      }
      size_t used[2] = {0, 0}, freeb[2] = {0, 0};
      for (const Held& h : held) used[E.runs[h.run].kind] += h.bytes;
      for (int kd = 0; kd < 2; ++kd) {
        for (const auto& x : E.free_[kd]) freeb[kd] += x.bytes;
        if (used[kd] != E.in_use[kd] || used[kd] + freeb[kd] != total[kd]) { std::fprintf(stderr, "pool: byte accounts\n"); return 1; }
      }
    }
    for (const Held& h : held) E.give_back(h.run, h.p, h.bytes);
    size_t extents = 0;
    for (int kd = 0; kd < 2; ++kd) extents += E.free_[kd].size();
    if (extents != E.runs.size() || E.in_use[0] || E.in_use[1]) { std::fprintf(stderr, "pool: not whole again (%zu extents, %zu runs)\n", extents, E.runs.size()); return 1; }
  }
  std::printf("launch maps %ld, equal split %ld, pool operations %ld\n", maps, equal_split, pool_ops);
  std::printf("iterations %ld accepted %ld rejected %ld tile items %ld\n", iterations, accepted, rejected, items);
  return accepted > 0 && rejected > 0 ? 0 : 1;
}
