// Robustness harness for the decoded-GOF container reader (tmc2-rs_amd/csrc/decoder_input.cpp, host code, CPU
// only): mutates a valid .vpccgof image, parses it, and — when the parser accepts it — reads every byte
// range the runtime would upload (vpcc_runtime.hip::gof_create_impl: tight planes, chroma_elems for U/V) and
// runs the frame validation and planning on it.  Built by tests/test_container_fuzz.py with
// -fsanitize=address,undefined: an accepted container whose planes reach outside the file, an arithmetic
// wrap or an uncaught exception fails the test.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "decoder.hpp"
#include "vpcc_host.hpp"

static uint64_t rng_state = 0xD1B54A32D192ED03ull;
static uint64_t rnd() {
  rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
  return rng_state;
}

static volatile unsigned sink;
static void touch(const void* p, size_t bytes) {          // ASan checks both ends and a stride through the range
  const unsigned char* q = static_cast<const unsigned char*>(p);
  if (!bytes) return;
  unsigned s = q[0] + q[bytes - 1];
  for (size_t i = 0; i < bytes; i += 4096) s += q[i];
  sink = s;
}

static int parse_and_walk(const std::vector<unsigned char>& buf, long* frames, long* valid) {
  std::vector<tmc2rs::DecodedGof> gofs;
  std::string err;
  if (!tmc2rs::parse_container(buf, &gofs, &err)) return 1;
  for (const tmc2rs::DecodedGof& g : gofs)
    for (const vpcc_frame_desc& d : g.frames) {
      ++*frames;
      touch(d.patches, sizeof(vpcc_patch) * (size_t)d.patch_count);
      touch(d.occupancy.y, (size_t)d.occupancy.width * d.occupancy.height);
      for (uint32_t m = 0; m < d.map_count; ++m) {
        touch(d.geometry[m].y, (size_t)d.geometry[m].width * d.geometry[m].height * 2);
        if (d.attribute_count) {
          touch(d.attribute[m].y, (size_t)d.attribute[m].width * d.attribute[m].height * 2);
          touch(d.attribute[m].u, vpcc::chroma_elems(d.attribute[m]) * 2);
          touch(d.attribute[m].v, vpcc::chroma_elems(d.attribute[m]) * 2);
        }
      }
      vpcc::FrameShape shape;
      if (vpcc::validate_frame(&d, &shape) == VPCC_OK) {
        ++*valid;
        std::vector<uint32_t> vb_base(shape.n_patches + 1);
        std::vector<vpcc::TileItem> items(shape.n_patches);
        std::vector<vpcc::DevPatch> patches(shape.n_patches);
        vpcc::write_frame_records(d, vb_base.data(), shape.tile_eligible ? items.data() : nullptr, patches.data());
      }
    }
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 3) return 64;
  std::ifstream in(argv[1], std::ios::binary);
  const std::vector<unsigned char> seed((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
  const long iterations = std::atol(argv[2]);
  long frames = 0, valid = 0, ok = 0, rejected = 0;
  if (parse_and_walk(seed, &frames, &valid) != 0 || valid == 0) { std::fprintf(stderr, "seed container does not parse\n"); return 2; }
  for (long it = 0; it < iterations; ++it) {
    std::vector<unsigned char> d = seed;
    const int kind = (int)(rnd() % 6);
    const int n = 1 + (int)(rnd() % 3);
    for (int k = 0; k < n; ++k) {
      // headers are where the damage is interesting: bias positions towards the first bytes of the file and of frames
      size_t pos = (size_t)(rnd() % d.size());
      if (rnd() % 4) pos = (size_t)(rnd() % (d.size() < 256 ? d.size() : 256));
      if (kind == 0) d[pos] ^= (unsigned char)(1u << (rnd() % 8));
      else if (kind == 1) d[pos] = (unsigned char)rnd();
      else if (kind == 2) d[pos] = (rnd() & 1) ? 0xFF : 0x00;
      else if (kind == 3) { d.resize(1 + pos); break; }
      else if (kind == 4) {                                          // a whole header word set to a large value
        const size_t w = pos & ~size_t(3);
        const uint32_t big[4] = {0xFFFFFFFFu, 0x80000000u, 0x00010001u, 32769u};
        if (w + 4 <= d.size()) std::memcpy(&d[w], &big[rnd() % 4], 4);
      } else d.insert(d.begin() + (long)pos, (unsigned char)rnd());
    }
    const int st = parse_and_walk(d, &frames, &valid);
    if (st == 0) ++ok; else ++rejected;
  }
  std::printf("iterations %ld parsed %ld rejected %ld (frames %ld valid %ld)\n", iterations, ok, rejected, frames, valid);
  return 0;
}
