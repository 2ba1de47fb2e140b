// Robustness harness for the V3C syntax parser (host code, CPU only): mutates a valid sample stream and
// parses every GOF.  Built by tests/test_v3c_fuzz.py with -fsanitize=address,undefined: any out-of-bounds
// access, overflow-dependent behaviour or uncaught exception fails the test; a SyntaxError is the expected
// way for a malformed stream to end.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <vector>

#include "bitstream.hpp"
#include "v3c_syntax.hpp"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
  rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
  return rng_state;
}

static int parse_all(const std::vector<uint8_t>& data, long* gofs, long* frames) {
  std::vector<tmc2rs::V3CUnit> units;
  try {
    tmc2rs::Bitstream bs(data);
    size_t header = 0;
    units = tmc2rs::split_sample_stream(bs, &header);
  } catch (const std::exception&) {
    return 1;
  }
  size_t next = 0;
  while (next < units.size()) {
    tmc2rs::GofSyntax g;
    try {
      next = tmc2rs::parse_gof(units, next, &g);
      const tmc2rs::GofParams p = tmc2rs::build_gof_params(g);
      const std::vector<tmc2rs::PatchFrame> f = tmc2rs::build_patch_frames(g);
      (void)p;
      ++*gofs;
      *frames += (long)f.size();
      for (const tmc2rs::VideoSubstream& v : g.videos) {       // NAL length prefixes -> Annex-B start codes
        try {
          (void)tmc2rs::sample_stream_to_bytestream(v.data, tmc2rs::CodecId::H265, 4);
        } catch (const std::exception&) {
        }
      }
    } catch (const tmc2rs::SyntaxError&) {
      return 2;
    }
  }
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 3) return 64;
  std::ifstream in(argv[1], std::ios::binary);
  const std::vector<uint8_t> seed((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
  const long iterations = std::atol(argv[2]);
  long gofs = 0, frames = 0, ok = 0, rejected = 0;
  if (parse_all(seed, &gofs, &frames) != 0) { std::fprintf(stderr, "seed stream does not parse\n"); return 2; }
  for (long it = 0; it < iterations; ++it) {
    std::vector<uint8_t> d = seed;
    const int kind = (int)(rnd() % 5);
    const int n = 1 + (int)(rnd() % 4);
    for (int k = 0; k < n; ++k) {
      const size_t pos = (size_t)(rnd() % d.size());
      if (kind == 0) d[pos] ^= (uint8_t)(1u << (rnd() % 8));            // bit flip
      else if (kind == 1) d[pos] = (uint8_t)rnd();                       // random byte
      else if (kind == 2) d[pos] = (rnd() & 1) ? 0xFF : 0x00;            // saturate
      else if (kind == 3) { d.resize(1 + pos); break; }                  // truncate
      else d.insert(d.begin() + (long)pos, (uint8_t)rnd());              // insert
    }
    const int st = parse_all(d, &gofs, &frames);
    if (st == 0) ++ok; else ++rejected;
  }
  std::printf("iterations %ld parsed %ld rejected %ld (gofs %ld frames %ld)\n", iterations, ok, rejected, gofs, frames);
  return 0;
}
