/* vpcc_smoothing_spec.c — CPU specification of the smoothing filters (see vpcc_smoothing_spec.h).
 * TEST INFRASTRUCTURE ONLY.  Not derived from the reference (which has none). */
#include "vpcc_smoothing_spec.h"

#include <stdlib.h>
#include <string.h>

typedef struct { uint32_t count, s[3], minp, maxp; } cell_t;

static uint32_t cell_coord(uint32_t p, uint32_t G, uint32_t w) { uint32_t q = p / G; return q < w ? q : w - 1; }

static cell_t* build_cells(const vpcc_point3* xyz, const uint32_t* val3 /* 3 per point or NULL -> xyz */,
                           const uint16_t* patch, size_t n, uint32_t G, uint32_t w) {
  cell_t* cells = (cell_t*)calloc((size_t)w * w * w, sizeof(cell_t));
  if (!cells) return NULL;
  for (size_t i = 0; i < n; ++i) {
    const uint32_t p[3] = {xyz[i].x, xyz[i].y, xyz[i].z};
    cell_t* c = &cells[((size_t)cell_coord(p[2], G, w) * w + cell_coord(p[1], G, w)) * w + cell_coord(p[0], G, w)];
    if (c->count == 0) { c->minp = patch[i]; c->maxp = patch[i]; }
    if (patch[i] < c->minp) c->minp = patch[i];
    if (patch[i] > c->maxp) c->maxp = patch[i];
    c->count++;
    for (int a = 0; a < 3; ++a) c->s[a] += val3 ? val3[3 * i + a] : p[a];
  }
  return cells;
}

/* neighbourhood of one point: lower cell and weights per axis */
static void axis_setup(uint32_t p, uint32_t G, uint32_t w, int64_t* s, int64_t wt[2]) {
  const int64_t q = cell_coord(p, G, w), r = (int64_t)p - q * (int64_t)G, h = G / 2;
  *s = (r < h) ? q - 1 : q;
  const int64_t t = 2 * ((int64_t)p - (*s * (int64_t)G + h)) + 1;
  wt[0] = 2 * (int64_t)G - t;
  wt[1] = t;
}

int vpcc_spec_smooth_geometry(vpcc_point3* xyz, const uint16_t* patch, size_t n, uint32_t bitdepth, uint32_t G,
                              uint32_t T) {
  if (!xyz || !patch || G < 2 || bitdepth < 1 || bitdepth > 16) return VPCC_ERR_INVALID_ARG;
  const uint32_t w = ((1u << bitdepth) + G - 1) / G;
  cell_t* cells = build_cells(xyz, NULL, patch, n, G, w);
  if (!cells) return VPCC_ERR_INVALID_ARG;
  for (size_t i = 0; i < n; ++i) {
    const uint32_t p[3] = {xyz[i].x, xyz[i].y, xyz[i].z};
    int64_t s[3], wt[3][2];
    for (int a = 0; a < 3; ++a) axis_setup(p[a], G, w, &s[a], wt[a]);
    int64_t num[3] = {0, 0, 0}, den = 0;
    int mixed = 0;
    for (int dz = 0; dz < 2; ++dz) for (int dy = 0; dy < 2; ++dy) for (int dx = 0; dx < 2; ++dx) {
      const int64_t cx = s[0] + dx, cy = s[1] + dy, cz = s[2] + dz;
      if (cx < 0 || cy < 0 || cz < 0 || cx >= w || cy >= w || cz >= w) continue;
      const cell_t* c = &cells[((size_t)cz * w + cy) * w + cx];
      if (!c->count) continue;
      const int64_t W = wt[0][dx] * wt[1][dy] * wt[2][dz];
      for (int a = 0; a < 3; ++a) num[a] += W * c->s[a];
      den += W * c->count;
      mixed |= c->minp != c->maxp;
    }
    if (!mixed || den <= 0) continue;
    int64_t C[3], d2 = 0;
    for (int a = 0; a < 3; ++a) {
      C[a] = (16 * num[a] + den / 2) / den;
      const int64_t d = 16 * (int64_t)p[a] - C[a];
      d2 += d * d;
    }
    if (d2 > 256 * (int64_t)T * T) {
      uint16_t o[3];
      for (int a = 0; a < 3; ++a) { int64_t v = (C[a] + 8) >> 4; o[a] = (uint16_t)(v > 65535 ? 65535 : v); }
      xyz[i].x = o[0]; xyz[i].y = o[1]; xyz[i].z = o[2];
    }
  }
  free(cells);
  return VPCC_OK;
}

int vpcc_spec_smooth_color(const vpcc_point3* xyz, vpcc_color3* rgb, const uint16_t* patch, size_t n, uint32_t bitdepth,
                           uint32_t G, uint32_t Ts, uint32_t Td) {
  if (!xyz || !rgb || !patch || G < 2 || bitdepth < 1 || bitdepth > 16) return VPCC_ERR_INVALID_ARG;
  const uint32_t w = ((1u << bitdepth) + G - 1) / G;
  uint32_t* col = (uint32_t*)malloc(sizeof(uint32_t) * 3 * (n ? n : 1));
  if (!col) return VPCC_ERR_INVALID_ARG;
  for (size_t i = 0; i < n; ++i) { col[3 * i] = rgb[i].r; col[3 * i + 1] = rgb[i].g; col[3 * i + 2] = rgb[i].b; }
  cell_t* cells = build_cells(xyz, col, patch, n, G, w);
  if (!cells) { free(col); return VPCC_ERR_INVALID_ARG; }
  for (size_t i = 0; i < n; ++i) {
    const uint32_t p[3] = {xyz[i].x, xyz[i].y, xyz[i].z};
    int64_t s[3], wt[3][2];
    for (int a = 0; a < 3; ++a) axis_setup(p[a], G, w, &s[a], wt[a]);
    const cell_t* cc = &cells[((size_t)cell_coord(p[2], G, w) * w + cell_coord(p[1], G, w)) * w + cell_coord(p[0], G, w)];
    int64_t mc[3];
    for (int a = 0; a < 3; ++a) mc[a] = cc->s[a] / cc->count;
    int64_t num[3] = {0, 0, 0}, den = 0;
    int mixed = 0;
    for (int dz = 0; dz < 2; ++dz) for (int dy = 0; dy < 2; ++dy) for (int dx = 0; dx < 2; ++dx) {
      const int64_t cx = s[0] + dx, cy = s[1] + dy, cz = s[2] + dz;
      if (cx < 0 || cy < 0 || cz < 0 || cx >= w || cy >= w || cz >= w) continue;
      const cell_t* c = &cells[((size_t)cz * w + cy) * w + cx];
      if (!c->count) continue;
      if (c != cc) {
        int64_t diff = 0;
        for (int a = 0; a < 3; ++a) { int64_t m = c->s[a] / c->count - mc[a]; diff += m < 0 ? -m : m; }
        if (diff > Td) continue;
      }
      const int64_t W = wt[0][dx] * wt[1][dy] * wt[2][dz];
      for (int a = 0; a < 3; ++a) num[a] += W * c->s[a];
      den += W * c->count;
      mixed |= c->minp != c->maxp;
    }
    if (!mixed || den <= 0) continue;
    int64_t m[3], dist = 0;
    for (int a = 0; a < 3; ++a) {
      m[a] = (2 * num[a] + den) / (2 * den);
      const int64_t d = (int64_t)col[3 * i + a] - m[a];
      dist += d < 0 ? -d : d;
    }
    if (dist >= Ts) { rgb[i].r = (uint8_t)m[0]; rgb[i].g = (uint8_t)m[1]; rgb[i].b = (uint8_t)m[2]; }
  }
  free(cells);
  free(col);
  return VPCC_OK;
}
