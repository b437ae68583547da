// hdr_writer.cpp — Radiance .hdr export. Replaces what BDPT's "Export HDR" gets from
// stbi_write_hdr(path, w, h, 4, pixels) (src/Node/BDPT.cpp:313-337; stb_image_write v1.16 is the vendored
// dependency). The file is byte-compatible with that writer: same header text, same RGBE quantisation
// (truncating, shared exponent from frexp of the largest channel), same per-channel run-length packets (runs of
// >= 3 equal bytes, at most 127 per run packet and 128 per literal packet). tests/test_post.py pins the bytes
// against a fixture produced by the vendored writer itself (tests/golden/make_hdr_golden.py).
#include <math.h>
#include <stdint.h>
#include <stdio.h>

#include <vector>

#include "../../include/sthip.h"

namespace {

void rgbe_of(const float* px, uint8_t out[4]) {
  const float m = px[0] > (px[1] > px[2] ? px[1] : px[2]) ? px[0] : (px[1] > px[2] ? px[1] : px[2]);
  if (m < 1e-32f) {
    out[0] = out[1] = out[2] = out[3] = 0;
    return;
  }
  int e;
  const float scale = (float)frexp((double)m, &e) * 256.0f / m;
  for (int k = 0; k < 3; k++) out[k] = (uint8_t)(int32_t)(px[k] * scale);
  out[3] = (uint8_t)(e + 128);
}

// one colour plane of a scanline as Radiance "new RLE" packets
void pack_plane(const uint8_t* v, uint32_t n, std::vector<uint8_t>& out) {
  uint32_t pos = 0;
  while (pos < n) {
    // literals run up to the next triple of equal bytes (or the end of the row)
    uint32_t run = pos;
    bool found = false;
    for (; run + 2 < n; run++)
      if (v[run] == v[run + 1] && v[run] == v[run + 2]) {
        found = true;
        break;
      }
    const uint32_t lit_end = found ? run : n;
    while (pos < lit_end) {
      const uint32_t len = lit_end - pos > 128 ? 128 : lit_end - pos;
      out.push_back((uint8_t)len);
      out.insert(out.end(), v + pos, v + pos + len);
      pos += len;
    }
    if (found) {
      uint32_t end = run;
      while (end < n && v[end] == v[run]) end++;
      while (pos < end) {
        const uint32_t len = end - pos > 127 ? 127 : end - pos;
        out.push_back((uint8_t)(128 + len));
        out.push_back(v[run]);
        pos += len;
      }
    }
  }
}

}  // namespace

extern "C" int sthip_write_hdr(const char* path, uint32_t width, uint32_t height, const float* rgba) {
  if (!path || !rgba || width == 0 || height == 0 || width > 0x7FFFFFFFu || height > 0x7FFFFFFFu) return STHIP_ERR_INVALID_ARGUMENT;
  std::vector<uint8_t> out;
  char head[160];
  const int hl = snprintf(head, sizeof(head), "#?RADIANCE\n# Written by stb_image_write.h\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=          1.0000000000000\n\n-Y %d +X %d\n", (int)height, (int)width);
  out.insert(out.end(), head, head + hl);
  const bool rle = width >= 8 && width < 32768;
  std::vector<uint8_t> planes((size_t)width * 4);
  for (uint32_t y = 0; y < height; y++) {
    const float* row = rgba + (size_t)y * width * 4;
    if (!rle) {
      for (uint32_t x = 0; x < width; x++) {
        uint8_t q[4];
        rgbe_of(row + (size_t)x * 4, q);
        out.insert(out.end(), q, q + 4);
      }
      continue;
    }
    for (uint32_t x = 0; x < width; x++) {
      uint8_t q[4];
      rgbe_of(row + (size_t)x * 4, q);
      for (int k = 0; k < 4; k++) planes[(size_t)k * width + x] = q[k];
    }
    const uint8_t mark[4] = {2, 2, (uint8_t)(width >> 8), (uint8_t)(width & 0xFF)};
    out.insert(out.end(), mark, mark + 4);
    for (int k = 0; k < 4; k++) pack_plane(planes.data() + (size_t)k * width, width, out);
  }
  FILE* f = fopen(path, "wb");
  if (!f) return STHIP_ERR_INVALID_ARGUMENT;
  const size_t w = fwrite(out.data(), 1, out.size(), f);
  const int c = fclose(f);
  return (w == out.size() && c == 0) ? STHIP_OK : STHIP_ERR_INVALID_ARGUMENT;
}
