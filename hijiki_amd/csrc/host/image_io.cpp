// Image output of the reference's `Renderer::save_image` (src/main.rs:1395-1419): three FLOAT channels R, G, B,
// written as an OpenEXR scan-line file — here without the OpenEXR library, as an uncompressed single-part file
// (any EXR reader accepts it) — plus the trivially inspectable PFM.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace hijiki {

void write_pfm(const std::string& path, uint32_t w, uint32_t h, const float* rgb) {
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) throw std::runtime_error("cannot create " + path);
  std::fprintf(f, "PF\n%u %u\n-1.0\n", w, h);   // negative scale = little endian
  for (uint32_t y = 0; y < h; y++)              // PFM stores the bottom row first
    std::fwrite(rgb + (size_t)(h - 1 - y) * w * 3, sizeof(float), (size_t)w * 3, f);
  if (std::fclose(f) != 0) throw std::runtime_error("write failed: " + path);
}

namespace {
void put_bytes(std::vector<uint8_t>& b, const void* p, size_t n) { b.insert(b.end(), (const uint8_t*)p, (const uint8_t*)p + n); }
void put_str(std::vector<uint8_t>& b, const char* s) { put_bytes(b, s, std::strlen(s) + 1); }
void put_i32(std::vector<uint8_t>& b, int32_t v) { put_bytes(b, &v, 4); }
void put_f32(std::vector<uint8_t>& b, float v) { put_bytes(b, &v, 4); }
void attr(std::vector<uint8_t>& b, const char* name, const char* type, const std::vector<uint8_t>& data) {
  put_str(b, name);
  put_str(b, type);
  put_i32(b, (int32_t)data.size());
  put_bytes(b, data.data(), data.size());
}
}  // namespace

void write_exr(const std::string& path, uint32_t w, uint32_t h, const float* rgb) {
  std::vector<uint8_t> hd;
  const uint8_t magic[4] = {0x76, 0x2f, 0x31, 0x01};
  put_bytes(hd, magic, 4);
  put_i32(hd, 2);   // version 2, no flags: single-part scan-line image
  {
    std::vector<uint8_t> ch;
    for (const char* name : {"B", "G", "R"}) {   // channel list is sorted by name
      put_str(ch, name);
      put_i32(ch, 2);                            // FLOAT (the reference writes PixelType::FLOAT, src/main.rs:1411-1413)
      const uint8_t lin[4] = {0, 0, 0, 0};
      put_bytes(ch, lin, 4);
      put_i32(ch, 1);
      put_i32(ch, 1);
    }
    ch.push_back(0);
    attr(hd, "channels", "chlist", ch);
  }
  attr(hd, "compression", "compression", {0});   // NO_COMPRESSION
  {
    std::vector<uint8_t> box;
    put_i32(box, 0); put_i32(box, 0); put_i32(box, (int32_t)w - 1); put_i32(box, (int32_t)h - 1);
    attr(hd, "dataWindow", "box2i", box);
    attr(hd, "displayWindow", "box2i", box);
  }
  attr(hd, "lineOrder", "lineOrder", {0});       // INCREASING_Y
  { std::vector<uint8_t> v; put_f32(v, 1.0f); attr(hd, "pixelAspectRatio", "float", v); }
  { std::vector<uint8_t> v; put_f32(v, 0.0f); put_f32(v, 0.0f); attr(hd, "screenWindowCenter", "v2f", v); }
  { std::vector<uint8_t> v; put_f32(v, 1.0f); attr(hd, "screenWindowWidth", "float", v); }
  hd.push_back(0);   // end of header

  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) throw std::runtime_error("cannot create " + path);
  std::fwrite(hd.data(), 1, hd.size(), f);
  const uint64_t line_bytes = 8 + (uint64_t)w * 3 * 4;
  uint64_t off = hd.size() + (uint64_t)h * 8;
  for (uint32_t y = 0; y < h; y++, off += line_bytes) std::fwrite(&off, 8, 1, f);   // line offset table
  std::vector<float> row((size_t)w * 3);
  for (uint32_t y = 0; y < h; y++) {
    const int32_t yy = (int32_t)y, size = (int32_t)(w * 3 * 4);
    std::fwrite(&yy, 4, 1, f);
    std::fwrite(&size, 4, 1, f);
    for (int c = 0; c < 3; c++)                  // planar per line, channels in list order B, G, R
      for (uint32_t x = 0; x < w; x++) row[(size_t)c * w + x] = rgb[((size_t)y * w + x) * 3 + (2 - c)];
    std::fwrite(row.data(), 4, row.size(), f);
  }
  if (std::fclose(f) != 0) throw std::runtime_error("write failed: " + path);
}

// What the reference shows in its preview window (shader/preview.glsl:9-12 writes rgb/w to an sRGB swapchain): an
// 8-bit sRGB PNG.  No zlib here: the IDAT stream uses stored (uncompressed) deflate blocks, which every PNG reader
// accepts.
namespace {
uint32_t crc32_update(uint32_t c, const uint8_t* p, size_t n) {
  static uint32_t table[256];
  static bool init = false;
  if (!init) {
    for (uint32_t i = 0; i < 256; i++) {
      uint32_t k = i;
      for (int j = 0; j < 8; j++) k = (k & 1u) ? 0xEDB88320u ^ (k >> 1) : k >> 1;
      table[i] = k;
    }
    init = true;
  }
  for (size_t i = 0; i < n; i++) c = table[(c ^ p[i]) & 0xFFu] ^ (c >> 8);
  return c;
}
void put_be32(std::vector<uint8_t>& b, uint32_t v) {
  const uint8_t x[4] = {(uint8_t)(v >> 24), (uint8_t)(v >> 16), (uint8_t)(v >> 8), (uint8_t)v};
  put_bytes(b, x, 4);
}
void png_chunk(std::vector<uint8_t>& out, const char type[4], const std::vector<uint8_t>& data) {
  put_be32(out, (uint32_t)data.size());
  const size_t start = out.size();
  put_bytes(out, type, 4);
  put_bytes(out, data.data(), data.size());
  put_be32(out, crc32_update(0xFFFFFFFFu, out.data() + start, out.size() - start) ^ 0xFFFFFFFFu);
}
uint8_t srgb8(float v) {
  if (!(v > 0.0f)) return 0;                    // negatives and NaN
  if (v >= 1.0f) return 255;
  const float e = v <= 0.0031308f ? 12.92f * v : 1.055f * std::pow(v, 1.0f / 2.4f) - 0.055f;
  return (uint8_t)(e * 255.0f + 0.5f);
}
}  // namespace

void write_png(const std::string& path, uint32_t w, uint32_t h, const float* rgb) {
  std::vector<uint8_t> raw;                      // filter byte 0 + RGB8 per scan line
  raw.reserve((size_t)h * (1 + (size_t)w * 3));
  for (uint32_t y = 0; y < h; y++) {
    raw.push_back(0);
    for (size_t i = 0; i < (size_t)w * 3; i++) raw.push_back(srgb8(rgb[(size_t)y * w * 3 + i]));
  }
  std::vector<uint8_t> z = {0x78, 0x01};         // zlib header, then stored blocks of at most 65535 bytes
  uint32_t a = 1, b = 0;                         // Adler-32 of the raw data
  for (size_t pos = 0; pos < raw.size() || pos == 0;) {
    const size_t n = std::min<size_t>(65535, raw.size() - pos);
    z.push_back(pos + n >= raw.size() ? 1 : 0);
    const uint8_t len[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)~n, (uint8_t)(~n >> 8)};
    put_bytes(z, len, 4);
    put_bytes(z, raw.data() + pos, n);
    for (size_t i = 0; i < n; i++) { a = (a + raw[pos + i]) % 65521u; b = (b + a) % 65521u; }
    pos += n;
    if (n == 0) break;
  }
  put_be32(z, (b << 16) | a);
  std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  std::vector<uint8_t> ihdr;
  put_be32(ihdr, w);
  put_be32(ihdr, h);
  const uint8_t fmt[5] = {8, 2, 0, 0, 0};        // 8 bits, truecolour, deflate, adaptive filtering, no interlace
  put_bytes(ihdr, fmt, 5);
  png_chunk(out, "IHDR", ihdr);
  png_chunk(out, "sRGB", std::vector<uint8_t>{0});
  png_chunk(out, "IDAT", z);
  png_chunk(out, "IEND", {});
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) throw std::runtime_error("cannot create " + path);
  const bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
  if (std::fclose(f) != 0 || !ok) throw std::runtime_error("write failed: " + path);
}

}  // namespace hijiki
