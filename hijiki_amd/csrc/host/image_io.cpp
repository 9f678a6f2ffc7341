// Image output of the reference's `Renderer::save_image` (src/main.rs:1395-1419): three FLOAT channels R, G, B,
// written as an OpenEXR scan-line file — here without the OpenEXR library, as an uncompressed single-part file
// (any EXR reader accepts it) — plus the trivially inspectable PFM.
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

}  // namespace hijiki
