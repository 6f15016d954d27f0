// png_io.h -- PNG reading/writing for text-line images without libpng (absent in this image): the
// chunk parser, zlib inflate (system zlib) and scan-line unfiltering are done here; the pixel
// conversions restate what the reference asks libpng for (extras.cc:313-429: STRIP_16, STRIP_ALPHA,
// PACKING, EXPAND) and read_png's final mapping (extras.cc:529-545):
//   colour images -> (r+g+b)/(3*255) in [0,1];   grey images -> the raw 0..255 value (NOT scaled:
//   a quirk of the reference, kept).  image(x, y): x = column, y = row.
#pragma once
#include <zlib.h>

#include "hostutil.h"

namespace clstmhost {

inline unsigned be32(const unsigned char* p) { return (p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }

inline void read_png(Image& image, const string& name) {
  std::ifstream f(name, std::ios::binary);
  if (!f) fail("error on open: " + name);
  vector<unsigned char> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (buf.size() < 8 || memcmp(buf.data(), sig, 8)) fail("not a PNG file: " + name);
  unsigned w = 0, h = 0, depth = 0, ctype = 0, interlace = 0;
  vector<unsigned char> idat, plte;
  size_t pos = 8;
  while (pos + 12 <= buf.size()) {
    unsigned len = be32(&buf[pos]);
    string type((char*)&buf[pos + 4], 4);
    const unsigned char* data = &buf[pos + 8];
    if (pos + 12 + len > buf.size()) fail("truncated PNG: " + name);
    if (type == "IHDR") {
      w = be32(data); h = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
    } else if (type == "PLTE") plte.assign(data, data + len);
    else if (type == "IDAT") idat.insert(idat.end(), data, data + len);
    else if (type == "IEND") break;
    pos += 12 + len;
  }
  if (w == 0 || h == 0) fail("bad PNG header: " + name);
  if (interlace) fail("interlaced PNG is not supported: " + name);
  const int chans = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
  if (!chans) fail("bad PNG colour type");
  const size_t bpp_bits = (size_t)chans * depth;
  const size_t stride = (w * bpp_bits + 7) / 8, bpp = (bpp_bits + 7) / 8;
  vector<unsigned char> raw((stride + 1) * h);
  uLongf rawlen = raw.size();
  if (uncompress(raw.data(), &rawlen, idat.data(), idat.size()) != Z_OK || rawlen != raw.size())
    fail("PNG inflate failed: " + name);
  // undo the scan-line filters (PNG spec 9.2)
  vector<unsigned char> pix(stride * h);
  for (unsigned y = 0; y < h; y++) {
    const unsigned char ft = raw[y * (stride + 1)];
    const unsigned char* in = &raw[y * (stride + 1) + 1];
    unsigned char* out = &pix[y * stride];
    const unsigned char* up = y ? &pix[(y - 1) * stride] : nullptr;
    for (size_t i = 0; i < stride; i++) {
      int a = i >= bpp ? out[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0, x = in[i];
      int v;
      switch (ft) {
        case 0: v = x; break;
        case 1: v = x + a; break;
        case 2: v = x + b; break;
        case 3: v = x + ((a + b) >> 1); break;
        case 4: {
          int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
          v = x + ((pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c));
          break;
        }
        default: fail("bad PNG filter");
      }
      out[i] = (unsigned char)v;
    }
  }
  auto sample = [&](unsigned y, unsigned x, int ch) -> int {  // 8-bit sample after STRIP_16 / EXPAND
    const unsigned char* row = &pix[y * stride];
    if (depth == 8) return row[x * chans + ch];
    if (depth == 16) return row[(x * chans + ch) * 2];  // STRIP_16: high byte
    const unsigned idx = x * chans + ch, per = 8 / depth;  // packed 1, 2, 4 bit samples (one channel)
    const int v = (row[idx / per] >> ((per - 1 - idx % per) * depth)) & ((1 << depth) - 1);
    return ctype == 3 ? v : v * 255 / ((1 << depth) - 1);  // EXPAND scales grey to 8 bit
  };
  image.resize(w, h);
  for (unsigned y = 0; y < h; y++)
    for (unsigned x = 0; x < w; x++) {
      if (ctype == 0 || ctype == 4) {
        image(x, y) = (float)sample(y, x, 0);  // spp == 1: raw value (extras.cc:537-538)
      } else if (ctype == 3) {
        const int i = sample(y, x, 0);
        if ((size_t)3 * i + 2 >= plte.size()) fail("PNG palette index out of range");
        image(x, y) = (plte[3 * i] + plte[3 * i + 1] + plte[3 * i + 2]) / (3 * 255.0);
      } else {
        image(x, y) = (sample(y, x, 0) + sample(y, x, 1) + sample(y, x, 2)) / (3 * 255.0);  // :540-541
      }
    }
}

// write_png (extras.cc:546-561): grey value floor(clip(v*256, 0, 255.999999)) replicated to RGB
inline void write_png(const string& name, const Image& image) {
  const unsigned w = image.w, h = image.h;
  vector<unsigned char> raw((size_t)(3 * w + 1) * h);
  for (unsigned y = 0; y < h; y++) {
    raw[(size_t)y * (3 * w + 1)] = 0;
    for (unsigned x = 0; x < w; x++) {
      double v = image(x, y) * 256;
      v = v < 0.0 ? 0.0 : v > 255.999999 ? 255.999999 : v;
      const unsigned char b = (unsigned char)floor(v);
      unsigned char* p = &raw[(size_t)y * (3 * w + 1) + 1 + 3 * x];
      p[0] = p[1] = p[2] = b;
    }
  }
  uLongf clen = compressBound(raw.size());
  vector<unsigned char> comp(clen);
  if (compress(comp.data(), &clen, raw.data(), raw.size()) != Z_OK) fail("PNG deflate failed");
  std::ofstream f(name, std::ios::binary);
  if (!f) fail("error on open: " + name);
  auto put32 = [](unsigned char* p, unsigned v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; };
  auto chunk = [&](const char* type, const unsigned char* data, unsigned len) {
    vector<unsigned char> c(len + 12);
    put32(&c[0], len);
    memcpy(&c[4], type, 4);
    if (len) memcpy(&c[8], data, len);
    put32(&c[8 + len], (unsigned)crc32(0, &c[4], len + 4));
    f.write((const char*)c.data(), c.size());
  };
  static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  f.write((const char*)sig, 8);
  unsigned char ihdr[13];
  put32(ihdr, w); put32(ihdr + 4, h);
  ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = ihdr[11] = ihdr[12] = 0;
  chunk("IHDR", ihdr, 13);
  chunk("IDAT", comp.data(), (unsigned)clen);
  chunk("IEND", nullptr, 0);
}

}  // namespace clstmhost
