/*
 * har_image_io.cpp -- developed film -> image files (HDRFilm::write, src/films/hdrfilm.cpp:414-560,
 * which hands the developed bitmap to Bitmap::write, src/core/bitmap.cpp).
 *   har_image_write_exr  OpenEXR 2 scanline file, uncompressed (NO_COMPRESSION), FLOAT channels named like the
 *                        reference's rgb / rgba pixel formats ("R", "G", "B" [, "A"]), increasing-y line order
 *   har_image_write_pfm  Portable Float Map ("PF" / "Pf", little endian, bottom-to-top rows)
 * Host pointers, row-major H x W x C float32.  No third-party library (ext/openexr is an empty submodule).
 */
#include "../../include/hip_ad_rgb.h"
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

extern int har_set_error(const std::string &msg);

namespace {
struct Out {
    std::vector<uint8_t> b;
    void raw(const void *p, size_t n) { const uint8_t *q = (const uint8_t *) p; b.insert(b.end(), q, q + n); }
    void u8(uint8_t v) { b.push_back(v); }
    void i32(int32_t v) { raw(&v, 4); }   void u64(uint64_t v) { raw(&v, 8); }   void f32(float v) { raw(&v, 4); }
    void str(const char *s) { raw(s, strlen(s) + 1); }
    void attr(const char *name, const char *type, const void *data, int32_t size) { str(name); str(type); i32(size); raw(data, (size_t) size); }
};
}

extern "C" {

int har_image_write_exr(const char *filename, const float *image, uint32_t width, uint32_t height, uint32_t channels) {
    if (!filename || !image) return har_set_error("null argument");
    if (channels != 3 && channels != 4) return har_set_error("har_image_write_exr: 3 (rgb) or 4 (rgba) channels are supported");
    if (width == 0 || height == 0) return har_set_error("har_image_write_exr: empty image");
    Out o;
    o.i32(20000630); o.i32(2);                                   /* magic, version 2, single-part scanline */
    /* channels in alphabetical order: A, B, G, R; each: name, pixel type (2 = FLOAT), pLinear, reserved[3], xSampling, ySampling */
    const char *names[4] = { "A", "B", "G", "R" }; const int src[4] = { 3, 2, 1, 0 };
    Out ch;
    for (int k = channels == 4 ? 0 : 1; k < 4; ++k) { ch.str(names[k]); ch.i32(2); ch.u8(0); ch.u8(0); ch.u8(0); ch.u8(0); ch.i32(1); ch.i32(1); }
    ch.u8(0);
    o.attr("channels", "chlist", ch.b.data(), (int32_t) ch.b.size());
    uint8_t comp = 0; o.attr("compression", "compression", &comp, 1);
    int32_t win[4] = { 0, 0, (int32_t) width - 1, (int32_t) height - 1 };
    o.attr("dataWindow", "box2i", win, 16); o.attr("displayWindow", "box2i", win, 16);
    uint8_t lo = 0; o.attr("lineOrder", "lineOrder", &lo, 1);
    float one = 1.f, zero2[2] = { 0.f, 0.f };
    o.attr("pixelAspectRatio", "float", &one, 4); o.attr("screenWindowCenter", "v2f", zero2, 8); o.attr("screenWindowWidth", "float", &one, 4);
    o.u8(0);                                                     /* end of header */
    const uint32_t nch = channels; const uint64_t line_bytes = 8 + (uint64_t) nch * width * 4;
    uint64_t offset = o.b.size() + 8ull * height;
    for (uint32_t y = 0; y < height; ++y) { o.u64(offset); offset += line_bytes; }
    std::vector<float> row(width);
    for (uint32_t y = 0; y < height; ++y) {
        o.i32((int32_t) y); o.i32((int32_t) (nch * width * 4));
        for (int k = channels == 4 ? 0 : 1; k < 4; ++k) {
            for (uint32_t x = 0; x < width; ++x) row[x] = image[((size_t) y * width + x) * channels + src[k]];
            o.raw(row.data(), (size_t) width * 4);
        }
    }
    FILE *f = fopen(filename, "wb");
    if (!f) return har_set_error(std::string("cannot open \"") + filename + "\" for writing");
    size_t n = fwrite(o.b.data(), 1, o.b.size(), f); fclose(f);
    return n == o.b.size() ? 0 : har_set_error("short write");
}

int har_image_write_pfm(const char *filename, const float *image, uint32_t width, uint32_t height, uint32_t channels) {
    if (!filename || !image) return har_set_error("null argument");
    if (channels != 1 && channels != 3) return har_set_error("har_image_write_pfm: 1 or 3 channels are supported");
    FILE *f = fopen(filename, "wb");
    if (!f) return har_set_error(std::string("cannot open \"") + filename + "\" for writing");
    fprintf(f, "%s\n%u %u\n-1.000000\n", channels == 3 ? "PF" : "Pf", width, height);     /* negative scale = little endian */
    for (uint32_t y = 0; y < height; ++y)                      /* rows are stored bottom to top */
        fwrite(image + (size_t) (height - 1 - y) * width * channels, sizeof(float), (size_t) width * channels, f);
    fclose(f);
    return 0;
}

} // extern "C"
