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
#include <new>
#include <stdexcept>
#include <vector>
#include <cstdlib>
#include <cmath>
#include <zlib.h>

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
    if (channels != 1 && channels != 3 && channels != 4) return har_set_error("har_image_write_exr: 1 (luminance), 3 (rgb) or 4 (rgba) channels are supported");
    if (width == 0 || height == 0) return har_set_error("har_image_write_exr: empty image");
    Out o;
    o.i32(20000630); o.i32(2);                                   /* magic, version 2, single-part scanline */
    /* channels in alphabetical order: A, B, G, R; each: name, pixel type (2 = FLOAT), pLinear, reserved[3], xSampling, ySampling */
    const char *names[4] = { "A", "B", "G", "R" }; int src[4] = { 3, 2, 1, 0 };
    if (channels == 1) { names[3] = "Y"; src[3] = 0; }          /* PixelFormat::Y */
    const int first = channels == 4 ? 0 : channels == 3 ? 1 : 3;
    Out ch;
    for (int k = first; k < 4; ++k) { ch.str(names[k]); ch.i32(2); ch.u8(0); ch.u8(0); ch.u8(0); ch.u8(0); ch.i32(1); ch.i32(1); }
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
        for (int k = first; k < 4; ++k) {
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


/* ---- reading (Bitmap(filename), src/core/bitmap.cpp read_exr / read_pfm), as far as the environment-map emitter needs it:
 *      OpenEXR scanline files with NO / ZIPS / ZIP compression, HALF / FLOAT / UINT channels named R G B [A] or Y (a layer prefix is
 *      ignored), any data window; PFM "PF" / "Pf" of either endianness.  The result is H x W x C float32 with C = 1, 3 or 4. */
static float half_to_float(uint16_t h) {
    uint32_t sign = (uint32_t) (h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu, bits;
    if (e == 0) {
        if (m == 0) bits = sign;
        else { int sh = 0; while (!(m & 0x400u)) { m <<= 1; ++sh; } bits = sign | ((uint32_t) (113 - sh) << 23) | ((m & 0x3ffu) << 13); }
    } else if (e == 31) bits = sign | 0x7f800000u | (m << 13);
    else bits = sign | ((e + 112u) << 23) | (m << 13);
    float f; memcpy(&f, &bits, 4); return f;
}

static int image_read_impl(const char *filename, HarImage *out);
/* the file's contents are untrusted: every size taken from it is checked against the bytes that are there (and against 2^31 pixels), allocations
 * are checked, and no C++ exception crosses the C boundary */
int har_image_read(const char *filename, HarImage *out) {
    try { return image_read_impl(filename, out); }
    catch (const std::bad_alloc &) { if (out) { free(out->data); out->data = nullptr; } return har_set_error(std::string("Error while loading \"") + (filename ? filename : "") + "\": out of memory"); }
    catch (const std::exception &e) { if (out) { free(out->data); out->data = nullptr; } return har_set_error(std::string("Error while loading \"") + (filename ? filename : "") + "\": " + e.what()); }
}
static int image_read_impl(const char *filename, HarImage *out) {
    if (!filename || !out) return har_set_error("null argument");
    out->data = nullptr; out->width = out->height = out->channels = 0;
    FILE *f = fopen(filename, "rb");
    if (!f) return har_set_error(std::string("\"") + filename + "\": file not found / unreadable");
    std::vector<uint8_t> d; { uint8_t buf[1 << 16]; size_t n; while ((n = fread(buf, 1, sizeof(buf), f)) > 0) d.insert(d.end(), buf, buf + n); } fclose(f);
    auto fail = [&](const std::string &m) { free(out->data); out->data = nullptr; return har_set_error(std::string("Error while loading \"") + filename + "\": " + m); };
    if (d.size() >= 2 && d[0] == 'P' && (d[1] == 'F' || d[1] == 'f')) {                       /* ---- PFM */
        size_t pos = 0; std::string tok[4]; int nt = 0;
        while (nt < 4 && pos < d.size()) {
            while (pos < d.size() && isspace(d[pos])) ++pos;
            while (pos < d.size() && !isspace(d[pos])) tok[nt] += (char) d[pos++];
            ++nt;
        }
        ++pos;                                                                               /* the single whitespace after the scale */
        const uint32_t c = tok[0] == "PF" ? 3u : 1u; const long w = atol(tok[1].c_str()), h = atol(tok[2].c_str()); const double scale = atof(tok[3].c_str());
        if (nt < 4 || w <= 0 || h <= 0 || scale == 0.0 || !(scale == scale)) return fail("invalid PFM header");
        if (w > 0x7fffffffl || h > 0x7fffffffl || (uint64_t) w * (uint64_t) h > 0x7fffffffull) return fail("image too large");
        if (pos > d.size() || (uint64_t) w * h * c * 4 > d.size() - pos) return fail("unexpected end of file");
        out->data = (float *) malloc((size_t) w * h * c * 4);
        if (!out->data) return fail("out of memory");
        out->width = (uint32_t) w; out->height = (uint32_t) h; out->channels = c;
        const bool big = scale > 0.0;
        for (long y = 0; y < h; ++y)
            for (size_t i = 0; i < (size_t) w * c; ++i) {
                uint8_t b[4]; memcpy(b, d.data() + pos + ((size_t) (h - 1 - y) * w * c + i) * 4, 4);
                if (big) { std::swap(b[0], b[3]); std::swap(b[1], b[2]); }
                memcpy(out->data + (size_t) y * w * c + i, b, 4);
            }
        return 0;
    }
    /* ---- OpenEXR */
    size_t pos = 0;
    auto need = [&](size_t n) { return pos + n <= d.size(); };
    auto rd32 = [&]() { int32_t v = 0; if (need(4)) { memcpy(&v, d.data() + pos, 4); pos += 4; } else pos = d.size(); return v; };
    auto rdstr = [&]() { std::string r; while (pos < d.size() && d[pos]) r += (char) d[pos++]; ++pos; return r; };
    if (d.size() < 8 || rd32() != 20000630) return fail("unknown file format (OpenEXR and PFM are implemented)");
    const int32_t version = rd32();
    if ((version & 0xff) != 2 || (version & 0x1a00)) return fail("only single-part scanline OpenEXR files are supported");     /* tiled / deep / multipart bits */
    struct Chan { std::string name; int32_t type; };
    std::vector<Chan> chans; int comp = -1; int32_t win[4] = { 0, 0, -1, -1 }; int line_order = 0;
    while (pos < d.size() && d[pos]) {
        std::string name = rdstr(), type = rdstr(); int32_t size = rd32(); size_t start = pos;
        if (size < 0 || !need((size_t) size)) return fail("corrupt header");
        if (name == "channels") {
            while (pos < start + (size_t) size && d[pos]) {
                Chan c; c.name = rdstr(); c.type = rd32(); if (!need(12)) return fail("corrupt channel list"); pos += 4; int32_t xs = rd32(), ys = rd32();
                if (xs != 1 || ys != 1) return fail("subsampled channels are not supported");
                if (c.type < 0 || c.type > 2) return fail("unknown channel type");
                if (pos > start + (size_t) size) return fail("corrupt channel list");
                chans.push_back(c);
                if (chans.size() > 1024) return fail("too many channels");
            }
        } else if (name == "compression") { if (size < 1) return fail("corrupt header"); comp = d[pos]; }
        else if (name == "dataWindow") { if (size != 16) return fail("corrupt dataWindow attribute"); memcpy(win, d.data() + pos, 16); }
        else if (name == "lineOrder") { if (size < 1) return fail("corrupt header"); line_order = d[pos]; }
        pos = start + (size_t) size;
    }
    ++pos;
    if (chans.empty() || comp < 0 || win[2] < win[0] || win[3] < win[1]) return fail("incomplete header");
    if (comp != 0 && comp != 2 && comp != 3) return fail("unsupported compression (NO_COMPRESSION, ZIPS and ZIP are implemented)");
    (void) line_order;                                          /* chunks carry their y coordinate */
    const int64_t W64 = (int64_t) win[2] - (int64_t) win[0] + 1, H64 = (int64_t) win[3] - (int64_t) win[1] + 1;       /* no int32 overflow in the subtraction */
    if (W64 <= 0 || H64 <= 0 || W64 > 0x7fffffffll || H64 > 0x7fffffffll || W64 * H64 > 0x7fffffffll) return fail("invalid data window");
    const uint32_t W = (uint32_t) W64, H = (uint32_t) H64;
    /* map file channels to output channels */
    int map_[4] = { -1, -1, -1, -1 }; bool lum = false;
    for (size_t k = 0; k < chans.size(); ++k) {
        std::string n = chans[k].name; size_t dot = n.rfind('.'); if (dot != std::string::npos) n = n.substr(dot + 1);
        if (n == "R") map_[0] = (int) k; else if (n == "G") map_[1] = (int) k; else if (n == "B") map_[2] = (int) k; else if (n == "A") map_[3] = (int) k;
        else if (n == "Y") { map_[0] = (int) k; lum = true; }
    }
    uint32_t C = 0;
    if (lum && map_[1] < 0) C = 1; else if (map_[0] >= 0 && map_[1] >= 0 && map_[2] >= 0) C = map_[3] >= 0 ? 4 : 3;
    else return fail("expected channels R, G, B [, A] or Y");
    std::vector<size_t> ch_off(chans.size()); size_t line_bytes = 0;
    for (size_t k = 0; k < chans.size(); ++k) { ch_off[k] = line_bytes; line_bytes += (size_t) W * (chans[k].type == 1 ? 2 : 4); }
    const uint32_t lines_per_chunk = comp == 3 ? 16u : 1u, n_chunks = (H + lines_per_chunk - 1) / lines_per_chunk;
    if (!need(8ull * n_chunks)) return fail("unexpected end of file");
    /* the pixel data cannot be smaller than ~1/1000 of the decoded size (zlib's best ratio): refuse windows the file cannot possibly fill */
    if ((uint64_t) W * H * C * 4 / 1100 > d.size() + 1024) return fail("data window larger than the file can hold");
    out->data = (float *) malloc((size_t) W * H * C * 4);
    if (!out->data) return fail("out of memory");
    memset(out->data, 0, (size_t) W * H * C * 4);
    out->width = W; out->height = H; out->channels = C;
    std::vector<uint8_t> raw, tmp;
    for (uint32_t c = 0; c < n_chunks; ++c) {
        uint64_t off; memcpy(&off, d.data() + pos + 8ull * c, 8);
        if (off > d.size() || d.size() - off < 8) return fail("corrupt chunk table");
        int32_t y0, sz; memcpy(&y0, d.data() + off, 4); memcpy(&sz, d.data() + off + 4, 4);
        if (sz < 0 || (uint64_t) sz > d.size() - off - 8 || y0 < win[1] || y0 > win[3]) return fail("corrupt chunk");
        const uint32_t nl = std::min<uint32_t>(lines_per_chunk, (uint32_t) (win[3] - y0 + 1));
        const size_t expect = line_bytes * nl;
        const uint8_t *src = d.data() + off + 8;
        if (comp != 0 && (size_t) sz < expect) {                 /* zlib + predictor + byte interleave (ImfZip.cpp) */
            tmp.resize(expect); uLongf got = (uLongf) expect;
            if (uncompress(tmp.data(), &got, src, (uLong) sz) != Z_OK || got != expect) return fail("zlib: corrupt chunk");
            for (size_t i = 1; i < expect; ++i) tmp[i] = (uint8_t) (tmp[i - 1] + tmp[i] - 128);
            raw.resize(expect); const size_t half = (expect + 1) / 2;
            for (size_t i = 0; i < expect; ++i) raw[i] = (i & 1) ? tmp[half + i / 2] : tmp[i / 2];
            src = raw.data();
        } else if ((size_t) sz != expect) return fail("corrupt chunk size");
        for (uint32_t l = 0; l < nl; ++l) {
            const uint8_t *line = src + line_bytes * l; float *dst = out->data + (size_t) (y0 - win[1] + (int32_t) l) * W * C;
            for (uint32_t oc = 0; oc < C; ++oc) {
                const int k = map_[oc]; const uint8_t *p = line + ch_off[k];
                for (uint32_t x = 0; x < W; ++x) {
                    float v;
                    if (chans[k].type == 1) { uint16_t hv; memcpy(&hv, p + 2 * (size_t) x, 2); v = half_to_float(hv); }
                    else if (chans[k].type == 2) memcpy(&v, p + 4 * (size_t) x, 4);
                    else { uint32_t u; memcpy(&u, p + 4 * (size_t) x, 4); v = (float) u; }
                    dst[(size_t) x * C + oc] = v;
                }
            }
        }
    }
    return 0;
}

void har_image_free(HarImage *img) { if (img) { free(img->data); img->data = nullptr; img->width = img->height = img->channels = 0; } }

} // extern "C"
