/*
 * har_mesh_formats.cpp -- OBJ and `serialized` mesh files for the hip_ad_rgb path (SURVEY.md 8f rank 2, next to har_mesh_io.cpp's PLY).
 *
 *   har_mesh_load_obj         OBJMesh ctor (src/shapes/obj.cpp:98-296): v / vn / vt / f records, 1-based `p`, `p/t`, `p//n`, `p/t/n`
 *                             corners, polygons of any size, with the welding rules of Mesh::from_corners
 *                             (src/render/mesh_utils.cpp:210-560) -- see the contract stated above mesh_load_obj_impl; the implementation
 *                             is this repository's own (scanner + hash table + prefix sum).  Missing normals are regenerated per SURFACE POINT
 *                             (Mesh::pack: normal_index = position_index, src/render/mesh.cpp:573-582), i.e. UV seams stay smooth.
 *   har_mesh_load_serialized  SerializedMesh ctor + load_legacy (src/shapes/serialized.cpp:225-370): Mitsuba 0.x / 2 / 3 `.serialized`
 *                             container versions 3 and 4 (zlib stream per sub-mesh, offset table at the end of the file).
 * Output: the packed layout har_scene_create ingests (8 f32 per vertex, 4 u32 per face), malloc'ed.
 */
#include "../../include/hip_ad_rgb.h"
#include "har_math.h"

#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <new>
#include <stdexcept>
#include <vector>

using namespace har;

extern int har_set_error(const std::string &msg);
extern int har_mesh_finalize(std::vector<float> &V, std::vector<uint32_t> &F, bool stored_normals, bool regenerate, const float *to_world32,
                             bool flip_normals, const std::vector<uint32_t> *position_index, uint32_t position_count, bool packed_records = false);     /* har_mesh_io.cpp */

namespace {

bool read_file(const char *filename, std::string &data) {
    std::ifstream f(filename, std::ios::binary);
    if (!f) return false;
    f.seekg(0, std::ios::end); std::streamoff n = f.tellg(); f.seekg(0);
    if (n < 0) return false;
    data.resize((size_t) n);
    if (n) f.read(&data[0], n);
    return (bool) f;
}

int emit(const std::vector<float> &V, const std::vector<uint32_t> &F, uint32_t flags, HarMeshData *out) {
    out->vertices = (float *) malloc(std::max<size_t>(V.size(), 1) * sizeof(float));
    out->faces = (uint32_t *) malloc(std::max<size_t>(F.size(), 1) * sizeof(uint32_t));
    if (!out->vertices || !out->faces) { har_mesh_free(out); return har_set_error("out of memory"); }
    memcpy(out->vertices, V.data(), V.size() * sizeof(float)); memcpy(out->faces, F.data(), F.size() * sizeof(uint32_t));
    out->vertex_count = (uint32_t) (V.size() / 8); out->face_count = (uint32_t) (F.size() / 4); out->flags = flags; out->reserved = 0;
    return 0;
}

const uint32_t kMissing = 0xffffffffu;

/* the words a corner attribute contributes to a welding key: the record's IEEE bit patterns with the sign of a zero cleared (so -0 welds with +0);
 * a corner without that attribute contributes zeros */
void fetch_key(const std::vector<float> &pool, uint32_t index, int dim, uint32_t *out) {
    for (int k = 0; k < dim; ++k) {
        uint32_t word = 0;
        if (index != kMissing) memcpy(&word, &pool[(size_t) index * dim + k], sizeof(word));
        out[k] = (word << 1) == 0u ? 0u : word;
    }
}

} // namespace

extern "C" {

static int mesh_load_obj_impl(const char *filename, int face_normals, int flip_tex_coords, const float *to_world, int flip_normals, HarMeshData *out);
static int mesh_load_serialized_impl(const char *filename, int shape_index, int face_normals, const float *to_world, int flip_normals, HarMeshData *out);
static void mesh_reset(HarMeshData *out) { if (out) { free(out->vertices); free(out->faces); out->vertices = nullptr; out->faces = nullptr; out->vertex_count = out->face_count = out->flags = out->reserved = 0; } }
/* no C++ exception (std::bad_alloc of a file-controlled size, ...) crosses the C boundary */
int har_mesh_load_obj(const char *filename, int face_normals, int flip_tex_coords, const float *to_world, int flip_normals, HarMeshData *out) {
    try { return mesh_load_obj_impl(filename, face_normals, flip_tex_coords, to_world, flip_normals, out); }
    catch (const std::bad_alloc &) { mesh_reset(out); return har_set_error(std::string("Error while loading OBJ file \"") + (filename ? filename : "") + "\": out of memory"); }
    catch (const std::exception &e) { mesh_reset(out); return har_set_error(std::string("Error while loading OBJ file \"") + (filename ? filename : "") + "\": " + e.what()); }
}
int har_mesh_load_serialized(const char *filename, int shape_index, int face_normals, const float *to_world, int flip_normals, HarMeshData *out) {
    try { return mesh_load_serialized_impl(filename, shape_index, face_normals, to_world, flip_normals, out); }
    catch (const std::bad_alloc &) { mesh_reset(out); return har_set_error(std::string("Error while loading serialized file \"") + (filename ? filename : "") + "\": out of memory!"); }
    catch (const std::exception &e) { mesh_reset(out); return har_set_error(std::string("Error while loading serialized file \"") + (filename ? filename : "") + "\": " + e.what() + "!"); }
}
/* ---------------------------------------------------------------------------------------------------------------- Wavefront OBJ
 *
 * Written from the observable contract of the reference's loader (what a file produces, which files are refused and with which message),
 * not from its text:
 *   records    `v x y z`, `vn x y z`, `vt u v`, `f c c c ...` with corners `p`, `p/t`, `p//n`, `p/t/n` (1-based); everything else is skipped;
 *              `vn` is ignored with face_normals, `vt`'s v is mirrored with flip_tex_coords; a line of 1024+ characters is refused;
 *   topology   a polygon c0 c1 ... c(n-1) becomes the fan (c0, ci, ci+1); corners of ONE source point stay one vertex when their normal,
 *              texcoord and the orientation of their triangle in UV space agree bit for bit (-0 == +0, a missing attribute == zeros);
 *   numbering  vertices are numbered point by point in `v` order and, within a point, in the order its variants first occur in the triangle
 *              list; points no face refers to do not exist in the output.
 * Structure here: a character scanner per line, one open-addressing hash table keyed on (point, attribute words) that hands out a variant
 * rank per point, and a prefix sum over the points' variant counts that turns (point, rank) into the final vertex id.
 */
namespace {

struct LineScanner {
    const char *p, *end;
    static bool blank(char c) { return c == ' ' || c == '\t' || c == '\r'; }
    void skip_blanks() { while (p < end && blank(*p)) ++p; }
    bool at_end() { skip_blanks(); return p >= end; }
    /* `word` followed by a blank: a record tag */
    bool tag(const char *word) {
        const size_t n = strlen(word);
        if ((size_t) (end - p) <= n || memcmp(p, word, n) != 0 || !(p[n] == ' ' || p[n] == '\t')) return false;
        p += n + 1; return true;
    }
    /* one real number; the buffer behind the line is NUL-terminated, the line itself ends at `end` (strtof would walk over the newline) */
    bool real(float &v) {
        skip_blanks();
        if (p >= end) return false;
        char *stop = nullptr; v = strtof(p, &stop);
        if (stop == p || stop > end) return false;
        p = stop; return true;
    }
    /* a decimal integer; a sign is read like C's strtoul reads it (the value wraps: `-1` is 4294967295, which no pool can satisfy, so relative
     * OBJ indices end in the "invalid vertex -1" message exactly like in the reference) */
    bool index(uint32_t &v) {
        const char *q = p; bool negative = false;
        if (q < end && (*q == '-' || *q == '+')) { negative = *q == '-'; ++q; }
        if (q >= end || *q < '0' || *q > '9') return false;
        uint64_t acc = 0;
        while (q < end && *q >= '0' && *q <= '9') { acc = std::min<uint64_t>(acc * 10u + (uint64_t) (*q - '0'), 0xffffffffull); ++q; }
        v = negative ? (uint32_t) (0u - (uint32_t) acc) : (uint32_t) acc; p = q; return true;
    }
};

struct ObjSoup {
    std::vector<float> xyz, nrm, uv;                        /* the three attribute pools as read */
    std::vector<uint32_t> c_point, c_uv, c_nrm;            /* per face corner: 0-based pool indices, kMissing = not given */
    std::vector<uint32_t> poly_first;                       /* first corner of every polygon + one past the last */
    bool any_uv = false, any_nrm = false;
};

/* parse one `f` record: a corner is up to three slash-separated indices, corners are separated by blanks */
enum CornerStatus { CORNER_OK, CORNER_NONE, CORNER_MALFORMED };
CornerStatus scan_corner(LineScanner &L, uint32_t ref[3]) {
    ref[0] = ref[1] = ref[2] = 0;
    L.skip_blanks();
    if (L.p >= L.end) return CORNER_NONE;
    int field = 0; bool got_any = false;
    for (;;) {
        uint32_t v;
        if (L.index(v)) { ref[field] = v; got_any = true; }
        if (L.p < L.end && *L.p == '/') { if (++field > 2) return CORNER_MALFORMED; ++L.p; continue; }
        break;
    }
    if (!got_any) return CORNER_NONE;                        /* not a number: the record ends here (trailing junk is ignored like the reference does) */
    if (L.p < L.end && !LineScanner::blank(*L.p)) return CORNER_MALFORMED;
    return CORNER_OK;
}

/* hash table (point, words...) -> rank of that attribute variant within its point */
struct VariantTable {
    size_t words; std::vector<uint32_t> slots;              /* per slot: point, rank, words...; point == kMissing: empty */
    size_t mask = 0, used = 0, stride = 0;
    explicit VariantTable(size_t words_, size_t expected) : words(words_) {
        stride = 2 + words; size_t cap = 16; while (cap < 2 * expected + 8) cap <<= 1;
        mask = cap - 1; slots.assign(cap * stride, kMissing);
    }
    static uint64_t mix(uint64_t h, uint32_t v) { h ^= v; h *= 0x9E3779B97F4A7C15ull; return h ^ (h >> 29); }
    /* returns the rank; `fresh` tells whether the variant is new (its rank is then next_rank) */
    uint32_t find_or_add(uint32_t point, const uint32_t *w, uint32_t next_rank, bool &fresh) {
        uint64_t h = mix(0x243F6A8885A308D3ull, point);
        for (size_t k = 0; k < words; ++k) h = mix(h, w[k]);
        for (size_t i = (size_t) h & mask;; i = (i + 1) & mask) {
            uint32_t *s = slots.data() + i * stride;
            if (s[0] == kMissing) { s[0] = point; s[1] = next_rank; memcpy(s + 2, w, words * sizeof(uint32_t)); ++used; fresh = true; return next_rank; }
            if (s[0] == point && memcmp(s + 2, w, words * sizeof(uint32_t)) == 0) { fresh = false; return s[1]; }
        }
    }
};

} // namespace

static int mesh_load_obj_impl(const char *filename, int face_normals, int flip_tex_coords, const float *to_world, int flip_normals, HarMeshData *out) {
    if (!filename || !out) return har_set_error("null argument");
    out->vertices = nullptr; out->faces = nullptr; out->vertex_count = out->face_count = out->flags = out->reserved = 0;
    auto fail = [&](const std::string &d) { return har_set_error("Error while loading OBJ file \"" + std::string(filename) + "\": " + d); };
    std::string text;
    if (!read_file(filename, text)) return fail("file not found / unreadable");

    /* ---- pass 1: records */
    ObjSoup soup; soup.poly_first.push_back(0u);
    const char *cursor = text.data(), *const file_end = cursor + text.size();
    while (cursor < file_end) {
        const char *eol = (const char *) memchr(cursor, '\n', (size_t) (file_end - cursor));
        if (!eol) eol = file_end;
        const size_t length = (size_t) (eol - cursor);
        if (length >= 1024) return fail("file contains an excessively long line! (" + std::to_string(length) + " characters)");
        LineScanner L{ cursor, eol };
        auto bad_line = [&]() { return fail("could not parse line \"" + std::string(cursor, length) + "\""); };
        L.skip_blanks();
        if (L.tag("v")) {
            float q[3];
            if (!L.real(q[0]) || !L.real(q[1]) || !L.real(q[2])) return bad_line();
            if (!finite_(q[0]) || !finite_(q[1]) || !finite_(q[2])) return fail("mesh contains invalid vertex position data");
            soup.xyz.insert(soup.xyz.end(), q, q + 3);
        } else if (L.tag("vn")) {
            if (!face_normals) {
                float q[3];
                if (!L.real(q[0]) || !L.real(q[1]) || !L.real(q[2])) return bad_line();
                if (!finite_(q[0]) || !finite_(q[1]) || !finite_(q[2])) return fail("mesh contains invalid vertex normal data");
                soup.nrm.insert(soup.nrm.end(), q, q + 3);
            }
        } else if (L.tag("vt")) {
            float q[2];
            if (!L.real(q[0]) || !L.real(q[1])) return bad_line();
            soup.uv.push_back(q[0]); soup.uv.push_back(flip_tex_coords ? 1.f - q[1] : q[1]);
        } else if (L.tag("f")) {
            for (;;) {
                uint32_t ref[3];
                const CornerStatus st = scan_corner(L, ref);
                if (st == CORNER_NONE) break;
                if (st == CORNER_MALFORMED) return bad_line();
                if (ref[0] == 0 || (size_t) ref[0] > soup.xyz.size() / 3) return fail("reference to invalid vertex " + std::to_string((int32_t) ref[0]) + "!");
                if (ref[1] != 0 && (size_t) ref[1] > soup.uv.size() / 2) return fail("reference to invalid texture coordinate " + std::to_string((int32_t) ref[1]) + "!");
                if (ref[2] != 0 && !face_normals && (size_t) ref[2] > soup.nrm.size() / 3) return fail("reference to invalid normal " + std::to_string((int32_t) ref[2]) + "!");
                soup.c_point.push_back(ref[0] - 1u);
                soup.c_uv.push_back(ref[1] ? ref[1] - 1u : kMissing);
                soup.c_nrm.push_back(ref[2] ? ref[2] - 1u : kMissing);
                soup.any_uv = soup.any_uv || ref[1] != 0; soup.any_nrm = soup.any_nrm || ref[2] != 0;
            }
            soup.poly_first.push_back((uint32_t) soup.c_point.size());
        }
        cursor = eol + 1;
    }

    /* ---- pass 2: fans.  tri[3t + j] = the face corner serving corner j of triangle t */
    const bool keep_normals = soup.any_nrm && !face_normals, keep_uv = soup.any_uv, orient_uv = keep_uv && !face_normals;
    std::vector<uint32_t> tri;
    for (size_t f = 0; f + 1 < soup.poly_first.size(); ++f) {
        const uint32_t c0 = soup.poly_first[f], sides = soup.poly_first[f + 1] - c0;
        for (uint32_t k = 2; k < sides; ++k) { tri.push_back(c0); tri.push_back(c0 + k - 1u); tri.push_back(c0 + k); }
    }
    const size_t n_tri = tri.size() / 3, n_points = soup.xyz.size() / 3;

    /* ---- pass 3: variants.  A variant's words: [normal bits x3][uv bits x2][uv orientation]; at least one word so that the table has a key */
    const size_t words = std::max<size_t>(1, (keep_normals ? 3 : 0) + (keep_uv ? 2 : 0) + (orient_uv ? 1 : 0));
    VariantTable table(words, tri.size());
    std::vector<uint32_t> variants_of(n_points, 0u);        /* distinct variants seen per point so far */
    std::vector<uint32_t> rank_of(tri.size());              /* per triangle corner */
    struct Variant { uint32_t point, rank; uint32_t w[6]; };
    std::vector<Variant> fresh_variants;
    for (size_t t = 0; t < n_tri; ++t) {
        uint32_t mirrored = 0;
        if (orient_uv) {
            float a[2], b[2], c[2]; uint32_t bits[2];
            fetch_key(soup.uv, soup.c_uv[tri[3 * t]], 2, bits);     memcpy(a, bits, 8);
            fetch_key(soup.uv, soup.c_uv[tri[3 * t + 1]], 2, bits); memcpy(b, bits, 8);
            fetch_key(soup.uv, soup.c_uv[tri[3 * t + 2]], 2, bits); memcpy(c, bits, 8);
            const float signed_area = (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0]);
            mirrored = signed_area > 0.f ? 0u : 1u;
        }
        for (int j = 0; j < 3; ++j) {
            const uint32_t fc = tri[3 * t + j], point = soup.c_point[fc];
            uint32_t w[6] = { 0, 0, 0, 0, 0, 0 }; size_t n = 0;
            if (keep_normals) { fetch_key(soup.nrm, soup.c_nrm[fc], 3, w + n); n += 3; }
            if (keep_uv) { fetch_key(soup.uv, soup.c_uv[fc], 2, w + n); n += 2; }
            if (orient_uv) w[n++] = mirrored;
            bool fresh = false;
            const uint32_t rank = table.find_or_add(point, w, variants_of[point], fresh);
            if (fresh) { Variant v; v.point = point; v.rank = rank; memcpy(v.w, w, sizeof(w)); fresh_variants.push_back(v); variants_of[point]++; }
            rank_of[3 * t + j] = rank;
        }
    }

    /* ---- pass 4: numbering.  first_vertex[point] = vertices of all earlier points; compact ids for the points that are in use */
    std::vector<uint32_t> first_vertex(n_points + 1, 0u), compact_point(n_points, kMissing);
    uint32_t points_in_use = 0;
    for (size_t p = 0; p < n_points; ++p) {
        first_vertex[p + 1] = first_vertex[p] + variants_of[p];
        if (variants_of[p]) compact_point[p] = points_in_use++;
    }
    const uint32_t n_vertices = first_vertex[n_points];
    std::vector<float> V((size_t) n_vertices * 8, 0.f);
    std::vector<uint32_t> surface_point(n_vertices, 0u);
    for (const Variant &v : fresh_variants) {
        const uint32_t id = first_vertex[v.point] + v.rank;
        float *rec = V.data() + (size_t) id * 8;
        memcpy(rec, soup.xyz.data() + (size_t) v.point * 3, 12);
        size_t n = 0;
        if (keep_normals) { memcpy(rec + 3, v.w + n, 12); n += 3; }
        if (keep_uv) { memcpy(rec + 6, v.w + n, 8); n += 2; }
        surface_point[id] = compact_point[v.point];
    }
    std::vector<uint32_t> F(4 * n_tri, 0u);
    for (size_t t = 0; t < n_tri; ++t)
        for (int j = 0; j < 3; ++j) F[4 * t + j] = first_vertex[soup.c_point[tri[3 * t + j]]] + rank_of[3 * t + j];

    const bool regenerate = !keep_normals && !face_normals;
    if (har_mesh_finalize(V, F, keep_normals, regenerate, to_world, flip_normals != 0, &surface_point, points_in_use)) return 1;
    return emit(V, F, ((keep_normals || regenerate) ? 1u : 0u) | (keep_uv ? 2u : 0u), out);
}

static int mesh_load_serialized_impl(const char *filename, int shape_index, int face_normals, const float *to_world, int flip_normals, HarMeshData *out) {
    if (!filename || !out) return har_set_error("null argument");
    out->vertices = nullptr; out->faces = nullptr; out->vertex_count = out->face_count = out->flags = out->reserved = 0;
    auto fail = [&](const std::string &d) { return har_set_error("Error while loading serialized file \"" + std::string(filename) + "\": " + d + "!"); };
    if (shape_index < 0) return fail("shape index must be nonnegative");
    std::string data;
    if (!read_file(filename, data)) return fail("file not found / unreadable");
    auto rd = [&](size_t off, void *dst, size_t n) { if (off > data.size() || n > data.size() - off) return false; memcpy(dst, data.data() + off, n); return true; };
    uint16_t format = 0, version = 0;
    if (!rd(0, &format, 2) || !rd(2, &version, 2)) return fail("unexpected end of file");
    if (format != 0x041C) return fail("encountered an invalid file format");
    if (version != 3 && version != 4 && version != 5) return fail("encountered an incompatible file version");
    size_t offset = 0;
    if (shape_index != 0) {                                  /* sub-mesh directory at the end of the file (serialized.cpp:264-297) */
        uint32_t count = 0;
        if (data.size() < 8 || !rd(data.size() - 4, &count, 4)) return fail("unexpected end of file");
        if (shape_index >= (int) count)
            return fail("Unable to unserialize mesh, shape index is out of range! (requested " + std::to_string(shape_index) + " out of 0.." + std::to_string((int) count - 1) + ")");
        if (version >= 4) { uint64_t o = 0; if (data.size() < 8ull * (count - shape_index) + 4 || !rd(data.size() - 8ull * (count - shape_index) - 4, &o, 8)) return fail("unexpected end of file"); offset = (size_t) o; }
        else { uint32_t o = 0; if (data.size() < 4ull * (count - shape_index + 1) || !rd(data.size() - 4ull * (count - shape_index + 1), &o, 4)) return fail("unexpected end of file"); offset = o; }
        if (offset + 4 > data.size()) return fail("invalid sub-mesh offset");
    }
    /* ZStream: one deflate stream per sub-mesh, right after its 4-byte header */
    std::vector<uint8_t> raw;
    {
        z_stream zs; memset(&zs, 0, sizeof(zs));
        if (inflateInit(&zs) != Z_OK) return fail("inflateInit(): failed");
        zs.next_in = (Bytef *) (data.data() + offset + 4); zs.avail_in = (uInt) std::min<size_t>(data.size() - offset - 4, 0xffffffffu);
        uint8_t chunk[1 << 16]; int rc = Z_OK;
        while (rc != Z_STREAM_END) {
            zs.next_out = chunk; zs.avail_out = sizeof(chunk);
            rc = inflate(&zs, Z_NO_FLUSH);
            if (rc != Z_OK && rc != Z_STREAM_END) { inflateEnd(&zs); return fail("inflate(): stream error"); }
            raw.insert(raw.end(), chunk, chunk + (sizeof(chunk) - zs.avail_out));
            if (raw.size() > ((size_t) 1 << 36)) { inflateEnd(&zs); return fail("inflate(): stream larger than 64 GiB"); }
            if (rc == Z_OK && zs.avail_in == 0 && zs.avail_out != 0) { inflateEnd(&zs); return fail("inflate(): unexpected end of file"); }
        }
        inflateEnd(&zs);
    }
    size_t pos = 0;
    auto take = [&](void *dst, size_t n) { if (n > raw.size() - std::min(pos, raw.size()) || pos > raw.size()) return false; if (dst) memcpy(dst, raw.data() + pos, n); pos += n; return true; };
    uint32_t flags = 0;
    if (!take(&flags, 4)) return fail("unexpected end of stream");
    if (version == 5) {
        /* SerializedMesh::load_v5 (serialized.cpp:393-450; writer Mesh::write_serialized, mesh.cpp:1091-1148): the packed representation verbatim -- flag word
         * (low bits = Layout: 1 normals, 2 tangents, 4 texcoords, 8 per-face BSDFs; 0x10 face normals; 0x1000 single precision), length-prefixed name, four
         * 64-bit counts, then the 8-float vertex records and 4-word face records AS THEY ARE, the optional vertex -> surface point / normal group maps and the
         * custom attributes; `to_world` / `flip_normals` are applied afterwards (PackedMesh::transform_records, mesh_utils.cpp:46-99). */
        auto take_string = [&](std::string &out) { uint32_t len = 0; if (!take(&len, 4) || len > raw.size() - pos) return false; out.assign((const char *) raw.data() + pos, len); pos += len; return true; };
        std::string name;
        if (!take_string(name)) return fail("unexpected end of stream");
        auto fail5 = [&](const std::string &d) { return har_set_error("\"" + name + "\": " + d); };
        if (!(flags & 0x1000u)) return fail5("version 5 serialized meshes are stored in single precision.");
        const uint32_t layout = flags & 0xfu;
        const bool l_normals = layout & 1u, l_tangents = layout & 2u, l_texcoords = layout & 4u;
        uint64_t vc = 0, fc = 0, pc = 0, nc = 0;
        if (!take(&vc, 8) || !take(&fc, 8) || !take(&pc, 8) || !take(&nc, 8)) return fail("unexpected end of stream");
        if (pc > vc || nc > vc || (l_tangents && !l_normals)) return fail5("invalid serialized mesh header.");
        if (layout & 8u) return fail5("per-face BSDF indices (Layout::FaceBSDFs) are not implemented by hip_ad_rgb");
        if (vc > 0xffffffffull || fc > 0xffffffffull) return fail("mesh too large");
        if (vc > (raw.size() - pos) / 32 || fc > (raw.size() - pos) / 16) return fail("unexpected end of stream");
        std::vector<float> V(8 * (size_t) vc); std::vector<uint32_t> F(4 * (size_t) fc), pidx;
        if (!take(V.data(), V.size() * 4) || !take(F.data(), F.size() * 4)) return fail("unexpected end of stream");
        if (pc) { pidx.resize((size_t) vc); if (!take(pidx.data(), pidx.size() * 4)) return fail("unexpected end of stream"); for (uint32_t g : pidx) if (g >= pc) return fail5("invalid serialized mesh header."); }
        if (nc && !take(nullptr, (size_t) vc * 4)) return fail("unexpected end of stream");
        uint32_t attr_count = 0;
        if (!take(&attr_count, 4)) return fail("unexpected end of stream");
        for (uint32_t a = 0; a < attr_count; ++a) {          /* custom attributes are read past: nothing on the hip_ad_rgb path looks them up */
            std::string an; uint8_t aflags = 0; uint32_t dim = 0;
            if (!take_string(an) || !take(&aflags, 1) || !take(&dim, 4)) return fail("unexpected end of stream");
            const uint64_t rows = an.compare(0, 5, "face_") == 0 ? fc : vc;
            if (dim != 0 && rows > (raw.size() - pos) / 4 / dim) return fail("unexpected end of stream");
            pos += (size_t) (rows * dim * 4);
        }
        for (size_t i = 0; i < (size_t) fc; ++i) for (int k = 0; k < 3; ++k) if (F[4 * i + k] >= vc) return fail("face index out of bounds");
        /* the stored FaceNormals flag applies when the scene description leaves the property unset (face_normals < 0) */
        const bool fn = face_normals < 0 ? (flags & 0x10u) != 0 : face_normals != 0;
        const bool store_normals = l_normals && !fn;
        if (l_tangents && store_normals)         /* frame_decode (mesh_utils.h:92-117): the normal of the packed (normal, tangent) frame; hip_ad_rgb keeps no tangents */
            for (size_t v = 0; v < (size_t) vc; ++v) {
                float *r = V.data() + 8 * v; const float px = r[3], py = r[4], pz = r[5];
                const float a = 1.f / (1.f + fmaf(pz, pz, fmaf(py, py, px * px))), A = 8.f * (a * a), B = fmaf(-4.f, a, A), X = A * px, Y = A * py, Z = A * pz, yy = py * Y, xz = px * Z, yz = py * Z, u = 1.f - yy;
                r[3] = fmaf(B, py, xz); r[4] = fmaf(-B, px, yz); r[5] = fmaf(-px, X, u);
            }
        if (!store_normals) for (size_t v = 0; v < (size_t) vc; ++v) { float *r = V.data() + 8 * v; r[3] = r[4] = r[5] = 0.f; }
        if (!l_texcoords) for (size_t v = 0; v < (size_t) vc; ++v) { float *r = V.data() + 8 * v; r[6] = r[7] = 0.f; }
        const bool regenerate = !store_normals && !fn;
        if (har_mesh_finalize(V, F, store_normals, regenerate, to_world, flip_normals != 0, pc ? &pidx : nullptr, (uint32_t) (pc ? pc : vc), true)) return 1;
        return emit(V, F, ((store_normals || regenerate) ? 1u : 0u) | (l_texcoords ? 2u : 0u), out);
    }
    if (face_normals < 0) face_normals = 0;
    if (version == 4) { while (true) { char ch; if (!take(&ch, 1)) return fail("unexpected end of stream"); if (!ch) break; } }
    uint64_t vertex_count = 0, face_count = 0;
    if (!take(&vertex_count, 8) || !take(&face_count, 8)) return fail("unexpected end of stream");
    if (vertex_count > 0xffffffffull || face_count > 0xffffffffull) return fail("mesh too large");
    /* the counts are bounded by what the inflated stream holds (positions: >= 12 bytes per vertex, indices: >= 12 bytes per face) before anything is allocated */
    if (vertex_count > (raw.size() - pos) / 12 || face_count > (raw.size() - pos) / 12) return fail("unexpected end of stream");
    const bool dp = flags & 0x2000u, has_normals = flags & 0x0001u, has_texcoords = flags & 0x0002u, has_colors = flags & 0x0008u;
    const bool store_normals = has_normals && !face_normals;
    auto read_array = [&](float *dst, size_t dim, size_t stride) {           /* read_helper: doubles are narrowed, null dst = skip */
        size_t count = (size_t) vertex_count * dim, esz = dp ? 8 : 4;
        if (pos + count * esz > raw.size()) return false;
        if (dst) for (size_t i = 0; i < (size_t) vertex_count; ++i) for (size_t k = 0; k < dim; ++k) {
            const uint8_t *src = raw.data() + pos + (i * dim + k) * esz;
            float v; if (dp) { double d; memcpy(&d, src, 8); v = (float) d; } else memcpy(&v, src, 4);
            dst[i * stride + k] = v;
        }
        pos += count * esz; return true;
    };
    std::vector<float> V(8 * (size_t) vertex_count, 0.f); std::vector<uint32_t> F(4 * (size_t) face_count, 0u);
    if (!read_array(V.data(), 3, 8)) return fail("unexpected end of stream");
    if (has_normals && !read_array(store_normals ? V.data() + 3 : nullptr, 3, 8)) return fail("unexpected end of stream");
    if (has_texcoords && !read_array(V.data() + 6, 2, 8)) return fail("unexpected end of stream");
    if (has_colors && !read_array(nullptr, 3, 8)) return fail("unexpected end of stream");
    for (size_t i = 0; i < (size_t) face_count; ++i) {
        uint32_t fi[3]; if (!take(fi, 12)) return fail("unexpected end of stream");
        for (int k = 0; k < 3; ++k) { if (fi[k] >= vertex_count) return fail("face index out of bounds"); F[4 * i + k] = fi[k]; }
    }
    const bool regenerate = !store_normals && !face_normals;
    if (har_mesh_finalize(V, F, store_normals, regenerate, to_world, flip_normals != 0, nullptr, 0)) return 1;
    return emit(V, F, ((store_normals || regenerate) ? 1u : 0u) | (has_texcoords ? 2u : 0u), out);
}

} // extern "C"
