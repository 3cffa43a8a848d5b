/*
 * har_mesh_formats.cpp -- OBJ and `serialized` mesh files for the hip_ad_rgb path (SURVEY.md 8f rank 2, next to har_mesh_io.cpp's PLY).
 *
 *   har_mesh_load_obj         OBJMesh ctor (src/shapes/obj.cpp:98-296): v / vn / vt / f records, 1-based `p`, `p/t`, `p//n`, `p/t/n`
 *                             corners, polygons of any size; then Mesh::from_corners -> corner_to_packed_mesh
 *                             (src/render/mesh_utils.cpp:210-560): fan triangulation around corner 0, corners of one source point
 *                             weld when normal, texcoord and the sign of the triangle's UV area agree, vertex ids follow the source
 *                             point order, unreferenced points are dropped.  Missing normals are regenerated per SURFACE POINT
 *                             (Mesh::pack: normal_index = position_index, src/render/mesh.cpp:573-582), i.e. UV seams stay smooth.
 *   har_mesh_load_serialized  SerializedMesh ctor + load_legacy (src/shapes/serialized.cpp:225-370): Mitsuba 0.x / 2 / 3 `.serialized`
 *                             container versions 3 and 4 (zlib stream per sub-mesh, offset table at the end of the file).
 * Output: the packed layout har_scene_create ingests (8 f32 per vertex, 4 u32 per face), malloc'ed.
 */
#include "../../include/hip_ad_rgb.h"
#include "har_math.h"

#include <zlib.h>

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
                             bool flip_normals, const std::vector<uint32_t> *position_index, uint32_t position_count);     /* har_mesh_io.cpp */

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

/* bit pattern of one attribute record, -0.0 folded to +0.0 so that equal values compare equal (mesh_utils.cpp:185-207) */
void fetch_key(const std::vector<float> &pool, uint32_t index, int dim, uint32_t *out) {
    if (index == kMissing) memset(out, 0, dim * sizeof(uint32_t));
    else memcpy(out, pool.data() + (size_t) index * dim, dim * sizeof(float));
    for (int k = 0; k < dim; ++k) if (out[k] == 0x80000000u) out[k] = 0;
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
static int mesh_load_obj_impl(const char *filename, int face_normals, int flip_tex_coords, const float *to_world, int flip_normals, HarMeshData *out) {
    if (!filename || !out) return har_set_error("null argument");
    out->vertices = nullptr; out->faces = nullptr; out->vertex_count = out->face_count = out->flags = out->reserved = 0;
    auto fail = [&](const std::string &d) { return har_set_error("Error while loading OBJ file \"" + std::string(filename) + "\": " + d); };
    std::string data;
    if (!read_file(filename, data)) return fail("file not found / unreadable");

    std::vector<float> positions, normals, texcoords;
    std::vector<uint32_t> corner_vertex, corner_uv, corner_normal, face_offsets(1, 0u);
    bool has_uv_indices = false, has_normal_indices = false;
    const char *ptr = data.data(), *eof = ptr + data.size();
    char buf[1025];
    auto is_ws = [](char c) { return c == ' ' || c == '\t'; };
    while (ptr < eof) {
        const char *next = ptr; while (next != eof && *next != '\n') ++next;
        size_t size = (size_t) (next - ptr);
        if (size >= sizeof(buf) - 1) return fail("file contains an excessively long line! (" + std::to_string(size) + " characters)");
        memcpy(buf, ptr, size); buf[size] = '\0';
        const char *cur = buf; while (*cur == ' ' || *cur == '\t' || *cur == '\r') ++cur;
        bool parse_error = false;
        auto floats = [&](int n, float *dst) { for (int i = 0; i < n; ++i) { char *e = nullptr; dst[i] = strtof(cur, &e); parse_error |= e == cur; cur = e; } };
        if (cur[0] == 'v' && is_ws(cur[1])) {
            cur += 2; float p[3]; floats(3, p);
            if (!(finite_(p[0]) && finite_(p[1]) && finite_(p[2]))) return fail("mesh contains invalid vertex position data");
            positions.insert(positions.end(), p, p + 3);
        } else if (cur[0] == 'v' && cur[1] == 'n' && is_ws(cur[2])) {
            if (!face_normals) {
                cur += 3; float n[3]; floats(3, n);
                if (!(finite_(n[0]) && finite_(n[1]) && finite_(n[2]))) return fail("mesh contains invalid vertex normal data");
                normals.insert(normals.end(), n, n + 3);
            }
        } else if (cur[0] == 'v' && cur[1] == 't' && is_ws(cur[2])) {
            cur += 3; float uv[2]; floats(2, uv);
            if (flip_tex_coords) uv[1] = 1.f - uv[1];
            texcoords.insert(texcoords.end(), uv, uv + 2);
        } else if (cur[0] == 'f' && is_ws(cur[1])) {
            cur += 2;
            size_t type_index = 0; uint32_t key[3] = { 0, 0, 0 };
            while (true) {
                char *next2 = nullptr;
                uint32_t value = (uint32_t) strtoul(cur, &next2, 10);
                if (cur == next2) break;
                if (type_index < 3) key[type_index] = value; else { parse_error = true; break; }
                while (*next2 == '/') { type_index++; next2++; }
                if (*next2 == ' ' || *next2 == '\t' || *next2 == '\0' || *next2 == '\r') {
                    type_index = 0;
                    if (key[0] == 0 || (size_t) (key[0] - 1) * 3 >= positions.size()) return fail("reference to invalid vertex " + std::to_string(key[0]) + "!");
                    if (key[1] != 0 && (size_t) (key[1] - 1) * 2 >= texcoords.size()) return fail("reference to invalid texture coordinate " + std::to_string(key[1]) + "!");
                    if (key[2] != 0 && !face_normals && (size_t) (key[2] - 1) * 3 >= normals.size()) return fail("reference to invalid normal " + std::to_string(key[2]) + "!");
                    corner_vertex.push_back(key[0] - 1);
                    corner_uv.push_back(key[1] ? key[1] - 1 : kMissing);
                    corner_normal.push_back(key[2] ? key[2] - 1 : kMissing);
                    has_uv_indices |= key[1] != 0; has_normal_indices |= key[2] != 0;
                    key[1] = key[2] = 0;
                }
                cur = next2;
            }
            face_offsets.push_back((uint32_t) corner_vertex.size());
        }
        if (parse_error) return fail("could not parse line \"" + std::string(buf) + "\"");
        ptr = next + 1;
    }

    /* ---- Mesh::from_corners (mesh_utils.cpp:210-560) */
    const bool has_normals = has_normal_indices && !face_normals, has_uv = has_uv_indices, split_uv_sign = has_uv && !face_normals;
    const size_t n_points = positions.size() / 3;
    std::vector<uint32_t> tri_corner;                        /* triangle corner -> face corner: fan around corner 0 */
    for (size_t f = 0; f + 1 < face_offsets.size(); ++f) {
        uint32_t begin = face_offsets[f], n = face_offsets[f + 1] - begin;
        for (uint32_t i = 1; i + 1 < n; ++i) { tri_corner.push_back(begin); tri_corner.push_back(begin + i); tri_corner.push_back(begin + i + 1); }
    }
    const size_t n_tris = tri_corner.size() / 3, n_tc = tri_corner.size();
    std::vector<uint8_t> uv_flipped;
    if (split_uv_sign) {
        uv_flipped.resize(n_tris);
        for (size_t t = 0; t < n_tris; ++t) {
            uint32_t bits[6]; float uv[6];
            for (int j = 0; j < 3; ++j) fetch_key(texcoords, corner_uv[tri_corner[3 * t + j]], 2, bits + 2 * j);
            memcpy(uv, bits, sizeof(uv));
            float area2 = (uv[2] - uv[0]) * (uv[5] - uv[1]) - (uv[3] - uv[1]) * (uv[4] - uv[0]);
            uv_flipped[t] = !(area2 > 0.f);
        }
    }
    size_t vert_dim = (has_normals ? 3 : 0) + (has_uv ? 2 : 0) + (split_uv_sign ? 1 : 0);
    if (vert_dim == 0) vert_dim = 1;
    std::vector<uint32_t> key(vert_dim, 0u);
    auto build_key = [&](uint32_t c) {
        uint32_t sc = tri_corner[c], *k = key.data();
        if (has_normals) { fetch_key(normals, corner_normal[sc], 3, k); k += 3; }
        if (has_uv) { fetch_key(texcoords, corner_uv[sc], 2, k); k += 2; }
        if (split_uv_sign) *k++ = uv_flipped[c / 3];
    };
    /* stable counting sort of the triangle corners by source point */
    std::vector<uint32_t> point_offsets(n_points + 1, 0u), corner_order(n_tc);
    for (size_t c = 0; c < n_tc; ++c) point_offsets[corner_vertex[tri_corner[c]]]++;
    for (size_t p = 1; p <= n_points; ++p) point_offsets[p] += point_offsets[p - 1];
    for (size_t c = n_tc; c-- > 0; ) corner_order[--point_offsets[corner_vertex[tri_corner[c]]]] = (uint32_t) c;

    std::vector<float> V; std::vector<uint32_t> F(4 * n_tris, 0u), position_index, vert_keys;
    uint32_t vertex_count = 0, position_count = 0;
    for (size_t p = 0; p < n_points; ++p) {
        uint32_t begin = point_offsets[p], end = point_offsets[p + 1];
        if (begin == end) continue;                          /* unreferenced source vertices are dropped */
        uint32_t point_id = position_count++, vert_base = vertex_count;
        vert_keys.clear();
        for (uint32_t i = begin; i != end; ++i) {
            uint32_t c = corner_order[i];
            build_key(c);
            uint32_t n_local = vertex_count - vert_base, j = 0;
            while (j < n_local && memcmp(vert_keys.data() + (size_t) j * vert_dim, key.data(), vert_dim * sizeof(uint32_t)) != 0) ++j;
            uint32_t vid = vert_base + j;
            if (vid == vertex_count) {
                vertex_count++;
                position_index.push_back(point_id);
                vert_keys.insert(vert_keys.end(), key.begin(), key.end());
                float rec[8] = { positions[3 * p], positions[3 * p + 1], positions[3 * p + 2], 0.f, 0.f, 0.f, 0.f, 0.f };
                const uint32_t *k = key.data();
                if (has_normals) { memcpy(rec + 3, k, 12); k += 3; }
                if (has_uv) { memcpy(rec + 6, k, 8); k += 2; }
                V.insert(V.end(), rec, rec + 8);
            }
            F[(size_t) (c / 3) * 4 + c % 3] = vid;
        }
    }
    const bool regenerate = !has_normals && !face_normals;
    if (har_mesh_finalize(V, F, has_normals, regenerate, to_world, flip_normals != 0, &position_index, position_count)) return 1;
    return emit(V, F, ((has_normals || regenerate) ? 1u : 0u) | (has_uv ? 2u : 0u), out);
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
    if (version == 5) return fail("version 5 (packed-record) files are not supported by hip_ad_rgb yet; re-export as version 4");
    if (version != 3 && version != 4) return fail("encountered an incompatible file version");
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
    auto take = [&](void *dst, size_t n) { if (pos + n > raw.size()) return false; if (dst) memcpy(dst, raw.data() + pos, n); pos += n; return true; };
    uint32_t flags = 0;
    if (!take(&flags, 4)) return fail("unexpected end of stream");
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
