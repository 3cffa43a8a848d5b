/*
 * har_mesh_io.cpp -- host-side mesh ingestion for the hip_ad_rgb path.
 *
 *   har_mesh_load_ply      PLYMesh ctor (src/shapes/ply.cpp:113-345) + parse_ply_header / parse_ascii (src/shapes/ply.h):
 *                          ASCII, binary little- and big-endian; x/y/z [+ nx/ny/nz] [+ u/v | texture_u/texture_v | s/t] of any
 *                          scalar type, converted to f32; triangle faces (`vertex_indices` / `vertex_index` lists of 3)
 *   har_mesh_compute_normals  Mesh::compute_normals (src/render/mesh.cpp:1218-1267): angle-weighted vertex normals
 *                          (Thuermer & Wuethrich 1998) for meshes without stored normals unless `face_normals`
 * The result is the packed layout of include/mitsuba/render/mesh_utils.h:19-34 (8 f32 per vertex, 4 u32 per face) that
 * har_scene_create ingests.  Extra per-vertex / per-face properties (colours, ...) are parsed and skipped.
 */
#include "../../include/hip_ad_rgb.h"
#include "har_math.h"
#include "har_vertex_update.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <new>
#include <stdexcept>
#include <vector>

using namespace har;

extern int har_set_error(const std::string &msg);      /* har_capi.hip: thread-local message, returns 1 */

namespace {

enum PType { I8, U8, I16, U16, I32, U32, F32, F64, PInvalid };
struct Prop { std::string name; PType type; bool is_list; PType count_type; };
struct Element { std::string name; size_t count; std::vector<Prop> props; };

PType parse_type(const std::string &t) {
    if (t == "char" || t == "int8") return I8;      if (t == "uchar" || t == "uint8") return U8;
    if (t == "short" || t == "int16") return I16;   if (t == "ushort" || t == "uint16") return U16;
    if (t == "int" || t == "int32") return I32;     if (t == "uint" || t == "uint32") return U32;
    if (t == "float" || t == "float32") return F32; if (t == "double" || t == "float64") return F64;
    return PInvalid;
}
size_t type_size(PType t) { static const size_t s[] = { 1, 1, 2, 2, 4, 4, 4, 8, 0 }; return s[t]; }

struct Reader {
    const uint8_t *p, *end; bool ascii, swap;
    std::istringstream text;
    bool fail = false;
    double read(PType t) {
        if (ascii) {                                         /* strtod: accepts nan / inf like the reference's parser */
            std::string tok; if (!(text >> tok)) { fail = true; return 0; }
            char *endp = nullptr; double v = strtod(tok.c_str(), &endp);
            if (endp == tok.c_str() || *endp != '\0') { fail = true; return 0; }
            return v;
        }
        size_t n = type_size(t);
        if ((size_t) (end - p) < n) { fail = true; return 0; }
        uint8_t b[8]; memcpy(b, p, n); p += n;
        if (swap) for (size_t i = 0; i < n / 2; ++i) std::swap(b[i], b[n - 1 - i]);
        switch (t) {
            case I8: { int8_t v; memcpy(&v, b, 1); return v; }     case U8: { uint8_t v; memcpy(&v, b, 1); return v; }
            case I16: { int16_t v; memcpy(&v, b, 2); return v; }   case U16: { uint16_t v; memcpy(&v, b, 2); return v; }
            case I32: { int32_t v; memcpy(&v, b, 4); return v; }   case U32: { uint32_t v; memcpy(&v, b, 4); return v; }
            case F32: { float v; memcpy(&v, b, 4); return v; }     default: { double v; memcpy(&v, b, 8); return v; }
        }
    }
};

} // namespace

extern "C" {

int har_mesh_compute_normals(uint32_t vertex_count, float *vertices, uint32_t face_count, const uint32_t *faces) {
    if (vertex_count == 0 && face_count == 0) return 0;
    if (!vertices || (!faces && face_count)) return har_set_error("null mesh buffers");
    std::vector<float> acc(3 * (size_t) vertex_count, 0.f);
    for (uint32_t f = 0; f < face_count; ++f) {
        uint32_t fi[3] = { faces[4 * (size_t) f], faces[4 * (size_t) f + 1], faces[4 * (size_t) f + 2] };
        Vec3 p[3];
        for (int k = 0; k < 3; ++k) { if (fi[k] >= vertex_count) return har_set_error("face index out of bounds"); const float *v = vertices + 8 * (size_t) fi[k]; p[k] = Vec3(v[0], v[1], v[2]); }
        for (int k = 0; k < 3; ++k) {
            Vec3 t;
            if (!mesh_corner_term(p, k, t)) break;          /* a face without area adds nothing */
            float *a = acc.data() + 3 * (size_t) fi[k];
            a[0] += t.x; a[1] += t.y; a[2] += t.z;
        }
    }
    for (uint32_t v = 0; v < vertex_count; ++v) {
        Vec3 n(acc[3 * (size_t) v], acc[3 * (size_t) v + 1], acc[3 * (size_t) v + 2]);
        float length_sqr = dot3(n, n);
        n = length_sqr > 0.f ? n * rsqrt_(length_sqr) : Vec3(1.f, 0.f, 0.f);
        float *o = vertices + 8 * (size_t) v; o[3] = n.x; o[4] = n.y; o[5] = n.z;
    }
    return 0;
}

void har_mesh_free(HarMeshData *m) {
    if (!m) return;
    free(m->vertices); free(m->faces);
    m->vertices = nullptr; m->faces = nullptr; m->vertex_count = m->face_count = 0;
}

} // extern "C"

/* What the reference does between parsing a mesh file and Mesh::from_packed (PackedMesh::set_transform / set_vertex,
 * src/render/mesh_utils.cpp:33-44,101-133): positions take `to_world`, stored normals its inverse transpose and are normalised
 * (left alone when their length is 0 / not finite) and negated under `flip_normals`, the winding is reversed when
 * det(to_world) < 0 XOR flip_normals; normals missing from the file are regenerated AFTERWARDS, from the transformed positions
 * and the final winding (Mesh::pack, mesh.cpp:573-582,618-619), one per surface point when `position_index` splits vertices. */
int har_mesh_finalize(std::vector<float> &V, std::vector<uint32_t> &F, bool stored_normals, bool regenerate, const float *to_world32,
                      bool flip_normals, const std::vector<uint32_t> *position_index, uint32_t position_count, bool packed_records) {
    const size_t nv = V.size() / 8, nf = F.size() / 4;
    float m[3][4] = { { 1, 0, 0, 0 }, { 0, 1, 0, 0 }, { 0, 0, 1, 0 } }, it[3][3] = { { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } };
    bool transform = false;
    if (to_world32) {            /* har_transform_* layout: matrix (row major 4 x 4) followed by its inverse transpose */
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) { m[r][c] = to_world32[4 * r + c]; transform |= m[r][c] != (r == c ? 1.f : 0.f); }
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) it[r][c] = to_world32[16 + 4 * r + c];
    }
    const float det = m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
                      m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
    for (size_t i = 0; i < nv; ++i) {
        float *r = V.data() + 8 * i;
        if (transform) {         /* same operation order as Transform4f::point / ::normal in har_host.cpp */
            Vec3 p(r[0], r[1], r[2]), q(m[0][3], m[1][3], m[2][3]);
            q = Vec3(fma_(m[0][0], p.x, q.x), fma_(m[1][0], p.x, q.y), fma_(m[2][0], p.x, q.z));
            q = Vec3(fma_(m[0][1], p.y, q.x), fma_(m[1][1], p.y, q.y), fma_(m[2][1], p.y, q.z));
            q = Vec3(fma_(m[0][2], p.z, q.x), fma_(m[1][2], p.z, q.y), fma_(m[2][2], p.z, q.z));
            r[0] = q.x; r[1] = q.y; r[2] = q.z;
        }
        if (stored_normals) {
            Vec3 n(r[3], r[4], r[5]);
            if (transform) {
                Vec3 q(it[0][0] * n.x, it[1][0] * n.x, it[2][0] * n.x);
                q = Vec3(fma_(it[0][1], n.y, q.x), fma_(it[1][1], n.y, q.y), fma_(it[2][1], n.y, q.z));
                n = Vec3(fma_(it[0][2], n.z, q.x), fma_(it[1][2], n.z, q.y), fma_(it[2][2], n.z, q.z));
            }
            /* packed_records (serialized v5, PackedMesh::transform_records, mesh_utils.cpp:71-76): the stored normal is kept as it is unless a transform is applied,
             * and then dr::normalize()d without a guard; parsed files (set_vertex, :113-115) always normalise and leave zero / non-finite lengths alone */
            if (packed_records) { if (transform) n = n * rsqrt_(dot3(n, n)); }
            else { float il = rsqrt_(dot3(n, n)); if (finite_(il)) n = n * il; }
            if (flip_normals) n = Vec3(-n.x, -n.y, -n.z);
            r[3] = n.x; r[4] = n.y; r[5] = n.z;
        }
    }
    if ((det < 0.f) != flip_normals)
        for (size_t f = 0; f < nf; ++f) std::swap(F[4 * f], F[4 * f + 2]);
    if (regenerate) {
        if (position_index && position_count != nv) {
            std::vector<float> P(8 * (size_t) position_count, 0.f); std::vector<uint32_t> G(F.size(), 0u);
            for (size_t v = 0; v < nv; ++v) memcpy(P.data() + 8 * (size_t) (*position_index)[v], V.data() + 8 * v, 12);
            for (size_t i = 0; i < F.size(); ++i) if ((i & 3) != 3) G[i] = (*position_index)[F[i]];
            if (har_mesh_compute_normals(position_count, P.data(), (uint32_t) nf, G.data())) return 1;
            for (size_t v = 0; v < nv; ++v) memcpy(V.data() + 8 * v + 3, P.data() + 8 * (size_t) (*position_index)[v] + 3, 12);
        } else if (har_mesh_compute_normals((uint32_t) nv, V.data(), (uint32_t) nf, F.data())) return 1;
    }
    return 0;
}

extern "C" {

static int mesh_load_ply_impl(const char *filename, int face_normals, int flip_tex_coords, const float *to_world, int flip_normals, HarMeshData *out);
/* file contents are untrusted: element counts are bounded by the bytes that are left in the file and by the 2^32 - 1 indices of the packed layout,
 * and no C++ exception (std::bad_alloc of a header-controlled vector size, ...) crosses the C boundary */
int har_mesh_load_ply(const char *filename, int face_normals, int flip_tex_coords, const float *to_world, int flip_normals, HarMeshData *out) {
    try { return mesh_load_ply_impl(filename, face_normals, flip_tex_coords, to_world, flip_normals, out); }
    catch (const std::bad_alloc &) { if (out) har_mesh_free(out); return har_set_error(std::string("Error while loading PLY file \"") + (filename ? filename : "") + "\": out of memory!"); }
    catch (const std::exception &e) { if (out) har_mesh_free(out); return har_set_error(std::string("Error while loading PLY file \"") + (filename ? filename : "") + "\": " + e.what() + "!"); }
}
static int mesh_load_ply_impl(const char *filename, int face_normals, int flip_tex_coords, const float *to_world, int flip_normals, HarMeshData *out) {
    if (!filename || !out) return har_set_error("null argument");
    memset(out, 0, sizeof(*out));
    auto fail = [&](const std::string &d) { har_mesh_free(out); return har_set_error("Error while loading PLY file \"" + std::string(filename) + "\": " + d + "!"); };
    std::ifstream f(filename, std::ios::binary);
    if (!f) return fail("file not found");
    std::string data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    /* ---- header (parse_ply_header, src/shapes/ply.h) */
    size_t pos = 0; std::string line; bool first = true, have_format = false, ascii = false, big = false, ended = false;
    std::vector<Element> elements;
    auto next_line = [&]() { size_t e = data.find('\n', pos); if (e == std::string::npos) return false; line = data.substr(pos, e - pos); pos = e + 1;
                             if (!line.empty() && line.back() == '\r') line.pop_back(); return true; };
    while (next_line()) {
        std::istringstream ls(line); std::string tok; ls >> tok;
        if (first) { if (tok != "ply") return fail("invalid PLY header"); first = false; continue; }
        if (tok == "format") {
            std::string fmt, ver; ls >> fmt >> ver;
            if (fmt == "ascii") ascii = true; else if (fmt == "binary_little_endian") big = false; else if (fmt == "binary_big_endian") big = true;
            else return fail("invalid PLY header: unknown format");
            if (ver != "1.0") return fail("PLY file has unknown version number \"" + ver + "\"");
            have_format = true;
        } else if (tok == "comment" || tok == "obj_info" || tok.empty()) {
        } else if (tok == "element") {
            Element el; ls >> el.name >> el.count; if (!ls) return fail("invalid PLY header: \"element\" line");
            elements.push_back(el);
        } else if (tok == "property") {
            if (elements.empty()) return fail("invalid PLY header: \"property\" before \"element\"");
            Prop p; std::string t; ls >> t;
            if (t == "list") { std::string ct, it; ls >> ct >> it >> p.name; p.is_list = true; p.count_type = parse_type(ct); p.type = parse_type(it); if (p.count_type == PInvalid) return fail("invalid PLY header: unknown list count type"); }
            else { p.is_list = false; p.count_type = PInvalid; p.type = parse_type(t); ls >> p.name; }
            if (p.type == PInvalid || p.name.empty()) return fail("invalid PLY header: unknown property type \"" + t + "\"");
            elements.back().props.push_back(p);
        } else if (tok == "end_header") { ended = true; break; }
        else return fail("invalid PLY header: unknown token \"" + tok + "\"");
    }
    if (!ended || !have_format) return fail("invalid PLY header");
    Reader R; R.p = (const uint8_t *) data.data() + pos; R.end = (const uint8_t *) data.data() + data.size(); R.ascii = ascii;
    { uint16_t one = 1; bool host_little = *(uint8_t *) &one == 1; R.swap = !ascii && (big == host_little); }
    if (ascii) R.text.str(data.substr(pos));
    bool has_normals = false, has_uv = false;
    std::vector<float> V; std::vector<uint32_t> F; size_t nv = 0, nf = 0;
    for (const Element &el : elements) {
        /* every element occupies at least one byte (binary) / two characters (ascii) per property value */
        const size_t bytes_left = ascii ? data.size() - pos : (size_t) (R.end - R.p);
        if (el.count > 0xffffffffull || (!el.props.empty() && el.count > bytes_left)) return fail("element count exceeds the file size");
        if (el.name == "vertex") {
            int ix[8] = { -1, -1, -1, -1, -1, -1, -1, -1 };       /* x y z nx ny nz u v */
            for (size_t k = 0; k < el.props.size(); ++k) {
                const std::string &n = el.props[k].name;
                if (el.props[k].is_list) return fail("incompatible contents -- vertex element with a list property");
                if (n == "x") ix[0] = (int) k; else if (n == "y") ix[1] = (int) k; else if (n == "z") ix[2] = (int) k;
                else if (n == "nx") ix[3] = (int) k; else if (n == "ny") ix[4] = (int) k; else if (n == "nz") ix[5] = (int) k;
            }
            auto find2 = [&](const char *a, const char *b) { int ia = -1, ib = -1; for (size_t k = 0; k < el.props.size(); ++k) { if (el.props[k].name == a) ia = (int) k; if (el.props[k].name == b) ib = (int) k; }
                                                             if (ia >= 0 && ib >= 0 && ix[6] < 0) { ix[6] = ia; ix[7] = ib; } };
            find2("u", "v"); find2("texture_u", "texture_v"); find2("s", "t");
            if (ix[0] < 0 || ix[1] < 0 || ix[2] < 0) return fail("vertex positions (x, y, z) not found");
            has_normals = !face_normals && ix[3] >= 0 && ix[4] >= 0 && ix[5] >= 0;
            has_uv = ix[6] >= 0;
            nv = el.count; V.assign(8 * nv, 0.f);
            std::vector<double> rec(el.props.size());
            for (size_t i = 0; i < nv; ++i) {
                for (size_t k = 0; k < el.props.size(); ++k) rec[k] = R.read(el.props[k].type);
                if (R.fail) return fail("unexpected end of file");
                float *o = V.data() + 8 * i;
                for (int c = 0; c < 3; ++c) { o[c] = (float) rec[ix[c]]; if (!finite_(o[c])) return fail("mesh contains invalid vertex position data"); }
                if (has_normals) for (int c = 0; c < 3; ++c) o[3 + c] = (float) rec[ix[3 + c]];
                if (has_uv) { o[6] = (float) rec[ix[6]]; o[7] = (float) rec[ix[7]]; if (flip_tex_coords) o[7] = 1.f - o[7]; }
            }
        } else if (el.name == "face") {
            int il = -1;
            for (size_t k = 0; k < el.props.size(); ++k) if (el.props[k].is_list && (el.props[k].name == "vertex_index" || el.props[k].name == "vertex_indices")) il = (int) k;
            if (il < 0) return fail("vertex_index/vertex_indices property not found");
            nf = el.count; F.assign(4 * nf, 0u);
            for (size_t i = 0; i < nf; ++i) {
                for (size_t k = 0; k < el.props.size(); ++k) {
                    const Prop &p = el.props[k];
                    if (!p.is_list) { (void) R.read(p.type); continue; }
                    double cnt = R.read(p.count_type);
                    if ((int) k == il) {
                        if (R.fail || cnt != 3.0) return fail("incompatible contents -- is this a triangle mesh?");
                        for (int c = 0; c < 3; ++c) { double v = R.read(p.type); if (v < 0 || v > 4294967295.0) return fail("invalid face index"); F[4 * i + c] = (uint32_t) v; }
                    } else for (int c = 0; c < (int) cnt; ++c) (void) R.read(p.type);
                }
                if (R.fail) return fail("unexpected end of file");
            }
        } else {
            for (const Prop &p : el.props) if (p.is_list) return fail("cannot skip unknown element \"" + el.name + "\" with list properties");
            for (size_t i = 0; i < el.count; ++i) for (const Prop &p : el.props) (void) R.read(p.type);
            if (R.fail) return fail("unexpected end of file");
        }
    }
    if (!ascii && R.p != R.end) return fail("invalid file -- trailing content");
    if (ascii) { std::string rest; if (R.text >> rest) return fail("invalid file -- trailing content"); }
    for (size_t i = 0; i < nf; ++i) for (int c = 0; c < 3; ++c) if (F[4 * i + c] >= nv) return fail("face index out of bounds");
    const bool regenerate = !has_normals && !face_normals;       /* Mesh::from_packed -> pack(regenerate_normals = true), mesh.cpp:355-356 */
    if (har_mesh_finalize(V, F, has_normals, regenerate, to_world, flip_normals != 0, nullptr, 0, false)) return 1;
    out->vertices = (float *) malloc(std::max<size_t>(V.size(), 1) * sizeof(float));
    out->faces = (uint32_t *) malloc(std::max<size_t>(F.size(), 1) * sizeof(uint32_t));
    if (!out->vertices || !out->faces) return fail("out of memory");
    memcpy(out->vertices, V.data(), V.size() * sizeof(float)); memcpy(out->faces, F.data(), F.size() * sizeof(uint32_t));
    out->vertex_count = (uint32_t) nv; out->face_count = (uint32_t) nf;
    out->flags = ((has_normals || regenerate) ? 1u : 0u) | (has_uv ? 2u : 0u);
    return 0;
}

} // extern "C"
