/*
 * orc_scene.cpp -- ORACLE-side scene builders (test infrastructure only).
 *
 * Independent restatement of the host-side lowering the reference performs
 * when `mi.load_dict(mi.cornell_box())` runs: Transform4f algebra
 * (include/mitsuba/core/transform.h), PerspectiveCamera set-up
 * (src/sensors/perspective.cpp:174-198, include/mitsuba/render/sensor.h:234-269,
 * src/render/sensor.cpp:142-190), Rectangle / Cube records
 * (src/shapes/rectangle.cpp:108-156, src/shapes/cube.cpp:58-113) baked with
 * `to_world` (src/render/mesh.cpp:1160-1215).  Used by tests to cross-check the
 * product's own host code; not used by the product.
 *
 * A transform is 32 floats: row-major 4x4 `matrix` followed by row-major 4x4
 * `inverse_transpose`, exactly the pair the reference's Transform keeps.
 */
#include "mi_oracle.h"
#include "orc_math.h"
#include <cstring>
#include <string>

using namespace orc;

namespace {
struct M4 { float m[16]; float &operator()(int r, int c) { return m[4 * r + c]; } float operator()(int r, int c) const { return m[4 * r + c]; } };
static M4 ident() { M4 r{}; for (int i = 0; i < 4; ++i) r(i, i) = 1.f; return r; }
static M4 transpose(const M4 &a) { M4 r{}; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r(i, j) = a(j, i); return r; }
/* general dr::Matrix product: row_i = a(i,0)*b.row(0), then fma over j */
static M4 matmul(const M4 &a, const M4 &b) {
    M4 r{};
    for (int i = 0; i < 4; ++i) {
        float row[4];
        for (int c = 0; c < 4; ++c) row[c] = a(i, 0) * b(0, c);
        for (int j = 1; j < 4; ++j) for (int c = 0; c < 4; ++c) row[c] = fmadd(a(i, j), b(j, c), row[c]);
        for (int c = 0; c < 4; ++c) r(i, c) = row[c];
    }
    return r;
}
struct Xf { M4 m, it; };
static void store(const Xf &x, float *out) { std::memcpy(out, x.m.m, 64); std::memcpy(out + 16, x.it.m, 64); }
static Xf load(const float *in) { Xf x; std::memcpy(x.m.m, in, 64); std::memcpy(x.it.m, in + 16, 64); return x; }

/* Transform::operator*, affine branch (transform.h:364-400) */
static Xf mul_affine(const Xf &a, const Xf &o) {
    Xf r; r.m = ident(); r.it = ident();
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float sum = 0.f, sum_it = 0.f;
            for (int k = 0; k < 3; ++k) sum = fmadd(a.m(i, k), o.m(k, j), sum);
            for (int k = 0; k < 3; ++k) sum_it = fmadd(a.it(i, k), o.it(k, j), sum_it);
            r.m(i, j) = sum; r.it(i, j) = sum_it;
        }
    for (int l = 0; l < 3; ++l) {
        float sum = a.m(l, 3), sum_it = o.it(3, l);
        for (int k = 0; k < 3; ++k) sum = fmadd(a.m(l, k), o.m(k, 3), sum);
        for (int k = 0; k < 3; ++k) sum_it = fmadd(a.it(3, k), o.it(k, l), sum_it);
        r.m(l, 3) = sum; r.it(3, l) = sum_it;
    }
    return r;
}
static Xf mul_general(const Xf &a, const Xf &b) { Xf r; r.m = matmul(a.m, b.m); r.it = matmul(a.it, b.it); return r; }
static Xf inverse(const Xf &a) { Xf r; r.m = transpose(a.it); r.it = transpose(a.m); return r; }

static Xf translate(V3 v) {          // transform.h:132-135
    Xf r; r.m = ident(); r.it = ident();
    r.m(0, 3) = v.x; r.m(1, 3) = v.y; r.m(2, 3) = v.z;
    r.it(3, 0) = -v.x; r.it(3, 1) = -v.y; r.it(3, 2) = -v.z;
    return r;
}
static Xf scale(V3 v) {              // transform.h:138-140
    Xf r; r.m = ident(); r.it = ident();
    r.m(0, 0) = v.x; r.m(1, 1) = v.y; r.m(2, 2) = v.z;
    r.it(0, 0) = rcp(v.x); r.it(1, 1) = rcp(v.y); r.it(2, 2) = rcp(v.z);
    return r;
}
static Xf rotate(V3 axis, float angle_deg) {   // transform.h:143-147 over dr::rotate (Rodrigues)
    float angle = angle_deg * (Pi / 180.f);
    float c, s = sincos(angle, &c);
    float cm = 1.f - c;
    V3 sh1(axis.y, axis.z, axis.x), sh2(axis.z, axis.x, axis.y);
    V3 t0 = fmadd(axis * axis, cm, V3(c));
    V3 t1 = fmadd(axis * sh1, cm, sh2 * s);
    V3 t2(fmsub(axis.x * sh2.x, cm, sh1.x * s), fmsub(axis.y * sh2.y, cm, sh1.y * s), fmsub(axis.z * sh2.z, cm, sh1.z * s));
    Xf r; r.m = ident();
    r.m(0, 0) = t0.x; r.m(0, 1) = t2.y; r.m(0, 2) = t1.z;
    r.m(1, 0) = t1.x; r.m(1, 1) = t0.y; r.m(1, 2) = t2.z;
    r.m(2, 0) = t2.x; r.m(2, 1) = t1.y; r.m(2, 2) = t0.z;
    r.it = r.m;
    return r;
}
static Xf look_at(V3 origin, V3 target, V3 up) {   // transform.h:175-203
    V3 dir = normalize(target - origin), left = normalize(cross(up, dir)), new_up = cross(dir, left);
    Xf r; r.m = ident(); r.it = ident();
    const V3 cols[3] = { left, new_up, dir };
    for (int c = 0; c < 3; ++c) { r.m(0, c) = cols[c].x; r.m(1, c) = cols[c].y; r.m(2, c) = cols[c].z; r.it(0, c) = cols[c].x; r.it(1, c) = cols[c].y; r.it(2, c) = cols[c].z; }
    r.m(0, 3) = origin.x; r.m(1, 3) = origin.y; r.m(2, 3) = origin.z;
    // inverse[3] = transpose(inverse) * (-origin, 1)
    M4 tt = transpose(r.it);
    float arg[4] = { -origin.x, -origin.y, -origin.z, 1.f };
    for (int i = 0; i < 4; ++i) {
        float sum = tt(i, 0) * arg[0];
        for (int j = 1; j < 4; ++j) sum = fmadd(tt(i, j), arg[j], sum);
        r.it(3, i) = sum;
    }
    return r;
}
static Xf perspective(float fov, float near_, float far_) {   // transform.h:420-437
    float recip = 1.f / (far_ - near_);
    float tan = (float) std::tan((double) (fov * .5f * (Pi / 180.f))), cot = 1.f / tan;
    Xf r; r.m = M4{}; M4 inv{};
    r.m(0, 0) = cot; r.m(1, 1) = cot; r.m(2, 2) = far_ * recip; r.m(3, 3) = 0.f;
    r.m(2, 3) = -near_ * far_ * recip; r.m(3, 2) = 1.f;
    inv(0, 0) = tan; inv(1, 1) = tan; inv(2, 2) = 0.f; inv(3, 3) = rcp(near_);
    inv(2, 3) = 1.f; inv(3, 2) = (near_ - far_) / (far_ * near_);
    r.it = transpose(inv);
    return r;
}
static inline V3 apply_point(const M4 &m, V3 p) {      // affine, transform.h:325-335
    V3 r(m(0, 3), m(1, 3), m(2, 3));
    r = V3(fmadd(m(0, 0), p.x, r.x), fmadd(m(1, 0), p.x, r.y), fmadd(m(2, 0), p.x, r.z));
    r = V3(fmadd(m(0, 1), p.y, r.x), fmadd(m(1, 1), p.y, r.y), fmadd(m(2, 1), p.y, r.z));
    r = V3(fmadd(m(0, 2), p.z, r.x), fmadd(m(1, 2), p.z, r.y), fmadd(m(2, 2), p.z, r.z));
    return r;
}
static inline V3 apply_vec(const M4 &m, V3 v) {        // transform.h:288-299 (also normals with `it`)
    V3 r(m(0, 0) * v.x, m(1, 0) * v.x, m(2, 0) * v.x);
    r = V3(fmadd(m(0, 1), v.y, r.x), fmadd(m(1, 1), v.y, r.y), fmadd(m(2, 1), v.y, r.z));
    r = V3(fmadd(m(0, 2), v.z, r.x), fmadd(m(1, 2), v.z, r.y), fmadd(m(2, 2), v.z, r.z));
    return r;
}
static float det3(const M4 &m) {
    return m(0, 0) * (m(1, 1) * m(2, 2) - m(1, 2) * m(2, 1)) - m(0, 1) * (m(1, 0) * m(2, 2) - m(1, 2) * m(2, 0)) + m(0, 2) * (m(1, 0) * m(2, 1) - m(1, 1) * m(2, 0));
}
/* Mesh::transform + flip_winding (mesh.cpp:1160-1215) */
static void bake(const Xf &t, float *V, uint32_t nv, uint32_t *F, uint32_t nf) {
    for (uint32_t i = 0; i < nv; ++i) {
        float *r = V + 8 * (size_t) i;
        V3 p = apply_point(t.m, V3(r[0], r[1], r[2]));
        V3 n = normalize(apply_vec(t.it, V3(r[3], r[4], r[5])));
        r[0] = p.x; r[1] = p.y; r[2] = p.z; r[3] = n.x; r[4] = n.y; r[5] = n.z;
    }
    if (det3(t.m) < 0.f)
        for (uint32_t f = 0; f < nf; ++f) { uint32_t a = F[4 * f]; F[4 * f] = F[4 * f + 2]; F[4 * f + 2] = a; }
}
} // namespace

extern "C" {

void orc_look_at(const float o[3], const float t[3], const float u[3], float out[32]) { store(look_at(V3(o[0], o[1], o[2]), V3(t[0], t[1], t[2]), V3(u[0], u[1], u[2])), out); }
void orc_translate(const float v[3], float out[32]) { store(translate(V3(v[0], v[1], v[2])), out); }
void orc_scale(const float v[3], float out[32]) { store(scale(V3(v[0], v[1], v[2])), out); }
void orc_rotate(const float a[3], float deg, float out[32]) { store(rotate(V3(a[0], a[1], a[2]), deg), out); }
void orc_matmul(const float a[32], const float b[32], float out[32]) { store(mul_affine(load(a), load(b)), out); }
void orc_affine_inverse(const float m[32], float out[32]) { store(inverse(load(m)), out); }

void orc_perspective_sensor(const float to_world[32], double fov, const char *fov_axis_, float near_clip, float far_clip,
                            uint32_t width, uint32_t height, uint32_t cx, uint32_t cy, uint32_t cw, uint32_t ch,
                            uint32_t rfilter, float stddev, OrcSensor *out) {
    // parse_fov, src/render/sensor.cpp:142-190 (double precision)
    double aspect = width / (double) height;
    std::string axis = fov_axis_ ? fov_axis_ : "x";
    if (axis == "smaller") axis = aspect > 1 ? "y" : "x";
    else if (axis == "larger") axis = aspect > 1 ? "x" : "y";
    const double dpi = 3.14159265358979323846;
    double result = fov;
    if (axis == "y") result = (180.0 / dpi) * (2.0 * std::atan(std::tan(0.5 * fov * dpi / 180.0) * aspect));
    else if (axis == "diagonal") {
        double diagonal = 2.0 * std::tan(0.5 * fov * dpi / 180.0);
        double w = diagonal / std::sqrt(1.0 + 1.0 / (aspect * aspect));
        result = (180.0 / dpi) * (2.0 * std::atan(w * 0.5));
    }
    float x_fov = (float) result;
    // perspective_projection, sensor.h:234-269
    float fsx = (float) (int) width, fsy = (float) (int) height;
    float rel_sx = (float) (int) cw / fsx, rel_sy = (float) (int) ch / fsy;
    float rel_ox = (float) (int) cx / fsx, rel_oy = (float) (int) cy / fsy;
    float asp = fsx / fsy;
    Xf p = mul_general(scale(V3(1.f / rel_sx, 1.f / rel_sy, 1.f)),
           mul_general(translate(V3(-rel_ox, -rel_oy, 0.f)),
           mul_general(scale(V3(-0.5f, -0.5f * asp, 1.f)),
           mul_general(translate(V3(-1.f, -1.f / asp, 0.f)), perspective(x_fov, near_clip, far_clip)))));
    Xf s2c = inverse(p);
    std::memcpy(out->sample_to_camera, s2c.m.m, 64);
    std::memcpy(out->to_world, to_world, 64);
    out->near_clip = near_clip; out->far_clip = far_clip;
    out->film_width = width; out->film_height = height;
    out->crop_offset_x = cx; out->crop_offset_y = cy; out->crop_width = cw; out->crop_height = ch;
    out->rfilter = rfilter; out->rfilter_stddev = stddev; out->rfilter_param1 = 1.f / 3.f;
    out->principal_point_offset_x = 0.f; out->principal_point_offset_y = 0.f;
}

void orc_rectangle(const float to_world[32], float *vertices, uint32_t *faces, float normal[3], float *inv_area) {
    static const uint32_t face_records[8] = { 1, 2, 0, 0, 1, 3, 2, 0 };
    static const float vertex_records[32] = {
        -1, -1, 0, 0, 0, 1, 0, 0,   1, -1, 0, 0, 0, 1, 1, 0,
        -1,  1, 0, 0, 0, 1, 0, 1,   1,  1, 0, 0, 0, 1, 1, 1 };
    std::memcpy(vertices, vertex_records, sizeof(vertex_records));
    std::memcpy(faces, face_records, sizeof(face_records));
    Xf t = load(to_world);
    V3 n = normalize(apply_vec(t.it, V3(0.f, 0.f, 1.f)));
    V3 dp_du = apply_vec(t.m, V3(2.f, 0.f, 0.f)), dp_dv = apply_vec(t.m, V3(0.f, 2.f, 0.f));
    normal[0] = n.x; normal[1] = n.y; normal[2] = n.z;
    *inv_area = rcp(norm(cross(dp_du, dp_dv)));
    bake(t, vertices, 4, faces, 2);
}

void orc_cube(const float to_world[32], float *vertices, uint32_t *faces) {
    const float side_normals[6][3] = { { 0, -1, 0 }, { 0, 1, 0 }, { 1, 0, 0 }, { 0, 0, 1 }, { -1, 0, 0 }, { 0, 0, -1 } };
    const float side_uv[4][2] = { { 0, 1 }, { 1, 1 }, { 1, 0 }, { 0, 0 } };
    static const uint32_t position_index[24] = { 1, 5, 4, 0, 3, 2, 6, 7, 1, 3, 7, 5, 5, 7, 6, 4, 4, 6, 2, 0, 3, 1, 0, 2 };
    for (uint32_t s = 0; s < 6; ++s) {
        uint32_t v = 4 * s;
        for (uint32_t k = 0; k < 4; ++k) {
            uint32_t c = position_index[v + k];
            float *r = vertices + 8 * (size_t) (v + k);
            r[0] = c & 1 ? 1.f : -1.f; r[1] = c & 2 ? 1.f : -1.f; r[2] = c & 4 ? 1.f : -1.f;
            r[3] = side_normals[s][0]; r[4] = side_normals[s][1]; r[5] = side_normals[s][2];
            r[6] = side_uv[k][0]; r[7] = side_uv[k][1];
        }
        uint32_t *f0 = faces + 4 * (size_t) (2 * s), *f1 = f0 + 4;
        f0[0] = v; f0[1] = v + 1; f0[2] = v + 2; f0[3] = 0;
        f1[0] = v + 3; f1[1] = v; f1[2] = v + 2; f1[3] = 0;
    }
    bake(load(to_world), vertices, 24, faces, 12);
}

/* a triangle mesh given in object space with a `to_world`: the records are transformed in place (PackedMesh::set_transform + transform_records,
 * src/render/mesh_utils.cpp:33-88: positions by the matrix, normals by the inverse transpose and renormalised, winding reversed for a
 * mirroring transform) */
void orc_bake_mesh(const float to_world[32], float *vertices, uint32_t vertex_count, uint32_t *faces, uint32_t face_count) {
    bake(load(to_world), vertices, vertex_count, faces, face_count);
}

} // extern "C"
