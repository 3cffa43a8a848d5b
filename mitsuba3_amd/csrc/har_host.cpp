/*
 * har_host.cpp -- host C++ side of the hip_ad_rgb plugins: what the reference's
 * plugin constructors compute before any sample is drawn (Transform4f algebra,
 * PerspectiveCamera projection set-up, Rectangle / Cube / Mesh record baking).
 * Scalar fp32 with the reference's operation order, exposed through the C ABI
 * (include/hip_ad_rgb.h, "Host-side plugin lowering").
 */
#include "../../include/hip_ad_rgb.h"
#include "har_math.h"
#include <cmath>
#include <cstring>
#include <string>

namespace har { namespace host {

struct Matrix4f {
    float v[4][4];
    static Matrix4f zero() { Matrix4f m; std::memset(m.v, 0, sizeof(m.v)); return m; }
    static Matrix4f identity() { Matrix4f m = zero(); for (int i = 0; i < 4; ++i) m.v[i][i] = 1.f; return m; }
    Matrix4f transposed() const { Matrix4f r; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.v[i][j] = v[j][i]; return r; }
    /* dr::Matrix product: row i = a(i,0) * b.row(0), then fma-accumulate the other rows */
    Matrix4f operator*(const Matrix4f &b) const {
        Matrix4f r;
        for (int i = 0; i < 4; ++i)
            for (int c = 0; c < 4; ++c) {
                float acc = v[i][0] * b.v[0][c];
                for (int j = 1; j < 4; ++j) acc = fma_(v[i][j], b.v[j][c], acc);
                r.v[i][c] = acc;
            }
        return r;
    }
};

/* Transform<Point4f>: matrix + inverse transpose (include/mitsuba/core/transform.h:40-58) */
struct Transform4f {
    Matrix4f matrix = Matrix4f::identity(), inverse_transpose = Matrix4f::identity();

    static Transform4f from32(const float *p) { Transform4f t; std::memcpy(t.matrix.v, p, 64); std::memcpy(t.inverse_transpose.v, p + 16, 64); return t; }
    void to32(float *p) const { std::memcpy(p, matrix.v, 64); std::memcpy(p + 16, inverse_transpose.v, 64); }

    Transform4f inverse() const { Transform4f r; r.matrix = inverse_transpose.transposed(); r.inverse_transpose = matrix.transposed(); return r; }

    static Transform4f translate(const float t[3]) {
        Transform4f r;
        for (int i = 0; i < 3; ++i) { r.matrix.v[i][3] = t[i]; r.inverse_transpose.v[3][i] = -t[i]; }
        return r;
    }
    static Transform4f scale(const float s[3]) {
        Transform4f r;
        for (int i = 0; i < 3; ++i) { r.matrix.v[i][i] = s[i]; r.inverse_transpose.v[i][i] = rcp_(s[i]); }
        return r;
    }
    /* dr::rotate<Matrix4f>(axis, deg_to_rad(angle)): Rodrigues with the sincos of har_math.h */
    static Transform4f rotate(const float a[3], float angle_deg) {
        float sn, cs; sincos_(angle_deg * (HAR_PI / 180.f), sn, cs);
        const float cm = 1.f - cs;
        const float s1[3] = { a[1], a[2], a[0] }, s2[3] = { a[2], a[0], a[1] };
        float t0[3], t1[3], t2[3];
        for (int i = 0; i < 3; ++i) {
            t0[i] = fma_(a[i] * a[i], cm, cs);
            t1[i] = fma_(a[i] * s1[i], cm, s2[i] * sn);
            t2[i] = fms_(a[i] * s2[i], cm, s1[i] * sn);
        }
        Transform4f r;
        float (*m)[4] = r.matrix.v;
        m[0][0] = t0[0]; m[0][1] = t2[1]; m[0][2] = t1[2];
        m[1][0] = t1[0]; m[1][1] = t0[1]; m[1][2] = t2[2];
        m[2][0] = t2[0]; m[2][1] = t1[1]; m[2][2] = t0[2];
        r.inverse_transpose = r.matrix;
        return r;
    }
    static Transform4f look_at(const float o[3], const float t[3], const float u[3]) {
        Vec3 origin(o[0], o[1], o[2]);
        Vec3 dir = normalize3(Vec3(t[0], t[1], t[2]) - origin);
        Vec3 left = normalize3(cross3(Vec3(u[0], u[1], u[2]), dir));
        Vec3 new_up = cross3(dir, left);
        Transform4f r;
        const Vec3 cols[3] = { left, new_up, dir };
        for (int c = 0; c < 3; ++c) {
            const float col[3] = { cols[c].x, cols[c].y, cols[c].z };
            for (int i = 0; i < 3; ++i) { r.matrix.v[i][c] = col[i]; r.inverse_transpose.v[i][c] = col[i]; }
        }
        for (int i = 0; i < 3; ++i) r.matrix.v[i][3] = o[i];
        Matrix4f tt = r.inverse_transpose.transposed();
        const float arg[4] = { -o[0], -o[1], -o[2], 1.f };
        for (int i = 0; i < 4; ++i) {
            float acc = tt.v[i][0] * arg[0];
            for (int j = 1; j < 4; ++j) acc = fma_(tt.v[i][j], arg[j], acc);
            r.inverse_transpose.v[3][i] = acc;
        }
        return r;
    }
    /* Transform::perspective (transform.h:420-437) */
    static Transform4f perspective(float fov, float near_, float far_) {
        float recip = 1.f / (far_ - near_);
        float tan_ = (float) std::tan((double) (fov * .5f * (HAR_PI / 180.f))), cot = 1.f / tan_;
        Transform4f r; r.matrix = Matrix4f::zero();
        r.matrix.v[0][0] = cot; r.matrix.v[1][1] = cot; r.matrix.v[2][2] = far_ * recip;
        r.matrix.v[2][3] = -near_ * far_ * recip; r.matrix.v[3][2] = 1.f;
        Matrix4f inv = Matrix4f::zero();
        inv.v[0][0] = tan_; inv.v[1][1] = tan_; inv.v[3][3] = rcp_(near_);
        inv.v[2][3] = 1.f; inv.v[3][2] = (near_ - far_) / (far_ * near_);
        r.inverse_transpose = inv.transposed();
        return r;
    }
    /* affine concatenation (transform.h:364-400) */
    Transform4f operator*(const Transform4f &o) const {
        Transform4f r;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                float sum = 0.f, sum_it = 0.f;
                for (int k = 0; k < 3; ++k) sum = fma_(matrix.v[i][k], o.matrix.v[k][j], sum);
                for (int k = 0; k < 3; ++k) sum_it = fma_(inverse_transpose.v[i][k], o.inverse_transpose.v[k][j], sum_it);
                r.matrix.v[i][j] = sum; r.inverse_transpose.v[i][j] = sum_it;
            }
        for (int l = 0; l < 3; ++l) {
            float sum = matrix.v[l][3], sum_it = o.inverse_transpose.v[3][l];
            for (int k = 0; k < 3; ++k) sum = fma_(matrix.v[l][k], o.matrix.v[k][3], sum);
            for (int k = 0; k < 3; ++k) sum_it = fma_(inverse_transpose.v[3][k], o.inverse_transpose.v[k][l], sum_it);
            r.matrix.v[l][3] = sum; r.inverse_transpose.v[3][l] = sum_it;
        }
        return r;
    }
    /* projective concatenation (transform.h:401-404) */
    Transform4f mul_projective(const Transform4f &o) const { Transform4f r; r.matrix = matrix * o.matrix; r.inverse_transpose = inverse_transpose * o.inverse_transpose; return r; }

    Vec3 point(Vec3 p) const {
        const float (*m)[4] = matrix.v;
        Vec3 r(m[0][3], m[1][3], m[2][3]);
        r = Vec3(fma_(m[0][0], p.x, r.x), fma_(m[1][0], p.x, r.y), fma_(m[2][0], p.x, r.z));
        r = Vec3(fma_(m[0][1], p.y, r.x), fma_(m[1][1], p.y, r.y), fma_(m[2][1], p.y, r.z));
        r = Vec3(fma_(m[0][2], p.z, r.x), fma_(m[1][2], p.z, r.y), fma_(m[2][2], p.z, r.z));
        return r;
    }
    static Vec3 lin(const float (*m)[4], Vec3 v) {
        Vec3 r(m[0][0] * v.x, m[1][0] * v.x, m[2][0] * v.x);
        r = Vec3(fma_(m[0][1], v.y, r.x), fma_(m[1][1], v.y, r.y), fma_(m[2][1], v.y, r.z));
        r = Vec3(fma_(m[0][2], v.z, r.x), fma_(m[1][2], v.z, r.y), fma_(m[2][2], v.z, r.z));
        return r;
    }
    Vec3 vector(Vec3 v) const { return lin(matrix.v, v); }
    Vec3 normal(Vec3 n) const { return lin(inverse_transpose.v, n); }
    float det3() const {
        const float (*m)[4] = matrix.v;
        return m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
               m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
    }
};

/* Mesh::transform (mesh.cpp:1160-1200) + flip_winding (:1202-1215) */
static void bake_records(const Transform4f &t, uint32_t nv, float *V, uint32_t nf, uint32_t *F, bool normals) {
    for (uint32_t i = 0; i < nv; ++i) {
        float *r = V + 8 * (size_t) i;
        Vec3 p = t.point(Vec3(r[0], r[1], r[2]));
        r[0] = p.x; r[1] = p.y; r[2] = p.z;
        if (normals) { Vec3 n = normalize3(t.normal(Vec3(r[3], r[4], r[5]))); r[3] = n.x; r[4] = n.y; r[5] = n.z; }
    }
    if (t.det3() < 0.f)
        for (uint32_t f = 0; f < nf; ++f) { uint32_t a = F[4 * (size_t) f]; F[4 * (size_t) f] = F[4 * (size_t) f + 2]; F[4 * (size_t) f + 2] = a; }
}

}} // namespace har::host

using namespace har;
using namespace har::host;

extern "C" {

int har_transform_translate(const float v[3], float out[32]) { Transform4f::translate(v).to32(out); return 0; }
int har_transform_scale(const float v[3], float out[32]) { Transform4f::scale(v).to32(out); return 0; }
int har_transform_rotate(const float axis[3], float deg, float out[32]) { Transform4f::rotate(axis, deg).to32(out); return 0; }
int har_transform_look_at(const float o[3], const float t[3], const float u[3], float out[32]) { Transform4f::look_at(o, t, u).to32(out); return 0; }
int har_transform_mul(const float a[32], const float b[32], float out[32]) { (Transform4f::from32(a) * Transform4f::from32(b)).to32(out); return 0; }
int har_transform_inverse(const float a[32], float out[32]) { Transform4f::from32(a).inverse().to32(out); return 0; }

int har_perspective_sensor(const float to_world[32], double fov, const char *fov_axis_, float near_clip, float far_clip, uint32_t width,
                           uint32_t height, uint32_t cx, uint32_t cy, uint32_t cw, uint32_t ch, uint32_t rfilter, float stddev, HarSensor *out) {
    if (!out || !to_world || width == 0 || height == 0) return 1;
    const double aspect = width / (double) height, pi = 3.14159265358979323846;
    std::string axis = fov_axis_ ? fov_axis_ : "x";
    for (auto &c : axis) c = (char) std::tolower(c);
    if (axis == "smaller") axis = aspect > 1 ? "y" : "x";
    else if (axis == "larger") axis = aspect > 1 ? "x" : "y";
    double x_fov_d;
    if (axis == "x") x_fov_d = fov;
    else if (axis == "y") x_fov_d = (180.0 / pi) * (2.0 * std::atan(std::tan(0.5 * fov * pi / 180.0) * aspect));
    else if (axis == "diagonal") {
        double diagonal = 2.0 * std::tan(0.5 * fov * pi / 180.0);
        double w = diagonal / std::sqrt(1.0 + 1.0 / (aspect * aspect));
        x_fov_d = (180.0 / pi) * (2.0 * std::atan(w * 0.5));
    } else return 2;
    if (x_fov_d <= 0.0 || x_fov_d >= 180.0) return 3;
    const float x_fov = (float) x_fov_d;
    /* perspective_projection (sensor.h:234-269) */
    const float fsx = (float) (int) width, fsy = (float) (int) height;
    const float rel_size[2] = { (float) (int) cw / fsx, (float) (int) ch / fsy }, rel_off[2] = { (float) (int) cx / fsx, (float) (int) cy / fsy };
    const float asp = fsx / fsy;
    const float s1[3] = { 1.f / rel_size[0], 1.f / rel_size[1], 1.f }, t1[3] = { -rel_off[0], -rel_off[1], 0.f };
    const float s2[3] = { -0.5f, -0.5f * asp, 1.f }, t2[3] = { -1.f, -1.f / asp, 0.f };
    Transform4f proj = Transform4f::scale(s1).mul_projective(Transform4f::translate(t1).mul_projective(
                       Transform4f::scale(s2).mul_projective(Transform4f::translate(t2).mul_projective(Transform4f::perspective(x_fov, near_clip, far_clip)))));
    Transform4f s2c = proj.inverse();
    std::memcpy(out->sample_to_camera, s2c.matrix.v, 64);
    std::memcpy(out->to_world, to_world, 64);
    out->near_clip = near_clip; out->far_clip = far_clip; out->film_width = width; out->film_height = height;
    out->crop_offset_x = cx; out->crop_offset_y = cy; out->crop_width = cw; out->crop_height = ch;
    out->rfilter = rfilter; out->rfilter_stddev = stddev; out->rfilter_param1 = 1.f / 3.f;
    out->principal_point_offset_x = 0.f; out->principal_point_offset_y = 0.f; out->projection = 0u;
    return 0;
}

int har_orthographic_sensor(const float to_world[32], float near_clip, float far_clip, uint32_t width, uint32_t height, uint32_t cx, uint32_t cy, uint32_t cw, uint32_t ch,
                            uint32_t rfilter, float stddev, HarSensor *out) {
    if (!out || !to_world || width == 0 || height == 0) return 1;
    /* orthographic_projection (sensor.h:272-307): scale(1 / rel_size) * translate(-rel_offset) * scale(-1/2, -aspect/2, 1) * translate(-1, -1/aspect, 0) *
     * orthographic(near, far), the last being scale(1, 1, 1 / (far - near)) * translate(0, 0, -near) (transform.h:162-166) */
    const float fsx = (float) (int) width, fsy = (float) (int) height;
    const float rel_size[2] = { (float) (int) cw / fsx, (float) (int) ch / fsy }, rel_off[2] = { (float) (int) cx / fsx, (float) (int) cy / fsy };
    const float asp = fsx / fsy;
    const float s1[3] = { 1.f / rel_size[0], 1.f / rel_size[1], 1.f }, t1[3] = { -rel_off[0], -rel_off[1], 0.f };
    const float s2[3] = { -0.5f, -0.5f * asp, 1.f }, t2[3] = { -1.f, -1.f / asp, 0.f };
    const float s3[3] = { 1.f, 1.f, 1.f / (far_clip - near_clip) }, t3[3] = { 0.f, 0.f, -near_clip };
    Transform4f proj = Transform4f::scale(s1) * (Transform4f::translate(t1) * (Transform4f::scale(s2) * (Transform4f::translate(t2) * (Transform4f::scale(s3) * Transform4f::translate(t3)))));
    Transform4f s2c = proj.inverse();
    std::memset(out, 0, sizeof(*out));
    std::memcpy(out->sample_to_camera, s2c.matrix.v, 64);
    std::memcpy(out->to_world, to_world, 64);
    out->near_clip = near_clip; out->far_clip = far_clip; out->film_width = width; out->film_height = height;
    out->crop_offset_x = cx; out->crop_offset_y = cy; out->crop_width = cw; out->crop_height = ch;
    out->rfilter = rfilter; out->rfilter_stddev = stddev; out->rfilter_param1 = 1.f / 3.f;
    out->projection = 1u;
    return 0;
}

int har_shape_rectangle(const float to_world[32], int flip_normals, float vertices[32], uint32_t faces[8], float normal[3], float *inv_area) {
    static const uint32_t face_records[8] = { 1, 2, 0, 0, 1, 3, 2, 0 };
    static const float vertex_records[32] = { -1, -1, 0, 0, 0, 1, 0, 0,  1, -1, 0, 0, 0, 1, 1, 0,  -1, 1, 0, 0, 0, 1, 0, 1,  1, 1, 0, 0, 0, 1, 1, 1 };
    Transform4f t = Transform4f::from32(to_world);
    if (flip_normals) { const float s[3] = { 1.f, 1.f, -1.f }; t = t * Transform4f::scale(s); }
    Vec3 n = normalize3(t.normal(Vec3(0.f, 0.f, 1.f)));
    Vec3 dp_du = t.vector(Vec3(2.f, 0.f, 0.f)), dp_dv = t.vector(Vec3(0.f, 2.f, 0.f));
    normal[0] = n.x; normal[1] = n.y; normal[2] = n.z;
    *inv_area = rcp_(norm3(cross3(dp_du, dp_dv)));
    std::memcpy(vertices, vertex_records, sizeof(vertex_records));
    std::memcpy(faces, face_records, sizeof(face_records));
    bake_records(t, 4, vertices, 2, faces, true);
    return 0;
}

int har_shape_cube(const float to_world[32], float vertices[192], uint32_t faces[48]) {
    static const float side_normals[6][3] = { { 0, -1, 0 }, { 0, 1, 0 }, { 1, 0, 0 }, { 0, 0, 1 }, { -1, 0, 0 }, { 0, 0, -1 } };
    static const float side_uv[4][2] = { { 0, 1 }, { 1, 1 }, { 1, 0 }, { 0, 0 } };
    static const uint32_t corner_of[24] = { 1, 5, 4, 0, 3, 2, 6, 7, 1, 3, 7, 5, 5, 7, 6, 4, 4, 6, 2, 0, 3, 1, 0, 2 };
    for (uint32_t side = 0; side < 6; ++side) {
        const uint32_t v = 4 * side;
        for (uint32_t k = 0; k < 4; ++k) {
            const uint32_t c = corner_of[v + k];
            float *r = vertices + 8 * (size_t) (v + k);
            r[0] = (c & 1) ? 1.f : -1.f; r[1] = (c & 2) ? 1.f : -1.f; r[2] = (c & 4) ? 1.f : -1.f;
            std::memcpy(r + 3, side_normals[side], 12); std::memcpy(r + 6, side_uv[k], 8);
        }
        uint32_t *f = faces + 8 * (size_t) side;
        f[0] = v; f[1] = v + 1; f[2] = v + 2; f[3] = 0;
        f[4] = v + 3; f[5] = v; f[6] = v + 2; f[7] = 0;
    }
    bake_records(Transform4f::from32(to_world), 24, vertices, 12, faces, true);
    return 0;
}

int har_mesh_transform(const float to_world[32], uint32_t nv, float *vertices, uint32_t nf, uint32_t *faces, int has_normals) {
    if (!to_world || (nv && !vertices) || (nf && !faces)) return 1;
    bake_records(Transform4f::from32(to_world), nv, vertices, nf, faces, has_normals != 0);
    return 0;
}

} // extern "C"
