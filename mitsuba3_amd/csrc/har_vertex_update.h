/*
 * har_vertex_update.h -- what Mesh::parameters_changed does to a mesh whose vertex positions were written (src/render/mesh.cpp:848-899), as per-element
 * HAR_HD functions: the HIP kernels of har_refit.hip run them on the device arrays, the host harness runs the same code.
 *
 * In the reference's JIT variants the positions never leave the device: parameters_changed regenerates the vertex normals of a smooth mesh
 * (pack(regenerate_normals) -> compute_normals, mesh.cpp:876-878, 1216-1267), recomputes the bounding box, and Scene::parameters_changed hands the accel its
 * new vertices (scene.cpp:517-540).  Here (round 6, har_scene_update_vertices_device):
 *   1. set_position           the 3 floats of every packed 32-byte vertex record from the caller's DEVICE array;
 *   2. vertex_normal          Mesh::compute_normals as a GATHER: one thread per vertex walks the vertex's corners in (face, corner) order -- the order in which
 *                             the serial host loop (har_mesh_compute_normals) reaches them -- and adds  n_face * angle  in float32, so the sum has the host
 *                             loop's rounding sequence and does not depend on thread scheduling (the reference's scatter_reduce has no defined order at all);
 *                             the corner list (4 B offset per vertex + 4 B per corner) is built once per mesh on the host;
 *   3. shading_triangle       the face's three vertex records copied into its 96-byte shading triangle (DScene::shade_tris, what compute_si reads);
 *   4. the BLAS refit of har_refit.h;
 *   5. for a mesh inside a shape group: the world-space boxes of the group's instances (instance_box_*) and a refit of the instance level (the TLAS keeps its topology,
 *      like the BLAS), so that an instanced mesh is updated without the host as well.
 */
#pragma once
#include "har_refit.h"

namespace har {

/* dr::unit_angle (Dr.Jit, NOT IN TREE -- restated from its published definition): numerically robust angle between two unit vectors */
HAR_HD float mesh_unit_angle(Vec3 a, Vec3 b) {
    float dot_uv = dot3(a, b);
    Vec3 am(mulsign_(a.x, dot_uv), mulsign_(a.y, dot_uv), mulsign_(a.z, dot_uv));
    float temp = 2.f * asinf(.5f * norm3(b - am));
    return dot_uv >= 0.f ? temp : HAR_PI - temp;
}

/* what face (p0, p1, p2) adds to the normal sum of its corner k (mesh.cpp:1237-1251); false: a face without area adds nothing */
HAR_HD bool mesh_corner_term(const Vec3 p[3], int k, Vec3 &term) {
    Vec3 n = cross3(p[1] - p[0], p[2] - p[0]);
    const float length_sqr = dot3(n, n);
    if (!(length_sqr > 0.f)) return false;
    n = n * rsqrt_(length_sqr);
    const float angle = mesh_unit_angle(normalize3(p[(k + 1) % 3] - p[k]), normalize3(p[(k + 2) % 3] - p[k]));
    term = Vec3(n.x * angle, n.y * angle, n.z * angle);
    return true;
}

/* corner list entry: face index | corner << 30 (a mesh has < 2^30 faces) */
HAR_HD uint32_t corner_face(uint32_t e) { return e & 0x3fffffffu; }
HAR_HD uint32_t corner_slot(uint32_t e) { return e >> 30; }

/* vertex `v` of a mesh (verts / faces = the mesh's own records): normal = normalised sum over its corners, (1, 0, 0) without a valid contribution (mesh.cpp:1256-1264) */
HAR_HD void vertex_normal(float *verts, const uint32_t *faces, const uint32_t *corner_begin, const uint32_t *corners, uint32_t v) {
    float ax = 0.f, ay = 0.f, az = 0.f;
    for (uint32_t e = corner_begin[v]; e < corner_begin[v + 1]; ++e) {
        const uint32_t *fi = faces + 4 * (size_t) corner_face(corners[e]);
        Vec3 p[3];
        for (int k = 0; k < 3; ++k) { const float *q = verts + 8 * (size_t) fi[k]; p[k] = Vec3(q[0], q[1], q[2]); }
        Vec3 t;
        if (!mesh_corner_term(p, (int) corner_slot(corners[e]), t)) continue;
        ax += t.x; ay += t.y; az += t.z;
    }
    Vec3 n(ax, ay, az);
    const float length_sqr = dot3(n, n);
    n = length_sqr > 0.f ? n * rsqrt_(length_sqr) : Vec3(1.f, 0.f, 0.f);
    float *o = verts + 8 * (size_t) v; o[3] = n.x; o[4] = n.y; o[5] = n.z;
}

/* the 96-byte shading triangle of face `f` (three 32-byte vertex records in corner order) */
HAR_HD void shading_triangle(const float *verts, const uint32_t *faces, float *shade_tris, uint32_t f) {
    for (int k = 0; k < 3; ++k) {
        const float *src = verts + 8 * (size_t) faces[4 * (size_t) f + k]; float *dst = shade_tris + 24 * (size_t) f + 8 * k;
        for (int c = 0; c < 8; ++c) dst[c] = src[c];
    }
}

/* world-space box of an instance (a TLAS leaf) from the object-space vertices of its group -- the exact bound of the transformed vertices, as build_tlas takes it
 * (har_scene_host.cpp; Instance::bbox of the reference transforms the eight corners of the group's box, src/shapes/instance.cpp:93-103, which a TLAS leaf need not) */
HAR_HD RefitBox instance_box_empty() { RefitBox b; for (int a = 0; a < 3; ++a) { b.lo[a] = HAR_INF; b.hi[a] = -HAR_INF; } return b; }
HAR_HD void instance_box_grow(RefitBox &b, const float *to_world, const float *vertex_record) {
    const Vec3 q = xf_point(to_world, Vec3(vertex_record[0], vertex_record[1], vertex_record[2]));
    b.lo[0] = fminf(b.lo[0], q.x); b.lo[1] = fminf(b.lo[1], q.y); b.lo[2] = fminf(b.lo[2], q.z);
    b.hi[0] = fmaxf(b.hi[0], q.x); b.hi[1] = fmaxf(b.hi[1], q.y); b.hi[2] = fmaxf(b.hi[2], q.z);
}
HAR_HD void instance_box_finish(RefitBox &b) { if (b.lo[0] <= b.hi[0]) pad_box(b.lo, b.hi); }

/* Instance::parameters_changed (src/shapes/instance.cpp:79-91): a new to_world and its inverse.  m, inv: column-major 3 x 4 ({column 0, column 1, column 2, translation}).
 * The inverse is formed in double (cofactors of the 3 x 3 part) and rounded once; false: the matrix is singular or not finite (nothing is written) */
HAR_HD bool affine_inverse(const float *m, float *inv) {
    const double a = m[0], b = m[3], c = m[6], d = m[1], e = m[4], f = m[7], g = m[2], h = m[5], i = m[8];      /* rows of the 3 x 3 part: (a b c), (d e f), (g h i) */
    const double A = e * i - f * h, B = -(d * i - f * g), Cc = d * h - e * g;
    const double det = a * A + b * B + c * Cc;
    bool ok = det != 0.0 && det == det && fabs(det) < 1e300;
    for (int k = 0; k < 12; ++k) ok = ok && (m[k] - m[k] == 0.f);                        /* finite */
    if (!ok) return false;
    const double r = 1.0 / det;
    const double n00 = A * r, n01 = -(b * i - c * h) * r, n02 = (b * f - c * e) * r;
    const double n10 = B * r, n11 = (a * i - c * g) * r, n12 = -(a * f - c * d) * r;
    const double n20 = Cc * r, n21 = -(a * h - b * g) * r, n22 = (a * e - b * d) * r;
    const double tx = m[9], ty = m[10], tz = m[11];
    inv[0] = (float) n00; inv[1] = (float) n10; inv[2] = (float) n20;
    inv[3] = (float) n01; inv[4] = (float) n11; inv[5] = (float) n21;
    inv[6] = (float) n02; inv[7] = (float) n12; inv[8] = (float) n22;
    inv[9] = (float) -(n00 * tx + n01 * ty + n02 * tz); inv[10] = (float) -(n10 * tx + n11 * ty + n12 * tz); inv[11] = (float) -(n20 * tx + n21 * ty + n22 * tz);
    return true;
}

} // namespace har
