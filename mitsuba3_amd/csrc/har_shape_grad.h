/*
 * har_shape_grad.h -- vertex-position gradients of the PRB adjoint (HAR_HD: device kernel + host test harness).
 *
 * What the reference obtains by reverse-mode AD through the geometry-attached part of PRBIntegrator.sample
 * (src/python/python/ad/integrators/prb.py:124-141 attached surface interaction, :176-216 emitter sampling from the attached
 * point, :261-297 attached outgoing direction and solid-angle-to-area Jacobian) is written out here by hand for one path vertex
 * on a flat-shaded triangle with a `diffuse` BSDF (plain or inside `twosided`).  Per vertex the differentiable quantities are
 *
 *   p_att = b0 P0 + b1 P1 + b2 P2                     (barycentrics detached, mesh.cpp:2296)
 *   n     = normalize((P1 - P0) x (P2 - P0))          (= shading normal of a mesh without vertex normals)
 *   p     = o + d * <p_att - o, n_det> / <n_det, d>   (attach_motion without FollowShape, interaction.h:536-544: the point stays on the ray)
 *   b_i  += barycentric coordinates of (p - p_att)    (mesh.cpp:2308-2321) -> uv -> rho(uv)
 *
 * and two "direction blocks" use them, one for emitter sampling (target = the detached emitter sample) and one for the sampled
 * continuation (target = the detached next interaction): w = normalize(y - p), cos = <w, n>, J = |<m, w>| / |y - p|^2 with
 *
 *   Lr_dir = beta mis em_weight * rho/pi cos * relative_grad(J),   Lr_ind = L * relative_grad(rho/pi cos) * relative_grad(J).
 *
 * The caller reduces the colour channels to scalar adjoints (cos_bar = d objective / d cos, a = d objective / d log J, uv_bar); this
 * file turns them into the gradient w.r.t. P0, P1, P2.  The oracle (oracle/mi_oracle.cpp attach_si, over dual numbers) computes the
 * same thing without any of the formulas below.
 */
#pragma once
#include "har_scene.h"

namespace har {

struct ShapeDirTerm {
    bool on;            /* the term exists (visible emitter sample / the path continued) */
    bool attached;      /* w = normalize(target - p) and J exist (surface emitter, valid next interaction); otherwise w is a detached direction and J = 1 */
    Vec3 target, normal;/* y, m: detached */
    Vec3 w;             /* value of the direction (ds.d / ray_next.d) */
    float cos_bar, a;   /* adjoints of <w, n> and of log J */
};

struct ShapeVertex {
    Vec3 p0, p1, p2;
    float b1, b2;
    Vec3 d_in;                      /* direction of the (detached) ray the vertex lies on */
    bool has_uv; float duv0[2], duv1[2];   /* texcoord differences (P1 - P0, P2 - P0) of the triangle; without texcoords uv = (b1, b2) */
    float uv_bar[2];                /* adjoint of the texture coordinates (both terms) */
    ShapeDirTerm nee, ind;
};

/* adjoint of the point p a direction block starts from: w = normalize(y - p) enters cos = <w, n> (n = the normal the BSDF's cosine uses) and
 * log J = log |<m, w>| - 2 log |y - p|.  Zero for a detached direction (w given, J = 1). */
HAR_HD Vec3 dir_term_point_adjoint(const ShapeDirTerm &T, Vec3 p, Vec3 n) {
    if (!T.on || !T.attached) return Vec3(0.f);
    const Vec3 D = T.target - p;
    const float r2 = dot3(D, D), r = sqrtf(r2);
    const Vec3 u = D * rcp_(r);
    const float c = dot3(T.normal, u);
    Vec3 u_bar = n * T.cos_bar;                            /* d cos / d w */
    if (c != 0.f) u_bar = u_bar + T.normal * (T.a / c);    /* d log |<m, u>| / d u */
    /* u = D / |D|, D = y - p:  du = -(dp - u <u, dp>) / r;  log J also holds -2 log r, dr = -<u, dp> */
    const Vec3 proj = u_bar - u * dot3(u, u_bar);
    return u * (2.f * T.a / r) - proj * rcp_(r);
}

/* adds d objective / d P_k to g[k] */
HAR_HD void shape_vertex_adjoint(const ShapeVertex &v, Vec3 g[3]) {
    const float b0 = 1.f - v.b1 - v.b2;
    const Vec3 e1 = v.p1 - v.p0, e2 = v.p2 - v.p0;
    const Vec3 p = fma3(v.p0, b0, fma3(v.p1, v.b1, v.p2 * v.b2));
    const Vec3 N = cross3(e1, e2);
    const float len = norm3(N);
    const Vec3 n = N * rcp_(len);
    Vec3 p_bar(0.f), n_bar(0.f);
    const ShapeDirTerm *terms[2] = { &v.nee, &v.ind };
    for (int k = 0; k < 2; ++k) {
        const ShapeDirTerm &T = *terms[k];
        if (!T.on) continue;
        n_bar = n_bar + T.w * T.cos_bar;                       /* cos = <w, n> */
        p_bar = p_bar + dir_term_point_adjoint(T, p, n);
    }
    /* texture coordinates -> barycentric coordinates -> (p - p_att) */
    float b1_bar, b2_bar;
    if (v.has_uv) { b1_bar = v.uv_bar[0] * v.duv0[0] + v.uv_bar[1] * v.duv0[1]; b2_bar = v.uv_bar[0] * v.duv1[0] + v.uv_bar[1] * v.duv1[1]; }
    else { b1_bar = v.uv_bar[0]; b2_bar = v.uv_bar[1]; }
    Vec3 patt_bar(0.f);
    if (b1_bar != 0.f || b2_bar != 0.f) {
        const float a11 = dot3(e1, e1), a12 = dot3(e1, e2), a22 = dot3(e2, e2), inv_det = rcp_(a11 * a22 - a12 * a12);
        const float r1_bar = (a22 * b1_bar - a12 * b2_bar) * inv_det, r2_bar = (a11 * b2_bar - a12 * b1_bar) * inv_det;
        const Vec3 rel_bar = e1 * r1_bar + e2 * r2_bar;
        p_bar = p_bar + rel_bar; patt_bar = patt_bar - rel_bar;
    }
    /* p = o + d t,  t = <p_att - o, n_det> / <n_det, d> */
    patt_bar = patt_bar + n * (dot3(p_bar, v.d_in) / dot3(n, v.d_in));
    g[0] = g[0] + patt_bar * b0; g[1] = g[1] + patt_bar * v.b1; g[2] = g[2] + patt_bar * v.b2;
    /* n = N / |N|, N = e1 x e2 */
    const Vec3 N_bar = (n_bar - n * dot3(n, n_bar)) * rcp_(len);
    const Vec3 e1_bar = cross3(e2, N_bar), e2_bar = cross3(N_bar, e1);
    g[1] = g[1] + e1_bar; g[2] = g[2] + e2_bar; g[0] = g[0] - (e1_bar + e2_bar);
}

/* d rho_c / d (u, v) of the bilinear lookup whose taps are `l` */
HAR_HD void tex_fetch_grad(const DTexture &T, const TexTaps &l, Vec3 &d_du, Vec3 &d_dv) {
    float du[3], dv[3];
    for (int c = 0; c < 3; ++c) {
        const float v00 = T.data[3 * (size_t) l.idx[0] + c], v10 = T.data[3 * (size_t) l.idx[1] + c];
        const float v01 = T.data[3 * (size_t) l.idx[2] + c], v11 = T.data[3 * (size_t) l.idx[3] + c];
        du[c] = (float) T.w * (l.w0y * (v10 - v00) + l.w1y * (v11 - v01));
        dv[c] = (float) T.h * (l.w0x * (v01 - v00) + l.w1x * (v11 - v10));
    }
    d_du = Vec3(du[0], du[1], du[2]); d_dv = Vec3(dv[0], dv[1], dv[2]);
}

/* geometry record of one adjoint item, written by the shading stage when vertex-position gradients are requested */
struct ShapeItem {
    uint32_t shape, prim; float b1, b2;
    Vec3 d_in; uint32_t next_slot;             /* 0xffffffff: the path ended at this vertex */
    Vec3 q; uint32_t nee_flags;                /* bit 0: an emitter sample exists, bit 1: it lies on a surface, bit 2: the vertex is lit (cos_i > 0), bit 3: flipped */
    Vec3 n_e; float cos_em;
    Vec3 w_em;
};
#define HAR_SHAPE_NEE         1u
#define HAR_SHAPE_NEE_SURFACE 2u
#define HAR_SHAPE_LIT         4u
#define HAR_SHAPE_FLIPPED     8u          /* twosided BSDF seen from behind: wo is mirrored, cos = -<w, n> (twosided.cpp:124-127) */
#define HAR_SHAPE_NO_NEXT     0xffffffffu

/* One path vertex: the item's geometry record, the visibility of its emitter sample, the radiance accumulator L after this vertex's
 * subtraction (prb.py:227), the film adjoint dL, NEE's d Lr_dir / d rho (= beta mis em_weight cos / pi), and the next interaction
 * (position / geometric normal, `next_valid` = false for an escaped ray).  Adds to g[0..2]; returns false when the vertex's mesh is
 * not differentiated or nothing contributes. */
HAR_HD bool shape_item_adjoint(const DScene &S, const ShapeItem &it, uint32_t bsdf, bool visible, Vec3 L, Vec3 dl, Vec3 dLr_drho,
                               bool has_next, bool next_valid, Vec3 next_p, Vec3 next_n, Vec3 next_d, Vec3 g[3], uint32_t vid[3]) {
    const DMesh M = S.meshes[it.shape];
    const uint32_t *f = S.faces + 4 * (size_t) (M.foff + it.prim);
    vid[0] = f[0]; vid[1] = f[1]; vid[2] = f[2];
    const float *r0 = S.verts + 8 * (size_t) (M.voff + f[0]), *r1 = S.verts + 8 * (size_t) (M.voff + f[1]), *r2 = S.verts + 8 * (size_t) (M.voff + f[2]);
    ShapeVertex v;
    v.p0 = Vec3(r0[0], r0[1], r0[2]); v.p1 = Vec3(r1[0], r1[1], r1[2]); v.p2 = Vec3(r2[0], r2[1], r2[2]);
    v.b1 = it.b1; v.b2 = it.b2; v.d_in = it.d_in;
    v.has_uv = (M.flags & 2u) != 0u;
    float uv_x = it.b1, uv_y = it.b2;
    if (v.has_uv) {
        v.duv0[0] = r1[6] - r0[6]; v.duv0[1] = r1[7] - r0[7]; v.duv1[0] = r2[6] - r0[6]; v.duv1[1] = r2[7] - r0[7];
        uv_x = fma_(v.duv0[0], it.b1, fma_(v.duv1[0], it.b2, r0[6])); uv_y = fma_(v.duv0[1], it.b1, fma_(v.duv1[1], it.b2, r0[7]));
    }
    const DBsdf B = S.bsdfs[bsdf];             /* the record serving this side (TwoSidedBRDF picks front / back) */
    const float sign = (it.nee_flags & HAR_SHAPE_FLIPPED) ? -1.f : 1.f;
    TexTaps taps; const Vec3 rho = bsdf_reflectance(S, B, uv_x, uv_y, taps);
    Vec3 rho_du(0.f), rho_dv(0.f);
    if (B.texture >= 0) tex_fetch_grad(S.textures[B.texture], taps, rho_du, rho_dv);
    v.uv_bar[0] = 0.f; v.uv_bar[1] = 0.f;
    const bool lit = (it.nee_flags & HAR_SHAPE_LIT) != 0u;
    bool any = false;
    /* emitter sampling: sum_c dl_c W_c d(rho_c / pi cos) + sum_c dl_c W_c rho_c / pi cos dlogJ, with W_c cos / pi = dLr_drho_c */
    v.nee.on = false;
    if ((it.nee_flags & HAR_SHAPE_NEE) && visible && lit && it.cos_em > 0.f) {
        const Vec3 k = dl * dLr_drho;
        v.nee.on = true; v.nee.attached = (it.nee_flags & HAR_SHAPE_NEE_SURFACE) != 0u;
        v.nee.target = it.q; v.nee.normal = it.n_e; v.nee.w = it.w_em;
        const float s = k.x * rho.x + k.y * rho.y + k.z * rho.z;
        v.nee.cos_bar = sign * s / it.cos_em; v.nee.a = v.nee.attached ? s : 0.f;       /* cos_em = sign <w, n> */
        v.uv_bar[0] += k.x * rho_du.x + k.y * rho_du.y + k.z * rho_du.z;
        v.uv_bar[1] += k.x * rho_dv.x + k.y * rho_dv.y + k.z * rho_dv.z;
        any = true;
    }
    /* continuation: sum_c dl_c L_c (d(rho_c cos) / (rho_c cos) + dlogJ) */
    v.ind.on = false;
    if (has_next) {
        const Vec3 k = dl * L;
        const Vec3 e1 = v.p1 - v.p0, e2 = v.p2 - v.p0;
        const Vec3 n = normalize3(cross3(e1, e2));
        const float cos_ind = sign * dot3(next_d, n);
        const bool f_on = lit && cos_ind > 0.f;
        v.ind.on = true; v.ind.attached = next_valid;
        v.ind.target = next_p; v.ind.normal = next_n; v.ind.w = next_d;
        v.ind.cos_bar = 0.f;
        if (f_on) {
            float s = 0.f;
            if (rho.x != 0.f) { s += k.x; v.uv_bar[0] += k.x * rho_du.x / rho.x; v.uv_bar[1] += k.x * rho_dv.x / rho.x; }
            if (rho.y != 0.f) { s += k.y; v.uv_bar[0] += k.y * rho_du.y / rho.y; v.uv_bar[1] += k.y * rho_dv.y / rho.y; }
            if (rho.z != 0.f) { s += k.z; v.uv_bar[0] += k.z * rho_du.z / rho.z; v.uv_bar[1] += k.z * rho_dv.z / rho.z; }
            v.ind.cos_bar = sign * s / cos_ind;
        }
        v.ind.a = next_valid ? k.x + k.y + k.z : 0.f;
        any = any || k.x != 0.f || k.y != 0.f || k.z != 0.f;
    }
    if (!any) return false;
    shape_vertex_adjoint(v, g);
    return true;
}

/* The same vertex on INSTANCED geometry, differentiated w.r.t. the instance's `to_world` (Instance::compute_surface_interaction with an attached
 * transform, src/shapes/instance.cpp:150-266): the nested interaction is detached, `si.p = to_world * p_obj` carries the motion (:191-193), the
 * normals use dr::detach(to_world) and uv is not attached (:250-251), and without FollowShape the point is put back onto the ray,
 *   t = (<n, p_att> - <n, o>) / <n, d>,  p = ray(t)   (:240-249)   =>   p_att_bar = n <p_bar, d> / <n, d>,   to_world_bar = p_att_bar (x) (p_obj, 1).
 * Any vertex normals / texcoords of the nested mesh are fine (they are values here).  gM: 12 floats, column-major 3x4 like DInst::to_world. */
HAR_HD bool instance_item_adjoint(const DScene &S, const ShapeItem &it, uint32_t inst, uint32_t bsdf, bool visible, Vec3 L, Vec3 dl, Vec3 dLr_drho,
                                  bool has_next, bool next_valid, Vec3 next_p, Vec3 next_n, Vec3 next_d, float gM[12]) {
    const DMesh M = S.meshes[it.shape];
    const uint32_t *f = S.faces + 4 * (size_t) (M.foff + it.prim);
    const float *r0 = S.verts + 8 * (size_t) (M.voff + f[0]), *r1 = S.verts + 8 * (size_t) (M.voff + f[1]), *r2 = S.verts + 8 * (size_t) (M.voff + f[2]);
    const Vec3 p_obj = fma3(Vec3(r0[0], r0[1], r0[2]), 1.f - it.b1 - it.b2, fma3(Vec3(r1[0], r1[1], r1[2]), it.b1, Vec3(r2[0], r2[1], r2[2]) * it.b2));
    const SurfInt si = compute_si(S, it.d_in, 0.f, it.b1, it.b2, it.prim, it.shape, inst);      /* world-space p, geometric normal n, shading normal sn, uv */
    const DBsdf B = S.bsdfs[bsdf];
    const float sign = (it.nee_flags & HAR_SHAPE_FLIPPED) ? -1.f : 1.f;
    TexTaps taps; const Vec3 rho = bsdf_reflectance(S, B, si.uv_x, si.uv_y, taps);
    const bool lit = (it.nee_flags & HAR_SHAPE_LIT) != 0u;
    Vec3 p_bar(0.f);
    if ((it.nee_flags & HAR_SHAPE_NEE) && (it.nee_flags & HAR_SHAPE_NEE_SURFACE) && visible && lit && it.cos_em > 0.f) {
        const Vec3 k = dl * dLr_drho;
        const float s = k.x * rho.x + k.y * rho.y + k.z * rho.z;
        ShapeDirTerm T; T.on = true; T.attached = true; T.target = it.q; T.normal = it.n_e; T.w = it.w_em; T.cos_bar = sign * s / it.cos_em; T.a = s;
        p_bar = p_bar + dir_term_point_adjoint(T, si.p, si.sn);
    }
    if (has_next && next_valid) {
        const Vec3 k = dl * L;
        const float cos_ind = sign * dot3(next_d, si.sn);
        ShapeDirTerm T; T.on = true; T.attached = true; T.target = next_p; T.normal = next_n; T.w = next_d; T.cos_bar = 0.f;
        if (lit && cos_ind > 0.f) T.cos_bar = sign * ((rho.x != 0.f ? k.x : 0.f) + (rho.y != 0.f ? k.y : 0.f) + (rho.z != 0.f ? k.z : 0.f)) / cos_ind;
        T.a = k.x + k.y + k.z;
        p_bar = p_bar + dir_term_point_adjoint(T, si.p, si.sn);
    }
    if (p_bar.x == 0.f && p_bar.y == 0.f && p_bar.z == 0.f) return false;
    const Vec3 patt_bar = si.n * (dot3(p_bar, it.d_in) / dot3(si.n, it.d_in));
    const float ph[4] = { p_obj.x, p_obj.y, p_obj.z, 1.f };
    for (int c = 0; c < 4; ++c) { gM[3 * c] = patt_bar.x * ph[c]; gM[3 * c + 1] = patt_bar.y * ph[c]; gM[3 * c + 2] = patt_bar.z * ph[c]; }
    return true;
}

} // namespace har
