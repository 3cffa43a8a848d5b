/*
 * har_shape_grad.h -- vertex-position / instance-transform gradients of the PRB adjoint (HAR_HD: device kernel + host test harness).
 *
 * What the reference obtains by reverse-mode AD through the geometry-attached part of PRBIntegrator.sample
 * (src/python/python/ad/integrators/prb.py:124-141 attached surface interaction and si.wi, :176-216 emitter sampling from the attached
 * point, :261-297 attached outgoing direction and solid-angle-to-area Jacobian) is written out here by hand for one path vertex.
 * On a flat-shaded top-level triangle the differentiable quantities are
 *
 *   p_att = b0 P0 + b1 P1 + b2 P2                     (barycentrics detached, mesh.cpp:2296)
 *   n     = normalize((P1 - P0) x (P2 - P0))          (= shading normal of a mesh without vertex normals)
 *   s, t  = coordinate_system(n)                      (finalize_surface_interaction, interaction.h:570-600: no packed tangents)
 *   p     = o + d * <p_att - o, n_det> / <n_det, d>   (attach_motion without FollowShape, interaction.h:536-544: the point stays on the ray)
 *   b_i  += barycentric coordinates of (p - p_att)    (mesh.cpp:2308-2321) -> uv -> colour slot 0 (uv)
 *   wi    = to_local(-d) in the attached frame at the camera vertex; at later vertices the DETACHED frame applied to
 *           normalize(p_prev - p_det), p_prev = the previous vertex attached to ITS triangle with RayFlags::Minimal (prb.py:128-140)
 *
 * and two "direction blocks" use them, one for emitter sampling (target = the detached emitter sample) and one for the sampled
 * continuation (target = the detached next interaction): w = normalize(y - p), wo = to_local(w), J = |<m, w>| / |y - p|^2 with
 *
 *   Lr_dir = beta mis em_weight * f(wi, wo) * relative_grad(J),   Lr_ind = L * relative_grad(f(wi, wo)) * relative_grad(J),   f = BSDF value x cos.
 *
 * f may be any model (har_bsdf_dir.h supplies d f / d wi, d f / d wo); on an instance only p is attached (instance.cpp:191-193,240-251).  A vertex whose own
 * geometry does not move still contributes through wi when the PREVIOUS vertex's does.  The oracle (oracle/mi_oracle.cpp attach_si / attach_frame over dual
 * numbers, BSDF directions by finite differences) computes the same thing without any of the formulas below.
 */
#pragma once
#include "har_scene.h"
#include "har_bsdf_dir.h"

namespace har {

struct ShapeVertex {
    Vec3 p0, p1, p2;
    float b1, b2;
    Vec3 d_in;                      /* direction of the (detached) ray the vertex lies on */
    bool has_uv; float duv0[2], duv1[2];   /* texcoord differences (P1 - P0, P2 - P0) of the triangle; without texcoords uv = (b1, b2) */
    float uv_bar[2];                /* adjoint of the texture coordinates */
    float b_bar[2];                 /* direct adjoint of the attached barycentrics (b1, b2): interpolated vertex normals */
    Vec3 p_bar, n_bar;              /* adjoints of the attached point and of the attached (unit) GEOMETRIC normal */
};

/* adjoint of the point p a direction block starts from: w = normalize(y - p) carries the adjoint `w_bar` (from wo = to_local(w)) and
 * log J = log |<m, w>| - 2 log |y - p| carries `a`. */
HAR_HD Vec3 dir_point_adjoint(Vec3 target, Vec3 normal, Vec3 p, Vec3 w_bar, float a) {
    const Vec3 D = target - p;
    const float r2 = dot3(D, D), r = sqrtf(r2);
    const Vec3 u = D * rcp_(r);
    const float c = dot3(normal, u);
    Vec3 u_bar = w_bar;
    if (c != 0.f) u_bar = u_bar + normal * (a / c);        /* d log |<m, u>| / d u */
    /* u = D / |D|, D = y - p:  du = -(dp - u <u, dp>) / r;  log J also holds -2 log r, dr = -<u, dp> */
    const Vec3 proj = u_bar - u * dot3(u, u_bar);
    return u * (2.f * a / r) - proj * rcp_(r);
}
/* a * d log falloff_curve(to_world^-1 * -w) / d w of a spot light (spot.cpp:143-151, :252-259; record layout: spot_sample_direction in har_scene.h): zero inside the beam,
 * where the curve is constant; in the transition  E = (cutoff - acos(cos_theta)) / (cutoff - beam),  cos_theta = l.z,  l = m / |m|,  m = A (-w)  (A = the inverse's linear part) */
HAR_HD Vec3 spot_falloff_dir_adjoint(const DEmitter &E, Vec3 w, float a) {
    const Vec3 m = xf_vector(E.to_world, -w);
    const float len = norm3(m);
    const Vec3 l = m * rcp_(len);
    const float c = l.z;
    if (c >= E.normal[2] || !(c > E.normal[1])) return Vec3(0.f);
    const float fall = (E.normal[0] - acos_(c)) * E.inv_area;
    if (!(fall > 0.f)) return Vec3(0.f);
    const float dE_dc = E.inv_area * rcp_(sqrtf(fmaxf(1.f - c * c, 1e-30f)));        /* d (-acos c) / d c = 1 / sqrt(1 - c^2) */
    const float k = a * dE_dc / (fall * len);
    const Vec3 m_bar = (Vec3(0.f, 0.f, 1.f) - l * c) * k;                              /* d c / d m = (e_z - l c) / |m| */
    /* m = -A w: w_bar = -A^T m_bar (column-major 3 x 3 in to_world[0..8]: column j = to_world[3 j ..]) */
    return Vec3(-(E.to_world[0] * m_bar.x + E.to_world[1] * m_bar.y + E.to_world[2] * m_bar.z),
                -(E.to_world[3] * m_bar.x + E.to_world[4] * m_bar.y + E.to_world[5] * m_bar.z),
                -(E.to_world[6] * m_bar.x + E.to_world[7] * m_bar.y + E.to_world[8] * m_bar.z));
}
/* adjoint of coordinate_system(n) (vector.h:118-138): s = (sg nx^2 a + 1, sg b, -sg nx), t = (b, ny^2 a + sg, -ny), a = -1 / (sg + nz), b = nx ny a */
HAR_HD Vec3 coordinate_system_adjoint(Vec3 n, Vec3 s_bar, Vec3 t_bar) {
    const float sg = n.z >= 0.f ? 1.f : -1.f, a = -1.f / (sg + n.z);
    float a_bar = sg * n.x * n.x * s_bar.x + n.y * n.y * t_bar.y;
    const float b_bar = sg * s_bar.y + t_bar.x;
    a_bar += b_bar * n.x * n.y;
    return Vec3(2.f * sg * n.x * a * s_bar.x - sg * s_bar.z + b_bar * n.y * a, 2.f * n.y * a * t_bar.y - t_bar.z + b_bar * n.x * a, a_bar * a * a);
}

/* adds d objective / d P_k to g[k] */
HAR_HD void shape_vertex_adjoint(const ShapeVertex &v, Vec3 g[3]) {
    const float b0 = 1.f - v.b1 - v.b2;
    const Vec3 e1 = v.p1 - v.p0, e2 = v.p2 - v.p0;
    const Vec3 N = cross3(e1, e2);
    const float len = norm3(N);
    const Vec3 n = N * rcp_(len);
    Vec3 p_bar = v.p_bar; const Vec3 n_bar = v.n_bar;
    /* texture coordinates -> barycentric coordinates -> (p - p_att) */
    float b1_bar, b2_bar;
    if (v.has_uv) { b1_bar = v.uv_bar[0] * v.duv0[0] + v.uv_bar[1] * v.duv0[1]; b2_bar = v.uv_bar[0] * v.duv1[0] + v.uv_bar[1] * v.duv1[1]; }
    else { b1_bar = v.uv_bar[0]; b2_bar = v.uv_bar[1]; }
    b1_bar += v.b_bar[0]; b2_bar += v.b_bar[1];
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
    if (T.mode & 1u) { d_du = Vec3(0.f); d_dv = Vec3(0.f); return; }       /* FilterMode::Nearest: piecewise constant in uv */
    for (int c = 0; c < 3; ++c) {
        const float v00 = T.data[3 * (size_t) l.idx[0] + c], v10 = T.data[3 * (size_t) l.idx[1] + c];
        const float v01 = T.data[3 * (size_t) l.idx[2] + c], v11 = T.data[3 * (size_t) l.idx[3] + c];
        du[c] = (float) T.w * (l.w0y * (v10 - v00) + l.w1y * (v11 - v01));
        dv[c] = (float) T.h * (l.w0x * (v01 - v00) + l.w1x * (v11 - v10));
    }
    d_du = Vec3(du[0], du[1], du[2]); d_dv = Vec3(dv[0], dv[1], dv[2]);
    if (T.mode & HAR_TEX_HAS_UV_XF) {          /* the lookup ran at to_uv * (u, v): back to the surface's (u, v) by the transpose of the linear part (bitmap.cpp:591-598) */
        const Vec3 a = d_du, b = d_dv;
        d_du = a * T.uvm[0] + b * T.uvm[3]; d_dv = a * T.uvm[1] + b * T.uvm[4];
    }
}

/* geometry record of one adjoint item, written by the shading stage when vertex-position gradients are requested */
struct ShapeItem {
    uint32_t shape, prim, inst; float b1, b2;  /* the vertex: mesh, triangle, instance (0xffffffff: top-level geometry), barycentrics */
    Vec3 d_in; uint32_t next_slot;             /* 0xffffffff: the path ended at this vertex */
    Vec3 q; uint32_t nee_flags;                /* bit 0: an emitter sample exists, bit 1: it lies on a surface */
    Vec3 n_e;
    Vec3 w_em;                                 /* ds.d */
    Vec3 W;                                    /* beta * mis * em_weight: Lr_dir = W * f(wi, wo_em) */
    uint32_t prev_shape, prev_prim, prev_inst; float prev_b1, prev_b2; Vec3 prev_d;     /* the previous vertex and the ray IT lies on (prev_shape = 0xffffffff: this is the camera vertex) */
};
#define HAR_SHAPE_NEE         1u
#define HAR_SHAPE_NEE_SURFACE 2u
#define HAR_SHAPE_NEE_AT_POINT (HAR_SHAPE_NEE_SURFACE | HAR_SHAPE_NEE_POINT | HAR_SHAPE_NEE_SPOT)      /* ShapeItem::q is a POSITION (ds.p); otherwise it is the direction ds.d */
#define HAR_SHAPE_LIT         4u          /* (kept in the records for diagnostics; the side is recomputed from the geometry) */
#define HAR_SHAPE_FLIPPED     8u
#define HAR_SHAPE_NEE_POINT   16u         /* the sample lies on a point light: ds.d = normalize(ds.p - si.p) is re-attached (prb.py:191-192), no Jacobian (not a surface), and
                                           * PointLight::eval_direction divides by squared_norm(ds.p - it.p) with the attached it.p (point.cpp:155-165) */
#define HAR_SHAPE_NEE_SPOT    32u         /* ... on a spot light: the re-attached ds.d reaches falloff_curve; rcp(ds.dist) stays DETACHED (spot.cpp:252-274; ds.dist takes the zero
                                           * gradient of ds_diff, prb.py:188-193).  ShapeItem::n_e.x = bits of the emitter's record index */
#define HAR_SHAPE_NO_NEXT     0xffffffffu
#define HAR_SHAPE_NONE        0xffffffffu

/* what one vertex adds: to the three vertices of its own triangle (`self_mesh`) or to its instance's to_world (`self_inst`, 12 floats column-major 3x4
 * like DInst::to_world), and the same for the PREVIOUS vertex, which the attached si.wi follows */
struct ShapeGrad {
    bool self_mesh, self_inst, prev_mesh, prev_inst;
    Vec3 g[3]; uint32_t vid[3]; float gM[12];
    Vec3 gp[3]; uint32_t pvid[3]; float gpM[12];
    /* meshes with vertex normals: d objective / d (the three vertex normals of the triangle) -- the FIRST stage of the derivative through the regenerated normals
     * (mesh.cpp:876-878, compute_normals :1216-1267); face_normals_adjoint below is the second, run once per face after all path vertices have been added up */
    bool self_normals; Vec3 gn[3];
};

HAR_HD void shape_triangle(const DScene &S, uint32_t shape, uint32_t prim, uint32_t vid[3], const float *r[3]) {
    const DMesh M = S.meshes[shape];
    const uint32_t *f = S.faces + 4 * (size_t) (M.foff + prim);
    for (int k = 0; k < 3; ++k) { vid[k] = f[k]; r[k] = S.verts + 8 * (size_t) (M.voff + f[k]); }
}

/* One path vertex.  self_on / prev_on: the vertex's own geometry / the previous vertex's geometry is differentiated.  `visible` = the emitter sample is not
 * occluded; L = the radiance accumulator after this vertex's subtraction (prb.py:227); dl = the film adjoint; the next interaction (position / geometric
 * normal; next_valid = false for an escaped ray) is detached (prb.py:263-266).  Returns false when nothing contributes. */
HAR_HD bool shape_item_adjoint(const DScene &S, const ShapeItem &it, bool self_on, bool prev_on, bool visible, Vec3 L, Vec3 dl,
                               bool has_next, bool next_valid, Vec3 next_p, Vec3 next_n, Vec3 next_d, ShapeGrad &out, bool self_nested = false, bool prev_nested = false) {
    /* self_nested / prev_nested: the vertex lies on an INSTANCE and what moves is the nested mesh of its shape group (vertex positions shared by all instances,
     * the instance's to_world detached -- instance.cpp:162-166 refuses both at once): the nested Mesh::compute_surface_interaction is attached in OBJECT space on
     * to_object * ray (:181-189), then si.p = to_world * si.p, si.n / sh_frame.n = normalize(to_world * n) (:191-204) */
    out.self_mesh = out.self_inst = out.prev_mesh = out.prev_inst = out.self_normals = false;
    const bool depth0 = it.prev_shape == HAR_SHAPE_NONE;
    prev_on = prev_on && !depth0;
    if (!self_on && !prev_on) return false;
    const SurfInt si = compute_si(S, it.d_in, 0.f, it.b1, it.b2, it.prim, it.shape, it.inst);     /* detached values: p, n, frame, wi, uv */
    const DMesh M = S.meshes[it.shape];
    BsdfSide side; const bool side_ok = bsdf_side(S, M.bsdf, si.wi, side);
    const DBsdf B = S.bsdfs[side.index];
    TexTaps taps; const BsdfInputs bin = bsdf_inputs(S, B, si.uv_x, si.uv_y, taps);
    const bool nested = self_on && self_nested && it.inst != HAR_SHAPE_NONE;
    const bool self_mesh = self_on && (it.inst == HAR_SHAPE_NONE || nested);       /* attached triangle: p, n, frame, uv; an attached instance TRANSFORM moves p only */
    Vec3 rho_du(0.f), rho_dv(0.f);
    if (self_mesh && B.texture >= 0) tex_fetch_grad(S.textures[B.texture], taps, rho_du, rho_dv);
    SurfInt sp; Vec3 u_prev(0.f); float r_prev = 1.f;
    if (prev_on) {
        sp = compute_si(S, it.prev_d, 0.f, it.prev_b1, it.prev_b2, it.prev_prim, it.prev_shape, it.prev_inst);
        const Vec3 D = sp.p - si.p; r_prev = norm3(D); u_prev = D * rcp_(r_prev);
    }
    Vec3 p_bar(0.f), n_bar(0.f), s_bar(0.f), t_bar(0.f), pprev_bar(0.f); float uv_bar[2] = { 0.f, 0.f };
    bool any = false;
    /* ONE copy of the directional-derivative code for both terms: unrolled, the two copies of bsdf_weighted_value_dir (every BSDF model) took the kernel to 438
     * registers (accumulation registers as spill space), one wave per SIMD */
#if defined(__clang__)
#pragma clang loop unroll(disable)
#endif
    for (int term = 0; term < 2; ++term) {
        bool attached; Vec3 w, target, normal, A; float a = 0.f;
        BsdfEval e; e.value = Vec3(0.f); e.d_slot0 = Vec3(0.f);
        if (term == 0) {              /* emitter sampling: sum_c dl_c W_c (d f_c + f_c dlogJ) */
            if (!((it.nee_flags & HAR_SHAPE_NEE) && visible)) continue;
            w = it.w_em; attached = (it.nee_flags & HAR_SHAPE_NEE_AT_POINT) != 0u; target = it.q;
            normal = (it.nee_flags & HAR_SHAPE_NEE_SURFACE) ? it.n_e : Vec3(0.f);           /* a zero normal: dir_point_adjoint keeps the direction and the -2 log r, drops the cosine */
        } else {                      /* continuation: sum_c dl_c L_c (d f_c / f_c + dlogJ) */
            if (!has_next) continue;
            w = next_d; attached = next_valid; target = next_p; normal = next_n;
        }
        const Vec3 wo_l = si.to_local(w), wo_side(wo_l.x, wo_l.y, wo_l.z * side.wo_sign);
        if (side_ok) bsdf_eval_pdf_one(B, bin, side.wi, wo_side, e);
        if (term == 0) { A = dl * it.W; a = A.x * e.value.x + A.y * e.value.y + A.z * e.value.z; }
        else {
            const Vec3 k = dl * L;
            A = Vec3(e.value.x != 0.f ? k.x / e.value.x : 0.f, e.value.y != 0.f ? k.y / e.value.y : 0.f, e.value.z != 0.f ? k.z / e.value.z : 0.f);
            a = k.x + k.y + k.z;
        }
        if (!attached || !self_on) a = 0.f;
        float g6[6] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
        if (side_ok && (A.x != 0.f || A.y != 0.f || A.z != 0.f)) { bsdf_weighted_value_dir(B, bin, side.wi, wo_side, A, g6); g6[2] *= side.wo_sign; g6[5] *= side.wo_sign; }
        const Vec3 gwi(g6[0], g6[1], g6[2]), gwo(g6[3], g6[4], g6[5]);
        /* wo = to_local(w) */
        if (self_mesh) {
            s_bar = s_bar + w * gwo.x; t_bar = t_bar + w * gwo.y; n_bar = n_bar + w * gwo.z;
            const Vec3 c = A * e.d_slot0;                        /* colour slot 0 (uv) */
            uv_bar[0] += c.x * rho_du.x + c.y * rho_du.y + c.z * rho_du.z; uv_bar[1] += c.x * rho_dv.x + c.y * rho_dv.y + c.z * rho_dv.z;
        }
        if (self_on && attached) {
            Vec3 w_bar = si.ss * gwo.x + si.st * gwo.y + si.sn * gwo.z;
            if (term == 0 && (it.nee_flags & HAR_SHAPE_NEE_SPOT)) { w_bar = w_bar + spot_falloff_dir_adjoint(S.emitters[as_u32(it.n_e.x)], w, a); a = 0.f; }
            p_bar = p_bar + dir_point_adjoint(target, normal, si.p, w_bar, a);
        }
        /* wi */
        if (depth0) { if (self_mesh) { const Vec3 wv = -it.d_in; s_bar = s_bar + wv * gwi.x; t_bar = t_bar + wv * gwi.y; n_bar = n_bar + wv * gwi.z; } }
        else if (prev_on) { const Vec3 u_bar = si.ss * gwi.x + si.st * gwi.y + si.sn * gwi.z; pprev_bar = pprev_bar + (u_bar - u_prev * dot3(u_prev, u_bar)) * rcp_(r_prev); }
        any = true;
    }
    if (!any) return false;
    bool some = false;
    if (self_mesh) {
        const float *r[3]; shape_triangle(S, it.shape, it.prim, out.vid, r);
        ShapeVertex v;
        v.p0 = Vec3(r[0][0], r[0][1], r[0][2]); v.p1 = Vec3(r[1][0], r[1][1], r[1][2]); v.p2 = Vec3(r[2][0], r[2][1], r[2][2]);
        v.b1 = it.b1; v.b2 = it.b2; v.d_in = it.d_in;
        v.has_uv = (M.flags & 2u) != 0u;
        if (v.has_uv) { v.duv0[0] = r[1][6] - r[0][6]; v.duv0[1] = r[1][7] - r[0][7]; v.duv1[0] = r[2][6] - r[0][6]; v.duv1[1] = r[2][7] - r[0][7]; }
        v.uv_bar[0] = uv_bar[0]; v.uv_bar[1] = uv_bar[1]; v.b_bar[0] = 0.f; v.b_bar[1] = 0.f;
        Vec3 sn_bar = n_bar + coordinate_system_adjoint(si.sn, s_bar, t_bar);          /* adjoint of the (unit) SHADING normal */
        /* the unit shading normal in the MESH's space: the interpolated vertex normals, or the face normal */
        const Vec3 n0(r[0][3], r[0][4], r[0][5]), n1(r[1][3], r[1][4], r[1][5]), n2(r[2][3], r[2][4], r[2][5]);
        const Vec3 m = (M.flags & 1u) ? fma3(n1 - n0, it.b1, fma3(n2 - n0, it.b2, n0)) : cross3(v.p1 - v.p0, v.p2 - v.p0);
        Vec3 sn_mesh = si.sn;
        v.p_bar = p_bar;
        if (nested) {
            /* world <- object: p_w = A_w p_o + t, sn_w = normalize(A_o^T sn_o), d_o = A_o d_w  (A_w / A_o: linear parts of to_world / to_object) */
            const DInst &I = S.insts[it.inst];
            sn_mesh = normalize3(m);
            const float scale = norm3(xf_normal(I.to_object, sn_mesh));
            sn_bar = xf_vector(I.to_object, (sn_bar - si.sn * dot3(si.sn, sn_bar)) * rcp_(scale));
            v.p_bar = xf_normal(I.to_world, p_bar);
            v.d_in = xf_vector(I.to_object, it.d_in);
        }
        v.n_bar = sn_bar;
        if (M.flags & 1u) {
            /* mesh.cpp:2346-2356: sn = normalize(m), m = n0 + (n1 - n0) b1 + (n2 - n0) b2 over the attached barycentrics and the attached (regenerated) vertex
             * normals; the geometric normal no longer reaches any term */
            const Vec3 m_bar = (sn_bar - sn_mesh * dot3(sn_mesh, sn_bar)) * rcp_(norm3(m));
            out.gn[0] = m_bar * (1.f - it.b1 - it.b2); out.gn[1] = m_bar * it.b1; out.gn[2] = m_bar * it.b2;
            out.self_normals = true;
            v.b_bar[0] = dot3(m_bar, n1 - n0); v.b_bar[1] = dot3(m_bar, n2 - n0);
            v.n_bar = Vec3(0.f);
        }
        out.g[0] = out.g[1] = out.g[2] = Vec3(0.f);
        shape_vertex_adjoint(v, out.g);
        out.self_mesh = true; some = true;
    } else if (self_on && (p_bar.x != 0.f || p_bar.y != 0.f || p_bar.z != 0.f)) {
        /* Instance::compute_surface_interaction with an attached transform (instance.cpp:150-266): si.p = to_world * p_obj carries the motion (:191-193), and
         * without FollowShape the point is put back onto the ray, t = (<n, p_att> - <n, o>) / <n, d>, p = ray(t) (:240-249) */
        uint32_t vid[3]; const float *r[3]; shape_triangle(S, it.shape, it.prim, vid, r);
        const float b0 = 1.f - it.b1 - it.b2;
        const float ph[4] = { r[0][0] * b0 + r[1][0] * it.b1 + r[2][0] * it.b2, r[0][1] * b0 + r[1][1] * it.b1 + r[2][1] * it.b2, r[0][2] * b0 + r[1][2] * it.b1 + r[2][2] * it.b2, 1.f };
        const Vec3 patt_bar = si.n * (dot3(p_bar, it.d_in) / dot3(si.n, it.d_in));
        for (int c = 0; c < 4; ++c) { out.gM[3 * c] = patt_bar.x * ph[c]; out.gM[3 * c + 1] = patt_bar.y * ph[c]; out.gM[3 * c + 2] = patt_bar.z * ph[c]; }
        out.self_inst = true; some = true;
    }
    if (prev_on && (pprev_bar.x != 0.f || pprev_bar.y != 0.f || pprev_bar.z != 0.f)) {
        /* pi_prev.compute_surface_interaction(ray_prev, RayFlags::Minimal): only the point is attached -- p = ray_prev(t), t = <p_att - o, n_det> / <n_det, d> */
        const float *r[3]; shape_triangle(S, it.prev_shape, it.prev_prim, out.pvid, r);
        const float b0 = 1.f - it.prev_b1 - it.prev_b2;
        const Vec3 patt_bar = sp.n * (dot3(pprev_bar, it.prev_d) / dot3(sp.n, it.prev_d));
        if (it.prev_inst == HAR_SHAPE_NONE) { out.gp[0] = patt_bar * b0; out.gp[1] = patt_bar * it.prev_b1; out.gp[2] = patt_bar * it.prev_b2; out.prev_mesh = true; }
        else if (prev_nested) {
            /* the nested mesh of the previous vertex's instance moves: p_w = to_world * ray_obj(t), t = <p_att - o, n_o> / <n_o, d_o> in object space */
            const DInst &I = S.insts[it.prev_inst];
            const Vec3 q0(r[0][0], r[0][1], r[0][2]), q1(r[1][0], r[1][1], r[1][2]), q2(r[2][0], r[2][1], r[2][2]);
            const Vec3 n_o = cross3(q1 - q0, q2 - q0), d_o = xf_vector(I.to_object, it.prev_d);
            const Vec3 po = n_o * (dot3(pprev_bar, it.prev_d) / dot3(n_o, d_o));
            out.gp[0] = po * b0; out.gp[1] = po * it.prev_b1; out.gp[2] = po * it.prev_b2; out.prev_mesh = true;
        } else {
            const float ph[4] = { r[0][0] * b0 + r[1][0] * it.prev_b1 + r[2][0] * it.prev_b2, r[0][1] * b0 + r[1][1] * it.prev_b1 + r[2][1] * it.prev_b2, r[0][2] * b0 + r[1][2] * it.prev_b1 + r[2][2] * it.prev_b2, 1.f };
            for (int c = 0; c < 4; ++c) { out.gpM[3 * c] = patt_bar.x * ph[c]; out.gpM[3 * c + 1] = patt_bar.y * ph[c]; out.gpM[3 * c + 2] = patt_bar.z * ph[c]; }
            out.prev_inst = true;
        }
        some = true;
    }
    return some;
}

/* ---- second stage of the vertex-normal derivative.  Written in DOUBLE precision: real meshes hold slivers (the last ring of a UV sphere has triangles whose pole
 * edge is 1e-17 long) whose corner terms have derivatives of 1 / edge that cancel or meet a zero adjoint only in exact arithmetic -- in fp32 they come out as
 * inf - inf.  One thread per face, once per render: the fp64 rate is irrelevant. */
struct D3 { double x, y, z; };
HAR_HD D3 d3(Vec3 v) { return D3{ (double) v.x, (double) v.y, (double) v.z }; }
HAR_HD D3 operator+(D3 a, D3 b) { return D3{ a.x + b.x, a.y + b.y, a.z + b.z }; }
HAR_HD D3 operator-(D3 a, D3 b) { return D3{ a.x - b.x, a.y - b.y, a.z - b.z }; }
HAR_HD D3 operator-(D3 a) { return D3{ -a.x, -a.y, -a.z }; }
HAR_HD D3 operator*(D3 a, double s) { return D3{ a.x * s, a.y * s, a.z * s }; }
HAR_HD double ddot3(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
HAR_HD D3 dcross3(D3 a, D3 b) { return D3{ a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
/* dr::unit_angle(u, v) of two unit vectors: 2 asin(|v - u| / 2), or pi - 2 asin(|v + u| / 2) for an obtuse angle (restated from its published definition) */
HAR_HD double unit_angle_d(D3 u, D3 v) {
    const bool acute = ddot3(u, v) >= 0.0;
    const D3 w = acute ? v - u : v + u;
    const double t = 2.0 * asin(fmin(0.5 * sqrt(ddot3(w, w)), 1.0));
    return acute ? t : 3.14159265358979323846 - t;
}
/* Mesh::compute_normals (mesh.cpp:1216-1267) adds  c_k = n_face * angle_k  to the sum of the face's k-th vertex (n_face = N / |N|, N = (P1 - P0) x (P2 - P0);
 * angle_k = unit_angle(normalize(P_{k+1} - P_k), normalize(P_{k+2} - P_k))); false for a face without area, which contributes nothing */
HAR_HD bool face_corner_normals(const Vec3 Pf[3], Vec3 c[3]) {
    const D3 P[3] = { d3(Pf[0]), d3(Pf[1]), d3(Pf[2]) };
    const D3 N = dcross3(P[1] - P[0], P[2] - P[0]);
    const double l2 = ddot3(N, N);
    if (!(l2 > 0.0)) return false;
    const D3 n = N * (1.0 / sqrt(l2));
    for (int k = 0; k < 3; ++k) {
        const D3 a = P[(k + 1) % 3] - P[k], b = P[(k + 2) % 3] - P[k];
        const double angle = unit_angle_d(a * (1.0 / sqrt(ddot3(a, a))), b * (1.0 / sqrt(ddot3(b, b))));
        c[k] = Vec3((float) (n.x * angle), (float) (n.y * angle), (float) (n.z * angle));
    }
    return true;
}
/* ... and its reverse: given the adjoints a_bar[k] of the three sums the face adds to, g[j] += d objective / d P_j */
HAR_HD void face_normals_adjoint(const Vec3 Pf[3], const Vec3 a_bar_f[3], Vec3 g[3]) {
    const D3 P[3] = { d3(Pf[0]), d3(Pf[1]), d3(Pf[2]) }, a_bar[3] = { d3(a_bar_f[0]), d3(a_bar_f[1]), d3(a_bar_f[2]) };
    const D3 e1 = P[1] - P[0], e2 = P[2] - P[0], N = dcross3(e1, e2);
    const double l2 = ddot3(N, N);
    if (!(l2 > 0.0)) return;
    const double len = sqrt(l2);
    const D3 n = N * (1.0 / len);
    D3 n_bar{ 0.0, 0.0, 0.0 }, gd[3] = { D3{ 0.0, 0.0, 0.0 }, D3{ 0.0, 0.0, 0.0 }, D3{ 0.0, 0.0, 0.0 } };
    for (int k = 0; k < 3; ++k) {
        const int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
        const D3 a = P[k1] - P[k], b = P[k2] - P[k];
        const double la = sqrt(ddot3(a, a)), lb = sqrt(ddot3(b, b));
        const D3 u = a * (1.0 / la), v = b * (1.0 / lb);
        const bool acute = ddot3(u, v) >= 0.0;
        const D3 w = acute ? v - u : v + u;
        const double r = sqrt(ddot3(w, w)), s = fmin(0.5 * r, 1.0), t = 2.0 * asin(s), angle = acute ? t : 3.14159265358979323846 - t;
        n_bar = n_bar + a_bar[k] * angle;
        const double angle_bar = ddot3(n, a_bar[k]);
        if (!(r > 0.0) || angle_bar == 0.0) continue;
        /* angle = +-2 asin(r / 2) (+ pi): d angle / d r = +-1 / sqrt(1 - r^2 / 4) */
        const double r_bar = (acute ? angle_bar : -angle_bar) / sqrt(fmax(1.0 - s * s, 1e-300));
        const D3 w_bar = w * (r_bar / r);
        const D3 v_bar = w_bar, u_bar = acute ? -w_bar : w_bar;
        const D3 ab = (u_bar - u * ddot3(u, u_bar)) * (1.0 / la), bb = (v_bar - v * ddot3(v, v_bar)) * (1.0 / lb);
        gd[k1] = gd[k1] + ab; gd[k2] = gd[k2] + bb; gd[k] = gd[k] - (ab + bb);
    }
    const D3 N_bar = (n_bar - n * ddot3(n, n_bar)) * (1.0 / len);
    const D3 e1_bar = dcross3(e2, N_bar), e2_bar = dcross3(N_bar, e1);
    gd[1] = gd[1] + e1_bar; gd[2] = gd[2] + e2_bar; gd[0] = gd[0] - (e1_bar + e2_bar);
    for (int k = 0; k < 3; ++k) g[k] = g[k] + Vec3((float) gd[k].x, (float) gd[k].y, (float) gd[k].z);
}

} // namespace har
