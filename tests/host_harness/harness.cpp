/*
 * harness.cpp -- HOST TEST HARNESS (test tool, not a product path).
 *
 * Compiles the product's HAR_HD device functions (headers of mitsuba3_amd/csrc: BVH8
 * traversal, surface interaction, shading stages, film footprint) and its host
 * scene lowering with g++ and drives them lane by lane, so that the logic the
 * HIP kernels execute can be checked against the oracle on a machine without
 * a GPU.  The shipped library (libhip_ad_rgb.so) never contains or calls this.
 */
#include "../../mitsuba3_amd/csrc/har_cpu.h"
#include "../../mitsuba3_amd/csrc/har_path.h"
#include "../../mitsuba3_amd/csrc/har_shape_grad.h"
#include "../../mitsuba3_amd/csrc/har_scene_host.h"
#include "../../mitsuba3_amd/csrc/har_vertex_update.h"
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <string>
#include <vector>

using namespace har;

struct HostStack {
    static constexpr int Capacity = 24;
    uint32_t x[Capacity], y[Capacity];
    void push(int l, uint32_t a, uint32_t b) { x[l] = a; y[l] = b; }
    void pop(int l, uint32_t &a, uint32_t &b) { a = x[l]; b = y[l]; }
};

struct HScene {
    HostScene hs;
    std::vector<DTexture> dtex;
    DScene ds;
};

static void bind(HScene &H) {
    HostScene &hs = H.hs;
    H.dtex.clear();
    for (size_t k = 0; k < hs.textures.size(); ++k) H.dtex.push_back(hs.device_texture(k, hs.textures[k].data.data()));
    DScene &S = H.ds;
    S.accel.nodes = hs.nodes.data(); S.accel.tris = hs.tris.data(); S.accel.insts = hs.inst_recs.data(); S.accel.mesh_info = nullptr;
    S.accel.root = hs.root; S.accel.has_tlas = hs.has_tlas; S.accel.n_tris = (uint32_t) hs.tris.size(); S.accel.n_insts = (uint32_t) hs.inst_recs.size();
    S.accel.top_root = hs.top_root; S.accel.top_first = hs.top_first; S.accel.top_count = hs.top_count; S.accel.top_last = hs.top_last;
    S.blas_tri_ranges = hs.blas_tri_ranges.data();
#if HAR_SHADING_TRIS
    S.shade_tris = hs.shade_tris.data();
#endif
    S.verts = hs.verts.data(); S.faces = hs.faces.data(); S.meshes = hs.meshes.data(); S.bsdfs = hs.bsdfs.data();
    S.textures = H.dtex.data(); S.emitters = hs.emitters.data(); S.insts = hs.insts.data(); S.bsdf_tables = hs.bsdf_tables.data();
    S.n_emitters = (uint32_t) hs.emitters.size(); S.n_meshes = (uint32_t) hs.meshes.size();
    S.n_bsdfs = (uint32_t) hs.bsdfs.size(); S.n_insts = (uint32_t) hs.insts.size(); S.n_textures = (uint32_t) hs.textures.size();
    S.env_emitter = hs.env_emitter;
    S.bsdf_types = 0; for (const DBsdf &b : hs.bsdfs) S.bsdf_types |= (1u << b.type) | ((b.flags & BF_TWOSIDED) ? 0x80000000u : 0u);
    S.envmap = nullptr; S.emitter_cdf = hs.emitter_cdf.data();
    hs.bind_tables(S, hs.emitter_distr.data());
    if (hs.has_mesh_emitters || hs.has_point_emitters || !hs.emitter_distr.empty()) S.bsdf_types |= HAR_SCENE_ENVMAP;
    if (hs.has_envmap) { hs.envmap.tex = hs.env_tex.data(); hs.envmap.warp = hs.env_warp.data(); S.envmap = &hs.envmap; S.bsdf_types |= HAR_SCENE_ENVMAP; }
}

extern "C" {

void *hh_scene_create(const HarSceneDesc *d, char *err, int errlen) {
    HScene *H = new HScene();
    std::string e;
    if (!lower_scene(*d, H->hs, e)) { snprintf(err, errlen, "%s", e.c_str()); delete H; return nullptr; }
    bind(*H);
    return H;
}
void hh_scene_destroy(void *h) { delete (HScene *) h; }
/* Emitter::sample_direction of emitter `index` alone through the product's shading headers (the dispatch of shade_lane, har_path.h): reference points p[n][3], samples
 * s[n][2] -> d[n][3], dist[n], pdf[n], delta[n], weight[n][3] (the counterpart of orc_emitter_sample_direction) */
void hh_emitter_sample_direction(void *h, uint32_t index, uint32_t n, const float *p, const float *s, float *d, float *dist, float *pdf, uint8_t *delta, float *weight) {
    HScene *H = (HScene *) h; const DScene &S = H->ds;
    for (uint32_t i = 0; i < n; ++i) {
        DirSample ds; ds.pdf = 0.f; ds.d = Vec3(0.f); ds.p = Vec3(0.f); ds.n = Vec3(0.f); ds.dist = 0.f;
        Vec3 w(0.f); bool dl = false;
        const Vec3 ref(p[3 * i], p[3 * i + 1], p[3 * i + 2]);
        const DEmitter &E = S.emitters[index];
        if (E.type == 2u) envmap_sample_direction(*S.envmap, ref, s[2 * i], s[2 * i + 1], ds, w);
        else if (E.type == 3u) mesh_emitter_sample_direction(S, E, ref, s[2 * i], s[2 * i + 1], ds, w, nullptr);
        else if (E.type == 4u) { point_sample_direction(E, ref, ds, w, nullptr); dl = true; }
        else if (E.type == 5u) { spot_sample_direction(E, ref, ds, w, nullptr); dl = true; }
        else if (E.type == 6u) { directional_sample_direction(E, ref, ds, w, nullptr); dl = true; }
        else emitter_sample_direction(E, ref, s[2 * i], s[2 * i + 1], ds, w, nullptr);
        d[3 * i] = ds.d.x; d[3 * i + 1] = ds.d.y; d[3 * i + 2] = ds.d.z; dist[i] = ds.dist; pdf[i] = ds.pdf; delta[i] = dl ? 1 : 0;
        weight[3 * i] = w.x; weight[3 * i + 1] = w.y; weight[3 * i + 2] = w.z;
    }
}
/* the incremental updates of include/hip_ad_rgb.h on the host arrays (the refit runs the kernels' HAR_HD code sequentially): 0 ok, 2 = needs a new scene, 1 = error */
int hh_scene_update_instances(void *h, uint32_t first, uint32_t count, const float *to_world, const float *to_object, char *err, int errlen) {
    HScene *H = (HScene *) h; std::string e;
    if (!scene_set_instances_host(H->hs, first, count, to_world, to_object, e)) { snprintf(err, errlen, "%s", e.c_str()); return 1; }
    bind(*H);
    return 0;
}
int hh_scene_update_vertices(void *h, uint32_t mesh, const float *vertices, double *area, char *err, int errlen) {
    HScene *H = (HScene *) h; std::string e;
    BlasInfo *B = scene_set_vertices_host(H->hs, mesh, vertices, e);
    if (!B) { snprintf(err, errlen, "%s", e.c_str()); return 2; }
    const double a = refit_blas_host(H->hs, *B);
    if (area) *area = a;
    if (!scene_after_refit_host(H->hs, B, e)) { snprintf(err, errlen, "%s", e.c_str()); return 1; }
    bind(*H);
    return 0;
}
/* har_scene_update_vertices_device on the host arrays: the per-element code of the three update kernels (har_vertex_update.h: positions into the packed records, the
 * per-vertex gather of Mesh::compute_normals over the mesh's corner list, the shading triangles), run sequentially, then the refit.  positions: vertex_count x 3 */
int hh_scene_update_positions(void *h, uint32_t mesh, const float *positions, double *area, char *err, int errlen) {
    HScene *H = (HScene *) h; HostScene &hs = H->hs; std::string e;
    if (mesh >= hs.meshes.size()) { snprintf(err, errlen, "invalid mesh index"); return 1; }
    const DMesh m = hs.meshes[mesh];
    if (m.emitter >= 0) { snprintf(err, errlen, "the mesh carries an area emitter"); return 2; }
    float *V = hs.verts.data() + 8 * (size_t) m.voff; const uint32_t *F = hs.faces.data() + 4 * (size_t) m.foff;
    for (uint32_t v = 0; v < m.vertex_count; ++v) for (int a = 0; a < 3; ++a) V[8 * (size_t) v + a] = positions[3 * (size_t) v + a];
    if (m.flags & 1u) {
        std::vector<uint32_t> begin((size_t) m.vertex_count + 1, 0u), corners(3 * (size_t) m.face_count);
        for (uint32_t f = 0; f < m.face_count; ++f) for (int k = 0; k < 3; ++k) ++begin[(size_t) F[4 * (size_t) f + k] + 1];
        for (uint32_t v = 0; v < m.vertex_count; ++v) begin[v + 1] += begin[v];
        std::vector<uint32_t> cursor(begin.begin(), begin.end() - 1);
        for (uint32_t f = 0; f < m.face_count; ++f) for (int k = 0; k < 3; ++k) corners[cursor[F[4 * (size_t) f + k]]++] = f | ((uint32_t) k << 30);
        for (uint32_t v = 0; v < m.vertex_count; ++v) vertex_normal(V, F, begin.data(), corners.data(), v);
    }
#if HAR_SHADING_TRIS
    for (uint32_t f = 0; f < m.face_count; ++f) shading_triangle(V, F, hs.shade_tris.data() + 24 * (size_t) m.foff, f);
#endif
    BlasInfo *B = mesh < hs.top_mesh_count ? &hs.blas_top : nullptr;
    for (size_t g = 0; !B && g < hs.groups.size(); ++g) if (mesh >= hs.groups[g].first_mesh && mesh < hs.groups[g].first_mesh + hs.groups[g].mesh_count) B = &hs.blas_groups[g];
    if (!B) { snprintf(err, errlen, "mesh belongs to no BLAS"); return 1; }
    const double a = refit_blas_host(hs, *B);
    if (area) *area = a;
    if (!scene_after_refit_host(hs, B, e)) { snprintf(err, errlen, "%s", e.c_str()); return 1; }
    bind(*H);
    return 0;
}
/* the same for a mesh INSIDE a shape group as the device does it when no emitter follows the scene bounds: BLAS refit, then the instance level REFITTED (instance boxes from
 * the vertices, TLAS nodes deepest level first: refit_tlas_host = the per-element code of k_instance_boxes / k_refit_nodes) instead of rebuilt */
int hh_scene_update_positions_instanced(void *h, uint32_t mesh, const float *positions, char *err, int errlen) {
    HScene *H = (HScene *) h; HostScene &hs = H->hs;
    if (mesh < hs.top_mesh_count || mesh >= hs.meshes.size()) { snprintf(err, errlen, "not a mesh of a shape group"); return 1; }
    const DMesh m = hs.meshes[mesh];
    float *V = hs.verts.data() + 8 * (size_t) m.voff; const uint32_t *F = hs.faces.data() + 4 * (size_t) m.foff;
    for (uint32_t v = 0; v < m.vertex_count; ++v) for (int a = 0; a < 3; ++a) V[8 * (size_t) v + a] = positions[3 * (size_t) v + a];
    if (m.flags & 1u) {
        std::vector<uint32_t> begin((size_t) m.vertex_count + 1, 0u), corners(3 * (size_t) m.face_count);
        for (uint32_t f = 0; f < m.face_count; ++f) for (int k = 0; k < 3; ++k) ++begin[(size_t) F[4 * (size_t) f + k] + 1];
        for (uint32_t v = 0; v < m.vertex_count; ++v) begin[v + 1] += begin[v];
        std::vector<uint32_t> cursor(begin.begin(), begin.end() - 1);
        for (uint32_t f = 0; f < m.face_count; ++f) for (int k = 0; k < 3; ++k) corners[cursor[F[4 * (size_t) f + k]]++] = f | ((uint32_t) k << 30);
        for (uint32_t v = 0; v < m.vertex_count; ++v) vertex_normal(V, F, begin.data(), corners.data(), v);
    }
#if HAR_SHADING_TRIS
    for (uint32_t f = 0; f < m.face_count; ++f) shading_triangle(V, F, hs.shade_tris.data() + 24 * (size_t) m.foff, f);
#endif
    BlasInfo *B = nullptr;
    for (size_t g = 0; !B && g < hs.groups.size(); ++g) if (mesh >= hs.groups[g].first_mesh && mesh < hs.groups[g].first_mesh + hs.groups[g].mesh_count) B = &hs.blas_groups[g];
    if (!B) { snprintf(err, errlen, "mesh belongs to no BLAS"); return 1; }
    (void) refit_blas_host(hs, *B);
    refit_tlas_host(hs);
    bind(*H);
    return 0;
}
/* packed vertex records of a mesh as the harness scene holds them */
int hh_scene_get_vertices(void *h, uint32_t mesh, float *out) {
    HostScene &hs = ((HScene *) h)->hs;
    if (mesh >= hs.meshes.size()) return 1;
    const DMesh &m = hs.meshes[mesh];
    std::memcpy(out, hs.verts.data() + 8 * (size_t) m.voff, 32 * (size_t) m.vertex_count);
    return 0;
}
/* Scene::sample_emitter / pdf_emitter of the product's shading headers (scene_sample_emitter, har_scene.h) and the weight update of har_scene_set_emitter_sampling_weights */
void hh_scene_sample_emitter(void *h, uint32_t n, const float *sample, int jit, uint32_t *index, float *weight, float *reused) {
    const DScene &S = ((HScene *) h)->ds;
    for (uint32_t k = 0; k < n; ++k) index[k] = scene_sample_emitter(S, sample[k], jit != 0, weight[k], reused[k]);
}
void hh_scene_pdf_emitter(void *h, uint32_t n, const uint32_t *index, float *pdf) {
    const DScene &S = ((HScene *) h)->ds;
    for (uint32_t k = 0; k < n; ++k) pdf[k] = index[k] < S.n_emitters ? scene_pdf_emitter(S, index[k]) : 0.f;
}
int hh_scene_set_emitter_weights(void *h, const float *w, uint32_t n, char *err, int errlen) {
    HScene *H = (HScene *) h; std::string e;
    if (n != H->hs.emitters.size()) { snprintf(err, errlen, "one weight per emitter"); return 1; }
    if (!build_emitter_distribution(H->hs, w, n, e)) { snprintf(err, errlen, "%s", e.c_str()); return 1; }
    bind(*H);
    return 0;
}
/* FNV-1a over the bytes of the node array, the triangle records and the instance records: the builder's output, for tests that compare builds */
void hh_accel_hash(void *h, uint64_t out[3]) {
    HScene *H = (HScene *) h;
    auto fnv = [](const void *p, size_t n) { uint64_t x = 1469598103934665603ull; const unsigned char *b = (const unsigned char *) p; for (size_t i = 0; i < n; ++i) { x ^= b[i]; x *= 1099511628211ull; } return x; };
    out[0] = fnv(H->hs.nodes.data(), H->hs.nodes.size() * sizeof(Node8));
    out[1] = fnv(H->hs.tris.data(), H->hs.tris.size() * sizeof(TriRec));
    out[2] = fnv(H->hs.inst_recs.data(), H->hs.inst_recs.size() * sizeof(InstRec));
}
void hh_scene_info(void *h, uint64_t info[4]) {
    HScene *H = (HScene *) h; info[0] = H->hs.nodes.size(); info[1] = H->hs.tris.size(); info[2] = H->hs.stats.max_depth; info[3] = H->hs.inst_recs.size();
}

/* product BSDF code on the host: BSDF::eval_pdf / sample of scene BSDF `bsdf` (twosided handled) */
void hh_bsdf_eval_pdf(void *h, uint32_t bsdf, const float wi[3], const float uv[2], const float wo[3], float value[3], float *pdf) {
    HScene *H = (HScene *) h; const DScene &S = H->ds;
    BsdfSide side; bool ok = bsdf_side(S, bsdf, Vec3(wi[0], wi[1], wi[2]), side);
    TexTaps taps; BsdfInputs in = bsdf_inputs(S, S.bsdfs[side.index], uv[0], uv[1], taps);
    BsdfEval e; bsdf_eval_pdf(S, side, in, ok, Vec3(wo[0], wo[1], wo[2]), e);
    value[0] = e.value.x; value[1] = e.value.y; value[2] = e.value.z; *pdf = e.pdf;
}
void hh_bsdf_sample(void *h, uint32_t bsdf, const float wi[3], const float uv[2], float s1, const float s2[2], float wo[3], float *pdf, float weight[3], float *eta, int *delta) {
    HScene *H = (HScene *) h; const DScene &S = H->ds;
    BsdfSide side; bool ok = bsdf_side(S, bsdf, Vec3(wi[0], wi[1], wi[2]), side);
    TexTaps taps; BsdfInputs in = bsdf_inputs(S, S.bsdfs[side.index], uv[0], uv[1], taps);
    BsdfSample b; bsdf_sample(S, side, in, ok, s1, s2[0], s2[1], b);
    wo[0] = b.wo.x; wo[1] = b.wo.y; wo[2] = b.wo.z; *pdf = b.pdf; weight[0] = b.weight.x; weight[1] = b.weight.y; weight[2] = b.weight.z; *eta = b.eta; *delta = b.delta;
}
/* ... with the BSDFContext (mode, type_mask, component) and the Mask argument, exactly as k_api_bsdf_eval_pdf / k_api_bsdf_sample (har_kernels.hip) do it */
void hh_bsdf_eval_pdf_ctx(void *h, uint32_t bsdf, uint32_t mode, uint32_t type_mask, uint32_t component, int active, const float wi[3], const float uv[2], const float wo[3],
                          float value[3], float *pdf) {
    HScene *H = (HScene *) h; const DScene &S = H->ds;
    BsdfCtx ctx; ctx.mode = mode; ctx.type_mask = type_mask; ctx.component = component;
    BsdfEval e; e.value = Vec3(0.f); e.pdf = 0.f;
    if (active) {
        BsdfSide side; bool ok = bsdf_side(S, bsdf, Vec3(wi[0], wi[1], wi[2]), side);
        TexTaps taps; BsdfInputs in = bsdf_inputs(S, S.bsdfs[side.index], uv[0], uv[1], taps);
        bsdf_eval_pdf<HAR_BSDF_ALL_TYPES, true>(S, side, in, ok, Vec3(wo[0], wo[1], wo[2]), e, bsdf_side_ctx(S, bsdf, side, ctx));
    }
    value[0] = e.value.x; value[1] = e.value.y; value[2] = e.value.z; *pdf = e.pdf;
}
void hh_bsdf_sample_ctx(void *h, uint32_t bsdf, uint32_t mode, uint32_t type_mask, uint32_t component, int active, const float wi[3], const float uv[2], float s1, const float s2[2],
                        float wo[3], float *pdf, float weight[3], float *eta, uint32_t *stype, uint32_t *scomp) {
    HScene *H = (HScene *) h; const DScene &S = H->ds;
    BsdfCtx ctx; ctx.mode = mode; ctx.type_mask = type_mask; ctx.component = component;
    BsdfSample b; b.wo = Vec3(0.f); b.pdf = 0.f; b.weight = Vec3(0.f); b.eta = 0.f; b.delta = false; b.type = 0u; b.comp = 0u;
    if (active) {
        BsdfSide side; bool ok = bsdf_side(S, bsdf, Vec3(wi[0], wi[1], wi[2]), side);
        TexTaps taps; BsdfInputs in = bsdf_inputs(S, S.bsdfs[side.index], uv[0], uv[1], taps);
        bsdf_sample<HAR_BSDF_ALL_TYPES, true>(S, side, in, ok, s1, s2[0], s2[1], b, bsdf_side_ctx(S, bsdf, side, ctx));
    }
    wo[0] = b.wo.x; wo[1] = b.wo.y; wo[2] = b.wo.z; *pdf = b.pdf; weight[0] = b.weight.x; weight[1] = b.weight.y; weight[2] = b.weight.z; *eta = b.eta; *stype = b.type; *scomp = b.comp;
}
/* compute_si + compute_si_partials under RayFlags and a mask, exactly as k_api_si (har_kernels.hip): out[33] */
void hh_surface_interaction_flags(void *h, const float d[3], float t, float u, float v, uint32_t prim, uint32_t shape, uint32_t inst, uint32_t ray_flags, int active, float out[33]) {
    HScene *H = (HScene *) h; const DScene &S = H->ds;
    const Vec3 D(d[0], d[1], d[2]);
    const bool shading = (ray_flags & RAY_SHADING) != 0u, valid = active && t != HAR_INF;
    Vec3 vs[6] = { Vec3(0.f), Vec3(0.f), Vec3(0.f), Vec3(0.f), Vec3(0.f), Vec3(0.f) };
    float uvx = 0.f, uvy = 0.f;
    SurfPartials P; P.dp_du = Vec3(0.f); P.dp_dv = Vec3(0.f); P.dn_du = Vec3(0.f); P.dn_dv = Vec3(0.f);
    if (valid) {
        const SurfInt si = compute_si(S, D, t, u, v, prim, shape, inst);
        vs[0] = si.p; vs[1] = si.n;
        if (shading) { vs[2] = si.sn; vs[3] = si.ss; vs[4] = si.st; vs[5] = si.wi; uvx = si.uv_x; uvy = si.uv_y; compute_si_partials(S, u, v, prim, shape, inst, (ray_flags & RAY_NORMAL_PARTIALS) != 0u, P); }
    } else if (shading) { coordinate_system(Vec3(0.f), vs[3], vs[4]); vs[5] = -D; }
    for (int k = 0; k < 6; ++k) { out[3 * k] = vs[k].x; out[3 * k + 1] = vs[k].y; out[3 * k + 2] = vs[k].z; }
    out[18] = uvx; out[19] = uvy; out[20] = valid ? t : HAR_INF;
    const Vec3 ps[4] = { P.dp_du, P.dp_dv, P.dn_du, P.dn_dv };
    for (int k = 0; k < 4; ++k) { out[21 + 3 * k] = ps[k].x; out[22 + 3 * k] = ps[k].y; out[23 + 3 * k] = ps[k].z; }
}
void hh_microfacet_eval(int type, float alpha_u, float alpha_v, int sample_visible, const float wi[3], const float m[3], float out[3]) {
    Microfacet d(type != 0, alpha_u, alpha_v, sample_visible != 0);
    Vec3 w(wi[0], wi[1], wi[2]), mm(m[0], m[1], m[2]);
    out[0] = d.eval(mm); out[1] = d.pdf(w, mm); out[2] = d.smith_g1(w, mm);
}
void hh_microfacet_sample(int type, float alpha_u, float alpha_v, int sample_visible, const float wi[3], const float sample[2], float m[3], float *pdf) {
    Microfacet d(type != 0, alpha_u, alpha_v, sample_visible != 0);
    Vec3 r = d.sample(Vec3(wi[0], wi[1], wi[2]), sample[0], sample[1], *pdf);
    m[0] = r.x; m[1] = r.y; m[2] = r.z;
}
/* product's hand-derived d value / d {alpha_u, alpha_v, eta, k} of a rough BSDF record built from the arguments (har_bsdf.h bsdf_eval_extra_one) next to
 * the value itself, so that the test can difference the value in its parameters */
void hh_bsdf_eval_extra(int type, int ggx, int sample_visible, float alpha_u, float alpha_v, float eta, const float eta_c[3], const float k_c[3], const float slot0[3],
                        const float slot1[3], const float wi[3], const float wo[3], float value[3], float out[12]) {
    DBsdf B{}; B.type = (uint32_t) type; B.texture = -1; B.flags = (ggx ? BF_GGX : 0u) | (sample_visible ? BF_SAMPLE_VISIBLE : 0u);
    B.alpha_u = alpha_u; B.alpha_v = alpha_v; B.eta = eta; B.back = -1; B.table = -1;
    for (int k = 0; k < 3; ++k) { B.eta_c[k] = eta_c[k]; B.k_c[k] = k_c[k]; }
    B.inv_eta_2 = 1.f / (eta * eta); B.internal_reflectance = 0.f; B.spec_sampling_weight = .5f;
    static float table[64]; for (int i = 0; i < 64; ++i) table[i] = 1.f;          /* the (detached) transmittance table does not matter for the derivatives */
    BsdfInputs in; in.slot0 = Vec3(slot0[0], slot0[1], slot0[2]); in.slot1 = Vec3(slot1[0], slot1[1], slot1[2]); in.table = table;
    BsdfEval e; bsdf_eval_pdf_one(B, in, Vec3(wi[0], wi[1], wi[2]), Vec3(wo[0], wo[1], wo[2]), e);
    value[0] = e.value.x; value[1] = e.value.y; value[2] = e.value.z;
    BsdfEvalExtra x; bsdf_eval_extra_one(B, in, Vec3(wi[0], wi[1], wi[2]), Vec3(wo[0], wo[1], wo[2]), x);
    const Vec3 g[4] = { x.d_alpha_u, x.d_alpha_v, x.d_eta, x.d_k };
    for (int k = 0; k < 4; ++k) { out[3 * k] = g[k].x; out[3 * k + 1] = g[k].y; out[3 * k + 2] = g[k].z; }
}
void hh_fresnel(float cos_theta_i, float eta, float out[4]) { fresnel_dielectric(cos_theta_i, eta, out[0], out[1], out[2], out[3]); }
float hh_fresnel_conductor(float cos_theta_i, float eta, float k) { return fresnel_conductor(cos_theta_i, eta, k); }
/* har_cpu.h: the worker-thread count the product's host code uses for "all cores" */
unsigned hh_usable_cores() { return har_usable_cores(); }
/* elementary functions of har_math.h / har_bsdf.h, same numbering as orc_math_fn */
float hh_math_fn(int fn, float x, float y) {
    switch (fn) {
        case 0: return exp_(x);   case 1: return log_(x);  case 2: return erf_(x); case 3: return atan2_(x, y);
        case 4: return acos_(x);  case 5: return tan_(x);  case 6: return erfinv_(x);
        case 7: { float sn, cs; sincos_(x, sn, cs); return sn; }  case 8: { float sn, cs; sincos_(x, sn, cs); return cs; }
    }
    return 0.f;
}
void hh_math_fn_array(int fn, uint32_t n, const float *x, const float *y, float *out) { for (uint32_t i = 0; i < n; ++i) out[i] = hh_math_fn(fn, x[i], y ? y[i] : 0.f); }
void hh_coordinate_system(const float n[3], float s[3], float t[3]) {      /* coordinate_system of har_math.h (vector.h:118-138) */
    Vec3 a, b; coordinate_system(Vec3(n[0], n[1], n[2]), a, b);
    s[0] = a.x; s[1] = a.y; s[2] = a.z; t[0] = b.x; t[1] = b.y; t[2] = b.z;
}
void hh_gauss_legendre(int n, float *nodes, float *weights) {       /* the product's host-side quadrature rule (har_scene_host.cpp) */
    std::vector<float> a, b; quad_gauss_legendre(n, a, b);
    for (int i = 0; i < n; ++i) { nodes[i] = a[i]; weights[i] = b[i]; }
}
void hh_roughplastic_tables(void *h, uint32_t bsdf, float out[66]) {
    HScene *H = (HScene *) h; const DBsdf &b = H->hs.bsdfs[bsdf];
    for (int i = 0; i < 64; ++i) out[i] = b.table >= 0 ? H->hs.bsdf_tables[b.table + i] : 0.f;
    out[64] = b.internal_reflectance; out[65] = b.spec_sampling_weight;
}

int hh_trace(void *h, uint32_t n, const float *o, const float *d, const float *maxt, int naive, int anyhit,
             float *t, float *u, float *v, uint32_t *prim, uint32_t *shape, uint32_t *inst, uint8_t *hitflag) {
    HScene *H = (HScene *) h; int status = 0;
    for (uint32_t i = 0; i < n; ++i) {
        Vec3 O(o[i], o[n + i], o[2 * (size_t) n + i]), D(d[i], d[n + i], d[2 * (size_t) n + i]);
        Hit hit; HostStack st; bool r;
        if (naive >= 2) {
            /* the persistent kernels' resumable traversal, stepped to the end: 2 = the instantiation the launchers pick (FLAT for a scene without a TLAS),
             * 3 = always the generic one */
            const Accel &A = H->ds.accel; const bool tl = (A.top_last & (anyhit ? 1u : 2u)) != 0u;
            auto run = [&](auto &T) {
                T.begin(A, O, D, maxt[i], tl);
                if (anyhit) { while (!T.template step<true, HostStack, NoProbe, 0>(A, st, status)) { } }
                else        { while (!T.template step<false, HostStack, NoProbe, 0>(A, st, status)) { } }
                hit = T.hit; return T.found;
            };
            if (naive == 2 && !A.has_tlas) { Traversal<0, true> T; r = run(T); } else { Traversal<0, false> T; r = run(T); }
        }
        else if (naive) r = anyhit ? accel_trace_naive<true>(H->ds.accel, H->ds.blas_tri_ranges, O, D, maxt[i], hit)
                              : accel_trace_naive<false>(H->ds.accel, H->ds.blas_tri_ranges, O, D, maxt[i], hit);
        else       r = anyhit ? accel_trace<true>(H->ds.accel, O, D, maxt[i], hit, st, status)
                              : accel_trace<false>(H->ds.accel, O, D, maxt[i], hit, st, status);
        if (anyhit) hitflag[i] = r;
        else { t[i] = hit.t; u[i] = hit.u; v[i] = hit.v; prim[i] = hit.prim; shape[i] = hit.shape; inst[i] = hit.inst; }
    }
    return status;
}

/* lane-by-lane emulation of the wavefront pipeline (raygen -> {trace, shade, shadow}* -> splat) */
int hh_render(void *h, const HarSensor *sensor, int mode, uint32_t seed, uint32_t spp, int32_t max_depth, int32_t rr_depth,
              uint64_t lane_begin, uint64_t lane_end, float *film) {
    HScene *H = (HScene *) h; const DScene &S = H->ds;
    DSensor C; std::string e; if (!lower_sensor(*sensor, C, e)) return -1;
    uint64_t total = (uint64_t) C.samp_w * C.samp_h * spp;       /* the sample grid: crop + border with Film::sample_border */
    if (lane_begin == 0 && lane_end == 0) lane_end = total;
    uint32_t log_spp = 0xffffffffu; for (uint32_t k = 0; k < 32; ++k) if ((1u << k) == spp) log_spp = k;
    ShadeParams P{ seed, (uint32_t) max_depth, (uint32_t) rr_depth };
    int status = 0;
    for (uint64_t lane = lane_begin; lane < lane_end; ++lane) {
        LaneSample ls; PathState st = raygen_lane(C, seed, spp, log_spp, (uint32_t) lane, ls);
        Vec3 result(0.f);
        bool alive = P.max_depth != 0;
        while (alive) {
            Hit hit; HostStack stack;
            accel_trace<false>(S.accel, st.o, st.d, st.maxt, hit, stack, status);
            ShadeResult R;
            constexpr uint32_t ENV = HAR_BSDF_ALL_TYPES | HAR_SCENE_ENVMAP | HAR_SCENE_TEXLIGHT;
            if (S.bsdf_types & HAR_SCENE_ENVMAP) { if (mode == MODE_PATH) shade_lane<MODE_PATH, ENV>(S, P, st, hit, R); else shade_lane<MODE_PRB_PRIMAL, ENV>(S, P, st, hit, R); }
            else if (mode == MODE_PATH) shade_lane<MODE_PATH>(S, P, st, hit, R); else shade_lane<MODE_PRB_PRIMAL>(S, P, st, hit, R);
            if (R.add_emission) result = mode == MODE_PATH ? fma3(R.em_a, R.em_b, result) : result + R.em_b;
            if (R.item && R.item_ray) {
                Hit sh; HostStack s2;
                if (!accel_trace<true>(S.accel, R.sh_o, R.sh_d, R.sh_maxt, sh, s2, status)) result = result + R.contrib;
            }
            alive = R.alive; st = R.next;
        }
        LaneSample fp = lane_film_pos(C, seed, spp, log_spp, (uint32_t) lane);
        Footprint F; film_footprint(C, fp, F);
        const float val[4] = { result.x, result.y, result.z, 1.f };
        for (uint32_t ys = 0; ys < F.count; ++ys) for (uint32_t xs = 0; xs < F.count; ++xs) {
            uint32_t x = F.x0 + xs, y = F.y0 + ys;
            if (x < C.crop_w && y < C.crop_h) { float w = F.wx[xs] * F.wy[ys]; float *p = film + 4 * ((size_t) y * C.crop_w + x); for (int k = 0; k < 4; ++k) p[k] += val[k] * w; }
        }
    }
    return status;
}

/* ReconstructionFilter::eval of the product's HAR_HD code (any filter type of HarSensor) */
float hh_rfilter_eval(const HarSensor *sensor, float x) {
    DSensor C; std::string e; if (!lower_sensor(*sensor, C, e)) return -1e30f;
    return C.rfilter == 0 ? (x >= -.5f && x < .5f ? 1.f : 0.f) : rfilter_eval(C, x);
}

/* vertex-position gradients: the lane-by-lane equivalent of (k_shade<ADJOINT, diffuse, SHAPE> -> k_resolve -> k_shape_adjoint), i.e. the product's
 * hand-derived adjoint (har_shape_grad.h) driven exactly as the kernels drive it.  adj = grad_in / W (H x W x 3); grad[m] = 3 doubles per vertex
 * of mesh m or NULL. */
static int backward_shape_impl(void *h, const HarSensor *sensor, const float *adj, uint32_t seed, uint32_t spp, int32_t max_depth, int32_t rr_depth,
                               double *const *grad, double *inst_grad) {
    HScene *H = (HScene *) h; const DScene &S = H->ds;
    DSensor C; std::string e; if (!lower_sensor(*sensor, C, e)) return -1;
    const uint64_t total = (uint64_t) C.samp_w * C.samp_h * spp;
    uint32_t log_spp = 0xffffffffu; for (uint32_t k = 0; k < 32; ++k) if ((1u << k) == spp) log_spp = k;
    ShadeParams P{ seed, (uint32_t) max_depth, (uint32_t) rr_depth };
    int status = 0;
    /* meshes with vertex normals: first-stage adjoints of the vertex normals (3 per vertex), pushed through compute_normals after the lane loop */
    std::vector<std::vector<double>> nbar(H->hs.meshes.size());
    if (grad) for (size_t m = 0; m < H->hs.meshes.size(); ++m) if (grad[m] && (H->hs.meshes[m].flags & 1u)) nbar[m].assign(3 * (size_t) H->hs.meshes[m].vertex_count, 0.0);
    for (uint64_t lane = 0; lane < total; ++lane) {
        LaneSample ls; const PathState st0 = raygen_lane(C, seed, spp, log_spp, (uint32_t) lane, ls);
        Footprint F; film_footprint(C, ls, F);
        Vec3 dl(0.f);
        for (uint32_t ys = 0; ys < F.count; ++ys) for (uint32_t xs = 0; xs < F.count; ++xs) {
            uint32_t x = F.x0 + xs, y = F.y0 + ys;
            if (x < C.crop_w && y < C.crop_h) { float w = F.wx[xs] * F.wy[ys]; const float *a = adj + 3 * ((size_t) y * C.crop_w + x); dl = Vec3(fma_(a[0], w, dl.x), fma_(a[1], w, dl.y), fma_(a[2], w, dl.z)); }
        }
        /* primal pass: L */
        Vec3 L(0.f);
        {
            PathState st = st0; bool alive = P.max_depth != 0;
            while (alive) {
                Hit hit; HostStack stack; accel_trace<false>(S.accel, st.o, st.d, st.maxt, hit, stack, status);
                ShadeResult R;
                if (S.bsdf_types & HAR_SCENE_ENVMAP) shade_lane<MODE_PRB_PRIMAL, HAR_BSDF_ALL_TYPES | HAR_SCENE_ENVMAP | HAR_SCENE_TEXLIGHT>(S, P, st, hit, R);       /* mesh / delta / textured lights, environment maps, sampling weights */
                else shade_lane<MODE_PRB_PRIMAL, HAR_BSDF_ALL_TYPES>(S, P, st, hit, R);
                if (R.add_emission) L = L + R.em_b;
                if (R.item && R.item_ray) { Hit sh; HostStack s2; if (!accel_trace<true>(S.accel, R.sh_o, R.sh_d, R.sh_maxt, sh, s2, status)) L = L + R.contrib; }
                alive = R.alive; st = R.next;
            }
        }
        /* adjoint replay */
        PathState st = st0; bool alive = P.max_depth != 0;
        Hit hit; { HostStack stack; if (alive) accel_trace<false>(S.accel, st.o, st.d, st.maxt, hit, stack, status); }
        Hit prev; prev.t = HAR_INF; prev.shape = HAR_SHAPE_NONE; prev.prim = 0; prev.inst = HAR_SHAPE_NONE; prev.u = prev.v = 0.f; Vec3 prev_d(0.f);
        /* a vertex on an instance moves with the instance's to_world (inst_grad) or with the nested mesh of its shape group (grad[nested mesh]) */
        auto nested_on = [&](uint32_t shape, uint32_t inst) { return shape != HAR_SHAPE_NONE && inst != HAR_SHAPE_NONE && grad && grad[shape]; };
        auto moving = [&](uint32_t shape, uint32_t inst) { return shape != HAR_SHAPE_NONE && (inst == HAR_SHAPE_NONE ? (grad && grad[shape]) : (inst_grad != nullptr || nested_on(shape, inst))); };
        while (alive) {
            ShadeResult R;
            if (S.bsdf_types & HAR_SCENE_ENVMAP) shade_lane<MODE_PRB_ADJOINT, HAR_BSDF_ALL_TYPES | HAR_SCENE_ENVMAP | HAR_SCENE_TEXLIGHT>(S, P, st, hit, R);
            else shade_lane<MODE_PRB_ADJOINT, HAR_BSDF_ALL_TYPES>(S, P, st, hit, R);
            if (R.add_emission) L = L - R.em_b;
            bool visible = false;
            if (R.item && R.item_ray) { Hit sh; HostStack s2; visible = !accel_trace<true>(S.accel, R.sh_o, R.sh_d, R.sh_maxt, sh, s2, status); if (visible) L = L - R.contrib; }
            Hit next; bool next_valid = false; Vec3 np(0.f), nn(0.f);
            if (R.alive) {
                HostStack stack; accel_trace<false>(S.accel, R.next.o, R.next.d, R.next.maxt, next, stack, status);
                next_valid = next.t != HAR_INF;
                if (next_valid) { SurfInt sn = compute_si(S, R.next.d, next.t, next.u, next.v, next.prim, next.shape, next.inst); np = sn.p; nn = sn.n; }
            }
            const bool self_on = hit.t != HAR_INF && moving(hit.shape, hit.inst), prev_on = moving(prev.shape, prev.inst);
            if ((R.item || R.alive) && hit.t != HAR_INF && (self_on || prev_on)) {       /* the record k_shade<SHAPE> writes, the call k_shape_adjoint makes */
                ShapeItem it; it.shape = hit.shape; it.prim = hit.prim; it.inst = hit.inst; it.b1 = hit.u; it.b2 = hit.v; it.d_in = st.d;
                it.next_slot = R.alive ? 0u : HAR_SHAPE_NO_NEXT;
                it.q = R.nee_p; it.n_e = R.nee_n; it.nee_flags = R.item_ray ? R.nee_flags : (R.nee_flags & HAR_SHAPE_LIT); it.W = R.nee_w;
                it.prev_shape = prev.shape; it.prev_prim = prev.prim; it.prev_inst = prev.inst; it.prev_b1 = prev.u; it.prev_b2 = prev.v; it.prev_d = prev_d;
                const SurfInt si = compute_si(S, st.d, hit.t, hit.u, hit.v, hit.prim, hit.shape, hit.inst);
                it.w_em = (it.nee_flags & HAR_SHAPE_NEE_AT_POINT) ? normalize3(it.q - si.p) : it.q;
                ShapeGrad G;
                if (shape_item_adjoint(S, it, self_on, prev_on, visible, L, dl, R.alive, next_valid, np, nn, R.next.d, G, nested_on(hit.shape, hit.inst), nested_on(prev.shape, prev.inst))) {
                    if (G.self_mesh) { double *dst = grad[hit.shape]; for (int k = 0; k < 3; ++k) { dst[3 * (size_t) G.vid[k]] += G.g[k].x; dst[3 * (size_t) G.vid[k] + 1] += G.g[k].y; dst[3 * (size_t) G.vid[k] + 2] += G.g[k].z; } }
                    if (G.self_normals && !nbar[hit.shape].empty()) { double *dst = nbar[hit.shape].data(); for (int k = 0; k < 3; ++k) { dst[3 * (size_t) G.vid[k]] += G.gn[k].x; dst[3 * (size_t) G.vid[k] + 1] += G.gn[k].y; dst[3 * (size_t) G.vid[k] + 2] += G.gn[k].z; } }
                    if (G.self_inst) for (int k = 0; k < 12; ++k) inst_grad[12 * (size_t) hit.inst + k] += G.gM[k];
                    if (G.prev_mesh) { double *dst = grad[prev.shape]; for (int k = 0; k < 3; ++k) { dst[3 * (size_t) G.pvid[k]] += G.gp[k].x; dst[3 * (size_t) G.pvid[k] + 1] += G.gp[k].y; dst[3 * (size_t) G.pvid[k] + 2] += G.gp[k].z; } }
                    if (G.prev_inst) for (int k = 0; k < 12; ++k) inst_grad[12 * (size_t) prev.inst + k] += G.gpM[k];
                }
            }
            if (hit.t != HAR_INF) { prev = hit; prev_d = st.d; }
            alive = R.alive; st = R.next; hit = next;
        }
    }
    /* second stage (k_normals_sums + k_normals_adjoint on the device): n_v = normalize(sum of the corner contributions), face by face */
    for (size_t m = 0; m < nbar.size(); ++m) {
        if (nbar[m].empty()) continue;
        const DMesh &M = S.meshes[m];
        std::vector<Vec3> acc(M.vertex_count, Vec3(0.f));
        auto tri = [&](uint32_t f, uint32_t vid[3], Vec3 Pt[3]) {
            const uint32_t *fi = S.faces + 4 * (size_t) (M.foff + f);
            for (int k = 0; k < 3; ++k) { vid[k] = fi[k]; const float *r = S.verts + 8 * (size_t) (M.voff + fi[k]); Pt[k] = Vec3(r[0], r[1], r[2]); }
        };
        for (uint32_t f = 0; f < M.face_count; ++f) { uint32_t vid[3]; Vec3 Pt[3], c[3]; tri(f, vid, Pt); if (face_corner_normals(Pt, c)) for (int k = 0; k < 3; ++k) acc[vid[k]] = acc[vid[k]] + c[k]; }
        for (uint32_t f = 0; f < M.face_count; ++f) {
            uint32_t vid[3]; Vec3 Pt[3], ab[3]; tri(f, vid, Pt);
            for (int k = 0; k < 3; ++k) {
                const Vec3 a = acc[vid[k]]; const float l2 = dot3(a, a);
                const double *b = &nbar[m][3 * (size_t) vid[k]]; const Vec3 nb((float) b[0], (float) b[1], (float) b[2]);
                if (!(l2 > 0.f)) { ab[k] = Vec3(0.f); continue; }
                const float il = rsqrt_(l2); const Vec3 n = a * il;
                ab[k] = (nb - n * dot3(n, nb)) * il;
            }
            Vec3 g[3] = { Vec3(0.f), Vec3(0.f), Vec3(0.f) };
            face_normals_adjoint(Pt, ab, g);
            for (int k = 0; k < 3; ++k) { double *dst = grad[m] + 3 * (size_t) vid[k]; dst[0] += g[k].x; dst[1] += g[k].y; dst[2] += g[k].z; }
        }
    }
    return status;
}

int hh_render_backward_shape(void *h, const HarSensor *sensor, const float *adj, uint32_t seed, uint32_t spp, int32_t max_depth, int32_t rr_depth,
                             double *const *grad) {
    return backward_shape_impl(h, sensor, adj, seed, spp, max_depth, rr_depth, grad, nullptr);
}
/* instance to_world gradients: inst_grad = 12 doubles per instance (column-major 3x4) */
int hh_render_backward_instances(void *h, const HarSensor *sensor, const float *adj, uint32_t seed, uint32_t spp, int32_t max_depth, int32_t rr_depth,
                                 double *inst_grad) {
    return backward_shape_impl(h, sensor, adj, seed, spp, max_depth, rr_depth, nullptr, inst_grad);
}

/* gradients w.r.t. the texels of bitmap-radiance area lights: what k_shade<ADJOINT, .. | TEXLIGHT, INLINE> commits under HAR_SHADE_LIGHT_TEXELS, lane by lane on the host --
 * shade_lane's lt_* fields, em_unit / contrib_unit and the visibility of the emitter sample; grad_tex[t] = 3 doubles per texel of texture t or NULL.  adj = grad_in / W. */
int hh_render_backward_light_texels(void *h, const HarSensor *sensor, const float *adj, uint32_t seed, uint32_t spp, int32_t max_depth, int32_t rr_depth, double *const *grad_tex) {
    HScene *H = (HScene *) h; const DScene &S = H->ds;
    DSensor C; std::string e; if (!lower_sensor(*sensor, C, e)) return -1;
    const uint64_t total = (uint64_t) C.samp_w * C.samp_h * spp;
    uint32_t log_spp = 0xffffffffu; for (uint32_t k = 0; k < 32; ++k) if ((1u << k) == spp) log_spp = k;
    ShadeParams P{ seed, (uint32_t) max_depth, (uint32_t) rr_depth, HAR_SHADE_LIGHT_TEXELS };
    int status = 0;
    auto scatter = [&](int32_t index, const float uv[2], Vec3 g) {
        const uint32_t t = as_u32(S.emitters[index].radiance[0]);
        if (!grad_tex[t]) return;
        TexTaps taps; tex_taps(S.textures[t], uv[0], uv[1], taps);
        const float w[4] = { taps.w0x * taps.w0y, taps.w1x * taps.w0y, taps.w0x * taps.w1y, taps.w1x * taps.w1y };
        for (int k = 0; k < 4; ++k) { double *q = grad_tex[t] + 3 * (size_t) taps.idx[k]; q[0] += (double) (g.x * w[k]); q[1] += (double) (g.y * w[k]); q[2] += (double) (g.z * w[k]); }
    };
    for (uint64_t lane = 0; lane < total; ++lane) {
        LaneSample ls; PathState st = raygen_lane(C, seed, spp, log_spp, (uint32_t) lane, ls);
        Footprint F; film_footprint(C, ls, F);
        Vec3 dl(0.f);
        for (uint32_t ys = 0; ys < F.count; ++ys) for (uint32_t xs = 0; xs < F.count; ++xs) {
            uint32_t x = F.x0 + xs, y = F.y0 + ys;
            if (x < C.crop_w && y < C.crop_h) { float w = F.wx[xs] * F.wy[ys]; const float *a = adj + 3 * ((size_t) y * C.crop_w + x); dl = Vec3(fma_(a[0], w, dl.x), fma_(a[1], w, dl.y), fma_(a[2], w, dl.z)); }
        }
        bool alive = P.max_depth != 0;
        while (alive) {
            Hit hit; HostStack stack; accel_trace<false>(S.accel, st.o, st.d, st.maxt, hit, stack, status);
            ShadeResult R; shade_lane<MODE_PRB_ADJOINT, HAR_BSDF_ALL_TYPES | HAR_SCENE_ENVMAP | HAR_SCENE_TEXLIGHT>(S, P, st, hit, R);
            if (R.add_emission && R.lt_hit_emitter >= 0) scatter(R.lt_hit_emitter, R.lt_hit_uv, R.em_unit * dl);
            if (R.item && R.item_ray && R.lt_nee_emitter >= 0) {
                Hit sh; HostStack s2;
                if (!accel_trace<true>(S.accel, R.sh_o, R.sh_d, R.sh_maxt, sh, s2, status)) scatter(R.lt_nee_emitter, R.lt_nee_uv, R.contrib_unit * dl);
            }
            alive = R.alive; st = R.next;
        }
    }
    return status;
}

/* --- traversal statistics (tools/trace_stats.py): per-ray event counts and a lock-step SIMT model of the
 * static traversal kernel.  Rays of bounce b are the rays of all lanes alive at bounce b, in lane order
 * (the order the compacting wavefront keeps), grouped into waves of 64. */
/* statistics of the reference loop (accel_trace), [0] closest-hit / [1] any-hit queries: rays, node visits / triangle tests of the top-level BLAS phase, node visits in the
 * TLAS and inside instances, triangle tests and entries of instances, node visits that hit no child, ... whose own box lies beyond the current tmax, instance entries that
 * gave the ray nothing and their node visits.  Filled through the Probe interface of har_accel.h (this used to sit inside the product header). */
enum { HS_RAYS = 0, HS_TOP_NODES, HS_TOP_TRIS, HS_TLAS_NODES, HS_INST_NODES, HS_INST_TRIS, HS_INST_ENTRIES, HS_EMPTY_NODES, HS_STALE_NODES, HS_FALSE_ENTRIES, HS_FALSE_ENTRY_NODES, HS_ENTRY_MARK };
static unsigned long long g_host_stat[2][12] = { { 0 }, { 0 } };
static int g_host_child_order = 0;      /* what-if: 1 = back-to-front */
static std::vector<unsigned long long> g_node_hist[2];      /* visits per node index (closest-hit / any-hit): which nodes a per-block LDS copy would have to hold */
struct StatHooks {
    int q = 0;
    void ray(bool any_hit) { q = any_hit ? 1 : 0; ++g_host_stat[q][HS_RAYS]; }
    void visited(const Accel &A, const RaySetup &R, float tmax, uint32_t child, uint32_t ng_y, uint32_t tg_y, int phase) {
        ++g_host_stat[q][phase == 0 ? HS_TOP_NODES : phase == 1 ? HS_TLAS_NODES : HS_INST_NODES];
        if (g_node_hist[q].size() <= child) g_node_hist[q].resize((size_t) child + 1, 0ull);
        ++g_node_hist[q][child];
        /* is the node's own (quantisation-frame) box beyond the current tmax, i.e. was it queued under an older tmax? */
        const Node8 &N = A.nodes[child];
        const float lo[3] = { N.px, N.py, N.pz }, sc[3] = { as_f32((uint32_t) N.ex << 23), as_f32((uint32_t) N.ey << 23), as_f32((uint32_t) N.ez << 23) };
        const float oo[3] = { R.o.x, R.o.y, R.o.z }, id[3] = { R.idir.x, R.idir.y, R.idir.z };
        float tn = 0.f, tf = tmax;
        for (int a = 0; a < 3; ++a) { float t0 = (lo[a] - oo[a]) * id[a], t1 = (lo[a] + 255.f * sc[a] - oo[a]) * id[a]; if (t0 > t1) { float w = t0; t0 = t1; t1 = w; } tn = fmaxf(tn, t0); tf = fminf(tf, t1); }
        if (tn > tf) ++g_host_stat[q][HS_STALE_NODES];
        if (ng_y <= 0x00ffffffu && tg_y == 0u) ++g_host_stat[q][HS_EMPTY_NODES];        /* a visit that hit none of the node's children */
    }
    void leaf(int phase) { ++g_host_stat[q][phase == 0 ? HS_TOP_TRIS : HS_INST_TRIS]; }
    void entered() { ++g_host_stat[q][HS_INST_ENTRIES]; g_host_stat[q][HS_ENTRY_MARK] = g_host_stat[q][HS_INST_NODES]; }
    void left(bool top_phase, bool useful) {
        if (!top_phase && !useful) { ++g_host_stat[q][HS_FALSE_ENTRIES]; g_host_stat[q][HS_FALSE_ENTRY_NODES] += g_host_stat[q][HS_INST_NODES] - g_host_stat[q][HS_ENTRY_MARK]; }
    }
    bool back_to_front() const { return g_host_child_order == 1; }
};
struct EvProbe : StatHooks {
    std::vector<uint32_t> *ev;     /* one entry per outer iteration: bit0 node, bit1 inst, bits 8.. triangle tests */
    explicit EvProbe(std::vector<uint32_t> *e) : ev(e) {}
    void iter() { ev->push_back(0u); }
    void node() { ev->back() |= 1u; }
    void inst() { ev->back() |= 2u; }
    void tri()  { ev->back() += 256u; }
};

/* out[b*32 + ..]: closest rays in 0..15, shadow rays in 16..31:
 * 0 rays, 1 iters, 2 nodes, 3 tris, 4 insts, 5 wave_steps, 6 wave_node_blocks, 7 wave_tri_blocks, 8 wave_inst_blocks, 9 waves, 10 mismatches
 * policy: -1 reference loop (all triangles of a node per iteration), 0 / 1 = Traversal<POLICY>;
 * refill: 0 = static waves of 64 consecutive rays, R > 0 = persistent wave that refills when >= R lanes are idle */
} // extern "C"
static int g_max_sp = 0;
static double g_regroup_stat[2][8] = { { 0 }, { 0 } };      /* [closest / shadow]: node issues, triangle issues, instance issues, rounds, rays, ops */
template <bool AnyHit, typename T, int ORDER>
static void run_traversal_o(const Accel &A, Vec3 o, Vec3 d, float maxt, Hit &hit, bool &found, std::vector<uint32_t> &ev, int &status) {
    T tr; HostStack stack; EvProbe pr{ &ev };
    tr.begin(A, o, d, maxt, (A.top_last & (AnyHit ? 1u : 2u)) != 0u);      /* HAR_TOP_LAST=0..3 selects the order (har_scene_host.cpp) */
    while (!tr.template step<AnyHit, HostStack, EvProbe, ORDER>(A, stack, status, pr)) { if (tr.sp > g_max_sp) g_max_sp = tr.sp; }
    hit = tr.hit; found = tr.found;
}
static int g_order = 2;
template <bool AnyHit, typename T>
static void run_traversal(const Accel &A, Vec3 o, Vec3 d, float maxt, Hit &hit, bool &found, std::vector<uint32_t> &ev, int &status) {
    if (g_order == 0) run_traversal_o<AnyHit, T, 0>(A, o, d, maxt, hit, found, ev, status);
    else if (g_order == 1) run_traversal_o<AnyHit, T, 1>(A, o, d, maxt, hit, found, ev, status);
    else run_traversal_o<AnyHit, T, 2>(A, o, d, maxt, hit, found, ev, status);
}
extern "C" {

void hh_set_order(int o) { g_order = o; g_max_sp = 0; }
void hh_regroup_stats(double out[16]) { for (int k = 0; k < 2; ++k) for (int j = 0; j < 8; ++j) { out[8 * k + j] = g_regroup_stat[k][j]; g_regroup_stat[k][j] = 0.0; } }
void hh_top_phase_stats(double out[24]) { for (int q = 0; q < 2; ++q) for (int k = 0; k < 12; ++k) out[12 * q + k] = (double) g_host_stat[q][k]; }
int hh_max_sp() { return g_max_sp; }
/* visits per node of query class q since the process started; returns the number of entries (call with out = nullptr first) */
uint64_t hh_node_hist(int q, unsigned long long *out, uint64_t cap) {
    const auto &h = g_node_hist[q ? 1 : 0];
    if (out) for (size_t i = 0; i < h.size() && i < cap; ++i) out[i] = h[i];
    return h.size();
}
/* layout of the node array for the same tool: first TLAS node (or 0xffffffff), BLAS count, then per BLAS root / node count */
uint32_t hh_node_layout(void *h, uint32_t *out, uint32_t cap) {
    HScene *H = (HScene *) h; const HostScene &hs = H->hs;
    uint32_t k = 0;
    auto put = [&](uint32_t v) { if (out && k < cap) out[k] = v; ++k; };
    put(hs.has_tlas ? hs.tlas_first : 0xffffffffu); put((uint32_t) hs.nodes.size()); put((uint32_t) hs.blas_groups.size() + (hs.blas_top.node_count ? 1u : 0u));
    if (hs.blas_top.node_count) { put(hs.blas_top.root); put(hs.blas_top.node_count); }
    for (const auto &b : hs.blas_groups) { put(b.root); put(b.node_count); }
    put(hs.stack_need());
    return k;
}
int hh_trace_stats(void *h, const HarSensor *sensor, uint32_t seed, uint32_t spp, int32_t max_depth, int32_t rr_depth,
                   uint64_t lane_begin, uint64_t lane_end, uint32_t max_bounces, int policy, int refill, double *out) {
    HScene *H = (HScene *) h; const DScene &S = H->ds;
    DSensor C; std::string e; if (!lower_sensor(*sensor, C, e)) return -1;
    uint32_t log_spp = 0xffffffffu; for (uint32_t k = 0; k < 32; ++k) if ((1u << k) == spp) log_spp = k;
    ShadeParams P{ seed, (uint32_t) max_depth, (uint32_t) rr_depth };
    std::vector<PathState> cur, next; LaneSample ls;
    for (uint64_t lane = lane_begin; lane < lane_end; ++lane) cur.push_back(raygen_lane(C, seed, spp, log_spp, (uint32_t) lane, ls));
    int status = 0;
    for (uint32_t b = 0; b < max_bounces && !cur.empty(); ++b) {
        double *o = out + 32 * b;
        struct Sh { Vec3 o, d; float maxt; uint32_t pixel; };
        std::vector<Sh> shadow; next.clear();
        auto account = [&](std::vector<std::vector<uint32_t>> &evs, double *q) {
            size_t n = evs.size();
            if (refill <= 0) {
                for (size_t w = 0; w < n; w += 64) {
                    size_t steps = 0;
                    for (size_t i = w; i < std::min(n, w + 64); ++i) steps = std::max(steps, evs[i].size());
                    q[5] += (double) steps; q[9] += 1;
                    for (size_t k = 0; k < steps; ++k) {
                        uint32_t anyn = 0, anyi = 0, mt = 0;
                        for (size_t i = w; i < std::min(n, w + 64); ++i) if (k < evs[i].size()) { uint32_t v = evs[i][k]; anyn |= v & 1u; anyi |= (v >> 1) & 1u; mt = std::max(mt, v >> 8); }
                        q[6] += anyn; q[7] += mt; q[8] += anyi;
                    }
                }
            } else {
                /* persistent waves: the bounce's rays are dealt to W concurrent waves in batches of 128 (like the kernel) */
                const size_t W = std::max<size_t>(1, std::min<size_t>(n / 1024, 4096));
                std::vector<size_t> cursor_of(W);
                size_t next_batch = 0;
                for (size_t w = 0; w < W; ++w) {
                    size_t slot_ray[64], slot_pos[64]; bool busy[64];
                    for (int l = 0; l < 64; ++l) busy[l] = false;
                    size_t pool = 0, pool_end = 0; bool exhausted = false;
                    q[9] += 1;
                    for (;;) {
                        int idle = 0; for (int l = 0; l < 64; ++l) idle += !busy[l];
                        if (idle >= refill) {
                            for (int l = 0; l < 64; ++l) {
                                if (busy[l]) continue;
                                if (pool == pool_end && !exhausted) {
                                    /* round-robin batches across the waves: wave w takes batches w, w+W, ... */
                                    size_t bidx = w + W * cursor_of[w]++;
                                    if (bidx * 128 >= n) exhausted = true; else { pool = bidx * 128; pool_end = std::min(n, pool + 128); }
                                }
                                if (pool < pool_end) { busy[l] = true; slot_ray[l] = pool++; slot_pos[l] = 0; }
                            }
                            idle = 0; for (int l = 0; l < 64; ++l) idle += !busy[l];
                            if (idle == 64) break;
                        }
                        uint32_t anyn = 0, anyi = 0, mt = 0;
                        /* what-if: lanes whose next step enters an instance wait until `defer` of them are pending (or nothing else can run) */
                        static const int defer = getenv("HH_DEFER_INST") ? atoi(getenv("HH_DEFER_INST")) : 0;
                        bool run_inst = true;
                        if (defer > 0) {
                            int pend = 0, other = 0;
                            for (int l = 0; l < 64; ++l) if (busy[l]) { uint32_t v = evs[slot_ray[l]][slot_pos[l]]; if ((v >> 1) & 1u) ++pend; else ++other; }
                            run_inst = pend >= defer || other == 0;
                        }
                        /* what-if: lanes whose next step tests a triangle wait until `defer_tri` of them do (or nothing else can run) */
                        static const int defer_tri = getenv("HH_DEFER_TRI") ? atoi(getenv("HH_DEFER_TRI")) : 0;
                        bool run_tri = true;
                        if (defer_tri > 0) {
                            int pend = 0, other = 0;
                            for (int l = 0; l < 64; ++l) if (busy[l]) { uint32_t v = evs[slot_ray[l]][slot_pos[l]]; if (v >> 8) ++pend; else ++other; }
                            run_tri = pend >= defer_tri || other == 0;
                        }
                        for (int l = 0; l < 64; ++l) if (busy[l]) {
                            uint32_t v = evs[slot_ray[l]][slot_pos[l]];
                            if (!run_inst && ((v >> 1) & 1u)) continue;
                            if (!run_tri && (v >> 8)) continue;
                            slot_pos[l]++; anyn |= v & 1u; anyi |= (v >> 1) & 1u; mt = std::max(mt, v >> 8);
                            if (slot_pos[l] == evs[slot_ray[l]].size()) busy[l] = false;
                        }
                        q[5] += 1; q[6] += anyn; q[7] += mt; q[8] += anyi;
                    }
                }
                (void) next_batch;
            }
            for (auto &v : evs) { q[0] += 1; q[1] += (double) v.size(); for (uint32_t x : v) { q[2] += x & 1u; q[4] += (x >> 1) & 1u; q[3] += x >> 8; } }
            /* what-if (HH_REGROUP=R, tools/regroup_model.py): a block keeps R rays' states in LDS and, every round, deals them to its waves BY WHAT THEY DO NEXT -- each wave
             * issue runs ONE block (node visit / triangle test / instance entry) for up to 64 rays that all need it.  Counted: issues per block type (a ray's step may hold a node
             * visit AND a leaf item: two passes of the round) and rounds (each costs every wave of the block a sort + barrier).  Rays refill when a fifth of the block is idle. */
            static const int regroup = getenv("HH_REGROUP") ? atoi(getenv("HH_REGROUP")) : 0;
            if (regroup > 0 && n > 0) {
                double *g = g_regroup_stat[q == o ? 0 : 1];
                const size_t R = (size_t) regroup, W = std::max<size_t>(1, std::min<size_t>(n / (4 * R), 1024));
                for (size_t w = 0; w < W; ++w) {
                    std::vector<size_t> ray(R), pos(R); std::vector<char> busy(R, 0);
                    size_t cursor = 0, pool = 0, pool_end = 0; bool exhausted = false;
                    for (;;) {
                        size_t idle = 0; for (size_t l = 0; l < R; ++l) idle += !busy[l];
                        if (idle >= std::max<size_t>(1, R / 5)) {
                            for (size_t l = 0; l < R; ++l) {
                                if (busy[l]) continue;
                                if (pool == pool_end && !exhausted) { const size_t bidx = w + W * cursor++; if (bidx * R >= n) exhausted = true; else { pool = bidx * R; pool_end = std::min(n, pool + R); } }
                                if (pool < pool_end) { busy[l] = 1; ray[l] = pool++; pos[l] = 0; }
                            }
                            idle = 0; for (size_t l = 0; l < R; ++l) idle += !busy[l];
                            if (idle == R) break;
                        }
                        size_t cn = 0, ct = 0, ci = 0;
                        for (size_t l = 0; l < R; ++l) if (busy[l]) {
                            const uint32_t v = evs[ray[l]][pos[l]++];
                            cn += v & 1u; ci += (v >> 1) & 1u; ct += (v >> 8) ? 1u : 0u;       /* POLICY 0 steps hold at most one leaf item */
                            if (pos[l] == evs[ray[l]].size()) busy[l] = 0;
                        }
                        g[0] += (double) ((cn + 63) / 64); g[1] += (double) ((ct + 63) / 64); g[2] += (double) ((ci + 63) / 64); g[3] += 1.0; g[5] += (double) (cn + ct + ci);
                    }
                }
                g[4] += (double) n;
            }
        };
        /* what-if (HH_SORT=1 origin Morton, 2 = direction octant + origin Morton): the wavefront re-ordered before each trace launch */
        static const int sort_mode = getenv("HH_SORT") ? atoi(getenv("HH_SORT")) : 0;
        auto sort_key = [&](const Vec3 &o, const Vec3 &d, const float lo[3], const float hi[3]) {
            uint32_t q[3]; const float oo[3] = { o.x, o.y, o.z };
            for (int a = 0; a < 3; ++a) { float t = (oo[a] - lo[a]) / std::max(hi[a] - lo[a], 1e-20f); q[a] = (uint32_t) std::min(1023.f, std::max(0.f, t * 1024.f)); }
            uint64_t m = 0; for (int bit = 9; bit >= 0; --bit) for (int a = 0; a < 3; ++a) m = (m << 1) | ((q[a] >> bit) & 1u);
            uint64_t oct = (d.x < 0 ? 1u : 0u) | (d.y < 0 ? 2u : 0u) | (d.z < 0 ? 4u : 0u);
            static const int bits = getenv("HH_SORT_BITS") ? atoi(getenv("HH_SORT_BITS")) : 30;
            m >>= (30 - bits);
            return sort_mode == 2 ? (oct << 30) | m : sort_mode == 3 ? (m << 3) | oct : m;
        };
        if (sort_mode && b > 0) {
            float lo[3] = { 1e30f, 1e30f, 1e30f }, hi[3] = { -1e30f, -1e30f, -1e30f };
            for (auto &c : cur) { const float oo[3] = { c.o.x, c.o.y, c.o.z }; for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], oo[a]); hi[a] = std::max(hi[a], oo[a]); } }
            std::vector<std::pair<uint64_t, uint32_t>> keys(cur.size());
            for (size_t i = 0; i < cur.size(); ++i) keys[i] = { sort_key(cur[i].o, cur[i].d, lo, hi), (uint32_t) i };
            std::stable_sort(keys.begin(), keys.end());
            std::vector<PathState> tmp(cur.size()); for (size_t i = 0; i < cur.size(); ++i) tmp[i] = cur[keys[i].second];
            cur.swap(tmp);
        }
        std::vector<std::vector<uint32_t>> evs(cur.size());
        std::vector<Hit> hits(cur.size());
        for (size_t i = 0; i < cur.size(); ++i) {
            HostStack stack; EvProbe pr{ &evs[i] };
            accel_trace<false>(S.accel, cur[i].o, cur[i].d, cur[i].maxt, hits[i], stack, status, pr);
            if (policy >= 0) {
                Hit h2; bool f2; evs[i].clear();
                if (policy == 0) run_traversal<false, Traversal<0>>(S.accel, cur[i].o, cur[i].d, cur[i].maxt, h2, f2, evs[i], status);
                else if (policy == 1) run_traversal<false, Traversal<1>>(S.accel, cur[i].o, cur[i].d, cur[i].maxt, h2, f2, evs[i], status);
                else             run_traversal<false, Traversal<2>>(S.accel, cur[i].o, cur[i].d, cur[i].maxt, h2, f2, evs[i], status);
                if (memcmp(&h2, &hits[i], sizeof(Hit)) != 0 || f2 != (hits[i].t != HAR_INF)) o[10] += 1;
            }
        }
        account(evs, o);
        for (size_t i = 0; i < cur.size(); ++i) {
            ShadeResult R; shade_lane<MODE_PATH>(S, P, cur[i], hits[i], R);
            if (R.item && R.item_ray) shadow.push_back(Sh{ R.sh_o, R.sh_d, R.sh_maxt, cur[i].lane / spp });
            if (R.alive) next.push_back(R.next);
        }
        if (sort_mode && !shadow.empty()) {
            float lo[3] = { 1e30f, 1e30f, 1e30f }, hi[3] = { -1e30f, -1e30f, -1e30f };
            for (auto &c : shadow) { const float oo[3] = { c.o.x, c.o.y, c.o.z }; for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], oo[a]); hi[a] = std::max(hi[a], oo[a]); } }
            std::vector<std::pair<uint64_t, uint32_t>> keys(shadow.size());
            for (size_t i = 0; i < shadow.size(); ++i) keys[i] = { sort_key(shadow[i].o, shadow[i].d, lo, hi), (uint32_t) i };
            std::stable_sort(keys.begin(), keys.end());
            std::vector<Sh> tmp(shadow.size()); for (size_t i = 0; i < shadow.size(); ++i) tmp[i] = shadow[keys[i].second];
            shadow.swap(tmp);
        }
        std::vector<std::vector<uint32_t>> sev(shadow.size());
        static const int shadow_order = getenv("HH_SHADOW_ORDER") ? atoi(getenv("HH_SHADOW_ORDER")) : 0;
        g_host_child_order = shadow_order;
        for (size_t i = 0; i < shadow.size(); ++i) {
            HostStack stack; EvProbe pr{ &sev[i] }; Hit hh;
            bool f1 = accel_trace<true>(S.accel, shadow[i].o, shadow[i].d, shadow[i].maxt, hh, stack, status, pr);
            if (policy >= 0) {
                Hit h2; bool f2; sev[i].clear();
                if (policy == 0) run_traversal<true, Traversal<0>>(S.accel, shadow[i].o, shadow[i].d, shadow[i].maxt, h2, f2, sev[i], status);
                else if (policy == 1) run_traversal<true, Traversal<1>>(S.accel, shadow[i].o, shadow[i].d, shadow[i].maxt, h2, f2, sev[i], status);
                else             run_traversal<true, Traversal<2>>(S.accel, shadow[i].o, shadow[i].d, shadow[i].maxt, h2, f2, sev[i], status);
                if (f1 != f2) o[26] += 1;
            }
            /* occluded vs unoccluded shadow rays: count and node visits of the occluded ones (slots 27 / 28) */
            if (f1) { o[27] += 1; for (uint32_t x : sev[i]) o[28] += x & 1u; }
            /* occluder-cache what-if: is the occluder of this ray the one that occluded the pixel's previous occluded shadow ray of this bounce?
             * slot 29: same instance (upper bound of a cache's hit rate), slot 30: same triangle (lower bound) */
            if (f1) {
                static std::vector<Hit> last; static uint32_t last_bounce = 0xffffffffu;
                if (last_bounce != b) { last.assign((size_t) C.samp_w * C.samp_h, Hit{ HAR_INF, 0.f, 0.f, 0u, 0u, 0u }); last_bounce = b; }
                Hit occ; { HostStack s3; int st3 = 0; accel_trace<false>(S.accel, shadow[i].o, shadow[i].d, shadow[i].maxt, occ, s3, st3); }      /* any-hit reports no identity: take the nearest occluder */
                hh = occ;
                Hit &L = last[shadow[i].pixel];
                if (L.t != HAR_INF) { if (L.inst == hh.inst && L.shape == hh.shape) { o[29] += 1; if (L.prim == hh.prim) o[30] += 1; } }
                L = hh; L.t = 1.f;
            }
        }
        g_host_child_order = 0;
        account(sev, o + 16);
        cur.swap(next);
    }
    return status;
}

} // extern "C"

/* --- wave-shared ("packet") descent model (tools/packet_stats.py): the 64 rays of a wave walk the BVH TOGETHER -- one conservative interval test per child box for the
 * whole packet, exact per-lane Moeller-Trumbore tests at the leaves (so the hits stay those of the brute-force kernel: the BVH only prunes), one shared stack.  This is a
 * MODEL of the candidate kernel for coherent launches (camera rays / first shadow rays at >= 64 spp: a wave is one pixel); it counts the blocks such a wave would issue. */
namespace {
struct PkBounds { float o_lo[3], o_hi[3], id_lo[3], id_hi[3]; bool mixed[3]; uint32_t octinv; };
struct PkCount { double packets = 0, rays = 0, nodes = 0, tris = 0, insts = 0, mismatches = 0, lane_nodes = 0, lane_tris = 0, lane_insts = 0, coherent = 0; };
static PkBounds pk_bounds(const RaySetup *R, const bool *act, int n) {
    PkBounds B; for (int a = 0; a < 3; ++a) { B.o_lo[a] = HAR_INF; B.o_hi[a] = -HAR_INF; B.id_lo[a] = HAR_INF; B.id_hi[a] = -HAR_INF; }
    for (int l = 0; l < n; ++l) if (act[l]) {
        const float o[3] = { R[l].o.x, R[l].o.y, R[l].o.z }, id[3] = { R[l].idir.x, R[l].idir.y, R[l].idir.z };
        for (int a = 0; a < 3; ++a) { B.o_lo[a] = fminf(B.o_lo[a], o[a]); B.o_hi[a] = fmaxf(B.o_hi[a], o[a]); B.id_lo[a] = fminf(B.id_lo[a], id[a]); B.id_hi[a] = fmaxf(B.id_hi[a], id[a]); }
    }
    B.octinv = 0;
    for (int a = 0; a < 3; ++a) { B.mixed[a] = !(B.id_lo[a] > 0.f || B.id_hi[a] < 0.f); if (!(B.id_hi[a] < 0.f)) B.octinv |= 4u >> a; }      /* octinv bit set = non-negative direction (ray_setup) */
    return B;
}
/* conservative version of node_visit: same outputs (child group / triangle group), a child is kept unless NO ray of the packet can meet its box */
static void pk_node_visit(const Accel &A, const PkBounds &B, float tmax_wave, uint32_t index, uint32_t &ng_x, uint32_t &ng_y, uint32_t &tg_x, uint32_t &tg_y) {
    const Node8 &N = A.nodes[index];
    const float p[3] = { N.px, N.py, N.pz }, sc[3] = { as_f32((uint32_t) N.ex << 23), as_f32((uint32_t) N.ey << 23), as_f32((uint32_t) N.ez << 23) };
    const uint8_t *qlo[3] = { N.qlox, N.qloy, N.qloz }, *qhi[3] = { N.qhix, N.qhiy, N.qhiz };
    uint32_t hitmask = 0;
    for (int i = 0; i < 8; ++i) {
        const bool inner = (N.imask >> i) & 1u, leaf = (N.lmask >> i) & 1u;
        if (!inner && !leaf) continue;
        const uint32_t bits = 1u, idx = inner ? 24u + ((uint32_t) i ^ B.octinv) : (uint32_t) i;      /* inner child: position by the octant (node_visit); leaf: its slot */
        float lb = 0.f, ub = tmax_wave;
        for (int a = 0; a < 3; ++a) {
            if (B.mixed[a]) continue;                                        /* directions of both signs on this axis: no constraint from its slab */
            const float lo = fma_((float) qlo[a][i], sc[a], p[a]), hi = fma_((float) qhi[a][i], sc[a], p[a]);
            float tn, tf;
            if (B.id_lo[a] > 0.f) {                                          /* t = (plane - o) * idir, idir in [id_lo, id_hi] > 0 */
                const float dn = lo - B.o_hi[a], df = hi - B.o_lo[a];
                tn = dn * (dn >= 0.f ? B.id_lo[a] : B.id_hi[a]); tf = df * (df >= 0.f ? B.id_hi[a] : B.id_lo[a]);
            } else {                                                         /* idir in [id_lo, id_hi] < 0: near plane = hi */
                const float dn = hi - B.o_lo[a], df = lo - B.o_hi[a];      /* dn <= ... : (hi - o) largest at o_lo; t = dn * idir, idir negative */
                tn = dn * (dn >= 0.f ? B.id_lo[a] : B.id_hi[a]);            /* lower bound: most negative product */
                tf = df * (df >= 0.f ? B.id_hi[a] : B.id_lo[a]);
                /* with negative idir: t_near = (hi - o) * idir is smallest for the LARGEST (hi - o) times the most negative idir when hi - o >= 0 */
            }
            lb = fmaxf(lb, tn); ub = fminf(ub, tf);
        }
        if (lb <= ub * 1.000002f) hitmask |= bits << idx;
    }
    ng_x = N.child_base; tg_x = N.tri_base;
    ng_y = (hitmask & 0xff000000u) | N.imask; tg_y = (hitmask & 0xffu) ? ((hitmask & 0xffu) | ((uint32_t) N.lmask << 8)) : 0u;
}
/* one BLAS (or the TLAS) walked by the packet; R / act / tmax / hit are per lane */
template <bool AnyHit>
static void pk_walk(const Accel &A, uint32_t root, bool tlas, const Vec3 *o_w, const Vec3 *d_w, RaySetup *R, bool *act, float *tmax, Hit *hit, bool *found, int n, uint32_t cur_inst, PkCount &C) {
    PkBounds B = pk_bounds(R, act, n);
    uint32_t sx[64], sy[64]; int sp = 0;
    uint32_t ng_x = root, ng_y = 0x80000000u, tg_x = 0, tg_y = 0;
    for (;;) {
        bool any = false; for (int l = 0; l < n; ++l) any = any || act[l];
        if (!any) return;
        if (ng_y > 0x00ffffffu) {
            uint32_t px = ng_x, py = ng_y;
            const uint32_t child = ng_next_child(px, py, B.octinv);
            if (py > 0x00ffffffu) { sx[sp] = px; sy[sp] = py; ++sp; }
            float tw = 0.f; for (int l = 0; l < n; ++l) if (act[l]) tw = fmaxf(tw, tmax[l]);
            pk_node_visit(A, B, tw, child, ng_x, ng_y, tg_x, tg_y); C.nodes += 1;
        } else { tg_x = ng_x; tg_y = ng_y; ng_x = 0; ng_y = 0; }
        while (tg_y != 0u) {
            const uint32_t idx = tg_next_leaf(tg_x, tg_y);
            if (tlas) {
                const InstRec &I = A.insts[idx];
                RaySetup Ro[64];
                for (int l = 0; l < n; ++l) if (act[l]) Ro[l] = I.identity ? R[l] : ray_setup(xf_point(I.to_object, o_w[l]), xf_vector(I.to_object, d_w[l]));
                C.insts += 1;
                pk_walk<AnyHit>(A, I.blas_root, false, o_w, d_w, Ro, act, tmax, hit, found, n, I.inst_index, C);
            } else {
                C.tris += 1;
                for (int l = 0; l < n; ++l) if (act[l]) {
                    if (tri_visit<AnyHit>(A, R[l], tmax[l], idx, cur_inst, hit[l])) { found[l] = true; act[l] = false; }
                }
            }
        }
        if (ng_y <= 0x00ffffffu) {
            if (sp == 0) return;
            --sp; ng_x = sx[sp]; ng_y = sy[sp];
        }
    }
}
template <bool AnyHit>
static void pk_trace(const Accel &A, const Vec3 *o, const Vec3 *d, const float *maxt, int n, Hit *hit, bool *found, PkCount &C) {
    RaySetup R[64]; bool act[64]; float tmax[64];
    for (int l = 0; l < n; ++l) { R[l] = ray_setup(o[l], d[l]); act[l] = true; tmax[l] = maxt[l]; found[l] = false; hit[l].t = HAR_INF; hit[l].u = 0.f; hit[l].v = 0.f; hit[l].prim = 0; hit[l].shape = 0; hit[l].inst = 0xffffffffu; }
    C.packets += 1; C.rays += n;
    const bool two_level = A.has_tlas != 0;
    const bool top_last = (A.top_last & (AnyHit ? 1u : 2u)) != 0u;
    if (!two_level) { pk_walk<AnyHit>(A, A.root, false, o, d, R, act, tmax, hit, found, n, 0xffffffffu, C); }
    else {
        for (int phase = 0; phase < 2; ++phase) {
            const bool top = (phase == 0) != top_last;
            if (top) { if (A.top_root != HAR_NO_NODE) pk_walk<AnyHit>(A, A.top_root, false, o, d, R, act, tmax, hit, found, n, 0xffffffffu, C); }
            else pk_walk<AnyHit>(A, A.root, true, o, d, R, act, tmax, hit, found, n, 0xffffffffu, C);
        }
    }
    if (!AnyHit) for (int l = 0; l < n; ++l) found[l] = hit[l].t != HAR_INF;
}
struct CountProbe : NoProbe { double *nodes, *tris, *insts; CountProbe(double *a, double *b, double *c) : nodes(a), tris(b), insts(c) {} void iter() {} void node() { *nodes += 1; } void tri() { *tris += 1; } void inst() { *insts += 1; } };
}

extern "C" {
/* out[bounce][2][12]: per bounce, closest (0) and shadow (1) launches: packets, rays, packet node visits, packet leaf (triangle) blocks, packet instance entries, mismatches
 * against the per-ray reference loop, and the per-ray loop's node visits / triangle tests / instance entries summed over the same rays; slots 9.. : histogram helper (unused) */
int hh_packet_stats(void *h, const HarSensor *sensor, uint32_t seed, uint32_t spp, int32_t max_depth, int32_t rr_depth, uint64_t lane_begin, uint64_t lane_end,
                    uint32_t max_bounces, double *out, double *per_packet /* nullable: [bounce 0 closest | bounce 0 shadow] x (nodes, tris, insts, lane nodes) per packet, 4 doubles each, cap in out */,
                    uint64_t per_packet_cap) {
    HScene *H = (HScene *) h; const DScene &S = H->ds;
    DSensor C; std::string e; if (!lower_sensor(*sensor, C, e)) return -1;
    uint32_t log_spp = 0xffffffffu; for (uint32_t k = 0; k < 32; ++k) if ((1u << k) == spp) log_spp = k;
    ShadeParams P{ seed, (uint32_t) max_depth, (uint32_t) rr_depth };
    std::vector<PathState> cur, next; LaneSample ls;
    for (uint64_t lane = lane_begin; lane < lane_end; ++lane) cur.push_back(raygen_lane(C, seed, spp, log_spp, (uint32_t) lane, ls));
    int status = 0; uint64_t pp = 0;
    for (uint32_t b = 0; b < max_bounces && !cur.empty(); ++b) {
        struct Sh { Vec3 o, d; float maxt; };
        std::vector<Sh> shadow; next.clear();
        std::vector<Hit> hits(cur.size());
        PkCount cc;
        for (size_t w = 0; w < cur.size(); w += 64) {
            const int n = (int) std::min<size_t>(64, cur.size() - w);
            Vec3 o[64], d[64]; float mt[64]; Hit ph[64]; bool pf[64];
            for (int l = 0; l < n; ++l) { o[l] = cur[w + l].o; d[l] = cur[w + l].d; mt[l] = cur[w + l].maxt; }
            const double n0 = cc.nodes, t0 = cc.tris, i0 = cc.insts, ln0 = cc.lane_nodes;
            pk_trace<false>(S.accel, o, d, mt, n, ph, pf, cc);
            for (int l = 0; l < n; ++l) {
                HostStack stack; CountProbe pr{ &cc.lane_nodes, &cc.lane_tris, &cc.lane_insts };
                accel_trace<false>(S.accel, o[l], d[l], mt[l], hits[w + l], stack, status, pr);
                if (memcmp(&hits[w + l], &ph[l], sizeof(Hit)) != 0) cc.mismatches += 1;
            }
            if (per_packet && b == 0 && pp < per_packet_cap) { double *q = per_packet + 8 * pp; q[0] = cc.nodes - n0; q[1] = cc.tris - t0; q[2] = cc.insts - i0; q[3] = cc.lane_nodes - ln0; }
            if (b == 0) ++pp;
        }
        double *q = out + 24 * b;
        q[0] = cc.packets; q[1] = cc.rays; q[2] = cc.nodes; q[3] = cc.tris; q[4] = cc.insts; q[5] = cc.mismatches; q[6] = cc.lane_nodes; q[7] = cc.lane_tris; q[8] = cc.lane_insts;
        for (size_t i = 0; i < cur.size(); ++i) {
            ShadeResult R; shade_lane<MODE_PATH>(S, P, cur[i], hits[i], R);
            if (R.item && R.item_ray) shadow.push_back(Sh{ R.sh_o, R.sh_d, R.sh_maxt });
            if (R.alive) next.push_back(R.next);
        }
        PkCount sc; uint64_t sp2 = 0;
        for (size_t w = 0; w < shadow.size(); w += 64) {
            const int n = (int) std::min<size_t>(64, shadow.size() - w);
            Vec3 o[64], d[64]; float mt[64]; Hit ph[64]; bool pf[64];
            for (int l = 0; l < n; ++l) { o[l] = shadow[w + l].o; d[l] = shadow[w + l].d; mt[l] = shadow[w + l].maxt; }
            const double n0 = sc.nodes, t0 = sc.tris, i0 = sc.insts, ln0 = sc.lane_nodes;
            pk_trace<true>(S.accel, o, d, mt, n, ph, pf, sc);
            for (int l = 0; l < n; ++l) {
                HostStack stack; CountProbe pr{ &sc.lane_nodes, &sc.lane_tris, &sc.lane_insts }; Hit hh;
                const bool f = accel_trace<true>(S.accel, o[l], d[l], mt[l], hh, stack, status, pr);
                if (f != pf[l]) sc.mismatches += 1;
            }
            if (per_packet && b == 0 && sp2 < per_packet_cap) { double *qq = per_packet + 8 * sp2 + 4; qq[0] = sc.nodes - n0; qq[1] = sc.tris - t0; qq[2] = sc.insts - i0; qq[3] = sc.lane_nodes - ln0; }
            if (b == 0) ++sp2;
        }
        q += 12;
        q[0] = sc.packets; q[1] = sc.rays; q[2] = sc.nodes; q[3] = sc.tris; q[4] = sc.insts; q[5] = sc.mismatches; q[6] = sc.lane_nodes; q[7] = sc.lane_tris; q[8] = sc.lane_insts;
        cur.swap(next);
    }
    return status;
}
}

extern "C" {
/* environment map: the product's host lowering (build_envmap) + the HAR_HD lookup / sampling code */
void hh_envmap_storage(void *h, uint32_t *info /* w, h, n_levels, warp size */, uint32_t *table /* [n_levels][2]: width, offset */, float *warp) {
    HScene *H = (HScene *) h; const DEnvmap &E = H->hs.envmap;
    info[0] = E.w; info[1] = E.h; info[2] = E.n_levels; info[3] = (uint32_t) H->hs.env_warp.size();
    if (table) for (uint32_t l = 0; l < E.n_levels; ++l) { table[2 * l] = E.lvl_width[l]; table[2 * l + 1] = E.lvl_offset[l]; }
    if (warp) std::copy(H->hs.env_warp.begin(), H->hs.env_warp.end(), warp);
}
void hh_envmap_eval(void *h, uint32_t n, const float *d, float *rgb, float *pdf) {
    const DEnvmap &E = *((HScene *) h)->ds.envmap;
    for (uint32_t i = 0; i < n; ++i) {
        Vec3 dd(d[3 * i], d[3 * i + 1], d[3 * i + 2]); Vec3 v = envmap_eval(E, dd);
        rgb[3 * i] = v.x; rgb[3 * i + 1] = v.y; rgb[3 * i + 2] = v.z; pdf[i] = envmap_pdf_direction(E, dd);
    }
}
void hh_envmap_sample_direction(void *h, uint32_t n, const float *p, const float *s, float *d, float *dist, float *pdf, float *weight) {
    const DEnvmap &E = *((HScene *) h)->ds.envmap;
    for (uint32_t i = 0; i < n; ++i) {
        DirSample ds; Vec3 w; envmap_sample_direction(E, Vec3(p[3 * i], p[3 * i + 1], p[3 * i + 2]), s[2 * i], s[2 * i + 1], ds, w);
        d[3 * i] = ds.d.x; d[3 * i + 1] = ds.d.y; d[3 * i + 2] = ds.d.z; dist[i] = ds.dist; pdf[i] = ds.pdf;
        weight[3 * i] = w.x; weight[3 * i + 1] = w.y; weight[3 * i + 2] = w.z;
    }
}
} // extern "C"
