/*
 * har_scene_host.cpp -- Scene lowering for the hip_ad_rgb path (host C++).
 * Follows SceneIRBuilder::build (src/render/scene_ir.cpp:12-92): top-level
 * triangle geometry -> one BLAS, each ShapeGroup -> one BLAS, TLAS with an
 * identity entry for the top-level BLAS and one entry per Instance.
 */
#include "har_scene_host.h"
#include "har_refit.h"
#include "har_vertex_update.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <cstdlib>

namespace har {

namespace {

/* depth of every node of a BLAS (children sit behind their parents, so one forward sweep does it) -> its nodes sorted deepest first */
void blas_refit_order(HostScene &hs, BlasInfo &B) {
    B.order_first = (uint32_t) hs.refit_order.size(); B.level_begin.clear();
    if (B.empty || B.node_count == 0) return;
    std::vector<uint32_t> depth(B.node_count, 0u); uint32_t deepest = 0;
    for (uint32_t i = 0; i < B.node_count; ++i) {
        const Node8 &n = hs.nodes[B.root + i];
        uint32_t k = 0;
        for (int s = 0; s < 8; ++s) if (n.imask & (1u << s)) { depth[n.child_base + k - B.root] = depth[i] + 1u; deepest = std::max(deepest, depth[i] + 1u); ++k; }
    }
    std::vector<uint32_t> count(deepest + 2, 0u);
    for (uint32_t d : depth) count[deepest - d + 1]++;
    for (uint32_t l = 0; l <= deepest; ++l) count[l + 1] += count[l];
    B.level_begin.assign(count.begin(), count.end());
    hs.refit_order.resize((size_t) B.order_first + B.node_count);
    std::vector<uint32_t> cursor(count.begin(), count.end() - 1);
    for (uint32_t i = 0; i < B.node_count; ++i) hs.refit_order[B.order_first + cursor[deepest - depth[i]]++] = B.root + i;
}

/* the same for the TLAS (its nodes are the tail of hs.nodes, root first) */
void tlas_refit_order(HostScene &hs) {
    hs.tlas_order.clear(); hs.tlas_levels.clear(); ++hs.tlas_serial;
    const uint32_t first = hs.root, count = (uint32_t) hs.nodes.size() - first;
    if (!hs.has_tlas || count == 0) return;
    std::vector<uint32_t> depth(count, 0u); uint32_t deepest = 0;
    for (uint32_t i = 0; i < count; ++i) {
        const Node8 &n = hs.nodes[first + i];
        uint32_t k = 0;
        for (int s = 0; s < 8; ++s) if (n.imask & (1u << s)) { depth[n.child_base + k - first] = depth[i] + 1u; deepest = std::max(deepest, depth[i] + 1u); ++k; }
    }
    std::vector<uint32_t> cnt(deepest + 2, 0u);
    for (uint32_t d : depth) cnt[deepest - d + 1]++;
    for (uint32_t l = 0; l <= deepest; ++l) cnt[l + 1] += cnt[l];
    hs.tlas_levels.assign(cnt.begin(), cnt.end());
    hs.tlas_order.resize(count);
    std::vector<uint32_t> cursor(cnt.begin(), cnt.end() - 1);
    for (uint32_t i = 0; i < count; ++i) hs.tlas_order[cursor[deepest - depth[i]]++] = first + i;
}

BlasInfo build_blas(HostScene &hs, uint32_t first_mesh, uint32_t mesh_count, bool optimal_collapse = false) {
    std::vector<PrimBox> prims; std::vector<TriRec> recs;
    for (uint32_t s = first_mesh; s < first_mesh + mesh_count; ++s) {
        const DMesh &m = hs.meshes[s];
        const float *vertex_ptr = hs.verts.data() + 8 * (size_t) m.voff; const uint32_t *index_ptr = hs.faces.data() + 4 * (size_t) m.foff;
        for (uint32_t f = 0; f < m.face_count; ++f) {
            float p[3][3];
            for (int k = 0; k < 3; ++k) { const float *v = vertex_ptr + 8 * (size_t) index_ptr[4 * (size_t) f + k]; p[k][0] = v[0]; p[k][1] = v[1]; p[k][2] = v[2]; }
            TriRec t;
            t.p0x = p[0][0]; t.p0y = p[0][1]; t.p0z = p[0][2];
            t.e1x = p[1][0] - p[0][0]; t.e1y = p[1][1] - p[0][1]; t.e1z = p[1][2] - p[0][2];
            t.e2x = p[2][0] - p[0][0]; t.e2y = p[2][1] - p[0][1]; t.e2z = p[2][2] - p[0][2];
            t.prim = f; t.shape = s; t.pad = 0;
            PrimBox b;
            for (int a = 0; a < 3; ++a) { b.lo[a] = std::min(p[0][a], std::min(p[1][a], p[2][a])); b.hi[a] = std::max(p[0][a], std::max(p[1][a], p[2][a])); }
            pad_prim_box(b);
            prims.push_back(b); recs.push_back(t);
        }
    }
    BlasInfo info{};
    info.first_tri = (uint32_t) hs.tris.size(); info.tri_count = (uint32_t) recs.size(); info.empty = recs.empty();
    std::vector<uint32_t> order;
    const uint32_t blas_leaf = 1u;           /* one primitive per leaf (Node8::lmask): 1 beat 2 and 3 in round 1 (fewer wasted triangle tests, esp. for any-hit rays) */
    static const uint32_t blas_dp_min = getenv("HAR_BVH_DP_MIN") ? (uint32_t) atol(getenv("HAR_BVH_DP_MIN")) : 128u;    /* measured: the 36-triangle Cornell box is 5 % faster with the greedy collapse */
    static const float tri_cost = getenv("HAR_BVH_CTRI") ? (float) atof(getenv("HAR_BVH_CTRI")) : 0.3f;     /* triangle test vs node visit (VALU instructions) */
    /* optimal_collapse: the top-level BLAS of a two-level scene, which EVERY ray walks -- the SAH-optimal collapse whatever its size (round 3, host model:
     * 2.55 -> 1.85 node visits per ray for the 12 wall triangles of the benchmark scene; the small-scene exception above is about stand-alone scenes) */
    const uint32_t before = (uint32_t) hs.nodes.size();
    info.root = build_bvh8(prims, hs.nodes, info.first_tri, order, &hs.stats, blas_leaf, tri_cost, optimal_collapse ? 0u : blas_dp_min);
    info.node_count = (uint32_t) hs.nodes.size() - before;
    for (uint32_t i : order) hs.tris.push_back(recs[i]);
    for (int a = 0; a < 3; ++a) { info.lo[a] = INFINITY; info.hi[a] = -INFINITY; }
    for (const PrimBox &b : prims) for (int a = 0; a < 3; ++a) { info.lo[a] = std::min(info.lo[a], b.lo[a]); info.hi[a] = std::max(info.hi[a], b.hi[a]); }
    blas_refit_order(hs, info);
    return info;
}

/* quad::gauss_legendre (include/mitsuba/core/quad.h:27-90): nodes / weights on [-1, 1] */
void gauss_legendre(int n, std::vector<float> &nodes, std::vector<float> &weights) {
    nodes.assign(n, 0.f); weights.assign(n, 0.f);
    auto legendre_pd = [](int l, double x, double &p, double &dp) {          /* math::legendre_pd */
        double l_cur = x, l_p_pred = 1, d_cur = 1, d_p_pred = 0, k_cur = 1;
        if (l == 0) { p = 1; dp = 0; return; }
        for (int k = 1; k < l; ++k) {
            double l_next = ((2 * k_cur + 1) * x * l_cur - k_cur * l_p_pred) / (k_cur + 1), d_next = d_p_pred + (2 * k_cur + 1) * l_cur;
            l_p_pred = l_cur; l_cur = l_next; d_p_pred = d_cur; d_cur = d_next; k_cur += 1;
        }
        p = l_cur; dp = d_cur;
    };
    int N = n - 1, m = (N + 1) / 2;
    for (int i = 0; i < m; ++i) {
        double x = -std::cos((double) (2 * i + 1) / (double) (2 * N + 2) * 3.14159265358979323846);
        for (int it = 0; it < 20; ++it) {
            double p, dp; legendre_pd(N + 1, x, p, dp);
            double step = p / dp; x -= step;
            if (std::fabs(step) <= 4 * std::fabs(x) * 1.1102230246251565e-16) break;
        }
        double p, dp; legendre_pd(N + 1, x, p, dp);
        weights[i] = weights[N - i] = (float) (2 / ((1 - x * x) * (dp * dp)));
        nodes[i] = (float) x; nodes[N - i] = (float) -x;
    }
    if ((N % 2) == 0) {
        double p, dp; legendre_pd(N + 1, 0.0, p, dp);
        weights[N / 2] = (float) (2 / (dp * dp)); nodes[N / 2] = 0.f;
    }
}

/* eval_transmittance / eval_reflectance (include/mitsuba/render/microfacet.h:465-567) for one direction */
float rough_transmittance(const Microfacet &distr, Vec3 wi, float eta, bool reflectance) {
    int res = eta > 1.f ? 32 : 128;
    std::vector<float> nodes, weights; gauss_legendre(res, nodes, weights);
    float result = 0.f;
    for (int jy = 0; jy < res; ++jy)                     /* dr::meshgrid(nodes, nodes): x varies fastest */
        for (int jx = 0; jx < res; ++jx) {
            float nx = fma_(nodes[jx], 0.5f, 0.5f), ny = fma_(nodes[jy], 0.5f, 0.5f);
            float pdf; Vec3 m = distr.sample(wi, nx, ny, pdf);
            float f, cos_theta_t, eta_it, eta_ti; fresnel_dielectric(dot3(wi, m), eta, f, cos_theta_t, eta_it, eta_ti);
            float smith;
            if (reflectance) {
                Vec3 wo = reflect_m(wi, m);
                smith = distr.smith_g1(wo, m) * f;
                if (wo.z <= 0.f || wi.z <= 0.f) smith = 0.f;
            } else {
                Vec3 wo = refract_m(wi, m, cos_theta_t, eta_ti);
                smith = distr.smith_g1(wo, m) * (1.f - f);
                if (wo.z * wi.z >= 0.f) smith = 0.f;
            }
            result += smith * (weights[jx] * weights[jy]) * 0.25f;
        }
    return result;
}

/* RoughPlastic::parameters_changed (src/bsdfs/roughplastic.cpp:204-242) */
void build_roughplastic_tables(HostScene &hs, uint32_t index) {
    DBsdf &b = hs.bsdfs[index];
    b.inv_eta_2 = 1.f / (b.eta * b.eta);
    Microfacet distr((b.flags & BF_GGX) != 0, b.alpha_u, b.alpha_u, true);
    if (b.table < 0) { b.table = (int32_t) hs.bsdf_tables.size(); hs.bsdf_tables.resize(hs.bsdf_tables.size() + HAR_ROUGH_TRANSMITTANCE_RES); }      /* a parameter update rewrites the record's table in place */
    double mean = 0;
    for (int i = 0; i < HAR_ROUGH_TRANSMITTANCE_RES; ++i) {
        float mu = std::max(1e-6f, (float) i / (float) (HAR_ROUGH_TRANSMITTANCE_RES - 1));
        Vec3 wi(std::sqrt(1.f - mu * mu), 0.f, mu);
        hs.bsdf_tables[(size_t) b.table + i] = rough_transmittance(distr, wi, b.eta, false);
        mean += (double) (rough_transmittance(distr, wi, 1.f / b.eta, true) * wi.z);
    }
    b.internal_reflectance = (float) (mean / HAR_ROUGH_TRANSMITTANCE_RES) * 2.f;
}

} // namespace

/* the part of an emitter's record that depends on nothing but its own description: every field of the area / constant / point records, SpotLight::update, the direction of
 * a directional light (lower_scene adds what hangs on the scene: texel tables, face tables, the bounding sphere) */
bool lower_plain_emitter(const HarEmitter &e, DEmitter &de, std::string &err) {
    de.type = e.type;
    std::memcpy(de.radiance, e.radiance, 12); de.inv_area = e.inv_area;
    std::memcpy(de.to_world, e.to_world, 48); std::memcpy(de.normal, e.normal, 12); de.mesh = e.mesh;
    if (e.type == 5) {            /* SpotLight::update (spot.cpp:300-312); the record keeps what sample_direction reads: the inverse's linear part, the position, the cone */
        const float deg = 0.017453292519943295f, cutoff_rad = e.normal[0] * deg, beam_rad = e.normal[1] * deg;
        if (!(cutoff_rad >= beam_rad) || !(cutoff_rad > 0.f)) { err = "spot: cutoff_angle must be positive and not smaller than beam_width"; return false; }
        float s_, cos_cutoff, cos_beam; sincos_(cutoff_rad, s_, cos_cutoff); sincos_(beam_rad, s_, cos_beam);
        for (int k = 0; k < 9; ++k) de.to_world[k] = e.to_local[k];
        de.to_world[9] = e.to_world[9]; de.to_world[10] = e.to_world[10]; de.to_world[11] = e.to_world[11];
        de.normal[0] = cutoff_rad; de.normal[1] = cos_cutoff; de.normal[2] = cos_beam; de.inv_area = 1.0f / (cutoff_rad - beam_rad);
    }
    if (e.type == 6) {            /* the record of a directional light keeps its direction of travel (third column of to_world) in [0..2]; [3..6] = the scene's bounding sphere (update_scene_bounds) */
        de.to_world[0] = e.to_world[6]; de.to_world[1] = e.to_world[7]; de.to_world[2] = e.to_world[8];
    }
    return true;
}
/* a DELTA emitter's record (point / spot / directional: HarEmitter types 4 - 6) re-lowered in place -- PointLight / SpotLight / DirectionalEmitter::parameters_changed after a
 * `position` / `to_world` / cone update; the sampling weight stays the distribution's business (har_scene_set_emitter_sampling_weights) */
bool scene_set_delta_emitter_host(HostScene &hs, uint32_t index, const HarEmitter &e, std::string &err) {
    if (index >= hs.emitters.size()) { err = "invalid emitter index"; return false; }
    if (e.type < 4 || e.type > 6 || hs.emitters[index].type != e.type) { err = "only the record of a point / spot / directional emitter can be replaced in place, by one of the same type"; return false; }
    for (int k = 0; k < 12; ++k) if (!std::isfinite(e.to_world[k]) || !std::isfinite(e.to_local[k])) { err = "emitter transform is not finite"; return false; }
    DEmitter de = hs.emitters[index];
    if (!lower_plain_emitter(e, de, err)) return false;
    hs.emitters[index] = de;
    update_scene_bounds(hs);          /* a directional light's record carries the scene's bounding sphere */
    return true;
}

void quad_gauss_legendre(int n, std::vector<float> &nodes, std::vector<float> &weights) { gauss_legendre(n, nodes, weights); }

/* the non-colour parameters of ONE record re-lowered in place (RoughConductor / RoughPlastic / SmoothPlastic / SmoothDielectric::parameters_changed): alpha, eta, the complex IOR and
 * the colour of slot 1; the checks of lower_scene for them */
bool scene_set_bsdf_params_host(HostScene &hs, uint32_t index, const HarBSDF &in, std::string &err) {
    if (index >= hs.bsdfs.size()) { err = "invalid bsdf index"; return false; }
    DBsdf &b = hs.bsdfs[index];
    for (float v : { in.alpha_u, in.alpha_v, in.eta, in.eta_c[0], in.eta_c[1], in.eta_c[2], in.k_c[0], in.k_c[1], in.k_c[2], in.reflectance2[0], in.reflectance2[1], in.reflectance2[2] })
        if (!std::isfinite(v)) { err = "BSDF parameter is not finite"; return false; }
    const bool needs_eta = b.type == BSDF_DIELECTRIC || b.type == BSDF_ROUGHPLASTIC || b.type == BSDF_PLASTIC;
    if (needs_eta && !(in.eta > 0.f)) { err = "The interior and exterior indices of refraction must be positive!"; return false; }
    if (b.type == BSDF_ROUGHPLASTIC && in.eta == 1.f) { err = "The interior and exterior indices of refraction must be positive and differ!"; return false; }
    if (b.type == BSDF_ROUGHPLASTIC && in.alpha_u != in.alpha_v) { err = "The 'roughplastic' plugin currently does not support anisotropic microfacet distributions!"; return false; }
    b.alpha_u = in.alpha_u; b.alpha_v = in.alpha_v; b.eta = in.eta;
    for (int k = 0; k < 3; ++k) { b.eta_c[k] = in.eta_c[k]; b.k_c[k] = in.k_c[k]; }
    b.r2 = in.reflectance2[0]; b.g2 = in.reflectance2[1]; b.b2 = in.reflectance2[2];
    if (b.type == BSDF_ROUGHPLASTIC) { build_roughplastic_tables(hs, index); update_roughplastic_sampling_weight(hs, index); }
    if (b.type == BSDF_PLASTIC) { b.inv_eta_2 = 1.f / (b.eta * b.eta); b.internal_reflectance = fresnel_diffuse_reflectance(1.f / b.eta); update_roughplastic_sampling_weight(hs, index); }
    return true;
}

void update_roughplastic_sampling_weight(HostScene &hs, uint32_t index) {
    DBsdf &b = hs.bsdfs[index];
    float d_mean;
    if (b.texture >= 0) {                                /* BitmapTexture::mean */
        const HostTexture &t = hs.textures[b.texture];
        double acc = 0; for (float v : t.data) acc += v;
        d_mean = (float) (acc / (double) t.data.size());
    } else d_mean = (b.r + b.g + b.b) / 3.f;             /* SRGBReflectanceSpectrum::mean, srgb.cpp:117-122 */
    float s_mean = (b.r2 + b.g2 + b.b2) / 3.f;
    b.spec_sampling_weight = s_mean / (d_mean + s_mean);
}

/* EnvironmentMapEmitter ctor (src/emitters/envmap.cpp:113-180) + rebuild_distribution (:476-529) + Hierarchical2D<Float, 0> ctor
 * (include/mitsuba/core/distr_2d.h:405-515): the radiance image is padded to >= 2 x 3 (Bitmap::pad_to), stored with one halo column per side,
 * and the (W + 1) x H grid of luminance * sin(theta) becomes a normalised bilinear interpolant with a MIP pyramid of 2 x 2-blocked sums. */
static bool build_envmap(HostScene &hs, const HarTexture &img, const HarEmitter &e, std::string &err) {
    if (!img.data || img.width == 0 || img.height == 0) { err = "envmap: empty bitmap"; return false; }
    DEnvmap &E = hs.envmap;
    const uint32_t W = std::max(img.width, 2u), H = std::max(img.height, 3u), sw = W + 2;
    E.w = W; E.h = H; E.scale = e.radiance[0];
    std::memcpy(E.to_world, e.to_world, 48); std::memcpy(E.to_local, e.to_local, 48);
    E.center[0] = E.center[1] = E.center[2] = 0.f; E.radius = 1.f;
    std::vector<float> &T = hs.env_tex; T.assign((size_t) H * sw * 3, 0.f);
    for (uint32_t y = 0; y < H; ++y) {
        float *row = T.data() + 3 * (size_t) y * sw;
        for (uint32_t x = 0; x < W; ++x) std::memcpy(row + 3 * (x + 1), img.data + 3 * ((size_t) std::min(y, img.height - 1) * img.width + std::min(x, img.width - 1)), 12);
        std::memcpy(row, row + 3 * W, 12); std::memcpy(row + 3 * (W + 1), row + 3, 12);          /* refresh_halo */
    }
    const uint32_t rx = W + 1, ry = H;
    std::vector<float> lum((size_t) rx * ry);
    for (uint32_t y = 0; y < ry; ++y)
        for (uint32_t x = 0; x < rx; ++x) { const float *c = T.data() + 3 * ((size_t) y * sw + x + 1); lum[(size_t) y * rx + x] = c[0] * 0.212671f + c[1] * 0.715160f + c[2] * 0.072169f; }
    float offset = 0.f;
    if (e.radiance[1] != 0.f) {         /* mis_compensation (Karlik et al. 2019) */
        float min_lum = INFINITY; double acc = 0.0;
        for (uint32_t y = 0; y < ry; ++y) for (uint32_t x = 0; x + 1 < rx; ++x) { float l = lum[(size_t) y * rx + x]; min_lum = std::min(min_lum, l); acc += (double) l; }
        offset = (float) (acc / (double) ((size_t) (rx - 1) * ry));
        if (offset - min_lum <= 0.01f * offset) offset = 0.f;
    }
    const float theta_scale = 1.f / (float) (ry - 1) * HAR_PI;
    for (uint32_t y = 0; y < ry; ++y) {
        const float sin_theta = std::sin((float) y * theta_scale);
        for (uint32_t x = 0; x < rx; ++x) { float &l = lum[(size_t) y * rx + x]; l = std::max(l - offset, 0.f) * sin_theta; }
    }
    /* hierarchy layout */
    const uint32_t np[2] = { rx - 1, ry - 1 };
    uint32_t max_level = 0; while ((1u << max_level) < std::max(np[0], np[1])) ++max_level;
    if (max_level + 1 > HAR_ENV_MAX_LEVELS) { err = "envmap: resolution too large"; return false; }
    std::vector<uint32_t> lw, lh; lw.push_back(rx); lh.push_back(ry);
    { uint32_t a = np[0], b = np[1]; for (uint32_t l = 0; l < max_level; ++l) { a += a & 1u; b += b & 1u; lw.push_back(a); lh.push_back(b); a >>= 1; b >>= 1; } }
    E.n_levels = (uint32_t) lw.size();
    uint32_t total = 0;
    for (uint32_t l = 0; l < E.n_levels; ++l) { total = (total + 3u) & ~3u; E.lvl_offset[l] = total; E.lvl_width[l] = lw[l]; total += lw[l] * lh[l]; }
    total = (total + 3u) & ~3u;
    std::vector<float> &Wd = hs.env_warp; Wd.assign(total, 0.f);
    const bool has_mip = E.n_levels > 1;
    double sum = 0.0;
    for (uint32_t y = 0; y < np[1]; ++y)
        for (uint32_t x = 0; x < np[0]; ++x) {
            const float *q = lum.data() + (size_t) y * rx + x;
            const float avg = .25f * (q[0] + q[1] + q[rx] + q[rx + 1]);
            sum += (double) avg;
            if (has_mip) Wd[E.lvl_offset[1] + hier_index(x, y, E.lvl_width[1])] = avg;
        }
    const float scale = (float) ((double) ((uint64_t) np[0] * np[1]) / sum);
    for (size_t i = 0; i < lum.size(); ++i) Wd[E.lvl_offset[0] + i] = lum[i] * scale;
    if (has_mip) for (uint32_t i = 0; i < lw[1] * lh[1]; ++i) Wd[E.lvl_offset[1] + i] *= scale;
    uint32_t cur[2] = { np[0], np[1] };
    for (uint32_t l = 2; l < E.n_levels; ++l) {
        cur[0] = (cur[0] + 1u) >> 1; cur[1] = (cur[1] + 1u) >> 1;
        for (uint32_t y = 0; y < cur[1]; ++y)
            for (uint32_t x = 0; x < cur[0]; ++x) {
                const float *d0 = Wd.data() + E.lvl_offset[l - 1] + hier_index(2 * x, 2 * y, E.lvl_width[l - 1]);
                Wd[E.lvl_offset[l] + hier_index(x, y, E.lvl_width[l])] = d0[0] + d0[1] + d0[2] + d0[3];
            }
    }
    E.tex = nullptr; E.warp = nullptr;
    hs.has_envmap = true;
    return true;
}

bool lower_scene(const HarSceneDesc &d, HostScene &hs, std::string &err) {
    if (d.top_mesh_count > d.mesh_count) { err = "top_mesh_count exceeds mesh_count"; return false; }
    for (uint32_t i = 0; i < d.mesh_count; ++i) {
        const HarMesh &m = d.meshes[i];
        if (m.bsdf >= d.bsdf_count) { err = "mesh references a BSDF that does not exist"; return false; }
        if (m.emitter >= (int32_t) d.emitter_count) { err = "mesh references an emitter that does not exist"; return false; }
        if ((m.vertex_count && !m.vertex_ptr) || (m.face_count && !m.index_ptr)) { err = "mesh has null buffers"; return false; }
        for (uint32_t f = 0; f < m.face_count; ++f)
            for (int k = 0; k < 3; ++k)
                if (m.index_ptr[4 * (size_t) f + k] >= m.vertex_count) { err = "face index out of bounds"; return false; }
        hs.top_mesh_count = d.top_mesh_count;
        DMesh dm{};
        dm.voff = (uint32_t) (hs.verts.size() / 8); dm.foff = (uint32_t) (hs.faces.size() / 4);
        dm.bsdf = m.bsdf; dm.emitter = m.emitter; dm.flags = m.flags; dm.face_count = m.face_count; dm.vertex_count = m.vertex_count;
        hs.verts.insert(hs.verts.end(), m.vertex_ptr, m.vertex_ptr + 8 * (size_t) m.vertex_count);
        hs.faces.insert(hs.faces.end(), m.index_ptr, m.index_ptr + 4 * (size_t) m.face_count);
#if HAR_SHADING_TRIS
        for (uint32_t f = 0; f < m.face_count; ++f)
            for (int k = 0; k < 3; ++k) { const float *v = m.vertex_ptr + 8 * (size_t) m.index_ptr[4 * (size_t) f + k]; hs.shade_tris.insert(hs.shade_tris.end(), v, v + 8); }
#endif
        hs.meshes.push_back(dm);
    }
    for (uint32_t i = 0; i < d.texture_count; ++i) {
        const HarTexture &t = d.textures[i];
        if (!t.data || !t.width || !t.height) { err = "empty texture"; return false; }
        if ((t.mode & ~7u) != 0u || (t.mode & 6u) == 6u) { err = "HarTexture::mode: HAR_TEX_BILINEAR / _NEAREST combined with HAR_TEX_REPEAT / _MIRROR / _CLAMP"; return false; }
        HostTexture ht; ht.w = t.width; ht.h = t.height; ht.mode = t.mode; ht.data.assign(t.data, t.data + 3 * (size_t) t.width * t.height);
        bool zero = true, ident = true;          /* to_uv (bitmap.cpp:175): six zeros = the identity of a zero-initialised record */
        const float id6[6] = { 1.f, 0.f, 0.f, 0.f, 1.f, 0.f };
        for (int k = 0; k < 6; ++k) { if (!std::isfinite(t.to_uv[k])) { err = "HarTexture::to_uv must be finite"; return false; } zero = zero && t.to_uv[k] == 0.f; ident = ident && t.to_uv[k] == id6[k]; }
        if (!zero && !ident) {
            if (t.to_uv[0] * t.to_uv[4] - t.to_uv[1] * t.to_uv[3] == 0.f) { err = "HarTexture::to_uv is singular"; return false; }
            for (int k = 0; k < 6; ++k) ht.uvm[k] = t.to_uv[k];
            ht.mode |= HAR_TEX_HAS_UV_XF;
        }
        hs.textures.push_back(std::move(ht));
    }
    for (uint32_t i = 0; i < d.bsdf_count; ++i) {
        const HarBSDF &b = d.bsdfs[i];
        if (b.type >= BSDF_TYPE_COUNT) { err = "unsupported BSDF type (diffuse, dielectric, conductor, plastic, roughconductor, roughplastic and twosided are implemented in hip_ad_rgb)"; return false; }
        if (b.texture >= (int32_t) d.texture_count) { err = "BSDF references a texture that does not exist"; return false; }
        if ((b.flags & BF_TWOSIDED) && b.back >= (int32_t) d.bsdf_count) { err = "twosided BSDF references a back-side BSDF that does not exist"; return false; }
        if ((b.flags & BF_TWOSIDED) && b.type == BSDF_DIELECTRIC) { err = "Only materials without a transmission component can be nested!"; return false; }   /* twosided.cpp:79-83 */
        if ((b.type == BSDF_DIELECTRIC || b.type == BSDF_ROUGHPLASTIC || b.type == BSDF_PLASTIC) && !(b.eta > 0.f)) { err = "The interior and exterior indices of refraction must be positive!"; return false; }
        if (b.type == BSDF_ROUGHPLASTIC && b.eta == 1.f) { err = "The interior and exterior indices of refraction must be positive and differ!"; return false; }
        if (b.type == BSDF_ROUGHPLASTIC && b.alpha_u != b.alpha_v) { err = "The 'roughplastic' plugin currently does not support anisotropic microfacet distributions!"; return false; }
        DBsdf db{}; db.type = b.type; db.texture = b.texture; db.r = b.reflectance[0]; db.g = b.reflectance[1]; db.b = b.reflectance[2];
        db.flags = b.flags; db.r2 = b.reflectance2[0]; db.g2 = b.reflectance2[1]; db.b2 = b.reflectance2[2];
        db.alpha_u = b.alpha_u; db.alpha_v = b.alpha_v; db.eta = b.eta;
        for (int k = 0; k < 3; ++k) { db.eta_c[k] = b.eta_c[k]; db.k_c[k] = b.k_c[k]; }
        db.back = (b.flags & BF_TWOSIDED) ? b.back : -1; db.table = -1;
        hs.bsdfs.push_back(db);
        if (b.type == BSDF_ROUGHPLASTIC) { build_roughplastic_tables(hs, i); update_roughplastic_sampling_weight(hs, i); }
        if (b.type == BSDF_PLASTIC) {                       /* SmoothPlastic::parameters_changed (plastic.cpp:188-205) */
            DBsdf &p = hs.bsdfs[i];
            p.inv_eta_2 = 1.f / (p.eta * p.eta);
            p.internal_reflectance = fresnel_diffuse_reflectance(1.f / p.eta);
            update_roughplastic_sampling_weight(hs, i);
        }
    }
    for (uint32_t i = 0; i < d.emitter_count; ++i) {
        const HarEmitter &e = d.emitters[i];
        if (e.type > 7) { err = "unsupported emitter type (`area`, `constant`, `envmap`, `point`, `spot` and `directional` are implemented)"; return false; }
        if (!(e.sampling_weight >= 0.f) || !std::isfinite(e.sampling_weight)) { err = "DiscreteDistribution: entries must be non-negative!"; return false; }      /* distr_1d.h:247-248 */
        const bool area = e.type == 0 || e.type == 3 || e.type == 7, point = e.type >= 4 && e.type <= 6;      /* the delta emitters */
        if (area && e.mesh >= d.top_mesh_count) { err = "area emitter must be attached to a top-level mesh"; return false; }
        if (!area && !point && hs.env_emitter >= 0) { err = "Only one environment emitter can be specified per scene."; return false; }   /* scene.cpp:64-65 */
        if (!area && !point) hs.env_emitter = (int32_t) i;
        if (point) hs.has_point_emitters = true;
        if (e.type == 2) {
            if (e.mesh >= d.texture_count) { err = "envmap emitter references a bitmap that does not exist"; return false; }
            if (!build_envmap(hs, d.textures[e.mesh], e, err)) return false;
        }
        DEmitter de{};
        if (!lower_plain_emitter(e, de, err)) return false;
        if (e.type == 7) {            /* AreaLight with a bitmap radiance: the texel distribution (see texel_table_fill) lives in emitter_cdf, the record holds where */
            if (e.radiance_texture >= d.texture_count) { err = "area emitter references a bitmap that does not exist"; return false; }
            const HostTexture &t = hs.textures[e.radiance_texture];
            const uint32_t off = (uint32_t) hs.emitter_cdf.size(), tex = e.radiance_texture;
            hs.emitter_cdf.resize(off + HAR_TEXEL_TABLE_HEADER + (size_t) t.h + (size_t) t.w * t.h);
            std::memcpy(&de.radiance[0], &tex, 4); std::memcpy(&de.radiance[1], &off, 4);
            /* Rectangle::update (rectangle.cpp:118-124): dp_du = to_world * (2, 0, 0), dp_dv = to_world * (0, 2, 0); |dp_du x dp_dv| = the surface area */
            const Vec3 du(e.to_world[0] * 2.f, e.to_world[1] * 2.f, e.to_world[2] * 2.f), dv(e.to_world[3] * 2.f, e.to_world[4] * 2.f, e.to_world[5] * 2.f);
            const Vec3 c = cross3(du, dv);
            de.radiance[2] = sqrtf(dot3(c, c));
            if (!texel_table_fill(hs, t, off, err)) return false;
            hs.has_mesh_emitters = true;       /* = "emitter tables in emitter_cdf": the scene runs the kernels that carry the generic emitter code */
        }
        if (e.type == 3) {            /* Mesh::build_pmf (mesh.cpp:1358-1372): face areas of the (world-space) mesh + their running sum */
            const DMesh &M = hs.meshes[e.mesh];
            if (M.face_count == 0) { err = "Cannot create sampling table for an empty mesh"; return false; }
            const uint32_t off = (uint32_t) hs.emitter_cdf.size();
            hs.emitter_cdf.resize(off + 2 * (size_t) M.face_count);
            float acc = 0.f;
            for (uint32_t f = 0; f < M.face_count; ++f) {
                const uint32_t *fi = hs.faces.data() + 4 * ((size_t) M.foff + f);
                auto P = [&](uint32_t v) { const float *q = hs.verts.data() + 8 * ((size_t) M.voff + v); return Vec3(q[0], q[1], q[2]); };
                const Vec3 c = cross3(P(fi[1]) - P(fi[0]), P(fi[2]) - P(fi[0]));
                const float a = .5f * sqrtf(dot3(c, c));
                acc += a; hs.emitter_cdf[off + f] = a; hs.emitter_cdf[off + M.face_count + f] = acc;
            }
            uint32_t nf = M.face_count;
            std::memcpy(&de.to_world[0], &off, 4); std::memcpy(&de.to_world[1], &nf, 4); de.to_world[2] = acc;
            de.inv_area = rcp_(acc);
            hs.has_mesh_emitters = true;
        }
        hs.emitters.push_back(de);
    }
    {
        std::vector<float> weights(d.emitter_count);
        for (uint32_t i = 0; i < d.emitter_count; ++i) weights[i] = d.emitters[i].sampling_weight;
        if (!build_emitter_distribution(hs, weights.data(), d.emitter_count, err)) return false;
    }
    for (uint32_t g = 0; g < d.group_count; ++g)
        if (d.groups[g].first_mesh < d.top_mesh_count || d.groups[g].first_mesh + d.groups[g].mesh_count > d.mesh_count) { err = "shapegroup mesh range invalid"; return false; }
    for (uint32_t i = 0; i < d.instance_count; ++i) {
        if (d.instances[i].group >= d.group_count) { err = "instance references a shapegroup that does not exist"; return false; }
        DInst di; std::memcpy(di.to_world, d.instances[i].to_world, 48); std::memcpy(di.to_object, d.instances[i].to_object, 48);
        hs.insts.push_back(di);
    }

    hs.groups.assign(d.groups, d.groups + d.group_count);
    hs.inst_group.resize(d.instance_count);
    for (uint32_t i = 0; i < d.instance_count; ++i) hs.inst_group[i] = d.instances[i].group;
    update_scene_bounds(hs);

    hs.blas_top = build_blas(hs, 0, d.top_mesh_count, d.instance_count != 0);
    hs.blas_depth = hs.stats.max_depth;
    if (d.instance_count == 0) {
        hs.root = hs.blas_top.root; hs.has_tlas = false; hs.tlas_first = (uint32_t) hs.nodes.size();
        hs.blas_tri_ranges = { hs.blas_top.first_tri, hs.blas_top.tri_count };
        return true;
    }
    for (uint32_t g = 0; g < d.group_count; ++g) hs.blas_groups.push_back(build_blas(hs, d.groups[g].first_mesh, d.groups[g].mesh_count));
    hs.blas_depth = hs.stats.max_depth;
    hs.tlas_first = (uint32_t) hs.nodes.size();
    hs.inst_boxes.resize(d.instance_count); hs.inst_box_valid.assign(d.instance_count, 0);
    return build_tlas(hs, err);
}

/* ConstantBackgroundEmitter::set_scene (constant.cpp:72-87): bounding sphere of Scene::bbox() (all shapes; an Instance
 * contributes the 8 transformed corners of its group's box, instance.cpp:93-103), radius * (1 + RayEpsilon) */
void update_scene_bounds(HostScene &hs) {
    bool needs_bounds = hs.env_emitter >= 0;
    for (const DEmitter &E : hs.emitters) needs_bounds = needs_bounds || E.type == 6u;      /* DirectionalEmitter::set_scene (directional.cpp:99-109): the same sphere */
    if (!needs_bounds) return;
    float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
    auto grow = [&](float x, float y, float z) { const float q[3] = { x, y, z }; for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], q[a]); hi[a] = std::max(hi[a], q[a]); } };
    auto vertex = [&](const DMesh &m, uint32_t v) { return hs.verts.data() + 8 * ((size_t) m.voff + v); };
    for (uint32_t s = 0; s < hs.top_mesh_count; ++s)
        for (uint32_t v = 0; v < hs.meshes[s].vertex_count; ++v) { const float *p = vertex(hs.meshes[s], v); grow(p[0], p[1], p[2]); }
    for (uint32_t i = 0; i < hs.insts.size(); ++i) {
        const HarShapeGroup &sg = hs.groups[hs.inst_group[i]];
        float glo[3] = { INFINITY, INFINITY, INFINITY }, ghi[3] = { -INFINITY, -INFINITY, -INFINITY };
        for (uint32_t s = sg.first_mesh; s < sg.first_mesh + sg.mesh_count; ++s)
            for (uint32_t v = 0; v < hs.meshes[s].vertex_count; ++v) { const float *p = vertex(hs.meshes[s], v); for (int a = 0; a < 3; ++a) { glo[a] = std::min(glo[a], p[a]); ghi[a] = std::max(ghi[a], p[a]); } }
        if (!(glo[0] <= ghi[0])) continue;
        for (int c = 0; c < 8; ++c) { Vec3 q = xf_point(hs.insts[i].to_world, Vec3(c & 1 ? ghi[0] : glo[0], c & 2 ? ghi[1] : glo[1], c & 4 ? ghi[2] : glo[2])); grow(q.x, q.y, q.z); }
    }
    float bs[4] = { 0.f, 0.f, 0.f, HAR_RAY_EPS };        /* centre, radius */
    if (lo[0] <= hi[0]) {
        Vec3 c((hi[0] + lo[0]) * .5f, (hi[1] + lo[1]) * .5f, (hi[2] + lo[2]) * .5f);
        float r = norm3(c - Vec3(hi[0], hi[1], hi[2]));
        bs[0] = c.x; bs[1] = c.y; bs[2] = c.z; bs[3] = std::max(HAR_RAY_EPS, r * (1.f + HAR_RAY_EPS));
    }
    for (DEmitter &D : hs.emitters)
        if (D.type == 6u) {          /* the record of a directional light: its direction in [0..2] (set when the record was lowered), then the sphere */
            D.to_world[3] = bs[0]; D.to_world[4] = bs[1]; D.to_world[5] = bs[2]; D.to_world[6] = bs[3];
        }
    if (hs.env_emitter >= 0) {
        DEmitter &E = hs.emitters[hs.env_emitter];
        E.to_world[0] = bs[0]; E.to_world[1] = bs[1]; E.to_world[2] = bs[2]; E.to_world[3] = bs[3];
        E.mesh = 0xffffffffu;
    }
    if (hs.env_emitter >= 0 && hs.has_envmap) {
        DEmitter &E = hs.emitters[hs.env_emitter];            /* EnvironmentMapEmitter::set_scene (envmap.cpp:214-226): the same rule */
        for (int a = 0; a < 3; ++a) hs.envmap.center[a] = E.to_world[a];
        hs.envmap.radius = E.to_world[3];
    }
}

/* Texel distribution of a bitmap that an area light radiates (emitter type 7), written to hs.emitter_cdf[off ..]:
 *   [0] (float) sum, [1] (float) (1 / sum)   -- DiscreteDistribution2D::m_inv_normalization / m_normalization (include/mitsuba/core/distr_2d.h:93-118)
 *   [2..7] the inverse of the bitmap's to_uv (row-major 2 x 3): sample_position returns m_transform.inverse() * sample (bitmap.cpp:658)
 *   [8 ..] marginal running sums (h), then the conditional running sums row by row (w * h); both accumulated in double and stored per entry as float
 * over the texels' luminance (rebuild_internals, bitmap.cpp:876-955; luminance(): spectrum.h:439-442).  to_uv has to map the unit square's corners onto themselves
 * (check_sampling_transform, bitmap.cpp:976-992). */
bool texel_table_inputs_ok(const float uvm[6], const float *texels, uint32_t w, uint32_t h, std::string &err) {
    const float cx[4] = { 0.f, 1.f, 1.f, 0.f }, cy[4] = { 0.f, 0.f, 1.f, 1.f };
    uint32_t found = 0;
    for (int c = 0; c < 4; ++c) {
        const float qx = fma_(uvm[1], cy[c], fma_(uvm[0], cx[c], uvm[2])), qy = fma_(uvm[4], cy[c], fma_(uvm[3], cx[c], uvm[5]));
        for (uint32_t j = 0; j < 4; ++j) { const float dx = qx - cx[j], dy = qy - cy[j]; if (dx * dx + dy * dy < 1e-8f) found |= 1u << j; }
    }
    if (found != 0xFu) { err = "Bitmap texture: position sampling (e.g. of an area emitter's radiance) requires a 'to_uv' transformation that maps the unit square onto itself, such as a flip, a transpose or a multiple of a 90 degree rotation."; return false; }
    double total = 0.0;
    for (size_t i = 0; i < (size_t) w * h; ++i) {
        const float lum = texels[3 * i] * 0.212671f + texels[3 * i + 1] * 0.715160f + texels[3 * i + 2] * 0.072169f;
        if (!(lum >= 0.f) || !std::isfinite(lum)) { err = "area emitter: the radiance bitmap must be finite and non-negative"; return false; }
        total += (double) lum;
    }
    if (!(total > 0.0)) { err = "area emitter: the radiance bitmap has no luminance to sample"; return false; }
    return true;
}
bool texel_table_fill(HostScene &hs, const HostTexture &t, uint32_t off, std::string &err) {
    if (!texel_table_inputs_ok(t.uvm, t.data.data(), t.w, t.h, err)) return false;
    float *tab = hs.emitter_cdf.data() + off, *marg = tab + HAR_TEXEL_TABLE_HEADER, *cond = marg + t.h;
    double total = 0.0;
    for (uint32_t y = 0; y < t.h; ++y) {
        double row = 0.0;
        for (uint32_t x = 0; x < t.w; ++x) {
            const float *px = t.data.data() + 3 * ((size_t) y * t.w + x);
            const float lum = px[0] * 0.212671f + px[1] * 0.715160f + px[2] * 0.072169f;
            row += (double) lum; cond[(size_t) y * t.w + x] = (float) row;
        }
        total += row; marg[y] = (float) total;
    }
    if (!(total > 0.0)) { err = "area emitter: the radiance bitmap has no luminance to sample"; return false; }
    tab[0] = (float) total; tab[1] = (float) (1.0 / total);
    const float a = t.uvm[0], b = t.uvm[1], c = t.uvm[3], d = t.uvm[4], inv_det = 1.f / (a * d - b * c);
    tab[2] = d * inv_det; tab[3] = -b * inv_det; tab[5] = -c * inv_det; tab[6] = a * inv_det;
    tab[4] = -(tab[2] * t.uvm[2] + tab[3] * t.uvm[5]); tab[7] = -(tab[5] * t.uvm[2] + tab[6] * t.uvm[5]);
    return true;
}

/* The instance level: one TLAS leaf per Instance whose group is not empty, over the world-space box of the instance; hs.nodes is cut back to tlas_first and the new
 * TLAS appended, hs.inst_recs / hs.blas_tri_ranges rewritten in leaf order.  Boxes are recomputed for the instances marked invalid (all of them at creation). */
bool build_tlas(HostScene &hs, std::string &err) {
    const uint32_t n_inst = (uint32_t) hs.insts.size();
    const BlasInfo &top = hs.blas_top;
    std::vector<PrimBox> boxes; std::vector<InstRec> recs; std::vector<uint32_t> ranges;
    /* the top-level geometry is not a TLAS entry: rays walk its BLAS first and then the TLAS (Accel::top_root, har_accel.h) */
    if (!top.empty) { hs.top_root = top.root; hs.top_first = top.first_tri; hs.top_count = top.tri_count; }
    /* Accel::top_last: TLAS-first order when the top-level geometry is a handful of triangles around the instanced content.  HAR_TOP_LAST_MAX: the
     * threshold (triangles), HAR_TOP_LAST: force the bit mask (A/B: 0 = round-2 order, 1 = any-hit only, 3 = both) */
    static const uint32_t top_last_max = getenv("HAR_TOP_LAST_MAX") ? (uint32_t) atoi(getenv("HAR_TOP_LAST_MAX")) : 256u;
    static const int top_last_forced = getenv("HAR_TOP_LAST") ? atoi(getenv("HAR_TOP_LAST")) : -1;
    hs.top_last = top_last_forced >= 0 ? (uint32_t) top_last_forced : (!top.empty && top.tri_count <= top_last_max ? 3u : 0u);
    for (uint32_t i = 0; i < n_inst; ++i) {
        const BlasInfo &g = hs.blas_groups[hs.inst_group[i]];
        if (g.empty) continue;
        InstRec r{}; std::memcpy(r.to_world, hs.insts[i].to_world, 48); std::memcpy(r.to_object, hs.insts[i].to_object, 48);
        r.blas_root = g.root; r.inst_index = i; r.identity = 0;
        if (!hs.inst_box_valid[i]) {
            PrimBox b; for (int a = 0; a < 3; ++a) { b.lo[a] = INFINITY; b.hi[a] = -INFINITY; }
            auto grow = [&](Vec3 q) {
                const float qq[3] = { q.x, q.y, q.z };
                for (int a = 0; a < 3; ++a) { b.lo[a] = std::min(b.lo[a], qq[a]); b.hi[a] = std::max(b.hi[a], qq[a]); }
            };
            /* Instance::bbox (src/shapes/instance.cpp:93-103) transforms the 8 corners of the group's box, which inflates
             * the box of a rotated object by up to sqrt(3).  A TLAS leaf only has to bound the instance, so use the exact
             * bound of the transformed vertices when that is cheap (it cuts instance entries per ray by ~1/4 on the
             * 1M-triangle benchmark scene) and the reference's corner bound otherwise. */
            const HarShapeGroup &sg = hs.groups[hs.inst_group[i]];
            uint64_t nverts = 0;
            for (uint32_t s = sg.first_mesh; s < sg.first_mesh + sg.mesh_count; ++s) nverts += hs.meshes[s].vertex_count;
            if (nverts * (uint64_t) n_inst <= 200000000ull) {
                for (uint32_t s = sg.first_mesh; s < sg.first_mesh + sg.mesh_count; ++s) {
                    const DMesh &m = hs.meshes[s];
                    const float *vertex_ptr = hs.verts.data() + 8 * (size_t) m.voff; const uint32_t *index_ptr = hs.faces.data() + 4 * (size_t) m.foff;
                    for (uint32_t f = 0; f < m.face_count; ++f)          /* only referenced vertices */
                        for (int k = 0; k < 3; ++k) { const float *v = vertex_ptr + 8 * (size_t) index_ptr[4 * (size_t) f + k]; grow(xf_point(r.to_world, Vec3(v[0], v[1], v[2]))); }
                }
            } else {
                for (int c = 0; c < 8; ++c) grow(xf_point(r.to_world, Vec3(c & 1 ? g.hi[0] : g.lo[0], c & 2 ? g.hi[1] : g.lo[1], c & 4 ? g.hi[2] : g.lo[2])));
            }
            pad_prim_box(b);
            hs.inst_boxes[i] = b; hs.inst_box_valid[i] = 1;
        }
        boxes.push_back(hs.inst_boxes[i]); recs.push_back(r); ranges.push_back(g.first_tri); ranges.push_back(g.tri_count);
    }
    std::vector<uint32_t> order;
    Bvh8Stats tstats;
    const uint32_t tlas_leaf = 1u;
    static const float inst_cost = getenv("HAR_BVH_CINST") ? (float) atof(getenv("HAR_BVH_CINST")) : 1.5f;  /* instance entry = ray transform + a BLAS root visit */
    hs.nodes.resize(hs.tlas_first);
    hs.root = build_bvh8(boxes, hs.nodes, 0, order, &tstats, tlas_leaf, inst_cost, 0);
    hs.tlas_depth = tstats.max_depth; hs.stats.max_depth = std::max(hs.blas_depth, tstats.max_depth);
    hs.has_tlas = true;
    hs.inst_recs.clear(); hs.blas_tri_ranges.clear();
    for (uint32_t k : order) { hs.inst_recs.push_back(recs[k]); hs.blas_tri_ranges.push_back(ranges[2 * k]); hs.blas_tri_ranges.push_back(ranges[2 * k + 1]); }
    tlas_refit_order(hs);
    (void) err;
    return true;
}

/* Scene::update_emitter_sampling_distribution (scene.cpp:120-141): a DiscreteDistribution over the weights as soon as one differs from 1; its tables are built by
 * compute_cdf_scalar (distr_1d.h:236-266: running sums in double, rounded to float per entry; first / last bin with mass) in every variant, because the
 * constructor taking a ScalarFloat array is the one used */
bool build_emitter_distribution(HostScene &hs, const float *weights, uint32_t n, std::string &err) {
    bool non_uniform = false;
    for (uint32_t i = 0; i < n; ++i) {
        if (!(weights[i] >= 0.f) || !std::isfinite(weights[i])) { err = "DiscreteDistribution: entries must be non-negative!"; return false; }      /* distr_1d.h:247-248 */
        non_uniform = non_uniform || weights[i] != 1.f;
    }
    hs.emitter_distr.clear(); hs.emitter_sum = 0.f; hs.emitter_norm = 0.f; hs.emitter_valid_lo = 0; hs.emitter_valid_hi = 0;
    if (!non_uniform) return true;
    hs.emitter_distr.assign(2 * (size_t) n, 0.f);
    double sum = 0.0; uint32_t lo = 0xffffffffu, hi = 0xffffffffu;
    for (uint32_t i = 0; i < n; ++i) {
        const double v = (double) weights[i];
        sum += v; hs.emitter_distr[i] = weights[i]; hs.emitter_distr[n + i] = (float) sum;
        if (v > 0.0) { if (lo == 0xffffffffu) lo = i; hi = i; }
    }
    if (lo == 0xffffffffu) { hs.emitter_distr.clear(); err = "DiscreteDistribution: no probability mass found!"; return false; }
    hs.emitter_valid_lo = lo; hs.emitter_valid_hi = hi;
    hs.emitter_sum = hs.emitter_distr[n + hi]; hs.emitter_norm = rcp_(hs.emitter_sum);
    return true;
}

bool scene_set_instances_host(HostScene &hs, uint32_t first, uint32_t count, const float *to_world, const float *to_object, std::string &err) {
    if ((uint64_t) first + count > hs.insts.size()) { err = "instance range out of bounds"; return false; }
    if (!hs.has_tlas) { err = "the scene has no instances"; return false; }
    for (uint32_t k = 0; k < count; ++k) {
        for (int j = 0; j < 12; ++j) if (!std::isfinite(to_world[12 * k + j]) || !std::isfinite(to_object[12 * k + j])) { err = "instance transform is not finite"; return false; }
        std::memcpy(hs.insts[first + k].to_world, to_world + 12 * (size_t) k, 48); std::memcpy(hs.insts[first + k].to_object, to_object + 12 * (size_t) k, 48);
        hs.inst_box_valid[first + k] = 0;
    }
    if (!build_tlas(hs, err)) return false;
    update_scene_bounds(hs);
    return true;
}

BlasInfo *scene_set_vertices_host(HostScene &hs, uint32_t mesh, const float *vertices, std::string &err) {
    if (mesh >= hs.meshes.size()) { err = "invalid mesh index"; return nullptr; }
    const DMesh &m = hs.meshes[mesh];
    if (m.emitter >= 0) { err = "the mesh carries an area emitter (its sampling records are lowered from the positions): create a new scene"; return nullptr; }
    for (size_t k = 0; k < 8 * (size_t) m.vertex_count; ++k) if ((k & 7) < 3 && !std::isfinite(vertices[k])) { err = "vertex position is not finite"; return nullptr; }
    std::memcpy(hs.verts.data() + 8 * (size_t) m.voff, vertices, 32 * (size_t) m.vertex_count);
#if HAR_SHADING_TRIS
    for (uint32_t f = 0; f < m.face_count; ++f)
        for (int k = 0; k < 3; ++k) std::memcpy(hs.shade_tris.data() + 24 * ((size_t) m.foff + f) + 8 * k, vertices + 8 * (size_t) hs.faces[4 * ((size_t) m.foff + f) + k], 32);
#endif
    if (mesh < hs.top_mesh_count) return &hs.blas_top;
    for (size_t g = 0; g < hs.groups.size(); ++g)
        if (mesh >= hs.groups[g].first_mesh && mesh < hs.groups[g].first_mesh + hs.groups[g].mesh_count) return &hs.blas_groups[g];
    err = "mesh belongs to no BLAS"; return nullptr;
}

/* the group's box (corner bound of big groups, build_tlas) = union of the padded triangle boxes, as build_blas leaves it; the cached boxes of its instances are dropped */
static void recompute_group_box(HostScene &hs, size_t g) {
    BlasInfo *B = &hs.blas_groups[g];
    for (int a = 0; a < 3; ++a) { B->lo[a] = INFINITY; B->hi[a] = -INFINITY; }
    const HarShapeGroup &sg = hs.groups[g];
    for (uint32_t s = sg.first_mesh; s < sg.first_mesh + sg.mesh_count; ++s) {
        const DMesh &m = hs.meshes[s];
        for (uint32_t f = 0; f < m.face_count; ++f) {
            PrimBox b; for (int a = 0; a < 3; ++a) { b.lo[a] = INFINITY; b.hi[a] = -INFINITY; }
            for (int k = 0; k < 3; ++k) { const float *v = hs.verts.data() + 8 * ((size_t) m.voff + hs.faces[4 * ((size_t) m.foff + f) + k]); for (int a = 0; a < 3; ++a) { b.lo[a] = std::min(b.lo[a], v[a]); b.hi[a] = std::max(b.hi[a], v[a]); } }
            pad_prim_box(b);
            for (int a = 0; a < 3; ++a) { B->lo[a] = std::min(B->lo[a], b.lo[a]); B->hi[a] = std::max(B->hi[a], b.hi[a]); }
        }
    }
    for (size_t i = 0; i < hs.insts.size(); ++i) if (hs.inst_group[i] == g) hs.inst_box_valid[i] = 0;
    if (g < hs.group_box_stale.size()) hs.group_box_stale[g] = 0;
}
void recompute_stale_group_boxes(HostScene &hs) {
    for (size_t g = 0; g < hs.group_box_stale.size(); ++g) if (hs.group_box_stale[g]) recompute_group_box(hs, g);
}

bool scene_after_refit_host(HostScene &hs, BlasInfo *B, std::string &err) {
    B->refits++;
    if (B != &hs.blas_top) {
        recompute_group_box(hs, (size_t) (B - hs.blas_groups.data()));
        if (!build_tlas(hs, err)) return false;
    }
    update_scene_bounds(hs);
    return true;
}

double refit_blas_host(HostScene &hs, BlasInfo &B) {
    if (B.empty) return 0.0;
    std::vector<RefitBox> tri_box(hs.tris.size()), node_box(hs.nodes.size());
    for (uint32_t i = 0; i < B.tri_count; ++i) tri_box[B.first_tri + i] = refit_triangle(hs.meshes.data(), hs.verts.data(), hs.faces.data(), hs.tris.data(), B.first_tri + i);
    double area = 0.0;
    for (uint32_t k = 0; k < B.node_count; ++k) area += (double) refit_node(hs.nodes.data(), hs.refit_order[B.order_first + k], tri_box.data(), node_box.data());
    return area;
}

void refit_tlas_host(HostScene &hs) {
    if (!hs.has_tlas || hs.tlas_order.empty()) return;
    std::vector<RefitBox> inst_box(hs.inst_recs.size()), node_box(hs.nodes.size());
    for (size_t r = 0; r < hs.inst_recs.size(); ++r) {
        const HarShapeGroup &sg = hs.groups[hs.inst_group[hs.inst_recs[r].inst_index]];
        const uint32_t vfirst = hs.meshes[sg.first_mesh].voff; uint32_t vcount = 0;
        for (uint32_t s = sg.first_mesh; s < sg.first_mesh + sg.mesh_count; ++s) vcount += hs.meshes[s].vertex_count;
        RefitBox b = instance_box_empty();
        for (uint32_t v = 0; v < vcount; ++v) instance_box_grow(b, hs.inst_recs[r].to_world, hs.verts.data() + 8 * ((size_t) vfirst + v));
        instance_box_finish(b);
        inst_box[r] = b;
    }
    for (uint32_t idx : hs.tlas_order) (void) refit_node(hs.nodes.data(), idx, inst_box.data(), node_box.data());
}

static bool lower_sensor_filter(const HarSensor &in, DSensor &out, std::string &err) {
    if (in.rfilter == 0) { out.radius = 0.5f; return true; }
    if (in.rfilter == 2 || in.rfilter == 5) {                  /* tent: radius; lanczos: radius = lobes */
        if (!(in.rfilter_stddev > 0.f)) { err = "reconstruction filter: the radius / lobe count must be positive"; return false; }
        out.radius = in.rfilter == 5 ? (float) (int) in.rfilter_stddev : in.rfilter_stddev; return true;
    }
    if (in.rfilter == 3 || in.rfilter == 4) { out.radius = 2.f; return true; }
    float stddev = in.rfilter_stddev;
    out.radius = 4 * stddev;
    // Remez fit of exp(-x/2), scaled by 1/stddev^(2i) and shifted to reach 0 at the radius
    const double coeff[10] = { 9.992604880e-1, -4.977025247e-1, 1.222248550e-1, -1.932406282e-2, 2.136713061e-3,
                               -1.679873860e-4, 9.202145248e-6, -3.329417433e-7, 7.128382794e-9, -6.821193280e-11 };
    double scale = 1;
    for (int i = 0; i < 10; ++i) { out.coeff[i] = (float) (coeff[i] * scale); scale /= (double) stddev * (double) stddev; }
    out.coeff[0] -= estrin10(out.radius * out.radius, out.coeff);
    return true;
}

bool lower_sensor(const HarSensor &in, DSensor &out, std::string &err) {
    if (in.crop_width == 0 || in.crop_height == 0) { err = "empty crop window"; return false; }
    if (in.rfilter > 5) { err = "unsupported reconstruction filter (box, gaussian, tent, mitchell, catmullrom and lanczos are implemented)"; return false; }
    if (in.projection > 1) { err = "unsupported sensor projection (0 = perspective, 1 = orthographic)"; return false; }
    out.projection = in.projection;
    std::memcpy(out.s2c, in.sample_to_camera, 64); std::memcpy(out.to_world, in.to_world, 64);
    out.near_clip = in.near_clip; out.far_clip = in.far_clip;
    out.crop_x = in.crop_offset_x; out.crop_y = in.crop_offset_y; out.crop_w = in.crop_width; out.crop_h = in.crop_height;
    out.rfilter = in.rfilter;
    std::memset(out.coeff, 0, sizeof(out.coeff));
    out.rf_p0 = in.rfilter_stddev; out.rf_p1 = in.rfilter_param1;
    out.ppo_x = (float) in.film_width * in.principal_point_offset_x / (float) in.crop_width;
    out.ppo_y = (float) in.film_height * in.principal_point_offset_y / (float) in.crop_height;
    if (!lower_sensor_filter(in, out, err)) return false;
    /* ReconstructionFilter::init_discretization (rfilter.cpp:22): border_size = ceil(radius - 1/2 - 2 RayEpsilon) */
    out.border = in.sample_border ? (uint32_t) std::max(0, (int) std::ceil(out.radius - .5f - 2.f * HAR_RAY_EPS)) : 0u;
    out.samp_w = out.crop_w + 2u * out.border; out.samp_h = out.crop_h + 2u * out.border;
    return true;
}

} // namespace har
