/*
 * orc_bsdf.h -- ORACLE restatement of the reference's BSDF plugins (TEST INFRASTRUCTURE ONLY).
 *
 * Class-per-plugin restatement, written against the reference sources (not against the product):
 *   SmoothDiffuse      src/bsdfs/diffuse.cpp:100-179
 *   SmoothDielectric   src/bsdfs/dielectric.cpp:245-353
 *   RoughConductor     src/bsdfs/roughconductor.cpp:226-520
 *   RoughPlastic       src/bsdfs/roughplastic.cpp:204-420
 *   TwoSidedBRDF       src/bsdfs/twosided.cpp:112-270
 *   fresnel / fresnel_conductor / reflect / refract   include/mitsuba/render/fresnel.h:35-116,276-313
 *   MicrofacetDistribution, eval_reflectance / eval_transmittance   include/mitsuba/render/microfacet.h:64-567
 *   quad::gauss_legendre   include/mitsuba/core/quad.h:27-90
 *
 * Pinned by the golden vectors of src/render/tests/test_microfacet.py, src/bsdfs/tests/test_dielectric.py and
 * src/bsdfs/tests/test_twosided.py (tests/golden/reference_kats.json).  Parity unpinned: dr::erf / dr::erfinv /
 * dr::exp / dr::log / dr::tan are Dr.Jit polynomials that are not in the tree; Cephes-style restatements (orc_math.h)
 * and Giles' erfinv approximation are used here.
 */
#pragma once
#include "orc_math.h"
#include "mi_oracle.h"
#include <vector>

namespace orc {

static inline float safe_sqrt(float x) { return std::sqrt(std::fmax(x, 0.f)); }
static inline float lerp(float a, float b, float t) { return fmadd(b, t, fnmadd(a, t, a)); }

struct FresnelResult { float r, cos_theta_t, eta_it, eta_ti; };
static inline FresnelResult fresnel(float cos_theta_i, float eta) {
    FresnelResult f;
    bool outside_mask = cos_theta_i >= 0.f;
    float rcp_eta = rcp(eta);
    f.eta_it = outside_mask ? eta : rcp_eta;
    f.eta_ti = outside_mask ? rcp_eta : eta;
    float cos_theta_t_sqr = fnmadd(fnmadd(cos_theta_i, cos_theta_i, 1.f), f.eta_ti * f.eta_ti, 1.f);
    float cos_theta_i_abs = std::fabs(cos_theta_i), cos_theta_t_abs = safe_sqrt(cos_theta_t_sqr);
    bool index_matched = eta == 1.f, special_case = index_matched || (cos_theta_i_abs == 0.f);
    float r_sc = index_matched ? 0.f : 1.f;
    float a_s = fnmadd(f.eta_it, cos_theta_t_abs, cos_theta_i_abs) / fmadd(f.eta_it, cos_theta_t_abs, cos_theta_i_abs);
    float a_p = fnmadd(f.eta_it, cos_theta_i_abs, cos_theta_t_abs) / fmadd(f.eta_it, cos_theta_i_abs, cos_theta_t_abs);
    f.r = 0.5f * (sqr(a_s) + sqr(a_p));
    if (special_case) f.r = r_sc;
    f.cos_theta_t = mulsign_neg(cos_theta_t_abs, cos_theta_i);
    return f;
}

static inline float fresnel_conductor(float cos_theta_i, float eta_r, float eta_i) {
    float cos_theta_i_2 = cos_theta_i * cos_theta_i, sin_theta_i_2 = 1.f - cos_theta_i_2, sin_theta_i_4 = sin_theta_i_2 * sin_theta_i_2;
    float temp_1 = eta_r * eta_r - eta_i * eta_i - sin_theta_i_2,
          a_2_pb_2 = safe_sqrt(temp_1 * temp_1 + 4.f * eta_i * eta_i * eta_r * eta_r),
          a = safe_sqrt(.5f * (a_2_pb_2 + temp_1));
    float term_1 = a_2_pb_2 + cos_theta_i_2, term_2 = 2.f * cos_theta_i * a;
    float r_s = (term_1 - term_2) / (term_1 + term_2);
    float term_3 = a_2_pb_2 * cos_theta_i_2 + sin_theta_i_4, term_4 = term_2 * sin_theta_i_2;
    float r_p = r_s * (term_3 - term_4) / (term_3 + term_4);
    return 0.5f * (r_s + r_p);
}

static inline V3 reflect(V3 wi) { return V3(-wi.x, -wi.y, wi.z); }
static inline V3 reflect(V3 wi, V3 m) { float s = 2.f * dot(wi, m); return V3(fmsub(m.x, s, wi.x), fmsub(m.y, s, wi.y), fmsub(m.z, s, wi.z)); }
static inline V3 refract(V3 wi, float cos_theta_t, float eta_ti) { return V3(-eta_ti * wi.x, -eta_ti * wi.y, cos_theta_t); }
static inline V3 refract(V3 wi, V3 m, float cos_theta_t, float eta_ti) {
    float s = fmadd(dot(wi, m), eta_ti, cos_theta_t);
    return V3(fmsub(m.x, s, wi.x * eta_ti), fmsub(m.y, s, wi.y * eta_ti), fmsub(m.z, s, wi.z * eta_ti));
}

/* erfinv (Giles 2010, single precision).  dr::erfinv: NOT IN TREE, parity unpinned */
static inline float erfinv(float x) {
    float w = -log32((1.f - x) * (1.f + x)), p;
    if (w < 5.f) {
        w -= 2.5f;
        const float c[9] = { 2.81022636e-08f, 3.43273939e-07f, -3.5233877e-06f, -4.39150654e-06f, 0.00021858087f, -0.00125372503f, -0.00417768164f, 0.246640727f, 1.50140941f };
        p = c[0]; for (int i = 1; i < 9; ++i) p = fmadd(p, w, c[i]);
    } else {
        w = std::sqrt(w) - 3.f;
        const float c[9] = { -0.000200214257f, 0.000100950558f, 0.00134934322f, -0.00367342844f, 0.00573950773f, -0.0076224613f, 0.00943887047f, 1.00167406f, 2.83297682f };
        p = c[0]; for (int i = 1; i < 9; ++i) p = fmadd(p, w, c[i]);
    }
    return p * x;
}

enum class MicrofacetType { Beckmann = 0, GGX = 1 };

class MicrofacetDistribution {
public:
    MicrofacetDistribution(MicrofacetType type, float alpha_u, float alpha_v, bool sample_visible = true)
        : m_type(type), m_alpha_u(std::fmax(alpha_u, 1e-4f)), m_alpha_v(std::fmax(alpha_v, 1e-4f)), m_sample_visible(sample_visible) {}
    bool sample_visible() const { return m_sample_visible; }
    bool is_isotropic() const { return m_alpha_u == m_alpha_v; }

    float eval(V3 m) const {
        float alpha_uv = m_alpha_u * m_alpha_v, cos_theta = m.z, cos_theta_2 = sqr(cos_theta), result;
        if (m_type == MicrofacetType::Beckmann)
            result = exp32(-(sqr(m.x / m_alpha_u) + sqr(m.y / m_alpha_v)) / cos_theta_2) / (Pi * alpha_uv * sqr(cos_theta_2));
        else
            result = rcp(Pi * alpha_uv * sqr(sqr(m.x / m_alpha_u) + sqr(m.y / m_alpha_v) + sqr(m.z)));
        return (result * cos_theta > 1e-20f) ? result : 0.f;
    }
    float pdf(V3 wi, V3 m) const {
        float result = eval(m);
        if (m_sample_visible) result *= smith_g1(wi, m) * std::fabs(dot(wi, m)) / wi.z;
        else result *= m.z;
        return result;
    }
    float smith_g1(V3 v, V3 m) const {
        float xy_alpha_2 = sqr(m_alpha_u * v.x) + sqr(m_alpha_v * v.y), tan_theta_alpha_2 = xy_alpha_2 / sqr(v.z), result;
        if (m_type == MicrofacetType::Beckmann) {
            float a = rsqrt(tan_theta_alpha_2), a_sqr = sqr(a);
            result = (a >= 1.6f) ? 1.f : (3.535f * a + 2.181f * a_sqr) / (1.f + 2.276f * a + 2.577f * a_sqr);
        } else {
            result = 2.f / (1.f + std::sqrt(1.f + tan_theta_alpha_2));
        }
        if (xy_alpha_2 == 0.f) result = 1.f;
        if (dot(v, m) * v.z <= 0.f) result = 0.f;
        return result;
    }
    float G(V3 wi, V3 wo, V3 m) const { return smith_g1(wi, m) * smith_g1(wo, m); }

    void sample_visible_11(float cos_theta_i, float sx, float sy, float &out_x, float &out_y) const {
        const float InvSqrtPi = 0.56418958354775628695f;
        if (m_type == MicrofacetType::Beckmann) {
            float tan_theta_i = safe_sqrt(fnmadd(cos_theta_i, cos_theta_i, 1.f)) / cos_theta_i;
            float cot_theta_i = rcp(tan_theta_i);
            float maxval = erf32(cot_theta_i);
            sx = std::fmax(std::fmin(sx, 1.f - 1e-6f), 1e-6f); sy = std::fmax(std::fmin(sy, 1.f - 1e-6f), 1e-6f);
            float x = maxval - (maxval + 1.f) * erf32(std::sqrt(-log32(sx)));
            sx *= 1.f + maxval + InvSqrtPi * tan_theta_i * exp32(-sqr(cot_theta_i));
            for (int i = 0; i < 3; ++i) {
                float slope = erfinv(x), value = 1.f + x + InvSqrtPi * tan_theta_i * exp32(-sqr(slope)) - sx, derivative = 1.f - slope * tan_theta_i;
                x -= value / derivative;
            }
            out_x = erfinv(x); out_y = erfinv(fmsub(2.f, sy, 1.f));
        } else {
            // warp::square_to_uniform_disk_concentric (warp.h:54-90)
            float x = fmsub(2.f, sx, 1.f), y = fmsub(2.f, sy, 1.f);
            bool is_zero = x == 0.f && y == 0.f, quadrant_1_or_3 = std::fabs(x) < std::fabs(y);
            float r = quadrant_1_or_3 ? y : x, rp = quadrant_1_or_3 ? x : y;
            float phi = .25f * Pi * rp / r;
            if (quadrant_1_or_3) phi = .5f * Pi - phi;
            if (is_zero) phi = 0.f;
            float c, s = sincos(phi, &c);
            float px = r * c, py = r * s;
            float sc = 0.5f * (1.f + cos_theta_i);
            py = lerp(safe_sqrt(1.f - sqr(px)), py, sc);
            float z = safe_sqrt(1.f - fmadd(py, py, px * px));          // squared_norm(p) = dot(p, p)
            float sin_theta_i = safe_sqrt(1.f - sqr(cos_theta_i));
            float nrm = rcp(fmadd(sin_theta_i, py, cos_theta_i * z));
            out_x = fmsub(cos_theta_i, py, sin_theta_i * z) * nrm; out_y = px * nrm;
        }
    }

    V3 sample(V3 wi, float sx, float sy, float &pdf_out) const {
        if (!m_sample_visible) {
            float sin_phi, cos_phi, cos_theta, cos_theta_2, alpha_2;
            if (is_isotropic()) {
                sin_phi = sincos((2.f * Pi) * sy, &cos_phi);
                alpha_2 = m_alpha_u * m_alpha_u;
            } else {
                float ratio = m_alpha_v / m_alpha_u, tmp = ratio * tan32((2.f * Pi) * sy);
                cos_phi = rsqrt(fmadd(tmp, tmp, 1.f));
                cos_phi = mulsign(cos_phi, std::fabs(sy - .5f) - .25f);
                sin_phi = cos_phi * tmp;
                alpha_2 = rcp(sqr(cos_phi / m_alpha_u) + sqr(sin_phi / m_alpha_v));
            }
            if (m_type == MicrofacetType::Beckmann) {
                cos_theta = rsqrt(fnmadd(alpha_2, log32(1.f - sx), 1.f));
                cos_theta_2 = sqr(cos_theta);
                float cos_theta_3 = std::fmax(cos_theta_2 * cos_theta, 1e-20f);
                pdf_out = (1.f - sx) / (Pi * m_alpha_u * m_alpha_v * cos_theta_3);
            } else {
                float tan_theta_m_2 = alpha_2 * sx / (1.f - sx);
                cos_theta = rsqrt(1.f + tan_theta_m_2);
                cos_theta_2 = sqr(cos_theta);
                float temp = 1.f + tan_theta_m_2 / alpha_2, cos_theta_3 = std::fmax(cos_theta_2 * cos_theta, 1e-20f);
                pdf_out = rcp(Pi * m_alpha_u * m_alpha_v * cos_theta_3 * sqr(temp));
            }
            float sin_theta = std::sqrt(1.f - cos_theta_2);
            return V3(cos_phi * sin_theta, sin_phi * sin_theta, cos_theta);
        }
        V3 wi_p = normalize(V3(m_alpha_u * wi.x, m_alpha_v * wi.y, wi.z));
        // Frame3f::sincos_phi (frame.h:111-122)
        float sin_theta_2 = fmadd(wi_p.x, wi_p.x, sqr(wi_p.y)), inv_sin_theta = rsqrt(sin_theta_2);
        float rx = wi_p.x * inv_sin_theta, ry = wi_p.y * inv_sin_theta;
        rx = std::fmin(std::fmax(rx, -1.f), 1.f); ry = std::fmin(std::fmax(ry, -1.f), 1.f);
        if (std::fabs(sin_theta_2) <= 4.f * 0x1p-24f) { rx = 1.f; ry = 0.f; }
        float sin_phi = ry, cos_phi = rx, cos_theta = wi_p.z;
        float slope_x, slope_y; sample_visible_11(cos_theta, sx, sy, slope_x, slope_y);
        float s0 = fmsub(cos_phi, slope_x, sin_phi * slope_y) * m_alpha_u, s1 = fmadd(sin_phi, slope_x, cos_phi * slope_y) * m_alpha_v;
        V3 m = normalize(V3(-s0, -s1, 1.f));
        pdf_out = eval(m) * smith_g1(wi, m) * std::fabs(dot(wi, m)) / wi.z;
        return m;
    }

private:
    MicrofacetType m_type; float m_alpha_u, m_alpha_v; bool m_sample_visible;
};

static inline void gauss_legendre(int n, std::vector<float> &nodes, std::vector<float> &weights) {
    nodes.assign(n, 0.f); weights.assign(n, 0.f);
    auto legendre_pd = [](int l, double x) -> std::pair<double, double> {      // math.h:93-120
        double l_cur = 0, d_cur = 0;
        if (l > 1) {
            double l_p_pred = 1, l_pred = x, d_p_pred = 0, d_pred = 1, k0 = 3, k1 = 2, k2 = 1;
            for (int ki = 2; ki <= l; ++ki) {
                l_cur = (k0 * x * l_pred - k2 * l_p_pred) / k1; d_cur = d_p_pred + k0 * l_pred;
                l_p_pred = l_pred; l_pred = l_cur; d_p_pred = d_pred; d_pred = d_cur;
                k2 = k1; k0 += 2; k1 += 1;
            }
        } else if (l == 0) { l_cur = 1; d_cur = 0; } else { l_cur = x; d_cur = 1; }
        return { l_cur, d_cur };
    };
    n--;
    if (n == 0) { nodes[0] = 0; weights[0] = 2; }
    else if (n == 1) { nodes[0] = (float) -std::sqrt(1.0 / 3.0); nodes[1] = -nodes[0]; weights[0] = weights[1] = 1; }
    int m = (n + 1) / 2;
    for (int i = 0; i < m; ++i) {
        double x = -std::cos((double) (2 * i + 1) / (double) (2 * n + 2) * 3.14159265358979323846);
        int it = 0;
        while (true) {
            if (++it > 20) break;
            auto L = legendre_pd(n + 1, x);
            double step = L.first / L.second; x -= step;
            if (std::fabs(step) <= 4 * std::fabs(x) * 0x1p-53) break;
        }
        auto L = legendre_pd(n + 1, x);
        weights[i] = weights[n - i] = (float) (2 / ((1 - x * x) * (L.second * L.second)));
        nodes[i] = (float) x; nodes[n - i] = (float) -x;
    }
    if ((n % 2) == 0) {
        auto L = legendre_pd(n + 1, 0.0);
        weights[n / 2] = (float) (2 / (L.second * L.second)); nodes[n / 2] = 0.f;
    }
}

static inline float eval_reflectance(const MicrofacetDistribution &distr, V3 wi, float eta) {
    int res = eta > 1 ? 32 : 128;
    std::vector<float> nodes, weights; gauss_legendre(res, nodes, weights);
    float result = 0.f;
    for (int j = 0; j < res * res; ++j) {
        float nx = fmadd(nodes[j % res], 0.5f, 0.5f), ny = fmadd(nodes[j / res], 0.5f, 0.5f);
        float wgt = weights[j % res] * weights[j / res];
        float pdf; V3 m = distr.sample(wi, nx, ny, pdf);
        V3 wo = reflect(wi, m);
        float f = fresnel(dot(wi, m), eta).r;
        float smith = distr.smith_g1(wo, m) * f;
        if (wo.z <= 0.f || wi.z <= 0.f) smith = 0.f;
        result += smith * wgt * 0.25f;
    }
    return result;
}
static inline float eval_transmittance(const MicrofacetDistribution &distr, V3 wi, float eta) {
    int res = eta > 1 ? 32 : 128;
    std::vector<float> nodes, weights; gauss_legendre(res, nodes, weights);
    float result = 0.f;
    for (int j = 0; j < res * res; ++j) {
        float nx = fmadd(nodes[j % res], 0.5f, 0.5f), ny = fmadd(nodes[j / res], 0.5f, 0.5f);
        float wgt = weights[j % res] * weights[j / res];
        float pdf; V3 m = distr.sample(wi, nx, ny, pdf);
        FresnelResult fr = fresnel(dot(wi, m), eta);
        V3 wo = refract(wi, m, fr.cos_theta_t, fr.eta_ti);
        float smith = distr.smith_g1(wo, m) * (1.f - fr.r);
        if (wo.z * wi.z >= 0.f) smith = 0.f;
        result += smith * wgt * 0.25f;
    }
    return result;
}

/* fresnel_diffuse_reflectance (include/mitsuba/render/fresnel.h:327-355) */
static inline float fresnel_diffuse_reflectance(float eta) {
    float inv_eta = rcp(eta);
    float approx_1 = fmadd(0.0636f, inv_eta, fmadd(eta, fmadd(eta, -1.4399f, 0.7099f), 0.6681f));
    const float c[6] = { 0.919317f, -3.4793f, 6.75335f, -7.80989f, 4.98554f, -1.36881f };       // dr::horner
    float approx_2 = c[5];
    for (int i = 4; i >= 0; --i) approx_2 = fmadd(approx_2, inv_eta, c[i]);
    return eta < 1.f ? approx_1 : approx_2;
}

/* ------------------------------------------------------------------ plugins
 * types: 0 diffuse, 1 dielectric, 2 roughconductor, 3 roughplastic, 4 conductor (src/bsdfs/conductor.cpp:218-330), 5 plastic (src/bsdfs/plastic.cpp:150-360) */
struct BSDFSample { V3 wo; float pdf = 0.f, eta = 0.f; bool delta = false; };
/* value = f * cos, plus d value / d slot0 (for the hand-derived PRB adjoint of slot-0 colour parameters) */
struct BSDFEval { V3 value, d_slot0; float pdf = 0.f; };

constexpr int kRoughTransmittanceRes = 64;

struct BsdfRecord {
    OrcBSDF p;
    std::vector<float> external_transmittance;      /* roughplastic */
    float inv_eta_2 = 0.f, internal_reflectance = 0.f, specular_sampling_weight = 0.f;
    MicrofacetType mtype() const { return (p.flags & 2u) ? MicrofacetType::GGX : MicrofacetType::Beckmann; }
    bool sample_visible() const { return (p.flags & 4u) != 0; }
    bool nonlinear() const { return (p.flags & 8u) != 0; }
    bool smooth() const { return p.type != 1 && p.type != 4; }      // BSDFFlags::Smooth: `dielectric` (1) and `conductor` (4) only have delta lobes
};

static inline float lerp_gather(const std::vector<float> &data, float x) {
    size_t size = data.size();
    x *= (float) (size - 1);
    uint32_t index = std::min<uint32_t>((uint32_t) x, (uint32_t) (size - 2));
    return lerp(data[index], data[index + 1], x - (float) index);
}

static inline BSDFEval plugin_eval_pdf(const BsdfRecord &b, V3 slot0, V3 slot1, V3 wi, V3 wo) {
    BSDFEval e;
    float cos_theta_i = wi.z, cos_theta_o = wo.z;
    if (b.p.type == 0) {                                   // SmoothDiffuse::eval_pdf
        if (cos_theta_i > 0.f && cos_theta_o > 0.f) {
            e.value = (slot0 * InvPi) * cos_theta_o; e.pdf = InvPi * cos_theta_o; e.d_slot0 = V3(InvPi * cos_theta_o);
        }
    } else if (b.p.type == 2) {                            // RoughConductor::eval_pdf
        V3 H = normalize(wo + wi);
        bool active = cos_theta_i > 0.f && cos_theta_o > 0.f && dot(wi, H) > 0.f && dot(wo, H) > 0.f;
        if (!active) return e;
        MicrofacetDistribution distr(b.mtype(), b.p.alpha_u, b.p.alpha_v, b.sample_visible());
        float D = distr.eval(H);
        active = D != 0.f;
        float smith_g1_wi = distr.smith_g1(wi, H), G = smith_g1_wi * distr.smith_g1(wo, H);
        float value = D * G / (4.f * cos_theta_i);
        float c = dot(wi, H);
        V3 F(fresnel_conductor(c, b.p.eta_c[0], b.p.k_c[0]), fresnel_conductor(c, b.p.eta_c[1], b.p.k_c[1]), fresnel_conductor(c, b.p.eta_c[2], b.p.k_c[2]));
        e.pdf = b.sample_visible() ? D * smith_g1_wi / (4.f * cos_theta_i) : distr.pdf(wi, H) / (4.f * dot(wo, H));
        if (active) { e.d_slot0 = F * value; e.value = F * (slot0 * value); }
    } else if (b.p.type == 3) {                            // RoughPlastic::eval + pdf
        if (!(cos_theta_i > 0.f && cos_theta_o > 0.f)) return e;
        MicrofacetDistribution distr(b.mtype(), b.p.alpha_u, b.p.alpha_u, b.sample_visible());
        V3 H = normalize(wo + wi);
        float D = distr.eval(H);
        float F = fresnel(dot(wi, H), b.p.eta).r;
        float G = distr.G(wi, wo, H);
        float spec = F * D * G / (4.f * cos_theta_i);
        float t_i = lerp_gather(b.external_transmittance, cos_theta_i), t_o = lerp_gather(b.external_transmittance, cos_theta_o);
        V3 den = b.nonlinear() ? V3(1.f) - slot0 * b.internal_reflectance : V3(1.f - b.internal_reflectance);
        V3 diff(slot0.x / den.x, slot0.y / den.y, slot0.z / den.z);
        float k = InvPi * b.inv_eta_2 * cos_theta_o * t_i * t_o;
        e.value = slot1 * spec + diff * k;
        e.d_slot0 = b.nonlinear() ? V3(k / (den.x * den.x), k / (den.y * den.y), k / (den.z * den.z)) : V3(k / den.x, k / den.y, k / den.z);
        float prob_specular = (1.f - t_i) * b.specular_sampling_weight, prob_diffuse = t_i * (1.f - b.specular_sampling_weight);
        prob_specular = prob_specular / (prob_specular + prob_diffuse); prob_diffuse = 1.f - prob_specular;
        float result = b.sample_visible() ? D * distr.smith_g1(wi, H) / (4.f * cos_theta_i) : distr.pdf(wi, H) / (4.f * dot(wo, H));
        result *= prob_specular;
        result += prob_diffuse * (InvPi * cos_theta_o);     // warp::square_to_cosine_hemisphere_pdf
        e.pdf = result;
    }
    else if (b.p.type == 5) {                              // SmoothPlastic::eval_pdf (plastic.cpp:318-352); the delta lobe evaluates to zero
        if (!(cos_theta_i > 0.f && cos_theta_o > 0.f)) return e;
        float f_i = fresnel(cos_theta_i, b.p.eta).r, f_o = fresnel(cos_theta_o, b.p.eta).r;
        V3 den = b.nonlinear() ? V3(1.f) - slot0 * b.internal_reflectance : V3(1.f - b.internal_reflectance);
        V3 diff(slot0.x / den.x, slot0.y / den.y, slot0.z / den.z);
        float hemi_pdf = InvPi * cos_theta_o;
        float k = hemi_pdf * b.inv_eta_2 * (1.f - f_i) * (1.f - f_o);
        e.value = diff * k;
        e.d_slot0 = b.nonlinear() ? V3(k / (den.x * den.x), k / (den.y * den.y), k / (den.z * den.z)) : V3(k / den.x, k / den.y, k / den.z);
        float prob_specular = f_i * b.specular_sampling_weight, prob_diffuse = (1.f - f_i) * (1.f - b.specular_sampling_weight);
        prob_diffuse = prob_diffuse / (prob_specular + prob_diffuse);
        e.pdf = hemi_pdf * prob_diffuse;
    }
    return e;                                              // SmoothDielectric, SmoothConductor: eval = pdf = 0
}

static inline BSDFSample plugin_sample(const BsdfRecord &b, V3 slot0, V3 slot1, V3 wi, float sample1, float s2x, float s2y, V3 &weight) {
    BSDFSample bs; weight = V3(0.f);
    float cos_theta_i = wi.z;
    if (b.p.type == 0) {
        bs.wo = square_to_cosine_hemisphere(s2x, s2y); bs.pdf = InvPi * bs.wo.z; bs.eta = 1.f;
        if (cos_theta_i > 0.f && bs.pdf > 0.f) weight = slot0;
    } else if (b.p.type == 1) {                            // SmoothDielectric::sample, Radiance mode, both lobes enabled
        FresnelResult fr = fresnel(cos_theta_i, b.p.eta);
        float r_i = fr.r, t_i = 1.f - r_i;
        bool selected_r = sample1 <= r_i;
        bs.pdf = selected_r ? r_i : t_i; bs.delta = true;
        bs.wo = selected_r ? reflect(wi) : refract(wi, fr.cos_theta_t, fr.eta_ti);
        bs.eta = selected_r ? 1.f : fr.eta_it;
        weight = selected_r ? slot0 : slot1 * sqr(fr.eta_ti);
    } else if (b.p.type == 2) {
        if (!(cos_theta_i > 0.f)) return bs;
        MicrofacetDistribution distr(b.mtype(), b.p.alpha_u, b.p.alpha_v, b.sample_visible());
        V3 m = distr.sample(wi, s2x, s2y, bs.pdf);
        bs.wo = reflect(wi, m); bs.eta = 1.f;
        bool active = (bs.pdf != 0.f) && bs.wo.z > 0.f;
        float w = b.sample_visible() ? distr.smith_g1(bs.wo, m) : distr.G(wi, bs.wo, m) * dot(wi, m) / (cos_theta_i * m.z);
        bs.pdf /= 4.f * dot(bs.wo, m);
        float c = dot(wi, m);
        V3 F(fresnel_conductor(c, b.p.eta_c[0], b.p.k_c[0]), fresnel_conductor(c, b.p.eta_c[1], b.p.k_c[1]), fresnel_conductor(c, b.p.eta_c[2], b.p.k_c[2]));
        if (active) weight = F * (slot0 * w);
    } else if (b.p.type == 4) {                            // SmoothConductor::sample
        if (!(cos_theta_i > 0.f)) return bs;
        bs.wo = reflect(wi); bs.eta = 1.f; bs.pdf = 1.f; bs.delta = true;
        V3 F(fresnel_conductor(cos_theta_i, b.p.eta_c[0], b.p.k_c[0]), fresnel_conductor(cos_theta_i, b.p.eta_c[1], b.p.k_c[1]), fresnel_conductor(cos_theta_i, b.p.eta_c[2], b.p.k_c[2]));
        weight = slot0 * F;
    } else if (b.p.type == 5) {                            // SmoothPlastic::sample (plastic.cpp:208-266), both components enabled
        if (!(cos_theta_i > 0.f)) return bs;
        float f_i = fresnel(cos_theta_i, b.p.eta).r;
        float prob_specular = f_i * b.specular_sampling_weight, prob_diffuse = (1.f - f_i) * (1.f - b.specular_sampling_weight);
        prob_specular = prob_specular / (prob_specular + prob_diffuse);
        prob_diffuse = 1.f - prob_specular;
        bs.eta = 1.f;
        if (sample1 < prob_specular) {
            bs.wo = reflect(wi); bs.pdf = prob_specular; bs.delta = true;
            weight = slot1 * (f_i / bs.pdf);
        } else {
            bs.wo = square_to_cosine_hemisphere(s2x, s2y);
            bs.pdf = prob_diffuse * (InvPi * bs.wo.z);
            float f_o = fresnel(bs.wo.z, b.p.eta).r;
            V3 den = b.nonlinear() ? V3(1.f) - slot0 * b.internal_reflectance : V3(1.f - b.internal_reflectance);
            V3 value(slot0.x / den.x, slot0.y / den.y, slot0.z / den.z);
            weight = value * (b.inv_eta_2 * (1.f - f_i) * (1.f - f_o) / prob_diffuse);
        }
    } else {
        if (!(cos_theta_i > 0.f)) return bs;
        float t_i = lerp_gather(b.external_transmittance, cos_theta_i);
        float prob_specular = (1.f - t_i) * b.specular_sampling_weight, prob_diffuse = t_i * (1.f - b.specular_sampling_weight);
        prob_specular = prob_specular / (prob_specular + prob_diffuse);
        bool sample_specular = sample1 < prob_specular;
        bs.eta = 1.f;
        if (sample_specular) {
            MicrofacetDistribution distr(b.mtype(), b.p.alpha_u, b.p.alpha_u, b.sample_visible());
            float tmp; V3 m = distr.sample(wi, s2x, s2y, tmp);
            bs.wo = reflect(wi, m);
        } else bs.wo = square_to_cosine_hemisphere(s2x, s2y);
        BSDFEval e = plugin_eval_pdf(b, slot0, slot1, wi, bs.wo);
        bs.pdf = e.pdf;
        if (bs.pdf > 0.f) weight = V3(e.value.x / bs.pdf, e.value.y / bs.pdf, e.value.z / bs.pdf);
    }
    return bs;
}

/* SmoothPlastic::parameters_changed (plastic.cpp:188-205) */
static inline void plastic_precompute(BsdfRecord &b, float d_mean) {
    b.inv_eta_2 = 1.f / (b.p.eta * b.p.eta);
    b.internal_reflectance = fresnel_diffuse_reflectance(1.f / b.p.eta);          // m_fdr_int
    float s_mean = (b.p.reflectance2[0] + b.p.reflectance2[1] + b.p.reflectance2[2]) / 3.f;
    b.specular_sampling_weight = s_mean / (d_mean + s_mean);
}

/* RoughPlastic::parameters_changed */
static inline void roughplastic_precompute(BsdfRecord &b, float d_mean) {
    b.inv_eta_2 = 1.f / (b.p.eta * b.p.eta);
    float s_mean = (b.p.reflectance2[0] + b.p.reflectance2[1] + b.p.reflectance2[2]) / 3.f;
    b.specular_sampling_weight = s_mean / (d_mean + s_mean);
    if (!b.external_transmittance.empty()) return;
    MicrofacetDistribution distr(b.mtype(), b.p.alpha_u, b.p.alpha_u, true);
    double acc = 0;
    for (int i = 0; i < kRoughTransmittanceRes; ++i) {
        float mu = std::fmax(1e-6f, (float) i / (float) (kRoughTransmittanceRes - 1));
        V3 wi(std::sqrt(1.f - mu * mu), 0.f, mu);
        b.external_transmittance.push_back(eval_transmittance(distr, wi, b.p.eta));
        acc += (double) (eval_reflectance(distr, wi, 1.f / b.p.eta) * wi.z);
    }
    b.internal_reflectance = (float) (acc / kRoughTransmittanceRes) * 2.f;
}

} // namespace orc
