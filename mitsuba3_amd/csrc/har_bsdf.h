/*
 * har_bsdf.h -- BSDF models of the hip_ad_rgb path (HAR_HD: HIP kernels + host test harness).
 *
 *   type 0  diffuse          src/bsdfs/diffuse.cpp:100-179
 *   type 1  dielectric       src/bsdfs/dielectric.cpp:245-353           (delta reflection / transmission)
 *   type 2  roughconductor   src/bsdfs/roughconductor.cpp:226-520       (Beckmann / GGX microfacet, conductor Fresnel)
 *   type 3  roughplastic     src/bsdfs/roughplastic.cpp:244-420         (rough dielectric coating over a diffuse base)
 *   flag    twosided         src/bsdfs/twosided.cpp:112-270             (one nested BSDF, or a different one on the back)
 *
 * plus include/mitsuba/render/fresnel.h:35-91 (fresnel), :93-116 (fresnel_conductor), :276-313 (reflect / refract)
 * and  include/mitsuba/render/microfacet.h:185-421 (MicrofacetDistribution eval / pdf / sample / smith_g1).
 *
 * Every model returns f * cos(theta_o) like the reference.  `d_slot0` / `d_slot1` of an evaluation are the
 * per-channel derivatives of that value with respect to the BSDF's two colour parameters ("slots", see DBsdf),
 * which is all the hand-derived PRB adjoint needs (SURVEY.md App. B generalised): delta lobes evaluate to zero,
 * so -- exactly as in prb.py:288-297 -- they carry no parameter gradient.
 */
#pragma once
#include "har_math.h"

namespace har {

enum { BSDF_DIFFUSE = 0, BSDF_DIELECTRIC = 1, BSDF_ROUGHCONDUCTOR = 2, BSDF_ROUGHPLASTIC = 3, BSDF_CONDUCTOR = 4, BSDF_PLASTIC = 5, BSDF_TYPE_COUNT = 6 };
enum { BF_TWOSIDED = 1u, BF_GGX = 2u, BF_SAMPLE_VISIBLE = 4u, BF_NONLINEAR = 8u };
#define HAR_ROUGH_TRANSMITTANCE_RES 64          /* MI_ROUGH_TRANSMITTANCE_RES, include/mitsuba/render/microfacet.h */

/* colour parameter slots: slot 0 = diffuse.reflectance | roughconductor.specular_reflectance |
 * roughplastic.diffuse_reflectance | dielectric.specular_reflectance (may be a bitmap texture);
 * slot 1 = roughplastic.specular_reflectance | dielectric.specular_transmittance (constant) */
struct DBsdf {
    uint32_t type; int32_t texture; float r, g, b;
    uint32_t flags;
    float r2, g2, b2;
    float alpha_u, alpha_v, eta;
    float eta_c[3], k_c[3];
    int32_t back;                 /* twosided: BSDF record of the back side (-1: the same record) */
    int32_t table;                /* roughplastic: offset of its 64-entry external transmittance table, else -1 */
    float inv_eta_2, internal_reflectance, spec_sampling_weight;
    uint32_t pad;
};
static_assert(sizeof(DBsdf) == 96, "DBsdf layout");

struct BsdfEval { Vec3 value; float pdf; Vec3 d_slot0, d_slot1; };
struct BsdfSample { Vec3 wo; float pdf; Vec3 weight; float eta; bool delta; uint32_t type, comp; };     /* type / comp: BSDFSample3f::sampled_type / sampled_component (bsdf.h:212-216) */

/* BSDFFlags of the lobes these models have (include/mitsuba/render/bsdf.h:31-80) */
enum { LOBE_DIFFUSE_REFLECTION = 0x2u, LOBE_GLOSSY_REFLECTION = 0x8u, LOBE_DELTA_REFLECTION = 0x20u, LOBE_DELTA_TRANSMISSION = 0x40u };
/* BSDFContext (include/mitsuba/render/bsdf.h:140-186): transport mode (0 = Radiance, 1 = Importance), type mask, component index.  The wavefront kernels
 * run the default context -- the `CTX = false` instantiations below, in which every test of it folds away; the array-valued plugin surface
 * (har_bsdf_eval / _pdf / _eval_pdf / _sample) passes the caller's. */
struct BsdfCtx {
    uint32_t mode = 0u, type_mask = 0x1ffu, component = 0xffffffffu;
    HAR_HD bool is_enabled(uint32_t type, uint32_t comp = 0u) const {                   /* bsdf.h:177-181 */
        return (type_mask == 0xffffffffu || (type_mask & type) == type) && (component == 0xffffffffu || component == comp);
    }
};
/* BSDF::component_count() of a (not twosided) record: diffuse 1, dielectric 2 (reflection, transmission), roughconductor 1, roughplastic 2 (glossy, diffuse),
 * conductor 1, plastic 2 (delta reflection, diffuse) */
HAR_HD uint32_t bsdf_component_count(uint32_t type) { return (type == 1u || type == 3u || type == 5u) ? 2u : 1u; }

HAR_HD float safe_sqrt_(float x) { return sqrtf(fmaxf(x, 0.f)); }
HAR_HD float lerp_(float a, float b, float t) { return fma_(b, t, fnma_(a, t, a)); }        /* dr::lerp */

/* fresnel(), fresnel.h:35-91 */
HAR_HD void fresnel_dielectric(float cos_theta_i, float eta, float &r, float &cos_theta_t, float &eta_it, float &eta_ti) {
    bool outside = cos_theta_i >= 0.f;
    float rcp_eta = rcp_(eta);
    eta_it = outside ? eta : rcp_eta; eta_ti = outside ? rcp_eta : eta;
    float cos_theta_t_sqr = fnma_(fnma_(cos_theta_i, cos_theta_i, 1.f), eta_ti * eta_ti, 1.f);
    float cos_theta_i_abs = fabsf(cos_theta_i), cos_theta_t_abs = safe_sqrt_(cos_theta_t_sqr);
    bool index_matched = eta == 1.f, special_case = index_matched || cos_theta_i_abs == 0.f;
    float r_sc = index_matched ? 0.f : 1.f;
    float a_s = fnma_(eta_it, cos_theta_t_abs, cos_theta_i_abs) / fma_(eta_it, cos_theta_t_abs, cos_theta_i_abs);
    float a_p = fnma_(eta_it, cos_theta_i_abs, cos_theta_t_abs) / fma_(eta_it, cos_theta_i_abs, cos_theta_t_abs);
    r = 0.5f * (sqr_(a_s) + sqr_(a_p));
    if (special_case) r = r_sc;
    cos_theta_t = mulsign_neg_(cos_theta_t_abs, cos_theta_i);
}

/* fresnel_conductor(), fresnel.h:93-116 */
HAR_HD float fresnel_conductor(float cos_theta_i, float eta_r, float eta_i) {
    float cos_theta_i_2 = cos_theta_i * cos_theta_i, sin_theta_i_2 = 1.f - cos_theta_i_2, sin_theta_i_4 = sin_theta_i_2 * sin_theta_i_2;
    float temp_1 = eta_r * eta_r - eta_i * eta_i - sin_theta_i_2;
    float a_2_pb_2 = safe_sqrt_(temp_1 * temp_1 + 4.f * eta_i * eta_i * eta_r * eta_r);
    float a = safe_sqrt_(.5f * (a_2_pb_2 + temp_1));
    float term_1 = a_2_pb_2 + cos_theta_i_2, term_2 = 2.f * cos_theta_i * a;
    float r_s = (term_1 - term_2) / (term_1 + term_2);
    float term_3 = a_2_pb_2 * cos_theta_i_2 + sin_theta_i_4, term_4 = term_2 * sin_theta_i_2;
    float r_p = r_s * (term_3 - term_4) / (term_3 + term_4);
    return 0.5f * (r_s + r_p);
}

/* fresnel_conductor and its derivatives w.r.t. eta (real part) and k (imaginary part): forward-mode chain rule through the function above */
HAR_HD float fresnel_conductor_grad(float cos_theta_i, float eta_r, float eta_i, float &dF_deta, float &dF_dk) {
    const float c2 = cos_theta_i * cos_theta_i, s2 = 1.f - c2, s4 = s2 * s2;
    const float temp_1 = eta_r * eta_r - eta_i * eta_i - s2;
    const float rad = temp_1 * temp_1 + 4.f * eta_i * eta_i * eta_r * eta_r;
    const float ab = safe_sqrt_(rad);                                   /* a^2 + b^2 */
    const float a = safe_sqrt_(.5f * (ab + temp_1));
    const float term_1 = ab + c2, term_2 = 2.f * cos_theta_i * a;
    const float r_s = (term_1 - term_2) / (term_1 + term_2);
    const float term_3 = ab * c2 + s4, term_4 = term_2 * s2;
    const float q = (term_3 - term_4) / (term_3 + term_4), r_p = r_s * q;
    float out[2];
    for (int k = 0; k < 2; ++k) {                                       /* k = 0: d / d eta_r, 1: d / d eta_i */
        const float d_temp_1 = k == 0 ? 2.f * eta_r : -2.f * eta_i;
        const float d_rad = 2.f * temp_1 * d_temp_1 + (k == 0 ? 8.f * eta_i * eta_i * eta_r : 8.f * eta_i * eta_r * eta_r);
        const float d_ab = ab > 0.f ? .5f * d_rad / ab : 0.f;
        const float d_a = a > 0.f ? .25f * (d_ab + d_temp_1) / a : 0.f;
        const float d_t1 = d_ab, d_t2 = 2.f * cos_theta_i * d_a;
        const float d_rs = ((d_t1 - d_t2) * (term_1 + term_2) - (term_1 - term_2) * (d_t1 + d_t2)) / sqr_(term_1 + term_2);
        const float d_t3 = d_ab * c2, d_t4 = d_t2 * s2;
        const float d_q = ((d_t3 - d_t4) * (term_3 + term_4) - (term_3 - term_4) * (d_t3 + d_t4)) / sqr_(term_3 + term_4);
        out[k] = .5f * (d_rs + d_rs * q + r_s * d_q);
    }
    dF_deta = out[0]; dF_dk = out[1];
    return .5f * (r_s + r_p);
}

HAR_HD Vec3 reflect_local(Vec3 wi) { return Vec3(-wi.x, -wi.y, wi.z); }                                        /* fresnel.h:276 */
HAR_HD Vec3 reflect_m(Vec3 wi, Vec3 m) { float k = 2.f * dot3(wi, m); return Vec3(fms_(m.x, k, wi.x), fms_(m.y, k, wi.y), fms_(m.z, k, wi.z)); }   /* :282 */
HAR_HD Vec3 refract_local(Vec3 wi, float cos_theta_t, float eta_ti) { return Vec3(-eta_ti * wi.x, -eta_ti * wi.y, cos_theta_t); }                   /* :293 */
HAR_HD Vec3 refract_m(Vec3 wi, Vec3 m, float cos_theta_t, float eta_ti) {                                                                            /* :311 */
    float k = fma_(dot3(wi, m), eta_ti, cos_theta_t);
    return Vec3(fms_(m.x, k, wi.x * eta_ti), fms_(m.y, k, wi.y * eta_ti), fms_(m.z, k, wi.z * eta_ti));
}

/* erfinv, single precision (M. Giles, "Approximating the erfinv function"); dr::erfinv is NOT IN TREE (parity unpinned) */
HAR_HD float erfinv_(float x) {
    float w = -log_((1.f - x) * (1.f + x)), p;
    if (w < 5.f) {
        w = w - 2.5f;
        p = 2.81022636e-08f; p = fma_(p, w, 3.43273939e-07f); p = fma_(p, w, -3.5233877e-06f); p = fma_(p, w, -4.39150654e-06f);
        p = fma_(p, w, 0.00021858087f); p = fma_(p, w, -0.00125372503f); p = fma_(p, w, -0.00417768164f); p = fma_(p, w, 0.246640727f);
        p = fma_(p, w, 1.50140941f);
    } else {
        w = sqrtf(w) - 3.f;
        p = -0.000200214257f; p = fma_(p, w, 0.000100950558f); p = fma_(p, w, 0.00134934322f); p = fma_(p, w, -0.00367342844f);
        p = fma_(p, w, 0.00573950773f); p = fma_(p, w, -0.0076224613f); p = fma_(p, w, 0.00943887047f); p = fma_(p, w, 1.00167406f);
        p = fma_(p, w, 2.83297682f);
    }
    return p * x;
}

/* MicrofacetDistribution, microfacet.h:64-447 */
struct Microfacet {
    bool ggx, sample_visible; float alpha_u, alpha_v;
    HAR_HD Microfacet(bool g, float au, float av, bool sv) : ggx(g), sample_visible(sv), alpha_u(fmaxf(au, 1e-4f)), alpha_v(fmaxf(av, 1e-4f)) {}

    HAR_HD float eval(Vec3 m) const {                                                  /* :185-207 */
        float alpha_uv = alpha_u * alpha_v, cos_theta = m.z, cos_theta_2 = sqr_(cos_theta), result;
        if (!ggx) result = exp_(-(sqr_(m.x / alpha_u) + sqr_(m.y / alpha_v)) / cos_theta_2) / (HAR_PI * alpha_uv * sqr_(cos_theta_2));
        else      result = rcp_(HAR_PI * alpha_uv * sqr_(sqr_(m.x / alpha_u) + sqr_(m.y / alpha_v) + sqr_(m.z)));
        return result * cos_theta > 1e-20f ? result : 0.f;
    }
    HAR_HD float smith_g1(Vec3 v, Vec3 m) const {                                      /* :341-365 */
        float xy_alpha_2 = sqr_(alpha_u * v.x) + sqr_(alpha_v * v.y), tan_theta_alpha_2 = xy_alpha_2 / sqr_(v.z), result;
        if (!ggx) {
            float a = rsqrt_(tan_theta_alpha_2), a_sqr = sqr_(a);
            result = a >= 1.6f ? 1.f : (3.535f * a + 2.181f * a_sqr) / (1.f + 2.276f * a + 2.577f * a_sqr);
        } else result = 2.f / (1.f + sqrtf(1.f + tan_theta_alpha_2));
        if (xy_alpha_2 == 0.f) result = 1.f;
        if (dot3(v, m) * v.z <= 0.f) result = 0.f;
        return result;
    }
    HAR_HD float G(Vec3 wi, Vec3 wo, Vec3 m) const { return smith_g1(wi, m) * smith_g1(wo, m); }
    /* d ln D(m) / d alpha_u, d alpha_v (hand-derived from eval() above; zero where eval() clamps D to zero or alpha to 1e-4):
     *   Beckmann: ln D = -(x^2/au^2 + y^2/av^2)/c^2 - ln(pi au av c^4);   GGX: ln D = -ln(pi au av) - 2 ln(x^2/au^2 + y^2/av^2 + z^2) */
    HAR_HD void dlog_eval(Vec3 m, float &du, float &dv) const {
        const float c2 = sqr_(m.z);
        if (!ggx) { du = 2.f * sqr_(m.x) / (alpha_u * alpha_u * alpha_u * c2) - 1.f / alpha_u; dv = 2.f * sqr_(m.y) / (alpha_v * alpha_v * alpha_v * c2) - 1.f / alpha_v; }
        else {
            const float S = sqr_(m.x / alpha_u) + sqr_(m.y / alpha_v) + c2;
            du = 4.f * sqr_(m.x) / (alpha_u * alpha_u * alpha_u * S) - 1.f / alpha_u; dv = 4.f * sqr_(m.y) / (alpha_v * alpha_v * alpha_v * S) - 1.f / alpha_v;
        }
    }
    /* d ln G1(v, m) / d alpha_u, d alpha_v: G1 is a function of t = (au^2 vx^2 + av^2 vy^2) / vz^2 */
    HAR_HD void dlog_smith_g1(Vec3 v, Vec3 m, float &du, float &dv) const {
        du = 0.f; dv = 0.f;
        const float xy_alpha_2 = sqr_(alpha_u * v.x) + sqr_(alpha_v * v.y), inv_z2 = 1.f / sqr_(v.z), t = xy_alpha_2 * inv_z2;
        if (xy_alpha_2 == 0.f || dot3(v, m) * v.z <= 0.f) return;
        float dG_dt, G1;
        if (!ggx) {
            const float a = rsqrt_(t), a2 = a * a;
            if (a >= 1.6f) return;                                                     /* the rational fit is cut off at 1 */
            const float num = 3.535f * a + 2.181f * a2, den = 1.f + 2.276f * a + 2.577f * a2;
            G1 = num / den;
            const float dG_da = ((3.535f + 4.362f * a) * den - num * (2.276f + 5.154f * a)) / (den * den);
            dG_dt = dG_da * (-.5f * a / t);                                            /* a = t^(-1/2) */
        } else {
            const float sq = sqrtf(1.f + t);
            G1 = 2.f / (1.f + sq); dG_dt = -1.f / (sq * sqr_(1.f + sq));
        }
        const float k = dG_dt / G1;
        du = k * 2.f * alpha_u * sqr_(v.x) * inv_z2; dv = k * 2.f * alpha_v * sqr_(v.y) * inv_z2;
    }
    HAR_HD float pdf(Vec3 wi, Vec3 m) const {                                          /* :219-228 */
        float result = eval(m);
        if (sample_visible) result *= smith_g1(wi, m) * fabsf(dot3(wi, m)) / wi.z; else result *= m.z;
        return result;
    }
    HAR_HD void sample_visible_11(float cos_theta_i, float sx, float sy, float &slope_x, float &slope_y) const {   /* :368-421 */
        if (!ggx) {
            float tan_theta_i = safe_sqrt_(fnma_(cos_theta_i, cos_theta_i, 1.f)) / cos_theta_i, cot_theta_i = rcp_(tan_theta_i);
            float maxval = erf_(cot_theta_i);
            sx = fmaxf(fminf(sx, 1.f - 1e-6f), 1e-6f); sy = fmaxf(fminf(sy, 1.f - 1e-6f), 1e-6f);
            float x = maxval - (maxval + 1.f) * erf_(sqrtf(-log_(sx)));
            sx *= 1.f + maxval + 0.56418958354775628695f * tan_theta_i * exp_(-sqr_(cot_theta_i));
            for (int i = 0; i < 3; ++i) {
                float slope = erfinv_(x), value = 1.f + x + 0.56418958354775628695f * tan_theta_i * exp_(-sqr_(slope)) - sx, derivative = 1.f - slope * tan_theta_i;
                x -= value / derivative;
            }
            slope_x = erfinv_(x); slope_y = erfinv_(fms_(2.f, sy, 1.f));
        } else {
            /* warp::square_to_uniform_disk_concentric, warp.h:54-90 */
            float x = fms_(2.f, sx, 1.f), y = fms_(2.f, sy, 1.f);
            bool is_zero = x == 0.f && y == 0.f, q13 = fabsf(x) < fabsf(y);
            float r = q13 ? y : x, rp = q13 ? x : y, phi = 0.25f * HAR_PI * rp / r;
            if (q13) phi = 0.5f * HAR_PI - phi;
            if (is_zero) phi = 0.f;
            float s, c; sincos_(phi, s, c);
            float px = r * c, py = r * s;
            float sc = 0.5f * (1.f + cos_theta_i);
            py = lerp_(safe_sqrt_(1.f - sqr_(px)), py, sc);
            float z = safe_sqrt_(1.f - fma_(py, py, px * px));
            float sin_theta_i = safe_sqrt_(1.f - sqr_(cos_theta_i));
            float norm = rcp_(fma_(sin_theta_i, py, cos_theta_i * z));
            slope_x = fms_(cos_theta_i, py, sin_theta_i * z) * norm; slope_y = px * norm;
        }
    }
    HAR_HD Vec3 sample(Vec3 wi, float sx, float sy, float &pdf_out) const {            /* :244-326 */
        if (!sample_visible) {
            float sin_phi, cos_phi, cos_theta, cos_theta_2, alpha_2;
            if (alpha_u == alpha_v) { sincos_((2.f * HAR_PI) * sy, sin_phi, cos_phi); alpha_2 = alpha_u * alpha_u; }
            else {
                float ratio = alpha_v / alpha_u, tmp = ratio * tan_((2.f * HAR_PI) * sy);
                cos_phi = rsqrt_(fma_(tmp, tmp, 1.f)); cos_phi = mulsign_(cos_phi, fabsf(sy - .5f) - .25f);
                sin_phi = cos_phi * tmp;
                alpha_2 = rcp_(sqr_(cos_phi / alpha_u) + sqr_(sin_phi / alpha_v));
            }
            if (!ggx) {
                cos_theta = rsqrt_(fnma_(alpha_2, log_(1.f - sx), 1.f)); cos_theta_2 = sqr_(cos_theta);
                float cos_theta_3 = fmaxf(cos_theta_2 * cos_theta, 1e-20f);
                pdf_out = (1.f - sx) / (HAR_PI * alpha_u * alpha_v * cos_theta_3);
            } else {
                float tan_theta_m_2 = alpha_2 * sx / (1.f - sx);
                cos_theta = rsqrt_(1.f + tan_theta_m_2); cos_theta_2 = sqr_(cos_theta);
                float temp = 1.f + tan_theta_m_2 / alpha_2, cos_theta_3 = fmaxf(cos_theta_2 * cos_theta, 1e-20f);
                pdf_out = rcp_(HAR_PI * alpha_u * alpha_v * cos_theta_3 * sqr_(temp));
            }
            float sin_theta = sqrtf(1.f - cos_theta_2);
            return Vec3(cos_phi * sin_theta, sin_phi * sin_theta, cos_theta);
        }
        Vec3 wi_p = normalize3(Vec3(alpha_u * wi.x, alpha_v * wi.y, wi.z));
        /* Frame3f::sincos_phi, frame.h:103-117 */
        float sin_theta_2 = fma_(wi_p.x, wi_p.x, wi_p.y * wi_p.y), inv_sin_theta = rsqrt_(sin_theta_2);
        float cos_phi = fminf(fmaxf(wi_p.x * inv_sin_theta, -1.f), 1.f), sin_phi = fminf(fmaxf(wi_p.y * inv_sin_theta, -1.f), 1.f);
        if (fabsf(sin_theta_2) <= 4.f * 5.9604644775390625e-8f) { sin_phi = 0.f; cos_phi = 1.f; }
        float slx, sly; sample_visible_11(wi_p.z, sx, sy, slx, sly);
        float s0 = fms_(cos_phi, slx, sin_phi * sly) * alpha_u, s1 = fma_(sin_phi, slx, cos_phi * sly) * alpha_v;
        Vec3 m = normalize3(Vec3(-s0, -s1, 1.f));
        pdf_out = eval(m) * smith_g1(wi, m) * fabsf(dot3(wi, m)) / wi.z;
        return m;
    }
};

/* lerp_gather, roughplastic.cpp:338-349 */
HAR_HD float lerp_gather(const float *data, float x, uint32_t size) {
    x *= (float) (size - 1);
    uint32_t index = (uint32_t) x; if (index > size - 2) index = size - 2;
    float v0 = data[index], v1 = data[index + 1];
    return lerp_(v0, v1, x - (float) index);
}

struct BsdfInputs { Vec3 slot0, slot1; const float *table; };       /* evaluated colour parameters + roughplastic table */

HAR_HD bool bsdf_is_smooth(const DBsdf &B) { return B.type != BSDF_DIELECTRIC && B.type != BSDF_CONDUCTOR; }          /* BSDFFlags::Smooth */
/* material class of a BSDF record for the per-material shading queues: its model when both sides of the surface evaluate the same model (plain records,
 * `twosided` with one nested BSDF or a same-model pair), HAR_MAT_GENERIC for a `twosided` pair of two different models (the generic kernel serves it) */
#define HAR_MAT_GENERIC 6u
#define HAR_MAT_CLASSES 7u

/* TYPES = bit mask of the BSDF types a scene contains (1 << type): the shading kernels are specialised for
 * diffuse-only scenes, where the other models (and their registers) compile out */
#define HAR_BSDF_ALL_TYPES 0xffu
#define HAR_BSDF_ONLY_DIFFUSE 0x1u
#define HAR_BSDF_CLASSIC_TYPES 0xfu       /* diffuse, dielectric, roughconductor, roughplastic: scenes without `conductor` / `plastic` run kernels without that code */
/* models without microfacet code (diffuse, dielectric, conductor, plastic): the cheap half of a split shading pass (k_shade, SPLIT) */
#define HAR_BSDF_CHEAP_TYPES ((1u << BSDF_DIFFUSE) | (1u << BSDF_DIELECTRIC) | (1u << BSDF_CONDUCTOR) | (1u << BSDF_PLASTIC))
#define HAR_BSDF_HAS(TYPES, T) (((TYPES) & (1u << (T))) != 0u)
/* per-material shading queues (k_classify + one k_shade launch per BSDF model, har_kernels.hip): TYPES = (1 << type) | HAR_BSDF_QUEUED.  With ONE model
 * bit set the `switch (B.type)` of the eval / sample code folds to that model at compile time; the QUEUED bit keeps such a mask distinct from
 * HAR_BSDF_ONLY_DIFFUSE, which additionally promises "no twosided record in the scene" */
#define HAR_BSDF_QUEUED 0x200u
#define HAR_BSDF_SINGLE(TYPES) ((((TYPES) & 0xffu) != 0u) && ((((TYPES) & 0xffu) & (((TYPES) & 0xffu) - 1u)) == 0u))
#define HAR_BSDF_SINGLE_TYPE(TYPES) ((((TYPES) & 0xffu) == 1u) ? 0u : (((TYPES) & 0xffu) == 2u) ? 1u : (((TYPES) & 0xffu) == 4u) ? 2u : (((TYPES) & 0xffu) == 8u) ? 3u : (((TYPES) & 0xffu) == 16u) ? 4u : 5u)
/* ... and an area light that radiates a BITMAP (emitter type 7): its texel-distribution code costs the generic kernels registers they do not have (128, spills), so
 * only scenes with such a light run kernels that carry it: TYPES = HAR_BSDF_ALL_TYPES | HAR_SCENE_ENVMAP | HAR_SCENE_TEXLIGHT */
#define HAR_SCENE_TEXLIGHT 0x400u
#define HAR_SCENE_ENVMAP 0x100u           /* the scene has an environment MAP (emitter type 2), a MESH area light (type 3) or a POINT light (type 4): kernels of other scenes compile that code out */

/* fresnel_diffuse_reflectance (include/mitsuba/render/fresnel.h:327-355), evaluated on the host when a `plastic` record is (re)built */
HAR_HD float fresnel_diffuse_reflectance(float eta) {
    float inv_eta = 1.f / eta;
    float approx_1 = fmaf(0.0636f, inv_eta, fmaf(eta, fmaf(eta, -1.4399f, 0.7099f), 0.6681f));
    float approx_2 = fmaf(fmaf(fmaf(fmaf(fmaf(-1.36881f, inv_eta, 4.98554f), inv_eta, -7.80989f), inv_eta, 6.75335f), inv_eta, -3.4793f), inv_eta, 0.919317f);
    return eta < 1.f ? approx_1 : approx_2;
}

/* eval_pdf of one (not twosided) record: value = f * cos(theta_o) */
template <uint32_t TYPES = HAR_BSDF_ALL_TYPES, bool CTX = false>
HAR_HD void bsdf_eval_pdf_one(const DBsdf &B, const BsdfInputs &in, Vec3 wi, Vec3 wo, BsdfEval &e, const BsdfCtx &ctx = BsdfCtx()) {
    e.value = Vec3(0.f); e.pdf = 0.f; e.d_slot0 = Vec3(0.f); e.d_slot1 = Vec3(0.f);
    float cos_theta_i = wi.z, cos_theta_o = wo.z;
    const uint32_t type = HAR_BSDF_SINGLE(TYPES) ? (uint32_t) HAR_BSDF_SINGLE_TYPE(TYPES) : B.type;
    switch (type) {
    case BSDF_DIFFUSE: {                                                     /* diffuse.cpp:159-179 */
        if (CTX && !ctx.is_enabled(LOBE_DIFFUSE_REFLECTION)) return;
        if (!(cos_theta_i > 0.f && cos_theta_o > 0.f)) return;
        float k = HAR_INV_PI * cos_theta_o;
        e.value = (in.slot0 * HAR_INV_PI) * cos_theta_o; e.pdf = k; e.d_slot0 = Vec3(k);
    } break;
    case BSDF_DIELECTRIC: break;                                             /* dielectric.cpp:340-348: delta lobes */
    case BSDF_ROUGHCONDUCTOR: {                                              /* roughconductor.cpp:429-520 */
        if (!HAR_BSDF_HAS(TYPES, BSDF_ROUGHCONDUCTOR)) return;
        if (CTX && !ctx.is_enabled(LOBE_GLOSSY_REFLECTION)) return;
        Vec3 H = normalize3(wo + wi);
        if (!(cos_theta_i > 0.f && cos_theta_o > 0.f && dot3(wi, H) > 0.f && dot3(wo, H) > 0.f)) return;
        Microfacet distr((B.flags & BF_GGX) != 0, B.alpha_u, B.alpha_v, (B.flags & BF_SAMPLE_VISIBLE) != 0);
        float D = distr.eval(H);
        bool active = D != 0.f;
        float smith_g1_wi = distr.smith_g1(wi, H), G = smith_g1_wi * distr.smith_g1(wo, H);
        float value = D * G / (4.f * cos_theta_i);
        float c = dot3(wi, H);
        Vec3 F(fresnel_conductor(c, B.eta_c[0], B.k_c[0]), fresnel_conductor(c, B.eta_c[1], B.k_c[1]), fresnel_conductor(c, B.eta_c[2], B.k_c[2]));
        float pdf = distr.sample_visible ? D * smith_g1_wi / (4.f * cos_theta_i) : distr.pdf(wi, H) / (4.f * dot3(wo, H));
        if (active) { e.d_slot0 = F * value; e.value = F * (in.slot0 * value); }       /* roughconductor.cpp:386-389: (F * (result * reflectance)) */
        e.pdf = pdf;
    } break;
    case BSDF_ROUGHPLASTIC: {                                                /* roughplastic.cpp:296-336 (eval), :351-395 (pdf) */
        if (!HAR_BSDF_HAS(TYPES, BSDF_ROUGHPLASTIC)) return;
        /* component 0 = the glossy coating, 1 = the diffuse base (roughplastic.cpp:301-302) */
        const bool has_specular = !CTX || ctx.is_enabled(LOBE_GLOSSY_REFLECTION, 0u), has_diffuse = !CTX || ctx.is_enabled(LOBE_DIFFUSE_REFLECTION, 1u);
        if (CTX && !has_specular && !has_diffuse) return;
        if (!(cos_theta_i > 0.f && cos_theta_o > 0.f)) return;
        Microfacet distr((B.flags & BF_GGX) != 0, B.alpha_u, B.alpha_u, (B.flags & BF_SAMPLE_VISIBLE) != 0);
        Vec3 H = normalize3(wo + wi);
        float D = distr.eval(H);
        float F, ct, eit, eti; fresnel_dielectric(dot3(wi, H), B.eta, F, ct, eit, eti);
        float G = distr.G(wi, wo, H);
        float spec = F * D * G / (4.f * cos_theta_i);
        float t_i = lerp_gather(in.table, cos_theta_i, HAR_ROUGH_TRANSMITTANCE_RES), t_o = lerp_gather(in.table, cos_theta_o, HAR_ROUGH_TRANSMITTANCE_RES);
        const bool nonlinear = (B.flags & BF_NONLINEAR) != 0;
        Vec3 den = nonlinear ? Vec3(1.f) - in.slot0 * B.internal_reflectance : Vec3(1.f - B.internal_reflectance);
        Vec3 diff(in.slot0.x / den.x, in.slot0.y / den.y, in.slot0.z / den.z);
        float k = HAR_INV_PI * B.inv_eta_2 * cos_theta_o * t_i * t_o;
        e.value = in.slot1 * spec + diff * k;
        e.d_slot1 = Vec3(spec);
        e.d_slot0 = nonlinear ? Vec3(k / (den.x * den.x), k / (den.y * den.y), k / (den.z * den.z)) : Vec3(k / den.x, k / den.y, k / den.z);
        if (CTX && !has_specular) { e.value = diff * k; e.d_slot1 = Vec3(0.f); }                  /* :305-327: the disabled lobe adds nothing */
        if (CTX && !has_diffuse) { e.value = in.slot1 * spec; e.d_slot0 = Vec3(0.f); }
        float prob_specular = (1.f - t_i) * B.spec_sampling_weight, prob_diffuse = t_i * (1.f - B.spec_sampling_weight);
        if (CTX && has_specular != has_diffuse) prob_specular = has_specular ? 1.f : 0.f;         /* :372-375 */
        else prob_specular = prob_specular / (prob_specular + prob_diffuse);
        prob_diffuse = 1.f - prob_specular;
        float result = distr.sample_visible ? D * distr.smith_g1(wi, H) / (4.f * cos_theta_i) : distr.pdf(wi, H) / (4.f * dot3(wo, H));
        result *= prob_specular;
        result += prob_diffuse * (HAR_INV_PI * cos_theta_o);
        e.pdf = result;
    } break;
    case BSDF_CONDUCTOR: break;                                              /* conductor.cpp:306-318: a delta lobe */
    case BSDF_PLASTIC: {                                                     /* plastic.cpp:318-352 (the delta lobe evaluates to zero) */
        if (!HAR_BSDF_HAS(TYPES, BSDF_PLASTIC)) return;
        if (CTX && !ctx.is_enabled(LOBE_DIFFUSE_REFLECTION, 1u)) return;        /* component 1 = the diffuse base, 0 = the delta coating (plastic.cpp:270,322) */
        if (!(cos_theta_i > 0.f && cos_theta_o > 0.f)) return;
        float f_i, f_o, ct, eit, eti;
        fresnel_dielectric(cos_theta_i, B.eta, f_i, ct, eit, eti);
        fresnel_dielectric(cos_theta_o, B.eta, f_o, ct, eit, eti);
        const bool nonlinear = (B.flags & BF_NONLINEAR) != 0;
        Vec3 den = nonlinear ? Vec3(1.f) - in.slot0 * B.internal_reflectance : Vec3(1.f - B.internal_reflectance);
        Vec3 diff(in.slot0.x / den.x, in.slot0.y / den.y, in.slot0.z / den.z);
        float hemi_pdf = HAR_INV_PI * cos_theta_o;
        float k = hemi_pdf * B.inv_eta_2 * (1.f - f_i) * (1.f - f_o);
        e.value = diff * k;
        e.d_slot0 = nonlinear ? Vec3(k / (den.x * den.x), k / (den.y * den.y), k / (den.z * den.z)) : Vec3(k / den.x, k / den.y, k / den.z);
        float prob_specular = f_i * B.spec_sampling_weight, prob_diffuse = (1.f - f_i) * (1.f - B.spec_sampling_weight);
        prob_diffuse = prob_diffuse / (prob_specular + prob_diffuse);
        if (CTX && !ctx.is_enabled(LOBE_DELTA_REFLECTION, 0u)) prob_diffuse = 1.f;               /* :339-346 */
        e.pdf = hemi_pdf * prob_diffuse;
    } break;
    }
}

/* Derivatives of eval()'s value (f * cos theta_o, per channel) with respect to the NON-colour parameters of the rough models -- what the PRB adjoint
 * needs beyond d_slot0 / d_slot1 (prb.py:288-313 with `alpha`, `eta`, `k` attached; roughconductor.cpp:429-520, roughplastic.cpp:296-336):
 *   d_alpha_u / d_alpha_v : every channel scales with D * G, so d value_c = value_spec_c * d ln(D G)
 *   d_eta / d_k           : roughconductor's complex IOR, channel-diagonal through the conductor Fresnel term
 * (roughplastic's `eta` and its transmittance tables are not differentiable in the reference either: ScalarFloat members.) */
struct BsdfEvalExtra { Vec3 d_alpha_u, d_alpha_v, d_eta, d_k; };
HAR_HD void bsdf_eval_extra_one(const DBsdf &B, const BsdfInputs &in, Vec3 wi, Vec3 wo, BsdfEvalExtra &x) {
    x.d_alpha_u = Vec3(0.f); x.d_alpha_v = Vec3(0.f); x.d_eta = Vec3(0.f); x.d_k = Vec3(0.f);
    const float cos_theta_i = wi.z, cos_theta_o = wo.z;
    if (!(cos_theta_i > 0.f && cos_theta_o > 0.f)) return;
    if (B.type == BSDF_ROUGHCONDUCTOR) {
        const Vec3 H = normalize3(wo + wi);
        if (!(dot3(wi, H) > 0.f && dot3(wo, H) > 0.f)) return;
        Microfacet distr((B.flags & BF_GGX) != 0, B.alpha_u, B.alpha_v, (B.flags & BF_SAMPLE_VISIBLE) != 0);
        const float D = distr.eval(H);
        if (D == 0.f) return;
        const float V = D * distr.G(wi, wo, H) / (4.f * cos_theta_i), c = dot3(wi, H);
        float du, dv, gu, gv, hu, hv; distr.dlog_eval(H, du, dv); distr.dlog_smith_g1(wi, H, gu, gv); distr.dlog_smith_g1(wo, H, hu, hv);
        float Fc[3], dE[3], dK[3];
        for (int k = 0; k < 3; ++k) Fc[k] = fresnel_conductor_grad(c, B.eta_c[k], B.k_c[k], dE[k], dK[k]);
        const Vec3 F(Fc[0], Fc[1], Fc[2]), value = (F * V) * in.slot0;
        x.d_alpha_u = value * (du + gu + hu); x.d_alpha_v = value * (dv + gv + hv);
        x.d_eta = (Vec3(dE[0], dE[1], dE[2]) * V) * in.slot0; x.d_k = (Vec3(dK[0], dK[1], dK[2]) * V) * in.slot0;
    } else if (B.type == BSDF_ROUGHPLASTIC) {
        Microfacet distr((B.flags & BF_GGX) != 0, B.alpha_u, B.alpha_u, (B.flags & BF_SAMPLE_VISIBLE) != 0);
        const Vec3 H = normalize3(wo + wi);
        const float D = distr.eval(H);
        if (D == 0.f) return;
        float F, ct, eit, eti; fresnel_dielectric(dot3(wi, H), B.eta, F, ct, eit, eti);
        const float spec = F * D * distr.G(wi, wo, H) / (4.f * cos_theta_i);
        float du, dv, gu, gv, hu, hv; distr.dlog_eval(H, du, dv); distr.dlog_smith_g1(wi, H, gu, gv); distr.dlog_smith_g1(wo, H, hu, hv);
        /* one `alpha` for both axes: d / d alpha = d / d alpha_u + d / d alpha_v, reported in d_alpha_u */
        x.d_alpha_u = (in.slot1 * spec) * ((du + gu + hu) + (dv + gv + hv));
    }
}

/* sample of one (not twosided) record */
template <uint32_t TYPES = HAR_BSDF_ALL_TYPES, bool CTX = false>
HAR_HD void bsdf_sample_one(const DBsdf &B, const BsdfInputs &in, Vec3 wi, float sample1, float s2x, float s2y, BsdfSample &bs, const BsdfCtx &ctx = BsdfCtx()) {
    bs.wo = Vec3(0.f); bs.pdf = 0.f; bs.weight = Vec3(0.f); bs.eta = 0.f; bs.delta = false; bs.type = 0u; bs.comp = 0u;     /* dr::zeros<BSDFSample3f>() */
    float cos_theta_i = wi.z;
    const uint32_t type = HAR_BSDF_SINGLE(TYPES) ? (uint32_t) HAR_BSDF_SINGLE_TYPE(TYPES) : B.type;
    switch (type) {
    case BSDF_DIFFUSE: {                                                     /* diffuse.cpp:100-124 */
        if (CTX && !ctx.is_enabled(LOBE_DIFFUSE_REFLECTION)) return;
        bs.wo = square_to_cosine_hemisphere(s2x, s2y);
        bs.pdf = HAR_INV_PI * bs.wo.z; bs.eta = 1.f; bs.type = LOBE_DIFFUSE_REFLECTION;
        bs.weight = (cos_theta_i > 0.f && bs.pdf > 0.f) ? in.slot0 : Vec3(0.f);
    } break;
    case BSDF_DIELECTRIC: {                                                  /* dielectric.cpp:245-370 */
        /* component 0 = reflection, 1 = transmission; with one of them disabled the other is taken with probability 1 and carries the Fresnel
         * term in its weight (:262-272, 348-350); TransportMode::Importance drops the eta_ti^2 radiance scaling of the transmitted lobe (:362-367) */
        const bool has_reflection = !CTX || ctx.is_enabled(LOBE_DELTA_REFLECTION, 0u), has_transmission = !CTX || ctx.is_enabled(LOBE_DELTA_TRANSMISSION, 1u);
        if (CTX && !has_reflection && !has_transmission) return;
        float r_i, cos_theta_t, eta_it, eta_ti; fresnel_dielectric(cos_theta_i, B.eta, r_i, cos_theta_t, eta_it, eta_ti);
        float t_i = 1.f - r_i;
        const bool both = has_reflection && has_transmission;
        bool selected_r = both ? sample1 <= r_i : has_reflection;
        bs.pdf = both ? (selected_r ? r_i : t_i) : 1.f;
        bs.delta = true; bs.type = selected_r ? LOBE_DELTA_REFLECTION : LOBE_DELTA_TRANSMISSION; bs.comp = selected_r ? 0u : 1u;
        bs.wo = selected_r ? reflect_local(wi) : refract_local(wi, cos_theta_t, eta_ti);
        bs.eta = selected_r ? 1.f : eta_it;
        const float factor = (CTX && ctx.mode != 0u) ? 1.f : eta_ti;
        bs.weight = selected_r ? in.slot0 : in.slot1 * sqr_(factor);
        if (CTX && !both) bs.weight = selected_r ? in.slot0 * r_i : (in.slot1 * t_i) * sqr_(factor);
    } break;
    case BSDF_ROUGHCONDUCTOR: {                                              /* roughconductor.cpp:226-320 */
        if (!HAR_BSDF_HAS(TYPES, BSDF_ROUGHCONDUCTOR)) return;
        if (CTX && !ctx.is_enabled(LOBE_GLOSSY_REFLECTION)) return;
        if (!(cos_theta_i > 0.f)) return;
        Microfacet distr((B.flags & BF_GGX) != 0, B.alpha_u, B.alpha_v, (B.flags & BF_SAMPLE_VISIBLE) != 0);
        float pdf; Vec3 m = distr.sample(wi, s2x, s2y, pdf);
        bs.wo = reflect_m(wi, m); bs.eta = 1.f; bs.pdf = pdf; bs.type = LOBE_GLOSSY_REFLECTION;
        bool active = pdf != 0.f && bs.wo.z > 0.f;
        float weight = distr.sample_visible ? distr.smith_g1(bs.wo, m) : distr.G(wi, bs.wo, m) * dot3(wi, m) / (cos_theta_i * m.z);
        bs.pdf /= 4.f * dot3(bs.wo, m);
        float c = dot3(wi, m);
        Vec3 F(fresnel_conductor(c, B.eta_c[0], B.k_c[0]), fresnel_conductor(c, B.eta_c[1], B.k_c[1]), fresnel_conductor(c, B.eta_c[2], B.k_c[2]));
        bs.weight = active ? F * (in.slot0 * weight) : Vec3(0.f);
    } break;
    case BSDF_ROUGHPLASTIC: {                                                /* roughplastic.cpp:244-294 */
        if (!HAR_BSDF_HAS(TYPES, BSDF_ROUGHPLASTIC)) return;
        const bool has_specular = !CTX || ctx.is_enabled(LOBE_GLOSSY_REFLECTION, 0u), has_diffuse = !CTX || ctx.is_enabled(LOBE_DIFFUSE_REFLECTION, 1u);
        if (CTX && !has_specular && !has_diffuse) return;
        if (!(cos_theta_i > 0.f)) return;
        float t_i = lerp_gather(in.table, cos_theta_i, HAR_ROUGH_TRANSMITTANCE_RES);
        float prob_specular = (1.f - t_i) * B.spec_sampling_weight, prob_diffuse = t_i * (1.f - B.spec_sampling_weight);
        if (CTX && has_specular != has_diffuse) prob_specular = has_specular ? 1.f : 0.f;         /* :261-264 */
        else prob_specular = prob_specular / (prob_specular + prob_diffuse);
        bool sample_specular = sample1 < prob_specular;
        bs.eta = 1.f;
        if (sample_specular) {
            Microfacet distr((B.flags & BF_GGX) != 0, B.alpha_u, B.alpha_u, (B.flags & BF_SAMPLE_VISIBLE) != 0);
            float pdf_m; Vec3 m = distr.sample(wi, s2x, s2y, pdf_m);
            bs.wo = reflect_m(wi, m); bs.type = LOBE_GLOSSY_REFLECTION; bs.comp = 0u;
        } else { bs.wo = square_to_cosine_hemisphere(s2x, s2y); bs.type = LOBE_DIFFUSE_REFLECTION; bs.comp = 1u; }
        BsdfEval e; bsdf_eval_pdf_one<TYPES, CTX>(B, in, wi, bs.wo, e, ctx);
        bs.pdf = e.pdf;
        bool active = bs.pdf > 0.f;
        bs.weight = active ? Vec3(e.value.x / bs.pdf, e.value.y / bs.pdf, e.value.z / bs.pdf) : Vec3(0.f);
    } break;
    case BSDF_CONDUCTOR: {                                                   /* conductor.cpp:264-304 */
        if (!HAR_BSDF_HAS(TYPES, BSDF_CONDUCTOR)) return;
        if (CTX && !ctx.is_enabled(LOBE_DELTA_REFLECTION)) return;
        if (!(cos_theta_i > 0.f)) return;
        bs.wo = reflect_local(wi); bs.eta = 1.f; bs.pdf = 1.f; bs.delta = true; bs.type = LOBE_DELTA_REFLECTION;
        Vec3 F(fresnel_conductor(cos_theta_i, B.eta_c[0], B.k_c[0]), fresnel_conductor(cos_theta_i, B.eta_c[1], B.k_c[1]), fresnel_conductor(cos_theta_i, B.eta_c[2], B.k_c[2]));
        bs.weight = in.slot0 * F;
    } break;
    case BSDF_PLASTIC: {                                                     /* plastic.cpp:208-266 */
        if (!HAR_BSDF_HAS(TYPES, BSDF_PLASTIC)) return;
        const bool has_specular = !CTX || ctx.is_enabled(LOBE_DELTA_REFLECTION, 0u), has_diffuse = !CTX || ctx.is_enabled(LOBE_DIFFUSE_REFLECTION, 1u);
        if (CTX && !has_specular && !has_diffuse) return;
        if (!(cos_theta_i > 0.f)) return;
        float f_i, ct, eit, eti; fresnel_dielectric(cos_theta_i, B.eta, f_i, ct, eit, eti);
        float prob_specular = f_i * B.spec_sampling_weight, prob_diffuse = (1.f - f_i) * (1.f - B.spec_sampling_weight);
        if (CTX && has_specular != has_diffuse) prob_specular = has_specular ? 1.f : 0.f;         /* :231-234 */
        else prob_specular = prob_specular / (prob_specular + prob_diffuse);
        prob_diffuse = 1.f - prob_specular;
        bs.eta = 1.f;
        if (sample1 < prob_specular) {
            bs.wo = reflect_local(wi); bs.pdf = prob_specular; bs.delta = true; bs.type = LOBE_DELTA_REFLECTION; bs.comp = 0u;
            bs.weight = in.slot1 * (f_i / bs.pdf);
        } else {
            bs.wo = square_to_cosine_hemisphere(s2x, s2y); bs.type = LOBE_DIFFUSE_REFLECTION; bs.comp = 1u;
            bs.pdf = prob_diffuse * (HAR_INV_PI * bs.wo.z);
            float f_o; fresnel_dielectric(bs.wo.z, B.eta, f_o, ct, eit, eti);
            const bool nonlinear = (B.flags & BF_NONLINEAR) != 0;
            Vec3 den = nonlinear ? Vec3(1.f) - in.slot0 * B.internal_reflectance : Vec3(1.f - B.internal_reflectance);
            Vec3 value(in.slot0.x / den.x, in.slot0.y / den.y, in.slot0.z / den.z);
            bs.weight = value * (B.inv_eta_2 * (1.f - f_i) * (1.f - f_o) / prob_diffuse);
        }
    } break;
    }
}

} // namespace har
