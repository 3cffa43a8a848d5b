/*
 * har_bsdf_dir.h -- directional derivatives of BSDF::eval for the vertex-position gradients of the PRB adjoint (HAR_HD: device kernel + host harness).
 *
 * With moving geometry the reference evaluates `bsdf.eval(ctx, si, wo)` on an attached `si.wi` and an attached `wo` (prb.py:128-140, 276-288), so the
 * adjoint needs d value / d wi and d value / d wo (local frame, 3 + 3 numbers) of every model with a non-delta lobe:
 *   diffuse.cpp:159-179, roughconductor.cpp:429-520, roughplastic.cpp:296-336, plastic.cpp:318-352 (microfacet.h:185-207,341-365, fresnel.h:35-116).
 * A vertex needs them for ONE scalar only -- F = sum_c A_c value_c(wi, wo), with the channel weights A_c the caller folds in (dL x throughput for emitter
 * sampling, dL x L / value for the continuation) -- so the models are written once more here over a 6-wide forward dual (value + 6 partials, fp32):
 * 7 floats per intermediate instead of 21, no per-model derivative formulas to get wrong, and every dr::select takes the branch the value takes, as the
 * reference's AD does.  The oracle (oracle/mi_oracle.cpp bsdf_dir_grad_fd) differentiates a double-precision restatement numerically instead.
 * Only the shape-gradient kernel (k_shape_adjoint) instantiates this; the rendering kernels never see it.
 */
#pragma once
#include "har_bsdf.h"

namespace har {

struct D6 {
    float v, d[6];
    HAR_HD D6() {}
    HAR_HD D6(float c) : v(c) { for (int i = 0; i < 6; ++i) d[i] = 0.f; }
    HAR_HD static D6 input(float value, int slot) { D6 r(value); r.d[slot] = 1.f; return r; }
};
HAR_HD D6 operator+(const D6 &a, const D6 &b) { D6 r; r.v = a.v + b.v; for (int i = 0; i < 6; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
HAR_HD D6 operator-(const D6 &a, const D6 &b) { D6 r; r.v = a.v - b.v; for (int i = 0; i < 6; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
HAR_HD D6 operator-(const D6 &a) { D6 r; r.v = -a.v; for (int i = 0; i < 6; ++i) r.d[i] = -a.d[i]; return r; }
HAR_HD D6 operator*(const D6 &a, const D6 &b) { D6 r; r.v = a.v * b.v; for (int i = 0; i < 6; ++i) r.d[i] = fma_(a.d[i], b.v, a.v * b.d[i]); return r; }
HAR_HD D6 operator*(const D6 &a, float s) { D6 r; r.v = a.v * s; for (int i = 0; i < 6; ++i) r.d[i] = a.d[i] * s; return r; }
HAR_HD D6 operator+(const D6 &a, float s) { D6 r = a; r.v += s; return r; }
HAR_HD D6 operator/(const D6 &a, const D6 &b) { D6 r; const float ib = 1.f / b.v; r.v = a.v * ib; for (int i = 0; i < 6; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * ib; return r; }
HAR_HD D6 d6_rcp(const D6 &a) { D6 r; r.v = 1.f / a.v; const float k = -r.v * r.v; for (int i = 0; i < 6; ++i) r.d[i] = a.d[i] * k; return r; }
HAR_HD D6 d6_sqr(const D6 &a) { D6 r; r.v = a.v * a.v; const float k = 2.f * a.v; for (int i = 0; i < 6; ++i) r.d[i] = a.d[i] * k; return r; }
/* dr::safe_sqrt: sqrt(max(x, 0)); the clamped branch has a zero derivative */
HAR_HD D6 d6_safe_sqrt(const D6 &a) {
    D6 r; r.v = sqrtf(fmaxf(a.v, 0.f)); const float k = (a.v > 0.f && r.v > 0.f) ? .5f / r.v : 0.f;
    for (int i = 0; i < 6; ++i) r.d[i] = a.d[i] * k;
    return r;
}
HAR_HD D6 d6_exp(const D6 &a) { D6 r; r.v = exp_(a.v); for (int i = 0; i < 6; ++i) r.d[i] = a.d[i] * r.v; return r; }
HAR_HD D6 d6_abs(const D6 &a) { return a.v < 0.f ? -a : a; }

struct D6Vec { D6 x, y, z; };
HAR_HD D6 d6_dot(const D6Vec &a, const D6Vec &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

/* fresnel(cos_theta_i, eta).r, fresnel.h:35-91 */
HAR_HD D6 d6_fresnel_dielectric(const D6 &cos_theta_i, float eta) {
    const bool outside = cos_theta_i.v >= 0.f;
    const float rcp_eta = 1.f / eta, eta_it = outside ? eta : rcp_eta, eta_ti = outside ? rcp_eta : eta;
    if (eta == 1.f) return D6(0.f);
    if (cos_theta_i.v == 0.f) return D6(1.f);
    const D6 ct2 = (d6_sqr(cos_theta_i) * -1.f + 1.f) * (-eta_ti * eta_ti) + 1.f;
    const D6 ci = d6_abs(cos_theta_i), ct = d6_safe_sqrt(ct2);
    const D6 a_s = (ci - ct * eta_it) / (ci + ct * eta_it), a_p = (ct - ci * eta_it) / (ct + ci * eta_it);
    return (d6_sqr(a_s) + d6_sqr(a_p)) * .5f;
}
/* fresnel_conductor(cos_theta_i, eta + i k), fresnel.h:93-116 */
HAR_HD D6 d6_fresnel_conductor(const D6 &c, float eta_r, float eta_i) {
    const D6 c2 = d6_sqr(c), s2 = c2 * -1.f + 1.f, s4 = d6_sqr(s2);
    const D6 temp_1 = s2 * -1.f + (eta_r * eta_r - eta_i * eta_i);
    const D6 ab = d6_safe_sqrt(d6_sqr(temp_1) + 4.f * eta_i * eta_i * eta_r * eta_r);
    const D6 a = d6_safe_sqrt((ab + temp_1) * .5f);
    const D6 term_1 = ab + c2, term_2 = c * a * 2.f;
    const D6 r_s = (term_1 - term_2) / (term_1 + term_2);
    const D6 term_3 = ab * c2 + s4, term_4 = term_2 * s2;
    const D6 r_p = r_s * (term_3 - term_4) / (term_3 + term_4);
    return (r_s + r_p) * .5f;
}
/* MicrofacetDistribution::eval(m), microfacet.h:185-207 */
HAR_HD D6 d6_microfacet_D(bool ggx, float au, float av, const D6Vec &m) {
    const float alpha_uv = au * av;
    const D6 c2 = d6_sqr(m.z), q = d6_sqr(m.x * (1.f / au)) + d6_sqr(m.y * (1.f / av));
    D6 r;
    if (!ggx) r = d6_exp(-(q / c2)) / (d6_sqr(c2) * (HAR_PI * alpha_uv));
    else r = d6_rcp(d6_sqr(q + c2) * (HAR_PI * alpha_uv));
    return r.v * m.z.v > 1e-20f ? r : D6(0.f);
}
/* MicrofacetDistribution::smith_g1(v, m), microfacet.h:341-365 */
HAR_HD D6 d6_smith_g1(bool ggx, float au, float av, const D6Vec &v, const D6Vec &m) {
    const D6 xy = d6_sqr(v.x * au) + d6_sqr(v.y * av), t = xy / d6_sqr(v.z);
    if (xy.v == 0.f) return D6(1.f);
    if (d6_dot(v, m).v * v.z.v <= 0.f) return D6(0.f);
    if (!ggx) {
        const D6 a = d6_rcp(d6_safe_sqrt(t));
        if (a.v >= 1.6f) return D6(1.f);
        const D6 a2 = d6_sqr(a);
        return (a * 3.535f + a2 * 2.181f) / (a * 2.276f + a2 * 2.577f + 1.f);
    }
    return d6_rcp(d6_safe_sqrt(t + 1.f) + 1.f) * 2.f;
}
/* lerp_gather(table, x), roughplastic.cpp:338-349: piecewise linear, the derivative is the slope of the cell the value falls into */
HAR_HD D6 d6_lerp_gather(const float *data, const D6 &x, uint32_t size) {
    const float xs = x.v * (float) (size - 1);
    uint32_t index = (uint32_t) xs; if (index > size - 2) index = size - 2;
    const float v0 = data[index], v1 = data[index + 1], slope = (v1 - v0) * (float) (size - 1);
    D6 r; r.v = lerp_(v0, v1, xs - (float) index);
    for (int i = 0; i < 6; ++i) r.d[i] = x.d[i] * slope;
    return r;
}

/* F = sum_c A_c value_c(wi, wo) of ONE (not twosided) record and its partials: g[0..2] = dF / d wi, g[3..5] = dF / d wo.  `wi`, `wo` are the directions the
 * record is evaluated with (the caller has mirrored them for the back side of a twosided BSDF and mirrors g[2], g[5] back).  Returns the value. */
HAR_HD float bsdf_weighted_value_dir(const DBsdf &B, const BsdfInputs &in, Vec3 wi_v, Vec3 wo_v, Vec3 A, float g[6]) {
    for (int i = 0; i < 6; ++i) g[i] = 0.f;
    if (!(wi_v.z > 0.f && wo_v.z > 0.f)) return 0.f;
    const D6Vec wi{ D6::input(wi_v.x, 0), D6::input(wi_v.y, 1), D6::input(wi_v.z, 2) }, wo{ D6::input(wo_v.x, 3), D6::input(wo_v.y, 4), D6::input(wo_v.z, 5) };
    const bool nonlinear = (B.flags & BF_NONLINEAR) != 0;
    /* the diffuse base of the two plastic models: value / (1 - fdr_int * value) or value / (1 - fdr_int), weighted */
    const Vec3 den = nonlinear ? Vec3(1.f) - in.slot0 * B.internal_reflectance : Vec3(1.f - B.internal_reflectance);
    const float a_diff = A.x * in.slot0.x / den.x + A.y * in.slot0.y / den.y + A.z * in.slot0.z / den.z;
    D6 F(0.f);
    switch (B.type) {
    case BSDF_DIFFUSE: F = wo.z * (HAR_INV_PI * (A.x * in.slot0.x + A.y * in.slot0.y + A.z * in.slot0.z)); break;
    case BSDF_PLASTIC:
        F = wo.z * (d6_fresnel_dielectric(wi.z, B.eta) * -1.f + 1.f) * (d6_fresnel_dielectric(wo.z, B.eta) * -1.f + 1.f) * (HAR_INV_PI * B.inv_eta_2 * a_diff);
        break;
    case BSDF_ROUGHCONDUCTOR: case BSDF_ROUGHPLASTIC: {
        const bool rc = B.type == BSDF_ROUGHCONDUCTOR, ggx = (B.flags & BF_GGX) != 0;
        const float au = fmaxf(B.alpha_u, 1e-4f), av = rc ? fmaxf(B.alpha_v, 1e-4f) : au;
        D6Vec H{ wi.x + wo.x, wi.y + wo.y, wi.z + wo.z };
        const D6 il = d6_rcp(d6_safe_sqrt(d6_dot(H, H)));
        H.x = H.x * il; H.y = H.y * il; H.z = H.z * il;
        const D6 wih = d6_dot(wi, H);
        if (rc && !(wih.v > 0.f && d6_dot(wo, H).v > 0.f)) break;
        const D6 V = d6_microfacet_D(ggx, au, av, H) * d6_smith_g1(ggx, au, av, wi, H) * d6_smith_g1(ggx, au, av, wo, H) / (wi.z * 4.f);
        if (rc) {
            F = V * (d6_fresnel_conductor(wih, B.eta_c[0], B.k_c[0]) * (A.x * in.slot0.x) + d6_fresnel_conductor(wih, B.eta_c[1], B.k_c[1]) * (A.y * in.slot0.y) +
                     d6_fresnel_conductor(wih, B.eta_c[2], B.k_c[2]) * (A.z * in.slot0.z));
        } else {
            const D6 t_i = d6_lerp_gather(in.table, wi.z, HAR_ROUGH_TRANSMITTANCE_RES), t_o = d6_lerp_gather(in.table, wo.z, HAR_ROUGH_TRANSMITTANCE_RES);
            F = d6_fresnel_dielectric(wih, B.eta) * V * (A.x * in.slot1.x + A.y * in.slot1.y + A.z * in.slot1.z) + wo.z * t_i * t_o * (HAR_INV_PI * B.inv_eta_2 * a_diff);
        }
    } break;
    default: break;                 /* dielectric, conductor: eval() is zero */
    }
    for (int i = 0; i < 6; ++i) g[i] = F.d[i];
    return F.v;
}

} // namespace har
