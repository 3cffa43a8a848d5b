/*
 * har_math.h -- fp32 / integer device arithmetic of the hip_ad_rgb path.
 *
 * Every helper is HAR_HD (host + device) so that the identical source can be
 * compiled by g++ into the host test harness (tests/host_harness) -- a test
 * tool, never a product path: the shipped library only launches HIP kernels.
 *
 * Rounding contract (must match what the reference's LLVM JIT variant emits,
 * see DESIGN.md "Arithmetic contract"): explicit single-rounding fma where the
 * reference calls dr::fmadd/fmsub/fnmadd, IEEE divide and sqrt for
 * dr::rcp/dr::rsqrt/dr::sqrt, no implicit contraction (-ffp-contract=off).
 */
#pragma once
#include <stdint.h>
#include <math.h>
#include <string.h>

#if defined(__HIPCC__)
#  include <hip/hip_runtime.h>
#  define HAR_HD __host__ __device__ __forceinline__
#else
#  define HAR_HD inline
#endif

namespace har {

HAR_HD float fma_(float a, float b, float c)  { return fmaf(a, b, c); }
HAR_HD float fms_(float a, float b, float c)  { return fmaf(a, b, -c); }
HAR_HD float fnma_(float a, float b, float c) { return fmaf(-a, b, c); }
HAR_HD float rcp_(float x)   { return 1.0f / x; }
HAR_HD float rsqrt_(float x) { return 1.0f / sqrtf(x); }
HAR_HD float sqr_(float x)   { return x * x; }

HAR_HD uint32_t as_u32(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(f);
#else
    uint32_t u; memcpy(&u, &f, 4); return u;
#endif
}
HAR_HD float as_f32(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f; memcpy(&f, &u, 4); return f;
#endif
}
/* dr::mulsign / mulsign_neg: flip the sign of a by the sign bit of b (resp. ~b) */
HAR_HD float mulsign_(float a, float b)     { return as_f32(as_u32(a) ^ (as_u32(b) & 0x80000000u)); }
HAR_HD float mulsign_neg_(float a, float b) { return as_f32(as_u32(a) ^ (~as_u32(b) & 0x80000000u)); }
HAR_HD float sign_(float x) { return as_f32(0x3f800000u | (as_u32(x) & 0x80000000u)); }
HAR_HD bool  finite_(float x) { return (as_u32(x) & 0x7f800000u) != 0x7f800000u; }

#define HAR_PI        3.14159265358979323846f
#define HAR_EPSILON   5.9604644775390625e-8f      /* dr::Epsilon<float> = 2^-24 */
#define HAR_INV_PI    0.31830988618379067154f
#define HAR_INF       as_f32(0x7f800000u)
#define HAR_LARGEST   3.402823466e+38f
/* include/mitsuba/core/math.h:17-22 (MI_ENABLE_EMBREE build, dr::Epsilon<float> = 2^-24) */
#define HAR_RAY_EPS    (5.9604644775390625e-8f * 1500.f)
#define HAR_SHADOW_EPS (HAR_RAY_EPS * 10.f)

struct Vec3 {
    float x, y, z;
    HAR_HD Vec3() : x(0.f), y(0.f), z(0.f) {}
    HAR_HD explicit Vec3(float a) : x(a), y(a), z(a) {}
    HAR_HD Vec3(float a, float b, float c) : x(a), y(b), z(c) {}
};
HAR_HD Vec3 operator+(Vec3 a, Vec3 b) { return Vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
HAR_HD Vec3 operator-(Vec3 a, Vec3 b) { return Vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
HAR_HD Vec3 operator*(Vec3 a, Vec3 b) { return Vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
HAR_HD Vec3 operator*(Vec3 a, float s) { return Vec3(a.x * s, a.y * s, a.z * s); }
HAR_HD Vec3 operator-(Vec3 a) { return Vec3(-a.x, -a.y, -a.z); }
HAR_HD Vec3 fma3(Vec3 a, float b, Vec3 c) { return Vec3(fma_(a.x, b, c.x), fma_(a.y, b, c.y), fma_(a.z, b, c.z)); }
HAR_HD Vec3 fma3(Vec3 a, Vec3 b, Vec3 c)  { return Vec3(fma_(a.x, b.x, c.x), fma_(a.y, b.y, c.y), fma_(a.z, b.z, c.z)); }
/* dr::dot: mul, then fma chain; dr::cross: fmsub of rotated components */
HAR_HD float dot3(Vec3 a, Vec3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
HAR_HD Vec3 cross3(Vec3 a, Vec3 b) {
    return Vec3(fms_(a.y, b.z, a.z * b.y), fms_(a.z, b.x, a.x * b.z), fms_(a.x, b.y, a.y * b.x));
}
HAR_HD float norm3(Vec3 a) { return sqrtf(dot3(a, a)); }
HAR_HD Vec3 normalize3(Vec3 a) { return a * rsqrt_(dot3(a, a)); }
HAR_HD Vec3 div3(Vec3 a, float s) { return a * rcp_(s); }   /* Vector / Float => * rcp */
HAR_HD float hmax3(Vec3 a) { return fmaxf(fmaxf(a.x, a.y), a.z); }
HAR_HD Vec3 abs3(Vec3 a) { return Vec3(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }

/* dr::sincos, single precision (Cephes-style reduction by pi/4 + minimax polynomials) */
HAR_HD void sincos_(float x, float &s_out, float &c_out) {
    float xa = fabsf(x);
    int32_t j = (int32_t) (xa * 1.2732395447351626862f);
    j = (j + 1) & ~1;
    float y = (float) j;
    uint32_t sign_sin = ((uint32_t) j << 29) ^ as_u32(x);
    uint32_t sign_cos = (uint32_t) (~(j - 2)) << 29;
    float r = fnma_(y, 0.78515625f, xa);
    r = fnma_(y, 2.4187564849853515625e-4f, r);
    r = fnma_(y, 3.77489497744594108e-8f, r);
    float z = r * r, z2 = z * z;
    float s = fma_(z2, -1.9515295891e-4f, fma_(z, 8.3321608736e-3f, -1.6666654611e-1f)) * z;
    float c = fma_(z2, 2.443315711809948e-5f, fma_(z, -1.388731625493765e-3f, 4.166664568298827e-2f)) * z;
    s = fma_(s, r, r);
    c = fma_(c, z, fma_(z, -0.5f, 1.0f));
    bool poly = (j & 2) == 0;
    s_out = mulsign_(poly ? s : c, as_f32(sign_sin));
    c_out = mulsign_(poly ? c : s, as_f32(sign_cos));
}

/*
 * dr::exp / log / erf / atan2 / acos, single precision.  Dr.Jit implements them as Cephes-style range reductions + minimax polynomials (NOT IN TREE: parity
 * unpinned); the same shape is written out here with explicit fused operations, so that the device, the host build of these headers and the oracle's own
 * restatement (oracle/orc_math.h) produce the SAME BITS -- libm and the device library differ in the last ulp, which used to flip a discrete decision in about
 * 1e-5 of the paths through a rough BSDF.  Maximum errors against double precision: exp 1.0, log 0.8, erf 1.0, acos 1.3, atan2 3.1 ulp.
 */
HAR_HD float exp_(float x) {
    if (!(x > -86.6f)) return x != x ? x : 0.f;                 /* results below 2^-125 are flushed */
    if (x > 88.72283f) return as_f32(0x7f800000u);
    float n = floorf(fma_(x, 1.44269504088896341f, 0.5f));
    float r = fnma_(n, 0.693359375f, x);
    r = fnma_(n, -2.12194440e-4f, r);
    float p = fma_(1.9875691500e-4f, r, 1.3981999507e-3f);
    p = fma_(p, r, 8.3334519073e-3f); p = fma_(p, r, 4.1665795894e-2f); p = fma_(p, r, 1.6666665459e-1f); p = fma_(p, r, 5.0000001201e-1f);
    float y = fma_(p, r * r, r) + 1.f;
    int32_t e = (int32_t) n, h = e >> 1;                        /* 2^n in two exact factors (n = 128 has no single one) */
    return y * as_f32((uint32_t) (h + 127) << 23) * as_f32((uint32_t) (e - h + 127) << 23);
}
HAR_HD float log_(float x) {
    if (!(x > 0.f)) return x == 0.f ? as_f32(0xff800000u) : as_f32(0x7fc00000u);
    uint32_t b = as_u32(x);
    if (b == 0x7f800000u) return x;
    int32_t e = -126;
    if (b < 0x00800000u) { b = as_u32(x * 8388608.f); e = -149; }
    e += (int32_t) (b >> 23);
    float m = as_f32((b & 0x007fffffu) | 0x3f000000u);          /* mantissa in [0.5, 1) */
    if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.f; } else m = m - 1.f;
    float fe = (float) e, z = m * m;
    float p = fma_(7.0376836292e-2f, m, -1.1514610310e-1f);
    p = fma_(p, m, 1.1676998740e-1f); p = fma_(p, m, -1.2420140846e-1f); p = fma_(p, m, 1.4249322787e-1f); p = fma_(p, m, -1.6668057665e-1f);
    p = fma_(p, m, 2.0000714765e-1f); p = fma_(p, m, -2.4999993993e-1f); p = fma_(p, m, 3.3333331174e-1f);
    float y = fma_(fe, -2.12194440e-4f, p * m * z);
    y = fma_(z, -0.5f, y);
    return fma_(fe, 0.693359375f, m + y);
}
HAR_HD float erf_(float a) {
    float t = fabsf(a), s = a * a, r;
    if (t > 0.927734375f) {                                     /* 1 - exp(polynomial) */
        r = fma_(-1.72853470e-5f, t, 3.83197126e-4f);
        r = fma_(r, s, fma_(-3.88396438e-3f, t, 2.42546219e-2f));
        r = fma_(r, t, -1.06777877e-1f); r = fma_(r, t, -6.34846687e-1f); r = fma_(r, t, -1.28717512e-1f);
        r = mulsign_(1.f - exp_(fma_(r, t, -t)), a);
    } else {
        r = fma_(-5.96761703e-4f, s, 4.99119423e-3f);
        r = fma_(r, s, -2.67681349e-2f); r = fma_(r, s, 1.12819925e-1f); r = fma_(r, s, -3.76125336e-1f); r = fma_(r, s, 1.28379166e-1f);
        r = fma_(r, a, a);
    }
    return r;
}
HAR_HD float atan2_(float y, float x) {
    if (x == 0.f) return y == 0.f ? 0.f : mulsign_(0.5f * HAR_PI, y);
    if (y == 0.f) return x < 0.f ? HAR_PI : 0.f;
    const float q = y / x;
    float a = fabsf(q), base = 0.f;
    if (a > 2.414213562373095f) { base = 0.5f * HAR_PI; a = -(1.f / a); }
    else if (a > 0.4142135623730950f) { base = 0.25f * HAR_PI; a = (a - 1.f) / (a + 1.f); }
    const float z = a * a;
    float p = fma_(8.05374449538e-2f, z, -1.38776856032e-1f); p = fma_(p, z, 1.99777106478e-1f); p = fma_(p, z, -3.33329491539e-1f);
    const float at = mulsign_(base + fma_(p * z, a, a), q);
    return (x < 0.f ? mulsign_(HAR_PI, y) : 0.f) + at;
}
HAR_HD float asin_half_(float a) {                              /* asin on [0, 0.5] */
    const float z = a * a;
    float p = fma_(4.2163199048e-2f, z, 2.4181311049e-2f); p = fma_(p, z, 4.5470025998e-2f); p = fma_(p, z, 7.4953002686e-2f); p = fma_(p, z, 1.6666752422e-1f);
    return fma_(p * z, a, a);
}
HAR_HD float acos_(float x) {
    if (x < -0.5f) return HAR_PI - 2.f * asin_half_(sqrtf(0.5f * (1.f + x)));
    if (x > 0.5f) return 2.f * asin_half_(sqrtf(0.5f * (1.f - x)));
    return 0.5f * HAR_PI - mulsign_(asin_half_(fabsf(x)), x);
}
HAR_HD float tan_(float x) { float s, c; sincos_(x, s, c); return s / c; }

/* sample_tea_32, include/mitsuba/core/random.h:76-90 */
HAR_HD void tea32(uint32_t v0, uint32_t v1, uint32_t &o0, uint32_t &o1) {
    uint32_t sum = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i < 4; ++i) {
        sum += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + sum) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    o0 = v0; o1 = v1;
}

/* dr::PCG32 (XSH-RR 64/32) */
#define HAR_PCG32_MULT 0x5851f42d4c957f2dull
HAR_HD uint32_t pcg32_next(uint64_t &state, uint64_t inc) {
    uint64_t old = state;
    state = old * HAR_PCG32_MULT + inc;
    uint32_t xorshifted = (uint32_t) (((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t) (old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((0u - rot) & 31u));
}
HAR_HD float pcg32_next_float(uint64_t &state, uint64_t inc) {
    return as_f32((pcg32_next(state, inc) >> 9) | 0x3f800000u) - 1.0f;
}
/* PCG32Sampler::seed, src/render/sampler.cpp:129-148: lane `idx` of the wavefront */
HAR_HD void sampler_seed(uint32_t seed_value, uint32_t idx, uint64_t &state, uint64_t &inc) {
    uint32_t v0, v1;
    tea32(seed_value, idx, v0, v1);
    state = 0;
    inc = ((uint64_t) v1 << 1) | 1u;
    pcg32_next(state, inc);
    state += (uint64_t) v0;
    pcg32_next(state, inc);
}
HAR_HD uint64_t sampler_inc(uint32_t seed_value, uint32_t idx) {
    uint32_t v0, v1;
    tea32(seed_value, idx, v0, v1);
    return ((uint64_t) v1 << 1) | 1u;
}

/* coordinate_system(), include/mitsuba/core/vector.h:118-138 */
HAR_HD void coordinate_system(Vec3 n, Vec3 &s, Vec3 &t) {
    float sign = sign_(n.z), a = -rcp_(sign + n.z), b = n.x * n.y * a;
    s = Vec3(mulsign_(sqr_(n.x) * a, n.z) + 1.0f, mulsign_(b, n.z), mulsign_neg_(n.x, n.z));
    t = Vec3(b, fma_(n.y, n.y * a, sign), -n.y);
}

/* warp::square_to_cosine_hemisphere, include/mitsuba/core/warp.h:54-90,412-423 */
HAR_HD Vec3 square_to_cosine_hemisphere(float sx, float sy) {
    float x = fms_(2.f, sx, 1.f), y = fms_(2.f, sy, 1.f);
    bool is_zero = (x == 0.f) && (y == 0.f), q13 = fabsf(x) < fabsf(y);
    float r = q13 ? y : x, rp = q13 ? x : y;
    float phi = 0.25f * HAR_PI * rp / r;
    if (q13) phi = 0.5f * HAR_PI - phi;
    if (is_zero) phi = 0.f;
    float s, c;
    sincos_(phi, s, c);
    float px = r * c, py = r * s;
    float z = sqrtf(fmaxf(1.f - fma_(py, py, px * px), 0.f));
    return Vec3(px, py, z);
}

/* AffineTransform (column-major 3x4) * Point / Vector / Normal,
 * include/mitsuba/core/transform.h:285-335 */
HAR_HD Vec3 xf_point(const float *m, Vec3 p) {
    Vec3 r(m[9], m[10], m[11]);
    r = Vec3(fma_(m[0], p.x, r.x), fma_(m[1], p.x, r.y), fma_(m[2], p.x, r.z));
    r = Vec3(fma_(m[3], p.y, r.x), fma_(m[4], p.y, r.y), fma_(m[5], p.y, r.z));
    r = Vec3(fma_(m[6], p.z, r.x), fma_(m[7], p.z, r.y), fma_(m[8], p.z, r.z));
    return r;
}
HAR_HD Vec3 xf_vector(const float *m, Vec3 v) {
    Vec3 r(m[0] * v.x, m[1] * v.x, m[2] * v.x);
    r = Vec3(fma_(m[3], v.y, r.x), fma_(m[4], v.y, r.y), fma_(m[5], v.y, r.z));
    r = Vec3(fma_(m[6], v.z, r.x), fma_(m[7], v.z, r.y), fma_(m[8], v.z, r.z));
    return r;
}
/* normal: multiply by inverse_transpose = transpose(to_object 3x3) */
HAR_HD Vec3 xf_normal(const float *inv, Vec3 n) {
    Vec3 r(inv[0] * n.x, inv[3] * n.x, inv[6] * n.x);
    r = Vec3(fma_(inv[1], n.y, r.x), fma_(inv[4], n.y, r.y), fma_(inv[7], n.y, r.z));
    r = Vec3(fma_(inv[2], n.z, r.x), fma_(inv[5], n.z, r.y), fma_(inv[8], n.z, r.z));
    return r;
}

/* power heuristic, src/integrators/path.cpp:359-364 */
HAR_HD float mis_weight(float pdf_a, float pdf_b) {
    pdf_a *= pdf_a; pdf_b *= pdf_b;
    float w = pdf_a / (pdf_a + pdf_b);
    return finite_(w) ? w : 0.f;
}

} // namespace har
