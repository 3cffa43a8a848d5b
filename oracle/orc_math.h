/*
 * orc_math.h -- scalar fp32 arithmetic of the ORACLE (test infrastructure only).
 *
 * Restates the arithmetic the reference obtains from Dr.Jit 1.5.0 (ext/drjit,
 * NOT IN TREE => "parity unpinned" for the exact rounding of these helpers):
 *   - fmadd/fmsub/fnmadd are single-rounding fused ops,
 *   - dot()  = x*x' then fma chain over y, z         (array_router.h, published),
 *   - cross() = fmsub(a.yzx, b.zxy, a.zxy * b.yzx)    (published),
 *   - rcp(x) = 1/x, rsqrt(x) = 1/sqrt(x) on the LLVM backend (IEEE divide/sqrt),
 *   - normalize(v) = v * rsqrt(dot(v, v)),  v / s = v * rcp(s) for vector/scalar,
 *   - sincos(): Cephes-style range reduction + polynomials (drjit/math.h),
 *   - PCG32 XSH-RR (pcg-random.org; drjit/random.h) and next_float32.
 * Everything in-tree is cited at its use site in mi_oracle.cpp.
 *
 * Compile with -ffp-contract=off: every contraction below is explicit.
 */
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <initializer_list>

namespace orc {

static inline float fmadd(float a, float b, float c)  { return std::fmaf(a, b, c); }
static inline float fmsub(float a, float b, float c)  { return std::fmaf(a, b, -c); }
static inline float fnmadd(float a, float b, float c) { return std::fmaf(-a, b, c); }
static inline float rcp(float x)   { return 1.0f / x; }
static inline float rsqrt(float x) { return 1.0f / std::sqrt(x); }
static inline float sqr(float x)   { return x * x; }
static inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline float mulsign(float a, float b) { return u2f(f2u(a) ^ (f2u(b) & 0x80000000u)); }
static inline float mulsign_neg(float a, float b) { return u2f(f2u(a) ^ (~f2u(b) & 0x80000000u)); }
static inline float sign1(float x) { return std::copysign(1.0f, x); }

constexpr float Pi       = 3.14159265358979323846f;
constexpr float InvPi    = 0.31830988618379067154f;
constexpr float Infinity = std::numeric_limits<float>::infinity();
constexpr float Largest  = std::numeric_limits<float>::max();
/* include/mitsuba/core/math.h:17-22 with dr::Epsilon<float> = 2^-24 (unpinned) */
constexpr float Epsilon       = 0x1p-24f;            /* dr::Epsilon<float> */
constexpr float RayEpsilon    = 0x1p-24f * 1500.f;
constexpr float ShadowEpsilon = RayEpsilon * 10.f;

struct V3 {
    float x, y, z;
    V3() : x(0), y(0), z(0) {}
    V3(float a) : x(a), y(a), z(a) {}
    V3(float a, float b, float c) : x(a), y(b), z(c) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
static inline V3 operator+(V3 a, V3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline V3 operator-(V3 a, V3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline V3 operator*(V3 a, V3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline V3 operator*(V3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline V3 operator*(float s, V3 a) { return V3(a.x * s, a.y * s, a.z * s); }
static inline V3 operator-(V3 a) { return V3(-a.x, -a.y, -a.z); }
static inline V3 fmadd(V3 a, float b, V3 c) { return V3(fmadd(a.x, b, c.x), fmadd(a.y, b, c.y), fmadd(a.z, b, c.z)); }
static inline V3 fmadd(V3 a, V3 b, V3 c) { return V3(fmadd(a.x, b.x, c.x), fmadd(a.y, b.y, c.y), fmadd(a.z, b.z, c.z)); }
static inline V3 fnmadd(V3 a, float b, V3 c) { return V3(fnmadd(a.x, b, c.x), fnmadd(a.y, b, c.y), fnmadd(a.z, b, c.z)); }
static inline float dot(V3 a, V3 b) { return fmadd(a.z, b.z, fmadd(a.y, b.y, a.x * b.x)); }
static inline V3 cross(V3 a, V3 b) {
    return V3(fmsub(a.y, b.z, a.z * b.y), fmsub(a.z, b.x, a.x * b.z), fmsub(a.x, b.y, a.y * b.x));
}
static inline float squared_norm(V3 a) { return dot(a, a); }
static inline float norm(V3 a) { return std::sqrt(dot(a, a)); }
static inline V3 normalize(V3 a) { return a * rsqrt(dot(a, a)); }
static inline V3 div(V3 a, float s) { return a * rcp(s); } /* vector / scalar => * rcp */
static inline float hmax(V3 a) { return std::fmax(std::fmax(a.x, a.y), a.z); }
static inline V3 vabs(V3 a) { return V3(std::fabs(a.x), std::fabs(a.y), std::fabs(a.z)); }

/* drjit/math.h sincos (single precision, Cephes-derived); returns sin, writes cos */
static inline float sincos(float x, float *c_out) {
    float xa = std::fabs(x);
    int32_t j = (int32_t) (xa * 1.2732395447351626862f);
    j = (j + 1) & ~1;
    float y = (float) j;
    uint32_t sign_sin = ((uint32_t) j << 29) ^ f2u(x);
    uint32_t sign_cos = (uint32_t) (~(j - 2)) << 29;
    float r = fnmadd(y, 0.78515625f, xa);
    r = fnmadd(y, 2.4187564849853515625e-4f, r);
    r = fnmadd(y, 3.77489497744594108e-8f, r);
    float z = r * r, z2 = z * z;
    float s = fmadd(z2, -1.9515295891e-4f, fmadd(z, 8.3321608736e-3f, -1.6666654611e-1f)) * z;
    float c = fmadd(z2, 2.443315711809948e-5f, fmadd(z, -1.388731625493765e-3f, 4.166664568298827e-2f)) * z;
    s = fmadd(s, r, r);
    c = fmadd(c, z, fmadd(z, -0.5f, 1.0f));
    bool poly = (j & 2) == 0;
    *c_out = mulsign(poly ? c : s, u2f(sign_cos));
    return mulsign(poly ? s : c, u2f(sign_sin));
}

/*
 * drjit/math.h exp / log / erf / atan2 / acos / tan, single precision (NOT IN TREE: parity unpinned).  Dr.Jit evaluates these as Cephes-derived range
 * reductions followed by a polynomial in Horner / Estrin form with fused operations; libm rounds differently in the last place.  Restated here the Cephes way
 * (S. Moshier's single-precision routines; erf: one polynomial below 0.93, 1 - exp(polynomial) above), every contraction explicit, so that a result is a
 * function of the algorithm and not of the C library the checker happens to be linked with.
 */
template <size_t N> static inline float horner(float x, const float (&c)[N]) {      /* c[0] is the leading coefficient */
    float acc = c[0];
    for (size_t i = 1; i < N; ++i) acc = fmadd(acc, x, c[i]);
    return acc;
}
static inline float exp32(float x) {
    if (std::isnan(x)) return x;
    if (x <= -86.6f) return 0.f;                      /* below 2^-125: flushed */
    if (x > 88.72283f) return Infinity;
    static const float P[] = { 1.9875691500e-4f, 1.3981999507e-3f, 8.3334519073e-3f, 4.1665795894e-2f, 1.6666665459e-1f, 5.0000001201e-1f };
    const float n = std::floor(fmadd(x, 1.44269504088896341f, 0.5f));
    float r = fnmadd(n, 0.693359375f, x);             /* ln 2 in two pieces */
    r = fnmadd(n, -2.12194440e-4f, r);
    const float y = fmadd(horner(r, P), r * r, r) + 1.f;
    const int e = (int) n, half = e >> 1;
    return std::ldexp(std::ldexp(y, half), e - half);
}
static inline float log32(float x) {
    if (std::isnan(x) || x < 0.f) return std::numeric_limits<float>::quiet_NaN();
    if (x == 0.f) return -Infinity;
    if (std::isinf(x)) return x;
    static const float P[] = { 7.0376836292e-2f, -1.1514610310e-1f, 1.1676998740e-1f, -1.2420140846e-1f, 1.4249322787e-1f,
                               -1.6668057665e-1f, 2.0000714765e-1f, -2.4999993993e-1f, 3.3333331174e-1f };
    int e; float m = std::frexp(x, &e);               /* x = m 2^e, m in [0.5, 1) */
    if (m < 0.707106781186547524f) { --e; m = m + m - 1.f; } else m = m - 1.f;
    const float fe = (float) e, z = m * m;
    float y = fmadd(fe, -2.12194440e-4f, horner(m, P) * m * z);
    y = fmadd(z, -0.5f, y);
    return fmadd(fe, 0.693359375f, m + y);
}
static inline float erf32(float a) {
    const float t = std::fabs(a), s = a * a;
    if (t > 0.927734375f) {
        float r = fmadd(fmadd(-1.72853470e-5f, t, 3.83197126e-4f), s, fmadd(-3.88396438e-3f, t, 2.42546219e-2f));
        for (float q : { -1.06777877e-1f, -6.34846687e-1f, -1.28717512e-1f }) r = fmadd(r, t, q);
        return std::copysign(1.f - exp32(fmadd(r, t, -t)), a);
    }
    static const float P[] = { -5.96761703e-4f, 4.99119423e-3f, -2.67681349e-2f, 1.12819925e-1f, -3.76125336e-1f, 1.28379166e-1f };
    return fmadd(horner(s, P), a, a);
}
static inline float atan2_32(float y, float x) {
    const float HalfPi = 0.5f * Pi, QuarterPi = 0.25f * Pi;
    if (x == 0.f) return y == 0.f ? 0.f : std::copysign(HalfPi, y);
    if (y == 0.f) return x < 0.f ? Pi : 0.f;
    static const float P[] = { 8.05374449538e-2f, -1.38776856032e-1f, 1.99777106478e-1f, -3.33329491539e-1f };
    const float q = y / x;
    float a = std::fabs(q), offset = 0.f;
    if (a > 2.414213562373095f)       { offset = HalfPi;    a = -(1.f / a); }             /* tan(3 pi / 8) */
    else if (a > 0.4142135623730950f) { offset = QuarterPi; a = (a - 1.f) / (a + 1.f); }  /* tan(pi / 8) */
    const float z = a * a, at = std::copysign(offset + fmadd(horner(z, P) * z, a, a), q);
    return (x < 0.f ? std::copysign(Pi, y) : 0.f) + at;
}
static inline float asin_small(float a) {             /* 0 <= a <= 0.5 */
    static const float P[] = { 4.2163199048e-2f, 2.4181311049e-2f, 4.5470025998e-2f, 7.4953002686e-2f, 1.6666752422e-1f };
    const float z = a * a;
    return fmadd(horner(z, P) * z, a, a);
}
static inline float acos32(float x) {
    if (x < -0.5f) return Pi - 2.f * asin_small(std::sqrt(0.5f * (1.f + x)));
    if (x > 0.5f)  return 2.f * asin_small(std::sqrt(0.5f * (1.f - x)));
    return 0.5f * Pi - std::copysign(asin_small(std::fabs(x)), x);
}
static inline float tan32(float x) { float c, s = sincos(x, &c); return s / c; }

/* include/mitsuba/core/random.h:76-90 */
static inline void sample_tea_32(uint32_t v0, uint32_t v1, int rounds, uint32_t &o0, uint32_t &o1) {
    uint32_t sum = 0;
    for (int i = 0; i < rounds; ++i) {
        sum += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + sum) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    o0 = v0; o1 = v1;
}

/* PCG32 XSH-RR 64/32 (O'Neill; drjit/random.h -- NOT IN TREE, pinned by the
 * published pcg32 demo vector in tests) */
struct Pcg32 {
    uint64_t state, inc;
    void seed(uint64_t initstate, uint64_t initseq) {
        state = 0;
        inc = (initseq << 1) | 1u;
        next_uint32();
        state += initstate;
        next_uint32();
    }
    uint32_t next_uint32() {
        uint64_t old = state;
        state = old * 0x5851f42d4c957f2dull + inc;
        uint32_t xorshift = (uint32_t) (((old >> 18) ^ old) >> 27);
        uint32_t rot = (uint32_t) (old >> 59);
        return (xorshift >> rot) | (xorshift << ((32 - rot) & 31));
    }
    float next_float32() { return u2f((next_uint32() >> 9) | 0x3f800000u) - 1.0f; }
};

/* src/render/sampler.cpp:129-148: per-lane stream of a wavefront sampler */
static inline Pcg32 sampler_seed(uint32_t seed_value, uint32_t lane) {
    uint32_t v0, v1;
    sample_tea_32(seed_value, lane, 4, v0, v1);
    Pcg32 r;
    r.seed(v0, v1);
    return r;
}

/* include/mitsuba/core/vector.h:118-138 (Duff et al. orthonormal basis) */
static inline void coordinate_system(V3 n, V3 &s, V3 &t) {
    float sign = sign1(n.z), a = -rcp(sign + n.z), b = n.x * n.y * a;
    s = V3(mulsign(sqr(n.x) * a, n.z) + 1.0f, mulsign(b, n.z), mulsign_neg(n.x, n.z));
    t = V3(b, fmadd(n.y, n.y * a, sign), -n.y);
}

/* include/mitsuba/core/warp.h:54-90 */
static inline void square_to_uniform_disk_concentric(float sx, float sy, float &ox, float &oy) {
    float x = fmsub(2.f, sx, 1.f), y = fmsub(2.f, sy, 1.f);
    bool is_zero = (x == 0.f) && (y == 0.f), q13 = std::fabs(x) < std::fabs(y);
    float r = q13 ? y : x, rp = q13 ? x : y;
    float phi = 0.25f * Pi * rp / r;
    if (q13) phi = 0.5f * Pi - phi;
    if (is_zero) phi = 0.f;
    float c, s = sincos(phi, &c);
    ox = r * c; oy = r * s;
}
/* include/mitsuba/core/warp.h:412-436 */
static inline V3 square_to_cosine_hemisphere(float sx, float sy) {
    float px, py;
    square_to_uniform_disk_concentric(sx, sy, px, py);
    float z = std::sqrt(std::fmax(1.f - fmadd(py, py, px * px), 0.f));
    return V3(px, py, z);
}

/* column-major 3x4 affine (ShapeIR::to_world); transform.h:285-335 op order */
static inline V3 xf_point(const float *m, V3 p) {
    V3 r(m[9], m[10], m[11]);
    r = V3(fmadd(m[0], p.x, r.x), fmadd(m[1], p.x, r.y), fmadd(m[2], p.x, r.z));
    r = V3(fmadd(m[3], p.y, r.x), fmadd(m[4], p.y, r.y), fmadd(m[5], p.y, r.z));
    r = V3(fmadd(m[6], p.z, r.x), fmadd(m[7], p.z, r.y), fmadd(m[8], p.z, r.z));
    return r;
}
static inline V3 xf_vector(const float *m, V3 v) {
    V3 r(m[0] * v.x, m[1] * v.x, m[2] * v.x);
    r = V3(fmadd(m[3], v.y, r.x), fmadd(m[4], v.y, r.y), fmadd(m[5], v.y, r.z));
    r = V3(fmadd(m[6], v.z, r.x), fmadd(m[7], v.z, r.y), fmadd(m[8], v.z, r.z));
    return r;
}
/* normal: multiply with the inverse transpose = (to_object 3x3)^T, i.e.
 * result[i] = sum_j inv[j][i] * n[j] with inv given column-major */
static inline V3 xf_normal(const float *inv, V3 n) {
    V3 r(inv[0] * n.x, inv[3] * n.x, inv[6] * n.x);
    r = V3(fmadd(inv[1], n.y, r.x), fmadd(inv[4], n.y, r.y), fmadd(inv[7], n.y, r.z));
    r = V3(fmadd(inv[2], n.z, r.x), fmadd(inv[5], n.z, r.y), fmadd(inv[8], n.z, r.z));
    return r;
}

} // namespace orc
