/*
 * orc_envmap.h -- ORACLE restatement of the environment-map emitter (TEST INFRASTRUCTURE ONLY).
 *
 *   Hier2D      Hierarchical2D<Float, 0>            include/mitsuba/core/distr_2d.h:370-860 (ctor, sample, invert, eval, Level::index)
 *   bilinear warps                                  include/mitsuba/core/warp.h:446-521
 *   EnvMap      EnvironmentMapEmitter               src/emitters/envmap.cpp:113-180 (halo storage), :214-226 (set_scene), :228-236 (eval),
 *               :284-323 (sample_direction), :325-339 (pdf_direction), :436-459 (uv <-> direction), :476-529 (rebuild_distribution),
 *               :531-548,589-597 (eval_spectrum, RGB branch)
 *
 * Pinned by the Mathematica spot checks of src/core/tests/test_distr_2d.py:7-50 and the weight bounds of
 * src/emitters/tests/test_envmap.py:45-95 (tests/test_envmap_cpu.py).  Parity unpinned: dr::Texture bilinear/clamp lookup,
 * dr::sincos / atan2 / acos lowering (Cephes-style restatements in orc_math.h).
 */
#pragma once
#include "orc_math.h"
#include <algorithm>
#include <cmath>
#include <vector>

namespace orc {

static inline float lerp_(float a, float b, float t) { return fmadd(b, t, fnmadd(a, t, a)); }     // dr::lerp
static inline float clip01(float x) { return std::fmin(std::fmax(x, 0.f), 1.f); }

/* warp.h:446-463 */
static inline float interval_to_linear(float v0, float v1, float sample) {
    if (std::fabs(v0 - v1) > 1e-4f * (v0 + v1))
        return (v0 - std::sqrt(std::fmax(lerp_(sqr(v0), sqr(v1), sample), 0.f))) / (v0 - v1);
    return sample;
}
static inline float linear_to_interval(float v0, float v1, float sample) {
    if (std::fabs(v0 - v1) > 1e-4f * (v0 + v1))
        return sample * ((2.f - sample) * v0 + sample * v1) / (v0 + v1);
    return sample;
}

class Hier2D {
public:
    struct Level {
        uint32_t width = 0, size = 0, offset = 0;
        uint32_t index(uint32_t x, uint32_t y) const { return ((x & 1u) | (((x & ~1u) | (y & 1u)) << 1)) + ((y & ~1u) * width); }
    };
    std::vector<float> data;
    std::vector<Level> levels;
    float patch_size[2] = { 0, 0 }, inv_patch_size[2] = { 0, 0 };
    uint32_t max_patch_index[2] = { 0, 0 };

    bool build(const float *in, uint32_t sx, uint32_t sy, bool normalize = true) {
        if (sx < 2 || sy < 2) return false;                                      // distr_2d.h:233-234
        const uint32_t np[2] = { sx - 1, sy - 1 };
        for (int a = 0; a < 2; ++a) { patch_size[a] = 1.f / (float) np[a]; inv_patch_size[a] = (float) np[a]; max_patch_index[a] = np[a] - 1; }
        uint32_t m = std::max(np[0], np[1]), max_level = 0;
        while ((1u << max_level) < m) ++max_level;                                 // math::log2i_ceil
        levels.clear();
        auto add_level = [&](uint32_t w, uint32_t h) { Level l; l.width = w; l.size = w * h; levels.push_back(l); };
        add_level(sx, sy);
        uint32_t ls[2] = { np[0], np[1] };
        for (uint32_t l = 0; l < max_level; ++l) {
            ls[0] += ls[0] & 1u; ls[1] += ls[1] & 1u;
            add_level(ls[0], ls[1]);
            ls[0] >>= 1; ls[1] >>= 1;
        }
        uint32_t total = 0;
        for (Level &l : levels) { total = (total + 3u) & ~3u; l.offset = total; total += l.size; }
        total = (total + 3u) & ~3u;
        data.assign(total, 0.f);
        const bool has_mip = levels.size() > 1;
        double sum = 0.0;
        for (uint32_t y = 0; y < np[1]; ++y)
            for (uint32_t x = 0; x < np[0]; ++x) {
                const float *p = in + (size_t) y * sx + x;
                float avg = .25f * (p[0] + p[1] + p[sx] + p[sx + 1]);
                sum += (double) avg;
                if (has_mip) data[levels[1].offset + levels[1].index(x, y)] = avg;
            }
        float scale = normalize ? (float) ((double) ((uint64_t) np[0] * np[1]) / sum) : 1.f;
        // note: dr::prod(n_patches) is a uint32 product divided by a double
        for (uint32_t i = 0; i < levels[0].size; ++i) data[levels[0].offset + i] = in[i] * scale;
        if (has_mip) for (uint32_t i = 0; i < levels[1].size; ++i) data[levels[1].offset + i] *= scale;
        uint32_t lsz[2] = { np[0], np[1] };
        for (size_t l = 2; l < levels.size(); ++l) {
            const Level &la = levels[l - 1], &lb = levels[l];
            lsz[0] = (lsz[0] + 1u) >> 1; lsz[1] = (lsz[1] + 1u) >> 1;
            for (uint32_t y = 0; y < lsz[1]; ++y)
                for (uint32_t x = 0; x < lsz[0]; ++x) {
                    const float *d0 = data.data() + la.offset + la.index(x * 2, y * 2);
                    data[lb.offset + lb.index(x, y)] = d0[0] + d0[1] + d0[2] + d0[3];
                }
        }
        return true;
    }

    void sample(float sx, float sy, float out[2], float &pdf) const {
        sx = clip01(sx); sy = clip01(sy);
        uint32_t ox = 0, oy = 0;
        for (int l = (int) levels.size() - 1; l > 0; --l) {
            const Level &lv = levels[l];
            ox <<= 1; oy <<= 1;
            const float *v = data.data() + ((lv.offset + lv.index(ox, oy)) >> 2) * 4;
            float v00 = v[0], v10 = v[1], v01 = v[2], v11 = v[3];
            sx = clip01(sx); sy = clip01(sy);
            float r0 = v00 + v10, r1 = v01 + v11;
            sy *= r0 + r1;
            bool ym = sy > r0;
            if (ym) { oy += 1; sy -= r0; }
            float dy = ym ? r1 : r0;
            float c0 = ym ? v01 : v00, c1 = ym ? v11 : v10;
            sx *= dy;
            bool xm = sx > c0;
            if (xm) { sx -= c0; ox += 1; }
            float dx = xm ? c1 : c0;
            float inv = rcp(dy * dx);
            sy *= dx * inv; sx *= dy * inv;
        }
        const Level &l0 = levels[0];
        const float *d = data.data() + l0.offset + ox + oy * l0.width;
        float v00 = d[0], v10 = d[1], v01 = d[l0.width], v11 = d[l0.width + 1];
        // warp::square_to_bilinear (warp.h:478-494)
        float r0 = v00 + v10, r1 = v01 + v11;
        sy = interval_to_linear(r0, r1, sy);
        float c0 = lerp_(v00, v01, sy), c1 = lerp_(v10, v11, sy);
        sx = interval_to_linear(c0, c1, sx);
        pdf = lerp_(c0, c1, sx);
        out[0] = ((float) (int32_t) ox + sx) * patch_size[0];
        out[1] = ((float) (int32_t) oy + sy) * patch_size[1];
    }

    void invert(float px, float py, float out[2], float &pdf) const {
        px = clip01(px) * inv_patch_size[0]; py = clip01(py) * inv_patch_size[1];
        uint32_t ox = std::min((uint32_t) (int32_t) px, max_patch_index[0]), oy = std::min((uint32_t) (int32_t) py, max_patch_index[1]);
        const Level &l0 = levels[0];
        const float *d = data.data() + l0.offset + ox + oy * l0.width;
        float v00 = d[0], v10 = d[1], v01 = d[l0.width], v11 = d[l0.width + 1];
        px -= (float) (int32_t) ox; py -= (float) (int32_t) oy;
        {   // warp::bilinear_to_square (warp.h:497-513)
            float r0 = v00 + v10, r1 = v01 + v11, c0 = lerp_(v00, v01, py), c1 = lerp_(v10, v11, py);
            pdf = lerp_(c0, c1, px);
            px = linear_to_interval(c0, c1, px);
            py = linear_to_interval(r0, r1, py);
        }
        for (size_t l = 1; l < levels.size(); ++l) {
            const Level &lv = levels[l];
            const float *v = data.data() + ((lv.offset + lv.index(ox & ~1u, oy & ~1u)) >> 2) * 4;
            v00 = v[0]; v10 = v[1]; v01 = v[2]; v11 = v[3];
            bool xm = ox & 1u, ym = oy & 1u;
            float r0 = v00 + v10, r1 = v01 + v11, c0 = ym ? v01 : v00, c1 = ym ? v11 : v10;
            float dy = r0 + r1, dx = c0 + c1;
            py *= ym ? r1 : r0; if (ym) py += r0;
            px *= xm ? c1 : c0; if (xm) px += c0;
            float inv = rcp(dy * dx);
            py *= dx * inv; px *= dy * inv;
            px = clip01(px); py = clip01(py);
            ox >>= 1; oy >>= 1;
        }
        out[0] = px; out[1] = py;
    }

    float eval(float px, float py) const {
        px = clip01(px) * inv_patch_size[0]; py = clip01(py) * inv_patch_size[1];
        uint32_t ox = std::min((uint32_t) (int32_t) px, max_patch_index[0]), oy = std::min((uint32_t) (int32_t) py, max_patch_index[1]);
        px -= (float) (int32_t) ox; py -= (float) (int32_t) oy;
        const Level &l0 = levels[0];
        const float *d = data.data() + l0.offset + ox + oy * l0.width;
        return lerp_(lerp_(d[0], d[1], px), lerp_(d[l0.width], d[l0.width + 1], px), py);      // square_to_bilinear_pdf
    }
};

class EnvMap {
public:
    uint32_t w = 0, h = 0;                  // real resolution (after pad_to(2, 3))
    std::vector<float> tex;                 // H x (W + 2) x 3: halo columns make WrapMode::Clamp periodic in phi
    Hier2D warp;                            // (W + 1) x H luminance * sin(theta)
    float scale = 1.f;
    float to_world[12], to_local[12];       // column-major 3 x 4
    float center[3] = { 0, 0, 0 }, radius = 1.f;

    void init(const float *rgb, uint32_t iw, uint32_t ih, float scale_, bool mis_compensation, const float *tw, const float *tl) {
        w = std::max(iw, 2u); h = std::max(ih, 3u);                                  // Bitmap::pad_to (bitmap.cpp:397-427): replicate last pixel / row
        scale = scale_;
        std::copy(tw, tw + 12, to_world); std::copy(tl, tl + 12, to_local);
        const uint32_t sw = w + 2;
        tex.assign((size_t) h * sw * 3, 0.f);
        for (uint32_t y = 0; y < h; ++y)
            for (uint32_t x = 0; x < w; ++x) {
                const float *src = rgb + 3 * ((size_t) std::min(y, ih - 1) * iw + std::min(x, iw - 1));
                float *dst = tex.data() + 3 * ((size_t) y * sw + x + 1);
                dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
            }
        for (uint32_t y = 0; y < h; ++y) {                                              // refresh_halo (envmap.cpp:463-472)
            float *r = tex.data() + 3 * (size_t) y * sw;
            for (int c = 0; c < 3; ++c) { r[c] = r[3 * w + c]; r[3 * (w + 1) + c] = r[3 + c]; }
        }
        // rebuild_distribution (envmap.cpp:476-529)
        const uint32_t rx = w + 1, ry = h;
        std::vector<float> lum((size_t) rx * ry);
        for (uint32_t y = 0; y < ry; ++y)
            for (uint32_t x = 0; x < rx; ++x) {
                const float *c = tex.data() + 3 * ((size_t) y * sw + x + 1);
                lum[(size_t) y * rx + x] = c[0] * 0.212671f + c[1] * 0.715160f + c[2] * 0.072169f;
            }
        float offset = 0.f;
        if (mis_compensation) {
            float min_lum = Infinity; double acc = 0.0;
            for (uint32_t y = 0; y < ry; ++y) for (uint32_t x = 0; x + 1 < rx; ++x) { float l = lum[(size_t) y * rx + x]; min_lum = std::fmin(min_lum, l); acc += (double) l; }
            offset = (float) (acc / (double) ((size_t) (rx - 1u) * ry));
            if (offset - min_lum <= 0.01f * offset) offset = 0.f;
        }
        float theta_scale = 1.f / (float) (ry - 1) * Pi;
        for (uint32_t y = 0; y < ry; ++y) {
            float sin_theta = std::sin((float) y * theta_scale);
            for (uint32_t x = 0; x < rx; ++x) { float &l = lum[(size_t) y * rx + x]; l = std::fmax(l - offset, 0.f) * sin_theta; }
        }
        warp.build(lum.data(), rx, ry);
    }

    float half_texel() const { return .5f / (float) w; }

    V3 uv_to_direction(float u, float v, float &inv_sin_theta) const {
        float theta = v * Pi, phi = u * (2.f * Pi);
        float ct, st = sincos(theta, &ct), cp, sp = sincos(phi, &cp);
        inv_sin_theta = rcp(std::fmax(st, 0x1p-24f));                                // dr::Epsilon<float>
        return V3(sp * st, ct, -cp * st);
    }
    static void direction_to_uv(V3 d, float uv[2]) {
        uv[0] = atan2_32(d.x, -d.z) * (0.5f * InvPi);
        uv[1] = acos32(std::fmin(std::fmax(d.y, -1.f), 1.f)) * InvPi;
    }

    /* eval_spectrum, RGB branch: dr::Texture<Float, 2>::eval, Linear + Clamp, on the halo'ed storage */
    V3 eval_uv(float u_, float v_) const {
        const float rx = (float) w, ry = (float) h;
        float u = u_ - std::floor(u_), v = clip01(v_);
        float pos_x = fmadd(u, rx, 1.f) / (rx + 2.f), pos_y = fmadd(v, ry - 1.f, 0.5f) / ry;
        const uint32_t sw = w + 2;
        float px = fmadd(pos_x, (float) sw, -0.5f), py = fmadd(pos_y, ry, -0.5f);
        float fx = std::floor(px), fy = std::floor(py);
        int32_t ix = (int32_t) fx, iy = (int32_t) fy;
        float w1x = px - fx, w1y = py - fy, w0x = 1.f - w1x, w0y = 1.f - w1y;
        auto clampi = [](int32_t i, int32_t n) { return (uint32_t) std::min(std::max(i, 0), n - 1); };
        uint32_t x0 = clampi(ix, (int32_t) sw), x1 = clampi(ix + 1, (int32_t) sw), y0 = clampi(iy, (int32_t) h), y1 = clampi(iy + 1, (int32_t) h);
        float out[3];
        for (int c = 0; c < 3; ++c) {
            float v00 = tex[3 * ((size_t) y0 * sw + x0) + c], v10 = tex[3 * ((size_t) y0 * sw + x1) + c],
                  v01 = tex[3 * ((size_t) y1 * sw + x0) + c], v11 = tex[3 * ((size_t) y1 * sw + x1) + c];
            float a = fmadd(w0x, v00, w1x * v10), b = fmadd(w0x, v01, w1x * v11);
            out[c] = fmadd(w0y, a, w1y * b) * scale;
        }
        return V3(out[0], out[1], out[2]);
    }

    /* eval(si): radiance arriving along direction `d_world` = -si.wi (envmap.cpp:228-236) */
    V3 eval(V3 d_world) const { float uv[2]; direction_to_uv(xf_vector(to_local, d_world), uv); return eval_uv(uv[0], uv[1]); }

    /* sample_direction (envmap.cpp:284-323) */
    void sample_direction(V3 ref_p, float s0, float s1, V3 &d, float &dist, float &pdf_out, V3 &weight, float uv_out[2]) const {
        float uv[2], pdf; warp.sample(s0, s1, uv, pdf);
        uv[0] += half_texel();
        bool active = pdf > 0.f;
        float inv_sin_theta; V3 dl = uv_to_direction(uv[0], uv[1], inv_sin_theta);
        V3 c(center[0], center[1], center[2]);
        float r = std::fmax(radius, norm(ref_p - c));
        dist = 2.f * r;
        d = xf_vector(to_world, dl);
        pdf_out = active ? pdf * inv_sin_theta * (1.f / (2.f * sqr(Pi))) : 0.f;
        weight = active ? div(eval_uv(uv[0], uv[1]), pdf_out) : V3(0.f);
        uv_out[0] = uv[0]; uv_out[1] = uv[1];
    }

    /* pdf_direction (envmap.cpp:325-339) */
    float pdf_direction(V3 d_world) const {
        V3 d = xf_vector(to_local, d_world);
        float uv[2]; direction_to_uv(d, uv);
        uv[0] -= half_texel();
        uv[0] -= std::floor(uv[0]); uv[1] -= std::floor(uv[1]);
        float s2 = std::fmax(sqr(d.x) + sqr(d.z), sqr(0x1p-24f));
        float inv_sin_theta = s2 > 0.f ? rsqrt(s2) : 0.f;                                 // dr::safe_rsqrt
        return warp.eval(uv[0], uv[1]) * inv_sin_theta * (1.f / (2.f * sqr(Pi)));
    }
};

} // namespace orc
