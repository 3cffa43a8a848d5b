// TEST INFRASTRUCTURE (see mi_oracle.h): part of the CPU oracle, never linked into the product.
//
// AreaLight whose `radiance` is a bitmap, sitting on a Rectangle: the "very different strategy" of src/emitters/area.cpp:133-165 (sample_direction),
// :185-191 (pdf_direction) and :83-90 (eval) -- the texture is importance-sampled (BitmapTexture::sample_position / pdf_position,
// src/textures/bitmap.cpp:622-703, over a DiscreteDistribution2D of the texels' luminance, include/mitsuba/core/distr_2d.h:76-180, built by
// rebuild_internals, bitmap.cpp:876-955) and the uv is mapped onto the shape by Rectangle::eval_parameterization (src/shapes/rectangle.cpp:215-237).
// Included by mi_oracle.cpp after `struct Texture`.  Parity: DiscreteDistribution2D::sample is pinned by the six known answers of the reference's
// test05_discrete_distribution_2d (src/core/tests/test_distr_2d.py:168-180, tests/test_textured_area_light_cpu.py); sample_position / pdf_position by the
// reference's own chi^2 and consistency tests re-hosted there (src/textures/tests/test_bitmap.py:8-29, 250-268).  dr::binary_search and the fused / unfused
// evaluation of luminance() in JIT variants are not in the tree (parity unpinned, as for the 1-D distributions).
#pragma once
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

struct TexelTable {
    uint32_t w = 0, h = 0;
    std::vector<float> marginal, conditional;          // m_marg_cdf (h), m_cond_cdf (w * h): running sums, accumulated in double, stored as float
    float inv_normalization = 0.f, normalization = 0.f; // (float) accum_marg and (float) (1 / accum_marg)
    float back[2][3] = { { 1.f, 0.f, 0.f }, { 0.f, 1.f, 0.f } };   // m_transform.inverse(), rows of the affine map
    float span = 0.f;                                  // |dp_du x dp_dv| of the rectangle (rectangle.cpp:120-123, 181-183)
    bool ready = false;
};

/* luminance(Color3f), include/mitsuba/core/spectrum.h:439-442 */
static inline float texel_luminance(const float *rgb) { return rgb[0] * 0.212671f + rgb[1] * 0.715160f + rgb[2] * 0.072169f; }

/* BitmapTextureImpl::check_sampling_transform (bitmap.cpp:976-992): position sampling needs a to_uv that maps the unit square's corners onto themselves */
static inline bool texel_table_transform_ok(const Texture &t) {
    const float corner[4][2] = { { 0.f, 0.f }, { 1.f, 0.f }, { 1.f, 1.f }, { 0.f, 1.f } };
    uint32_t found = 0;
    for (int c = 0; c < 4; ++c) {
        float q[2]; t.to_texture_space(corner[c], q);
        for (uint32_t j = 0; j < 4; ++j) { const float dx = q[0] - corner[j][0], dy = q[1] - corner[j][1]; if (dx * dx + dy * dy < 1e-8f) found |= 1u << j; }
    }
    return found == 0xFu;
}

/* rebuild_internals(.., init_distr = true) + DiscreteDistribution2D(data, size) + the rectangle's frame.  `to_world`: column-major 3 x 4 */
static inline bool texel_table_build(const Texture &t, const float *to_world, TexelTable &tab, std::string &why) {
    if (!texel_table_transform_ok(t)) { why = "position sampling requires a 'to_uv' transformation that maps the unit square onto itself"; return false; }
    tab.w = t.w; tab.h = t.h; tab.marginal.assign(t.h, 0.f); tab.conditional.assign((size_t) t.w * t.h, 0.f);
    double rows = 0.0;
    for (uint32_t y = 0; y < t.h; ++y) {
        double row = 0.0;
        for (uint32_t x = 0; x < t.w; ++x) {
            const size_t i = (size_t) y * t.w + x;
            row += (double) texel_luminance(t.data.data() + 3 * i);
            tab.conditional[i] = (float) row;
        }
        rows += row; tab.marginal[y] = (float) rows;
    }
    if (!(rows > 0.0)) { why = "the radiance texture has no luminance to sample"; return false; }
    tab.inv_normalization = (float) rows; tab.normalization = (float) (1.0 / rows);
    /* inverse of the 2-D affine map (Transform::inverse of a 3 x 3 affine matrix) */
    const float a = t.xf[0][0], b = t.xf[0][1], c = t.xf[1][0], d = t.xf[1][1], det = a * d - b * c, id = 1.f / det;
    tab.back[0][0] = d * id; tab.back[0][1] = -b * id; tab.back[1][0] = -c * id; tab.back[1][1] = a * id;
    tab.back[0][2] = -(tab.back[0][0] * t.xf[0][2] + tab.back[0][1] * t.xf[1][2]);
    tab.back[1][2] = -(tab.back[1][0] * t.xf[0][2] + tab.back[1][1] * t.xf[1][2]);
    /* Rectangle::update (rectangle.cpp:118-124): dp_du = to_world * (2, 0, 0), dp_dv = to_world * (0, 2, 0); surface_area = |dp_du x dp_dv| */
    const V3 du(to_world[0] * 2.f, to_world[1] * 2.f, to_world[2] * 2.f), dv(to_world[3] * 2.f, to_world[4] * 2.f, to_world[5] * 2.f);
    tab.span = norm(cross(du, dv));
    tab.ready = true;
    return true;
}

/* dr::binary_search(0, last, [cdf[i] < value]): first index whose running sum is not below the value */
static inline uint32_t texel_search(const float *cdf, uint32_t last, float value) {
    uint32_t lo = 0, hi = last, rounds = 0;
    if (lo < hi) { uint32_t width = hi - lo; rounds = 1; while (width >>= 1) ++rounds; }
    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t mid = (lo + hi) >> 1;
        if (cdf[mid] < value) lo = std::min(mid + 1, hi); else hi = mid;
    }
    return lo;
}

/* DiscreteDistribution2D::sample (distr_2d.h:141-180): column, row, discrete probability, re-uniformised sample */
static inline void texel_table_sample(const TexelTable &tab, float sx, float sy, uint32_t &col, uint32_t &row, float &pmf, float &rx, float &ry) {
    const float tiny = 1.17549435e-38f, below_one = 0x1.fffffep-1f;           // dr::Smallest<float>, dr::OneMinusEpsilon<float>
    sx = std::min(std::max(sx, tiny), below_one); sy = std::min(std::max(sy, tiny), below_one);
    sy *= tab.inv_normalization;
    row = texel_search(tab.marginal.data(), tab.h - 1, sy);
    const float *cond = tab.conditional.data() + (size_t) row * tab.w;
    sx *= cond[tab.w - 1];
    col = texel_search(cond, tab.w - 1, sx);
    const float c0 = col > 0 ? cond[col - 1] : 0.f, c1 = cond[col], r0 = row > 0 ? tab.marginal[row - 1] : 0.f, r1 = tab.marginal[row];
    sx -= c0; sy -= r0;
    if (c1 != c0) sx /= c1 - c0;
    if (r1 != r0) sy /= r1 - r0;
    pmf = (c1 - c0) * tab.normalization; rx = sx; ry = sy;
}
/* DiscreteDistribution2D::pdf (distr_2d.h:121-131) */
static inline float texel_table_pdf(const TexelTable &tab, uint32_t x, uint32_t y) {
    const size_t i = (size_t) y * tab.w + x;
    return (tab.conditional[i] - (x > 0 ? tab.conditional[i - 1] : 0.f)) * tab.normalization;
}

/* BitmapTexture::pdf_texture (bitmap.cpp:673-703): density of sample_position in the texture's own parameterisation */
static inline float bitmap_pdf_texture(const Texture &t, const TexelTable &tab, const float pos[2]) {
    const float texels = (float) ((int) t.w * (int) t.h);                     // dr::prod(res)
    if (t.mode & 1u) {                                                         // FilterMode::Nearest
        const int64_t x = (int64_t) std::floor(pos[0] * (float) t.w), y = (int64_t) std::floor(pos[1] * (float) t.h);
        return texel_table_pdf(tab, tex_wrap_pos(x, t.w, t.mode), tex_wrap_pos(y, t.h, t.mode)) * texels;
    }
    const float px = fmadd(pos[0], (float) t.w, -0.5f), py = fmadd(pos[1], (float) t.h, -0.5f);       // bilinear_weights, :775-781
    const float fx = std::floor(px), fy = std::floor(py);
    const int64_t ix = (int64_t) fx, iy = (int64_t) fy;
    const float w1x = px - fx, w1y = py - fy, w0x = 1.f - w1x, w0y = 1.f - w1y;
    const uint32_t x0 = tex_wrap_pos(ix, t.w, t.mode), x1 = tex_wrap_pos(ix + 1, t.w, t.mode), y0 = tex_wrap_pos(iy, t.h, t.mode), y1 = tex_wrap_pos(iy + 1, t.h, t.mode);
    const float v00 = texel_table_pdf(tab, x0, y0), v10 = texel_table_pdf(tab, x1, y0), v01 = texel_table_pdf(tab, x0, y1), v11 = texel_table_pdf(tab, x1, y1);
    const float v0 = fmadd(w0x, v00, w1x * v10), v1 = fmadd(w0x, v01, w1x * v11);
    return fmadd(w0y, v0, w1y * v1) * texels;
}
/* warp::interval_to_tent (warp.h:196-200) */
static inline float interval_to_tent(float s) {
    s -= 0.5f;
    const float root = std::sqrt(std::fmax(fmadd(std::fabs(s), -2.f, 1.f), 0.f));      // dr::safe_sqrt
    return std::copysign(1.f - root, s);
}
/* BitmapTexture::sample_position (bitmap.cpp:622-660): uv in the SURFACE's parameterisation and its density */
static inline void bitmap_sample_position(const Texture &t, const TexelTable &tab, float sx, float sy, float uv[2], float &pdf) {
    uint32_t col, row; float pmf, rx, ry;
    texel_table_sample(tab, sx, sy, col, row, pmf, rx, ry);
    const float inv_w = 1.f / (float) t.w, inv_h = 1.f / (float) t.h;
    float q[2];
    if (t.mode & 1u) { q[0] = ((float) col + rx) * inv_w; q[1] = ((float) row + ry) * inv_h; }
    else {
        q[0] = ((float) col + 0.5f + interval_to_tent(rx)) * inv_w; q[1] = ((float) row + 0.5f + interval_to_tent(ry)) * inv_h;
        for (int k = 0; k < 2; ++k) {
            if (!(t.mode & 6u)) { if (q[k] < 0.f) q[k] += 1.f; if (q[k] > 1.f) q[k] -= 1.f; }      // repeat
            else { if (q[k] < 0.f) q[k] = -q[k]; if (q[k] > 1.f) q[k] = 2.f - q[k]; }               // clamp / mirror: one row of texels beyond the edge folds back
        }
    }
    for (int r = 0; r < 2; ++r) uv[r] = fmadd(tab.back[r][1], q[1], fmadd(tab.back[r][0], q[0], tab.back[r][2]));     // m_transform.inverse() * sample2
    pdf = bitmap_pdf_texture(t, tab, q);
}
/* BitmapTexture::pdf_position (bitmap.cpp:663-670) */
static inline float bitmap_pdf_position(const Texture &t, const TexelTable &tab, const float uv[2]) {
    float q[2]; t.to_texture_space(uv, q);
    return bitmap_pdf_texture(t, tab, q);
}
