/*
 * har_scene.h -- device-side scene records and the per-vertex shading math of
 * the hip_ad_rgb path (surface interaction, diffuse BSDF, bitmap/srgb texture,
 * area emitter, perspective sensor, reconstruction filter).  HAR_HD so that the
 * same source runs in the HIP kernels and in the host test harness.
 * Each function cites the reference code it reproduces.
 */
#pragma once
#include "har_accel.h"
#include "har_bsdf.h"

namespace har {

struct DMesh    { uint32_t voff, foff, bsdf; int32_t emitter; uint32_t flags, face_count, vertex_count, pad1; };
/* mode: HarTexture::mode (bit 0 nearest, bit 1 mirror, bit 2 clamp), bit 31 (HAR_TEX_HAS_UV_XF): `uvm` is not the identity; uvm = BitmapTexture's `to_uv`
 * (bitmap.cpp:175), row-major 2 x 3 -- tex_taps applies it to the surface's uv before the lookup (bitmap.cpp:565,792,831,847) */
#define HAR_TEX_HAS_UV_XF 0x80000000u
struct DTexture { const float *data; uint32_t w, h; uint32_t mode, pad; float uvm[6]; };
/* type 0: AreaLight on a rectangle (to_world, normal, inv_area, mesh); type 1: ConstantBackgroundEmitter
 * (src/emitters/constant.cpp): to_world[0..2] = bounding sphere centre, to_world[3] = radius, mesh = 0xffffffff; type 2: environment map
 * (DEnvmap); type 3: AreaLight on a triangle mesh: mesh, inv_area = 1 / surface area, to_world[0] / [1] = bit patterns of the offset of its
 * table in DScene::emitter_cdf and of its face count, to_world[2] = sum of the face areas */
#ifndef HAR_SHADING_TRIS
/* 1 (default since round 5): compute_si reads ONE pre-gathered 96-byte block per face (its three vertex records, DScene::shade_tris) instead of face -> three scattered
 * vertices: one level of dependent loads less, one or two cache lines instead of three or four.  Bracketed A/B on one box (profiles/r05_ab_shading_tris.txt): k_shade
 * 10.5 -> 10.2 ms on instanced1m, 24.5 -> 23.85 ms on materials1m, flat1m unchanged; forward +0.5 %; GPU suite green on both builds.  Costs 96 B per face of HBM. */
#define HAR_SHADING_TRIS 1
#endif
struct DEmitter { float radiance[3]; float inv_area; float to_world[12]; float normal[3]; uint32_t mesh; uint32_t type; };
/* type 7 (area light on a rectangle with a BITMAP radiance): radiance[0] = bits of the texture index, radiance[1] = bits of the offset of its texel distribution in
 * DScene::emitter_cdf, radiance[2] = |dp_du x dp_dv| of the rectangle; the table: HAR_TEXEL_TABLE_HEADER floats { sum, 1 / sum, inverse to_uv (2 x 3) }, the marginal
 * running sums (h), the conditional ones (w * h) -- texel_table_fill in har_scene_host.cpp */
#define HAR_TEXEL_TABLE_HEADER 8u
struct DInst    { float to_world[12]; float to_object[12]; };
/* EnvironmentMapEmitter (src/emitters/envmap.cpp), emitter type 2.  `tex` = H x (W + 2) x 3 radiance with one halo column on each side
 * (:140-172), `warp` = storage of the Hierarchical2D<Float, 0> over the (W + 1) x H luminance * sin(theta) grid (distr_2d.h:405-560):
 * level 0 row-major, levels >= 1 in 2 x 2 blocks; lvl_offset / lvl_width index it. */
#define HAR_ENV_MAX_LEVELS 18
struct DEnvmap {
    const float *tex, *warp;
    uint32_t w, h, n_levels; float scale;
    float to_world[12], to_local[12];       /* column-major 3 x 4 */
    float center[3], radius;                /* scene bounding sphere (set_scene, :214-226) */
    uint32_t lvl_offset[HAR_ENV_MAX_LEVELS], lvl_width[HAR_ENV_MAX_LEVELS];
};

struct DScene {
    Accel accel;
    const uint32_t *blas_tri_ranges;   /* per TLAS record: first, count (brute-force kernel) */
    const float    *verts;             /* packed vertices, 8 f32 each (mesh_utils.h:19-34) */
#if HAR_SHADING_TRIS
    const float    *shade_tris;        /* the three vertex records of every face, 24 f32 per face in face order (see compute_si) */
#endif
    const uint32_t *faces;             /* packed faces, 4 u32 each */
    const DMesh    *meshes;
    const DBsdf    *bsdfs;
    const DTexture *textures;
    const DEmitter *emitters;
    const DInst    *insts;
    const float    *bsdf_tables;       /* roughplastic external transmittance tables, 64 floats each */
    uint32_t n_emitters, n_meshes, n_bsdfs, n_textures, n_insts;      /* n_insts: records of `insts` (the scene's instances) */
    int32_t  env_emitter;              /* index of the environment emitter (Scene::environment()), or -1 */
    uint32_t bsdf_types;               /* bit mask (1 << type) of the BSDF types present (+ bit 31: some record is twosided) */
    const DEnvmap *envmap;             /* device record of the environment map when emitters[env_emitter].type == 2 */
    const float   *emitter_cdf;        /* face-area tables of the mesh emitters (type 3): per emitter `count` unnormalised pmf values, then `count` cdf values */
    /* Scene::m_emitter_distr (scene.cpp:120-141): NULL = uniform emitter selection (every sampling_weight is 1); otherwise n_emitters unnormalised pmf values (the
     * weights), then n_emitters cdf values (running sums accumulated in double, distr_1d.h:236-266), with their sum, 1 / sum and the first / last bin of non-zero weight */
    const float   *emitter_distr;
    float    emitter_sum, emitter_norm;
    uint32_t emitter_valid_lo, emitter_valid_hi;
    /* copy of emitters[0] for scenes with exactly one emitter (kernel argument = scalar registers); emitter0_valid = 0: use the array; 2: the radiance was updated on the
     * device (har_scene_set_emitter_radiance_device) -- those three floats come from the array, everything else from this copy (round 6: an emitter-optimisation loop
     * keeps the scalar-register path) */
    DEmitter emitter0; uint32_t emitter0_valid;
};

struct DSensor {
    float s2c[16];
    float to_world[16];
    float near_clip, far_clip;
    uint32_t crop_x, crop_y, crop_w, crop_h;
    uint32_t samp_w, samp_h, border;   /* the sample grid of render(): the crop window + `border` pixels on every side (Film::sample_border, integrator.cpp:162-165); border = 0: the crop */
    uint32_t rfilter;            /* 0 box, 1 gaussian, 2 tent, 3 mitchell, 4 catmullrom, 5 lanczos */
    float rf_p0, rf_p1;          /* filter parameters (HarSensor::rfilter_stddev / rfilter_param1) */
    float radius;
    float coeff[10];             /* GaussianFilter::m_coeff (LLVM branch) */
    float ppo_x, ppo_y;          /* scaled principal point offset: film size * principal_point_offset / crop size (perspective.cpp:213-214) */
    uint32_t projection;         /* 0 perspective, 1 orthographic (HarSensor::projection) */
};

struct SurfInt {
    float t;
    Vec3 p, n, sn, ss, st, wi;
    float uv_x, uv_y;
    uint32_t mesh;
    HAR_HD bool valid() const { return t != HAR_INF; }
    HAR_HD Vec3 to_local(Vec3 v) const { return Vec3(dot3(v, ss), dot3(v, st), dot3(v, sn)); }   /* frame.h:34 */
    HAR_HD Vec3 to_world(Vec3 v) const { return fma3(sn, v.z, fma3(st, v.y, ss * v.x)); }         /* frame.h:39 */
};

/* PreliminaryIntersection::compute_surface_interaction (interaction.h:804-829) ->
 * Mesh::compute_surface_interaction (src/render/mesh.cpp:2255-2437) [+ Instance::
 * compute_surface_interaction, src/shapes/instance.cpp:150-266] ->
 * finalize_surface_interaction (interaction.h:559-605; `diffuse` packs no tangents so
 * the frame comes from coordinate_system(sh_frame.n)) */
HAR_HD SurfInt compute_si(const DScene &S, Vec3 ray_d, float t, float bu, float bv, uint32_t prim, uint32_t shape, uint32_t inst) {
    SurfInt si;
    si.t = t; si.uv_x = 0.f; si.uv_y = 0.f; si.mesh = 0;
    if (t == HAR_INF) { si.wi = -ray_d; return si; }
    const DMesh M = S.meshes[shape];
#if HAR_SHADING_TRIS
    /* the face's three vertex records pre-gathered into ONE 96-byte block -- hit -> mesh record -> block instead of hit -> mesh record -> face -> three scattered
     * vertices */
    const float *r0 = S.shade_tris + 24 * (size_t) (M.foff + prim), *r1 = r0 + 8, *r2 = r0 + 16;
#else
    const uint32_t *f = S.faces + 4 * (size_t) (M.foff + prim);
    const float *r0 = S.verts + 8 * (size_t) (M.voff + f[0]);
    const float *r1 = S.verts + 8 * (size_t) (M.voff + f[1]);
    const float *r2 = S.verts + 8 * (size_t) (M.voff + f[2]);
#endif
#if defined(__HIP_DEVICE_COMPILE__)
    const float4 qa0 = reinterpret_cast<const float4 *>(r0)[0], qa1 = reinterpret_cast<const float4 *>(r0)[1];
    const float4 qb0 = reinterpret_cast<const float4 *>(r1)[0], qb1 = reinterpret_cast<const float4 *>(r1)[1];
    const float4 qc0 = reinterpret_cast<const float4 *>(r2)[0], qc1 = reinterpret_cast<const float4 *>(r2)[1];
    const float v0[8] = { qa0.x, qa0.y, qa0.z, qa0.w, qa1.x, qa1.y, qa1.z, qa1.w };
    const float v1[8] = { qb0.x, qb0.y, qb0.z, qb0.w, qb1.x, qb1.y, qb1.z, qb1.w };
    const float v2[8] = { qc0.x, qc0.y, qc0.z, qc0.w, qc1.x, qc1.y, qc1.z, qc1.w };
#else
    const float *v0 = r0, *v1 = r1, *v2 = r2;
#endif
    Vec3 p0(v0[0], v0[1], v0[2]), p1(v1[0], v1[1], v1[2]), p2(v2[0], v2[1], v2[2]);
    float b1 = bu, b2 = bv, b0 = 1.f - b1 - b2;
    Vec3 e1 = p1 - p0, e2 = p2 - p0;
    si.p = fma3(p0, b0, fma3(p1, b1, p2 * b2));
    si.n = normalize3(cross3(e1, e2));                       /* mesh.h:568-572 */
    if (M.flags & 1u) {
        Vec3 n0(v0[3], v0[4], v0[5]), dn1 = Vec3(v1[3], v1[4], v1[5]) - n0, dn2 = Vec3(v2[3], v2[4], v2[5]) - n0;
        Vec3 n = fma3(dn1, b1, fma3(dn2, b2, n0));
        si.sn = n * rsqrt_(dot3(n, n));
    } else si.sn = si.n;
    if (M.flags & 2u) {
        float u0 = v0[6], w0 = v0[7], du0 = v1[6] - u0, dv0 = v1[7] - w0, du1 = v2[6] - u0, dv1 = v2[7] - w0;
        si.uv_x = fma_(du0, b1, fma_(du1, b2, u0));
        si.uv_y = fma_(dv0, b1, fma_(dv1, b2, w0));
    } else { si.uv_x = b1; si.uv_y = b2; }
    si.mesh = shape;
    if (inst != 0xffffffffu) {                              /* instance.cpp:196-224 */
        const DInst &I = S.insts[inst];
        si.p = xf_point(I.to_world, si.p);
        si.n = normalize3(xf_normal(I.to_object, si.n));
        Vec3 n = xf_normal(I.to_object, si.sn);
        si.sn = n * rcp_(norm3(n));
    }
    coordinate_system(si.sn, si.ss, si.st);
    si.wi = si.to_local(-ray_d);
    return si;
}

/* the same interaction from a hit record that carries the face's index in the shading-triangle array and the mesh's flags (HAR_HIT_MATINFO): no load of the mesh record */
HAR_HD SurfInt compute_si_record(const DScene &S, Vec3 ray_d, float t, float bu, float bv, uint32_t gface, uint32_t mesh_flags, uint32_t shape, uint32_t inst) {
    SurfInt si;
    si.t = t; si.uv_x = 0.f; si.uv_y = 0.f; si.mesh = 0;
    if (t == HAR_INF) { si.wi = -ray_d; return si; }
#if HAR_SHADING_TRIS
    const float *r0 = S.shade_tris + 24 * (size_t) gface, *r1 = r0 + 8, *r2 = r0 + 16;
#else
    const uint32_t *f = S.faces + 4 * (size_t) gface; const uint32_t voff = S.meshes[shape].voff;
    const float *r0 = S.verts + 8 * (size_t) (voff + f[0]), *r1 = S.verts + 8 * (size_t) (voff + f[1]), *r2 = S.verts + 8 * (size_t) (voff + f[2]);
#endif
#if defined(__HIP_DEVICE_COMPILE__)
    const float4 qa0 = reinterpret_cast<const float4 *>(r0)[0], qa1 = reinterpret_cast<const float4 *>(r0)[1];
    const float4 qb0 = reinterpret_cast<const float4 *>(r1)[0], qb1 = reinterpret_cast<const float4 *>(r1)[1];
    const float4 qc0 = reinterpret_cast<const float4 *>(r2)[0], qc1 = reinterpret_cast<const float4 *>(r2)[1];
    const float v0[8] = { qa0.x, qa0.y, qa0.z, qa0.w, qa1.x, qa1.y, qa1.z, qa1.w };
    const float v1[8] = { qb0.x, qb0.y, qb0.z, qb0.w, qb1.x, qb1.y, qb1.z, qb1.w };
    const float v2[8] = { qc0.x, qc0.y, qc0.z, qc0.w, qc1.x, qc1.y, qc1.z, qc1.w };
#else
    const float *v0 = r0, *v1 = r1, *v2 = r2;
#endif
    Vec3 p0(v0[0], v0[1], v0[2]), p1(v1[0], v1[1], v1[2]), p2(v2[0], v2[1], v2[2]);
    float b1 = bu, b2 = bv, b0 = 1.f - b1 - b2;
    Vec3 e1 = p1 - p0, e2 = p2 - p0;
    si.p = fma3(p0, b0, fma3(p1, b1, p2 * b2));
    si.n = normalize3(cross3(e1, e2));
    if (mesh_flags & 1u) {
        Vec3 n0(v0[3], v0[4], v0[5]), dn1 = Vec3(v1[3], v1[4], v1[5]) - n0, dn2 = Vec3(v2[3], v2[4], v2[5]) - n0;
        Vec3 n = fma3(dn1, b1, fma3(dn2, b2, n0));
        si.sn = n * rsqrt_(dot3(n, n));
    } else si.sn = si.n;
    if (mesh_flags & 2u) {
        float u0 = v0[6], w0 = v0[7], du0 = v1[6] - u0, dv0 = v1[7] - w0, du1 = v2[6] - u0, dv1 = v2[7] - w0;
        si.uv_x = fma_(du0, b1, fma_(du1, b2, u0));
        si.uv_y = fma_(dv0, b1, fma_(dv1, b2, w0));
    } else { si.uv_x = b1; si.uv_y = b2; }
    si.mesh = shape;
    if (inst != 0xffffffffu) {
        const DInst &I = S.insts[inst];
        si.p = xf_point(I.to_world, si.p);
        si.n = normalize3(xf_normal(I.to_object, si.n));
        Vec3 n = xf_normal(I.to_object, si.sn);
        si.sn = n * rcp_(norm3(n));
    }
    coordinate_system(si.sn, si.ss, si.st);
    si.wi = si.to_local(-ray_d);
    return si;
}

/* RayFlags (include/mitsuba/render/interaction.h:19-87).  The wavefront kernels compute what RayFlags::Default asks for (compute_si above); the array-valued
 * Scene::ray_intersect / PreliminaryIntersection::compute_surface_interaction entry points honour the caller's flags through compute_si_flags.  FollowShape /
 * DetachShape only choose which AD dependence the reference tracks through the hit; the primal values are the same (the C ABI carries no AD graph). */
enum { RAY_MINIMAL = 0u, RAY_SHADING = 1u, RAY_NORMAL_PARTIALS = 2u, RAY_FOLLOW_SHAPE = 4u, RAY_DETACH_SHAPE = 8u, RAY_KNOWN_FLAGS = 15u };
struct SurfPartials { Vec3 dp_du, dp_dv, dn_du, dn_dv; };
/* Position / normal partials of Mesh::compute_surface_interaction (src/render/mesh.cpp:2334-2395: dn / d barycentric of the normalised interpolated normal,
 * change of variables to the texture parameterisation) and their transport through an instance (src/shapes/instance.cpp:206-212,250-253) */
HAR_HD void compute_si_partials(const DScene &S, float bu, float bv, uint32_t prim, uint32_t shape, uint32_t inst, bool normal_partials, SurfPartials &P) {
    const DMesh M = S.meshes[shape];
    const uint32_t *f = S.faces + 4 * (size_t) (M.foff + prim);
    const float *v0 = S.verts + 8 * (size_t) (M.voff + f[0]), *v1 = S.verts + 8 * (size_t) (M.voff + f[1]), *v2 = S.verts + 8 * (size_t) (M.voff + f[2]);
    const Vec3 p0(v0[0], v0[1], v0[2]), e1 = Vec3(v1[0], v1[1], v1[2]) - p0, e2 = Vec3(v2[0], v2[1], v2[2]) - p0;
    const float b1 = bu, b2 = bv;
    const bool need_dn = (M.flags & 1u) && normal_partials;
    Vec3 dn_db1(0.f), dn_db2(0.f), sn(0.f);
    if (M.flags & 1u) {
        Vec3 n0(v0[3], v0[4], v0[5]), dn1 = Vec3(v1[3], v1[4], v1[5]) - n0, dn2 = Vec3(v2[3], v2[4], v2[5]) - n0;
        Vec3 n = fma3(dn1, b1, fma3(dn2, b2, n0));
        const float il = rsqrt_(dot3(n, n));
        n = n * il; sn = n;
        if (need_dn) {
            dn1 = dn1 * il; dn2 = dn2 * il;
            dn_db1 = fma3(n, -dot3(n, dn1), dn1); dn_db2 = fma3(n, -dot3(n, dn2), dn2);        /* dr::fnmadd(n, dot(n, dn), dn) */
        }
    }
    P.dn_du = Vec3(0.f); P.dn_dv = Vec3(0.f);
    if (M.flags & 2u) {
        const float u0 = v0[6], w0 = v0[7], du0x = v1[6] - u0, du0y = v1[7] - w0, du1x = v2[6] - u0, du1y = v2[7] - w0;
        const float det = fms_(du0x, du1y, du0y * du1x), inv_det = det != 0.f ? rcp_(det) : 0.f;
        /* to_uv_basis(d1, d2) = ( fmsub(duv1.y, d1, duv0.y * d2), fnmadd(duv1.x, d1, duv0.x * d2) ) * inv_det */
        P.dp_du = Vec3(fms_(du1y, e1.x, du0y * e2.x), fms_(du1y, e1.y, du0y * e2.y), fms_(du1y, e1.z, du0y * e2.z)) * inv_det;
        P.dp_dv = Vec3(fnma_(du1x, e1.x, du0x * e2.x), fnma_(du1x, e1.y, du0x * e2.y), fnma_(du1x, e1.z, du0x * e2.z)) * inv_det;
        if (need_dn) {
            P.dn_du = Vec3(fms_(du1y, dn_db1.x, du0y * dn_db2.x), fms_(du1y, dn_db1.y, du0y * dn_db2.y), fms_(du1y, dn_db1.z, du0y * dn_db2.z)) * inv_det;
            P.dn_dv = Vec3(fnma_(du1x, dn_db1.x, du0x * dn_db2.x), fnma_(du1x, dn_db1.y, du0x * dn_db2.y), fnma_(du1x, dn_db1.z, du0x * dn_db2.z)) * inv_det;
        }
    } else {
        P.dp_du = e1; P.dp_dv = e2;
        if (need_dn) { P.dn_du = dn_db1; P.dn_dv = dn_db2; }
    }
    if (inst != 0xffffffffu) {
        const DInst &I = S.insts[inst];
        if (need_dn) {                  /* instance.cpp:206-212 (meshes without vertex normals: the partials are zero and stay zero) */
            Vec3 n = xf_normal(I.to_object, sn);
            const float inv_len = rcp_(norm3(n));
            n = n * inv_len;
            const Vec3 du = xf_normal(I.to_object, P.dn_du) * inv_len, dv = xf_normal(I.to_object, P.dn_dv) * inv_len;
            P.dn_du = fma3(n, -dot3(n, du), du); P.dn_dv = fma3(n, -dot3(n, dv), dv);
        }
        P.dp_du = xf_vector(I.to_world, P.dp_du); P.dp_dv = xf_vector(I.to_world, P.dp_dv);
    }
}

/* Interaction::offset_p / spawn_ray / spawn_ray_to (interaction.h:161-191) */
HAR_HD Vec3 offset_p(const SurfInt &si, Vec3 d) {
    float mag = (1.f + hmax3(abs3(si.p))) * HAR_RAY_EPS;
    mag = mulsign_(mag, dot3(si.n, d));
    return fma3(si.n, mag, si.p);
}
HAR_HD void spawn_ray_to(const SurfInt &si, Vec3 target, Vec3 &o, Vec3 &d, float &maxt) {
    o = offset_p(si, target - si.p);
    Vec3 dd = target - o;
    float dist = norm3(dd);
    d = div3(dd, dist);
    maxt = dist * (1.f - HAR_SHADOW_EPS);
}

/* dr::Texture<Float, 2>::eval (call site bitmap.cpp:842-850; Dr.Jit's texture.h is NOT IN TREE -- parity unpinned): texel centres at (i + 1/2) / res; the integer
 * texel index is wrapped by WrapMode Repeat (i mod res), Clamp (clip to [0, res - 1]) or Mirror (the image flipped every other repetition, so that index -1 is
 * texel 0 and index res is texel res - 1); FilterMode::Nearest takes the texel under floor(uv * res), Linear blends the four around uv * res - 1/2. */
struct TexTaps { uint32_t idx[4]; float w0x, w1x, w0y, w1y; };
HAR_HD int32_t tex_wrap(int32_t i, int32_t n, uint32_t mode) {
    if (mode & 4u) return i < 0 ? 0 : (i > n - 1 ? n - 1 : i);                       /* clamp */
    const int32_t shifted = i < 0 ? i + 1 : i, div = shifted / n;                     /* truncating division, as the reference's integer divisor */
    int32_t mod = i - div * n;
    if (mod < 0) mod += n;
    if ((mode & 2u) && (((div & 1) == 0) == (i < 0))) mod = n - 1 - mod;              /* mirror: flip unless (even repetition) xor (negative side) */
    return mod;
}
HAR_HD void tex_taps(const DTexture &T, float u, float v, TexTaps &l) {
    const int32_t W = (int32_t) T.w, H = (int32_t) T.h;
    if (T.mode & HAR_TEX_HAS_UV_XF) {                                                    /* uv = m_transform * si.uv (bitmap.cpp:847; AffineTransform * Point, transform.h:322-335) */
        const float tu = fma_(T.uvm[1], v, fma_(T.uvm[0], u, T.uvm[2])), tv = fma_(T.uvm[4], v, fma_(T.uvm[3], u, T.uvm[5]));
        u = tu; v = tv;
    }
    if (T.mode & 1u) {                                                                 /* nearest: one texel, written as four coincident taps of weight (1, 0) */
        const int32_t x = tex_wrap((int32_t) floorf(u * (float) T.w), W, T.mode), y = tex_wrap((int32_t) floorf(v * (float) T.h), H, T.mode);
        l.w1x = 0.f; l.w1y = 0.f; l.w0x = 1.f; l.w0y = 1.f;
        l.idx[0] = l.idx[1] = l.idx[2] = l.idx[3] = (uint32_t) (y * W + x);
        return;
    }
    float px = fma_(u, (float) T.w, -0.5f), py = fma_(v, (float) T.h, -0.5f);
    float fx = floorf(px), fy = floorf(py);
    int32_t ix = (int32_t) fx, iy = (int32_t) fy;
    l.w1x = px - fx; l.w1y = py - fy; l.w0x = 1.f - l.w1x; l.w0y = 1.f - l.w1y;
    int32_t x0, x1, y0, y1;
    if ((T.mode & 7u) == 0u) {                                                         /* bilinear + repeat, the defaults: the round-1 code path */
        x0 = ix % W; if (x0 < 0) x0 += W;
        x1 = (ix + 1) % W; if (x1 < 0) x1 += W;
        y0 = iy % H; if (y0 < 0) y0 += H;
        y1 = (iy + 1) % H; if (y1 < 0) y1 += H;
    } else { x0 = tex_wrap(ix, W, T.mode); x1 = tex_wrap(ix + 1, W, T.mode); y0 = tex_wrap(iy, H, T.mode); y1 = tex_wrap(iy + 1, H, T.mode); }
    l.idx[0] = (uint32_t) (y0 * W + x0); l.idx[1] = (uint32_t) (y0 * W + x1);
    l.idx[2] = (uint32_t) (y1 * W + x0); l.idx[3] = (uint32_t) (y1 * W + x1);
}
HAR_HD Vec3 tex_fetch(const DTexture &T, const TexTaps &l) {
    float out[3];
    if (T.mode & 1u) { for (int c = 0; c < 3; ++c) out[c] = T.data[3 * (size_t) l.idx[0] + c]; return Vec3(out[0], out[1], out[2]); }
    for (int c = 0; c < 3; ++c) {
        float v00 = T.data[3 * (size_t) l.idx[0] + c], v10 = T.data[3 * (size_t) l.idx[1] + c];
        float v01 = T.data[3 * (size_t) l.idx[2] + c], v11 = T.data[3 * (size_t) l.idx[3] + c];
        float v0 = fma_(l.w0x, v00, l.w1x * v10), v1 = fma_(l.w0x, v01, l.w1x * v11);
        out[c] = fma_(l.w0y, v0, l.w1y * v1);
    }
    return Vec3(out[0], out[1], out[2]);
}
/* Texture::eval for `reflectance`: srgb constant (src/spectra/srgb.cpp) or bitmap */
HAR_HD Vec3 bsdf_reflectance(const DScene &S, const DBsdf &B, float u, float v, TexTaps &taps) {
    if (B.texture < 0) return Vec3(B.r, B.g, B.b);
    const DTexture T = S.textures[B.texture];
    tex_taps(T, u, v, taps);
    return tex_fetch(T, taps);
}
/* evaluated colour parameters of a BSDF record at (u, v) */
HAR_HD BsdfInputs bsdf_inputs(const DScene &S, const DBsdf &B, float u, float v, TexTaps &taps) {
    BsdfInputs in;
    in.slot0 = bsdf_reflectance(S, B, u, v, taps);
    in.slot1 = Vec3(B.r2, B.g2, B.b2);
    in.table = B.table >= 0 ? S.bsdf_tables + B.table : nullptr;
    return in;
}

/* TwoSidedBRDF (src/bsdfs/twosided.cpp:112-270): which record serves this side, and how wi / wo are mirrored.
 * Returns false when the BSDF is zero on this side (two different nested BSDFs and wi.z == 0). */
struct BsdfSide { uint32_t index; Vec3 wi; float wo_sign; };
HAR_HD bool bsdf_side(const DScene &S, uint32_t index, Vec3 wi, BsdfSide &side) {
    const DBsdf &B = S.bsdfs[index];
    side.index = index; side.wi = wi; side.wo_sign = 1.f;
    if (!(B.flags & BF_TWOSIDED)) return true;
    if (B.back < 0) {                      /* m_brdf[0] == m_brdf[1]: wo.z = mulsign(wo.z, wi.z); wi.z = abs(wi.z) */
        side.wo_sign = sign_(wi.z); side.wi.z = fabsf(wi.z);
        return true;
    }
    if (wi.z > 0.f) return true;
    if (wi.z < 0.f) { side.index = (uint32_t) B.back; side.wi.z = -wi.z; side.wo_sign = -1.f; return true; }
    return false;
}
/* The context a nested BSDF of a `twosided` record sees (twosided.cpp:129-146, 164-180): the front side and a one-BSDF `twosided` get the caller's context as it
 * is; the back side of a two-BSDF pair gets `component - component_count(front)` (uint32 arithmetic, as in the reference) unless the component is "all" */
HAR_HD BsdfCtx bsdf_side_ctx(const DScene &S, uint32_t index, const BsdfSide &side, BsdfCtx ctx) {
    const DBsdf &B = S.bsdfs[index];
    if ((B.flags & BF_TWOSIDED) && B.back >= 0 && side.wo_sign < 0.f && ctx.component != 0xffffffffu) ctx.component -= bsdf_component_count(B.type);
    return ctx;
}
template <uint32_t TYPES = HAR_BSDF_ALL_TYPES, bool CTX = false>
HAR_HD void bsdf_eval_pdf(const DScene &S, const BsdfSide &side, const BsdfInputs &in, bool side_ok, Vec3 wo, BsdfEval &e, const BsdfCtx &ctx = BsdfCtx()) {
    if (!side_ok) { e.value = Vec3(0.f); e.pdf = 0.f; e.d_slot0 = Vec3(0.f); e.d_slot1 = Vec3(0.f); return; }
    bsdf_eval_pdf_one<TYPES, CTX>(S.bsdfs[side.index], in, side.wi, Vec3(wo.x, wo.y, wo.z * side.wo_sign), e, ctx);
}
/* d value / d {alpha, eta, k} of the record serving this side (see bsdf_eval_extra_one) */
HAR_HD void bsdf_eval_extra(const DScene &S, const BsdfSide &side, const BsdfInputs &in, bool side_ok, Vec3 wo, BsdfEvalExtra &x) {
    if (!side_ok) { x.d_alpha_u = Vec3(0.f); x.d_alpha_v = Vec3(0.f); x.d_eta = Vec3(0.f); x.d_k = Vec3(0.f); return; }
    bsdf_eval_extra_one(S.bsdfs[side.index], in, side.wi, Vec3(wo.x, wo.y, wo.z * side.wo_sign), x);
}
template <uint32_t TYPES = HAR_BSDF_ALL_TYPES, bool CTX = false>
HAR_HD void bsdf_sample(const DScene &S, const BsdfSide &side, const BsdfInputs &in, bool side_ok, float s1, float s2x, float s2y, BsdfSample &bs, const BsdfCtx &ctx = BsdfCtx()) {
    if (!side_ok) { bs.wo = Vec3(0.f); bs.pdf = 0.f; bs.weight = Vec3(0.f); bs.eta = 0.f; bs.delta = false; bs.type = 0u; bs.comp = 0u; return; }
    bsdf_sample_one<TYPES, CTX>(S.bsdfs[side.index], in, side.wi, s1, s2x, s2y, bs, ctx);
    bs.wo.z *= side.wo_sign;
}

/* SmoothDiffuse::eval_pdf / sample (src/bsdfs/diffuse.cpp:159-179, 100-124) */
HAR_HD void diffuse_eval_pdf(Vec3 refl, Vec3 wi, Vec3 wo, Vec3 &value, float &pdf) {
    bool active = wi.z > 0.f && wo.z > 0.f;
    value = active ? (refl * HAR_INV_PI) * wo.z : Vec3(0.f);
    pdf = active ? HAR_INV_PI * wo.z : 0.f;
}
HAR_HD void diffuse_sample(Vec3 refl, Vec3 wi, float s2x, float s2y, Vec3 &wo, float &pdf, Vec3 &weight) {
    wo = square_to_cosine_hemisphere(s2x, s2y);
    pdf = HAR_INV_PI * wo.z;
    weight = (wi.z > 0.f && pdf > 0.f) ? refl : Vec3(0.f);
}

/* AreaLight::sample_direction (src/emitters/area.cpp:118-168), Shape::sample_direction
 * (src/render/shape.cpp:93-110), Rectangle::sample_position (src/shapes/rectangle.cpp:159-173) */
struct DirSample { Vec3 p, n, d; float dist, pdf; };
#define HAR_INV_FOUR_PI 0.07957747154594766788f
/* warp::square_to_uniform_sphere (include/mitsuba/core/warp.h:250-255) */
HAR_HD Vec3 square_to_uniform_sphere(float sx, float sy) {
    float z = fnma_(2.f, sy, 1.f), r = sqrtf(fmaxf(fnma_(z, z, 1.f), 0.f));
    float s, c; sincos_(2.f * HAR_PI * sx, s, c);
    return Vec3(r * c, r * s, z);
}
/* ---- environment map (src/emitters/envmap.cpp) ------------------------------------------------------------------------- */
HAR_HD float clip01_(float x) { return fminf(fmaxf(x, 0.f), 1.f); }
/* warp::interval_to_linear (include/mitsuba/core/warp.h:446-453) */
HAR_HD float interval_to_linear(float v0, float v1, float sample) {
    if (fabsf(v0 - v1) > 1e-4f * (v0 + v1))
        return (v0 - sqrtf(fmaxf(lerp_(v0 * v0, v1 * v1, sample), 0.f))) / (v0 - v1);
    return sample;
}
HAR_HD uint32_t hier_index(uint32_t x, uint32_t y, uint32_t width) { return ((x & 1u) | (((x & ~1u) | (y & 1u)) << 1)) + ((y & ~1u) * width); }
/* Hierarchical2D<Float, 0>::sample (distr_2d.h:520-600) + warp::square_to_bilinear (warp.h:478-494) */
HAR_HD void hier2d_sample(const DEnvmap &E, float sx, float sy, float &u, float &v, float &pdf) {
    sx = clip01_(sx); sy = clip01_(sy);
    uint32_t ox = 0, oy = 0;
    for (int l = (int) E.n_levels - 1; l > 0; --l) {
        ox <<= 1; oy <<= 1;
        const float *q = E.warp + (((E.lvl_offset[l] + hier_index(ox, oy, E.lvl_width[l])) >> 2) << 2);
        const float v00 = q[0], v10 = q[1], v01 = q[2], v11 = q[3];
        sx = clip01_(sx); sy = clip01_(sy);
        const float r0 = v00 + v10, r1 = v01 + v11;
        sy *= r0 + r1;
        const bool ym = sy > r0;
        if (ym) { oy += 1u; sy -= r0; }
        const float dy = ym ? r1 : r0, c0 = ym ? v01 : v00, c1 = ym ? v11 : v10;
        sx *= dy;
        const bool xm = sx > c0;
        if (xm) { sx -= c0; ox += 1u; }
        const float dx = xm ? c1 : c0;
        const float inv = rcp_(dy * dx);
        sy *= dx * inv; sx *= dy * inv;
    }
    const uint32_t W0 = E.lvl_width[0];
    const float *d = E.warp + E.lvl_offset[0] + ox + oy * W0;
    const float v00 = d[0], v10 = d[1], v01 = d[W0], v11 = d[W0 + 1];
    const float r0 = v00 + v10, r1 = v01 + v11;
    sy = interval_to_linear(r0, r1, sy);
    const float c0 = lerp_(v00, v01, sy), c1 = lerp_(v10, v11, sy);
    sx = interval_to_linear(c0, c1, sx);
    pdf = lerp_(c0, c1, sx);
    u = ((float) (int32_t) ox + sx) * (1.f / (float) E.w);                  /* m_patch_size = 1 / (size - 1), size = (W + 1, H) */
    v = ((float) (int32_t) oy + sy) * (1.f / (float) (E.h - 1u));
}
/* Hierarchical2D::eval (distr_2d.h:700-725) */
HAR_HD float hier2d_eval(const DEnvmap &E, float px, float py) {
    px = clip01_(px) * (float) E.w; py = clip01_(py) * (float) (E.h - 1u);
    uint32_t ox = (uint32_t) (int32_t) px, oy = (uint32_t) (int32_t) py;
    if (ox > E.w - 1u) ox = E.w - 1u;
    if (oy > E.h - 2u) oy = E.h - 2u;
    px -= (float) (int32_t) ox; py -= (float) (int32_t) oy;
    const uint32_t W0 = E.lvl_width[0];
    const float *d = E.warp + E.lvl_offset[0] + ox + oy * W0;
    return lerp_(lerp_(d[0], d[1], px), lerp_(d[W0], d[W0 + 1], px), py);
}
/* eval_spectrum, RGB branch (envmap.cpp:531-548,589-597): dr::Texture bilinear lookup with WrapMode::Clamp on the halo'ed storage */
HAR_HD Vec3 envmap_eval_uv(const DEnvmap &E, float u_, float v_) {
    const float rx = (float) E.w, ry = (float) E.h;
    const float u = u_ - floorf(u_), v = clip01_(v_);
    const float pos_x = fma_(u, rx, 1.f) / (rx + 2.f), pos_y = fma_(v, ry - 1.f, 0.5f) / ry;
    const int32_t sw = (int32_t) E.w + 2, H = (int32_t) E.h;
    const float px = fma_(pos_x, (float) sw, -0.5f), py = fma_(pos_y, ry, -0.5f);
    const float fx = floorf(px), fy = floorf(py);
    const int32_t ix = (int32_t) fx, iy = (int32_t) fy;
    const float w1x = px - fx, w1y = py - fy, w0x = 1.f - w1x, w0y = 1.f - w1y;
    const int32_t x0 = ix < 0 ? 0 : (ix > sw - 1 ? sw - 1 : ix), x1 = ix + 1 < 0 ? 0 : (ix + 1 > sw - 1 ? sw - 1 : ix + 1),
                  y0 = iy < 0 ? 0 : (iy > H - 1 ? H - 1 : iy), y1 = iy + 1 < 0 ? 0 : (iy + 1 > H - 1 ? H - 1 : iy + 1);
    const float *p00 = E.tex + 3 * ((size_t) y0 * sw + x0), *p10 = E.tex + 3 * ((size_t) y0 * sw + x1),
                *p01 = E.tex + 3 * ((size_t) y1 * sw + x0), *p11 = E.tex + 3 * ((size_t) y1 * sw + x1);
    float out[3];
    for (int c = 0; c < 3; ++c) {
        const float a = fma_(w0x, p00[c], w1x * p10[c]), b = fma_(w0x, p01[c], w1x * p11[c]);
        out[c] = fma_(w0y, a, w1y * b) * E.scale;
    }
    return Vec3(out[0], out[1], out[2]);
}
HAR_HD void envmap_direction_to_uv(Vec3 d, float &u, float &v) {                                   /* envmap.cpp:454-459 */
    u = atan2_(d.x, -d.z) * (0.5f * HAR_INV_PI);
    v = acos_(fminf(fmaxf(d.y, -1.f), 1.f)) * HAR_INV_PI;
}
/* EnvironmentMapEmitter::eval (envmap.cpp:228-236): radiance arriving along world direction d (= -si.wi of the escaping ray) */
HAR_HD Vec3 envmap_eval(const DEnvmap &E, Vec3 d_world) {
    float u, v; envmap_direction_to_uv(xf_vector(E.to_local, d_world), u, v);
    return envmap_eval_uv(E, u, v);
}
/* EnvironmentMapEmitter::pdf_direction (envmap.cpp:325-339) */
HAR_HD float envmap_pdf_direction(const DEnvmap &E, Vec3 d_world) {
    const Vec3 d = xf_vector(E.to_local, d_world);
    float u, v; envmap_direction_to_uv(d, u, v);
    u -= .5f / (float) E.w;
    u -= floorf(u); v -= floorf(v);
    const float inv_sin_theta = rsqrt_(fmaxf(d.x * d.x + d.z * d.z, 0x1p-24f * 0x1p-24f));
    return hier2d_eval(E, u, v) * inv_sin_theta * (1.f / (2.f * (HAR_PI * HAR_PI)));
}
/* EnvironmentMapEmitter::sample_direction (envmap.cpp:284-323) */
HAR_HD void envmap_sample_direction(const DEnvmap &E, Vec3 ref_p, float sx, float sy, struct DirSample &ds, Vec3 &spec);

/* PointLight::sample_direction (src/emitters/point.cpp:119-148): emitter type 4, position in to_world[9..11], `radiance` = radiant intensity; pdf 1, delta */
HAR_HD void point_sample_direction(const DEmitter &E, Vec3 ref_p, DirSample &ds, Vec3 &spec, float *unit = nullptr) {
    ds.p = Vec3(E.to_world[9], E.to_world[10], E.to_world[11]); ds.n = Vec3(0.f); ds.pdf = 1.f;
    ds.d = ds.p - ref_p;
    const float dist2 = dot3(ds.d, ds.d), inv_dist = rsqrt_(dist2);
    ds.dist = sqrtf(dist2);
    ds.d = ds.d * inv_dist;
    const float w = inv_dist * inv_dist;
    spec = Vec3(E.radiance[0], E.radiance[1], E.radiance[2]) * w;
    if (unit) *unit = w;
}
/* SpotLight::sample_direction (src/emitters/spot.cpp:177-211) with falloff_curve (:143-151): emitter type 5; the record holds the linear part of to_world^-1 in
 * to_world[0..8], the position in [9..11], normal = (cutoff angle in radians, its cosine, the beam width's cosine), inv_area = 1 / transition width */
HAR_HD void spot_sample_direction(const DEmitter &E, Vec3 ref_p, DirSample &ds, Vec3 &spec, float *unit = nullptr) {
    ds.p = Vec3(E.to_world[9], E.to_world[10], E.to_world[11]); ds.n = Vec3(0.f); ds.pdf = 1.f;
    ds.d = ds.p - ref_p;
    ds.dist = norm3(ds.d);
    const float inv_dist = rcp_(ds.dist);
    ds.d = ds.d * inv_dist;
    const Vec3 local_dir = normalize3(xf_vector(E.to_world, -ds.d));            /* to_world[0..8] = the inverse's linear part: a vector needs no more */
    const float cos_theta = local_dir.z;
    const float beam_res = cos_theta >= E.normal[2] ? 1.f : (E.normal[0] - acos_(cos_theta)) * E.inv_area;
    const float falloff = cos_theta > E.normal[1] ? beam_res : 0.f;
    const bool active = falloff > 0.f;
    const float w = falloff * (inv_dist * inv_dist);
    spec = active ? Vec3(E.radiance[0], E.radiance[1], E.radiance[2]) * w : Vec3(0.f);
    if (unit) *unit = active ? w : 0.f;
}
/* DirectionalEmitter::sample_direction (src/emitters/directional.cpp:149-176): emitter type 6; the record holds the direction of travel in to_world[0..2], the scene's
 * bounding sphere (set_scene, :99-109) in [3..5] (centre) and [6] (radius); `radiance` = the irradiance */
HAR_HD void directional_sample_direction(const DEmitter &E, Vec3 ref_p, DirSample &ds, Vec3 &spec, float *unit = nullptr) {
    const Vec3 d(E.to_world[0], E.to_world[1], E.to_world[2]);
    const float radius = fmaxf(E.to_world[6], norm3(ref_p - Vec3(E.to_world[3], E.to_world[4], E.to_world[5])));
    const float dist = 2.f * radius;
    ds.p = ref_p - d * dist; ds.n = d; ds.pdf = 1.f; ds.d = -d; ds.dist = dist;
    spec = Vec3(E.radiance[0], E.radiance[1], E.radiance[2]);
    if (unit) *unit = 1.f;
}
/* `unit` (optional): the weight the sample would carry for a unit radiance, i.e. d spec / d radiance */
HAR_HD void emitter_sample_direction(const DEmitter &E, Vec3 ref_p, float sx, float sy, DirSample &ds, Vec3 &spec, float *unit = nullptr) {
    if (E.type == 1u) {                                     /* ConstantBackgroundEmitter::sample_direction, constant.cpp:127-153 */
        Vec3 d = square_to_uniform_sphere(sx, sy);
        Vec3 c(E.to_world[0], E.to_world[1], E.to_world[2]);
        float radius = fmaxf(E.to_world[3], norm3(ref_p - c)), dist = 2.f * radius;
        ds.p = fma3(d, dist, ref_p); ds.n = -d; ds.pdf = HAR_INV_FOUR_PI; ds.d = d; ds.dist = dist;
        spec = div3(Vec3(E.radiance[0], E.radiance[1], E.radiance[2]), ds.pdf);
        if (unit) *unit = rcp_(ds.pdf);
        return;
    }
    ds.p = xf_point(E.to_world, Vec3(fma_(sx, 2.f, -1.f), fma_(sy, 2.f, -1.f), 0.f));
    ds.n = Vec3(E.normal[0], E.normal[1], E.normal[2]);
    ds.pdf = E.inv_area;
    ds.d = ds.p - ref_p;
    float dist2 = dot3(ds.d, ds.d);
    ds.dist = sqrtf(dist2);
    ds.d = div3(ds.d, ds.dist);
    float dp = fabsf(dot3(ds.d, ds.n));
    float x = dist2 / dp;
    ds.pdf *= finite_(x) ? x : 0.f;
    bool active = dot3(ds.d, ds.n) < 0.f && ds.pdf != 0.f;
    spec = active ? div3(Vec3(E.radiance[0], E.radiance[1], E.radiance[2]), ds.pdf) : Vec3(0.f);
    if (unit) *unit = active ? rcp_(ds.pdf) : 0.f;
}
HAR_HD void envmap_sample_direction(const DEnvmap &E, Vec3 ref_p, float sx, float sy, DirSample &ds, Vec3 &spec) {
    float u, v, pdf; hier2d_sample(E, sx, sy, u, v, pdf);
    u += .5f / (float) E.w;
    const bool active = pdf > 0.f;
    float st, ct, sp, cp; sincos_(v * HAR_PI, st, ct); sincos_(u * (2.f * HAR_PI), sp, cp);
    const float inv_sin_theta = rcp_(fmaxf(st, 0x1p-24f));
    const Vec3 dl(sp * st, ct, -cp * st);
    const Vec3 c(E.center[0], E.center[1], E.center[2]);
    const float radius = fmaxf(E.radius, norm3(ref_p - c)), dist = 2.f * radius;
    const Vec3 d = xf_vector(E.to_world, dl);
    ds.p = fma3(d, dist, ref_p); ds.n = -d; ds.d = d; ds.dist = dist;
    ds.pdf = active ? pdf * inv_sin_theta * (1.f / (2.f * (HAR_PI * HAR_PI))) : 0.f;
    spec = active ? div3(envmap_eval_uv(E, u, v), ds.pdf) : Vec3(0.f);
}
/* DiscreteDistribution::sample / sample_reuse_pmf (include/mitsuba/core/distr_1d.h:117-140,205-219): value *= sum; dr::binary_search over [lo, hi] (m_valid) with the
 * predicate of the variant -- JIT: (cdf[i] < value || cdf[i] == 0) && cdf[i] != sum, scalar: cdf[i] < value -- then the index, the re-used sample
 * (value - cdf_normalized[index - 1]) / pmf_normalized[index] and the normalised pmf */
HAR_HD uint32_t discrete_sample_reuse_pmf(const float *pmf, const float *cdf, uint32_t lo, uint32_t hi, float sum, float normalization, float value01, bool jit,
                                          float &reused, float &pmf_out) {
    const float value = value01 * sum;
    uint32_t start = lo, end = hi, iterations = 0;
    if (start < end) { uint32_t span = end - start; iterations = 1; while (span >>= 1) ++iterations; }
    for (uint32_t i = 0; i < iterations; ++i) {
        const uint32_t middle = (start + end) >> 1;
        const float c = cdf[middle];
        const bool cond = jit ? (((c < value) || c == 0.f) && c != sum) : (c < value);
        if (cond) start = middle + 1u < end ? middle + 1u : end; else end = middle;
    }
    const float pmf_n = pmf[start] * normalization, cdf_n = start > 0u ? cdf[start - 1u] * normalization : 0.f;
    reused = (value01 - cdf_n) / pmf_n; pmf_out = pmf_n;
    return start;
}
/* Scene::sample_emitter (src/render/scene.cpp:248-271) and Scene::pdf_emitter (:273-279): the emitter a uniform sample picks, the weight 1 / probability and the re-used
 * sample.  Uniform choice unless the scene holds a distribution over the emitters' sampling weights (DScene::emitter_distr).  `jit`: the variant's predicate of
 * DiscreteDistribution::sample.  No emitters: index 0xffffffff, weight 0 (:251-256). */
HAR_HD uint32_t scene_sample_emitter(const DScene &S, float sample, bool jit, float &weight, float &reused) {
    weight = 1.f; reused = sample;
    if (S.n_emitters < 2u) { if (S.n_emitters == 0u) weight = 0.f; return S.n_emitters ? 0u : 0xffffffffu; }
    if (S.emitter_distr != nullptr) {
        float p;
        const uint32_t index = discrete_sample_reuse_pmf(S.emitter_distr, S.emitter_distr + S.n_emitters, S.emitter_valid_lo, S.emitter_valid_hi, S.emitter_sum, S.emitter_norm,
                                                         sample, jit, reused, p);
        weight = rcp_(p);
        return index;
    }
    const float scaled = sample * (float) S.n_emitters;
    uint32_t index = (uint32_t) scaled; if (index > S.n_emitters - 1u) index = S.n_emitters - 1u;
    weight = (float) S.n_emitters; reused = scaled - (float) index;
    return index;
}
HAR_HD float scene_pdf_emitter(const DScene &S, uint32_t index) {
    if (S.emitter_distr == nullptr) return S.n_emitters ? 1.f / (float) S.n_emitters : 0.f;      /* m_emitter_pmf, scene.cpp:139 */
    return S.emitter_distr[index] * S.emitter_norm;                                             /* eval_pmf_normalized */
}
/* AreaLight::sample_direction on a triangle mesh: Mesh::sample_position (src/render/mesh.cpp:1662-1712) -- DiscreteDistribution::sample_reuse over the
 * face areas (distr_1d.h:117-183, JIT predicate, dr::binary_search over [0, n - 1]), warp::square_to_uniform_triangle (warp.h:153-156), interpolated
 * vertex normals when the mesh has them -- then Shape::sample_direction (shape.cpp:93-110) and the one-sided test of area.cpp:118-168 */
HAR_HD void mesh_emitter_sample_direction(const DScene &S, const DEmitter &E, Vec3 ref_p, float sx, float sy, DirSample &ds, Vec3 &spec, float *unit = nullptr) {
    const uint32_t off = as_u32(E.to_world[0]), nf = as_u32(E.to_world[1]);
    const float sum = E.to_world[2], normalization = E.inv_area;
    const float *pmf = S.emitter_cdf + off, *cdf = pmf + nf;
    const float value = sy * sum;
    uint32_t start = 0, end = nf - 1u, iterations = 0;
    if (start < end) { uint32_t span = end - start; iterations = 1; while (span >>= 1) ++iterations; }
    for (uint32_t i = 0; i < iterations; ++i) {
        const uint32_t middle = (start + end) >> 1;
        const float c = cdf[middle];
        const bool cond = ((c < value) || c == 0.f) && c != sum;
        if (cond) start = middle + 1u < end ? middle + 1u : end; else end = middle;
    }
    const uint32_t idx = start;
    const float pmf_n = pmf[idx] * normalization, cdf_n = idx > 0 ? cdf[idx - 1u] * normalization : 0.f;
    sy = (sy - cdf_n) / pmf_n;
    const DMesh M = S.meshes[E.mesh];
    const uint32_t *f = S.faces + 4 * ((size_t) M.foff + idx);
    const float *v0 = S.verts + 8 * ((size_t) M.voff + f[0]), *v1 = S.verts + 8 * ((size_t) M.voff + f[1]), *v2 = S.verts + 8 * ((size_t) M.voff + f[2]);
    const Vec3 p0(v0[0], v0[1], v0[2]), e0 = Vec3(v1[0], v1[1], v1[2]) - p0, e1 = Vec3(v2[0], v2[1], v2[2]) - p0;
    const float t = sqrtf(fmaxf(1.f - sx, 0.f)), bx = 1.f - t, by = t * sy;
    ds.p = fma3(e0, bx, fma3(e1, by, p0));
    Vec3 n;
    if (M.flags & 1u) n = fma3(Vec3(v0[3], v0[4], v0[5]), 1.f - bx - by, fma3(Vec3(v1[3], v1[4], v1[5]), bx, Vec3(v2[3], v2[4], v2[5]) * by));
    else n = cross3(e0, e1);
    ds.n = normalize3(n);
    ds.pdf = normalization;
    ds.d = ds.p - ref_p;
    const float dist2 = dot3(ds.d, ds.d);
    ds.dist = sqrtf(dist2);
    ds.d = div3(ds.d, ds.dist);
    const float dp = fabsf(dot3(ds.d, ds.n));
    const float x = dist2 / dp;
    ds.pdf *= finite_(x) ? x : 0.f;
    const bool active = dot3(ds.d, ds.n) < 0.f && ds.pdf != 0.f;
    spec = active ? div3(Vec3(E.radiance[0], E.radiance[1], E.radiance[2]), ds.pdf) : Vec3(0.f);
    if (unit) *unit = active ? rcp_(ds.pdf) : 0.f;
}
/* AreaLight::pdf_direction (area.cpp:170-197) / Shape::pdf_direction (shape.cpp:112-124) */
HAR_HD float emitter_pdf_direction(const DEmitter &E, Vec3 d, Vec3 n, float dist) {
    float dp = dot3(d, n);
    if (!(dp < 0.f)) return 0.f;
    float adp = fabsf(dp);
    float pdf = E.inv_area;
    pdf *= (adp != 0.f) ? (dist * dist) / adp : 0.f;
    return pdf;
}

/* ---- AreaLight with a spatially varying radiance (emitter type 7; area.cpp:133-165, 185-191) */
/* dr::binary_search(0, last, cdf[i] < value) */
HAR_HD uint32_t cdf_search(const float *cdf, uint32_t last, float value) {
    uint32_t start = 0, end = last, iterations = 0;
    if (start < end) { uint32_t span = end - start; iterations = 1; while (span >>= 1) ++iterations; }
    for (uint32_t i = 0; i < iterations; ++i) {
        const uint32_t middle = (start + end) >> 1;
        if (cdf[middle] < value) start = middle + 1u < end ? middle + 1u : end; else end = middle;
    }
    return start;
}
/* DiscreteDistribution2D::pdf (distr_2d.h:121-131) of texel (x, y): difference of neighbouring conditional sums * normalization */
HAR_HD float texel_pdf(const float *cond, uint32_t w, float normalization, uint32_t x, uint32_t y) {
    const size_t i = (size_t) y * w + x;
    return (cond[i] - (x > 0u ? cond[i - 1u] : 0.f)) * normalization;
}
/* BitmapTexture::pdf_texture (bitmap.cpp:673-703), position in the TEXTURE's parameterisation */
HAR_HD float bitmap_pdf_texture(const DTexture &T, const float *tab, float u, float v) {
    const float *cond = tab + HAR_TEXEL_TABLE_HEADER + T.h;
    const float texels = (float) ((int32_t) T.w * (int32_t) T.h);
    const int32_t W = (int32_t) T.w, H = (int32_t) T.h;
    if (T.mode & 1u) {
        const int32_t x = tex_wrap((int32_t) floorf(u * (float) T.w), W, T.mode), y = tex_wrap((int32_t) floorf(v * (float) T.h), H, T.mode);
        return texel_pdf(cond, T.w, tab[1], (uint32_t) x, (uint32_t) y) * texels;
    }
    const float px = fma_(u, (float) T.w, -0.5f), py = fma_(v, (float) T.h, -0.5f), fx = floorf(px), fy = floorf(py);
    const int32_t ix = (int32_t) fx, iy = (int32_t) fy;
    const float w1x = px - fx, w1y = py - fy, w0x = 1.f - w1x, w0y = 1.f - w1y;
    const uint32_t x0 = (uint32_t) tex_wrap(ix, W, T.mode), x1 = (uint32_t) tex_wrap(ix + 1, W, T.mode), y0 = (uint32_t) tex_wrap(iy, H, T.mode), y1 = (uint32_t) tex_wrap(iy + 1, H, T.mode);
    const float v00 = texel_pdf(cond, T.w, tab[1], x0, y0), v10 = texel_pdf(cond, T.w, tab[1], x1, y0), v01 = texel_pdf(cond, T.w, tab[1], x0, y1), v11 = texel_pdf(cond, T.w, tab[1], x1, y1);
    const float v0 = fma_(w0x, v00, w1x * v10), v1 = fma_(w0x, v01, w1x * v11);
    return fma_(w0y, v0, w1y * v1) * texels;
}
/* warp::interval_to_tent (warp.h:196-200) */
HAR_HD float interval_to_tent(float s) {
    s -= 0.5f;
    return mulsign_(1.f - sqrtf(fmaxf(fma_(fabsf(s), -2.f, 1.f), 0.f)), s);
}
/* BitmapTexture::sample_position (bitmap.cpp:622-660) over DiscreteDistribution2D::sample (distr_2d.h:141-180): surface uv + its density */
HAR_HD void bitmap_sample_position(const DTexture &T, const float *tab, float sx, float sy, float &u, float &v, float &pdf) {
    const float *marg = tab + HAR_TEXEL_TABLE_HEADER, *cond_all = marg + T.h;
    sx = fminf(fmaxf(sx, 1.17549435e-38f), 0x1.fffffep-1f); sy = fminf(fmaxf(sy, 1.17549435e-38f), 0x1.fffffep-1f);
    sy *= tab[0];
    const uint32_t row = cdf_search(marg, T.h - 1u, sy);
    const float *cond = cond_all + (size_t) row * T.w;
    sx *= cond[T.w - 1u];
    const uint32_t col = cdf_search(cond, T.w - 1u, sx);
    const float c0 = col > 0u ? cond[col - 1u] : 0.f, c1 = cond[col], r0 = row > 0u ? marg[row - 1u] : 0.f, r1 = marg[row];
    sx -= c0; sy -= r0;
    if (c1 != c0) sx = sx / (c1 - c0);
    if (r1 != r0) sy = sy / (r1 - r0);
    const float inv_w = 1.f / (float) T.w, inv_h = 1.f / (float) T.h;
    float qx, qy;
    if (T.mode & 1u) { qx = ((float) col + sx) * inv_w; qy = ((float) row + sy) * inv_h; }
    else {
        qx = ((float) col + 0.5f + interval_to_tent(sx)) * inv_w; qy = ((float) row + 0.5f + interval_to_tent(sy)) * inv_h;
        if (!(T.mode & 6u)) { if (qx < 0.f) qx += 1.f; if (qx > 1.f) qx -= 1.f; if (qy < 0.f) qy += 1.f; if (qy > 1.f) qy -= 1.f; }
        else { if (qx < 0.f) qx = -qx; if (qx > 1.f) qx = 2.f - qx; if (qy < 0.f) qy = -qy; if (qy > 1.f) qy = 2.f - qy; }
    }
    u = fma_(tab[3], qy, fma_(tab[2], qx, tab[4])); v = fma_(tab[6], qy, fma_(tab[5], qx, tab[7]));
    pdf = bitmap_pdf_texture(T, tab, qx, qy);
}
HAR_HD Vec3 texture_eval_uv(const DTexture &T, float u, float v) { TexTaps taps; tex_taps(T, u, v, taps); return tex_fetch(T, taps); }
/* AreaLight::sample_direction, spatially varying branch (area.cpp:133-165) with Rectangle::eval_parameterization (rectangle.cpp:215-237) */
/* `unit` / `uv_out` (optional, prb adjoint w.r.t. the bitmap's texels -- `radiance` is a differentiable traverse entry, area.cpp:64-70): the weight the sample carries per unit of
 * radiance(ds.uv), and ds.uv itself; the texel distribution the sample was drawn from stays detached (prb.py:174-175) */
HAR_HD void textured_area_sample_direction(const DScene &S, const DEmitter &E, Vec3 ref_p, float sx, float sy, DirSample &ds, Vec3 &spec, float *unit = nullptr, float *uv_out = nullptr) {
    const DTexture T = S.textures[as_u32(E.radiance[0])];
    const float *tab = S.emitter_cdf + as_u32(E.radiance[1]);
    float u, v, pdf;
    bitmap_sample_position(T, tab, sx, sy, u, v, pdf);
    if (uv_out) { uv_out[0] = u; uv_out[1] = v; }
    bool active = pdf != 0.f;
    ds.p = xf_point(E.to_world, Vec3(fma_(u, 2.f, -1.f), fma_(v, 2.f, -1.f), 0.f));
    ds.n = Vec3(E.normal[0], E.normal[1], E.normal[2]);
    ds.d = ds.p - ref_p;
    const float dist2 = dot3(ds.d, ds.d);
    ds.dist = sqrtf(dist2);
    ds.d = div3(ds.d, ds.dist);
    const float dp = dot3(ds.d, ds.n);
    active = active && dp < 0.f;
    ds.pdf = active ? pdf / E.radiance[2] * dist2 / -dp : 0.f;
    spec = active ? div3(texture_eval_uv(T, u, v), ds.pdf) : Vec3(0.f);
    if (unit) *unit = active ? rcp_(ds.pdf) : 0.f;
}
/* AreaLight::pdf_direction, spatially varying branch (area.cpp:185-191): pdf_position(ds.uv) * dist^2 / (|dp_du x dp_dv| * -dp) */
HAR_HD float textured_area_pdf_direction(const DScene &S, const DEmitter &E, Vec3 d, Vec3 n, float dist, float u, float v) {
    const float dp = dot3(d, n);
    if (!(dp < 0.f)) return 0.f;
    const DTexture T = S.textures[as_u32(E.radiance[0])];
    float tu = u, tv = v;
    if (T.mode & HAR_TEX_HAS_UV_XF) { tu = fma_(T.uvm[1], v, fma_(T.uvm[0], u, T.uvm[2])); tv = fma_(T.uvm[4], v, fma_(T.uvm[3], u, T.uvm[5])); }
    return bitmap_pdf_texture(T, S.emitter_cdf + as_u32(E.radiance[1]), tu, tv) * (dist * dist) / (E.radiance[2] * -dp);
}

/* PerspectiveCamera::sample_ray (src/sensors/perspective.cpp:200-237) */
HAR_HD void sensor_sample_ray(const DSensor &C, float px, float py, Vec3 &o, Vec3 &d, float &maxt) {
    const float *M = C.s2c;
    float r0 = M[3], r1 = M[7], r2 = M[11], r3 = M[15];
    px = px + C.ppo_x; py = py + C.ppo_y;                              /* perspective.cpp:216-221 */
    r0 = fma_(M[0], px, r0); r1 = fma_(M[4], px, r1); r2 = fma_(M[8], px, r2);  r3 = fma_(M[12], px, r3);
    r0 = fma_(M[1], py, r0); r1 = fma_(M[5], py, r1); r2 = fma_(M[9], py, r2);  r3 = fma_(M[13], py, r3);
    r0 = fma_(M[2], 0.f, r0); r1 = fma_(M[6], 0.f, r1); r2 = fma_(M[10], 0.f, r2); r3 = fma_(M[14], 0.f, r3);
    if (C.projection == 1u) {            /* OrthographicCamera::sample_ray (orthographic.cpp:131-157): near_p = sample_to_camera * p (affine), o = to_world * near_p, d = normalize(to_world * +z) */
        const float *T = C.to_world;
        Vec3 ow(fma_(T[0], r0, T[3]), fma_(T[4], r0, T[7]), fma_(T[8], r0, T[11]));
        ow = Vec3(fma_(T[1], r1, ow.x), fma_(T[5], r1, ow.y), fma_(T[9], r1, ow.z));
        o = Vec3(fma_(T[2], r2, ow.x), fma_(T[6], r2, ow.y), fma_(T[10], r2, ow.z));
        d = normalize3(Vec3(T[2], T[6], T[10]));
        maxt = C.far_clip - C.near_clip;
        return;
    }
    float iw = rcp_(r3);
    Vec3 dl = normalize3(Vec3(r0 * iw, r1 * iw, r2 * iw));
    const float *T = C.to_world;
    Vec3 dw(T[0] * dl.x, T[4] * dl.x, T[8] * dl.x);
    dw = Vec3(fma_(T[1], dl.y, dw.x), fma_(T[5], dl.y, dw.y), fma_(T[9], dl.y, dw.z));
    dw = Vec3(fma_(T[2], dl.z, dw.x), fma_(T[6], dl.z, dw.y), fma_(T[10], dl.z, dw.z));
    float inv_z = rcp_(dl.z);
    float near_t = C.near_clip * inv_z, far_t = C.far_clip * inv_z;
    d = dw;
    o = Vec3(T[3], T[7], T[11]) + dw * near_t;
    maxt = far_t - near_t;
}

/* GaussianFilter::eval, LLVM branch: max(estrin(x^2, coeff), 0) (src/rfilters/gaussian.cpp:57-101) */
HAR_HD float estrin10(float x, const float *c) {
    float x2 = x * x, x4 = x2 * x2, x8 = x4 * x4;
    float a0 = fma_(x, c[1], c[0]), a1 = fma_(x, c[3], c[2]), a2 = fma_(x, c[5], c[4]), a3 = fma_(x, c[7], c[6]), a4 = fma_(x, c[9], c[8]);
    float b0 = fma_(x2, a1, a0), b1 = fma_(x2, a3, a2);
    float c0 = fma_(x4, b1, b0);
    return fma_(x8, a4, c0);
}
/* ... and the other reconstruction filters: tent.cpp:54-56, mitchell.cpp:60-79, catmullrom.cpp:39-54, lanczos.cpp:57-67 */
HAR_HD float rfilter_eval(const DSensor &C, float x) {
    switch (C.rfilter) {
    case 2: return fmaxf(0.f, 1.f - fabsf(x * (1.f / C.radius)));
    case 3: {
        x = fabsf(x); const float x2 = x * x, x3 = x2 * x, B = C.rf_p0, Cc = C.rf_p1;
        const float a3 = (12.f - 9.f * B - 6.f * Cc), a2 = (-18.f + 12.f * B + 6.f * Cc), a0 = (6.f - 2.f * B),
                    b3 = (-B - 6.f * Cc), b2 = (6.f * B + 30.f * Cc), b1 = (-12.f * B - 48.f * Cc), b0 = (8.f * B + 24.f * Cc);
        const float r = (1.f / 6.f) * (x < 1.f ? fma_(a3, x3, fma_(a2, x2, a0)) : fma_(b3, x3, fma_(b2, x2, fma_(b1, x, b0))));
        return x < 2.f ? r : 0.f;
    }
    case 4: {
        x = fabsf(x); const float x2 = x * x, x3 = x2 * x, B = 0.f, Cc = .5f;
        const float r = (1.f / 6.f) * (x < 1.f ? (12.f - 9.f * B - 6.f * Cc) * x3 + (-18.f + 12.f * B + 6.f * Cc) * x2 + (6.f - 2.f * B)
                                               : (-B - 6.f * Cc) * x3 + (6.f * B + 30.f * Cc) * x2 + (-12.f * B - 48.f * Cc) * x + (8.f * B + 24.f * Cc));
        return x < 2.f ? r : 0.f;
    }
    case 5: {
        x = fabsf(x);
        const float x1 = HAR_PI * x, x2 = x1 / C.radius;
        float s1, c1, s2, c2; sincos_(x1, s1, c1); sincos_(x2, s2, c2);
        const float r = (s1 * s2) / (x1 * x2);
        return x < HAR_EPSILON ? 1.f : (x > C.radius ? 0.f : r);
    }
    default: return fmaxf(estrin10(x * x, C.coeff), 0.f);
    }
}

/* lane -> pixel/sample mapping of SamplingIntegrator::render (integrator.cpp:322-339) and
 * render_sample (:448-466): returns integer pixel (crop-relative) and jittered film position */
struct LaneSample { float pos_x, pos_y, ipos_x, ipos_y; };
HAR_HD LaneSample lane_sample(const DSensor &C, uint32_t idx, uint32_t spp, uint32_t log_spp, float jx, float jy) {
    uint32_t p = (log_spp != 0xffffffffu) ? (idx >> log_spp) : (idx / spp);
    uint32_t y = p / C.samp_w, x = p - C.samp_w * y;                 /* integrator.cpp:330-331 over film_size = crop_size + 2 * border_size */
    LaneSample L;
    /* pos -= border_size; pos += crop_offset (integrator.cpp:333-336): a Vector2i, negative in the border of a crop window at the film's origin */
    L.ipos_x = (float) ((int32_t) (x + C.crop_x) - (int32_t) C.border); L.ipos_y = (float) ((int32_t) (y + C.crop_y) - (int32_t) C.border);
    L.pos_x = L.ipos_x + jx; L.pos_y = L.ipos_y + jy;
    return L;
}
HAR_HD void lane_camera_ray(const DSensor &C, const LaneSample &L, Vec3 &o, Vec3 &d, float &maxt) {
    float sx = 1.f / (float) C.crop_w, sy = 1.f / (float) C.crop_h;
    float ox = -(float) C.crop_x * sx, oy = -(float) C.crop_y * sy;
    sensor_sample_ray(C, fma_(L.pos_x, sx, ox), fma_(L.pos_y, sy, oy), o, d, maxt);
}

} // namespace har
