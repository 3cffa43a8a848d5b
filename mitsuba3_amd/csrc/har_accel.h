/*
 * har_accel.h -- compressed 8-wide BVH (two-level) for the hip_ad_rgb path.
 *
 * MI355X has no ray-tracing hardware, and the reference's accel backends
 * (Embree / OptiX / Metal; include/mitsuba/render/accel_*.h) are third-party
 * code that is not in the tree, so this traversal is new work.  Design:
 *   - 80-byte nodes with 8 children whose boxes are quantised to 8 bits per
 *     plane relative to the parent box (after Ylitie, Karras & Laine 2017):
 *     five 16-byte loads fetch 8 child boxes, vs. 8 x 24 bytes uncompressed;
 *   - children sit in slots ordered by ray octant so that a `clz` over the hit
 *     mask visits them front-to-back without sorting distances;
 *   - leaves reference runs of <= 3 pre-gathered 48-byte triangle records
 *     {p0, e1, e2, prim, shape}: three 16-byte loads, no index indirection;
 *   - one TLAS over InstanceEntry-like records (scene_ir.h:128-138) whose
 *     leaves switch the ray to object space (Instance::ray_intersect_preliminary,
 *     src/shapes/instance.cpp:121-132) and descend into the ShapeGroup's BLAS;
 *   - traversal state is two 64-bit "groups" in registers plus a short stack of
 *     8-byte entries that the kernels keep in LDS (one column per lane).
 * The exact triangle test is Mesh::moeller_trumbore (mesh.h:1130-1155), so the
 * BVH only prunes: results equal the brute-force kernel bit for bit.
 */
#pragma once
#include "har_math.h"

namespace har {

struct alignas(16) Node8 {
    float px, py, pz;
    uint8_t ex, ey, ez, imask;
    uint32_t child_base, tri_base;
    uint8_t meta[8];
    uint8_t qlox[8], qloy[8], qloz[8], qhix[8], qhiy[8], qhiz[8];
};
static_assert(sizeof(Node8) == 80, "Node8 must be 80 bytes");

struct alignas(16) TriRec { float p0x, p0y, p0z, e1x, e1y, e1z, e2x, e2y, e2z; uint32_t prim, shape, pad; };
static_assert(sizeof(TriRec) == 48, "TriRec must be 48 bytes");

struct alignas(16) InstRec {
    float to_world[12];
    float to_object[12];
    uint32_t blas_root;   /* node index of the BLAS root */
    uint32_t inst_index;  /* 0xffffffff: top-level geometry (identity) */
    uint32_t identity;
    uint32_t pad;
};

struct Accel {
    const Node8   *nodes;
    const TriRec  *tris;
    const InstRec *insts;      /* TLAS leaf records */
    uint32_t root;             /* TLAS root if has_tlas, else BLAS root of the top-level geometry */
    uint32_t has_tlas;
    uint32_t n_tris, n_insts;
};

struct Hit {
    float t, u, v;
    uint32_t prim, shape, inst;
};

/* Mesh::moeller_trumbore, include/mitsuba/render/mesh.h:1130-1155 (same op order) */
HAR_HD bool moeller_trumbore(Vec3 o, Vec3 d, float maxt, Vec3 p0, Vec3 e1, Vec3 e2, float &t, float &u, float &v) {
    Vec3 pvec = cross3(d, e2);
    float inv_det = rcp_(dot3(e1, pvec));
    Vec3 tvec = o - p0;
    u = dot3(tvec, pvec) * inv_det;
    bool active = u >= 0.f && u <= 1.f;
    Vec3 qvec = cross3(tvec, e1);
    v = dot3(d, qvec) * inv_det;
    active = active && v >= 0.f && u + v <= 1.f;
    t = dot3(e2, qvec) * inv_det;
    active = active && t >= 0.f && t <= maxt;
    return active;
}

/* closest-hit update with the tie rule of ShapeKDTree::ray_intersect_naive
 * (kdtree.h:2433-2460): on exactly equal t the later (inst, shape, prim) wins */
HAR_HD void hit_update(Hit &h, float t, float u, float v, uint32_t prim, uint32_t shape, uint32_t inst) {
    if (t == h.t) {
        uint32_t a0 = inst + 1u, b0 = h.inst + 1u;
        bool later = a0 != b0 ? a0 > b0 : (shape != h.shape ? shape > h.shape : prim > h.prim);
        if (!later) return;
    }
    h.t = t; h.u = u; h.v = v; h.prim = prim; h.shape = shape; h.inst = inst;
}

HAR_HD uint32_t clz32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t) __clz((int) x);
#else
    return x ? (uint32_t) __builtin_clz(x) : 32u;
#endif
}
HAR_HD uint32_t popc32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t) __popc(x);
#else
    return (uint32_t) __builtin_popcount(x);
#endif
}

struct RaySetup { Vec3 o, d, idir; uint32_t octinv; };
HAR_HD RaySetup ray_setup(Vec3 o, Vec3 d) {
    RaySetup r; r.o = o; r.d = d;
    const float eps = 1e-30f;
    r.idir = Vec3(1.f / (fabsf(d.x) > eps ? d.x : mulsign_(eps, d.x)),
                  1.f / (fabsf(d.y) > eps ? d.y : mulsign_(eps, d.y)),
                  1.f / (fabsf(d.z) > eps ? d.z : mulsign_(eps, d.z)));
    r.octinv = (d.x < 0.f ? 0u : 4u) | (d.y < 0.f ? 0u : 2u) | (d.z < 0.f ? 0u : 1u);
    return r;
}

#define HAR_STACK_OVERFLOW 0x7fffffff

/*
 * Stack concept: void push(int level, uint32_t x, uint32_t y); void pop(int level, uint32_t &x, uint32_t &y);
 * static constexpr int Capacity.
 * Returns true if (AnyHit and an intersection exists) or (closest hit found).
 * `status` is set to HAR_STACK_OVERFLOW if the stack capacity was exceeded.
 */
template <bool AnyHit, typename Stack>
HAR_HD bool accel_trace(const Accel &A, Vec3 o_w, Vec3 d_w, float maxt, Hit &hit, Stack &stack, int &status) {
    hit.t = HAR_INF; hit.u = 0.f; hit.v = 0.f; hit.prim = 0; hit.shape = 0; hit.inst = 0xffffffffu;
    float tmax = maxt;
    RaySetup R = ray_setup(o_w, d_w);
    bool in_tlas = A.has_tlas != 0;
    uint32_t cur_inst = 0xffffffffu;
    uint32_t ng_x = A.root, ng_y = 0x80000000u, tg_x = 0, tg_y = 0;
    int sp = 0, inst_sp = -1;
    for (;;) {
        if (ng_y > 0x00ffffffu) {
            uint32_t imask = ng_y & 0xffu;
            uint32_t bit = 31u - clz32(ng_y);
            ng_y &= ~(1u << bit);
            if (ng_y > 0x00ffffffu) {
                if (sp >= Stack::Capacity) { status = HAR_STACK_OVERFLOW; return false; }
                stack.push(sp++, ng_x, ng_y);
            }
            uint32_t slot = (bit - 24u) ^ R.octinv;
            uint32_t rel = popc32(imask & ~(0xffffffffu << slot));
            const uint32_t *np = reinterpret_cast<const uint32_t *>(A.nodes + (ng_x + rel));
#if defined(__HIP_DEVICE_COMPILE__)
            const uint4 n0 = reinterpret_cast<const uint4 *>(np)[0], n1 = reinterpret_cast<const uint4 *>(np)[1],
                        n2 = reinterpret_cast<const uint4 *>(np)[2], n3 = reinterpret_cast<const uint4 *>(np)[3],
                        n4 = reinterpret_cast<const uint4 *>(np)[4];
            const uint32_t w[20] = { n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w, n2.x, n2.y, n2.z, n2.w,
                                     n3.x, n3.y, n3.z, n3.w, n4.x, n4.y, n4.z, n4.w };
#else
            uint32_t w[20]; for (int i = 0; i < 20; ++i) w[i] = np[i];
#endif
            // w[0..2] origin, w[3] = ex | ey<<8 | ez<<16 | imask<<24, w[4] child_base, w[5] tri_base,
            // w[6..7] meta, w[8..9] qlox, w[10..11] qloy, w[12..13] qloz, w[14..15] qhix, w[16..17] qhiy, w[18..19] qhiz
            float sx = as_f32((w[3] & 0xffu) << 23), sy = as_f32(((w[3] >> 8) & 0xffu) << 23), sz = as_f32(((w[3] >> 16) & 0xffu) << 23);
            float ax = sx * R.idir.x, ay = sy * R.idir.y, az = sz * R.idir.z;
            float bx = (as_f32(w[0]) - R.o.x) * R.idir.x, by = (as_f32(w[1]) - R.o.y) * R.idir.y, bz = (as_f32(w[2]) - R.o.z) * R.idir.z;
            bool nx = R.idir.x < 0.f, ny = R.idir.y < 0.f, nz = R.idir.z < 0.f;
            uint32_t hitmask = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
            for (int i = 0; i < 8; ++i) {
                const int wi = i >> 2, sh = (i & 3) * 8;
                uint32_t meta = (w[6 + wi] >> sh) & 0xffu;
                float qlx = (float) ((w[8 + wi] >> sh) & 0xffu),  qly = (float) ((w[10 + wi] >> sh) & 0xffu), qlz = (float) ((w[12 + wi] >> sh) & 0xffu);
                float qhx = (float) ((w[14 + wi] >> sh) & 0xffu), qhy = (float) ((w[16 + wi] >> sh) & 0xffu), qhz = (float) ((w[18 + wi] >> sh) & 0xffu);
                float t0x = fma_(nx ? qhx : qlx, ax, bx), t1x = fma_(nx ? qlx : qhx, ax, bx);
                float t0y = fma_(ny ? qhy : qly, ay, by), t1y = fma_(ny ? qly : qhy, ay, by);
                float t0z = fma_(nz ? qhz : qlz, az, bz), t1z = fma_(nz ? qlz : qhz, az, bz);
                float tn = fmaxf(fmaxf(t0x, t0y), fmaxf(t0z, 0.f));
                float tf = fminf(fminf(t1x, t1y), fminf(t1z, tmax));
                bool isect = (meta != 0u) && (tn <= tf * 1.0000005f);
                if (isect) {
                    bool inner = (meta & 0x18u) == 0x18u;
                    uint32_t bits = meta >> 5;
                    uint32_t index = (meta ^ (inner ? R.octinv : 0u)) & 0x1fu;
                    hitmask |= bits << index;
                }
            }
            ng_x = w[4]; tg_x = w[5];
            ng_y = (hitmask & 0xff000000u) | (w[3] >> 24);
            tg_y = hitmask & 0x00ffffffu;
        } else {
            tg_x = ng_x; tg_y = ng_y; ng_x = 0; ng_y = 0;
        }

        while (tg_y != 0u) {
            uint32_t bit = 31u - clz32(tg_y);
            tg_y &= ~(1u << bit);
            uint32_t idx = tg_x + bit;
            if (in_tlas) {
                // InstanceEntry leaf: save the TLAS continuation, switch to object space
                if (ng_y > 0x00ffffffu) {
                    if (sp >= Stack::Capacity) { status = HAR_STACK_OVERFLOW; return false; }
                    stack.push(sp++, ng_x, ng_y);
                }
                if (tg_y != 0u) {
                    if (sp >= Stack::Capacity) { status = HAR_STACK_OVERFLOW; return false; }
                    stack.push(sp++, tg_x, tg_y);
                }
                const InstRec &I = A.insts[idx];
                inst_sp = sp; cur_inst = I.inst_index; in_tlas = false;
                if (!I.identity) R = ray_setup(xf_point(I.to_object, o_w), xf_vector(I.to_object, d_w));
                ng_x = I.blas_root; ng_y = 0x80000000u; tg_y = 0;
                break;
            } else {
                const float *tp = reinterpret_cast<const float *>(A.tris + idx);
#if defined(__HIP_DEVICE_COMPILE__)
                const float4 a = reinterpret_cast<const float4 *>(tp)[0], b = reinterpret_cast<const float4 *>(tp)[1],
                             c = reinterpret_cast<const float4 *>(tp)[2];
                const float f[12] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w };
#else
                float f[12]; for (int i = 0; i < 12; ++i) f[i] = tp[i];
#endif
                float t, u, v;
                if (moeller_trumbore(R.o, R.d, tmax, Vec3(f[0], f[1], f[2]), Vec3(f[3], f[4], f[5]), Vec3(f[6], f[7], f[8]), t, u, v)) {
                    if (AnyHit) return true;
                    hit_update(hit, t, u, v, as_u32(f[9]), as_u32(f[10]), cur_inst);
                    tmax = hit.t;
                }
            }
        }

        if (ng_y <= 0x00ffffffu) {
            if (!in_tlas && sp == inst_sp) {
                in_tlas = true; cur_inst = 0xffffffffu; inst_sp = -1;
                R = ray_setup(o_w, d_w);
            }
            if (sp == 0) break;
            stack.pop(--sp, ng_x, ng_y);
        }
    }
    return hit.t != HAR_INF;
}

/* Scene::ray_intersect_naive (scene.cpp:240-244): brute force over every triangle record */
template <bool AnyHit>
HAR_HD bool accel_trace_naive(const Accel &A, const uint32_t *blas_tri_ranges, Vec3 o_w, Vec3 d_w, float maxt, Hit &hit) {
    hit.t = HAR_INF; hit.u = 0.f; hit.v = 0.f; hit.prim = 0; hit.shape = 0; hit.inst = 0xffffffffu;
    float tmax = maxt;
    uint32_t n = A.has_tlas ? A.n_insts : 1u;
    for (uint32_t k = 0; k < n; ++k) {
        Vec3 o = o_w, d = d_w; uint32_t inst = 0xffffffffu, first, count;
        if (A.has_tlas) {
            const InstRec &I = A.insts[k];
            if (!I.identity) { o = xf_point(I.to_object, o_w); d = xf_vector(I.to_object, d_w); }
            inst = I.inst_index;
        }
        first = blas_tri_ranges[2 * k]; count = blas_tri_ranges[2 * k + 1];
        for (uint32_t i = first; i < first + count; ++i) {
            const TriRec &T = A.tris[i];
            float t, u, v;
            if (moeller_trumbore(o, d, tmax, Vec3(T.p0x, T.p0y, T.p0z), Vec3(T.e1x, T.e1y, T.e1z), Vec3(T.e2x, T.e2y, T.e2z), t, u, v)) {
                if (AnyHit) return true;
                hit_update(hit, t, u, v, T.prim, T.shape, inst);
                tmax = hit.t;
            }
        }
    }
    return hit.t != HAR_INF;
}

} // namespace har
