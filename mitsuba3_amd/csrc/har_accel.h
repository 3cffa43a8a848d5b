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
    uint8_t lmask, pad[7];          /* lmask: the child slots that are LEAVES (one primitive each: record tri_base + rank of the slot among the leaf slots); imask: the inner ones */
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

/* ---- node emission, shared by the host builder (har_accel_build.cpp) and the refit of har_refit.h (device + host): the frame of a node (origin + one power-of-two
 * scale per axis, after Ylitie et al. 2017) and the quantisation of a child box against it.  One implementation, so that a refit of geometry that did not move
 * reproduces the built nodes bit for bit (double arithmetic, exact on both sides). */
HAR_HD uint8_t node_exp_byte(double extent) {
    /* smallest e with extent / 2^e <= 255 */
    int e = -126;
    if (extent > 0.0) {
        e = (int) ceil(log2(extent / 255.0));
        while (extent / ldexp(1.0, e) > 255.0) ++e;
        while (e > -126 && extent / ldexp(1.0, e - 1) <= 255.0) --e;
    }
    e = e < -126 ? -126 : (e > 127 ? 127 : e);
    return (uint8_t) (e + 127);
}
HAR_HD void node_set_frame(Node8 &n, const float lo[3], const float hi[3]) {
    n.px = lo[0]; n.py = lo[1]; n.pz = lo[2];
    n.ex = node_exp_byte((double) hi[0] - (double) n.px);
    n.ey = node_exp_byte((double) hi[1] - (double) n.py);
    n.ez = node_exp_byte((double) hi[2] - (double) n.pz);
}
HAR_HD void node_quantise_child(Node8 &n, int s, const float clo[3], const float chi[3]) {
    const double sc[3] = { ldexp(1.0, (int) n.ex - 127), ldexp(1.0, (int) n.ey - 127), ldexp(1.0, (int) n.ez - 127) };
    const double org[3] = { n.px, n.py, n.pz };
    uint8_t *q[6] = { n.qlox, n.qloy, n.qloz, n.qhix, n.qhiy, n.qhiz };
    for (int a = 0; a < 3; ++a) {
        double lo = floor(((double) clo[a] - org[a]) / sc[a]), hi = ceil(((double) chi[a] - org[a]) / sc[a]);
        lo = lo < 0.0 ? 0.0 : (lo > 255.0 ? 255.0 : lo); hi = hi < 0.0 ? 0.0 : (hi > 255.0 ? 255.0 : hi);
        q[a][s] = (uint8_t) lo; q[3 + a][s] = (uint8_t) hi;
    }
}
/* conservative padding of a primitive box before it enters a node (builder and refit) */
HAR_HD void pad_box(float lo[3], float hi[3]) {
    float m = 1.f;
    for (int a = 0; a < 3; ++a) m = fmaxf(m, fmaxf(fabsf(lo[a]), fabsf(hi[a])));
    const float pad = 2e-5f * m;
    for (int a = 0; a < 3; ++a) { lo[a] -= pad; hi[a] += pad; }
}

struct Accel {
    const Node8   *nodes;
    const TriRec  *tris;
    const InstRec *insts;      /* TLAS leaf records */
    uint32_t root;             /* TLAS root if has_tlas, else BLAS root of the top-level geometry */
    uint32_t has_tlas;
    uint32_t n_tris, n_insts;
    /* two-level scenes: the top-level (non-instanced) geometry is NOT an entry of the TLAS.  A ray walks its BLAS first -- no instance entry, no
     * ray transform, and every lane of a wave does so at the same time -- and then starts at the TLAS root with the closest hit found so far
     * as tmax.  (As a TLAS entry its box spans the scene, so every ray entered it anyway: one instance-entry block per ray for nothing.)
     * top_root = 0xffffffff: no top-level geometry.  top_first / top_count: its triangle records (brute-force kernel). */
    uint32_t top_root, top_first, top_count;
    /* order of the two phases of a two-level scene (Traversal::begin; round 3).  A FEW top-level triangles around instanced content -- the walls of a box
     * scene -- are better walked AFTER the TLAS: half of the benchmark scene's shadow rays are occluded by an instance and then never pay the walls' node
     * visits (k_resolve 29.9 -> 26.5 ms), and a closest-hit ray that met an instance reaches the walls with tmax in front of them.  Many top-level
     * triangles (terrain under instanced trees) keep the top-level-first order, which enters the TLAS with tmax at the nearest top-level hit.
     * bit 0: any-hit rays walk the TLAS first, bit 1: closest-hit rays do. */
    uint32_t top_last;
    /* per mesh {first face of the mesh in the packed face / shading-triangle arrays, material word}: what a finished closest-hit ray adds to its hit record so that the
     * shading kernel starts its geometry and BSDF loads from the record instead of from a dependent load of the mesh record (HAR_HIT_MATINFO, har_kernels.h).
     * material word: bits 0-19 BSDF record, 20-21 mesh flags (vertex normals, texcoords), 22 carries an emitter, 24-27 material class (HAR_MAT_*) */
    const struct MeshInfo *mesh_info;
};
struct MeshInfo { uint32_t foff, matinfo; };
#define HAR_MATINFO_BSDF(m)    ((m) & 0xfffffu)
#define HAR_MATINFO_FLAGS(m)   (((m) >> 20) & 3u)
#define HAR_MATINFO_EMITTER(m) (((m) >> 22) & 1u)
#define HAR_MATINFO_CLASS(m)   (((m) >> 24) & 0xfu)
#define HAR_NO_NODE 0xffffffffu

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
/* reciprocal for the (conservative) box tests only: v_rcp_f32 (1 ulp) on the device -- the slab test has a
 * 5e-7 relative slack and leaf boxes are padded by 2e-5, the exact triangle test never uses it */
HAR_HD float box_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.f / x;
#endif
}
HAR_HD RaySetup ray_setup(Vec3 o, Vec3 d) {
    RaySetup r; r.o = o; r.d = d;
    const float eps = 1e-30f;
    r.idir = Vec3(box_rcp(fabsf(d.x) > eps ? d.x : mulsign_(eps, d.x)),
                  box_rcp(fabsf(d.y) > eps ? d.y : mulsign_(eps, d.y)),
                  box_rcp(fabsf(d.z) > eps ? d.z : mulsign_(eps, d.z)));
    r.octinv = (d.x < 0.f ? 0u : 4u) | (d.y < 0.f ? 0u : 2u) | (d.z < 0.f ? 0u : 1u);
    return r;
}

#define HAR_STACK_OVERFLOW 0x7fffffff
#define HAR_MAX_PARKED 3          /* Traversal<2>: stack entries beyond the depth-first bound (HostScene::stack_need) */

/* Probe: optional per-ray instrumentation of the traversal loops (tools/ and the host test harness implement it; tests/host_harness/harness.cpp).  The default
 * compiles to nothing, and nothing of the instrumentation lives in this header:
 *   iter / node / tri / inst        one outer iteration / node visit / triangle test / instance entry
 *   ray(any_hit)                    a reference-loop query starts
 *   visited(A, R, tmax, node, ng_y, tg_y, phase)   after a node visit of the reference loop: phase 0 = top-level BLAS of a two-level scene, 1 = TLAS, 2 = inside an instance
 *   leaf(phase) / entered()         a triangle test in that phase / an instance was entered
 *   left(top_phase, useful)         an instance (or the top-level phase) is done; useful = it gave the ray its closest hit
 *   back_to_front()                 what-if: take the FARTHEST pending child first */
struct Accel; struct RaySetup;
struct NoProbe {
    HAR_HD void node() {} HAR_HD void tri() {} HAR_HD void inst() {} HAR_HD void iter() {}
    HAR_HD void ray(bool) {} HAR_HD void visited(const Accel &, const RaySetup &, float, uint32_t, uint32_t, uint32_t, int) {} HAR_HD void leaf(int) {} HAR_HD void entered() {}
    HAR_HD void left(bool, bool) {} HAR_HD bool back_to_front() const { return false; }
};

/* (m << 1) | sign bit of x: v_alignbit_b32 on the device */
/* The degenerate cases of that sign test, pinned so that host (harness, har_render_scalar) and device agree and neither drops a box that could hold a hit:
 *  - NaN (inf - inf: an axis-parallel ray whose slab distances overflow on both sides): the device's fma returns the DEFAULT NaN, which is positive on gfx950 -> the slot
 *    counts as hit (conservative); x86 produces a negative default NaN -> the host build maps every NaN to "hit" explicitly;
 *  - tf = -0 with tn = 0 (the ray starts exactly on the far plane of a box and leaves it): fma(-0, c, -0) = -0 -> miss on both sides; a triangle that lies in that plane
 *    is still found through a neighbouring box or not at all, exactly as with the reference's kd-tree whose split planes have the same measure-zero ambiguity. */
HAR_HD uint32_t shift_in_sign(uint32_t m, float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(m, as_u32(x), 31u);
#else
    return (m << 1) | ((x != x) ? 0u : (as_u32(x) >> 31));
#endif
}
/* bit i of x moves to bit i ^ c (x: 8 bits, c: 3 bits) */
HAR_HD uint32_t xor_permute8(uint32_t x, uint32_t c) {
    const uint32_t k1 = c & 1u, k2 = c & 2u, k4 = c & 4u;
    x = ((x << k1) & 0xaau) | ((x >> k1) & ~0xaau);          /* (a & m) | (b & ~m): one v_bfi_b32 per stage */
    x = ((x << k2) & 0xccu) | ((x >> k2) & ~0xccu);
    x = ((x << k4) & 0xf0u) | ((x >> k4) & ~0xf0u);
    return x;
}
/* next leaf of a leaf group (tg_y = hit leaf slots | the node's leaf slots << 8; 0 when no leaf is pending): returns its record index and clears it */
HAR_HD uint32_t tg_next_leaf(uint32_t tg_x, uint32_t &tg_y) {
    const uint32_t hl = tg_y & 0xffu, bit = 31u - clz32(hl);
    const uint32_t idx = tg_x + popc32((tg_y >> 8) & ~(0xffffffffu << bit));
    tg_y = (hl & ~(1u << bit)) ? (tg_y & ~(1u << bit)) : 0u;
    return idx;
}

/*
 * One node visit: fetch the 80-byte node `index`, intersect the ray with its 8 quantised child
 * boxes and return the hits as a node group (child_base | hit inner children in traversal order << 24 |
 * imask) and a leaf group (tri_base | hit leaf slots | lmask << 8) -- after Ylitie et al.'s encoding.
 */
HAR_HD void node_visit(const Accel &A, const RaySetup &R, float tmax, uint32_t index, uint32_t &ng_x, uint32_t &ng_y, uint32_t &tg_x, uint32_t &tg_y) {
    const uint32_t *np = reinterpret_cast<const uint32_t *>(A.nodes + index);
#if defined(__HIP_DEVICE_COMPILE__)
    const uint4 n0 = reinterpret_cast<const uint4 *>(np)[0], n1 = reinterpret_cast<const uint4 *>(np)[1],
                n2 = reinterpret_cast<const uint4 *>(np)[2], n3 = reinterpret_cast<const uint4 *>(np)[3],
                n4 = reinterpret_cast<const uint4 *>(np)[4];
    const uint32_t w[20] = { n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w, n2.x, n2.y, n2.z, n2.w,
                             n3.x, n3.y, n3.z, n3.w, n4.x, n4.y, n4.z, n4.w };
#else
    uint32_t w[20]; for (int i = 0; i < 20; ++i) w[i] = np[i];
#endif
    // w[0..2] origin, w[3] = ex | ey<<8 | ez<<16 | imask<<24, w[4] child_base, w[5] tri_base, w[6] & 0xff = lmask,
    // w[8..9] qlox, w[10..11] qloy, w[12..13] qloz, w[14..15] qhix, w[16..17] qhiy, w[18..19] qhiz
    float sx = as_f32((w[3] & 0xffu) << 23), sy = as_f32(((w[3] >> 8) & 0xffu) << 23), sz = as_f32(((w[3] >> 16) & 0xffu) << 23);
    float ax = sx * R.idir.x, ay = sy * R.idir.y, az = sz * R.idir.z;
    float bx = (as_f32(w[0]) - R.o.x) * R.idir.x, by = (as_f32(w[1]) - R.o.y) * R.idir.y, bz = (as_f32(w[2]) - R.o.z) * R.idir.z;
    /* the near / far plane of a slab depends only on the ray's sign: swap the quantised words once per node */
    const bool nx = R.idir.x < 0.f, ny = R.idir.y < 0.f, nz = R.idir.z < 0.f;
    const uint32_t nxw[2] = { nx ? w[14] : w[8],  nx ? w[15] : w[9]  }, fxw[2] = { nx ? w[8]  : w[14], nx ? w[9]  : w[15] };
    const uint32_t nyw[2] = { ny ? w[16] : w[10], ny ? w[17] : w[11] }, fyw[2] = { ny ? w[10] : w[16], ny ? w[11] : w[17] };
    const uint32_t nzw[2] = { nz ? w[18] : w[12], nz ? w[19] : w[13] }, fzw[2] = { nz ? w[12] : w[18], nz ? w[13] : w[19] };
    /* ONE BIT PER CHILD SLOT (round 4): slot i is hit when  tf * (1 + 5e-7) - tn >= 0; the sign of that (fused) difference is shifted into `miss`, slot 7 first, so
     * that slot i ends at bit i -- an fma and a funnel shift per child where a multiply, a compare, a select and a shift-or (plus two byte extractions of a per-child
     * meta byte) used to assemble the mask.  Leaves hold one primitive each, so a slot needs no count. */
    uint32_t miss = 0u;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 7; i >= 0; --i) {
        const int wi = i >> 2, sh = (i & 3) * 8;
        float t0x = fma_((float) ((nxw[wi] >> sh) & 0xffu), ax, bx), t1x = fma_((float) ((fxw[wi] >> sh) & 0xffu), ax, bx);
        float t0y = fma_((float) ((nyw[wi] >> sh) & 0xffu), ay, by), t1y = fma_((float) ((fyw[wi] >> sh) & 0xffu), ay, by);
        float t0z = fma_((float) ((nzw[wi] >> sh) & 0xffu), az, bz), t1z = fma_((float) ((fzw[wi] >> sh) & 0xffu), az, bz);
        float tn = fmaxf(fmaxf(t0x, t0y), fmaxf(t0z, 0.f));
        float tf = fminf(fminf(t1x, t1y), fminf(t1z, tmax));
        miss = shift_in_sign(miss, fma_(tf, 1.0000005f, -tn));
    }
    const uint32_t imask = w[3] >> 24, lmask = w[6] & 0xffu;
    const uint32_t inner = ~miss & imask, leaf = ~miss & lmask;
    ng_x = w[4]; tg_x = w[5];
    /* inner children are taken front to back along the ray: the child in slot s at position s ^ octinv (the builder stores them in octant order) -- a permutation of the
     * eight bits by XOR on their index = three conditional swaps of neighbouring bits / pairs / nibbles, written as shifts by 0 or by the stage's distance */
    ng_y = (xor_permute8(inner, R.octinv) << 24) | imask;
    tg_y = leaf ? (leaf | (lmask << 8)) : 0u;                  /* leaf group: hit leaf slots | the node's leaf slots (for the rank of a slot: tg_next_leaf) */
}

/* pick the next child of a node group (front-to-back = highest bit); returns its node index.  back_to_front: a Probe's what-if (host models only) */
HAR_HD uint32_t ng_next_child(uint32_t ng_x, uint32_t &ng_y, uint32_t octinv, bool back_to_front = false) {
    uint32_t imask = ng_y & 0xffu;
    uint32_t bit = 31u - clz32(ng_y);
#if !defined(__HIP_DEVICE_COMPILE__)
    if (back_to_front) bit = (uint32_t) __builtin_ctz(ng_y & 0xff000000u);
#else
    (void) back_to_front;
#endif
    ng_y &= ~(1u << bit);
    uint32_t slot = (bit - 24u) ^ octinv;
    return ng_x + popc32(imask & ~(0xffffffffu << slot));
}

/* one triangle record against the (object-space) ray; closest-hit bookkeeping.  `tp`: a TriRec (12 words) */
template <bool AnyHit>
HAR_HD bool tri_visit_at(const float *tp, const RaySetup &R, float &tmax, uint32_t cur_inst, Hit &hit) {
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
        /* + 0.f: a hit at t = -0 (the origin lies in the triangle's plane and the ray looks along -n) must not leave tmax = -0 behind -- the box test's far distance would then be
         * min(.., -0) = -0, its sign test  fma(-0, 1.0000005, -tn)  reads "miss" for tn = 0, and every box that holds an equal-t candidate (the tie rule's later primitive) would
         * be pruned: the BVH would answer differently from the brute-force loop (found by tests/test_gpu_boundary.py::test_ray_queries_bitexact_on_adversarial_rays).  -0 + 0 = +0;
         * the reported hit.t keeps its sign */
        tmax = hit.t + 0.f;
    }
    return false;
}
template <bool AnyHit>
HAR_HD bool tri_visit(const Accel &A, const RaySetup &R, float &tmax, uint32_t idx, uint32_t cur_inst, Hit &hit) {
    return tri_visit_at<AnyHit>(reinterpret_cast<const float *>(A.tris + idx), R, tmax, cur_inst, hit);
}

/*
 * Reference traversal loop (depth first, every triangle of a visited node at once).  Used by the
 * array-valued API kernels and the host harness.
 *
 * Stack concept: void push(int level, uint32_t x, uint32_t y); void pop(int level, uint32_t &x, uint32_t &y);
 * static constexpr int Capacity.
 * Returns true if (AnyHit and an intersection exists) or (closest hit found).
 * `status` is set to HAR_STACK_OVERFLOW if the stack capacity was exceeded.
 */
template <bool AnyHit, typename Stack, typename Probe = NoProbe>
HAR_HD bool accel_trace(const Accel &A, Vec3 o_w, Vec3 d_w, float maxt, Hit &hit, Stack &stack, int &status, Probe probe = Probe()) {
    hit.t = HAR_INF; hit.u = 0.f; hit.v = 0.f; hit.prim = 0; hit.shape = 0; hit.inst = 0xffffffffu;
    float tmax = maxt + 0.f;                       /* (-0 -> +0: see tri_visit_at) */
    RaySetup R = ray_setup(o_w, d_w);
    bool in_tlas = A.has_tlas != 0;
    uint32_t cur_inst = 0xffffffffu;
    uint32_t ng_x = A.root, ng_y = 0x80000000u, tg_x = 0, tg_y = 0;
    int sp = 0, inst_sp = -1;
    bool tlas_pending = false;
    if (A.has_tlas && A.top_root != HAR_NO_NODE) { in_tlas = false; inst_sp = 0; ng_x = A.top_root; tlas_pending = true; }      /* top-level geometry first */
    probe.ray(AnyHit);
    for (;;) {
        probe.iter();
        if (ng_y > 0x00ffffffu) {
            probe.node();
            uint32_t px = ng_x, py = ng_y;
            uint32_t child = ng_next_child(px, py, R.octinv, probe.back_to_front());
            if (py > 0x00ffffffu) {
                if (sp >= Stack::Capacity) { status = HAR_STACK_OVERFLOW; return false; }
                stack.push(sp++, px, py);
            }
            node_visit(A, R, tmax, child, ng_x, ng_y, tg_x, tg_y);
            probe.visited(A, R, tmax, child, ng_y, tg_y, tlas_pending ? 0 : in_tlas ? 1 : 2);
        } else {
            tg_x = ng_x; tg_y = ng_y; ng_x = 0; ng_y = 0;
        }

        while (tg_y != 0u) {
            uint32_t idx = tg_next_leaf(tg_x, tg_y);
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
                probe.inst(); probe.entered();
                const InstRec &I = A.insts[idx];
                inst_sp = sp; cur_inst = I.inst_index; in_tlas = false;
                if (!I.identity) R = ray_setup(xf_point(I.to_object, o_w), xf_vector(I.to_object, d_w));
                ng_x = I.blas_root; ng_y = 0x80000000u; tg_y = 0;
                break;
            } else {
                probe.tri(); probe.leaf(tlas_pending ? 0 : 2);
                if (tri_visit<AnyHit>(A, R, tmax, idx, cur_inst, hit)) return true;
            }
        }

        if (ng_y <= 0x00ffffffu) {
            if (!in_tlas && sp == inst_sp) {
                probe.left(tlas_pending, hit.inst == cur_inst);
                in_tlas = true; cur_inst = 0xffffffffu; inst_sp = -1;
                if (tlas_pending) { tlas_pending = false; ng_x = A.root; ng_y = 0x80000000u; continue; }      /* the ray is still the world-space one */
                R = ray_setup(o_w, d_w);
            }
            if (sp == 0) break;
            stack.pop(--sp, ng_x, ng_y);
        }
    }
    return hit.t != HAR_INF;
}

/*
 * Resumable, DECOUPLED traversal for the persistent wavefront kernels.  One call of step() is one
 * SIMT-friendly iteration with a fixed shape: at most ONE node visit, then at most ONE leaf item
 * (one triangle test in a BLAS, one instance entry in the TLAS), then at most one stack pop -- so
 * that the 64 lanes of a wave execute the same three blocks whatever their rays are doing (the
 * reference loop above runs "all triangles of the node" inside every iteration, which costs a wave
 * max-over-lanes triangle tests per iteration: measured 6 % lane utilisation in that block).
 *
 * POLICY 0: a node is only visited when no triangle is pending (same stack need as the reference loop);
 * POLICY 1: nodes are visited every iteration, a still-pending triangle group is parked on the stack.
 * Both visit the same set of candidate triangles with the exact Moeller-Trumbore test and the same
 * tie rule, so the result is identical to accel_trace (order independent).
 */
/* FLAT = true: the specialisation for scenes WITHOUT a TLAS (Accel::has_tlas == 0: one BLAS over the top-level meshes -- the Cornell boxes, the flattened
 * 1M-triangle scene).  `in_tlas` is constantly false there, so the instance-entry block, the instance exit with its second ray_setup, the world-space copy
 * of the ray and the instance / phase flags all fold away at compile time; the persistent kernels are instantiated for both (launch_trace_closest /
 * launch_resolve pick by has_tlas).  Same visits, same tests, same result as the generic code on such a scene. */
#define HAR_INST_LEAVING  0xfffffffeu
#define HAR_INST_ENTERING 0xfffffffdu
#ifndef HAR_DEFER_XFORM
#define HAR_DEFER_XFORM 1     /* instance entry / exit only NOTE the transition; the object-space ray, cur_inst and the BLAS root are set at ONE place, the top of
                               * the next step (apply_pending) -- see Traversal::pend.  0: they are set where the transition happens (round-2 form) */
#endif
template <int POLICY, bool FLAT = false>
struct Traversal {
    Vec3 o_w, d_w;
    RaySetup R;
    float tmax;
    Hit hit;
    uint32_t ng_x, ng_y, tg_x, tg_y, cur_inst;
    int sp, inst_sp;
    uint32_t parked;            /* POLICY 2: triangle groups currently parked on the stack (<= HAR_MAX_PARKED) */
    bool in_tlas, found;
    bool top_last, top_pending; /* any-hit rays: TLAS first, then the top-level BLAS (begin); top_pending: it is still to be walked once the TLAS is exhausted */
    /* pending ray transition (HAR_DEFER_XFORM), kept in cur_inst: HAR_INST_LEAVING = leave the instance (world-space ray again), HAR_INST_ENTERING = enter the
     * instance record whose index waits in ng_x.  Why: `R`, `cur_inst` and `ng_x` re-defined inside the leaf and pop phases made the compiler keep TWO copies of
     * that state (15 registers) and shuffle them with v_mov every wave step (ISA audit). */

    /* top_last (any-hit queries only; pair it with step<AnyHit = true>): walk the TLAS FIRST and the top-level BLAS afterwards.  For a closest-hit ray
     * the top-level geometry goes first so that the TLAS is entered with tmax at the nearest wall; an occlusion query has no use for that, and on a
     * "box around instanced content" scene half of the shadow rays are occluded by an instance -- they never pay the 2.5 node visits of the walls */
    HAR_HD void begin(const Accel &A, Vec3 o, Vec3 d, float maxt, bool top_last = false) {
        o_w = o; d_w = d; tmax = maxt + 0.f /* -0 -> +0: see tri_visit_at */; top_pending = false; this->top_last = top_last && POLICY != 2;
        hit.t = HAR_INF; hit.u = 0.f; hit.v = 0.f; hit.prim = 0; hit.shape = 0; hit.inst = 0xffffffffu;
        R = ray_setup(o, d);
        in_tlas = !FLAT && A.has_tlas != 0; cur_inst = 0xffffffffu; found = false;
        ng_x = A.root; ng_y = 0x80000000u; tg_x = 0; tg_y = 0; sp = 0; inst_sp = -1; parked = 0;
        if (FLAT) { this->top_last = false; return; }
        /* top-level geometry first (see Accel): the state "in a BLAS, no instance" (in_tlas false, cur_inst none) only exists in this phase of a
         * two-level scene -- TLAS entries always carry an instance index -- so no extra flag is kept */
        if (A.has_tlas && A.top_root != HAR_NO_NODE) {
            if (this->top_last) top_pending = true;                   /* start at the TLAS root (set above) */
            else { in_tlas = false; inst_sp = 0; ng_x = A.top_root; }
        }
    }

    /* ---- node phase: at most one node visit */
    template <typename Stack, typename Probe>
    HAR_HD bool phase_node(const Accel &A, Stack &stack, int &status, Probe &probe) {
        if (ng_y > 0x00ffffffu && (POLICY == 1 || tg_y == 0u || (POLICY == 2 && parked < HAR_MAX_PARKED))) {
            probe.node();
            uint32_t child = ng_next_child(ng_x, ng_y, R.octinv);
            if (ng_y > 0x00ffffffu) {
                if (sp >= Stack::Capacity) return overflow(status);
                stack.push(sp++, ng_x, ng_y);
            }
            uint32_t cx, cy;
            node_visit(A, R, tmax, child, ng_x, ng_y, cx, cy);
            if (POLICY == 2 && ng_y <= 0x00ffffffu) ng_y = 0u;        /* no inner child hit: drop the leftover imask */
            if (POLICY >= 1 && tg_y != 0u && cy != 0u) {          /* park the NEW group, finish the nearer old one first */
                if (sp >= Stack::Capacity) return overflow(status);
                stack.push(sp++, cx, cy); ++parked;
            } else if (cy != 0u) { tg_x = cx; tg_y = cy; }
        }
        return false;
    }
    /* ---- leaf phase: one triangle test (BLAS) or one instance entry (TLAS) */
    template <bool AnyHit, typename Stack, typename Probe>
    HAR_HD bool phase_leaf(const Accel &A, Stack &stack, int &status, Probe &probe, bool allow_inst = true) {
        if (POLICY == 2 && tg_y == 0u && ng_y <= 0x00ffffffu && ng_y != 0u) { tg_x = ng_x; tg_y = ng_y; ng_x = 0; ng_y = 0; }
        /* allow_inst = false (persistent kernels, deferred instance entry): a lane whose next leaf item is an instance entry waits this step out -- the
         * entry block (ray transform, reciprocals, two pushes: ~100 instructions) then runs when several lanes of the wave need it, not for one */
        if (tg_y != 0u && (FLAT || allow_inst || !in_tlas)) {
            uint32_t idx = tg_next_leaf(tg_x, tg_y);
            if (!FLAT && in_tlas) {
                if (POLICY == 2 ? ng_y != 0u : ng_y > 0x00ffffffu) {      /* POLICY 2: ng may hold a waiting instance list */
                    if (sp >= Stack::Capacity) return overflow(status);
                    stack.push(sp++, ng_x, ng_y);
                }
                if (tg_y != 0u) {
                    if (sp >= Stack::Capacity) return overflow(status);
                    stack.push(sp++, tg_x, tg_y);
                }
                probe.inst();
                inst_sp = sp; in_tlas = false;
#if HAR_DEFER_XFORM
                cur_inst = HAR_INST_ENTERING; ng_x = idx;
#else
                const InstRec &I = A.insts[idx];
                cur_inst = I.inst_index;
                if (!I.identity) R = ray_setup(xf_point(I.to_object, o_w), xf_vector(I.to_object, d_w));
                ng_x = I.blas_root;
#endif
                ng_y = 0x80000000u; tg_y = 0;
            } else {
                probe.tri();
                if (tri_visit<AnyHit>(A, R, tmax, idx, FLAT ? 0xffffffffu : cur_inst, hit)) { found = true; return true; }
            }
        }
        return false;
    }
    /* ---- pop phase: leave the instance / finish / take the next group from the stack */
    template <typename Stack, bool TOP_LAST = false>
    HAR_HD bool phase_pop(const Accel &A, Stack &stack) {
        if (POLICY == 2) {
            /* pop as soon as no group is held in ng, even while triangles are pending (they are tested one per
             * iteration in parallel with the node visits); a popped triangle group waits in ng until tg is empty */
            if (ng_y == 0u) {
                if (!FLAT && !in_tlas && sp == inst_sp) {
                    if (tg_y != 0u) return false;                      /* drain the instance's triangles first */
                    const bool top_phase = cur_inst == 0xffffffffu;
                    in_tlas = true; inst_sp = -1;
                    if (top_phase) { ng_x = A.root; ng_y = 0x80000000u; return false; }
                    leave_instance();
                }
                if (sp == 0) { if (tg_y != 0u) return false; found = hit.t != HAR_INF; return true; }
                stack.pop(--sp, ng_x, ng_y);
                if ((FLAT || !in_tlas) && ng_y <= 0x00ffffffu) --parked;         /* in a BLAS only parked triangle groups are non-node entries */
            }
            return false;
        }
        if (ng_y <= 0x00ffffffu && tg_y == 0u) {
            if (!FLAT && !in_tlas && sp == inst_sp) {
                const bool top_phase = cur_inst == 0xffffffffu;
                in_tlas = true; inst_sp = -1;
                if (top_phase) {
                    if (TOP_LAST && top_last) { found = hit.t != HAR_INF; return true; } /* any-hit order: the top-level BLAS was the last thing to walk */
                    ng_x = A.root; ng_y = 0x80000000u; return false;                      /* top-level BLAS done: on to the TLAS with the same (world-space) ray */
                }
                leave_instance();
            }
            if (sp == 0) {
                if (!FLAT && TOP_LAST && top_pending) {        /* TLAS exhausted without an occluder: now the top-level geometry (world-space ray, no instance) */
                    top_pending = false; in_tlas = false; inst_sp = 0; ng_x = A.top_root; ng_y = 0x80000000u;
                    return false;
                }
                found = hit.t != HAR_INF; return true;
            }
            uint32_t x, y;
            stack.pop(--sp, x, y);
            if (y > 0x00ffffffu) { ng_x = x; ng_y = y; } else { tg_x = x; tg_y = y; ng_x = 0; ng_y = 0; }
        }
        return false;
    }
    /* field-wise select: this = c ? n : this (the persistent kernels' refill) */
    HAR_HD void merge(bool c, const Traversal &n) {
        static_assert(sizeof(Traversal) == 128, "Traversal gained or lost a member: bring merge() up to date");
#define HAR_SEL(f) f = c ? n.f : f
        HAR_SEL(o_w.x); HAR_SEL(o_w.y); HAR_SEL(o_w.z); HAR_SEL(d_w.x); HAR_SEL(d_w.y); HAR_SEL(d_w.z);
        HAR_SEL(R.o.x); HAR_SEL(R.o.y); HAR_SEL(R.o.z); HAR_SEL(R.d.x); HAR_SEL(R.d.y); HAR_SEL(R.d.z);
        HAR_SEL(R.idir.x); HAR_SEL(R.idir.y); HAR_SEL(R.idir.z); HAR_SEL(R.octinv);
        HAR_SEL(tmax); HAR_SEL(hit.t); HAR_SEL(hit.u); HAR_SEL(hit.v); HAR_SEL(hit.prim); HAR_SEL(hit.shape); HAR_SEL(hit.inst);
        HAR_SEL(ng_x); HAR_SEL(ng_y); HAR_SEL(tg_x); HAR_SEL(tg_y); HAR_SEL(cur_inst); HAR_SEL(sp); HAR_SEL(inst_sp); HAR_SEL(parked);
        HAR_SEL(in_tlas); HAR_SEL(found); HAR_SEL(top_last); HAR_SEL(top_pending);
#undef HAR_SEL
    }
    HAR_HD bool overflow(int &status) { status = HAR_STACK_OVERFLOW; hit.t = HAR_INF; found = false; return true; }
    HAR_HD void leave_instance() {
#if HAR_DEFER_XFORM
        cur_inst = HAR_INST_LEAVING;
#else
        cur_inst = 0xffffffffu; R = ray_setup(o_w, d_w);
#endif
    }
    /* the one place where the ray changes its space (HAR_DEFER_XFORM): called at the top of every step */
    HAR_HD void apply_pending(const Accel &A) {
#if HAR_DEFER_XFORM
        if (!FLAT && (cur_inst == HAR_INST_LEAVING || cur_inst == HAR_INST_ENTERING)) {
            if (cur_inst == HAR_INST_LEAVING) { cur_inst = 0xffffffffu; R = ray_setup(o_w, d_w); }
            else {
                const InstRec &I = A.insts[ng_x];
                cur_inst = I.inst_index; ng_x = I.blas_root;
                /* always from the WORLD-space ray (an identity record transforms to the same bits): a pending LEAVING that this ENTERING overwrote -- possible
                 * once leaf_round() runs between two steps -- must not leave the previous instance's object-space ray in R */
                R = ray_setup(xf_point(I.to_object, o_w), xf_vector(I.to_object, d_w));
            }
        }
#else
        (void) A;
#endif
    }

    /* extra leaf round of the persistent kernels: one more leaf item + pop for a lane that still has one pending,
     * so that it is ready for a node visit in the next iteration (the node block is the expensive one) */
    template <bool AnyHit, typename Stack>
    HAR_HD bool leaf_round(const Accel &A, Stack &stack, int &status) {
        NoProbe probe;
        if (phase_leaf<AnyHit>(A, stack, status, probe)) return true;
        return phase_pop<Stack, true>(A, stack);
    }
    HAR_HD bool leaf_pending() const { return tg_y != 0u; }

    /* one iteration; returns true when the ray is finished (`found` / `hit` hold the result).
     * ORDER 0: node, leaf, pop   1: leaf, node, pop   2: leaf, pop, node */
    HAR_HD bool wants_instance_entry() const { return !FLAT && in_tlas && tg_y != 0u; }
    template <bool AnyHit, typename Stack, typename Probe = NoProbe, int ORDER = 2>
    HAR_HD bool step(const Accel &A, Stack &stack, int &status, Probe probe = Probe(), bool allow_inst = true) {
        probe.iter();
        apply_pending(A);
        /* the TLAS-first order is switched on per ray by begin(..., top_last) from Accel::top_last */
        if (ORDER == 0) {
            if (phase_node(A, stack, status, probe)) return true;
            if (phase_leaf<AnyHit>(A, stack, status, probe, allow_inst)) return true;
            return phase_pop<Stack, true>(A, stack);
        } else if (ORDER == 1) {
            if (phase_leaf<AnyHit>(A, stack, status, probe, allow_inst)) return true;
            apply_pending(A);                       /* the node phase of THIS step already walks the instance just entered */
            if (phase_node(A, stack, status, probe)) return true;
            return phase_pop<Stack, true>(A, stack);
        } else {
            if (phase_leaf<AnyHit>(A, stack, status, probe, allow_inst)) return true;
            if (phase_pop<Stack, true>(A, stack)) return true;
            apply_pending(A);
            return phase_node(A, stack, status, probe);
        }
    }
};

/* Scene::ray_intersect_naive (scene.cpp:240-244): brute force over every triangle record */
template <bool AnyHit>
HAR_HD bool accel_trace_naive(const Accel &A, const uint32_t *blas_tri_ranges, Vec3 o_w, Vec3 d_w, float maxt, Hit &hit) {
    hit.t = HAR_INF; hit.u = 0.f; hit.v = 0.f; hit.prim = 0; hit.shape = 0; hit.inst = 0xffffffffu;
    float tmax = maxt;
    const uint32_t n_top = (A.has_tlas && A.top_root != HAR_NO_NODE) ? 1u : 0u;
    uint32_t n = A.has_tlas ? A.n_insts + n_top : 1u;
    for (uint32_t kk = 0; kk < n; ++kk) {
        Vec3 o = o_w, d = d_w; uint32_t inst = 0xffffffffu, first, count;
        if (kk < n_top) { first = A.top_first; count = A.top_count; }          /* the top-level geometry of a two-level scene */
        else {
            const uint32_t k = kk - n_top;
            if (A.has_tlas) {
                const InstRec &I = A.insts[k];
                if (!I.identity) { o = xf_point(I.to_object, o_w); d = xf_vector(I.to_object, d_w); }
                inst = I.inst_index;
            }
            first = blas_tri_ranges[2 * k]; count = blas_tri_ranges[2 * k + 1];
        }
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
