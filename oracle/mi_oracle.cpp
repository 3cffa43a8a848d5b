/*
 * mi_oracle.cpp -- CPU ORACLE: restatement of the reference hot path
 * (mitsuba3 v3.9.1 `llvm_ad_rgb`: forward `path` + `prb` adjoint on triangle
 * scenes).  TEST INFRASTRUCTURE ONLY -- see mi_oracle.h for the rules and for
 * the "parity unpinned" statement.  Each function cites the reference
 * file:line it follows (paths relative to /root/reference).
 */
#include "mi_oracle.h"
#include "orc_math.h"
#include "orc_bsdf.h"
#include "orc_bsdf_ctx.h"
#include "orc_envmap.h"
#include "orc_dual.h"

#include <algorithm>
#include <memory>
#include <atomic>
#include <fstream>
#include <sched.h>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

using namespace orc;

namespace {

// ---------------------------------------------------------------------------
//  Scene storage
// ---------------------------------------------------------------------------

struct Mesh {
    std::vector<float> V;      // 8 floats / vertex
    std::vector<uint32_t> F;   // 4 u32 / face
    uint32_t nv, nf, bsdf; int32_t emitter; uint32_t flags;
};
struct Tri { V3 p0, e1, e2; uint32_t prim, shape; };
struct BvhNode { float lo[3], hi[3]; uint32_t left, count; /* count>0: leaf, left=first */ uint32_t right; };
struct Bvh {
    std::vector<BvhNode> nodes;
    std::vector<Tri> tris;
    float lo[3], hi[3];
    bool empty() const { return tris.empty(); }
};
/* `xf`: the bitmap's m_transform (to_uv, bitmap.cpp:175) as the 3 x 3 matrix of the ScalarAffineTransform3f; `moved` = it is not the identity */
struct Texture {
    std::vector<float> data; uint32_t w, h, mode = 0;
    float xf[3][3] = { { 1.f, 0.f, 0.f }, { 0.f, 1.f, 0.f }, { 0.f, 0.f, 1.f } }; bool moved = false;
    /* m_transform * Point2f, affine branch of Transform::operator*(Point) (transform.h:322-335): start from the translation column, then one fmadd per input coordinate */
    void to_texture_space(const float uv_in[2], float uv_out[2]) const {
        if (!moved) { uv_out[0] = uv_in[0]; uv_out[1] = uv_in[1]; return; }
        for (int row = 0; row < 2; ++row) {
            float acc = xf[row][2];
            for (int col = 0; col < 2; ++col) acc = fmadd(xf[row][col], uv_in[col], acc);
            uv_out[row] = acc;
        }
    }
};

static inline uint32_t tex_wrap_pos(int64_t pos, int64_t res, uint32_t mode);
#include "orc_texlight.h"

struct Scene {
    std::vector<TexelTable> texel_tables;  // indexed by emitter: AreaLight with a bitmap radiance (type 7)
    std::vector<Mesh> meshes; uint32_t top_count;
    std::vector<OrcShapeGroup> groups;
    std::vector<OrcInstance> instances;
    std::vector<BsdfRecord> bsdfs;
    std::vector<Texture> textures;
    std::vector<OrcEmitter> emitters;
    int env = -1; float env_center[3] = { 0, 0, 0 }; float env_radius = 0.f;   // Scene::environment() + its bounding sphere
    bool hide_emitters = false;            // Integrator property `hide_emitters` (integrator.cpp:29), set by orc_scene_set_hide_emitters
    bool alpha_only = false;               // forward renders splat the alpha value (valid_ray ? 1 : 0) instead of the radiance (orc_scene_set_alpha_only)
    EnvMap envmap;                 // when emitters[env].type == 2
    /* AreaLight on a triangle mesh (emitter type 3): DiscreteDistribution over the face areas (Mesh::build_pmf, mesh.cpp:1358-1372) */
    struct AreaPmf { std::vector<float> pmf, cdf; float sum = 0.f, normalization = 0.f; };
    std::vector<AreaPmf> area_pmf;  // indexed by emitter
    /* Scene::m_emitter_distr (scene.cpp:120-141): set up when some emitter's sampling_weight is not 1; `first` / `last` = DiscreteDistribution::m_valid */
    struct EmitterChoice { bool weighted = false; AreaPmf table; uint32_t first = 0, last = 0; } choice;
    Bvh top;                       // all top-level meshes
    std::vector<Bvh> group_bvh;    // one per shapegroup
    std::vector<BvhNode> inst_nodes; // BVH over instance world boxes
    std::vector<uint32_t> inst_order;
};

struct Ray { V3 o, d; float maxt; };
struct PI { float t = Infinity, u = 0, v = 0; uint32_t prim = 0, shape = 0, inst = 0xffffffffu; bool valid() const { return t != Infinity; } };

// ---------------------------------------------------------------------------
//  BVH (the oracle's own accel; reference uses Embree / kd-tree: kdtree.h:2206)
// ---------------------------------------------------------------------------

struct BuildPrim { float lo[3], hi[3], c[3]; uint32_t id; };

static void build_rec(std::vector<BvhNode> &nodes, std::vector<BuildPrim> &prims,
                      uint32_t begin, uint32_t end, uint32_t node_idx, uint32_t leaf_size) {
    BvhNode n{};
    for (int a = 0; a < 3; ++a) { n.lo[a] = Infinity; n.hi[a] = -Infinity; }
    float clo[3] = { Infinity, Infinity, Infinity }, chi[3] = { -Infinity, -Infinity, -Infinity };
    for (uint32_t i = begin; i < end; ++i)
        for (int a = 0; a < 3; ++a) {
            n.lo[a] = std::min(n.lo[a], prims[i].lo[a]); n.hi[a] = std::max(n.hi[a], prims[i].hi[a]);
            clo[a] = std::min(clo[a], prims[i].c[a]);    chi[a] = std::max(chi[a], prims[i].c[a]);
        }
    uint32_t count = end - begin;
    int axis = 0;
    for (int a = 1; a < 3; ++a) if (chi[a] - clo[a] > chi[axis] - clo[axis]) axis = a;
    if (count <= leaf_size || !(chi[axis] > clo[axis])) {
        n.left = begin; n.count = count; n.right = 0;
        nodes[node_idx] = n;
        return;
    }
    // binned SAH along the widest centroid axis
    constexpr int NB = 16;
    struct Bin { float lo[3], hi[3]; uint32_t n; } bins[NB];
    for (auto &b : bins) { for (int a = 0; a < 3; ++a) { b.lo[a] = Infinity; b.hi[a] = -Infinity; } b.n = 0; }
    float scale = NB / (chi[axis] - clo[axis]);
    auto bin_of = [&](const BuildPrim &p) { int b = (int) ((p.c[axis] - clo[axis]) * scale); return std::min(std::max(b, 0), NB - 1); };
    for (uint32_t i = begin; i < end; ++i) {
        Bin &b = bins[bin_of(prims[i])]; b.n++;
        for (int a = 0; a < 3; ++a) { b.lo[a] = std::min(b.lo[a], prims[i].lo[a]); b.hi[a] = std::max(b.hi[a], prims[i].hi[a]); }
    }
    auto area = [](const float *lo, const float *hi) { float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2]; return 2.f * (dx * dy + dy * dz + dz * dx); };
    float best = Infinity; int best_split = -1;
    float rlo[NB][3], rhi[NB][3]; uint32_t rn[NB];
    { float lo[3] = { Infinity, Infinity, Infinity }, hi[3] = { -Infinity, -Infinity, -Infinity }; uint32_t c = 0;
      for (int i = NB - 1; i >= 0; --i) { c += bins[i].n; for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], bins[i].lo[a]); hi[a] = std::max(hi[a], bins[i].hi[a]); rlo[i][a] = lo[a]; rhi[i][a] = hi[a]; } rn[i] = c; } }
    { float lo[3] = { Infinity, Infinity, Infinity }, hi[3] = { -Infinity, -Infinity, -Infinity }; uint32_t c = 0;
      for (int i = 0; i < NB - 1; ++i) { c += bins[i].n; for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], bins[i].lo[a]); hi[a] = std::max(hi[a], bins[i].hi[a]); }
          if (c == 0 || rn[i + 1] == 0) continue;
          float cost = c * area(lo, hi) + rn[i + 1] * area(rlo[i + 1], rhi[i + 1]);
          if (cost < best) { best = cost; best_split = i; } } }
    uint32_t mid;
    if (best_split < 0) {
        mid = begin + count / 2;
        std::nth_element(prims.begin() + begin, prims.begin() + mid, prims.begin() + end,
                         [axis](const BuildPrim &a, const BuildPrim &b) { return a.c[axis] < b.c[axis]; });
    } else {
        auto it = std::partition(prims.begin() + begin, prims.begin() + end,
                                 [&](const BuildPrim &p) { return bin_of(p) <= best_split; });
        mid = (uint32_t) (it - prims.begin());
        if (mid == begin || mid == end) mid = begin + count / 2;
    }
    uint32_t l = (uint32_t) nodes.size(); nodes.emplace_back(); nodes.emplace_back();
    n.left = l; n.right = l + 1; n.count = 0;
    nodes[node_idx] = n;
    build_rec(nodes, prims, begin, mid, l, leaf_size);
    build_rec(nodes, prims, mid, end, l + 1, leaf_size);
}

static void pad_box(float *lo, float *hi) {
    float m = 1.f;
    for (int a = 0; a < 3; ++a) m = std::max(m, std::max(std::fabs(lo[a]), std::fabs(hi[a])));
    float pad = 2e-5f * m;
    for (int a = 0; a < 3; ++a) { lo[a] -= pad; hi[a] += pad; }
}

static void build_tri_bvh(Bvh &bvh, const std::vector<Mesh> &meshes, uint32_t first, uint32_t count) {
    std::vector<BuildPrim> prims; std::vector<Tri> tris;
    for (uint32_t s = first; s < first + count; ++s) {
        const Mesh &m = meshes[s];
        for (uint32_t f = 0; f < m.nf; ++f) {
            V3 p[3];
            for (int k = 0; k < 3; ++k) { const float *v = &m.V[8 * (size_t) m.F[4 * f + k]]; p[k] = V3(v[0], v[1], v[2]); }
            Tri t; t.p0 = p[0]; t.e1 = p[1] - p[0]; t.e2 = p[2] - p[0]; t.prim = f; t.shape = s;
            BuildPrim b; b.id = (uint32_t) tris.size();
            for (int a = 0; a < 3; ++a) {
                b.lo[a] = std::min(p[0][a], std::min(p[1][a], p[2][a]));
                b.hi[a] = std::max(p[0][a], std::max(p[1][a], p[2][a]));
                b.c[a] = 0.5f * (b.lo[a] + b.hi[a]);
            }
            pad_box(b.lo, b.hi);
            prims.push_back(b); tris.push_back(t);
        }
    }
    bvh.nodes.clear(); bvh.tris.clear();
    if (prims.empty()) return;
    bvh.nodes.emplace_back();
    build_rec(bvh.nodes, prims, 0, (uint32_t) prims.size(), 0, 4);
    bvh.tris.resize(tris.size());
    for (size_t i = 0; i < prims.size(); ++i) bvh.tris[i] = tris[prims[i].id];
    for (int a = 0; a < 3; ++a) { bvh.lo[a] = bvh.nodes[0].lo[a]; bvh.hi[a] = bvh.nodes[0].hi[a]; }
}

/* Mesh::moeller_trumbore, include/mitsuba/render/mesh.h:1130-1155 */
static inline bool moeller_trumbore(const Ray &ray, const Tri &tr, float &t, float &u, float &v) {
    V3 pvec = cross(ray.d, tr.e2);
    float inv_det = rcp(dot(tr.e1, pvec));
    V3 tvec = ray.o - tr.p0;
    u = dot(tvec, pvec) * inv_det;
    bool active = u >= 0.f && u <= 1.f;
    V3 qvec = cross(tvec, tr.e1);
    v = dot(ray.d, qvec) * inv_det;
    active &= v >= 0.f && u + v <= 1.f;
    t = dot(tr.e2, qvec) * inv_det;
    active &= t >= 0.f && t <= ray.maxt;
    return active;
}

/* std::fmin / std::fmax (a NaN operand is dropped) without the call into libm that -fno-fast-math leaves in place: twelve of them per box were most of a ray query's time */
static inline float fmin_nan(float a, float b) { return b != b ? a : (a < b ? a : b); }
static inline float fmax_nan(float a, float b) { return b != b ? a : (a > b ? a : b); }
static inline bool slab(const float *lo, const float *hi, const V3 &o, const V3 &id, float tmax, float &tnear) {
    float t0 = 0.f, t1 = tmax;
    for (int a = 0; a < 3; ++a) {
        float ta = (lo[a] - o[a]) * id[a], tb = (hi[a] - o[a]) * id[a];
        float mn = fmin_nan(ta, tb), mx = fmax_nan(ta, tb);
        t0 = fmax_nan(t0, mn); t1 = fmin_nan(t1, mx);   // fmin / fmax drop NaNs (0 * inf)
    }
    tnear = t0;
    return t0 <= t1 * 1.0000005f;
}

/* closest hit with the brute-force tie rule of ShapeKDTree::ray_intersect_naive
 * (kdtree.h:2433-2460): a later primitive replaces an earlier one when
 * t <= maxt, i.e. on exact ties the larger (shape, prim) wins. */
static inline void consider(const Ray &ray, const Tri &tr, PI &pi, uint32_t inst) {
    float t, u, v;
    Ray r = ray; r.maxt = std::fmin(ray.maxt, pi.t);
    if (!moeller_trumbore(r, tr, t, u, v)) return;
    if (t == pi.t && pi.valid()) {
        // scene order: top-level shapes first (inst = -1 -> 0), then instances
        auto key = [](uint32_t in, uint32_t sh, uint32_t pr, int which) -> uint64_t {
            return which == 0 ? (uint64_t) (uint32_t) (in + 1u) : (((uint64_t) sh << 32) | pr); };
        uint64_t a0 = key(inst, tr.shape, tr.prim, 0), b0 = key(pi.inst, pi.shape, pi.prim, 0);
        uint64_t a1 = key(inst, tr.shape, tr.prim, 1), b1 = key(pi.inst, pi.shape, pi.prim, 1);
        bool later = a0 != b0 ? a0 > b0 : a1 > b1;
        if (!later) return;
    }
    pi.t = t; pi.u = u; pi.v = v; pi.prim = tr.prim; pi.shape = tr.shape; pi.inst = inst;
}

/* Depth first, NEARER child first (the entry distance of a child's box rides on the stack; a popped entry is dropped when the closest hit found since lies in front of it).  The order
 * only decides how soon the closest hit shrinks the interval: `consider` resolves equal distances by the primitives' keys, not by the order they are met in, and an occlusion
 * query is a yes / no -- the answers are those of any other visiting order, and of `brute` (the first version of this loop pushed left, then right, whatever the ray's direction:
 * same answers, three times the node visits of a closest-hit query). */
struct StackEntry { uint32_t node; float tnear; };
template <bool Shadow>
static bool traverse(const Bvh &bvh, const Ray &ray, PI &pi, uint32_t inst) {
    if (bvh.empty()) return false;
    V3 id(rcp(ray.d.x), rcp(ray.d.y), rcp(ray.d.z));
    StackEntry stack[128]; int sp = 0;
    { float tn; if (!slab(bvh.nodes[0].lo, bvh.nodes[0].hi, ray.o, id, std::fmin(ray.maxt, pi.t), tn)) return false; stack[sp++] = StackEntry{ 0u, tn }; }
    while (sp) {
        const StackEntry e = stack[--sp];
        if (!(e.tnear <= std::fmin(ray.maxt, pi.t) * 1.0000005f)) continue;      /* what slab() would answer now: its far distance only shrank through the interval's end */
        const BvhNode &n = bvh.nodes[e.node];
        if (n.count) {
            for (uint32_t i = 0; i < n.count; ++i) {
                const Tri &tr = bvh.tris[n.left + i];
                if (Shadow) { float t, u, v; if (moeller_trumbore(ray, tr, t, u, v)) return true; }
                else consider(ray, tr, pi, inst);
            }
        } else {
            const float tmax = std::fmin(ray.maxt, pi.t);
            float tl, tr;
            const bool hl = slab(bvh.nodes[n.left].lo, bvh.nodes[n.left].hi, ray.o, id, tmax, tl), hr = slab(bvh.nodes[n.right].lo, bvh.nodes[n.right].hi, ray.o, id, tmax, tr);
            if (hl && hr) {
                if (tl <= tr) { stack[sp++] = StackEntry{ n.right, tr }; stack[sp++] = StackEntry{ n.left, tl }; }
                else          { stack[sp++] = StackEntry{ n.left, tl };  stack[sp++] = StackEntry{ n.right, tr }; }
            } else if (hl) stack[sp++] = StackEntry{ n.left, tl };
            else if (hr) stack[sp++] = StackEntry{ n.right, tr };
        }
    }
    return false;
}

template <bool Shadow>
static bool brute(const Bvh &bvh, const Ray &ray, PI &pi, uint32_t inst) {
    for (const Tri &tr : bvh.tris) {
        if (Shadow) { float t, u, v; if (moeller_trumbore(ray, tr, t, u, v)) return true; }
        else consider(ray, tr, pi, inst);
    }
    return false;
}

/* Scene::ray_intersect_preliminary / ray_test (src/render/scene.cpp:216-238);
 * instances: Instance::ray_intersect_preliminary_impl (src/shapes/instance.cpp:121-132)
 * transforms the ray with to_world.inverse() and keeps `t`. */
template <bool Shadow>
static bool scene_trace(const Scene &sc, const Ray &ray, PI &pi, int mode) {
    if (mode == 0) { if (traverse<Shadow>(sc.top, ray, pi, 0xffffffffu) && Shadow) return true; }
    else           { if (brute<Shadow>(sc.top, ray, pi, 0xffffffffu) && Shadow) return true; }
    auto do_inst = [&](uint32_t i) -> bool {
        const OrcInstance &in = sc.instances[i];
        Ray r; r.o = xf_point(in.to_object, ray.o); r.d = xf_vector(in.to_object, ray.d); r.maxt = ray.maxt;
        const Bvh &b = sc.group_bvh[in.group];
        return mode == 0 ? traverse<Shadow>(b, r, pi, i) : brute<Shadow>(b, r, pi, i);
    };
    if (mode != 0 || sc.inst_nodes.empty()) {
        for (uint32_t i = 0; i < sc.instances.size(); ++i) if (do_inst(i) && Shadow) return true;
    } else {
        V3 id(rcp(ray.d.x), rcp(ray.d.y), rcp(ray.d.z));
        StackEntry stack[64]; int sp = 0;
        { float tn; if (slab(sc.inst_nodes[0].lo, sc.inst_nodes[0].hi, ray.o, id, std::fmin(ray.maxt, pi.t), tn)) stack[sp++] = StackEntry{ 0u, tn }; }
        while (sp) {                        /* nearer instance boxes first, as in traverse() */
            const StackEntry e = stack[--sp];
            if (!(e.tnear <= std::fmin(ray.maxt, pi.t) * 1.0000005f)) continue;
            const BvhNode &n = sc.inst_nodes[e.node];
            if (n.count) { for (uint32_t i = 0; i < n.count; ++i) if (do_inst(sc.inst_order[n.left + i]) && Shadow) return true; }
            else {
                const float tmax = std::fmin(ray.maxt, pi.t);
                float tl, tr;
                const bool hl = slab(sc.inst_nodes[n.left].lo, sc.inst_nodes[n.left].hi, ray.o, id, tmax, tl), hr = slab(sc.inst_nodes[n.right].lo, sc.inst_nodes[n.right].hi, ray.o, id, tmax, tr);
                if (hl && hr) {
                    if (tl <= tr) { stack[sp++] = StackEntry{ n.right, tr }; stack[sp++] = StackEntry{ n.left, tl }; }
                    else          { stack[sp++] = StackEntry{ n.left, tl };  stack[sp++] = StackEntry{ n.right, tr }; }
                } else if (hl) stack[sp++] = StackEntry{ n.left, tl };
                else if (hr) stack[sp++] = StackEntry{ n.right, tr };
            }
        }
    }
    return Shadow ? false : pi.valid();
}

// ---------------------------------------------------------------------------
//  Surface interaction
// ---------------------------------------------------------------------------

struct SI {
    float t = Infinity; V3 p, n, sn, ss, st, wi; float uv[2] = { 0, 0 };
    uint32_t mesh = 0; bool valid() const { return t != Infinity; }
    V3 to_local(V3 v) const { return V3(dot(v, ss), dot(v, st), dot(v, sn)); }               // frame.h:34
    V3 to_world(V3 v) const { return fmadd(sn, v.z, fmadd(st, v.y, ss * v.x)); }              // frame.h:39
};

/* Mesh::compute_surface_interaction (src/render/mesh.cpp:2255-2437),
 * Instance::compute_surface_interaction (src/shapes/instance.cpp:150-266),
 * SurfaceInteraction::finalize_surface_interaction (interaction.h:559-605). */
static SI compute_si(const Scene &sc, const Ray &ray_w, const PI &pi) {
    SI si;
    if (!pi.valid()) { si.wi = -ray_w.d; return si; }     // interaction.h:812-818
    const Mesh &m = sc.meshes[pi.shape];
    const uint32_t *f = &m.F[4 * (size_t) pi.prim];
    const float *r0 = &m.V[8 * (size_t) f[0]], *r1 = &m.V[8 * (size_t) f[1]], *r2 = &m.V[8 * (size_t) f[2]];
    V3 p0(r0[0], r0[1], r0[2]), p1(r1[0], r1[1], r1[2]), p2(r2[0], r2[1], r2[2]);
    float b1 = pi.u, b2 = pi.v, b0 = 1.f - b1 - b2;
    V3 e1 = p1 - p0, e2 = p2 - p0;
    si.p = fmadd(p0, b0, fmadd(p1, b1, p2 * b2));
    si.n = normalize(cross(e1, e2));                       // mesh.h:568-572
    si.t = pi.t;
    if (m.flags & 1u) {
        V3 n0(r0[3], r0[4], r0[5]), dn1 = V3(r1[3], r1[4], r1[5]) - n0, dn2 = V3(r2[3], r2[4], r2[5]) - n0;
        V3 n = fmadd(dn1, b1, fmadd(dn2, b2, n0));
        float il = rsqrt(squared_norm(n));
        si.sn = n * il;
    } else si.sn = si.n;
    if (m.flags & 2u) {
        float u0 = r0[6], v0 = r0[7], du0 = r1[6] - u0, dv0 = r1[7] - v0, du1 = r2[6] - u0, dv1 = r2[7] - v0;
        si.uv[0] = fmadd(du0, b1, fmadd(du1, b2, u0));
        si.uv[1] = fmadd(dv0, b1, fmadd(dv1, b2, v0));
    } else { si.uv[0] = b1; si.uv[1] = b2; }
    si.mesh = pi.shape;
    if (pi.inst != 0xffffffffu) {                          // instance.cpp:196-224
        const OrcInstance &in = sc.instances[pi.inst];
        si.p = xf_point(in.to_world, si.p);
        si.n = normalize(xf_normal(in.to_object, si.n));
        V3 n = xf_normal(in.to_object, si.sn);
        float inv_len = rcp(norm(n));
        si.sn = n * inv_len;
    }
    // finalize: sh_frame.s == 0 for `diffuse` (no tangents packed) => coordinate_system()
    coordinate_system(si.sn, si.ss, si.st);
    si.wi = si.to_local(-ray_w.d);
    return si;
}

/* Interaction::offset_p / spawn_ray / spawn_ray_to (interaction.h:161-191) */
static inline V3 offset_p(const SI &si, V3 d) {
    float mag = (1.f + hmax(vabs(si.p))) * RayEpsilon;
    mag = mulsign(mag, dot(si.n, d));
    return fmadd(si.n, mag, si.p);
}
static inline Ray spawn_ray(const SI &si, V3 d) { Ray r; r.o = offset_p(si, d); r.d = d; r.maxt = Largest; return r; }
static inline Ray spawn_ray_to(const SI &si, V3 t) {
    Ray r; r.o = offset_p(si, t - si.p);
    V3 d = t - r.o; float dist = norm(d);
    r.d = div(d, dist); r.maxt = dist * (1.f - ShadowEpsilon);
    return r;
}

// ---------------------------------------------------------------------------
//  Textures, BSDF, emitter
// ---------------------------------------------------------------------------

struct TexLookup { uint32_t idx[4]; float w[4]; };
/* dr::Texture<Float,2>::eval (ext/drjit texture.h, NOT IN TREE, parity unpinned; call site src/textures/bitmap.cpp:842-850).  Restated from Dr.Jit's published
 * behaviour: WrapMode::Repeat / Mirror / Clamp act on the INTEGER texel position -- repeat: pos mod res; mirror: the texture is flipped in every other
 * repetition (counting repetitions of negative positions from -1); clamp: clip to [0, res - 1] -- and FilterMode::Nearest reads the texel under floor(uv * res),
 * Linear interpolates the four texels around uv * res - 1/2. */
static inline uint32_t tex_wrap_pos(int64_t pos, int64_t res, uint32_t mode) {
    if (mode & 4u) return (uint32_t) std::min<int64_t>(std::max<int64_t>(pos, 0), res - 1);
    /* repetition index r (floor division) and offset m in [0, res) */
    int64_t r = pos >= 0 ? pos / res : -((-pos - 1) / res) - 1, m = pos - r * res;
    if ((mode & 2u) && (r & 1)) m = res - 1 - m;                                   /* odd repetitions (.. -3, -1, 1, 3 ..) run backwards */
    return (uint32_t) m;
}
static inline void tex_lookup(const Texture &t, const float uv_surface[2], TexLookup &l) {
    float uv[2]; t.to_texture_space(uv_surface, uv);                   /* uv = m_transform * si.uv (bitmap.cpp:565,792,831,847) */
    if (t.mode & 1u) {
        int64_t x = (int64_t) std::floor(uv[0] * (float) t.w), y = (int64_t) std::floor(uv[1] * (float) t.h);
        uint32_t i = tex_wrap_pos(y, t.h, t.mode) * t.w + tex_wrap_pos(x, t.w, t.mode);
        l.idx[0] = l.idx[1] = l.idx[2] = l.idx[3] = i; l.w[0] = 1.f; l.w[1] = 0.f; l.w[2] = 1.f; l.w[3] = 0.f;
        return;
    }
    float px = fmadd(uv[0], (float) t.w, -0.5f), py = fmadd(uv[1], (float) t.h, -0.5f);
    float fx = std::floor(px), fy = std::floor(py);
    int64_t ix = (int64_t) fx, iy = (int64_t) fy;
    float w1x = px - fx, w1y = py - fy, w0x = 1.f - w1x, w0y = 1.f - w1y;
    uint32_t x0 = tex_wrap_pos(ix, t.w, t.mode), x1 = tex_wrap_pos(ix + 1, t.w, t.mode);
    uint32_t y0 = tex_wrap_pos(iy, t.h, t.mode), y1 = tex_wrap_pos(iy + 1, t.h, t.mode);
    l.idx[0] = y0 * t.w + x0; l.idx[1] = y0 * t.w + x1; l.idx[2] = y1 * t.w + x0; l.idx[3] = y1 * t.w + x1;
    l.w[0] = w0x; l.w[1] = w1x; l.w[2] = w0y; l.w[3] = w1y;
}
static inline V3 tex_eval(const Texture &t, const TexLookup &l) {
    float out[3];
    if (t.mode & 1u) return V3(t.data[3 * (size_t) l.idx[0]], t.data[3 * (size_t) l.idx[0] + 1], t.data[3 * (size_t) l.idx[0] + 2]);
    for (int c = 0; c < 3; ++c) {
        float v00 = t.data[3 * (size_t) l.idx[0] + c], v10 = t.data[3 * (size_t) l.idx[1] + c],
              v01 = t.data[3 * (size_t) l.idx[2] + c], v11 = t.data[3 * (size_t) l.idx[3] + c];
        float v0 = fmadd(l.w[0], v00, l.w[1] * v10), v1 = fmadd(l.w[0], v01, l.w[1] * v11);
        out[c] = fmadd(l.w[2], v0, l.w[3] * v1);
    }
    return V3(out[0], out[1], out[2]);
}
static inline V3 bsdf_reflectance(const Scene &sc, const OrcBSDF &b, const SI &si) {   /* (kept for the diffuse-only KAT entry points) */
    if (b.texture < 0) return V3(b.reflectance[0], b.reflectance[1], b.reflectance[2]);  // srgb.cpp: m_value
    TexLookup l; tex_lookup(sc.textures[b.texture], si.uv, l);
    return tex_eval(sc.textures[b.texture], l);
}

/* TwoSidedBRDF (src/bsdfs/twosided.cpp:112-270) around the nested plugins of orc_bsdf.h.
 * `used` = record whose parameters are evaluated (front or back), for gradient book-keeping. */
struct BsdfCtx { const BsdfRecord *rec; uint32_t used; V3 wi; float wo_sign; bool ok; V3 slot0, slot1; TexLookup tl; bool textured; };
static inline BsdfCtx bsdf_prepare(const Scene &sc, uint32_t index, const SI &si) {
    BsdfCtx c; c.used = index; c.wi = si.wi; c.wo_sign = 1.f; c.ok = true; c.textured = false;
    const BsdfRecord *b = &sc.bsdfs[index];
    if (b->p.flags & 1u) {
        if (b->p.back < 0) { c.wo_sign = sign1(si.wi.z); c.wi.z = std::fabs(si.wi.z); }       // twosided.cpp:124-127
        else if (si.wi.z > 0.f) { }
        else if (si.wi.z < 0.f) { c.used = (uint32_t) b->p.back; b = &sc.bsdfs[c.used]; c.wi.z = -si.wi.z; c.wo_sign = -1.f; }
        else c.ok = false;
    }
    c.rec = b;
    if (b->p.texture >= 0) { c.textured = true; tex_lookup(sc.textures[b->p.texture], si.uv, c.tl); c.slot0 = tex_eval(sc.textures[b->p.texture], c.tl); }
    else c.slot0 = V3(b->p.reflectance[0], b->p.reflectance[1], b->p.reflectance[2]);
    c.slot1 = V3(b->p.reflectance2[0], b->p.reflectance2[1], b->p.reflectance2[2]);
    return c;
}
static inline BSDFEval bsdf_eval_pdf(const BsdfCtx &c, V3 wo) {
    if (!c.ok) return BSDFEval();
    return plugin_eval_pdf(*c.rec, c.slot0, c.slot1, c.wi, V3(wo.x, wo.y, wo.z * c.wo_sign));
}
static inline BSDFSample bsdf_sample(const BsdfCtx &c, float s1, float s2x, float s2y, V3 &weight) {
    if (!c.ok) { weight = V3(0.f); return BSDFSample(); }
    BSDFSample bs = plugin_sample(*c.rec, c.slot0, c.slot1, c.wi, s1, s2x, s2y, weight);
    bs.wo.z *= c.wo_sign;
    return bs;
}

/* SmoothDiffuse::eval_pdf (src/bsdfs/diffuse.cpp:159-179) */
static inline void diffuse_eval_pdf(V3 refl, V3 wi, V3 wo, V3 &value, float &pdf) {
    bool active = wi.z > 0.f && wo.z > 0.f;
    value = active ? (refl * InvPi) * wo.z : V3(0.f);
    pdf = active ? InvPi * wo.z : 0.f;
}
/* SmoothDiffuse::sample (src/bsdfs/diffuse.cpp:100-124) */
static inline void diffuse_sample(V3 refl, V3 wi, float s2x, float s2y, V3 &wo, float &pdf, V3 &weight) {
    wo = square_to_cosine_hemisphere(s2x, s2y);
    pdf = InvPi * wo.z;
    weight = (wi.z > 0.f && pdf > 0.f) ? refl : V3(0.f);
}

struct DS { V3 p, n, d; float dist = 0, pdf = 0; int emitter = -1; bool delta = false; float uv[2] = { 0.f, 0.f }; /* ds.uv: where a bitmap radiance was sampled (area.cpp:139-146) */ };

/* AreaLight::sample_direction (src/emitters/area.cpp:118-168) over
 * Shape::sample_direction (src/render/shape.cpp:93-110) and
 * Rectangle::sample_position (src/shapes/rectangle.cpp:159-173) */
/* ConstantBackgroundEmitter (src/emitters/constant.cpp): bounding sphere kept per scene */
struct EnvSphere { V3 center; float radius = 0.f; };
static inline V3 square_to_uniform_sphere(float sx, float sy) {          // warp.h:250-255
    float z = fnmadd(2.f, sy, 1.f), r = std::sqrt(std::fmax(fnmadd(z, z, 1.f), 0.f));
    float c, s = sincos(2.f * Pi * sx, &c);
    return V3(r * c, r * s, z);
}
constexpr float InvFourPi = 0.07957747154594766788f;
/* `unit` (optional) = d spec / d radiance: the weight the sample would carry for a unit radiance (prb adjoint w.r.t. the emitter colour) */
static inline void constant_sample_direction(const OrcEmitter &e, const EnvSphere &bs, V3 ref_p, float sx, float sy, DS &ds, V3 &spec, float *unit = nullptr) {   // constant.cpp:127-153
    V3 d = square_to_uniform_sphere(sx, sy);
    float radius = std::fmax(bs.radius, norm(ref_p - bs.center)), dist = 2.f * radius;
    ds.p = fmadd(d, dist, ref_p); ds.n = -d; ds.pdf = InvFourPi; ds.d = d; ds.dist = dist;
    spec = div(V3(e.radiance[0], e.radiance[1], e.radiance[2]), ds.pdf);
    if (unit) *unit = rcp(ds.pdf);
}
/* Mesh::sample_position (src/render/mesh.cpp:1662-1712) with DiscreteDistribution::sample_reuse (include/mitsuba/core/distr_1d.h:117-183; the JIT
 * predicate of `sample`, dr::binary_search over [0, n - 1]) and warp::square_to_uniform_triangle (warp.h:153-156) */
/* DiscreteDistribution::sample (distr_1d.h:117-140, JIT branch): value *= sum; dr::binary_search over [0, n - 1] with the predicate
 * (cdf[i] < value || cdf[i] == 0) && cdf[i] != sum -- the first bucket whose running sum reaches the value, skipping empty buckets at either end */
static inline uint32_t discrete_sample(const float *cdf, uint32_t n, float sum, float value01, uint32_t first = 0, uint32_t last = 0xffffffffu, bool scalar_variant = false) {
    const float value = value01 * sum;
    uint32_t start = first, end = last == 0xffffffffu ? n - 1 : last, iterations = 0;      /* m_valid: [0, n - 1] for tables built on the device (compute_cdf), first / last bin with mass for compute_cdf_scalar */
    if (start < end) { uint32_t span = end - start; iterations = 1; while (span >>= 1) ++iterations; }
    for (uint32_t i = 0; i < iterations; ++i) {
        uint32_t middle = (start + end) >> 1;
        float c = cdf[middle];
        bool cond = scalar_variant ? c < value                                   /* distr_1d.h:126-127: the non-JIT predicate */
                                   : ((c < value) || c == 0.f) && c != sum;
        if (cond) start = std::min(middle + 1, end); else end = middle;
    }
    return start;
}
/* DiscreteDistribution::sample_reuse_pmf (distr_1d.h:159-183): the index, the re-used sample (value - cdf_normalized[index - 1]) / pmf_normalized[index] and the pmf */
static inline uint32_t discrete_sample_reuse(const float *pmf, const float *cdf, uint32_t n, float sum, float normalization, float value01, float &reused, float &pmf_out,
                                             uint32_t first = 0, uint32_t last = 0xffffffffu, bool scalar_variant = false) {
    const uint32_t idx = discrete_sample(cdf, n, sum, value01, first, last, scalar_variant);
    const float pmf_n = pmf[idx] * normalization, cdf_n = idx > 0 ? cdf[idx - 1] * normalization : 0.f;
    reused = (value01 - cdf_n) / pmf_n; pmf_out = pmf_n;
    return idx;
}
static inline void mesh_sample_position(const Mesh &m, const Scene::AreaPmf &d, float sx, float sy, V3 &p, V3 &n, float &pdf) {
    const uint32_t nf = (uint32_t) d.pmf.size();
    float pmf_n;
    const uint32_t idx = discrete_sample_reuse(d.pmf.data(), d.cdf.data(), nf, d.sum, d.normalization, sy, sy, pmf_n);
    const uint32_t *f = m.F.data() + 4 * (size_t) idx;
    const float *v0 = m.V.data() + 8 * (size_t) f[0], *v1 = m.V.data() + 8 * (size_t) f[1], *v2 = m.V.data() + 8 * (size_t) f[2];
    V3 p0(v0[0], v0[1], v0[2]), p1(v1[0], v1[1], v1[2]), p2(v2[0], v2[1], v2[2]);
    V3 e0 = p1 - p0, e1 = p2 - p0;
    float t = std::sqrt(std::fmax(1.f - sx, 0.f)), bx = 1.f - t, by = t * sy;
    p = fmadd(e0, bx, fmadd(e1, by, p0));
    if (m.flags & 1u) {
        V3 n0(v0[3], v0[4], v0[5]), n1(v1[3], v1[4], v1[5]), n2(v2[3], v2[4], v2[5]);
        n = fmadd(n0, 1.f - bx - by, fmadd(n1, bx, n2 * by));
    } else n = cross(e0, e1);
    n = normalize(n);
    pdf = d.normalization;
}
/* PointLight::sample_direction (src/emitters/point.cpp:119-148): the position is to_world[9..11], `radiance` holds the radiant intensity */
static inline void point_sample_direction(const OrcEmitter &e, V3 ref_p, DS &ds, V3 &spec, float *unit) {
    ds.p = V3(e.to_world[9], e.to_world[10], e.to_world[11]); ds.n = V3(0.f); ds.pdf = 1.f; ds.delta = true;
    ds.d = ds.p - ref_p;
    const float dist2 = squared_norm(ds.d), inv_dist = rsqrt(dist2);
    ds.dist = std::sqrt(dist2);
    ds.d = ds.d * inv_dist;
    const float w = sqr(inv_dist);
    spec = V3(e.radiance[0], e.radiance[1], e.radiance[2]) * w;
    if (unit) *unit = w;
}
/* SpotLight (src/emitters/spot.cpp): update() (:300-312), falloff_curve (:143-151), sample_direction (:177-211).  to_world / to_local = the emitter's transform and
 * its inverse, normal[0] = cutoff_angle, normal[1] = beam_width (degrees); `radiance` = the radiant intensity along the axis.  No `texture`. */
static inline void spot_sample_direction(const OrcEmitter &e, V3 ref_p, DS &ds, V3 &spec, float *unit) {
    const float deg = 0.017453292519943295f;                                  // dr::deg_to_rad: value * (Pi / 180)
    const float cutoff_rad = e.normal[0] * deg, beam_rad = e.normal[1] * deg;
    const float inv_transition_width = 1.0f / (cutoff_rad - beam_rad);
    float cos_cutoff, cos_beam; sincos(cutoff_rad, &cos_cutoff); sincos(beam_rad, &cos_beam);
    ds.p = V3(e.to_world[9], e.to_world[10], e.to_world[11]); ds.n = V3(0.f); ds.pdf = 1.f; ds.delta = true;
    ds.d = ds.p - ref_p;
    ds.dist = norm(ds.d);
    const float inv_dist = rcp(ds.dist);
    ds.d = ds.d * inv_dist;
    const V3 local_dir = normalize(xf_vector(e.to_local, -ds.d));              // falloff_curve: dr::normalize(to_world.inverse() * -ds.d)
    const float cos_theta = local_dir.z;
    const float beam_res = cos_theta >= cos_beam ? 1.f : (cutoff_rad - acos32(cos_theta)) * inv_transition_width;
    const float falloff = cos_theta > cos_cutoff ? beam_res : 0.f;
    const bool active = falloff > 0.f;
    const float w = falloff * sqr(inv_dist);
    spec = active ? V3(e.radiance[0], e.radiance[1], e.radiance[2]) * w : V3(0.f);
    if (unit) *unit = active ? w : 0.f;
}
static inline void emitter_sample_direction(const Scene &sc, uint32_t index, V3 ref_p, float sx, float sy, DS &ds, V3 &spec, float *unit = nullptr) {
    const OrcEmitter &e = sc.emitters[index];
    if (e.type == 4) { point_sample_direction(e, ref_p, ds, spec, unit); return; }
    if (e.type == 5) { spot_sample_direction(e, ref_p, ds, spec, unit); return; }
    if (e.type == 6) {          // DirectionalEmitter::sample_direction (directional.cpp:149-176): d = to_world * (0, 0, 1) = the third column; `radiance` = the irradiance
        const V3 d(e.to_world[6], e.to_world[7], e.to_world[8]);
        const float radius = std::fmax(sc.env_radius, norm(ref_p - V3(sc.env_center[0], sc.env_center[1], sc.env_center[2])));
        const float dist = 2.f * radius;
        ds.p = ref_p - d * dist; ds.n = d; ds.pdf = 1.f; ds.delta = true; ds.d = -d; ds.dist = dist;
        spec = V3(e.radiance[0], e.radiance[1], e.radiance[2]);
        if (unit) *unit = 1.f;
        return;
    }
    if (e.type == 7) {          // AreaLight::sample_direction, spatially varying radiance (area.cpp:133-165)
        const Texture &t = sc.textures[e.radiance_texture]; const TexelTable &tab = sc.texel_tables[index];
        float uv[2], pdf;
        bitmap_sample_position(t, tab, sx, sy, uv, pdf);
        bool active = pdf != 0.f;
        ds.p = xf_point(e.to_world, V3(fmadd(uv[0], 2.f, -1.f), fmadd(uv[1], 2.f, -1.f), 0.f));      // Rectangle::eval_parameterization (rectangle.cpp:215-237)
        ds.n = V3(e.normal[0], e.normal[1], e.normal[2]);
        ds.d = ds.p - ref_p;
        const float dist2 = squared_norm(ds.d);
        ds.dist = std::sqrt(dist2);
        ds.d = div(ds.d, ds.dist);
        const float dp = dot(ds.d, ds.n);
        active = active && dp < 0.f;
        ds.pdf = active ? pdf / tab.span * dist2 / -dp : 0.f;
        TexLookup l; tex_lookup(t, uv, l);                                                           // m_radiance->eval(si) at si.uv = uv
        spec = active ? div(tex_eval(t, l), ds.pdf) : V3(0.f);
        ds.uv[0] = uv[0]; ds.uv[1] = uv[1];
        if (unit) *unit = active ? rcp(ds.pdf) : 0.f;                                                // the weight per unit of radiance(ds.uv): the bitmap's texels are the parameter (area.cpp:64-70)
        return;
    }
    if (e.type == 3) mesh_sample_position(sc.meshes[e.mesh], sc.area_pmf[index], sx, sy, ds.p, ds.n, ds.pdf);
    else {          // Rectangle::sample_position (rectangle.cpp:159-170)
        ds.p = xf_point(e.to_world, V3(fmadd(sx, 2.f, -1.f), fmadd(sy, 2.f, -1.f), 0.f));
        ds.n = V3(e.normal[0], e.normal[1], e.normal[2]);
        ds.pdf = e.inv_area;
    }
    ds.d = ds.p - ref_p;
    float dist2 = squared_norm(ds.d);
    ds.dist = std::sqrt(dist2);
    ds.d = div(ds.d, ds.dist);
    float dp = std::fabs(dot(ds.d, ds.n));
    float x = dist2 / dp;
    ds.pdf *= std::isfinite(x) ? x : 0.f;
    bool active = dot(ds.d, ds.n) < 0.f && ds.pdf != 0.f;
    V3 rad(e.radiance[0], e.radiance[1], e.radiance[2]);
    spec = active ? div(rad, ds.pdf) : V3(0.f);
    if (unit) *unit = active ? rcp(ds.pdf) : 0.f;
}
/* AreaLight::pdf_direction (area.cpp:170-197) over Shape::pdf_direction (shape.cpp:112-124) */
static inline float emitter_pdf_direction(const OrcEmitter &e, const DS &ds) {
    float dp = dot(ds.d, ds.n);
    if (!(dp < 0.f)) return 0.f;
    float adp = std::fabs(dp);
    float pdf = e.inv_area;
    pdf *= (adp != 0.f) ? (ds.dist * ds.dist) / adp : 0.f;
    return pdf;
}

/* the two quantities an emitter HIT needs, for every surface emitter: AreaLight::pdf_direction (area.cpp:170-197; the textured branch :185-191 evaluates
 * pdf_position(ds.uv) * dist^2 / (|dp_du x dp_dv| * -dp)) and AreaLight::eval (area.cpp:83-90) */
static inline float surface_emitter_pdf_direction(const Scene &sc, uint32_t index, const DS &ds, const float uv[2]) {
    const OrcEmitter &e = sc.emitters[index];
    if (e.type != 7) return emitter_pdf_direction(e, ds);
    const float dp = dot(ds.d, ds.n);
    if (!(dp < 0.f)) return 0.f;
    const TexelTable &tab = sc.texel_tables[index];
    return bitmap_pdf_position(sc.textures[e.radiance_texture], tab, uv) * sqr(ds.dist) / (tab.span * -dp);
}
static inline V3 surface_emitter_radiance(const Scene &sc, const OrcEmitter &e, const float uv[2]) {
    if (e.type != 7) return V3(e.radiance[0], e.radiance[1], e.radiance[2]);
    TexLookup l; tex_lookup(sc.textures[e.radiance_texture], uv, l);
    return tex_eval(sc.textures[e.radiance_texture], l);
}

/* d (bitmap lookup at uv) / d texels, times g: the transpose of tex_eval -- what reverse mode sends to the texels of a bitmap `radiance` (area.cpp:64-70: a differentiable
 * traverse entry; the texel distribution the samples were drawn from is detached, prb.py:174-175) */
template <typename Commit>
static inline void tex_scatter(const Texture &t, const float uv[2], Commit commit) {
    TexLookup l; tex_lookup(t, uv, l);
    if (t.mode & 1u) { commit(l.idx[0], 1.f); return; }
    const float wts[4] = { l.w[0] * l.w[2], l.w[1] * l.w[2], l.w[0] * l.w[3], l.w[1] * l.w[3] };
    for (int k = 0; k < 4; ++k) commit(l.idx[k], wts[k]);
}

/* PathIntegrator::mis_weight (src/integrators/path.cpp:359-364), common.py:1344 */
static inline float mis_weight(float a, float b) {
    a *= a; b *= b;
    float w = a / (a + b);
    return std::isfinite(w) ? w : 0.f;
}

/* Scene::update_emitter_sampling_distribution (scene.cpp:120-141) -> DiscreteDistribution(const ScalarFloat *, size) -> compute_cdf_scalar (distr_1d.h:236-266); false: a
 * negative weight or no probability mass at all (the reference throws) */
static bool rebuild_emitter_choice(Scene &sc) {
    Scene::EmitterChoice ch;
    for (const OrcEmitter &e : sc.emitters) { if (!(e.sampling_weight >= 0.f)) return false; ch.weighted = ch.weighted || e.sampling_weight != 1.f; }
    if (ch.weighted) {
        double running = 0.0; bool seen = false;
        for (uint32_t i = 0; i < sc.emitters.size(); ++i) {
            const float w = sc.emitters[i].sampling_weight;
            running += (double) w; ch.table.pmf.push_back(w); ch.table.cdf.push_back((float) running);
            if (w > 0.f) { if (!seen) ch.first = i; ch.last = i; seen = true; }
        }
        if (!seen) return false;                                        /* "no probability mass found!" */
        ch.table.sum = ch.table.cdf[ch.last]; ch.table.normalization = rcp(ch.table.sum);
    }
    sc.choice = ch;
    return true;
}

/* Scene::pdf_emitter (scene.cpp:273-279) / the emitter_pmf of pdf_emitter_direction (:378-388): m_emitter_pmf = 1 / n, or weight * normalization of the distribution */
static inline float emitter_choice_pmf(const Scene &sc, uint32_t index) {
    if (!sc.choice.weighted) return 1.f / (float) sc.emitters.size();
    return sc.choice.table.pmf[index] * sc.choice.table.normalization;
}

/* Scene::sample_emitter_direction (src/render/scene.cpp:316-366), JIT branch */
static inline bool sample_emitter_direction(const Scene &sc, const SI &si, float sx, float sy, DS &ds, V3 &spec,
                                            OrcStats &st, Ray *shadow_out = nullptr, float *unit = nullptr, bool scalar_variant = false) {
    uint32_t n = (uint32_t) sc.emitters.size();
    if (n == 0) { ds = DS(); spec = V3(0.f); return false; }
    uint32_t index = 0; float weight = 1.f, pmf = 1.f / (float) n;
    const Scene::EmitterChoice &ch = sc.choice;
    if (n > 1 && ch.weighted) {                    // sample_emitter with m_emitter_distr (scene.cpp:258-261): sample_reuse_pmf, emitter_weight = rcp(pmf)
        float chosen_pmf;
        index = discrete_sample_reuse(ch.table.pmf.data(), ch.table.cdf.data(), n, ch.table.sum, ch.table.normalization, sx, sx, chosen_pmf, ch.first, ch.last, scalar_variant);
        weight = rcp(chosen_pmf);
    } else if (n > 1) {                            // sample_emitter, scene.cpp:248-271
        float scaled = sx * (float) n;
        index = std::min((uint32_t) scaled, n - 1u);
        weight = (float) n; sx = scaled - (float) index;
    }
    /* pdf_emitter (scene.cpp:273-279): eval_pmf_normalized -- in JIT variants also when there is one emitter (:326); scalar variants with one emitter take the
       branch at :351-354, which samples it directly and applies no selection probability */
    if (ch.weighted) pmf = (scalar_variant && n == 1) ? 1.f : emitter_choice_pmf(sc, index);
    if (unit) *unit = 0.f;
    if (sc.emitters[index].type == 1) { EnvSphere bs; bs.center = V3(sc.env_center[0], sc.env_center[1], sc.env_center[2]); bs.radius = sc.env_radius; constant_sample_direction(sc.emitters[index], bs, si.p, sx, sy, ds, spec, unit); }
    else if (sc.emitters[index].type == 2) {       // EnvironmentMapEmitter::sample_direction (envmap.cpp:284-323)
        float uv[2];
        sc.envmap.sample_direction(si.p, sx, sy, ds.d, ds.dist, ds.pdf, spec, uv);
        ds.p = fmadd(ds.d, ds.dist, si.p); ds.n = -ds.d;
    }
    else emitter_sample_direction(sc, index, si.p, sx, sy, ds, spec, unit);
    ds.emitter = (int) index;
    ds.pdf *= pmf;
    spec = spec * weight;
    if (unit) *unit *= weight;
    if (ds.pdf != 0.f) {
        Ray r = spawn_ray_to(si, ds.p);
        if (shadow_out) *shadow_out = r;
        PI dummy; st.shadow_rays++;
        if (scene_trace<true>(sc, r, dummy, 0)) { spec = V3(0.f); ds.pdf = 0.f; if (unit) *unit = 0.f; }
    }
    return true;
}

// ---------------------------------------------------------------------------
//  Sensor, film
// ---------------------------------------------------------------------------

/* PerspectiveCamera::sample_ray (src/sensors/perspective.cpp:200-237) */
static inline Ray sensor_sample_ray(const OrcSensor &s, float px, float py) {
    const float *M = s.sample_to_camera;
    float r[4];
    for (int i = 0; i < 4; ++i) r[i] = M[4 * i + 3];
    /* scaled_principal_point_offset = film size * offset / crop size, added to the position sample (perspective.cpp:213-221) */
    const float sppx = (float) s.film_width * s.principal_point_offset_x / (float) s.crop_width, sppy = (float) s.film_height * s.principal_point_offset_y / (float) s.crop_height;
    float arg[3] = { px + sppx, py + sppy, 0.f };
    for (int j = 0; j < 3; ++j) for (int i = 0; i < 4; ++i) r[i] = fmadd(M[4 * i + j], arg[j], r[i]);   // transform.h:337-345
    if (s.projection == 1) {      // OrthographicCamera::sample_ray (orthographic.cpp:131-157): an affine sample_to_camera, rays from the near plane along to_world's +z
        const float *T = s.to_world;
        Ray ray;
        V3 o(T[3], T[7], T[11]);
        for (int j = 0; j < 3; ++j) o = V3(fmadd(T[j], r[j], o.x), fmadd(T[4 + j], r[j], o.y), fmadd(T[8 + j], r[j], o.z));      // to_world * near_p (a point)
        ray.o = o;
        ray.d = normalize(V3(T[2], T[6], T[10]));                                                                           // to_world * (0, 0, 1)
        ray.maxt = s.far_clip - s.near_clip;
        return ray;
    }
    float iw = rcp(r[3]);
    V3 near_p(r[0] * iw, r[1] * iw, r[2] * iw);
    V3 d = normalize(near_p);
    const float *T = s.to_world;
    Ray ray;
    ray.o = V3(T[3], T[7], T[11]);
    V3 dw(T[0] * d.x, T[4] * d.x, T[8] * d.x);
    dw = V3(fmadd(T[1], d.y, dw.x), fmadd(T[5], d.y, dw.y), fmadd(T[9], d.y, dw.z));
    dw = V3(fmadd(T[2], d.z, dw.x), fmadd(T[6], d.z, dw.y), fmadd(T[10], d.z, dw.z));
    ray.d = dw;
    float inv_z = rcp(d.z);
    float near_t = s.near_clip * inv_z, far_t = s.far_clip * inv_z;
    ray.o = ray.o + ray.d * near_t;
    ray.maxt = far_t - near_t;
    return ray;
}

struct RFilter { uint32_t type; float radius; float coeff[10]; float p0, p1; };
/* GaussianFilter ctor + eval, LLVM branch (src/rfilters/gaussian.cpp:48-101) */
static inline float estrin10(float x, const float *c) {
    float x2 = x * x, x4 = x2 * x2, x8 = x4 * x4;
    float a0 = fmadd(x, c[1], c[0]), a1 = fmadd(x, c[3], c[2]), a2 = fmadd(x, c[5], c[4]), a3 = fmadd(x, c[7], c[6]), a4 = fmadd(x, c[9], c[8]);
    float b0 = fmadd(x2, a1, a0), b1 = fmadd(x2, a3, a2), b2 = a4;
    float c0 = fmadd(x4, b1, b0), c1 = b2;
    return fmadd(x8, c1, c0);
}
static RFilter make_rfilter(uint32_t type, float stddev, float param1 = 0.f) {
    RFilter f{}; f.type = type; f.p0 = stddev; f.p1 = param1;
    if (type == 0) { f.radius = 0.5f; return f; }
    if (type == 2) { f.radius = stddev; return f; }                       // TentFilter (tent.cpp:48-52)
    if (type == 3 || type == 4) { f.radius = 2.f; return f; }            // MitchellNetravaliFilter (mitchell.cpp:50-58), CatmullRomFilter
    if (type == 5) { f.radius = stddev; return f; }                       // LanczosSincFilter (lanczos.cpp:52-55): radius = lobes
    f.radius = 4 * stddev;
    const double coeff[10] = { 9.992604880e-1, -4.977025247e-1, 1.222248550e-1, -1.932406282e-2, 2.136713061e-3,
                               -1.679873860e-4, 9.202145248e-6, -3.329417433e-7, 7.128382794e-9, -6.821193280e-11 };
    double scale = 1;
    for (int i = 0; i < 10; ++i) { f.coeff[i] = (float) (coeff[i] * scale); scale /= (double) stddev * (double) stddev; }
    f.coeff[0] -= estrin10(f.radius * f.radius, f.coeff);
    return f;
}
static inline float rfilter_eval(const RFilter &f, float x) {
    switch (f.type) {
    case 2: return std::fmax(0.f, 1.f - std::fabs(x * (1.f / f.radius)));                        // tent.cpp:54-56
    case 3: {                                                                                     // mitchell.cpp:60-79
        x = std::fabs(x); float x2 = x * x, x3 = x2 * x; const float B = f.p0, Cc = f.p1;
        float a3 = (12.f - 9.f * B - 6.f * Cc), a2 = (-18.f + 12.f * B + 6.f * Cc), a0 = (6.f - 2.f * B),
              b3 = (-B - 6.f * Cc), b2 = (6.f * B + 30.f * Cc), b1 = (-12.f * B - 48.f * Cc), b0 = (8.f * B + 24.f * Cc);
        float r = (1.f / 6.f) * (x < 1.f ? fmadd(a3, x3, fmadd(a2, x2, a0)) : fmadd(b3, x3, fmadd(b2, x2, fmadd(b1, x, b0))));
        return x < 2.f ? r : 0.f;
    }
    case 4: {                                                                                     // catmullrom.cpp:39-54 (B = 0, C = 1/2, no fmadd)
        x = std::fabs(x); float x2 = x * x, x3 = x2 * x; const float B = 0.f, Cc = .5f;
        float r = (1.f / 6.f) * (x < 1.f ? (12.f - 9.f * B - 6.f * Cc) * x3 + (-18.f + 12.f * B + 6.f * Cc) * x2 + (6.f - 2.f * B)
                                         : (-B - 6.f * Cc) * x3 + (6.f * B + 30.f * Cc) * x2 + (-12.f * B - 48.f * Cc) * x + (8.f * B + 24.f * Cc));
        return x < 2.f ? r : 0.f;
    }
    case 5: {                                                                                     // lanczos.cpp:57-67
        x = std::fabs(x);
        float x1 = Pi * x, x2 = x1 / f.radius, c1, c2;
        float r = (sincos(x1, &c1) * sincos(x2, &c2)) / (x1 * x2);
        return x < Epsilon ? 1.f : (x > f.radius ? 0.f : r);
    }
    default: return std::fmax(estrin10(x * x, f.coeff), 0.f);
    }
}

/* ImageBlock::put (src/render/imageblock.cpp:187-258 box, :444-540 coalesced JIT) */
static inline void film_put(const OrcSensor &s, const RFilter &rf, float px, float py, const float v[4], float *film) {
    uint32_t W = s.crop_width, H = s.crop_height;
    if (rf.type == 0) {
        int32_t x = (int32_t) std::floor(px) - (int32_t) s.crop_offset_x, y = (int32_t) std::floor(py) - (int32_t) s.crop_offset_y;
        if ((uint32_t) x < W && (uint32_t) y < H) { float *p = film + 4 * ((size_t) y * W + x); for (int k = 0; k < 4; ++k) p[k] += v[k]; }
        return;
    }
    uint32_t n = (uint32_t) std::ceil(rf.radius - .5f), count = 2 * n + 1;
    int32_t ix = (int32_t) std::floor(px) - (int32_t) n, iy = (int32_t) std::floor(py) - (int32_t) n;
    uint32_t x0 = (uint32_t) (ix - (int32_t) s.crop_offset_x), y0 = (uint32_t) (iy - (int32_t) s.crop_offset_y);
    float relx = ((float) ix + .5f) - px, rely = ((float) iy + .5f) - py;
    for (uint32_t ys = 0; ys < count; ++ys) {
        float wy = rfilter_eval(rf, rely + (float) ys);
        uint32_t y = y0 + ys;
        if (!(y < H)) continue;
        for (uint32_t xs = 0; xs < count; ++xs) {
            uint32_t x = x0 + xs;
            if (!(x < W)) continue;
            float wx = rfilter_eval(rf, relx + (float) xs), w = wx * wy;
            float *p = film + 4 * ((size_t) y * W + x);
            for (int k = 0; k < 4; ++k) p[k] += v[k] * w;
        }
    }
}

// ---------------------------------------------------------------------------
//  Lane -> sample mapping (src/render/integrator.cpp:322-339, 448-520)
// ---------------------------------------------------------------------------

/* Film::sample_border: film_size = crop_size + 2 * rfilter->border_size() (integrator.cpp:162-165), border_size = ceil(radius - 1/2 - 2 RayEpsilon)
 * (rfilter.cpp:22); the lane map runs over that grid and is shifted back by the border (integrator.cpp:333-334) */
static inline uint32_t sample_border_size(const OrcSensor &s) {
    if (!s.sample_border) return 0;
    RFilter rf = make_rfilter(s.rfilter, s.rfilter_stddev, s.rfilter_param1);
    int b = (int) std::ceil(rf.radius - .5f - 2.f * RayEpsilon);
    return b > 0 ? (uint32_t) b : 0u;
}
static inline uint64_t sample_grid_pixels(const OrcSensor &s) {
    const uint64_t b = sample_border_size(s);
    return (s.crop_width + 2 * b) * (s.crop_height + 2 * b);
}

struct Lane { Pcg32 rng; float pos_x, pos_y, ipos_x, ipos_y; Ray ray; };
static inline Lane make_lane(const OrcSensor &s, uint32_t seed, uint32_t spp, uint64_t idx) {
    Lane L;
    L.rng = sampler_seed(seed, (uint32_t) idx);
    uint32_t lspp = 0; while ((1u << (lspp + 1)) <= spp) ++lspp;
    uint32_t p = ((1u << lspp) == spp) ? (uint32_t) idx >> lspp : (uint32_t) idx / spp;
    const uint32_t border = sample_border_size(s), grid_w = s.crop_width + 2 * border;
    uint32_t y = p / grid_w, x = p - grid_w * y;
    float jx = L.rng.next_float32(), jy = L.rng.next_float32();
    L.ipos_x = (float) ((int32_t) (x + s.crop_offset_x) - (int32_t) border); L.ipos_y = (float) ((int32_t) (y + s.crop_offset_y) - (int32_t) border);
    L.pos_x = L.ipos_x + jx;
    L.pos_y = L.ipos_y + jy;
    float sx = 1.f / (float) s.crop_width, sy = 1.f / (float) s.crop_height;
    float ox = -(float) s.crop_offset_x * sx, oy = -(float) s.crop_offset_y * sy;
    L.ray = sensor_sample_ray(s, fmadd(L.pos_x, sx, ox), fmadd(L.pos_y, sy, oy));
    return L;
}

// ---------------------------------------------------------------------------
//  PathIntegrator::sample (src/integrators/path.cpp:94-346), JIT semantics
// ---------------------------------------------------------------------------

/* hide_emitters: a camera ray that hits an area emitter continues through ALL area emitters along it
 * (path.cpp:177-190, prb.py:112-118; Integrator::skip_area_emitters, src/render/integrator.cpp:96-124).  Only the preliminary
 * intersection is replaced: the loop keeps the camera ray (position and normal come from the barycentric coordinates,
 * si.wi from the unchanged direction). */
static void skip_area_emitters(const Scene &sc, const Ray &ray, PI &pi, OrcStats &st) {
    if (!(pi.valid() && sc.meshes[pi.shape].emitter >= 0)) return;
    SI si = compute_si(sc, ray, pi);
    Ray r = spawn_ray(si, ray.d);
    for (;;) {
        PI q; st.closest_rays++; scene_trace<false>(sc, r, q, 0);
        pi = q;
        if (!(q.valid() && sc.meshes[q.shape].emitter >= 0)) return;
        SI s2 = compute_si(sc, r, q);
        r = spawn_ray(s2, r.d);
    }
}

static V3 path_sample(const Scene &sc, Pcg32 &rng, Ray ray, uint32_t max_depth, uint32_t rr_depth, bool &valid_ray, OrcStats &st, bool scalar = false) {
    V3 throughput(1.f), result(0.f);
    float eta = 1.f; uint32_t depth = 0; valid_ray = !sc.hide_emitters && sc.env >= 0;      // path.cpp:114-115
    V3 prev_p(0.f); float prev_bsdf_pdf = 1.f; bool prev_bsdf_delta = true;
    if (max_depth == 0) return V3(0.f);
    PI pi; st.closest_rays++; scene_trace<false>(sc, ray, pi, 0);
    if (sc.hide_emitters) skip_area_emitters(sc, ray, pi, st);
    bool active = true;
    while (active) {
        st.vertices++;
        SI si = compute_si(sc, ray, pi);
        int emitter = si.valid() ? sc.meshes[si.mesh].emitter : sc.env;     // si.emitter(scene), scene.h:822-832
        if (emitter >= 0) {                                   // path.cpp:206-221
            DS ds; ds.p = si.p; ds.n = si.sn;                 // records.h:77-79,173-180
            V3 rel = si.p - prev_p; ds.dist = norm(rel); ds.d = si.valid() ? div(rel, ds.dist) : -si.wi;
            const OrcEmitter &e = sc.emitters[emitter];
            float em_pdf = 0.f;
            if (!prev_bsdf_delta) em_pdf = (e.type == 1 ? InvFourPi : e.type == 2 ? sc.envmap.pdf_direction(ds.d) : surface_emitter_pdf_direction(sc, (uint32_t) emitter, ds, si.uv)) * emitter_choice_pmf(sc, (uint32_t) emitter);
            float mis_bsdf = mis_weight(prev_bsdf_pdf, em_pdf);
            bool facing = (e.type != 0 && e.type != 3 && e.type != 7) || si.wi.z > 0.f;                                                    // area.cpp:83-90, constant.cpp:90-94
            V3 rad = e.type == 2 ? sc.envmap.eval(-si.wi) : surface_emitter_radiance(sc, e, si.uv);                       // envmap.cpp:228-236
            V3 Le = (facing && prev_bsdf_pdf > 0.f) ? rad : V3(0.f);
            result = fmadd(throughput, Le * mis_bsdf, result);
        }
        bool active_next = (depth + 1 < max_depth) && si.valid();
        valid_ray |= si.valid();                              // path.cpp:307-308 (JIT: evaluated every iteration)
        if (!active_next) {
            /* JIT: the lane still executes the rest of this iteration masked; its (unmasked) sampler draws advance the stream by 6,
               which only matters when a later pass continues it (integrator.cpp:349-356).  Scalar: `break` at path.cpp:226-227. */
            if (!scalar) for (int k = 0; k < 6; ++k) rng.next_uint32();
            break;
        }
        BsdfCtx bsdf = bsdf_prepare(sc, sc.meshes[si.mesh].bsdf, si);          // si.bsdf(ray), path.cpp:232
        // emitter sampling, path.cpp:236-258: the samples are drawn by every lane, used where the BSDF is Smooth
        bool active_em = bsdf.rec->smooth();
        /* JIT variants: `if (dr::any_or<true>(active_em))` is always taken, every lane draws the two samples;
           scalar variants evaluate the real condition (path.cpp:244-249) */
        float ex = 0.f, ey = 0.f;
        if (!scalar || active_em) { ex = rng.next_float32(); ey = rng.next_float32(); }
        DS ds; V3 em_weight(0.f), wo(0.f);
        if (active_em) active_em = sample_emitter_direction(sc, si, ex, ey, ds, em_weight, st, nullptr, nullptr, scalar);
        active_em &= ds.pdf != 0.f;
        if (active_em) wo = si.to_local(ds.d);
        float s1 = rng.next_float32();
        float s2x = rng.next_float32(), s2y = rng.next_float32();
        BSDFEval ev = bsdf_eval_pdf(bsdf, wo);                                 // bsdf.cpp:21-31 eval_pdf_sample
        V3 bsdf_weight; BSDFSample bs = bsdf_sample(bsdf, s1, s2x, s2y, bsdf_weight);
        if (active_em) {                                      // path.cpp:271-281
            float mis_em = ds.delta ? 1.f : mis_weight(ds.pdf, ev.pdf);              // path.cpp:274: dr::select(ds.delta, 1.f, mis_weight(ds.pdf, bsdf_pdf))
            result = fmadd(throughput, (ev.value * em_weight) * mis_em, result);
        }
        ray = spawn_ray(si, si.to_world(bs.wo));              // path.cpp:287
        throughput = throughput * bsdf_weight; eta *= bs.eta;
        prev_p = si.p; prev_bsdf_pdf = bs.pdf; prev_bsdf_delta = bs.delta;
        depth += 1;                                           // path.cpp:317 (si valid here)
        float tmax = hmax(throughput);
        float rr_prob = std::fmin(tmax * sqr(eta), .95f);
        bool rr_active = depth >= rr_depth, rr_continue = rng.next_float32() < rr_prob;
        if (rr_active) throughput = throughput * rcp(rr_prob);
        active = active_next && (!rr_active || rr_continue) && (tmax != 0.f);
        if (active) { pi = PI(); st.closest_rays++; scene_trace<false>(sc, ray, pi, 0); }
    }
    return valid_ray ? result : V3(0.f);
}

// ---------------------------------------------------------------------------
//  PRBIntegrator.sample (src/python/python/ad/integrators/prb.py:68-339)
//  mode Primal:   returns L.
//  mode Backward: L_in = primal L, dL = film adjoint; accumulates the
//  hand-derived gradients of SURVEY.md Appendix B into `grad`.
// ---------------------------------------------------------------------------

// ---------------------------------------------------------------------------
//  Geometry-attached part of PRBIntegrator.sample (prb.py:124-141, 176-216, 261-297) for `diffuse` BSDFs:
//  derivatives w.r.t. the vertex positions of flat-shaded top-level meshes, by forward-mode duals (orc_dual.h).
//  Slots 0..8 = coordinates of the three vertices of the triangle hit at the current vertex (the next interaction is detached).
// ---------------------------------------------------------------------------

/* slots 0..11: the CURRENT vertex (9 vertex coordinates of a top-level triangle, or the 12 entries of an instance's to_world);
 * slots 12..23: the same for the PREVIOUS vertex, whose motion the attached si.wi follows (prb.py:128-140) */
constexpr int kShapeSlots = 33;
constexpr int kPrevSlot = 12;
/* slots 24..32: the three VERTEX NORMALS of the current triangle on a mesh with vertex normals.  The reference regenerates them from the positions whenever
 * the positions are written (mesh.cpp:876-878 -> pack(regenerate_normals) -> compute_normals, :1216-1267), so they are differentiable functions of the whole
 * one-ring; the oracle takes the derivative in two stages like reverse mode would: per path vertex w.r.t. the three normals (accumulated per mesh vertex),
 * then once per face through compute_normals (normals_backward below). */
constexpr int kNormalSlot = 24;
typedef Dual<kShapeSlots> Dn;
typedef Dual3<kShapeSlots> Dn3;
static inline Dn3 dn3(V3 v) { return Dn3((double) v.x, (double) v.y, (double) v.z); }

struct AttachedSI { Dn3 p, n, sn; Dn uv[2]; bool diff = false, smooth = false; uint32_t mesh = 0, vid[3] = { 0, 0, 0 }; uint32_t inst = 0xffffffffu; /* != none: the slots are to_world[0..11] of this instance */ };

/* Mesh::compute_surface_interaction with AD-attached vertex positions (src/render/mesh.cpp:2286-2323) and
 * SurfaceInteraction::attach_motion without FollowShape (include/mitsuba/render/interaction.h:525-545): the point stays on the
 * (detached) ray and follows the moving tangent plane; the barycentric coordinates pick up the motion of that point relative to
 * the triangle.  `shading` = false is RayFlags::Minimal (p, t, n only). */
static AttachedSI attach_si(const Scene &sc, const Ray &ray, const PI &pi, const SI &si, const uint8_t *mask, int slot, bool shading, const uint8_t *inst_mask = nullptr) {
    AttachedSI a; a.p = dn3(si.p); a.n = dn3(si.n); a.sn = dn3(si.sn); a.uv[0] = Dn((double) si.uv[0]); a.uv[1] = Dn((double) si.uv[1]);
    if (pi.valid() && pi.inst != 0xffffffffu && inst_mask && inst_mask[pi.inst]) {
        /* Instance::compute_surface_interaction with an attached `to_world` (src/shapes/instance.cpp:150-266): the nested interaction is computed
         * with gradients suspended (:181-189), "hit point si.p is only attached to the surface motion" (:191-193: si.p = to_world * si.p, the normals
         * use dr::detach(to_world)), and without FollowShape the point is re-intersected with the moving tangent plane (:240-249):
         * si.t = (<n, p> - <n, o>) / <n, d>, si.p = ray(si.t).  uv and both normals stay detached (the TODOs at :250-251). */
        const OrcInstance &in = sc.instances[pi.inst];
        const Mesh &m = sc.meshes[pi.shape];
        const uint32_t *f = &m.F[4 * (size_t) pi.prim];
        const float *r0 = &m.V[8 * (size_t) f[0]], *r1 = &m.V[8 * (size_t) f[1]], *r2 = &m.V[8 * (size_t) f[2]];
        const double b1 = pi.u, b2 = pi.v, b0 = 1.0 - b1 - b2;
        const double po[3] = { r0[0] * b0 + r1[0] * b1 + r2[0] * b2, r0[1] * b0 + r1[1] * b1 + r2[1] * b2, r0[2] * b0 + r1[2] * b1 + r2[2] * b2 };   /* object-space hit point */
        Dn M[12]; for (int k = 0; k < 12; ++k) M[k] = Dn::param(in.to_world[k], slot + k);
        Dn3 p_att(M[0] * po[0] + M[3] * po[1] + M[6] * po[2] + M[9], M[1] * po[0] + M[4] * po[1] + M[7] * po[2] + M[10], M[2] * po[0] + M[5] * po[1] + M[8] * po[2] + M[11]);
        Dn3 nd = dn3(si.n), o = dn3(ray.o), d = dn3(ray.d);
        Dn t_att = ddot(p_att - o, nd) / ddot(nd, d);
        a.p = replace_grad3(si.p.x, si.p.y, si.p.z, o + d * t_att);
        a.diff = true; a.inst = pi.inst; a.mesh = pi.shape;
        return a;
    }
    if (!pi.valid() || !mask || !mask[pi.shape]) return a;
    const Mesh &m = sc.meshes[pi.shape];
    const uint32_t *f = &m.F[4 * (size_t) pi.prim];
    a.diff = true; a.mesh = pi.shape; a.vid[0] = f[0]; a.vid[1] = f[1]; a.vid[2] = f[2];
    Dn3 P[3];
    for (int k = 0; k < 3; ++k) {
        const float *r = &m.V[8 * (size_t) f[k]];
        P[k] = Dn3(Dn::param(r[0], slot + 3 * k), Dn::param(r[1], slot + 3 * k + 1), Dn::param(r[2], slot + 3 * k + 2));
    }
    double b1 = pi.u, b2 = pi.v, b0 = 1.0 - b1 - b2;
    Dn3 e1 = P[1] - P[0], e2 = P[2] - P[0];
    Dn3 p_att = P[0] * b0 + P[1] * b1 + P[2] * b2;
    Dn3 n_geo = dnormalize(dcross(e1, e2));                       // face_normal
    /* A mesh INSIDE a shape group (differentiated vertex positions shared by all instances; the instance's to_world is detached -- the reference refuses both at
     * once, instance.cpp:162-166): the nested Mesh::compute_surface_interaction runs in OBJECT space on to_object * ray (instance.cpp:181-189), then
     * si.p = to_world * si.p, si.n = normalize(to_world * si.n), sh_frame.n likewise (:191-204).  `wp` / `wn` apply the detached transforms to attached values. */
    const bool nested = pi.inst != 0xffffffffu;
    const OrcInstance *in = nested ? &sc.instances[pi.inst] : nullptr;
    auto wp = [&](const Dn3 &q) { if (!nested) return q; const float *M = in->to_world;
        return Dn3(q.x * (double) M[0] + q.y * (double) M[3] + q.z * (double) M[6] + Dn((double) M[9]), q.x * (double) M[1] + q.y * (double) M[4] + q.z * (double) M[7] + Dn((double) M[10]),
                   q.x * (double) M[2] + q.y * (double) M[5] + q.z * (double) M[8] + Dn((double) M[11])); };
    auto wn = [&](const Dn3 &q) { if (!nested) return q; const float *T = in->to_object;          /* inverse transpose, then normalize */
        return dnormalize(Dn3(q.x * (double) T[0] + q.y * (double) T[1] + q.z * (double) T[2], q.x * (double) T[3] + q.y * (double) T[4] + q.z * (double) T[5],
                              q.x * (double) T[6] + q.y * (double) T[7] + q.z * (double) T[8])); };
    V3 ro = ray.o, rd = ray.d;
    if (nested) { ro = xf_point(in->to_object, ray.o); rd = xf_vector(in->to_object, ray.d); }
    Dn3 nd = dvalue(n_geo);                                       // dr::detach(n) (object space for a nested mesh)
    Dn3 o = dn3(ro), d = dn3(rd);
    Dn t_att = ddot(p_att - o, nd) / ddot(nd, d);
    Dn3 p_ray = o + d * t_att;                                    // ray(t)
    a.p = replace_grad3(si.p.x, si.p.y, si.p.z, wp(p_ray));
    a.n = replace_grad3(si.n.x, si.n.y, si.n.z, wn(n_geo));
    if (!shading) return a;
    // mesh.cpp:2308-2321: rel = si.p - p_att has a zero value, only its derivative matters
    Dn3 rel = replace_grad3(0.0, 0.0, 0.0, p_ray - p_att);
    Dn a11 = ddot(e1, e1), a12 = ddot(e1, e2), a22 = ddot(e2, e2), inv_det = Dn(1.0) / (a11 * a22 - a12 * a12);
    Dn r1 = ddot(e1, rel), r2 = ddot(e2, rel);
    Dn b1d = replace_grad(b1, Dn(b1) + (a22 * r1 - a12 * r2) * inv_det);
    Dn b2d = replace_grad(b2, Dn(b2) + (a11 * r2 - a12 * r1) * inv_det);
    a.sn = a.n;                                                   // meshes without vertex normals: sh_frame.n = n (mesh.cpp:2367-2369)
    if (m.flags & 1u) {
        /* mesh.cpp:2346-2356: n = normalize(n0 + (n1 - n0) b1 + (n2 - n0) b2) with the attached barycentrics and the attached (regenerated) vertex normals */
        Dn3 N[3];
        for (int k = 0; k < 3; ++k) {
            const float *r = &m.V[8 * (size_t) f[k]];
            N[k] = Dn3(Dn::param(r[3], kNormalSlot + 3 * k), Dn::param(r[4], kNormalSlot + 3 * k + 1), Dn::param(r[5], kNormalSlot + 3 * k + 2));
        }
        Dn3 nn = N[0] + (N[1] - N[0]) * b1d + (N[2] - N[0]) * b2d;
        a.sn = replace_grad3(si.sn.x, si.sn.y, si.sn.z, nested ? wn(dnormalize(nn)) : dnormalize(nn));
        a.smooth = true;
    }
    if (m.flags & 2u) {
        const float *r0 = &m.V[8 * (size_t) f[0]], *r1v = &m.V[8 * (size_t) f[1]], *r2v = &m.V[8 * (size_t) f[2]];
        double u0 = r0[6], v0 = r0[7], du0 = r1v[6] - u0, dv0 = r1v[7] - v0, du1 = r2v[6] - u0, dv1 = r2v[7] - v0;
        a.uv[0] = replace_grad((double) si.uv[0], b1d * du0 + b2d * du1 + Dn(u0));
        a.uv[1] = replace_grad((double) si.uv[1], b1d * dv0 + b2d * dv1 + Dn(v0));
    } else { a.uv[0] = b1d; a.uv[1] = b2d; }
    return a;
}

/* bilinear, repeat-wrapped texture lookup (tex_lookup / tex_eval above) as a function of attached texture coordinates */
static inline Dn3 tex_eval_dual(const Texture &t, const TexLookup &l, const Dn uv[2]) {
    if (t.mode & 1u) return Dn3(Dn((double) t.data[3 * (size_t) l.idx[0]]), Dn((double) t.data[3 * (size_t) l.idx[0] + 1]), Dn((double) t.data[3 * (size_t) l.idx[0] + 2]));     /* nearest: constant in uv */
    /* the lookup position in texture space: m_transform * uv, carried with its derivative (an affine map: the duals go through the linear part) */
    const Dn tu = uv[0] * (double) t.xf[0][0] + uv[1] * (double) t.xf[0][1] + Dn((double) t.xf[0][2]), tv = uv[0] * (double) t.xf[1][0] + uv[1] * (double) t.xf[1][1] + Dn((double) t.xf[1][2]);
    Dn px = tu * (double) t.w - Dn(0.5), py = tv * (double) t.h - Dn(0.5);
    Dn w1x = replace_grad((double) l.w[1], px), w1y = replace_grad((double) l.w[3], py), w0x = Dn(1.0) - w1x, w0y = Dn(1.0) - w1y;
    Dn out[3];
    for (int c = 0; c < 3; ++c) {
        double v00 = t.data[3 * (size_t) l.idx[0] + c], v10 = t.data[3 * (size_t) l.idx[1] + c],
               v01 = t.data[3 * (size_t) l.idx[2] + c], v11 = t.data[3 * (size_t) l.idx[3] + c];
        out[c] = w0y * (w0x * v00 + w1x * v10) + w1y * (w0x * v01 + w1x * v11);
    }
    return Dn3(out[0], out[1], out[2]);
}

/* solid_angle_to_area_jacobian (ad/integrators/common.py:1355-1384) and the direction it is evaluated along */
static inline void dir_and_jacobian(const Dn3 &o, const Dn3 &p, const Dn3 &n, Dn3 &dir, Dn &J) {
    Dn3 d = p - o; Dn d2 = ddot(d, d);
    dir = dnormalize(d);
    J = dabs(ddot(n, dir)) / d2;
}

/* finalize_surface_interaction (interaction.h:570-600) with an attached shading normal: meshes pack no tangents here, sh_frame.s stays zero and the
 * frame is coordinate_system(sh_frame.n) (vector.h:118-138) -- written over duals, so the tangents follow the normal exactly as the reference's AD sees it */
struct AttachedFrame { Dn3 s, t, n; };
static AttachedFrame attach_frame(const AttachedSI &a, const SI &si) {
    AttachedFrame f; const Dn3 &n = a.sn;
    const double sign = n.z.v >= 0.0 ? 1.0 : -1.0;                 /* dr::sign / mulsign: constants under differentiation */
    Dn av = Dn(-1.0) / (Dn(sign) + n.z), b = n.x * n.y * av;
    Dn3 s((n.x * n.x * av) * sign + Dn(1.0), b * sign, n.x * (-sign)), t(b, n.y * n.y * av + Dn(sign), -n.y);
    f.s = replace_grad3(si.ss.x, si.ss.y, si.ss.z, s); f.t = replace_grad3(si.st.x, si.st.y, si.st.z, t); f.n = replace_grad3(si.sn.x, si.sn.y, si.sn.z, n);
    return f;
}
static inline Dn3 frame_to_local(const AttachedFrame &f, const Dn3 &v) { return Dn3(ddot(v, f.s), ddot(v, f.t), ddot(v, f.n)); }

/* BSDF::eval (value = f * cos theta_o) of the plugins with a non-delta lobe -- diffuse.cpp:159-179, roughconductor.cpp:429-520, roughplastic.cpp:296-336,
 * plastic.cpp:318-352 over microfacet.h:185-207,341-365 and fresnel.h:35-116 -- in DOUBLE precision as a function of the two LOCAL directions.  The oracle
 * differentiates it numerically (central differences): the attached `si.wi` / `wo` of prb.py:128-140,276-288 reach the BSDF through these six numbers.
 * (wi0, wo0) = the unperturbed directions: every dr::select of the plugin takes the branch THEY take, as the reference's AD does. */
static double fresnel_r_d(double cos_i, double eta) {
    const bool outside = cos_i >= 0.0; const double eta_it = outside ? eta : 1.0 / eta, eta_ti = outside ? 1.0 / eta : eta;
    const double ct2 = 1.0 - (1.0 - cos_i * cos_i) * eta_ti * eta_ti, ci = std::fabs(cos_i), ct = std::sqrt(std::max(ct2, 0.0));
    if (eta == 1.0) return 0.0;
    if (ci == 0.0) return 1.0;
    const double a_s = (ci - eta_it * ct) / (ci + eta_it * ct), a_p = (ct - eta_it * ci) / (ct + eta_it * ci);
    return 0.5 * (a_s * a_s + a_p * a_p);
}
static void bsdf_value_dir_d(const BsdfRecord &b, const double s0[3], const double s1[3], const double wi[3], const double wo[3], const double wi0[3], const double wo0[3], double out[3]) {
    out[0] = out[1] = out[2] = 0.0;
    const uint32_t type = b.p.type;
    if (!(wi0[2] > 0.0 && wo0[2] > 0.0)) return;                      /* every model here is one-sided */
    if (type == 0) { for (int c = 0; c < 3; ++c) out[c] = s0[c] * (double) InvPi * wo[2]; return; }
    auto diffuse_base = [&](double k) {                               /* value / (1 - fdr_int * value) or value / (1 - fdr_int) */
        for (int c = 0; c < 3; ++c) { const double den = b.nonlinear() ? 1.0 - s0[c] * (double) b.internal_reflectance : 1.0 - (double) b.internal_reflectance; out[c] += s0[c] / den * k; }
    };
    if (type == 5) {
        diffuse_base((double) InvPi * wo[2] * (double) b.inv_eta_2 * (1.0 - fresnel_r_d(wi[2], b.p.eta)) * (1.0 - fresnel_r_d(wo[2], b.p.eta)));
        return;
    }
    if (type != 2 && type != 3) return;
    auto half = [](const double a[3], const double c[3], double H[3]) { for (int k = 0; k < 3; ++k) H[k] = a[k] + c[k]; const double l = std::sqrt(H[0] * H[0] + H[1] * H[1] + H[2] * H[2]); for (int k = 0; k < 3; ++k) H[k] /= l; };
    auto dot = [](const double a[3], const double c[3]) { return a[0] * c[0] + a[1] * c[1] + a[2] * c[2]; };
    double H[3], H0[3]; half(wi, wo, H); half(wi0, wo0, H0);
    const bool ggx = b.mtype() == MicrofacetType::GGX;
    const double au = std::max((double) b.p.alpha_u, 1e-4), av = type == 3 ? au : std::max((double) b.p.alpha_v, 1e-4);
    auto Dm = [&](const double m[3]) {
        const double c2 = m[2] * m[2], q = (m[0] / au) * (m[0] / au) + (m[1] / av) * (m[1] / av);
        return ggx ? 1.0 / (M_PI * au * av * (q + c2) * (q + c2)) : std::exp(-q / c2) / (M_PI * au * av * c2 * c2);
    };
    auto G1 = [&](const double v[3], const double v0[3]) {
        const double xy = (au * v[0]) * (au * v[0]) + (av * v[1]) * (av * v[1]), t = xy / (v[2] * v[2]);
        const double xy0 = (au * v0[0]) * (au * v0[0]) + (av * v0[1]) * (av * v0[1]), t0 = xy0 / (v0[2] * v0[2]);
        double r;
        if (!ggx) { const double a = 1.0 / std::sqrt(t); r = 1.0 / std::sqrt(t0) >= 1.6 ? 1.0 : (3.535 * a + 2.181 * a * a) / (1.0 + 2.276 * a + 2.577 * a * a); }
        else r = 2.0 / (1.0 + std::sqrt(1.0 + t));
        if (xy0 == 0.0) r = 1.0;
        if (dot(v0, H0) * v0[2] <= 0.0) r = 0.0;
        return r;
    };
    const double D = Dm(H0) * H0[2] > 1e-20 ? Dm(H) : 0.0;
    const double wih = dot(wi, H);
    if (type == 2) {
        if (!(dot(wi0, H0) > 0.0 && dot(wo0, H0) > 0.0) || D == 0.0) return;
        const double V = D * G1(wi, wi0) * G1(wo, wo0) / (4.0 * wi[2]);
        auto Fc = [&](double c, double er, double ei) {
            const double c2 = c * c, s2 = 1.0 - c2, s4 = s2 * s2, t1 = er * er - ei * ei - s2, ab = std::sqrt(std::max(t1 * t1 + 4.0 * ei * ei * er * er, 0.0));
            const double a = std::sqrt(std::max(0.5 * (ab + t1), 0.0)), T1 = ab + c2, T2 = 2.0 * c * a, rs = (T1 - T2) / (T1 + T2), T3 = ab * c2 + s4, T4 = T2 * s2;
            return 0.5 * (rs + rs * (T3 - T4) / (T3 + T4));
        };
        for (int c = 0; c < 3; ++c) out[c] = s0[c] * Fc(wih, b.p.eta_c[c], b.p.k_c[c]) * V;
        return;
    }
    const double spec = fresnel_r_d(wih, b.p.eta) * D * G1(wi, wi0) * G1(wo, wo0) / (4.0 * wi[2]);
    auto table = [&](double x, double x0) {                           /* lerp_gather: the cell is the one the unperturbed argument falls into */
        const std::vector<float> &d = b.external_transmittance; const size_t n = d.size();
        x *= (double) (n - 1); x0 *= (double) (n - 1);
        const uint32_t i = std::min<uint32_t>((uint32_t) x0, (uint32_t) (n - 2));
        return (double) d[i] + ((double) d[i + 1] - (double) d[i]) * (x - (double) i);
    };
    for (int c = 0; c < 3; ++c) out[c] = s1[c] * spec;
    diffuse_base((double) InvPi * (double) b.inv_eta_2 * wo[2] * table(wi[2], wi0[2]) * table(wo[2], wo0[2]));
}
/* value and its six directional partials: d[c][j] = d value_c / d wi_j (j < 3), d wo_(j-3) (j >= 3) */
struct DirGrad { double d[3][6]; };
static void bsdf_dir_grad_fd(const BsdfRecord &b, V3 slot0, V3 slot1, V3 wi, V3 wo, DirGrad &g) {
    const double s0[3] = { slot0.x, slot0.y, slot0.z }, s1[3] = { slot1.x, slot1.y, slot1.z };
    const double x0[6] = { wi.x, wi.y, wi.z, wo.x, wo.y, wo.z };
    const double h = 1e-6;
    for (int j = 0; j < 6; ++j) {
        double xp[6], xm[6], fp[3], fm[3];
        for (int k = 0; k < 6; ++k) { xp[k] = x0[k]; xm[k] = x0[k]; }
        xp[j] += h; xm[j] -= h;
        bsdf_value_dir_d(b, s0, s1, xp, xp + 3, x0, x0 + 3, fp);
        bsdf_value_dir_d(b, s0, s1, xm, xm + 3, x0, x0 + 3, fm);
        for (int c = 0; c < 3; ++c) g.d[c][j] = (fp[c] - fm[c]) / (2.0 * h);
    }
}

struct ShapeSink { double *const *pos; const uint8_t *mask; double *inst = nullptr; const uint8_t *inst_mask = nullptr; /* 12 per instance: d / d to_world (column-major 3x4) */
                   double *const *nrm = nullptr; /* per mesh with vertex normals: d / d (vertex normal), 3 per vertex (first stage of the two-stage derivative) */ };
static inline void shape_scatter(const ShapeSink &sk, const AttachedSI &a, int slot, const double g[kShapeSlots]) {
    if (!a.diff) return;
    if (a.inst != 0xffffffffu) { for (int k = 0; k < 12; ++k) sk.inst[12 * (size_t) a.inst + k] += g[slot + k]; return; }
    double *dst = sk.pos[a.mesh];
    for (int k = 0; k < 3; ++k) for (int c = 0; c < 3; ++c) dst[3 * (size_t) a.vid[k] + c] += g[slot + 3 * k + c];
    if (slot == 0 && a.smooth && sk.nrm && sk.nrm[a.mesh]) {
        double *dn = sk.nrm[a.mesh];
        for (int k = 0; k < 3; ++k) for (int c = 0; c < 3; ++c) dn[3 * (size_t) a.vid[k] + c] += g[kNormalSlot + 3 * k + c];
    }
}

/* Mesh::compute_normals (src/render/mesh.cpp:1216-1267): angle-weighted vertex normals (Thuermer & Wuethrich 1998), written over a scalar type so that the
 * same text regenerates the normals (T = double values) and differentiates one face's contributions (T = Dual<9>: the nine coordinates of the face).
 * corner[k] = unit face normal x interior angle at corner k; returns false for a face without area. */
template <typename T> struct T3 { T x, y, z; };
template <typename T> static inline T tsqrt(const T &a);
template <> inline double tsqrt<double>(const double &a) { return std::sqrt(a); }
template <> inline Dual<9> tsqrt<Dual<9>>(const Dual<9> &a) { return dsqrt(a); }
static inline double tasin(double x) { return std::asin(x); }
static inline Dual<9> tasin(const Dual<9> &x) { Dual<9> r; r.v = std::asin(x.v); const double k = 1.0 / std::sqrt(1.0 - x.v * x.v); for (int i = 0; i < 9; ++i) r.d[i] = x.d[i] * k; return r; }
static inline double tval(double x) { return x; }
static inline double tval(const Dual<9> &x) { return x.v; }
template <typename T> static bool face_corner_normals(const T P[3][3], T3<T> corner[3]) {
    auto sub = [](const T a[3], const T b[3], T o[3]) { for (int c = 0; c < 3; ++c) o[c] = a[c] - b[c]; };
    auto dot = [](const T a[3], const T b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
    T e1[3], e2[3]; sub(P[1], P[0], e1); sub(P[2], P[0], e2);
    T n[3] = { e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0] };
    T l2 = dot(n, n);
    if (!(tval(l2) > 0.0)) return false;
    T il = T(1.0) / tsqrt(l2);
    for (int c = 0; c < 3; ++c) n[c] = n[c] * il;
    for (int k = 0; k < 3; ++k) {
        T u[3], v[3]; sub(P[(k + 1) % 3], P[k], u); sub(P[(k + 2) % 3], P[k], v);
        T iu = T(1.0) / tsqrt(dot(u, u)), iv = T(1.0) / tsqrt(dot(v, v));
        for (int c = 0; c < 3; ++c) { u[c] = u[c] * iu; v[c] = v[c] * iv; }
        /* dr::unit_angle: 2 asin(|v - u| / 2), or pi - 2 asin(|v + u| / 2) for an obtuse angle */
        T d = dot(u, v); const bool acute = tval(d) >= 0.0;
        T w[3]; for (int c = 0; c < 3; ++c) w[c] = acute ? v[c] - u[c] : v[c] + u[c];
        T t = tasin(tsqrt(dot(w, w)) * T(0.5)) * T(2.0);
        T angle = acute ? t : T(M_PI) - t;
        corner[k].x = n[0] * angle; corner[k].y = n[1] * angle; corner[k].z = n[2] * angle;
    }
    return true;
}
/* accumulated (un-normalised) vertex normals in double */
static void mesh_normal_sums(const Mesh &m, std::vector<double> &acc) {
    acc.assign(3 * (size_t) m.nv, 0.0);
    for (uint32_t f = 0; f < m.nf; ++f) {
        const uint32_t *fi = &m.F[4 * (size_t) f];
        double P[3][3]; for (int k = 0; k < 3; ++k) for (int c = 0; c < 3; ++c) P[k][c] = m.V[8 * (size_t) fi[k] + c];
        T3<double> cn[3];
        if (!face_corner_normals<double>(P, cn)) continue;
        for (int k = 0; k < 3; ++k) { acc[3 * (size_t) fi[k]] += cn[k].x; acc[3 * (size_t) fi[k] + 1] += cn[k].y; acc[3 * (size_t) fi[k] + 2] += cn[k].z; }
    }
}
/* regenerate the vertex normals of a mesh from its positions (what Mesh::parameters_changed does when the positions are written, mesh.cpp:876-878) */
static void mesh_regenerate_normals(Mesh &m) {
    std::vector<double> acc; mesh_normal_sums(m, acc);
    for (uint32_t v = 0; v < m.nv; ++v) {
        const double *a = &acc[3 * (size_t) v]; const double l2 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
        float *o = &m.V[8 * (size_t) v + 3];
        if (l2 > 0.0) { const double il = 1.0 / std::sqrt(l2); o[0] = (float) (a[0] * il); o[1] = (float) (a[1] * il); o[2] = (float) (a[2] * il); }
        else { o[0] = 1.f; o[1] = 0.f; o[2] = 0.f; }
    }
}
/* second stage: nbar[v] = d objective / d (unit vertex normal v)  ->  gpos += d objective / d positions, through normalize(sum over the vertex's corners) */
static void normals_backward(const Mesh &m, const double *nbar, double *gpos) {
    std::vector<double> acc; mesh_normal_sums(m, acc);
    std::vector<double> abar(3 * (size_t) m.nv, 0.0);           /* adjoint of the accumulated sums: (nbar - n <n, nbar>) / |acc| */
    for (uint32_t v = 0; v < m.nv; ++v) {
        const double *a = &acc[3 * (size_t) v]; const double l2 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
        if (!(l2 > 0.0)) continue;
        const double il = 1.0 / std::sqrt(l2), n[3] = { a[0] * il, a[1] * il, a[2] * il }, *b = &nbar[3 * (size_t) v];
        const double nb = n[0] * b[0] + n[1] * b[1] + n[2] * b[2];
        for (int c = 0; c < 3; ++c) abar[3 * (size_t) v + c] = (b[c] - n[c] * nb) * il;
    }
    for (uint32_t f = 0; f < m.nf; ++f) {
        const uint32_t *fi = &m.F[4 * (size_t) f];
        Dual<9> P[3][3]; for (int k = 0; k < 3; ++k) for (int c = 0; c < 3; ++c) P[k][c] = Dual<9>::param(m.V[8 * (size_t) fi[k] + c], 3 * k + c);
        T3<Dual<9>> cn[3];
        if (!face_corner_normals<Dual<9>>(P, cn)) continue;
        for (int k = 0; k < 3; ++k) {
            const double *ab = &abar[3 * (size_t) fi[k]];
            for (int s = 0; s < 9; ++s) gpos[3 * (size_t) fi[s / 3] + s % 3] += ab[0] * cn[k].x.d[s] + ab[1] * cn[k].y.d[s] + ab[2] * cn[k].z.d[s];
        }
    }
}

struct GradSink { float *refl; float *const *tex; float *emit; /* 3 per emitter (radiance of `area` / `constant`), may be null */
                  const ShapeSink *shape = nullptr; /* vertex-position gradients of the meshes in its mask, may be null */
                  /* FORWARD mode (RBIntegrator.render_forward, common.py:497-623; prb.py:313 `dL += dr.forward_to(Lo)`): refl / tex / emit hold the
                   * parameters' tangents (read only) and every derivative term is contracted with them into *fwd; the caller passes dL = 1 */
                  V3 *fwd = nullptr;
                  /* gradients w.r.t. alpha_u, alpha_v, eta (RGB), k (RGB), colour slot 1 of the rough models: 15 floats per BSDF record, may be null */
                  float *extra = nullptr;
                  /* gradients w.r.t. the texels of bitmap `radiance` textures of area lights (their buffers are entries of `tex` like every bitmap's) */
                  bool light_texels = false; };

/* The specular part of RoughConductor::eval / RoughPlastic::eval (roughconductor.cpp:429-520, roughplastic.cpp:296-336, microfacet.h:185-207,341-365,
 * fresnel.h:93-116) in DOUBLE precision as a function of the parameters the adjoint differentiates: value_c(alpha_u, alpha_v, eta_c, k_c, slot1_c).
 * The oracle differentiates it NUMERICALLY (central differences, relative step 1e-6: ~1e-9 accurate in double) -- no hand-derived formula, so it is
 * an independent check of the product's analytic derivatives (har_bsdf.h). */
static void rough_spec_value_d(const BsdfRecord &b, V3 slot0, double au, double av, const double ec[3], const double kc[3], const double s1[3], V3 wi_f, V3 wo_f, double out[3]) {
    out[0] = out[1] = out[2] = 0.0;
    const double wi[3] = { wi_f.x, wi_f.y, wi_f.z }, wo[3] = { wo_f.x, wo_f.y, wo_f.z };
    if (!(wi[2] > 0.0 && wo[2] > 0.0)) return;
    double H[3] = { wi[0] + wo[0], wi[1] + wo[1], wi[2] + wo[2] }; const double hl = std::sqrt(H[0] * H[0] + H[1] * H[1] + H[2] * H[2]);
    for (double &h : H) h /= hl;
    const double wih = wi[0] * H[0] + wi[1] * H[1] + wi[2] * H[2], woh = wo[0] * H[0] + wo[1] * H[1] + wo[2] * H[2];
    const bool ggx = b.mtype() == MicrofacetType::GGX;
    au = std::max(au, 1e-4); av = std::max(av, 1e-4);
    const double au0 = std::max((double) b.p.alpha_u, 1e-4), av0 = b.p.type == 3 ? au0 : std::max((double) b.p.alpha_v, 1e-4);
    auto D = [&]() {
        const double c2 = H[2] * H[2], q = (H[0] / au) * (H[0] / au) + (H[1] / av) * (H[1] / av);
        const double r = ggx ? 1.0 / (M_PI * au * av * (q + c2) * (q + c2)) : std::exp(-q / c2) / (M_PI * au * av * c2 * c2);
        return r * H[2] > 1e-20 ? r : 0.0;
    };
    auto G1 = [&](const double v[3]) {
        const double xy = (au * v[0]) * (au * v[0]) + (av * v[1]) * (av * v[1]), t = xy / (v[2] * v[2]);
        double r;
        if (!ggx) {
            /* the rational fit is not exactly 1 at its cut-off (1.000057 at a = 1.6): the branch is the one the UNPERTURBED parameters take, as in
               the reference's AD of dr::select (microfacet.h:352-357) -- differencing across it would add jump / step to lanes that sit on it */
            const double t0 = ((au0 * v[0]) * (au0 * v[0]) + (av0 * v[1]) * (av0 * v[1])) / (v[2] * v[2]);
            const double a = 1.0 / std::sqrt(t); r = 1.0 / std::sqrt(t0) >= 1.6 ? 1.0 : (3.535 * a + 2.181 * a * a) / (1.0 + 2.276 * a + 2.577 * a * a);
        }
        else r = 2.0 / (1.0 + std::sqrt(1.0 + t));
        if (xy == 0.0) r = 1.0;
        if ((v[0] * H[0] + v[1] * H[1] + v[2] * H[2]) * v[2] <= 0.0) r = 0.0;
        return r;
    };
    auto Fc = [&](double c, double er, double ei) {
        const double c2 = c * c, s2 = 1.0 - c2, s4 = s2 * s2, t1 = er * er - ei * ei - s2, ab = std::sqrt(std::max(t1 * t1 + 4.0 * ei * ei * er * er, 0.0));
        const double a = std::sqrt(std::max(0.5 * (ab + t1), 0.0)), T1 = ab + c2, T2 = 2.0 * c * a, rs = (T1 - T2) / (T1 + T2), T3 = ab * c2 + s4, T4 = T2 * s2;
        return 0.5 * (rs + rs * (T3 - T4) / (T3 + T4));
    };
    if (b.p.type == 2) {
        if (!(wih > 0.0 && woh > 0.0)) return;
        const double d = D(); if (d == 0.0) return;
        const double V = d * G1(wi) * G1(wo) / (4.0 * wi[2]);
        const double s0[3] = { slot0.x, slot0.y, slot0.z };
        for (int c = 0; c < 3; ++c) out[c] = s0[c] * Fc(wih, ec[c], kc[c]) * V;
    } else if (b.p.type == 3) {
        av = au;
        const double d = D();
        const double F = (double) fresnel((float) wih, b.p.eta).r;                 /* not a function of the differentiated parameters */
        const double spec = F * d * G1(wi) * G1(wo) / (4.0 * wi[2]);
        for (int c = 0; c < 3; ++c) out[c] = s1[c] * spec;
    }
}
/* d value_c / d theta for the five parameter groups (alpha_u, alpha_v, eta, k, slot 1) at (wi, wo): out[g][c] */
static void rough_spec_grad_fd(const BsdfRecord &b, V3 slot0, V3 slot1, V3 wi, V3 wo, double out[5][3]) {
    for (int g = 0; g < 5; ++g) for (int c = 0; c < 3; ++c) out[g][c] = 0.0;
    if (b.p.type != 2 && b.p.type != 3) return;
    const double au = b.p.alpha_u, av = b.p.type == 3 ? b.p.alpha_u : b.p.alpha_v;
    double ec[3], kc[3], s1[3] = { slot1.x, slot1.y, slot1.z };
    for (int c = 0; c < 3; ++c) { ec[c] = b.p.eta_c[c]; kc[c] = b.p.k_c[c]; }
    auto diff = [&](int g, int c) {
        double p[2][3];
        for (int sgn = 0; sgn < 2; ++sgn) {
            double a1 = au, a2 = av, e[3] = { ec[0], ec[1], ec[2] }, k[3] = { kc[0], kc[1], kc[2] }, t[3] = { s1[0], s1[1], s1[2] };
            const double f = sgn ? 1.0 - 1e-6 : 1.0 + 1e-6;
            if (g == 0) { a1 *= f; if (b.p.type == 3) a2 = a1; } else if (g == 1) a2 *= f; else if (g == 2) e[c] *= f; else if (g == 3) k[c] *= f; else t[c] = t[c] * f + (sgn ? -1e-9 : 1e-9);
            rough_spec_value_d(b, slot0, a1, a2, e, k, t, wi, wo, p[sgn]);
        }
        const double base = g == 0 ? au : g == 1 ? av : g == 2 ? ec[c] : g == 3 ? kc[c] : s1[c];
        const double h = g == 4 ? 2.0 * (base * 1e-6 + 1e-9) : 2e-6 * base;
        if (g <= 1) { for (int cc = 0; cc < 3; ++cc) out[g][cc] = h != 0.0 ? (p[0][cc] - p[1][cc]) / h : 0.0; }
        else out[g][c] = h != 0.0 ? (p[0][c] - p[1][c]) / h : 0.0;
    };
    diff(0, 0);
    if (b.p.type == 2) { diff(1, 0); for (int c = 0; c < 3; ++c) { diff(2, c); diff(3, c); } }
    else for (int c = 0; c < 3; ++c) diff(4, c);
}
/* one derivative term: backward adds w * g to the parameter's gradient slot, forward adds w * g * tangent to the lane's differential radiance */
static inline void grad_commit(const GradSink *grad, float *slot, V3 g, float w = 1.f) {
    if (grad->fwd) { grad->fwd->x += g.x * w * slot[0]; grad->fwd->y += g.y * w * slot[1]; grad->fwd->z += g.z * w * slot[2]; }
    else { slot[0] += g.x * w; slot[1] += g.y * w; slot[2] += g.z * w; }
}

static V3 prb_sample(const Scene &sc, Pcg32 &rng, Ray ray, uint32_t max_depth, uint32_t rr_depth, bool primal,
                     V3 L_in, V3 dL, const GradSink *grad, bool &valid, OrcStats &st) {
    uint32_t depth = 0; V3 L = primal ? V3(0.f) : L_in; V3 beta(1.f); float eta = 1.f;
    bool active = true;
    PI pi; st.closest_rays++; scene_trace<false>(sc, ray, pi, 0);
    if (sc.hide_emitters) skip_area_emitters(sc, ray, pi, st);             // prb.py:112-118
    V3 prev_p(0.f); float bsdf_pdf_prev = 1.f; bool bsdf_delta_prev = true;
    Ray ray_prev = ray; PI pi_prev; SI si_prev;                 // prb.py:105-109: the previous vertex, for the attached si.wi
    uint32_t iter = 0;
    while (active && iter < max_depth) {                      // prb.py:121-123 (max_iterations)
        ++iter; st.vertices++;
        bool active_next = true;
        SI si = compute_si(sc, ray, pi);
        int emitter = si.valid() ? sc.meshes[si.mesh].emitter : sc.env;
        // prb.py:153-161
        V3 Le(0.f);
        {
            float em_pdf = 0.f;
            DS ds; ds.p = si.p; ds.n = si.sn;
            V3 rel = si.p - prev_p; ds.dist = norm(rel); ds.d = si.valid() ? div(rel, ds.dist) : -si.wi;
            if (emitter >= 0 && !bsdf_delta_prev) em_pdf = (sc.emitters[emitter].type == 1 ? InvFourPi : sc.emitters[emitter].type == 2 ? sc.envmap.pdf_direction(ds.d) : surface_emitter_pdf_direction(sc, (uint32_t) emitter, ds, si.uv)) * emitter_choice_pmf(sc, (uint32_t) emitter);
            float mis = mis_weight(bsdf_pdf_prev, em_pdf);
            if (emitter >= 0 && !(sc.hide_emitters && depth == 0 && !si.valid())) {          // prb.py:146-148: active_next masks emitter.eval
                const OrcEmitter &e = sc.emitters[emitter];
                V3 ev = e.type == 2 ? sc.envmap.eval(-si.wi) : (e.type == 1 || si.wi.z > 0.f) ? surface_emitter_radiance(sc, e, si.uv) : V3(0.f);
                Le = (beta * mis) * ev;
                if (!primal && grad && grad->emit && e.type != 2 && e.type != 7 && (e.type == 1 || si.wi.z > 0.f)) {   // d Le / d radiance = beta * mis (prb.py:160-161, attached emitter.eval)
                    grad_commit(grad, grad->emit + 3 * (size_t) emitter, (beta * mis) * dL);
                }
                if (!primal && grad && grad->tex && grad->light_texels && e.type == 7 && si.wi.z > 0.f) {            // ... with a bitmap radiance: the four texels under si.uv (area.cpp:83-90)
                    float *dst = grad->tex[e.radiance_texture]; const V3 g = (beta * mis) * dL;
                    tex_scatter(sc.textures[e.radiance_texture], si.uv, [&](uint32_t idx, float w) { grad_commit(grad, dst + 3 * (size_t) idx, g, w); });
                }
            }
        }
        active_next &= (depth + 1 < max_depth) && si.valid();  // prb.py:166
        BsdfCtx bsdf{}; bsdf.ok = false; bsdf.rec = nullptr; bsdf.wo_sign = 1.f;
        if (si.valid()) bsdf = bsdf_prepare(sc, sc.meshes[si.mesh].bsdf, si);
        bool active_em = active_next && bsdf.rec && bsdf.rec->smooth();          // prb.py:169
        float ex = rng.next_float32(), ey = rng.next_float32();
        DS ds; V3 em_weight(0.f); float em_unit = 0.f;
        if (active_em) { sample_emitter_direction(sc, si, ex, ey, ds, em_weight, st, nullptr, &em_unit); active_em &= ds.pdf != 0.f; }
        // prb.py:210-216
        V3 Lr_dir(0.f), dLr_dir_drho(0.f);
        float mis_em = 0.f; V3 wo_em(0.f); const V3 beta_cur = beta;
        if (active_em) {
            V3 wo = si.to_local(ds.d); wo_em = wo;
            BSDFEval ev = bsdf_eval_pdf(bsdf, wo);
            mis_em = ds.delta ? 1.f : mis_weight(ds.pdf, ev.pdf);                   // prb.py:211
            Lr_dir = ((beta * mis_em) * ev.value) * em_weight;
            dLr_dir_drho = ((beta * mis_em) * ev.d_slot0) * em_weight;           // d/d slot0 of the line above
            if (!primal && grad && grad->emit && sc.emitters[ds.emitter].type != 2 && sc.emitters[ds.emitter].type != 7) {   // em_weight = radiance * em_unit (prb.py:198-206, attached eval_emitter_direction)
                grad_commit(grad, grad->emit + 3 * (size_t) ds.emitter, (((beta * mis_em) * ev.value) * em_unit) * dL);
            }
            if (!primal && grad && grad->tex && grad->light_texels && sc.emitters[ds.emitter].type == 7) {                    // em_weight = radiance(ds.uv) * em_unit: the texels under the SAMPLED uv
                const OrcEmitter &e = sc.emitters[ds.emitter];
                float *dst = grad->tex[e.radiance_texture]; const V3 g = (((beta * mis_em) * ev.value) * em_unit) * dL;
                tex_scatter(sc.textures[e.radiance_texture], ds.uv, [&](uint32_t idx, float w) { grad_commit(grad, dst + 3 * (size_t) idx, g, w); });
            }
        }
        // detached BSDF sampling, prb.py:220-223 (masked lanes return zeros)
        float s1 = rng.next_float32();
        float s2x = rng.next_float32(), s2y = rng.next_float32();
        V3 bwo(0.f), bsdf_weight(0.f); float bs_pdf = 0.f, bs_eta = 0.f; bool bs_delta = false;
        if (active_next) { BSDFSample bs = bsdf_sample(bsdf, s1, s2x, s2y, bsdf_weight); bwo = bs.wo; bs_pdf = bs.pdf; bs_eta = bs.eta; bs_delta = bs.delta; }
        L = primal ? (L + Le) + Lr_dir : (L - Le) - Lr_dir;      // prb.py:227
        Ray ray_next = spawn_ray(si, si.to_world(bwo));
        eta *= bs_eta; beta = beta * bsdf_weight;
        prev_p = si.p; bsdf_pdf_prev = bs_pdf; bsdf_delta_prev = bs_delta;
        float beta_max = hmax(beta);
        active_next &= beta_max != 0.f;
        float rr_prob = std::fmin(beta_max * (eta * eta), .95f);
        bool rr_active = depth >= rr_depth;                      // prb.py:249 (depth NOT yet incremented)
        if (rr_active) beta = beta * rcp(rr_prob);
        bool rr_continue = rng.next_float32() < rr_prob;
        active_next &= !rr_active || rr_continue;
        PI pi_next;
        if (active_next) { st.closest_rays++; scene_trace<false>(sc, ray_next, pi_next, 0); }
        if (!primal && grad && si.valid() && bsdf.rec) {        // prb.py:263-313 specialised to slot-0 colour parameters
            V3 wo = si.to_local(ray_next.d);
            V3 g = dLr_dir_drho;
            if (active_next) {                                   // Lr_ind = L * relative_grad(bsdf.eval(si, wo, active_next))
                BSDFEval e2 = bsdf_eval_pdf(bsdf, wo);
                g = g + V3(e2.value.x != 0.f ? L.x * (e2.d_slot0.x / e2.value.x) : 0.f, e2.value.y != 0.f ? L.y * (e2.d_slot0.y / e2.value.y) : 0.f,
                           e2.value.z != 0.f ? L.z * (e2.d_slot0.z / e2.value.z) : 0.f);
            }
            g = g * dL;
            if (!bsdf.textured) grad_commit(grad, grad->refl + 3 * (size_t) bsdf.used, g);
            else {
                const TexLookup &tl = bsdf.tl;
                float *dst = grad->tex[bsdf.rec->p.texture];
                const float wts[4] = { tl.w[0] * tl.w[2], tl.w[1] * tl.w[2], tl.w[0] * tl.w[3], tl.w[1] * tl.w[3] };
                for (int k = 0; k < 4; ++k) grad_commit(grad, dst + 3 * (size_t) tl.idx[k], g, wts[k]);
            }
        }
        if (!primal && grad && grad->extra && si.valid() && bsdf.rec && bsdf.ok && (bsdf.rec->p.type == 2 || bsdf.rec->p.type == 3)) {
            /* the same two terms for alpha / eta / k / colour slot 1 (prb.py:288-313 with those parameters attached): dL * (d Lr_dir / d theta + L * (d f / d theta) / f) */
            double dem[5][3], dwo[5][3];
            float *dst = grad->extra + 15 * (size_t) bsdf.used;
            if (active_em) {
                rough_spec_grad_fd(*bsdf.rec, bsdf.slot0, bsdf.slot1, bsdf.wi, V3(wo_em.x, wo_em.y, wo_em.z * bsdf.wo_sign), dem);
                const double w[3] = { (double) beta_cur.x * mis_em * em_weight.x, (double) beta_cur.y * mis_em * em_weight.y, (double) beta_cur.z * mis_em * em_weight.z };
                const double dl[3] = { dL.x, dL.y, dL.z };
                for (int g = 0; g < 5; ++g) for (int c = 0; c < 3; ++c) dst[3 * g + c] += (float) (dl[c] * w[c] * dem[g][c]);
                if (getenv("ORC_DEBUG_EXTRA") && dl[0] != 0.0 && bsdf.rec->p.type == 3) fprintf(stderr, "EM depth %u bsdf %d wi %.9g %.9g %.9g wo %.9g %.9g %.9g dalpha %.9g w %.9g dl %.9g\n", depth, (int) bsdf.used, bsdf.wi.x, bsdf.wi.y, bsdf.wi.z,
                                                       wo_em.x, wo_em.y, wo_em.z * bsdf.wo_sign, dem[0][0], w[0], dl[0]);
            }
            if (active_next) {
                V3 wo = si.to_local(ray_next.d); const V3 wol(wo.x, wo.y, wo.z * bsdf.wo_sign);
                rough_spec_grad_fd(*bsdf.rec, bsdf.slot0, bsdf.slot1, bsdf.wi, wol, dwo);
                BSDFEval e2 = bsdf_eval_pdf(bsdf, wo);
                const double f[3] = { e2.value.x, e2.value.y, e2.value.z }, Lc[3] = { L.x, L.y, L.z }, dl[3] = { dL.x, dL.y, dL.z };
                for (int g = 0; g < 5; ++g) for (int c = 0; c < 3; ++c) if (f[c] != 0.0) dst[3 * g + c] += (float) (dl[c] * Lc[c] * dwo[g][c] / f[c]);
                if (getenv("ORC_DEBUG_EXTRA") && dl[0] != 0.0 && bsdf.rec->p.type == 3) fprintf(stderr, "REL depth %u bsdf %d wi %.9g %.9g %.9g wo %.9g %.9g %.9g dalpha %.9g f %.9g L %.9g dl %.9g\n", depth, (int) bsdf.used, bsdf.wi.x, bsdf.wi.y, bsdf.wi.z,
                                                       wol.x, wol.y, wol.z, dwo[0][0], f[0], Lc[0], dl[0]);
            }
        }
        if (!primal && grad && grad->shape && si.valid() && bsdf.rec) {
            /* prb.py:124-141 (attached si, si.wi re-attached to the motion of the PREVIOUS vertex), :176-216 (emitter sampling with the attached shading
               point), :261-297 (attached wo, J): d/d(vertex positions, instance transforms) of  Lr_dir + Lr_ind  for whatever BSDF the vertex carries.
               The BSDF value is a function of (si.wi, wo, si.uv); the oracle differentiates it numerically in the two directions (bsdf_dir_grad_fd) and
               chains the result with the duals of the attached geometry.  A vertex contributes when its own triangle is attached (a.diff) or when the
               previous vertex's is (ap.diff: only si.wi moves, so only BSDFs that depend on wi see it).  relative_grad(0) is taken as "no derivative". */
            const ShapeSink &sk = *grad->shape;
            AttachedSI a = attach_si(sc, ray, pi, si, sk.mask, 0, true, sk.inst_mask);
            AttachedSI ap;
            if (depth >= 1) ap = attach_si(sc, ray_prev, pi_prev, si_prev, sk.mask, kPrevSlot, false, sk.inst_mask);     /* pi_prev.compute_surface_interaction(ray_prev, Minimal) */
            if (a.diff || ap.diff) {
                const AttachedFrame fr = attach_frame(a, si);
                Dn3 wi_l;
                if (depth == 0) wi_l = frame_to_local(fr, dn3(-ray.d));            /* compute_surface_interaction: wi = to_local(-ray.d) in the attached frame */
                else {                                                             /* prb.py:137-140: si_detached.to_local(normalize(si_prev.p - si_detached.p)) */
                    AttachedFrame fd; fd.s = dn3(si.ss); fd.t = dn3(si.st); fd.n = dn3(si.sn);
                    wi_l = frame_to_local(fd, dnormalize(ap.p - dn3(si.p)));
                }
                wi_l = replace_grad3(si.wi.x, si.wi.y, si.wi.z, wi_l);
                Dn3 rho = bsdf.textured ? tex_eval_dual(sc.textures[bsdf.rec->p.texture], bsdf.tl, a.uv) : dn3(bsdf.slot0);
                double g[kShapeSlots]; for (int k = 0; k < kShapeSlots; ++k) g[k] = 0.0;
                const double dl[3] = { dL.x, dL.y, dL.z };
                /* bsdf.eval(ctx, si, si.to_local(wo_world)) with everything attached; TwoSidedBRDF mirrors both directions for the back side (twosided.cpp:124-127) */
                auto value_cos = [&](const Dn3 &wo_world, V3 wo_value, Dn out[3]) {
                    Dn3 wo_l = replace_grad3(wo_value.x, wo_value.y, wo_value.z, frame_to_local(fr, wo_world));
                    const double sg = bsdf.wo_sign;
                    const V3 wo_side(wo_value.x, wo_value.y, wo_value.z * bsdf.wo_sign);
                    BSDFEval ev; if (bsdf.ok) ev = plugin_eval_pdf(*bsdf.rec, bsdf.slot0, bsdf.slot1, bsdf.wi, wo_side);
                    DirGrad dg; for (int c = 0; c < 3; ++c) for (int j = 0; j < 6; ++j) dg.d[c][j] = 0.0;
                    if (bsdf.ok) bsdf_dir_grad_fd(*bsdf.rec, bsdf.slot0, bsdf.slot1, bsdf.wi, wo_side, dg);
                    const Dn *in[6] = { &wi_l.x, &wi_l.y, &wi_l.z, &wo_l.x, &wo_l.y, &wo_l.z };
                    const Dn *r[3] = { &rho.x, &rho.y, &rho.z };
                    const float val[3] = { ev.value.x, ev.value.y, ev.value.z }, ds0[3] = { ev.d_slot0.x, ev.d_slot0.y, ev.d_slot0.z };
                    for (int c = 0; c < 3; ++c) {
                        Dn f((double) val[c]);
                        for (int k = 0; k < kShapeSlots; ++k) {
                            double acc = (double) ds0[c] * r[c]->d[k];
                            for (int j = 0; j < 6; ++j) acc += dg.d[c][j] * in[j]->d[k] * ((j == 2 || j == 5) ? sg : 1.0);
                            f.d[k] = acc;
                        }
                        out[c] = f;
                    }
                };
                if (active_em) {
                    const uint32_t et = sc.emitters[ds.emitter].type;
                    const bool is_surface = et == 0 || et == 3 || et == 7;         // EmitterFlags::Surface (area.cpp:42)
                    const bool is_infinite = et == 1 || et == 2 || et == 6;        // EmitterFlags::Infinite (constant.cpp, envmap.cpp, directional.cpp)
                    Dn3 dsd = dn3(ds.d); Dn J(1.0);
                    Dn E(1.0);                                                     // the part of eval_emitter_direction(si, ds) that follows si.p (prb.py:203-206), up to a constant factor
                    if (is_surface) {                                              // prb.py:189-201: ds.d = normalize(ds.p - si.p), J(si.p, detach(ds.p), detach(ds.n))
                        Dn3 dir; dir_and_jacobian(a.p, dn3(ds.p), dn3(ds.n), dir, J);
                        dsd = replace_grad3(ds.d.x, ds.d.y, ds.d.z, dir);
                    } else if (!is_infinite) {
                        /* point / spot: prb.py:191-192 -- only ds.d = normalize(ds.p - si.p) is re-attached (ds.p, ds.dist, ds.n take the zero gradients of
                         * ds_diff); J = 1 (not a surface).  PointLight::eval_direction (point.cpp:155-165) divides by squared_norm(ds.p - it.p) with the attached
                         * it.p; SpotLight::eval_direction (spot.cpp:252-274) uses rcp(ds.dist) -- DETACHED -- and the falloff of the attached ds.d */
                        const Dn3 dvec = dn3(ds.p) - a.p;
                        dsd = replace_grad3(ds.d.x, ds.d.y, ds.d.z, dnormalize(dvec));
                        const OrcEmitter &e = sc.emitters[ds.emitter];
                        if (et == 4) E = Dn(1.0) / ddot(dvec, dvec);
                        else {
                            const float *T = e.to_local;                            // local_d = to_world.inverse() * -ds.d, then falloff_curve's normalize (spot.cpp:143-151)
                            const Dn3 md = dsd * -1.0;
                            const Dn3 ld = dnormalize(Dn3(md.x * (double) T[0] + md.y * (double) T[3] + md.z * (double) T[6], md.x * (double) T[1] + md.y * (double) T[4] + md.z * (double) T[7],
                                                          md.x * (double) T[2] + md.y * (double) T[5] + md.z * (double) T[8]));
                            const double deg = 0.017453292519943295, cutoff = (double) e.normal[0] * deg, beam = (double) e.normal[1] * deg;
                            if (ld.z.v < std::cos(beam)) {                          // inside the beam the curve is constant; in the transition it is (cutoff - acos(cos_theta)) / (cutoff - beam)
                                Dn ac; ac.v = std::acos(ld.z.v); const double k = -1.0 / std::sqrt(std::max(1.0 - ld.z.v * ld.z.v, 1e-300));
                                for (int i = 0; i < kShapeSlots; ++i) ac.d[i] = ld.z.d[i] * k;
                                E = (Dn(cutoff) - ac) * (1.0 / (cutoff - beam));
                            }
                        }
                    }
                    Dn f[3]; value_cos(dsd, wo_em, f);
                    const double w[3] = { (double) beta_cur.x * mis_em * em_weight.x, (double) beta_cur.y * mis_em * em_weight.y, (double) beta_cur.z * mis_em * em_weight.z };
                    for (int c = 0; c < 3; ++c) {
                        if (w[c] == 0.0 || J.v == 0.0 || E.v == 0.0) continue;
                        for (int k = 0; k < kShapeSlots; ++k) g[k] += dl[c] * w[c] * (f[c].d[k] + f[c].v * (J.d[k] / J.v + E.d[k] / E.v));      // em_weight = replace_grad(em_weight, em_val_diff / pdf) * relative_grad(J)
                    }
                }
                if (active_next) {                                                 // prb.py:261-297
                    /* si_next is computed OUTSIDE dr.resume_grad() (prb.py:263-266): its position and normal are detached, only the current
                       point moves in wo and in the Jacobian */
                    SI si_next = compute_si(sc, ray_next, pi_next);
                    Dn3 wo_world = dn3(ray_next.d); Dn J(1.0);
                    if (pi_next.valid()) {
                        Dn3 dir; dir_and_jacobian(a.p, dn3(si_next.p), dn3(si_next.n), dir, J);
                        wo_world = replace_grad3(ray_next.d.x, ray_next.d.y, ray_next.d.z, dir);
                    }
                    Dn f[3]; value_cos(wo_world, si.to_local(ray_next.d), f);
                    const double Lc[3] = { L.x, L.y, L.z };
                    for (int c = 0; c < 3; ++c) {
                        if (Lc[c] == 0.0) continue;
                        for (int k = 0; k < kShapeSlots; ++k)
                            g[k] += dl[c] * Lc[c] * ((f[c].v != 0.0 ? f[c].d[k] / f[c].v : 0.0) + (J.v != 0.0 ? J.d[k] / J.v : 0.0));
                    }
                }
                shape_scatter(sk, a, 0, g);
                shape_scatter(sk, ap, kPrevSlot, g);
            }
        }
        if (si.valid()) { ray_prev = ray; pi_prev = pi; si_prev = si; }          // prb.py:327-330 (only lanes that met a surface continue)
        if (si.valid()) depth += 1;                               // prb.py:326
        active = active_next; pi = pi_next; ray = ray_next;
    }
    valid = depth != 0;
    return L;
}

// ---------------------------------------------------------------------------
//  Drivers (SamplingIntegrator::render, integrator.cpp:276-388;
//  ADIntegrator.render / RBIntegrator.render_backward, common.py:46-110,625-783)
// ---------------------------------------------------------------------------

/* worker threads when the caller asks for "all": the CPUs of the affinity mask, capped by the container's bandwidth quota (cgroup v2 cpu.max or the v1
 * cfs files) -- the GPU boxes show 256 logical CPUs behind a quota of 16, and 256 threads then run at an eighth of the rate of 16 (tools/cpu_scaling.py) */
static int default_threads() {
    static const int cached = [] {
        int n = (int) std::thread::hardware_concurrency();
        cpu_set_t mask;
        if (sched_getaffinity(0, sizeof(mask), &mask) == 0 && CPU_COUNT(&mask) > 0) n = std::min(n, CPU_COUNT(&mask));
        double quota = 0.0, period = 0.0;
        std::ifstream v2("/sys/fs/cgroup/cpu.max");
        std::string q;
        if (v2 && (v2 >> q >> period)) { if (q != "max") quota = std::strtod(q.c_str(), nullptr); }
        else {
            std::ifstream fq("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), fp("/sys/fs/cgroup/cpu/cpu.cfs_period_us");
            if (!(fq >> quota) || !(fp >> period)) quota = period = 0.0;
        }
        if (quota > 0.0 && period > 0.0) n = std::min(n, (int) std::ceil(quota / period));
        return std::max(n, 1);
    }();
    return cached;
}
template <typename Fn>
static void parallel_lanes(uint64_t begin, uint64_t end, int threads, Fn fn) {
    if (threads <= 0) threads = default_threads();
    if (threads < 1) threads = 1;
    uint64_t n = end - begin;
    if (n < 4096 || threads == 1) { fn(0, begin, end); return; }
    std::vector<std::thread> pool;
    std::atomic<uint64_t> next(begin);
    const uint64_t chunk = std::min<uint64_t>(16384, std::max<uint64_t>(512, n / ((uint64_t) threads * 8)));   /* >= 8 chunks per worker: balanced tails on 256 threads too */
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([&, t]() {
            for (;;) { uint64_t b = next.fetch_add(chunk); if (b >= end) break; fn(t, b, std::min(end, b + chunk)); }
        });
    for (auto &th : pool) th.join();
}

static int resolve_threads(int threads) {
    if (threads <= 0) threads = default_threads();
    return threads < 1 ? 1 : threads;
}

/* one counter block per worker thread, each on its own cache line: the workers bump `vertices` at every path vertex, and neighbouring 32-byte blocks
 * made them fight over lines (the 256-thread CPU baseline of bench.py ran at a thirteenth of the single-thread rate per thread) */
struct alignas(64) ThreadStats : OrcStats { ThreadStats() : OrcStats{} {} };
static void merge_stats(OrcStats *dst, const std::vector<ThreadStats> &src) {
    if (!dst) return;
    for (const auto &s : src) { dst->paths += s.paths; dst->vertices += s.vertices; dst->closest_rays += s.closest_rays; dst->shadow_rays += s.shadow_rays; }
}

static int render_forward(Scene &sc, const OrcSensor &s, uint32_t seed, uint32_t spp, int32_t max_depth, int32_t rr_depth,
                          uint64_t lb, uint64_t le, float *film, OrcStats *stats, int threads, bool prb) {
    uint64_t total = sample_grid_pixels(s) * spp;
    if (lb == 0 && le == 0) le = total;
    if (le > total || lb > le || total > 0xffffffffull) return -1;
    RFilter rf = make_rfilter(s.rfilter, s.rfilter_stddev, s.rfilter_param1);
    threads = resolve_threads(threads);
    size_t fsz = (size_t) s.crop_width * s.crop_height * 4;
    std::vector<std::vector<float>> films(threads);
    std::vector<ThreadStats> sts(threads);
    uint32_t md = (uint32_t) max_depth, rd = (uint32_t) rr_depth;
    parallel_lanes(lb, le, threads, [&](int t, uint64_t b, uint64_t e) {
        if (films[t].empty()) films[t].assign(fsz, 0.f);
        for (uint64_t i = b; i < e; ++i) {
            Lane L = make_lane(s, seed, spp, i);
            bool valid; V3 rgb;
            if (prb) rgb = prb_sample(sc, L.rng, L.ray, md, rd, true, V3(0.f), V3(0.f), nullptr, valid, sts[t]);
            else     rgb = path_sample(sc, L.rng, L.ray, md, rd, valid, sts[t]);
            sts[t].paths++;
            if (sc.alpha_only) rgb = V3(valid ? 1.f : 0.f);             // the alpha channel of `rgba` films: integrator.cpp:497-504, common.py:150-152
            float v[4] = { rgb.x, rgb.y, rgb.z, 1.f };
            film_put(s, rf, rf.type == 0 ? L.ipos_x : L.pos_x, rf.type == 0 ? L.ipos_y : L.pos_y, v, films[t].data());
        }
    });
    for (auto &f : films) if (!f.empty()) for (size_t i = 0; i < fsz; ++i) film[i] += f[i];
    merge_stats(stats, sts);
    return 0;
}

/* SamplingIntegrator::render, JIT branch with several passes (integrator.cpp:173-183,276-356): n_passes wavefronts of W*H*spp_per_pass
 * lanes; lane i keeps its pixel and its sampler (seeded once, `sampler->advance()` does not reseed), the block accumulates all passes. */
static int render_forward_passes(Scene &sc, const OrcSensor &s, uint32_t seed, uint32_t spp, uint32_t spp_per_pass, int32_t max_depth, int32_t rr_depth,
                                 uint64_t lb, uint64_t le, float *film, OrcStats *stats, int threads) {
    if (spp_per_pass == 0 || spp % spp_per_pass != 0) return -2;               // integrator.cpp:177-179
    uint64_t total = sample_grid_pixels(s) * spp_per_pass;
    if (lb == 0 && le == 0) le = total;
    if (le > total || lb > le || total > 0xffffffffull) return -1;
    const uint32_t n_passes = spp / spp_per_pass;
    RFilter rf = make_rfilter(s.rfilter, s.rfilter_stddev, s.rfilter_param1);
    threads = resolve_threads(threads);
    size_t fsz = (size_t) s.crop_width * s.crop_height * 4;
    std::vector<std::vector<float>> films(threads);
    std::vector<ThreadStats> sts(threads);
    uint32_t md = max_depth < 0 ? 0xffffffffu : (uint32_t) max_depth, rd = (uint32_t) rr_depth;
    parallel_lanes(lb, le, threads, [&](int t, uint64_t b, uint64_t e) {
        if (films[t].empty()) films[t].assign(fsz, 0.f);
        for (uint64_t i = b; i < e; ++i) {
            Lane L = make_lane(s, seed, spp_per_pass, i);                      // pass 0: seeded, jitter drawn
            for (uint32_t pass = 0; pass < n_passes; ++pass) {
                if (pass) {                                                    // render_sample again with the SAME `pos`, stream continues
                    float jx = L.rng.next_float32(), jy = L.rng.next_float32();
                    L.pos_x = L.ipos_x + jx; L.pos_y = L.ipos_y + jy;
                    float sx = 1.f / (float) s.crop_width, sy = 1.f / (float) s.crop_height;
                    L.ray = sensor_sample_ray(s, fmadd(L.pos_x, sx, -(float) s.crop_offset_x * sx), fmadd(L.pos_y, sy, -(float) s.crop_offset_y * sy));
                }
                bool valid; V3 rgb = path_sample(sc, L.rng, L.ray, md, rd, valid, sts[t]);
                sts[t].paths++;
                if (sc.alpha_only) rgb = V3(valid ? 1.f : 0.f);             // the alpha channel of `rgba` films: integrator.cpp:497-504, common.py:150-152
            float v[4] = { rgb.x, rgb.y, rgb.z, 1.f };
                film_put(s, rf, rf.type == 0 ? L.ipos_x : L.pos_x, rf.type == 0 ? L.ipos_y : L.pos_y, v, films[t].data());
            }
        }
    });
    for (auto &f : films) if (!f.empty()) for (size_t i = 0; i < fsz; ++i) film[i] += f[i];
    merge_stats(stats, sts);
    return 0;
}

// ---------------------------------------------------------------------------
//  Scalar-variant driver (BASELINE config 1, `scalar_rgb`): SamplingIntegrator::render, CPU branch
//  (src/render/integrator.cpp:190-274) -> Spiral::next_block (src/render/spiral.cpp:27-73) ->
//  render_block (integrator.cpp:398-446: Morton pixel order, per-pixel reseed) -> render_sample ->
//  ImageBlock::put, non-coalesced branch with the discretised filter (src/render/imageblock.cpp:283-375,
//  include/mitsuba/core/rfilter.h:70-79, src/core/rfilter.cpp:11-26) -> HDRFilm::put_block
// ---------------------------------------------------------------------------

static inline void morton_decode2(uint32_t m, uint32_t &x, uint32_t &y) {     // dr::morton_decode<Point2u>: even bits -> x, odd bits -> y
    auto compact = [](uint32_t v) { v &= 0x55555555u; v = (v ^ (v >> 1)) & 0x33333333u; v = (v ^ (v >> 2)) & 0x0f0f0f0fu;
                                    v = (v ^ (v >> 4)) & 0x00ff00ffu; v = (v ^ (v >> 8)) & 0x0000ffffu; return v; };
    x = compact(m); y = compact(m >> 1);
}

struct SpiralBlock { int32_t off_x, off_y; uint32_t size_x, size_y, id; };
static std::vector<SpiralBlock> spiral_blocks(uint32_t size_x, uint32_t size_y, uint32_t off_x, uint32_t off_y, uint32_t block_size) {
    std::vector<SpiralBlock> out;
    int32_t bx = (int32_t) ((size_x + block_size - 1) / block_size), by = (int32_t) ((size_y + block_size - 1) / block_size);
    uint32_t block_count = (uint32_t) (bx * by), counter = 0;
    int direction = 0 /* Right, Down, Left, Up */; int32_t px = bx / 2, py = by / 2; uint32_t steps_left = 1, spiral_size = 1;
    while (counter != block_count) {
        SpiralBlock b; b.id = counter;                                  // single pass: block_id = m_block_counter
        uint32_t ox = (uint32_t) px * block_size, oy = (uint32_t) py * block_size;
        b.size_x = std::min(block_size, size_x - ox); b.size_y = std::min(block_size, size_y - oy);
        b.off_x = (int32_t) (ox + off_x); b.off_y = (int32_t) (oy + off_y);
        out.push_back(b);
        ++counter;
        if (counter != block_count) {
            do {
                switch (direction) { case 0: ++px; break; case 1: ++py; break; case 2: --px; break; default: --py; break; }
                if (--steps_left == 0) {
                    direction = (direction + 1) % 4;
                    if (direction == 2 || direction == 0) ++spiral_size;
                    steps_left = spiral_size;
                }
            } while (px < 0 || py < 0 || px >= bx || py >= by);
        }
    }
    return out;
}

static int render_scalar(Scene &sc, const OrcSensor &s, uint32_t seed, uint32_t spp, int32_t max_depth, int32_t rr_depth,
                         uint32_t n_threads, float *film, OrcStats *stats, uint32_t *block_size_out) {
    const uint32_t W = s.crop_width, H = s.crop_height;
    RFilter rf = make_rfilter(s.rfilter, s.rfilter_stddev, s.rfilter_param1);
    // discretised filter (rfilter.cpp:11-26): MI_FILTER_RESOLUTION = 31
    constexpr int RES = 31;
    float values[RES + 1];
    for (int i = 0; i < RES; ++i) values[i] = s.rfilter == 0 ? 1.f : rfilter_eval(rf, (rf.radius * (float) i) / (float) RES);
    if (s.rfilter == 0) for (int i = 0; i < RES; ++i) values[i] = (rf.radius * (float) i) / (float) RES <= .5f ? 1.f : 0.f;
    values[RES] = 0.f;
    const float scale_factor = (float) RES / rf.radius;
    const int border = (int) std::ceil(rf.radius - .5f - 2.f * RayEpsilon);
    auto eval_discretized = [&](float x) { uint32_t index = std::min<uint32_t>((uint32_t) std::fabs(x * scale_factor), (uint32_t) RES); return values[index]; };
    // block size (integrator.cpp:203-214)
    uint32_t block_size = 32;
    /* Film::sample_border: spiral over the enlarged film, every block shifted back by the border (integrator.cpp:162-165, 215, 248-249) */
    const uint32_t sb = sample_border_size(s), Wg = W + 2 * sb, Hg = H + 2 * sb;
    while (true) { if (block_size == 1 || ((Wg + block_size - 1) / block_size) * ((Hg + block_size - 1) / block_size) >= std::max(n_threads, 1u)) break; block_size /= 2; }
    if (block_size_out) *block_size_out = block_size;
    std::vector<SpiralBlock> blocks = spiral_blocks(Wg, Hg, s.crop_offset_x, s.crop_offset_y, block_size);
    for (SpiralBlock &b : blocks) { b.off_x -= (int32_t) sb; b.off_y -= (int32_t) sb; }
    uint32_t md = max_depth < 0 ? 0xffffffffu : (uint32_t) max_depth;
    seed *= Wg * Hg;                                                    // integrator.cpp:231 (dr::prod(film_size) of the crop window)
    OrcStats st{};
    const bool box = s.rfilter == 0;
    for (const SpiralBlock &b : blocks) {
        // ImageBlock(size, border = true): (size + 2 * border)^2 x 4, cleared per block (integrator.cpp:419)
        const uint32_t bw = b.size_x + 2 * (uint32_t) border, bh = b.size_y + 2 * (uint32_t) border;
        std::vector<float> blk((size_t) bw * bh * 4, 0.f);
        uint32_t bseed = seed + b.id * block_size * block_size;       // integrator.cpp:412
        for (uint32_t i = 0; i < block_size * block_size; ++i) {
            uint32_t px, py; morton_decode2(i, px, py);
            if (px >= b.size_x || py >= b.size_y) continue;
            Pcg32 rng = sampler_seed(bseed + i, 0);                   // sampler->seed(seed + i): wavefront of size 1 (sampler.cpp:129-148)
            float pos_fx = (float) ((int32_t) px + b.off_x), pos_fy = (float) ((int32_t) py + b.off_y);
            for (uint32_t j = 0; j < spp; ++j) {                      // render_sample, integrator.cpp:448-520
                float jx = rng.next_float32(), jy = rng.next_float32();
                float sx = pos_fx + jx, sy = pos_fy + jy;
                float isx = 1.f / (float) W, isy = 1.f / (float) H;
                Ray ray = sensor_sample_ray(s, fmadd(sx, isx, -(float) s.crop_offset_x * isx), fmadd(sy, isy, -(float) s.crop_offset_y * isy));
                bool valid; V3 rgb = path_sample(sc, rng, ray, md, (uint32_t) rr_depth, valid, st, /* scalar */ true);
                st.paths++;
                const float v[4] = { rgb.x, rgb.y, rgb.z, 1.f };
                float ppx = box ? pos_fx : sx, ppy = box ? pos_fy : sy;
                // ImageBlock::put, scalar branch (imageblock.cpp:283-375)
                if (box) {                                            // no filter: nearest pixel (imageblock.cpp:225-243)
                    int32_t x = (int32_t) std::floor(ppx) - b.off_x + border, y = (int32_t) std::floor(ppy) - b.off_y + border;
                    if (x >= 0 && y >= 0 && (uint32_t) x < bw && (uint32_t) y < bh) { float *p = blk.data() + 4 * ((size_t) y * bw + x); for (int k = 0; k < 4; ++k) p[k] += v[k]; }
                    continue;
                }
                float pfx = ppx + ((float) border - (float) b.off_x - .5f), pfy = ppy + ((float) border - (float) b.off_y - .5f);
                int32_t x0 = std::max((int32_t) std::ceil(pfx - rf.radius), 0), y0 = std::max((int32_t) std::ceil(pfy - rf.radius), 0);
                int32_t x1 = std::min((int32_t) std::floor(pfx + rf.radius), (int32_t) bw - 1), y1 = std::min((int32_t) std::floor(pfy + rf.radius), (int32_t) bh - 1);
                if (x0 > x1 || y0 > y1) continue;
                float wx[16], wy[16];
                for (int32_t x = x0; x <= x1; ++x) wx[x - x0] = eval_discretized((float) x0 - pfx + (float) (x - x0));
                for (int32_t y = y0; y <= y1; ++y) wy[y - y0] = eval_discretized((float) y0 - pfy + (float) (y - y0));
                for (int32_t y = y0; y <= y1; ++y)
                    for (int32_t x = x0; x <= x1; ++x) {
                        float w = wx[x - x0] * wy[y - y0];
                        float *p = blk.data() + 4 * ((size_t) y * bw + x);
                        for (int k = 0; k < 4; ++k) p[k] = fmadd(v[k], w, p[k]);
                    }
            }
        }
        // HDRFilm::put_block -> ImageBlock::put_block (imageblock.cpp:152-185): accumulate the overlap with the film
        for (uint32_t y = 0; y < bh; ++y) {
            int32_t fy = (int32_t) y + b.off_y - border - (int32_t) s.crop_offset_y;
            if (fy < 0 || fy >= (int32_t) H) continue;
            for (uint32_t x = 0; x < bw; ++x) {
                int32_t fx = (int32_t) x + b.off_x - border - (int32_t) s.crop_offset_x;
                if (fx < 0 || fx >= (int32_t) W) continue;
                const float *src = blk.data() + 4 * ((size_t) y * bw + x); float *dst = film + 4 * ((size_t) fy * W + fx);
                for (int k = 0; k < 4; ++k) dst[k] += src[k];
            }
        }
    }
    if (stats) *stats = st;
    return 0;
}

/* the BVH over the instances' world-space boxes (Instance::bbox, instance.cpp:93-103); rebuilt when a nested mesh or a transform changes */
static void build_instance_bvh(Scene *sc) {
    sc->inst_nodes.clear(); sc->inst_order.clear();
    if (!sc->instances.empty()) {
        std::vector<BuildPrim> prims;
        for (uint32_t i = 0; i < sc->instances.size(); ++i) {
            const Bvh &b = sc->group_bvh[sc->instances[i].group];
            BuildPrim p; p.id = i;
            for (int a = 0; a < 3; ++a) { p.lo[a] = Infinity; p.hi[a] = -Infinity; }
            if (!b.empty())
                for (int c = 0; c < 8; ++c) {
                    V3 q = xf_point(sc->instances[i].to_world, V3(c & 1 ? b.hi[0] : b.lo[0], c & 2 ? b.hi[1] : b.lo[1], c & 4 ? b.hi[2] : b.lo[2]));
                    for (int a = 0; a < 3; ++a) { p.lo[a] = std::min(p.lo[a], q[a]); p.hi[a] = std::max(p.hi[a], q[a]); }
                }
            pad_box(p.lo, p.hi);
            for (int a = 0; a < 3; ++a) p.c[a] = 0.5f * (p.lo[a] + p.hi[a]);
            prims.push_back(p);
        }
        sc->inst_nodes.emplace_back();
        build_rec(sc->inst_nodes, prims, 0, (uint32_t) prims.size(), 0, 2);
        for (auto &p : prims) sc->inst_order.push_back(p.id);
    }
}

/* bounding sphere of the scene for the environment emitters; recomputed when vertex positions change */
static void scene_update_bounds(Scene &sc) {
    bool needs_bounds = sc.env >= 0;
    for (const OrcEmitter &e : sc.emitters) needs_bounds |= e.type == 6;      // DirectionalEmitter::set_scene (directional.cpp:99-109): the same sphere
    if (needs_bounds) {                                // ConstantBackgroundEmitter::set_scene (constant.cpp:72-87)
        float lo[3] = { Infinity, Infinity, Infinity }, hi[3] = { -Infinity, -Infinity, -Infinity };
        for (uint32_t m = 0; m < sc.top_count; ++m)
            for (uint32_t v = 0; v < sc.meshes[m].nv; ++v) for (int a = 0; a < 3; ++a) { float q = sc.meshes[m].V[8 * (size_t) v + a]; lo[a] = std::min(lo[a], q); hi[a] = std::max(hi[a], q); }
        for (uint32_t i = 0; i < sc.instances.size(); ++i) {
            const OrcShapeGroup &g = sc.groups[sc.instances[i].group];
            float glo[3] = { Infinity, Infinity, Infinity }, ghi[3] = { -Infinity, -Infinity, -Infinity };
            for (uint32_t m = g.first_mesh; m < g.first_mesh + g.mesh_count; ++m)
                for (uint32_t v = 0; v < sc.meshes[m].nv; ++v) for (int a = 0; a < 3; ++a) { float q = sc.meshes[m].V[8 * (size_t) v + a]; glo[a] = std::min(glo[a], q); ghi[a] = std::max(ghi[a], q); }
            if (!(glo[0] <= ghi[0])) continue;
            for (int c = 0; c < 8; ++c) { V3 q = xf_point(sc.instances[i].to_world, V3(c & 1 ? ghi[0] : glo[0], c & 2 ? ghi[1] : glo[1], c & 4 ? ghi[2] : glo[2])); for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], q[a]); hi[a] = std::max(hi[a], q[a]); } }
        }
        if (lo[0] <= hi[0]) {
            V3 c((hi[0] + lo[0]) * .5f, (hi[1] + lo[1]) * .5f, (hi[2] + lo[2]) * .5f);
            sc.env_center[0] = c.x; sc.env_center[1] = c.y; sc.env_center[2] = c.z;
            sc.env_radius = std::max(RayEpsilon, norm(c - V3(hi[0], hi[1], hi[2])) * (1.f + RayEpsilon));
        } else sc.env_radius = RayEpsilon;
        for (int a = 0; a < 3; ++a) sc.envmap.center[a] = sc.env_center[a];          // EnvironmentMapEmitter::set_scene (envmap.cpp:214-226), same rule
        sc.envmap.radius = sc.env_radius;
    }
}

} // namespace

// ===========================================================================
//  C ABI
// ===========================================================================

extern "C" {

/* Texture::mean of slot 0: SRGBReflectanceSpectrum::mean (srgb.cpp:117-122) / BitmapTexture::mean (bitmap.cpp:730) */
static float slot0_mean(const Scene &sc, uint32_t i) {
    const OrcBSDF &p = sc.bsdfs[i].p;
    if (p.texture < 0) return (p.reflectance[0] + p.reflectance[1] + p.reflectance[2]) / 3.f;
    const Texture &t = sc.textures[p.texture]; double acc = 0; for (float v : t.data) acc += v;
    return (float) (acc / (double) t.data.size());
}

void *orc_scene_create(const OrcSceneDesc *d) {
    Scene *sc = new Scene();
    sc->top_count = d->top_mesh_count;
    for (uint32_t i = 0; i < d->mesh_count; ++i) {
        const OrcMesh &m = d->meshes[i]; Mesh o;
        o.V.assign(m.vertex_ptr, m.vertex_ptr + 8 * (size_t) m.vertex_count);
        o.F.assign(m.index_ptr, m.index_ptr + 4 * (size_t) m.face_count);
        o.nv = m.vertex_count; o.nf = m.face_count; o.bsdf = m.bsdf; o.emitter = m.emitter; o.flags = m.flags;
        sc->meshes.push_back(std::move(o));
    }
    sc->groups.assign(d->groups, d->groups + d->group_count);
    sc->instances.assign(d->instances, d->instances + d->instance_count);
    for (uint32_t i = 0; i < d->bsdf_count; ++i) { BsdfRecord r; r.p = d->bsdfs[i]; if (!(r.p.flags & 1u)) r.p.back = -1; sc->bsdfs.push_back(r); }
    for (uint32_t i = 0; i < d->texture_count; ++i) {
        Texture t; t.w = d->textures[i].width; t.h = d->textures[i].height; t.mode = d->textures[i].mode;
        t.data.assign(d->textures[i].data, d->textures[i].data + 3 * (size_t) t.w * t.h);
        const float *m = d->textures[i].to_uv; bool unset = true;
        for (int k = 0; k < 6; ++k) unset = unset && m[k] == 0.f;
        if (!unset) { for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c) { t.xf[r][c] = m[3 * r + c]; t.moved = t.moved || m[3 * r + c] != (r == c ? 1.f : 0.f); } }
        sc->textures.push_back(std::move(t));
    }
    sc->emitters.assign(d->emitters, d->emitters + d->emitter_count);
    sc->area_pmf.resize(sc->emitters.size()); sc->texel_tables.resize(sc->emitters.size());
    if (!rebuild_emitter_choice(*sc)) { delete sc; return nullptr; }
    for (uint32_t i = 0; i < sc->emitters.size(); ++i) {
        const OrcEmitter e = sc->emitters[i];
        if (e.type == 1 || e.type == 2) sc->env = (int) i;
        if (e.type == 3) {              // Mesh::build_pmf: face areas, sequential float prefix sum (dr::prefix_sum's summation order is not in the tree)
            if (e.mesh >= sc->top_count || sc->meshes[e.mesh].nf == 0) { delete sc; return nullptr; }
            const Mesh &m = sc->meshes[e.mesh];
            Scene::AreaPmf &d = sc->area_pmf[i];
            float acc = 0.f;
            for (uint32_t f = 0; f < m.nf; ++f) {
                const uint32_t *fi = m.F.data() + 4 * (size_t) f;
                auto P = [&](uint32_t v) { const float *q = m.V.data() + 8 * (size_t) v; return V3(q[0], q[1], q[2]); };
                float a = .5f * norm(cross(P(fi[1]) - P(fi[0]), P(fi[2]) - P(fi[0])));
                d.pmf.push_back(a); acc += a; d.cdf.push_back(acc);
            }
            d.sum = acc; d.normalization = rcp(acc);
            sc->emitters[i].inv_area = d.normalization;       // Mesh::pdf_position = m_area_pmf.normalization()
        }
        if (e.type == 7) {              // AreaLight with a bitmap radiance on a rectangle: the texel distribution is built up front (the reference builds it on first use, bitmap.cpp:963-971)
            std::string why;
            if (e.mesh >= sc->top_count || e.radiance_texture >= sc->textures.size() ||
                !texel_table_build(sc->textures[e.radiance_texture], e.to_world, sc->texel_tables[i], why)) { delete sc; return nullptr; }
        }
        if (e.type == 2) {
            if (e.mesh >= sc->textures.size()) { delete sc; return nullptr; }
            const Texture &t = sc->textures[e.mesh];
            sc->envmap.init(t.data.data(), t.w, t.h, e.radiance[0], e.radiance[1] != 0.f, e.to_world, e.to_local);
        }
    }
    for (uint32_t i = 0; i < sc->bsdfs.size(); ++i) {
        if (sc->bsdfs[i].p.type == 3) roughplastic_precompute(sc->bsdfs[i], slot0_mean(*sc, i));
        if (sc->bsdfs[i].p.type == 5) plastic_precompute(sc->bsdfs[i], slot0_mean(*sc, i));
    }
    build_tri_bvh(sc->top, sc->meshes, 0, sc->top_count);
    sc->group_bvh.resize(sc->groups.size());
    for (size_t g = 0; g < sc->groups.size(); ++g) build_tri_bvh(sc->group_bvh[g], sc->meshes, sc->groups[g].first_mesh, sc->groups[g].mesh_count);
    build_instance_bvh(sc);
    scene_update_bounds(*sc);
    return sc;
}
void orc_scene_destroy(void *s) { delete (Scene *) s; }
void orc_scene_set_reflectance(void *s, uint32_t b, const float rgb[3]) {
    Scene *sc = (Scene *) s; for (int i = 0; i < 3; ++i) sc->bsdfs[b].p.reflectance[i] = rgb[i];
    if (sc->bsdfs[b].p.type == 3) roughplastic_precompute(sc->bsdfs[b], slot0_mean(*sc, b));
    if (sc->bsdfs[b].p.type == 5) plastic_precompute(sc->bsdfs[b], slot0_mean(*sc, b));
}
/* Scene::sample_emitter / pdf_emitter as the JIT variants evaluate them (scene.cpp:248-279): n samples -> index, 1 / pmf, re-used sample; index -> pmf */
void orc_scene_sample_emitter(void *s, uint32_t n, const float *sample, uint32_t *index, float *weight, float *reused) {
    const Scene &sc = *(Scene *) s; const uint32_t ne = (uint32_t) sc.emitters.size();
    for (uint32_t k = 0; k < n; ++k) {
        if (ne < 2) { index[k] = ne ? 0u : 0xffffffffu; weight[k] = ne ? 1.f : 0.f; reused[k] = sample[k]; continue; }      /* :251-256 */
        if (sc.choice.weighted) {
            float p;
            index[k] = discrete_sample_reuse(sc.choice.table.pmf.data(), sc.choice.table.cdf.data(), ne, sc.choice.table.sum, sc.choice.table.normalization, sample[k], reused[k], p,
                                             sc.choice.first, sc.choice.last);
            weight[k] = rcp(p);
        } else {
            const float scaled = sample[k] * (float) ne;
            index[k] = std::min((uint32_t) scaled, ne - 1u); weight[k] = (float) ne; reused[k] = scaled - (float) index[k];
        }
    }
}
void orc_scene_pdf_emitter(void *s, uint32_t n, const uint32_t *index, float *pdf) {
    const Scene &sc = *(Scene *) s;
    for (uint32_t k = 0; k < n; ++k) pdf[k] = (sc.emitters.empty() || index[k] >= sc.emitters.size()) ? 0.f : emitter_choice_pmf(sc, index[k]);
}
int orc_scene_set_emitter_weights(void *s, const float *w, uint32_t n) {
    Scene &sc = *(Scene *) s;
    if (n != sc.emitters.size()) return 1;
    std::vector<float> old(n);
    for (uint32_t i = 0; i < n; ++i) { old[i] = sc.emitters[i].sampling_weight; sc.emitters[i].sampling_weight = w[i]; }
    if (!rebuild_emitter_choice(sc)) { for (uint32_t i = 0; i < n; ++i) sc.emitters[i].sampling_weight = old[i]; return 2; }
    return 0;
}
static void refresh_texel_tables(Scene &sc, uint32_t t);
void orc_scene_set_texture_to_uv(void *s, uint32_t t, const float m[6]) {
    Texture &x = ((Scene *) s)->textures[t];
    x.moved = false;
    for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c) { x.xf[r][c] = m[3 * r + c]; x.moved = x.moved || m[3 * r + c] != (r == c ? 1.f : 0.f); }
    refresh_texel_tables(*(Scene *) s, t);
}
/* the texel distributions of the area lights that read texture t (BitmapTexture::parameters_changed -> rebuild_internals, bitmap.cpp:484-493) */
static void refresh_texel_tables(Scene &sc, uint32_t t) {
    for (size_t i = 0; i < sc.emitters.size(); ++i)
        if (sc.emitters[i].type == 7 && sc.emitters[i].radiance_texture == t) { std::string why; texel_table_build(sc.textures[t], sc.emitters[i].to_world, sc.texel_tables[i], why); }
}
void orc_scene_set_texture(void *s, uint32_t t, const float *data) {
    Texture &x = ((Scene *) s)->textures[t]; x.data.assign(data, data + 3 * (size_t) x.w * x.h);
    refresh_texel_tables(*(Scene *) s, t);
}
/* the two texture-side functions of a textured area light, for the known-answer and chi^2 tests: BitmapTexture::sample_position / pdf_position of the bitmap behind
 * emitter `emitter` (type 7).  sample: n x 2 -> uv n x 2, pdf n;  pdf_only != 0: `sample` holds positions, only pdf is written */
int orc_emitter_texture_sample_position(void *s, uint32_t emitter, const float *sample, uint32_t n, float *uv, float *pdf, int pdf_only) {
    Scene &sc = *(Scene *) s;
    if (emitter >= sc.emitters.size() || sc.emitters[emitter].type != 7) return -1;
    const Texture &t = sc.textures[sc.emitters[emitter].radiance_texture]; const TexelTable &tab = sc.texel_tables[emitter];
    for (uint32_t i = 0; i < n; ++i) {
        if (pdf_only) pdf[i] = bitmap_pdf_position(t, tab, sample + 2 * (size_t) i);
        else bitmap_sample_position(t, tab, sample[2 * (size_t) i], sample[2 * (size_t) i + 1], uv + 2 * (size_t) i, pdf[i]);
    }
    return 0;
}
/* DiscreteDistribution2D over a w x h array of values: sample n points -> col, row (n x 2 uint32), pmf (n), re-used sample (n x 2) */
int orc_discrete_distribution_2d_sample(const float *values, uint32_t w, uint32_t h, const float *sample, uint32_t n, uint32_t *pos, float *pmf, float *reused) {
    TexelTable tab; tab.w = w; tab.h = h; tab.marginal.assign(h, 0.f); tab.conditional.assign((size_t) w * h, 0.f);
    double rows = 0.0;
    for (uint32_t y = 0; y < h; ++y) { double row = 0.0; for (uint32_t x = 0; x < w; ++x) { row += (double) values[(size_t) y * w + x]; tab.conditional[(size_t) y * w + x] = (float) row; } rows += row; tab.marginal[y] = (float) rows; }
    tab.inv_normalization = (float) rows; tab.normalization = (float) (1.0 / rows);
    for (uint32_t i = 0; i < n; ++i) texel_table_sample(tab, sample[2 * (size_t) i], sample[2 * (size_t) i + 1], pos[2 * (size_t) i], pos[2 * (size_t) i + 1], pmf[i], reused[2 * (size_t) i], reused[2 * (size_t) i + 1]);
    return 0;
}

/* `active` (nullable): the Mask argument of Scene::ray_intersect_preliminary / ray_test (scene.cpp:216-238) -- a masked lane is not traced and reports
 * dr::zeros<PreliminaryIntersection3f>() (t = inf) / false */
void orc_ray_intersect_masked(void *scene, uint32_t n, const float *o, const float *d, const float *maxt, const uint8_t *active, int mode,
                              float *t, float *u, float *v, uint32_t *prim, uint32_t *shape, uint32_t *inst) {
    const Scene &sc = *(Scene *) scene;
    parallel_lanes(0, n, 0, [&](int, uint64_t b, uint64_t e) {
        for (uint64_t i = b; i < e; ++i) {
            Ray r; r.o = V3(o[i], o[n + i], o[2 * (size_t) n + i]); r.d = V3(d[i], d[n + i], d[2 * (size_t) n + i]); r.maxt = maxt[i];
            PI pi; if (!active || active[i]) scene_trace<false>(sc, r, pi, mode);
            t[i] = pi.t; u[i] = pi.u; v[i] = pi.v; prim[i] = pi.prim; shape[i] = pi.shape; inst[i] = pi.inst;
        }
    });
}
void orc_ray_intersect(void *scene, uint32_t n, const float *o, const float *d, const float *maxt, int mode,
                       float *t, float *u, float *v, uint32_t *prim, uint32_t *shape, uint32_t *inst) {
    orc_ray_intersect_masked(scene, n, o, d, maxt, nullptr, mode, t, u, v, prim, shape, inst);
}
void orc_ray_test_masked(void *scene, uint32_t n, const float *o, const float *d, const float *maxt, const uint8_t *active, int mode, uint8_t *hit) {
    const Scene &sc = *(Scene *) scene;
    parallel_lanes(0, n, 0, [&](int, uint64_t b, uint64_t e) {
        for (uint64_t i = b; i < e; ++i) {
            Ray r; r.o = V3(o[i], o[n + i], o[2 * (size_t) n + i]); r.d = V3(d[i], d[n + i], d[2 * (size_t) n + i]); r.maxt = maxt[i];
            PI pi; hit[i] = (!active || active[i]) && scene_trace<true>(sc, r, pi, mode) ? 1 : 0;
        }
    });
}
void orc_ray_test(void *scene, uint32_t n, const float *o, const float *d, const float *maxt, int mode, uint8_t *hit) { orc_ray_test_masked(scene, n, o, d, maxt, nullptr, mode, hit); }

int orc_render_path(void *scene, const OrcSensor *s, uint32_t seed, uint32_t spp, int32_t max_depth, int32_t rr_depth,
                    uint64_t lb, uint64_t le, float *film, OrcStats *stats, int threads) {
    return render_forward(*(Scene *) scene, *s, seed, spp, max_depth, rr_depth, lb, le, film, stats, threads, false);
}
int orc_render_path_passes(void *scene, const OrcSensor *s, uint32_t seed, uint32_t spp, uint32_t spp_per_pass, int32_t max_depth, int32_t rr_depth,
                           uint64_t lb, uint64_t le, float *film, OrcStats *stats, int threads) {
    return render_forward_passes(*(Scene *) scene, *s, seed, spp, spp_per_pass, max_depth, rr_depth, lb, le, film, stats, threads);
}
int orc_render_prb(void *scene, const OrcSensor *s, uint32_t seed, uint32_t spp, int32_t max_depth, int32_t rr_depth,
                   uint64_t lb, uint64_t le, float *film, OrcStats *stats, int threads) {
    return render_forward(*(Scene *) scene, *s, seed, spp, max_depth, rr_depth, lb, le, film, stats, threads, true);
}

/* SamplingIntegrator::sample (include/mitsuba/render/integrator.h:432-437; PathIntegrator::sample path.cpp:94-346, PRBIntegrator.sample primal
 * prb.py:68-339) for n caller-supplied rays: ray i uses the sampler stream of wavefront lane lane_offset + i (Sampler::seed, sampler.cpp:129-148),
 * continued from state[i] when `state` is given.  rgb is 3 x n (SoA), valid[i] = the returned mask, state_out[i] (nullable) = the stream's
 * state after the call. */
int orc_integrator_sample_masked(void *scene, int prb, uint32_t n, const float *o, const float *d, const float *maxt, uint32_t seed, uint32_t lane_offset,
                                 const uint64_t *state, const uint8_t *active, int32_t max_depth, int32_t rr_depth, float *rgb, uint8_t *valid, uint64_t *state_out, int threads);
int orc_integrator_sample(void *scene, int prb, uint32_t n, const float *o, const float *d, const float *maxt, uint32_t seed, uint32_t lane_offset,
                          const uint64_t *state, int32_t max_depth, int32_t rr_depth, float *rgb, uint8_t *valid, uint64_t *state_out, int threads) {
    return orc_integrator_sample_masked(scene, prb, n, o, d, maxt, seed, lane_offset, state, nullptr, max_depth, rr_depth, rgb, valid, state_out, threads);
}
/* ... with the Mask argument of SamplingIntegrator::sample (integrator.h:432-437): the loop condition of a masked lane is false from the start (path.cpp:161-166
 * `active` enters the loop state; prb.py:97), so it draws nothing, returns zero radiance and valid = false */
int orc_integrator_sample_masked(void *scene, int prb, uint32_t n, const float *o, const float *d, const float *maxt, uint32_t seed, uint32_t lane_offset,
                                 const uint64_t *state, const uint8_t *active, int32_t max_depth, int32_t rr_depth, float *rgb, uint8_t *valid, uint64_t *state_out, int threads) {
    Scene &sc = *(Scene *) scene;
    threads = resolve_threads(threads);
    std::vector<ThreadStats> sts(threads);
    const uint32_t md = max_depth < 0 ? 0xffffffffu : (uint32_t) max_depth, rd = (uint32_t) rr_depth;
    parallel_lanes(0, n, threads, [&](int t, uint64_t b, uint64_t e) {
        for (uint64_t i = b; i < e; ++i) {
            Pcg32 rng = sampler_seed(seed, lane_offset + (uint32_t) i);
            if (state) rng.state = state[i];
            Ray ray; ray.o = V3(o[i], o[n + i], o[2 * (size_t) n + i]); ray.d = V3(d[i], d[n + i], d[2 * (size_t) n + i]); ray.maxt = maxt[i];
            bool v = false; V3 L(0.f);
            if (active && !active[i]) { }
            else if (prb) L = prb_sample(sc, rng, ray, md, rd, true, V3(0.f), V3(0.f), nullptr, v, sts[t]);
            else          L = path_sample(sc, rng, ray, md, rd, v, sts[t]);
            rgb[i] = L.x; rgb[n + i] = L.y; rgb[2 * (size_t) n + i] = L.z;
            if (valid) valid[i] = v ? 1 : 0;
            if (state_out) state_out[i] = rng.state;
        }
    });
    return 0;
}

int orc_render_prb_backward_ex(void *scene, const OrcSensor *sp, const float *grad_in, uint32_t seed, uint32_t spp,
                               int32_t max_depth, int32_t rr_depth, float *grad_reflectance, float *const *grad_textures,
                               float *grad_emitters, OrcStats *stats, int threads);
int orc_render_prb_backward(void *scene, const OrcSensor *sp, const float *grad_in, uint32_t seed, uint32_t spp,
                            int32_t max_depth, int32_t rr_depth, float *grad_reflectance, float *const *grad_textures,
                            OrcStats *stats, int threads) {
    return orc_render_prb_backward_ex(scene, sp, grad_in, seed, spp, max_depth, rr_depth, grad_reflectance, grad_textures, nullptr, stats, threads);
}
void orc_scene_set_alpha_only(void *scene, int on) { ((Scene *) scene)->alpha_only = on != 0; }
void orc_scene_set_hide_emitters(void *scene, int hide) { ((Scene *) scene)->hide_emitters = hide != 0; }
void orc_scene_set_emitter_radiance(void *scene, uint32_t emitter, const float rgb[3]) {
    Scene &sc = *(Scene *) scene;
    if (emitter < sc.emitters.size()) for (int c = 0; c < 3; ++c) sc.emitters[emitter].radiance[c] = rgb[c];
}
/* weight-only splat of lanes [lb, le): W[px] of the dummy L = 1 film (common.py:716-746) */
static void render_weights_impl(const OrcSensor &s, uint32_t seed, uint32_t spp, uint64_t lb, uint64_t le, float *film, int threads) {
    RFilter rf = make_rfilter(s.rfilter, s.rfilter_stddev, s.rfilter_param1);
    const size_t npx = (size_t) s.crop_width * s.crop_height;
    std::vector<std::vector<float>> films(threads);
    parallel_lanes(lb, le, threads, [&](int t, uint64_t b, uint64_t e) {
        if (films[t].empty()) films[t].assign(npx * 4, 0.f);
        for (uint64_t i = b; i < e; ++i) {
            Lane L = make_lane(s, seed, spp, i);
            float v[4] = { 0.f, 0.f, 0.f, 1.f };
            film_put(s, rf, rf.type == 0 ? L.ipos_x : L.pos_x, rf.type == 0 ? L.ipos_y : L.pos_y, v, films[t].data());
        }
    });
    for (auto &f : films) if (!f.empty()) for (size_t i = 0; i < npx * 4; ++i) film[i] += f[i];
}
int orc_render_weights(const OrcSensor *s, uint32_t seed, uint32_t spp, uint64_t lb, uint64_t le, float *film, int threads) {
    uint64_t total = sample_grid_pixels(*s) * spp;
    if (total > 0xffffffffull) return -1;
    if (lb == 0 && le == 0) le = total;
    if (lb > le || le > total) return -1;
    render_weights_impl(*s, seed, spp, lb, le, film, resolve_threads(threads));
    return 0;
}
/* lanes [lb, le) only (0, 0 = all); weight_film != NULL: the accumulated weights of ALL lanes come from the caller (a rank of a multi-GPU job
 * holds the all-reduced weight film), otherwise they are computed here over the whole wavefront */
static int prb_backward_impl(void *scene, const OrcSensor *sp, const float *grad_in, uint32_t seed, uint32_t spp,
                             int32_t max_depth, int32_t rr_depth, float *grad_reflectance, float *const *grad_textures,
                             float *grad_emitters, const uint8_t *pos_mask, double *const *grad_positions, OrcStats *stats, int threads,
                             uint64_t lb = 0, uint64_t le = 0, const float *weight_film = nullptr, float *grad_bsdf_params = nullptr,
                             const uint8_t *inst_mask = nullptr, double *grad_to_world = nullptr) {
    Scene &sc = *(Scene *) scene; const OrcSensor &s = *sp;
    /* any BSDF model may sit on (or next to) the moving geometry: the attached BSDF value is differentiated in si.wi / wo numerically (bsdf_dir_grad_fd) */
    if (pos_mask) {           /* flat-shaded top-level meshes */
        for (size_t m = 0; m < sc.meshes.size(); ++m) {
            if (!pos_mask[m]) continue;
            if (m >= sc.top_count && inst_mask) return -3;       /* instance.cpp:162-166: "Cannot differentiate instance parameters and shapegroup internal parameters at the same time!" */
        }
    }
    uint64_t total = sample_grid_pixels(s) * spp;
    if (total > 0xffffffffull) return -1;
    RFilter rf = make_rfilter(s.rfilter, s.rfilter_stddev, s.rfilter_param1);
    threads = resolve_threads(threads);
    uint32_t W = s.crop_width, H = s.crop_height;
    size_t npx = (size_t) W * H;
    if (lb == 0 && le == 0) le = total;
    if (lb > le || le > total) return -1;
    // (1) weight-only splat: W[px] of the dummy L=1 film (common.py:716-746)
    std::vector<float> wfilm(npx * 4, 0.f);
    if (weight_film) std::copy(weight_film, weight_film + npx * 4, wfilm.begin());
    else render_weights_impl(s, seed, spp, 0, total, wfilm.data(), threads);
    // adjoint image: grad_in / W  (adjoint of hdrfilm.cpp:398-399)
    std::vector<float> adj(npx * 3);
    for (size_t i = 0; i < npx; ++i) { float w = wfilm[4 * i + 3]; float iw = w == 0.f ? 1.f : w; for (int c = 0; c < 3; ++c) adj[3 * i + c] = grad_in[3 * i + c] / iw; }
    // per-thread gradient buffers
    size_t nb = sc.bsdfs.size();
    /* the scalar slots (constant albedos, emitter radiance, alpha / eta / k) receive a term from every vertex of every path: they are summed in float -- like the
     * scatter_reduce of the reference -- over blocks of at most 8192 lanes and the block sums in double, so that the total does not depend on how many lanes a
     * worker happens to process (a single float accumulator per worker stagnates: 0.3 % low after 10^6 paths, which 256 workers hid and 16 did not) */
    std::vector<std::vector<double>> g_refl(threads), g_emit(threads), g_extra(threads);
    std::vector<std::vector<std::vector<float>>> g_tex(threads);
    std::vector<std::vector<std::vector<double>>> g_pos(threads), g_nrm(threads);
    std::vector<std::vector<double>> g_inst(threads);
    std::vector<ThreadStats> sts(threads);
    uint32_t md = (uint32_t) max_depth, rd = (uint32_t) rr_depth;
    parallel_lanes(lb, le, threads, [&](int t, uint64_t b, uint64_t e) {
        if (g_refl[t].empty()) {
            g_refl[t].assign(3 * nb + 3, 0.0); g_emit[t].assign(3 * sc.emitters.size() + 3, 0.0); g_extra[t].assign(15 * nb + 15, 0.0);
            g_tex[t].resize(sc.textures.size());
            for (size_t k = 0; k < sc.textures.size(); ++k) g_tex[t][k].assign(3 * (size_t) sc.textures[k].w * sc.textures[k].h, 0.f);
        }
        std::vector<float *> tp(sc.textures.size() + 1, nullptr);
        for (size_t k = 0; k < sc.textures.size(); ++k) tp[k] = g_tex[t][k].data();
        std::vector<float> b_refl(g_refl[t].size()), b_emit(g_emit[t].size()), b_extra(g_extra[t].size());     /* sums of the current block of lanes */
        GradSink sink{ b_refl.data(), tp.data(), grad_emitters ? b_emit.data() : nullptr };
        sink.extra = grad_bsdf_params ? b_extra.data() : nullptr;
        sink.light_texels = grad_emitters != nullptr;       /* emitter gradients: colours go to grad_emitters, the texels of a bitmap radiance to that bitmap's entry of grad_textures */
        auto flush_block = [&]() {
            for (size_t k = 0; k < b_refl.size(); ++k) { g_refl[t][k] += (double) b_refl[k]; b_refl[k] = 0.f; }
            for (size_t k = 0; k < b_emit.size(); ++k) { g_emit[t][k] += (double) b_emit[k]; b_emit[k] = 0.f; }
            for (size_t k = 0; k < b_extra.size(); ++k) { g_extra[t][k] += (double) b_extra[k]; b_extra[k] = 0.f; }
        };
        std::vector<double *> pp(sc.meshes.size() + 1, nullptr), pn(sc.meshes.size() + 1, nullptr); ShapeSink shape{ pp.data(), pos_mask };
        if (pos_mask) {
            if (g_pos[t].empty()) {
                g_pos[t].resize(sc.meshes.size()); g_nrm[t].resize(sc.meshes.size());
                for (size_t m = 0; m < sc.meshes.size(); ++m) if (pos_mask[m]) { g_pos[t][m].assign(3 * (size_t) sc.meshes[m].nv, 0.0); if (sc.meshes[m].flags & 1u) g_nrm[t][m].assign(3 * (size_t) sc.meshes[m].nv, 0.0); }
            }
            for (size_t m = 0; m < sc.meshes.size(); ++m) { pp[m] = pos_mask[m] ? g_pos[t][m].data() : nullptr; pn[m] = (pos_mask[m] && !g_nrm[t][m].empty()) ? g_nrm[t][m].data() : nullptr; }
            shape.nrm = pn.data();
            sink.shape = &shape;
        }
        if (inst_mask) {
            if (g_inst[t].empty()) g_inst[t].assign(12 * sc.instances.size() + 12, 0.0);
            shape.inst = g_inst[t].data(); shape.inst_mask = inst_mask; sink.shape = &shape;
        }
        for (uint64_t i = b; i < e; ++i) {
            Lane L = make_lane(s, seed, spp, i);
            // dL = adjoint of the splat (gather over the filter footprint)
            V3 dL(0.f);
            {
                float px = rf.type == 0 ? L.ipos_x : L.pos_x, py = rf.type == 0 ? L.ipos_y : L.pos_y;
                if (rf.type == 0) {
                    int32_t x = (int32_t) std::floor(px) - (int32_t) s.crop_offset_x, y = (int32_t) std::floor(py) - (int32_t) s.crop_offset_y;
                    if ((uint32_t) x < W && (uint32_t) y < H) { const float *a = &adj[3 * ((size_t) y * W + x)]; dL = V3(a[0], a[1], a[2]); }
                } else {
                    uint32_t n = (uint32_t) std::ceil(rf.radius - .5f), count = 2 * n + 1;
                    int32_t ix = (int32_t) std::floor(px) - (int32_t) n, iy = (int32_t) std::floor(py) - (int32_t) n;
                    uint32_t x0 = (uint32_t) (ix - (int32_t) s.crop_offset_x), y0 = (uint32_t) (iy - (int32_t) s.crop_offset_y);
                    float relx = ((float) ix + .5f) - px, rely = ((float) iy + .5f) - py;
                    for (uint32_t ys = 0; ys < count; ++ys) {
                        uint32_t y = y0 + ys; if (!(y < H)) continue;
                        float wy = rfilter_eval(rf, rely + (float) ys);
                        for (uint32_t xs = 0; xs < count; ++xs) {
                            uint32_t x = x0 + xs; if (!(x < W)) continue;
                            float w = rfilter_eval(rf, relx + (float) xs) * wy;
                            const float *a = &adj[3 * ((size_t) y * W + x)];
                            dL = V3(fmadd(a[0], w, dL.x), fmadd(a[1], w, dL.y), fmadd(a[2], w, dL.z));
                        }
                    }
                }
            }
            bool valid;
            Pcg32 rng2 = L.rng;                               // sampler.clone(): identical stream (common.py:755,768)
            OrcStats dummy{};
            V3 Lp = prb_sample(sc, rng2, L.ray, md, rd, true, V3(0.f), V3(0.f), nullptr, valid, dummy);
            prb_sample(sc, L.rng, L.ray, md, rd, false, Lp, dL, &sink, valid, sts[t]);
            sts[t].paths++;
            if (((i - b) & 8191u) == 8191u) flush_block();
        }
        flush_block();
    });
    std::vector<double> t_refl(3 * nb + 3, 0.0), t_emit(3 * sc.emitters.size() + 3, 0.0), t_extra(15 * nb + 15, 0.0);
    for (int t = 0; t < threads; ++t) {
        if (g_refl[t].empty()) continue;
        for (size_t i = 0; i < 3 * nb; ++i) t_refl[i] += g_refl[t][i];
        for (size_t i = 0; i < 3 * sc.emitters.size(); ++i) t_emit[i] += g_emit[t][i];
        for (size_t i = 0; i < 15 * nb; ++i) t_extra[i] += g_extra[t][i];
    }
    if (grad_reflectance) for (size_t i = 0; i < 3 * nb; ++i) grad_reflectance[i] += (float) t_refl[i];
    if (grad_emitters) for (size_t i = 0; i < 3 * sc.emitters.size(); ++i) grad_emitters[i] += (float) t_emit[i];
    if (grad_bsdf_params) for (size_t i = 0; i < 15 * nb; ++i) grad_bsdf_params[i] += (float) t_extra[i];
    for (int t = 0; t < threads; ++t) {
        if (g_refl[t].empty()) continue;
        for (size_t k = 0; k < sc.textures.size(); ++k)
            if (grad_textures && grad_textures[k]) { float *dst = grad_textures[k]; for (size_t i = 0; i < g_tex[t][k].size(); ++i) dst[i] += g_tex[t][k][i]; }
        if (inst_mask && grad_to_world && !g_inst[t].empty()) for (size_t i = 0; i < 12 * sc.instances.size(); ++i) grad_to_world[i] += g_inst[t][i];
        if (pos_mask && !g_pos[t].empty())
            for (size_t m = 0; m < sc.meshes.size(); ++m) if (pos_mask[m] && grad_positions[m]) for (size_t i = 0; i < g_pos[t][m].size(); ++i) grad_positions[m][i] += g_pos[t][m][i];
    }
    if (pos_mask)           /* second stage of the vertex-normal derivative: the summed normal adjoints through compute_normals, once per face */
        for (size_t m = 0; m < sc.meshes.size(); ++m) {
            if (!pos_mask[m] || !grad_positions[m] || !(sc.meshes[m].flags & 1u)) continue;
            std::vector<double> nbar(3 * (size_t) sc.meshes[m].nv, 0.0);
            for (int t = 0; t < threads; ++t) if (!g_nrm[t].empty() && !g_nrm[t][m].empty()) for (size_t i = 0; i < nbar.size(); ++i) nbar[i] += g_nrm[t][m][i];
            normals_backward(sc.meshes[m], nbar.data(), grad_positions[m]);
        }
    merge_stats(stats, sts);
    return 0;
}
int orc_render_prb_backward_ex(void *scene, const OrcSensor *sp, const float *grad_in, uint32_t seed, uint32_t spp,
                               int32_t max_depth, int32_t rr_depth, float *grad_reflectance, float *const *grad_textures,
                               float *grad_emitters, OrcStats *stats, int threads) {
    return prb_backward_impl(scene, sp, grad_in, seed, spp, max_depth, rr_depth, grad_reflectance, grad_textures, grad_emitters, nullptr, nullptr, stats, threads);
}
/* RBIntegrator.render_forward (common.py:497-623): raw film (H x W x 4) of the lanes' differential radiance for the given parameter tangents
 * (layout of the gradient buffers of orc_render_prb_backward_ex; tangent_emitters may be NULL) */
int orc_render_prb_forward(void *scene, const OrcSensor *sp, uint32_t seed, uint32_t spp, int32_t max_depth, int32_t rr_depth, const float *tangent_reflectance,
                           const float *const *tangent_textures, const float *tangent_emitters, float *film, int threads) {
    Scene &sc = *(Scene *) scene; const OrcSensor &s = *sp;
    uint64_t total = sample_grid_pixels(s) * spp;
    if (total > 0xffffffffull) return -1;
    RFilter rf = make_rfilter(s.rfilter, s.rfilter_stddev, s.rfilter_param1);
    threads = resolve_threads(threads);
    size_t fsz = (size_t) s.crop_width * s.crop_height * 4;
    std::vector<std::vector<float>> films(threads);
    uint32_t md = (uint32_t) max_depth, rd = (uint32_t) rr_depth;
    std::vector<float> zero_emit(3 * sc.emitters.size() + 3, 0.f);
    parallel_lanes(0, total, threads, [&](int t, uint64_t b, uint64_t e) {
        if (films[t].empty()) films[t].assign(fsz, 0.f);
        for (uint64_t i = b; i < e; ++i) {
            Lane L = make_lane(s, seed, spp, i);
            bool valid; OrcStats dummy{};
            Pcg32 rng2 = L.rng;                               // sampler.clone() for the primal pass (common.py:569-578)
            V3 Lp = prb_sample(sc, rng2, L.ray, md, rd, true, V3(0.f), V3(0.f), nullptr, valid, dummy);
            V3 acc(0.f);
            GradSink sink{ const_cast<float *>(tangent_reflectance), const_cast<float *const *>(tangent_textures),
                           const_cast<float *>(tangent_emitters ? tangent_emitters : zero_emit.data()), nullptr, &acc };
            prb_sample(sc, L.rng, L.ray, md, rd, false, Lp, V3(1.f), &sink, valid, dummy);
            float v[4] = { acc.x, acc.y, acc.z, 1.f };
            film_put(s, rf, rf.type == 0 ? L.ipos_x : L.pos_x, rf.type == 0 ? L.ipos_y : L.pos_y, v, films[t].data());
        }
    });
    for (auto &f : films) if (!f.empty()) for (size_t i = 0; i < fsz; ++i) film[i] += f[i];
    return 0;
}
/* ... plus the gradients w.r.t. alpha_u, alpha_v, eta, k, colour slot 1 of the rough BSDF records: grad_bsdf_params = bsdf_count x 15, added to */
int orc_render_prb_backward_bsdf_params(void *scene, const OrcSensor *sp, const float *grad_in, uint32_t seed, uint32_t spp, int32_t max_depth, int32_t rr_depth,
                                        float *grad_reflectance, float *const *grad_textures, float *grad_bsdf_params, OrcStats *stats, int threads) {
    return prb_backward_impl(scene, sp, grad_in, seed, spp, max_depth, rr_depth, grad_reflectance, grad_textures, nullptr, nullptr, nullptr, stats, threads, 0, 0, nullptr, grad_bsdf_params);
}
int orc_render_prb_backward_lanes(void *scene, const OrcSensor *sp, const float *grad_in, const float *weight_film, uint32_t seed, uint32_t spp,
                                  int32_t max_depth, int32_t rr_depth, uint64_t lane_begin, uint64_t lane_end, float *grad_reflectance,
                                  float *const *grad_textures, float *grad_emitters, OrcStats *stats, int threads) {
    return prb_backward_impl(scene, sp, grad_in, seed, spp, max_depth, rr_depth, grad_reflectance, grad_textures, grad_emitters, nullptr, nullptr, stats, threads,
                             lane_begin, lane_end, weight_film);
}
/* + d/d(vertex positions) of the meshes with pos_mask[m] != 0: grad_positions[m] = 3 doubles per vertex (accumulated into).
 * Returns -2 when the scene holds a BSDF other than plain `diffuse`, -3 for a mesh with vertex normals / inside a shape group. */
int orc_render_prb_backward_shape(void *scene, const OrcSensor *sp, const float *grad_in, uint32_t seed, uint32_t spp,
                                  int32_t max_depth, int32_t rr_depth, float *grad_reflectance, float *const *grad_textures,
                                  const uint8_t *pos_mask, double *const *grad_positions, OrcStats *stats, int threads) {
    return prb_backward_impl(scene, sp, grad_in, seed, spp, max_depth, rr_depth, grad_reflectance, grad_textures, nullptr, pos_mask, grad_positions, stats, threads);
}
/* + d/d(to_world) of the instances with inst_mask[i] != 0 (Instance::compute_surface_interaction, instance.cpp:150-266, with an attached transform):
 * grad_to_world = 12 doubles per instance, column-major 3x4 like OrcInstance::to_world (accumulated into).  -2: a BSDF other than `diffuse`. */
int orc_render_prb_backward_instances(void *scene, const OrcSensor *sp, const float *grad_in, uint32_t seed, uint32_t spp,
                                      int32_t max_depth, int32_t rr_depth, float *grad_reflectance, float *const *grad_textures,
                                      const uint8_t *inst_mask, double *grad_to_world, OrcStats *stats, int threads) {
    return prb_backward_impl(scene, sp, grad_in, seed, spp, max_depth, rr_depth, grad_reflectance, grad_textures, nullptr, nullptr, nullptr, stats, threads,
                             0, 0, nullptr, nullptr, inst_mask, grad_to_world);
}
void orc_scene_set_vertex_positions(void *scene, uint32_t mesh, const float *positions) {       /* + rebuild of the acceleration structure */
    Scene &sc = *(Scene *) scene;
    if (mesh >= sc.meshes.size()) return;
    Mesh &m = sc.meshes[mesh];
    for (uint32_t i = 0; i < m.nv; ++i) for (int c = 0; c < 3; ++c) m.V[8 * (size_t) i + c] = positions[3 * (size_t) i + c];
    if (m.flags & 1u) mesh_regenerate_normals(m);             /* mesh.cpp:876-878: writing the positions regenerates the vertex normals */
    if (mesh < sc.top_count) build_tri_bvh(sc.top, sc.meshes, 0, sc.top_count);
    else {                                                    /* a mesh inside a shape group: that group's BVH and the boxes of its instances */
        for (size_t g = 0; g < sc.groups.size(); ++g)
            if (mesh >= sc.groups[g].first_mesh && mesh < sc.groups[g].first_mesh + sc.groups[g].mesh_count) build_tri_bvh(sc.group_bvh[g], sc.meshes, sc.groups[g].first_mesh, sc.groups[g].mesh_count);
        build_instance_bvh(&sc);
    }
    scene_update_bounds(sc);
}

void orc_film_develop(const float *film, uint32_t width, uint32_t height, float *image) {
    for (size_t i = 0; i < (size_t) width * height; ++i) {
        float w = film[4 * i + 3]; float dv = w == 0.f ? 1.f : w;
        for (int c = 0; c < 3; ++c) image[3 * i + c] = film[4 * i + c] / dv;
    }
}

// ---- unit-level ----
void orc_sample_tea_32(uint32_t v0, uint32_t v1, int rounds, uint32_t out[2]) { sample_tea_32(v0, v1, rounds, out[0], out[1]); }
/* random.h:135-140 / 160-166: mantissa fill + subtract 1 */
float orc_sample_tea_float32(uint32_t v0, uint32_t v1, int rounds) { uint32_t a, b; sample_tea_32(v0, v1, rounds, a, b); return u2f((b >> 9) | 0x3f800000u) - 1.f; }
double orc_sample_tea_float64(uint32_t v0, uint32_t v1, int rounds) {
    uint32_t a, b; sample_tea_32(v0, v1, rounds, a, b);
    uint64_t v = (uint64_t) a + ((uint64_t) b << 32);
    uint64_t bits = (v >> 12) | 0x3ff0000000000000ull; double dd; std::memcpy(&dd, &bits, 8); return dd - 1.0;
}
void orc_pcg32_seed(uint64_t initstate, uint64_t initseq, uint64_t si[2]) { Pcg32 r; r.seed(initstate, initseq); si[0] = r.state; si[1] = r.inc; }
uint32_t orc_pcg32_next_uint32(uint64_t si[2]) { Pcg32 r; r.state = si[0]; r.inc = si[1]; uint32_t v = r.next_uint32(); si[0] = r.state; return v; }
float orc_pcg32_next_float32(uint64_t si[2]) { Pcg32 r; r.state = si[0]; r.inc = si[1]; float v = r.next_float32(); si[0] = r.state; return v; }
void orc_sampler_stream(uint32_t seed, uint32_t lane, uint32_t n, float *out) { Pcg32 r = sampler_seed(seed, lane); for (uint32_t i = 0; i < n; ++i) out[i] = r.next_float32(); }
float orc_rfilter_eval2(uint32_t type, float p0, float p1, float x) { RFilter f = make_rfilter(type, p0, p1); return type == 0 ? (x >= -.5f && x < .5f ? 1.f : 0.f) : rfilter_eval(f, x); }
float orc_rfilter_eval(uint32_t type, float stddev, float x) { RFilter f = make_rfilter(type, stddev); return type == 0 ? (std::fabs(x) <= .5f ? 1.f : 0.f) : rfilter_eval(f, x); }
void orc_film_put(const OrcSensor *s, uint32_t n, const float *px, const float *py, const float *values4, float *film) {
    RFilter rf = make_rfilter(s->rfilter, s->rfilter_stddev, s->rfilter_param1);
    for (uint32_t i = 0; i < n; ++i) film_put(*s, rf, px[i], py[i], values4 + 4 * (size_t) i, film);
}
void orc_sensor_sample_ray(const OrcSensor *s, uint32_t n, const float *px, const float *py, float *o, float *d, float *maxt) {
    for (uint32_t i = 0; i < n; ++i) {
        Ray r = sensor_sample_ray(*s, px[i], py[i]);
        o[i] = r.o.x; o[n + i] = r.o.y; o[2 * (size_t) n + i] = r.o.z;
        d[i] = r.d.x; d[n + i] = r.d.y; d[2 * (size_t) n + i] = r.d.z; maxt[i] = r.maxt;
    }
}
void orc_diffuse_eval_pdf(const float refl[3], const float wi[3], const float wo[3], float value[3], float *pdf) {
    V3 v; diffuse_eval_pdf(V3(refl[0], refl[1], refl[2]), V3(wi[0], wi[1], wi[2]), V3(wo[0], wo[1], wo[2]), v, *pdf);
    value[0] = v.x; value[1] = v.y; value[2] = v.z;
}
void orc_diffuse_sample(const float refl[3], const float wi[3], float, const float s2[2], float wo[3], float *pdf, float weight[3]) {
    V3 w, wt; diffuse_sample(V3(refl[0], refl[1], refl[2]), V3(wi[0], wi[1], wi[2]), s2[0], s2[1], w, *pdf, wt);
    wo[0] = w.x; wo[1] = w.y; wo[2] = w.z; weight[0] = wt.x; weight[1] = wt.y; weight[2] = wt.z;
}
/* Emitter::sample_direction of emitter `index` alone (no emitter choice, no visibility test) from the reference points p[n][3] with samples s[n][2]:
 * d[n][3], dist[n], pdf[n], delta[n], weight[n][3] -- the quantities the reference's emitter tests look at (src/emitters/tests/test_*.py) */
void orc_emitter_sample_direction(void *scene, uint32_t index, uint32_t n, const float *p, const float *s, float *d, float *dist, float *pdf, uint8_t *delta, float *weight) {
    const Scene &sc = *(const Scene *) scene;
    for (uint32_t i = 0; i < n; ++i) {
        DS ds; V3 spec(0.f);
        const V3 ref(p[3 * i], p[3 * i + 1], p[3 * i + 2]);
        const OrcEmitter &e = sc.emitters[index];
        if (e.type == 1) { EnvSphere bs; bs.center = V3(sc.env_center[0], sc.env_center[1], sc.env_center[2]); bs.radius = sc.env_radius; constant_sample_direction(e, bs, ref, s[2 * i], s[2 * i + 1], ds, spec, nullptr); }
        else if (e.type == 2) { float uv[2]; sc.envmap.sample_direction(ref, s[2 * i], s[2 * i + 1], ds.d, ds.dist, ds.pdf, spec, uv); }
        else emitter_sample_direction(sc, index, ref, s[2 * i], s[2 * i + 1], ds, spec, nullptr);
        d[3 * i] = ds.d.x; d[3 * i + 1] = ds.d.y; d[3 * i + 2] = ds.d.z; dist[i] = ds.dist; pdf[i] = ds.pdf; delta[i] = ds.delta ? 1 : 0;
        weight[3 * i] = spec.x; weight[3 * i + 1] = spec.y; weight[3 * i + 2] = spec.z;
    }
}
/* mi.DiscreteDistribution([pmf...]) as the emitters' face tables build it (Mesh::build_pmf / DiscreteDistribution::update: running float sum, normalization = 1 / sum):
 * sample_reuse_pmf of `n_samples` values -> index, re-used sample, normalised pmf */
void orc_discrete_sample_reuse(const float *pmf, uint32_t n, uint32_t n_samples, const float *values, uint32_t *index, float *reused, float *pmf_out) {
    std::vector<float> cdf(n); float acc = 0.f;
    for (uint32_t i = 0; i < n; ++i) { acc += pmf[i]; cdf[i] = acc; }
    const float normalization = rcp(acc);
    for (uint32_t k = 0; k < n_samples; ++k) index[k] = discrete_sample_reuse(pmf, cdf.data(), n, acc, normalization, values[k], reused[k], pmf_out[k]);
}
void orc_square_to_cosine_hemisphere(const float s[2], float out[3]) { V3 v = square_to_cosine_hemisphere(s[0], s[1]); out[0] = v.x; out[1] = v.y; out[2] = v.z; }
void orc_square_to_uniform_sphere(const float s[2], float out[3]) { V3 v = square_to_uniform_sphere(s[0], s[1]); out[0] = v.x; out[1] = v.y; out[2] = v.z; }
void orc_square_to_uniform_disk_concentric(const float s[2], float out[2]) { square_to_uniform_disk_concentric(s[0], s[1], out[0], out[1]); }
/* Mesh::compute_normals (mesh.cpp:1216-1267) on packed vertices (8 floats each; the normals at offset 3 are overwritten), faces of 4 u32 */
void orc_mesh_compute_normals(uint32_t nv, float *vertices, uint32_t nf, const uint32_t *faces) {
    Mesh m; m.nv = nv; m.nf = nf; m.V.assign(vertices, vertices + 8 * (size_t) nv); m.F.assign(faces, faces + 4 * (size_t) nf); m.bsdf = 0; m.emitter = -1; m.flags = 1u;
    mesh_regenerate_normals(m);
    std::copy(m.V.begin(), m.V.end(), vertices);
}
void orc_coordinate_system(const float n[3], float s[3], float t[3]) { V3 a, b; coordinate_system(V3(n[0], n[1], n[2]), a, b); s[0] = a.x; s[1] = a.y; s[2] = a.z; t[0] = b.x; t[1] = b.y; t[2] = b.z; }
float orc_sincos(float x, float *c) { return sincos(x, c); }
int orc_default_threads(void) { return default_threads(); }
/* the restated Dr.Jit elementary functions of orc_math.h: 0 exp, 1 log, 2 erf, 3 atan2(x, y), 4 acos, 5 tan, 6 erfinv */
float orc_math_fn(int fn, float x, float y) {
    switch (fn) {
        case 0: return exp32(x);   case 1: return log32(x);  case 2: return erf32(x); case 3: return atan2_32(x, y);
        case 4: return acos32(x);  case 5: return tan32(x);  case 6: return erfinv(x);
        case 7: { float c; return sincos(x, &c); }  case 8: { float c; (void) sincos(x, &c); return c; }
    }
    return std::numeric_limits<float>::quiet_NaN();
}
void orc_math_fn_array(int fn, uint32_t n, const float *x, const float *y, float *out) { for (uint32_t i = 0; i < n; ++i) out[i] = orc_math_fn(fn, x[i], y ? y[i] : 0.f); }
/* PreliminaryIntersection::compute_surface_interaction(ray, ray_flags, active) (interaction.h:804-829) WITH its flags and mask, written out once more from
 * Mesh::compute_surface_interaction (src/render/mesh.cpp:2255-2437), Instance::compute_surface_interaction (src/shapes/instance.cpp:150-266) and
 * finalize_surface_interaction (interaction.h:559-605) -- independently of compute_si() above, which only knows RayFlags::Default.
 * out[33] = p, n, sh_frame.n, sh_frame.s, sh_frame.t, wi, uv, t, dp_du, dp_dv, dn_du, dn_dv.  ray_flags: Shading 1, NormalPartials 2 (FollowShape 4 / DetachShape 8
 * change nothing in a primal evaluation). */
void orc_surface_interaction_flags(void *scene, const float o[3], const float d[3], float t, float u, float v, uint32_t prim, uint32_t shape, uint32_t inst,
                                   uint32_t ray_flags, int active_, float out[33]) {
    const Scene &sc = *(Scene *) scene;
    (void) o;
    const bool shading = (ray_flags & 1u) != 0;
    bool active = active_ != 0 && t != Infinity;                                  // interaction.h:811 active &= is_valid()
    V3 si_p(0.f), si_n(0.f), sh_n(0.f), sh_s(0.f), sh_t(0.f), wi(0.f), dp_du(0.f), dp_dv(0.f), dn_du(0.f), dn_dv(0.f);
    float uv0 = 0.f, uv1 = 0.f, si_t = Infinity;
    const V3 ray_d(d[0], d[1], d[2]);
    if (active) {                                                                 // the masked vcall returns dr::zeros<SurfaceInteraction3f>()
        const Mesh &m = sc.meshes[shape];
        const uint32_t *f = &m.F[4 * (size_t) prim];
        const float *rec0 = &m.V[8 * (size_t) f[0]], *rec1 = &m.V[8 * (size_t) f[1]], *rec2 = &m.V[8 * (size_t) f[2]];
        const V3 p0(rec0[0], rec0[1], rec0[2]), p1(rec1[0], rec1[1], rec1[2]), p2(rec2[0], rec2[1], rec2[2]);
        float b1 = u, b2 = v, b0 = 1.f - b1 - b2;
        V3 e1 = p1 - p0, e2 = p2 - p0;
        si_p = fmadd(p0, b0, fmadd(p1, b1, p2 * b2));
        si_n = normalize(cross(e1, e2));
        si_t = t;
        const bool has_normals = (m.flags & 1u) != 0, has_texcoords = (m.flags & 2u) != 0;
        if (shading) {
            bool need_dn = has_normals && (ray_flags & 2u) != 0;
            V3 dn_db1(0.f), dn_db2(0.f);
            if (has_normals) {
                V3 n0(rec0[3], rec0[4], rec0[5]), dn1 = V3(rec1[3], rec1[4], rec1[5]) - n0, dn2 = V3(rec2[3], rec2[4], rec2[5]) - n0;
                V3 n = fmadd(dn1, b1, fmadd(dn2, b2, n0));
                float il = rsqrt(squared_norm(n));
                n = n * il;
                sh_n = n;
                if (need_dn) {
                    dn1 = dn1 * il; dn2 = dn2 * il;
                    dn_db1 = fmadd(n, -dot(n, dn1), dn1);                         // dr::fnmadd(n, dot(n, dn1), dn1)
                    dn_db2 = fmadd(n, -dot(n, dn2), dn2);
                }
            } else sh_n = si_n;
            if (has_texcoords) {
                float uvx0 = rec0[6], uvy0 = rec0[7];
                float duv0x = rec1[6] - uvx0, duv0y = rec1[7] - uvy0, duv1x = rec2[6] - uvx0, duv1y = rec2[7] - uvy0;
                uv0 = fmadd(duv0x, b1, fmadd(duv1x, b2, uvx0)); uv1 = fmadd(duv0y, b1, fmadd(duv1y, b2, uvy0));
                float det = fmsub(duv0x, duv1y, duv0y * duv1x), inv_det = det != 0.f ? rcp(det) : 0.f;
                auto to_uv_basis = [&](V3 d1, V3 d2, V3 &a, V3 &b) {
                    a = V3(fmsub(duv1y, d1.x, duv0y * d2.x), fmsub(duv1y, d1.y, duv0y * d2.y), fmsub(duv1y, d1.z, duv0y * d2.z)) * inv_det;
                    b = V3(fnmadd(duv1x, d1.x, duv0x * d2.x), fnmadd(duv1x, d1.y, duv0x * d2.y), fnmadd(duv1x, d1.z, duv0x * d2.z)) * inv_det;
                };
                to_uv_basis(e1, e2, dp_du, dp_dv);
                if (need_dn) to_uv_basis(dn_db1, dn_db2, dn_du, dn_dv);
            } else {
                uv0 = b1; uv1 = b2; dp_du = e1; dp_dv = e2;
                if (need_dn) { dn_du = dn_db1; dn_dv = dn_db2; }
            }
        }
        if (inst != 0xffffffffu) {                                                // instance.cpp:190-253
            const OrcInstance &in = sc.instances[inst];
            si_p = xf_point(in.to_world, si_p);
            si_n = normalize(xf_normal(in.to_object, si_n));
            if (shading) {
                V3 n = xf_normal(in.to_object, sh_n);
                float inv_len = rcp(norm(n));
                n = n * inv_len;
                sh_n = n;
                if (ray_flags & 2u) {
                    V3 a = xf_normal(in.to_object, dn_du) * inv_len, b = xf_normal(in.to_object, dn_dv) * inv_len;
                    dn_du = fmadd(n, -dot(n, a), a); dn_dv = fmadd(n, -dot(n, b), b);
                }
                dp_du = xf_vector(in.to_world, dp_du); dp_dv = xf_vector(in.to_world, dp_dv);
            }
        }
    }
    if (shading) {                                                                // finalize_surface_interaction: no packed tangents, sh_frame.s == 0 -> coordinate_system(n)
        coordinate_system(sh_n, sh_s, sh_t);
        V3 md = -ray_d;
        wi = active ? V3(dot(md, sh_s), dot(md, sh_t), dot(md, sh_n)) : md;
    }
    const V3 vs[6] = { si_p, si_n, sh_n, sh_s, sh_t, wi };
    for (int i = 0; i < 6; ++i) { out[3 * i] = vs[i].x; out[3 * i + 1] = vs[i].y; out[3 * i + 2] = vs[i].z; }
    out[18] = uv0; out[19] = uv1; out[20] = si_t;
    const V3 ps[4] = { dp_du, dp_dv, dn_du, dn_dv };
    for (int i = 0; i < 4; ++i) { out[21 + 3 * i] = ps[i].x; out[22 + 3 * i] = ps[i].y; out[23 + 3 * i] = ps[i].z; }
}
void orc_surface_interaction(void *scene, const float o[3], const float d[3], float t, float u, float v, uint32_t prim,
                             uint32_t shape, uint32_t inst, float out[24]) {
    const Scene &sc = *(Scene *) scene;
    Ray r; r.o = V3(o[0], o[1], o[2]); r.d = V3(d[0], d[1], d[2]); r.maxt = Largest;
    PI pi; pi.t = t; pi.u = u; pi.v = v; pi.prim = prim; pi.shape = shape; pi.inst = inst;
    SI si = compute_si(sc, r, pi);
    const V3 vs[6] = { si.p, si.n, si.sn, si.ss, si.st, si.wi };
    for (int i = 0; i < 6; ++i) { out[3 * i] = vs[i].x; out[3 * i + 1] = vs[i].y; out[3 * i + 2] = vs[i].z; }
    out[18] = si.uv[0]; out[19] = si.uv[1]; out[20] = si.t; out[21] = out[22] = out[23] = 0.f;
}


void orc_microfacet_eval(int type, float alpha_u, float alpha_v, int sample_visible, const float wi[3], const float m[3], float out[3]) {
    MicrofacetDistribution d(type ? MicrofacetType::GGX : MicrofacetType::Beckmann, alpha_u, alpha_v, sample_visible != 0);
    V3 w(wi[0], wi[1], wi[2]), mm(m[0], m[1], m[2]);
    out[0] = d.eval(mm); out[1] = d.pdf(w, mm); out[2] = d.smith_g1(w, mm);
}
void orc_microfacet_sample(int type, float alpha_u, float alpha_v, int sample_visible, const float wi[3], const float sample[2], float m[3], float *pdf) {
    MicrofacetDistribution d(type ? MicrofacetType::GGX : MicrofacetType::Beckmann, alpha_u, alpha_v, sample_visible != 0);
    V3 r = d.sample(V3(wi[0], wi[1], wi[2]), sample[0], sample[1], *pdf);
    m[0] = r.x; m[1] = r.y; m[2] = r.z;
}
void orc_fresnel(float cos_theta_i, float eta, float out[4]) { FresnelResult f = fresnel(cos_theta_i, eta); out[0] = f.r; out[1] = f.cos_theta_t; out[2] = f.eta_it; out[3] = f.eta_ti; }
float orc_fresnel_conductor(float cos_theta_i, float eta, float k) { return fresnel_conductor(cos_theta_i, eta, k); }
void orc_bsdf_eval_pdf(void *scene, uint32_t bsdf, const float wi[3], const float uv[2], const float wo[3], float value[3], float *pdf) {
    const Scene &sc = *(Scene *) scene;
    SI si; si.wi = V3(wi[0], wi[1], wi[2]); si.uv[0] = uv[0]; si.uv[1] = uv[1];
    BsdfCtx c = bsdf_prepare(sc, bsdf, si);
    BSDFEval e = bsdf_eval_pdf(c, V3(wo[0], wo[1], wo[2]));
    value[0] = e.value.x; value[1] = e.value.y; value[2] = e.value.z; *pdf = e.pdf;
}
void orc_bsdf_sample(void *scene, uint32_t bsdf, const float wi[3], const float uv[2], float sample1, const float sample2[2],
                     float wo[3], float *pdf, float weight[3], float *eta, int *delta) {
    const Scene &sc = *(Scene *) scene;
    SI si; si.wi = V3(wi[0], wi[1], wi[2]); si.uv[0] = uv[0]; si.uv[1] = uv[1];
    BsdfCtx c = bsdf_prepare(sc, bsdf, si);
    V3 w; BSDFSample bs = bsdf_sample(c, sample1, sample2[0], sample2[1], w);
    wo[0] = bs.wo.x; wo[1] = bs.wo.y; wo[2] = bs.wo.z; *pdf = bs.pdf; weight[0] = w.x; weight[1] = w.y; weight[2] = w.z; *eta = bs.eta; *delta = bs.delta ? 1 : 0;
}
/* BSDF::eval / pdf / eval_pdf / sample WITH the BSDFContext and Mask arguments (orc_bsdf_ctx.h): scene BSDF `bsdf`, twosided handled as twosided.cpp does */
static void ctx_setup(const Scene &sc, uint32_t bsdf, const float wi[3], const float uv[2], ctxapi::TwoSided &ts, ctxapi::Si &front, ctxapi::Si &back,
                      std::unique_ptr<ctxapi::Plugin> &p0, std::unique_ptr<ctxapi::Plugin> &p1, bool &twosided) {
    const BsdfRecord &r0 = sc.bsdfs[bsdf];
    twosided = (r0.p.flags & 1u) != 0;
    const BsdfRecord &r1 = (twosided && r0.p.back >= 0) ? sc.bsdfs[(uint32_t) r0.p.back] : r0;
    p0.reset(ctxapi::make_plugin(r0));
    if (&r1 != &r0) p1.reset(ctxapi::make_plugin(r1));
    ts.brdf[0] = p0.get(); ts.brdf[1] = p1 ? p1.get() : p0.get();
    SI si; si.wi = V3(wi[0], wi[1], wi[2]); si.uv[0] = uv[0]; si.uv[1] = uv[1];
    auto fill = [&](const BsdfRecord &r, ctxapi::Si &o) { o.wi = si.wi; o.slot0 = bsdf_reflectance(sc, r.p, si); o.slot1 = V3(r.p.reflectance2[0], r.p.reflectance2[1], r.p.reflectance2[2]); };
    fill(r0, front); fill(r1, back);
}
void orc_bsdf_evaluate_ctx(void *scene, uint32_t bsdf, uint32_t mode, uint32_t type_mask, uint32_t component, int which, int active, const float wi[3], const float uv[2],
                           const float wo[3], float value[3], float *pdf) {
    const Scene &sc = *(Scene *) scene;
    ctxapi::TwoSided ts; ctxapi::Si front, back; std::unique_ptr<ctxapi::Plugin> p0, p1; bool twosided;
    ctx_setup(sc, bsdf, wi, uv, ts, front, back, p0, p1, twosided);
    ctxapi::Context ctx; ctx.mode = mode; ctx.type_mask = type_mask; ctx.component = component;
    V3 v(0.f), w(wo[0], wo[1], wo[2]); float p = 0.f;
    if (twosided) ts.evaluate(which, ctx, front, back, w, active != 0, v, p);
    else if (which == 0) v = p0->eval(ctx, front, w, active != 0);
    else if (which == 1) p = p0->pdf(ctx, front, w, active != 0);
    else p0->eval_pdf(ctx, front, w, active != 0, v, p);
    value[0] = v.x; value[1] = v.y; value[2] = v.z; *pdf = p;
}
void orc_bsdf_sample_ctx(void *scene, uint32_t bsdf, uint32_t mode, uint32_t type_mask, uint32_t component, int active, const float wi[3], const float uv[2], float sample1,
                         const float sample2[2], float wo[3], float *pdf, float weight[3], float *eta, uint32_t *sampled_type, uint32_t *sampled_component) {
    const Scene &sc = *(Scene *) scene;
    ctxapi::TwoSided ts; ctxapi::Si front, back; std::unique_ptr<ctxapi::Plugin> p0, p1; bool twosided;
    ctx_setup(sc, bsdf, wi, uv, ts, front, back, p0, p1, twosided);
    ctxapi::Context ctx; ctx.mode = mode; ctx.type_mask = type_mask; ctx.component = component;
    V3 w(0.f);
    ctxapi::Sample bs = twosided ? ts.sample(ctx, front, back, sample1, sample2[0], sample2[1], active != 0, w)
                                 : p0->sample(ctx, front, sample1, sample2[0], sample2[1], active != 0, w);
    wo[0] = bs.wo.x; wo[1] = bs.wo.y; wo[2] = bs.wo.z; *pdf = bs.pdf; weight[0] = w.x; weight[1] = w.y; weight[2] = w.z; *eta = bs.eta;
    *sampled_type = bs.sampled_type; *sampled_component = bs.sampled_component;
}
/* quad::gauss_legendre (include/mitsuba/core/quad.h:27-90): n nodes and n weights on [-1, 1] (known answers: src/core/tests/test_quad.py:16-22) */
void orc_gauss_legendre(int n, float *nodes, float *weights) {
    std::vector<float> a, b; gauss_legendre(n, a, b);
    for (int i = 0; i < n; ++i) { nodes[i] = a[i]; weights[i] = b[i]; }
}
void orc_roughplastic_tables(void *scene, uint32_t bsdf, float out[66]) {
    const BsdfRecord &b = ((Scene *) scene)->bsdfs[bsdf];
    for (int i = 0; i < 64; ++i) out[i] = i < (int) b.external_transmittance.size() ? b.external_transmittance[i] : 0.f;
    out[64] = b.internal_reflectance; out[65] = b.specular_sampling_weight;
}


int orc_render_path_scalar(void *scene, const OrcSensor *s, uint32_t seed, uint32_t spp, int32_t max_depth, int32_t rr_depth,
                           uint32_t n_threads, float *film, OrcStats *stats, uint32_t *block_size) {
    return render_scalar(*(Scene *) scene, *s, seed, spp, max_depth, rr_depth, n_threads, film, stats, block_size);
}
void orc_morton_decode(uint32_t m, uint32_t out[2]) { morton_decode2(m, out[0], out[1]); }
uint32_t orc_spiral(uint32_t size_x, uint32_t size_y, uint32_t block_size, uint32_t max_blocks, int32_t *out /* [n][5]: off_x, off_y, size_x, size_y, id */) {
    std::vector<SpiralBlock> b = spiral_blocks(size_x, size_y, 0, 0, block_size);
    for (uint32_t i = 0; i < std::min<uint32_t>((uint32_t) b.size(), max_blocks); ++i) { out[5 * i] = b[i].off_x; out[5 * i + 1] = b[i].off_y; out[5 * i + 2] = (int32_t) b[i].size_x; out[5 * i + 3] = (int32_t) b[i].size_y; out[5 * i + 4] = (int32_t) b[i].id; }
    return (uint32_t) b.size();
}


void *orc_hier2d_create(const float *data, uint32_t w, uint32_t h, int normalize) { Hier2D *d = new Hier2D(); if (!d->build(data, w, h, normalize != 0)) { delete d; return nullptr; } return d; }
void orc_hier2d_destroy(void *h) { delete (Hier2D *) h; }
void orc_hier2d_sample(void *h, uint32_t n, const float *s, float *pos, float *pdf) { for (uint32_t i = 0; i < n; ++i) ((Hier2D *) h)->sample(s[2 * i], s[2 * i + 1], pos + 2 * i, pdf[i]); }
void orc_hier2d_invert(void *h, uint32_t n, const float *pos, float *s, float *pdf) { for (uint32_t i = 0; i < n; ++i) ((Hier2D *) h)->invert(pos[2 * i], pos[2 * i + 1], s + 2 * i, pdf[i]); }
void orc_hier2d_eval(void *h, uint32_t n, const float *pos, float *pdf) { for (uint32_t i = 0; i < n; ++i) pdf[i] = ((Hier2D *) h)->eval(pos[2 * i], pos[2 * i + 1]); }
uint32_t orc_hier2d_data(void *h, float *out, uint32_t *table, uint32_t *n_levels) {
    Hier2D *d = (Hier2D *) h;
    if (out) std::copy(d->data.begin(), d->data.end(), out);
    if (table) for (size_t l = 0; l < d->levels.size(); ++l) { table[3 * l] = d->levels[l].width; table[3 * l + 1] = d->levels[l].size; table[3 * l + 2] = d->levels[l].offset; }
    if (n_levels) *n_levels = (uint32_t) d->levels.size();
    return (uint32_t) d->data.size();
}
void *orc_envmap_create(const float *rgb, uint32_t w, uint32_t h, float scale, int mis, const float tw[12], const float tl[12]) { EnvMap *e = new EnvMap(); e->init(rgb, w, h, scale, mis != 0, tw, tl); return e; }
void orc_envmap_destroy(void *e) { delete (EnvMap *) e; }
void orc_envmap_set_bsphere(void *e, const float c[3], float r) { EnvMap *m = (EnvMap *) e; for (int a = 0; a < 3; ++a) m->center[a] = c[a]; m->radius = r; }
void orc_envmap_eval(void *e, uint32_t n, const float *d, float *rgb) { for (uint32_t i = 0; i < n; ++i) { V3 v = ((EnvMap *) e)->eval(V3(d[3 * i], d[3 * i + 1], d[3 * i + 2])); rgb[3 * i] = v.x; rgb[3 * i + 1] = v.y; rgb[3 * i + 2] = v.z; } }
void orc_envmap_sample_direction(void *e, uint32_t n, const float *p, const float *s, float *d, float *dist, float *pdf, float *weight) {
    for (uint32_t i = 0; i < n; ++i) {
        V3 dd, w; float uv[2];
        ((EnvMap *) e)->sample_direction(V3(p[3 * i], p[3 * i + 1], p[3 * i + 2]), s[2 * i], s[2 * i + 1], dd, dist[i], pdf[i], w, uv);
        d[3 * i] = dd.x; d[3 * i + 1] = dd.y; d[3 * i + 2] = dd.z; weight[3 * i] = w.x; weight[3 * i + 1] = w.y; weight[3 * i + 2] = w.z;
    }
}
void orc_envmap_pdf_direction(void *e, uint32_t n, const float *d, float *pdf) { for (uint32_t i = 0; i < n; ++i) pdf[i] = ((EnvMap *) e)->pdf_direction(V3(d[3 * i], d[3 * i + 1], d[3 * i + 2])); }

} // extern "C"
