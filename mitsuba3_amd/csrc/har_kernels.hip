/*
 * har_kernels.hip -- gfx950 kernels of the hip_ad_rgb wavefront path tracer.
 *
 * Pipeline per chunk of lanes (one lane = one Monte-Carlo sample; lane order and
 * seeding are the reference's, src/render/integrator.cpp:276-339):
 *
 *   raygen -> [ trace_closest -> shade -> resolve ]* -> splat
 *
 *  - path state lives in HBM as five packed SoA arrays (4 x float4 + 1 x uint2 =
 *    72 B/path, every access a full 16 B/lane coalesced transaction) and is
 *    PHYSICALLY compacted by `shade`: live paths are written densely to the other
 *    buffer (wave ballot + mbcnt prefix + one atomic per wave), so later bounces
 *    read contiguous memory and no lane idles;
 *  - `trace_closest` / `resolve` walk the compressed 8-wide BVH with a per-lane
 *    traversal stack held in LDS (one 8-byte column per lane, conflict-free);
 *  - kernels take their element count from device memory (written by the previous
 *    kernel's compaction), so a whole chunk is enqueued without host round trips;
 *  - `splat` accumulates the 5x5 Gaussian footprints of a block in an LDS tile
 *    (ds_add_f32) and flushes each tile pixel with one global atomic.
 *
 * All path logic is in har_path.h (shared with the host test harness); these
 * kernels only move data.  gfx950 only: 64-lane waves are assumed throughout.
 */
#include "har_kernels.h"

namespace har {

static constexpr int kBlock = 256;

struct LdsStack {
    static constexpr int Capacity = HAR_LDS_STACK_DEPTH;
    uint2 *col;   /* &lds[threadIdx.x]; entry l lives at col[l * kBlock] */
    __device__ __forceinline__ void push(int l, uint32_t x, uint32_t y) { col[l * kBlock] = make_uint2(x, y); }
    __device__ __forceinline__ void pop(int l, uint32_t &x, uint32_t &y) { uint2 v = col[l * kBlock]; x = v.x; y = v.y; }
};

__device__ __forceinline__ uint32_t wave_rank(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t) (mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) mask, 0u));
}
/* reserve `popc(mask)` slots with ONE atomic per wave; returns this lane's slot */
__device__ __forceinline__ uint32_t wave_reserve(uint32_t *counter, bool pred) {
    uint64_t mask = __ballot(pred);
    uint32_t cnt = (uint32_t) __popcll(mask), base = 0;
    if (cnt == 0) return 0;
    uint32_t rank = wave_rank(mask);
    if (rank == 0 && pred) base = atomicAdd(counter, cnt);
    /* broadcast from the first active lane */
    uint32_t leader = (uint32_t) __ffsll((long long) mask) - 1u;
    base = __shfl(base, (int) leader, 64);
    return base + rank;
}

__device__ __forceinline__ void store_state(const WaveState &W, uint32_t i, const PathState &s) {
    W.a0[i] = make_float4(s.o.x, s.o.y, s.o.z, s.maxt);
    W.a1[i] = make_float4(s.d.x, s.d.y, s.d.z, s.prev_bsdf_pdf);
    W.a2[i] = make_float4(s.throughput.x, s.throughput.y, s.throughput.z, __uint_as_float(s.flags));
    W.a3[i] = make_float4(s.prev_p.x, s.prev_p.y, s.prev_p.z, __uint_as_float(s.lane));
    W.a4[i] = make_uint2((uint32_t) s.rng, (uint32_t) (s.rng >> 32));
}
__device__ __forceinline__ PathState load_state(const WaveState &W, uint32_t i) {
    float4 a0 = W.a0[i], a1 = W.a1[i], a2 = W.a2[i], a3 = W.a3[i]; uint2 a4 = W.a4[i];
    PathState s;
    s.o = Vec3(a0.x, a0.y, a0.z); s.maxt = a0.w;
    s.d = Vec3(a1.x, a1.y, a1.z); s.prev_bsdf_pdf = a1.w;
    s.throughput = Vec3(a2.x, a2.y, a2.z); s.flags = __float_as_uint(a2.w);
    s.prev_p = Vec3(a3.x, a3.y, a3.z); s.lane = __float_as_uint(a3.w);
    s.rng = (uint64_t) a4.x | ((uint64_t) a4.y << 32);
    return s;
}

/* ------------------------------------------------------------------ raygen */
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_raygen(DSensor C, uint32_t seed, uint32_t spp, uint32_t log_spp, uint32_t lane_base,
                                                   uint32_t n, WaveState out, float4 *result, uint32_t *count,
                                                   const float *adj, float4 *dL) {
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i == 0) *count = n;
    if (i >= n) return;
    LaneSample ls;
    PathState st = raygen_lane(C, seed, spp, log_spp, lane_base + i, ls);
    store_state(out, i, st);
    if (MODE != MODE_PRB_ADJOINT) result[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MODE == MODE_PRB_ADJOINT) {
        /* adjoint of ImageBlock::put + develop (common.py:696-746): gather grad_in / W over the footprint */
        Footprint F; film_footprint(C, ls, F);
        Vec3 g(0.f);
        for (uint32_t ys = 0; ys < F.count; ++ys) {
            uint32_t y = F.y0 + ys; if (!(y < C.crop_h)) continue;
            for (uint32_t xs = 0; xs < F.count; ++xs) {
                uint32_t x = F.x0 + xs; if (!(x < C.crop_w)) continue;
                float w = F.wx[xs] * F.wy[ys];
                const float *a = adj + 3 * ((size_t) y * C.crop_w + x);
                g = Vec3(fma_(a[0], w, g.x), fma_(a[1], w, g.y), fma_(a[2], w, g.z));
            }
        }
        dL[i] = make_float4(g.x, g.y, g.z, 0.f);
    }
}

/* ----------------------------------------------------------- trace_closest */
__global__ __launch_bounds__(kBlock) void k_trace_closest(Accel A, const uint32_t *count, const float4 *a0, const float4 *a1,
                                                          float4 *h0, uint2 *h1, int *status) {
    __shared__ uint2 lds[HAR_LDS_STACK_DEPTH * kBlock];
    LdsStack stack{ lds + threadIdx.x };
    const uint32_t n = *count;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        float4 o = a0[i], d = a1[i];
        Hit hit; int st = 0;
        accel_trace<false>(A, Vec3(o.x, o.y, o.z), Vec3(d.x, d.y, d.z), o.w, hit, stack, st);
        if (st) atomicMax(status, st);
        h0[i] = make_float4(hit.t, hit.u, hit.v, __uint_as_float(hit.prim));
        h1[i] = make_uint2(hit.shape, hit.inst);
    }
}

/* ------------------------------------------------------------------- shade */
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_shade(DScene S, ShadeParams P, uint32_t lane_base, const uint32_t *count_in, WaveState in,
                                                  const float4 *h0, const uint2 *h1, WaveState out, uint32_t *count_out,
                                                  ItemArrays items, uint32_t *item_count, float4 *result) {
    const uint32_t n = *count_in;
    for (uint32_t base = blockIdx.x * kBlock; base < n; base += gridDim.x * kBlock) {
        const uint32_t i = base + threadIdx.x;
        const bool in_range = i < n;
        ShadeResult R; R.alive = false; R.item = false; R.add_emission = false;
        uint32_t lane = 0;
        if (in_range) {
            PathState st = load_state(in, i);
            float4 hh = h0[i]; uint2 hs = h1[i];
            Hit hit; hit.t = hh.x; hit.u = hh.y; hit.v = hh.z; hit.prim = __float_as_uint(hh.w); hit.shape = hs.x; hit.inst = hs.y;
            shade_lane<MODE>(S, P, st, hit, R);
            lane = st.lane - lane_base;
            if (R.add_emission) {
                float4 r = result[lane];
                if (MODE == MODE_PATH)            r = make_float4(fma_(R.em_a.x, R.em_b.x, r.x), fma_(R.em_a.y, R.em_b.y, r.y), fma_(R.em_a.z, R.em_b.z, r.z), 0.f);
                else if (MODE == MODE_PRB_PRIMAL) r = make_float4(r.x + R.em_b.x, r.y + R.em_b.y, r.z + R.em_b.z, 0.f);
                else                              r = make_float4(r.x - R.em_b.x, r.y - R.em_b.y, r.z - R.em_b.z, 0.f);
                result[lane] = r;
            }
        }
        const bool alive = in_range && R.alive;
        uint32_t slot = wave_reserve(count_out, alive);
        if (alive) store_state(out, slot, R.next);
        const bool item = in_range && R.item;
        uint32_t islot = wave_reserve(item_count, item);
        if (item) {
            items.s0[islot] = make_float4(R.sh_o.x, R.sh_o.y, R.sh_o.z, R.item_ray ? R.sh_maxt : -1.f);
            items.s1[islot] = make_float4(R.sh_d.x, R.sh_d.y, R.sh_d.z, __uint_as_float(lane));
            items.s2[islot] = make_float4(R.contrib.x, R.contrib.y, R.contrib.z, __uint_as_float(MODE == MODE_PRB_ADJOINT ? (R.bsdf | (R.ind_active ? 0x80000000u : 0u)) : 0u));
            if (MODE == MODE_PRB_ADJOINT) {
                items.s3[islot] = make_float4(R.dLr_drho.x, R.dLr_drho.y, R.dLr_drho.z, R.uv_x);
                items.s4[islot] = make_float4(R.refl.x, R.refl.y, R.refl.z, R.uv_y);
            }
        }
    }
}

/* ------------------------------------------------- resolve (shadow rays + NEE) */
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_resolve(DScene S, const uint32_t *item_count, ItemArrays items, float4 *result,
                                                    const float4 *dL, float *grad_refl, float *const *grad_tex, int *status) {
    __shared__ uint2 lds[HAR_LDS_STACK_DEPTH * kBlock];
    LdsStack stack{ lds + threadIdx.x };
    const uint32_t n = *item_count;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        float4 s0 = items.s0[i], s1 = items.s1[i], s2 = items.s2[i];
        const uint32_t lane = __float_as_uint(s1.w);
        bool visible = false;
        if (s0.w >= 0.f) {
            Hit hit; int st = 0;
            visible = !accel_trace<true>(S.accel, Vec3(s0.x, s0.y, s0.z), Vec3(s1.x, s1.y, s1.z), s0.w, hit, stack, st);
            if (st) { atomicMax(status, st); visible = false; }
        }
        if (MODE == MODE_PATH || MODE == MODE_PRB_PRIMAL) {
            if (visible) { float4 r = result[lane]; result[lane] = make_float4(r.x + s2.x, r.y + s2.y, r.z + s2.z, 0.f); }
        } else {
            /* L <- L - Lr_dir; g = dL * (dLr_dir/drho + [bsdf_val != 0] L / rho)  (prb.py:227,288-313) */
            float4 L = result[lane];
            if (visible) { L = make_float4(L.x - s2.x, L.y - s2.y, L.z - s2.z, 0.f); result[lane] = L; }
            float4 s3 = items.s3[i], s4 = items.s4[i], dl = dL[lane];
            const uint32_t tag = __float_as_uint(s2.w), bsdf = tag & 0x7fffffffu;
            Vec3 g = visible ? Vec3(s3.x, s3.y, s3.z) : Vec3(0.f);
            if (tag & 0x80000000u)
                g = g + Vec3(s4.x != 0.f ? L.x / s4.x : 0.f, s4.y != 0.f ? L.y / s4.y : 0.f, s4.z != 0.f ? L.z / s4.z : 0.f);
            g = g * Vec3(dl.x, dl.y, dl.z);
            if (g.x != 0.f || g.y != 0.f || g.z != 0.f) {
                const DBsdf B = S.bsdfs[bsdf];
                if (B.texture < 0) {
                    float *dst = grad_refl + 3 * (size_t) bsdf;
                    atomicAdd(dst, g.x); atomicAdd(dst + 1, g.y); atomicAdd(dst + 2, g.z);
                } else {
                    TexTaps taps; tex_taps(S.textures[B.texture], s3.w, s4.w, taps);
                    float *dst = grad_tex[B.texture];
                    const float w[4] = { taps.w0x * taps.w0y, taps.w1x * taps.w0y, taps.w0x * taps.w1y, taps.w1x * taps.w1y };
                    for (int k = 0; k < 4; ++k) {
                        float *q = dst + 3 * (size_t) taps.idx[k];
                        atomicAdd(q, g.x * w[k]); atomicAdd(q + 1, g.y * w[k]); atomicAdd(q + 2, g.z * w[k]);
                    }
                }
            }
        }
    }
}

/* ------------------------------------------------------------------- splat */
/* ImageBlock::put (imageblock.cpp:444-540) for 256 consecutive lanes, LDS tile + one global atomic per tile pixel */
__global__ __launch_bounds__(kBlock) void k_splat(DSensor C, uint32_t seed, uint32_t spp, uint32_t log_spp, uint32_t lane_base, uint32_t n,
                                                  const float4 *result, int weights_only, float *film) {
    __shared__ float tile[HAR_SPLAT_TILE_FLOATS];
    __shared__ int tx0, ty0, tx1, ty1;
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    const bool act = i < n;
    if (threadIdx.x == 0) { tx0 = 0x7fffffff; ty0 = 0x7fffffff; tx1 = -0x7fffffff; ty1 = -0x7fffffff; }
    __syncthreads();
    Footprint F; F.count = 0; F.x0 = 0; F.y0 = 0;
    float val[4] = { 0.f, 0.f, 0.f, 1.f };
    if (act) {
        LaneSample ls = lane_film_pos(C, seed, spp, log_spp, lane_base + i);
        film_footprint(C, ls, F);
        if (!weights_only) { float4 r = result[i]; val[0] = r.x; val[1] = r.y; val[2] = r.z; }
        atomicMin(&tx0, (int) F.x0); atomicMin(&ty0, (int) F.y0);
        atomicMax(&tx1, (int) F.x0 + (int) F.count); atomicMax(&ty1, (int) F.y0 + (int) F.count);
    }
    __syncthreads();
    const int ox = tx0, oy = ty0, tw = tx1 - tx0, th = ty1 - ty0;
    const bool use_tile = tw > 0 && th > 0 && (size_t) tw * th * 4 <= HAR_SPLAT_TILE_FLOATS;
    if (use_tile) {
        for (int k = threadIdx.x; k < tw * th * 4; k += kBlock) tile[k] = 0.f;
        __syncthreads();
        if (act)
            for (uint32_t ys = 0; ys < F.count; ++ys)
                for (uint32_t xs = 0; xs < F.count; ++xs) {
                    float w = F.wx[xs] * F.wy[ys];
                    float *p = tile + 4 * (((int) F.y0 + (int) ys - oy) * tw + ((int) F.x0 + (int) xs - ox));
                    if (!weights_only) { atomicAdd(p, val[0] * w); atomicAdd(p + 1, val[1] * w); atomicAdd(p + 2, val[2] * w); }
                    atomicAdd(p + 3, val[3] * w);
                }
        __syncthreads();
        for (int k = threadIdx.x; k < tw * th * 4; k += kBlock) {
            float v = tile[k];
            int px = k >> 2, x = ox + px % tw, y = oy + px / tw;
            if (v != 0.f && (uint32_t) x < C.crop_w && (uint32_t) y < C.crop_h)
                atomicAdd(film + 4 * ((size_t) y * C.crop_w + x) + (k & 3), v);
        }
    } else if (act) {
        for (uint32_t ys = 0; ys < F.count; ++ys)
            for (uint32_t xs = 0; xs < F.count; ++xs) {
                uint32_t x = F.x0 + xs, y = F.y0 + ys;
                if (x < C.crop_w && y < C.crop_h) {
                    float w = F.wx[xs] * F.wy[ys];
                    float *p = film + 4 * ((size_t) y * C.crop_w + x);
                    if (!weights_only) { atomicAdd(p, val[0] * w); atomicAdd(p + 1, val[1] * w); atomicAdd(p + 2, val[2] * w); }
                    atomicAdd(p + 3, val[3] * w);
                }
            }
    }
}

/* --------------------------------------------------------- small utilities */
__global__ void k_develop(const float *film, uint32_t npx, float *image) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npx) return;
    float4 f = reinterpret_cast<const float4 *>(film)[i];
    float w = f.w == 0.f ? 1.f : f.w;
    image[3 * (size_t) i] = f.x / w; image[3 * (size_t) i + 1] = f.y / w; image[3 * (size_t) i + 2] = f.z / w;
}
__global__ void k_adjoint_image(const float *grad_in, const float *wfilm, uint32_t npx, float *adj) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npx) return;
    float w = wfilm[4 * (size_t) i + 3]; float iw = w == 0.f ? 1.f : w;
    for (int c = 0; c < 3; ++c) adj[3 * (size_t) i + c] = grad_in[3 * (size_t) i + c] / iw;
}
__global__ void k_accumulate_stats(const uint32_t *counters, uint32_t n_bounces, unsigned long long *totals, uint32_t paths) {
    if (threadIdx.x || blockIdx.x) return;
    unsigned long long v = 0, s = 0;
    for (uint32_t b = 0; b < n_bounces; ++b) { v += counters[b]; s += counters[HAR_MAX_BOUNCE_SLOTS + b]; }
    totals[0] += paths; totals[1] += v; totals[2] += v; totals[3] += s;
}

/* --- array-valued plugin surface (Scene::ray_intersect*, Sampler, BSDF, Sensor, ImageBlock) --- */
template <bool NAIVE>
__global__ __launch_bounds__(kBlock) void k_api_intersect(DScene S, uint32_t n, const float *o, const float *d, const float *maxt,
                                                          float *t, float *u, float *v, uint32_t *prim, uint32_t *shape, uint32_t *inst, int *status) {
    __shared__ uint2 lds[HAR_LDS_STACK_DEPTH * kBlock];
    LdsStack stack{ lds + threadIdx.x };
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    Vec3 O(o[i], o[n + i], o[2 * (size_t) n + i]), D(d[i], d[n + i], d[2 * (size_t) n + i]);
    Hit hit; int st = 0;
    if (NAIVE) accel_trace_naive<false>(S.accel, S.blas_tri_ranges, O, D, maxt[i], hit);
    else accel_trace<false>(S.accel, O, D, maxt[i], hit, stack, st);
    if (st) atomicMax(status, st);
    t[i] = hit.t; u[i] = hit.u; v[i] = hit.v; prim[i] = hit.prim; shape[i] = hit.shape; inst[i] = hit.inst;
}
template <bool NAIVE>
__global__ __launch_bounds__(kBlock) void k_api_ray_test(DScene S, uint32_t n, const float *o, const float *d, const float *maxt, uint8_t *out, int *status) {
    __shared__ uint2 lds[HAR_LDS_STACK_DEPTH * kBlock];
    LdsStack stack{ lds + threadIdx.x };
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    Vec3 O(o[i], o[n + i], o[2 * (size_t) n + i]), D(d[i], d[n + i], d[2 * (size_t) n + i]);
    Hit hit; int st = 0; bool r;
    if (NAIVE) r = accel_trace_naive<true>(S.accel, S.blas_tri_ranges, O, D, maxt[i], hit);
    else r = accel_trace<true>(S.accel, O, D, maxt[i], hit, stack, st);
    if (st) atomicMax(status, st);
    out[i] = r ? 1 : 0;
}
__global__ void k_api_si(DScene S, uint32_t n, const float *o, const float *d, const float *t, const float *u, const float *v,
                         const uint32_t *prim, const uint32_t *shape, const uint32_t *inst, float *out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    (void) o;
    SurfInt si = compute_si(S, Vec3(d[i], d[n + i], d[2 * (size_t) n + i]), t[i], u[i], v[i], prim[i], shape[i], inst[i]);
    const Vec3 vs[6] = { si.p, si.n, si.sn, si.ss, si.st, si.wi };
    for (int k = 0; k < 6; ++k) { out[(3 * k) * (size_t) n + i] = vs[k].x; out[(3 * k + 1) * (size_t) n + i] = vs[k].y; out[(3 * k + 2) * (size_t) n + i] = vs[k].z; }
    out[18 * (size_t) n + i] = si.uv_x; out[19 * (size_t) n + i] = si.uv_y; out[20 * (size_t) n + i] = si.t;
}
__global__ void k_api_sampler_seed(uint32_t seed, uint32_t lane_offset, uint32_t n, uint64_t *state, uint64_t *inc) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t s, c; sampler_seed(seed, lane_offset + i, s, c); state[i] = s; inc[i] = c;
}
__global__ void k_api_sampler_next(uint32_t n, uint64_t *state, const uint64_t *inc, const uint8_t *active, float *out, int dims) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (active && !active[i]) { for (int k = 0; k < dims; ++k) out[k * (size_t) n + i] = 0.f; return; }
    uint64_t s = state[i];
    for (int k = 0; k < dims; ++k) out[k * (size_t) n + i] = pcg32_next_float(s, inc[i]);
    state[i] = s;
}
__global__ void k_api_bsdf_eval_pdf(DScene S, uint32_t bsdf, uint32_t n, const float *wi, const float *uv, const float *wo, float *value, float *pdf) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    TexTaps taps; Vec3 refl = bsdf_reflectance(S, S.bsdfs[bsdf], uv[i], uv[n + i], taps);
    Vec3 val; float p;
    diffuse_eval_pdf(refl, Vec3(wi[i], wi[n + i], wi[2 * (size_t) n + i]), Vec3(wo[i], wo[n + i], wo[2 * (size_t) n + i]), val, p);
    value[i] = val.x; value[n + i] = val.y; value[2 * (size_t) n + i] = val.z; pdf[i] = p;
}
__global__ void k_api_bsdf_sample(DScene S, uint32_t bsdf, uint32_t n, const float *wi, const float *uv, const float *s2, float *wo, float *pdf, float *weight) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    TexTaps taps; Vec3 refl = bsdf_reflectance(S, S.bsdfs[bsdf], uv[i], uv[n + i], taps);
    Vec3 w, wt; float p;
    diffuse_sample(refl, Vec3(wi[i], wi[n + i], wi[2 * (size_t) n + i]), s2[i], s2[n + i], w, p, wt);
    wo[i] = w.x; wo[n + i] = w.y; wo[2 * (size_t) n + i] = w.z; pdf[i] = p;
    weight[i] = wt.x; weight[n + i] = wt.y; weight[2 * (size_t) n + i] = wt.z;
}
__global__ void k_api_sensor_ray(DSensor C, uint32_t n, const float *px, const float *py, float *o, float *d, float *maxt) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Vec3 O, D; float mt; sensor_sample_ray(C, px[i], py[i], O, D, mt);
    o[i] = O.x; o[n + i] = O.y; o[2 * (size_t) n + i] = O.z; d[i] = D.x; d[n + i] = D.y; d[2 * (size_t) n + i] = D.z; maxt[i] = mt;
}
__global__ void k_api_film_put(DSensor C, uint32_t n, const float *px, const float *py, const float *values4, float *film) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    LaneSample L; L.pos_x = px[i]; L.pos_y = py[i]; L.ipos_x = px[i]; L.ipos_y = py[i];
    Footprint F; film_footprint(C, L, F);
    for (uint32_t ys = 0; ys < F.count; ++ys)
        for (uint32_t xs = 0; xs < F.count; ++xs) {
            uint32_t x = F.x0 + xs, y = F.y0 + ys;
            if (x < C.crop_w && y < C.crop_h) {
                float w = F.wx[xs] * F.wy[ys];
                for (int k = 0; k < 4; ++k) atomicAdd(film + 4 * ((size_t) y * C.crop_w + x) + k, values4[4 * (size_t) i + k] * w);
            }
        }
}

// ---------------------------------------------------------------------------
//  launch wrappers
// ---------------------------------------------------------------------------

static inline uint32_t blocks_for(uint32_t n) { return (n + kBlock - 1) / kBlock; }

void launch_raygen(int mode, hipStream_t s, const DSensor &C, uint32_t seed, uint32_t spp, uint32_t log_spp, uint32_t lane_base, uint32_t n,
                   const WaveState &out, float4 *result, uint32_t *count, const float *adj, float4 *dL) {
    dim3 g(blocks_for(n)), b(kBlock);
    if (mode == MODE_PRB_ADJOINT) hipLaunchKernelGGL(k_raygen<MODE_PRB_ADJOINT>, g, b, 0, s, C, seed, spp, log_spp, lane_base, n, out, result, count, adj, dL);
    else hipLaunchKernelGGL(k_raygen<MODE_PATH>, g, b, 0, s, C, seed, spp, log_spp, lane_base, n, out, result, count, adj, dL);
}
void launch_trace_closest(hipStream_t s, uint32_t grid, const Accel &A, const uint32_t *count, const WaveState &in, float4 *h0, uint2 *h1, int *status) {
    hipLaunchKernelGGL(k_trace_closest, dim3(grid), dim3(kBlock), 0, s, A, count, in.a0, in.a1, h0, h1, status);
}
void launch_shade(int mode, hipStream_t s, uint32_t grid, const DScene &S, const ShadeParams &P, uint32_t lane_base, const uint32_t *count_in,
                  const WaveState &in, const float4 *h0, const uint2 *h1, const WaveState &out, uint32_t *count_out, const ItemArrays &items,
                  uint32_t *item_count, float4 *result) {
    dim3 g(grid), b(kBlock);
    if (mode == MODE_PATH) hipLaunchKernelGGL(k_shade<MODE_PATH>, g, b, 0, s, S, P, lane_base, count_in, in, h0, h1, out, count_out, items, item_count, result);
    else if (mode == MODE_PRB_PRIMAL) hipLaunchKernelGGL(k_shade<MODE_PRB_PRIMAL>, g, b, 0, s, S, P, lane_base, count_in, in, h0, h1, out, count_out, items, item_count, result);
    else hipLaunchKernelGGL(k_shade<MODE_PRB_ADJOINT>, g, b, 0, s, S, P, lane_base, count_in, in, h0, h1, out, count_out, items, item_count, result);
}
void launch_resolve(int mode, hipStream_t s, uint32_t grid, const DScene &S, const uint32_t *item_count, const ItemArrays &items, float4 *result,
                    const float4 *dL, float *grad_refl, float *const *grad_tex, int *status) {
    dim3 g(grid), b(kBlock);
    if (mode == MODE_PRB_ADJOINT) hipLaunchKernelGGL(k_resolve<MODE_PRB_ADJOINT>, g, b, 0, s, S, item_count, items, result, dL, grad_refl, grad_tex, status);
    else hipLaunchKernelGGL(k_resolve<MODE_PATH>, g, b, 0, s, S, item_count, items, result, dL, grad_refl, grad_tex, status);
}
void launch_splat(hipStream_t s, const DSensor &C, uint32_t seed, uint32_t spp, uint32_t log_spp, uint32_t lane_base, uint32_t n,
                  const float4 *result, int weights_only, float *film) {
    hipLaunchKernelGGL(k_splat, dim3(blocks_for(n)), dim3(kBlock), 0, s, C, seed, spp, log_spp, lane_base, n, result, weights_only, film);
}
void launch_develop(hipStream_t s, const float *film, uint32_t npx, float *image) {
    hipLaunchKernelGGL(k_develop, dim3(blocks_for(npx)), dim3(kBlock), 0, s, film, npx, image);
}
void launch_adjoint_image(hipStream_t s, const float *grad_in, const float *wfilm, uint32_t npx, float *adj) {
    hipLaunchKernelGGL(k_adjoint_image, dim3(blocks_for(npx)), dim3(kBlock), 0, s, grad_in, wfilm, npx, adj);
}
void launch_accumulate_stats(hipStream_t s, const uint32_t *counters, uint32_t n_bounces, unsigned long long *totals, uint32_t paths) {
    hipLaunchKernelGGL(k_accumulate_stats, dim3(1), dim3(1), 0, s, counters, n_bounces, totals, paths);
}
void launch_api_intersect(hipStream_t s, const DScene &S, uint32_t n, const float *o, const float *d, const float *maxt, int naive,
                          float *t, float *u, float *v, uint32_t *prim, uint32_t *shape, uint32_t *inst, int *status) {
    dim3 g(blocks_for(n)), b(kBlock);
    if (naive) hipLaunchKernelGGL(k_api_intersect<true>, g, b, 0, s, S, n, o, d, maxt, t, u, v, prim, shape, inst, status);
    else hipLaunchKernelGGL(k_api_intersect<false>, g, b, 0, s, S, n, o, d, maxt, t, u, v, prim, shape, inst, status);
}
void launch_api_ray_test(hipStream_t s, const DScene &S, uint32_t n, const float *o, const float *d, const float *maxt, int naive, uint8_t *out, int *status) {
    dim3 g(blocks_for(n)), b(kBlock);
    if (naive) hipLaunchKernelGGL(k_api_ray_test<true>, g, b, 0, s, S, n, o, d, maxt, out, status);
    else hipLaunchKernelGGL(k_api_ray_test<false>, g, b, 0, s, S, n, o, d, maxt, out, status);
}
void launch_api_si(hipStream_t s, const DScene &S, uint32_t n, const float *o, const float *d, const float *t, const float *u, const float *v,
                   const uint32_t *prim, const uint32_t *shape, const uint32_t *inst, float *out) {
    hipLaunchKernelGGL(k_api_si, dim3(blocks_for(n)), dim3(kBlock), 0, s, S, n, o, d, t, u, v, prim, shape, inst, out);
}
void launch_api_sampler_seed(hipStream_t s, uint32_t seed, uint32_t lane_offset, uint32_t n, uint64_t *state, uint64_t *inc) {
    hipLaunchKernelGGL(k_api_sampler_seed, dim3(blocks_for(n)), dim3(kBlock), 0, s, seed, lane_offset, n, state, inc);
}
void launch_api_sampler_next(hipStream_t s, uint32_t n, uint64_t *state, const uint64_t *inc, const uint8_t *active, float *out, int dims) {
    hipLaunchKernelGGL(k_api_sampler_next, dim3(blocks_for(n)), dim3(kBlock), 0, s, n, state, inc, active, out, dims);
}
void launch_api_bsdf_eval_pdf(hipStream_t s, const DScene &S, uint32_t bsdf, uint32_t n, const float *wi, const float *uv, const float *wo, float *value, float *pdf) {
    hipLaunchKernelGGL(k_api_bsdf_eval_pdf, dim3(blocks_for(n)), dim3(kBlock), 0, s, S, bsdf, n, wi, uv, wo, value, pdf);
}
void launch_api_bsdf_sample(hipStream_t s, const DScene &S, uint32_t bsdf, uint32_t n, const float *wi, const float *uv, const float *s2, float *wo, float *pdf, float *weight) {
    hipLaunchKernelGGL(k_api_bsdf_sample, dim3(blocks_for(n)), dim3(kBlock), 0, s, S, bsdf, n, wi, uv, s2, wo, pdf, weight);
}
void launch_api_sensor_ray(hipStream_t s, const DSensor &C, uint32_t n, const float *px, const float *py, float *o, float *d, float *maxt) {
    hipLaunchKernelGGL(k_api_sensor_ray, dim3(blocks_for(n)), dim3(kBlock), 0, s, C, n, px, py, o, d, maxt);
}
void launch_api_film_put(hipStream_t s, const DSensor &C, uint32_t n, const float *px, const float *py, const float *values4, float *film) {
    hipLaunchKernelGGL(k_api_film_put, dim3(blocks_for(n)), dim3(kBlock), 0, s, C, n, px, py, values4, film);
}

} // namespace har
