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
#include "har_shape_grad.h"

namespace har {

static constexpr int kBlock = 256;


template <int CAP> struct LdsStack {
    static constexpr int Capacity = CAP;
    static constexpr bool kSelectRefill = true;
    uint2 *col;   /* &lds[threadIdx.x]; entry l lives at col[l * kBlock] */
    __device__ __forceinline__ void push(int l, uint32_t x, uint32_t y) { col[l * kBlock] = make_uint2(x, y); }
    __device__ __forceinline__ void pop(int l, uint32_t &x, uint32_t &y) { uint2 v = col[l * kBlock]; x = v.x; y = v.y; }
};
/* first CAP entries in LDS, the next SPILL in a per-thread column of HBM (entry l of thread t at spill[(l - CAP) * stride + t]) */
template <int CAP, int SPILL> struct HybridStack {
    static constexpr int Capacity = CAP + SPILL;
    static constexpr bool kSelectRefill = false;        /* the branch-free refill lets LLVM fold push()'s two stores into one store through a selected (LDS-or-global) pointer, which its gfx950 back end cannot select */
    uint2 *col, *spill; uint32_t stride;
    __device__ __forceinline__ void push(int l, uint32_t x, uint32_t y) {
        if (l < CAP) col[l * kBlock] = make_uint2(x, y); else spill[(size_t) (l - CAP) * stride] = make_uint2(x, y);
    }
    __device__ __forceinline__ void pop(int l, uint32_t &x, uint32_t &y) {
        uint2 v; if (l < CAP) v = col[l * kBlock]; else v = spill[(size_t) (l - CAP) * stride];
        x = v.x; y = v.y;
    }
};
/* SPILL = false: scenes whose depth-first bound fits the LDS entries (no spill code in the loop) */
template <bool SPILL> struct WaveStackOf { typedef HybridStack<HAR_LDS_STACK_SMALL, HAR_STACK_SPILL> type; };
template <> struct WaveStackOf<false> { typedef LdsStack<HAR_LDS_STACK_SMALL> type; };
template <bool SPILL> __device__ __forceinline__ typename WaveStackOf<SPILL>::type make_wave_stack(uint2 *lds, uint2 *spill);
template <> __device__ __forceinline__ WaveStackOf<true>::type make_wave_stack<true>(uint2 *lds, uint2 *spill) {
    const uint32_t stride = gridDim.x * kBlock;
    return WaveStackOf<true>::type{ lds + threadIdx.x, spill + (size_t) blockIdx.x * kBlock + threadIdx.x, stride };
}
template <> __device__ __forceinline__ WaveStackOf<false>::type make_wave_stack<false>(uint2 *lds, uint2 *) { return WaveStackOf<false>::type{ lds + threadIdx.x }; }

__device__ __forceinline__ uint32_t wave_rank(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t) (mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) mask, 0u));
}

/*
 * XCD-private queues.  A chunk's paths are dealt tile-by-tile (256 lanes) to HAR_SHARDS = 8
 * queues; block b always works on shard b % 8 -- the XCD the dispatcher places it on -- and
 * compacts survivors back into the SAME shard, so a shard can never overflow its region and
 * its state stays in one XCD's L2.  Slot reservation costs ONE atomic per 256-thread block and
 * the eight counters live on separate cache lines: a single shared counter saturates at ~88
 * atomics/us on MI355X, which made one-atomic-per-wave compaction the top cost of `shade`.
 */
struct ShardLoop {
    uint32_t shard, n, base;
    __device__ __forceinline__ ShardLoop(const uint32_t *counts, uint32_t shard_cap) {
        shard = blockIdx.x & (HAR_SHARDS - 1); n = counts[shard * HAR_COUNTER_STRIDE]; base = shard * shard_cap;
    }
    __device__ __forceinline__ uint32_t first_tile() const { return blockIdx.x / HAR_SHARDS; }
    __device__ __forceinline__ uint32_t tile_step() const { return gridDim.x / HAR_SHARDS; }
};

/* reserve slots for two predicates at once (live paths, NEE items); 2 barriers per tile */
__device__ __forceinline__ void block_reserve2(uint32_t *cnt_a, bool pa, uint32_t *cnt_b, bool pb, uint32_t *lds /* [12] */,
                                               uint32_t &slot_a, uint32_t &slot_b) {
    const uint64_t ma = __ballot(pa), mb = __ballot(pb);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    if (lane == 0) { lds[wave] = (uint32_t) __popcll(ma); lds[4 + wave] = (uint32_t) __popcll(mb); }
    __syncthreads();
    if (threadIdx.x == 0)  { uint32_t t = lds[0] + lds[1] + lds[2] + lds[3]; lds[8] = t ? atomicAdd(cnt_a, t) : 0u; }
    if (threadIdx.x == 64) { uint32_t t = lds[4] + lds[5] + lds[6] + lds[7]; lds[9] = t ? atomicAdd(cnt_b, t) : 0u; }
    __syncthreads();
    uint32_t oa = lds[8], ob = lds[9];
    for (uint32_t w = 0; w < wave; ++w) { oa += lds[w]; ob += lds[4 + w]; }
    slot_a = oa + wave_rank(ma); slot_b = ob + wave_rank(mb);
    __syncthreads();
}

/* the same reservation per WAVE: one returning atomic per predicate and wave, no barrier.  HAR_WAVE_RESERVE = 1 (default): the generic (all-BSDF) shading kernels, whose
 * waves run material code of very different length between two reservations -- with block_reserve2 all four waves of a block wait at three barriers per round for the
 * slowest one and for the round trip of the block's atomic: generic k_shade 22.5 -> 21.8 ms per frame on materials1m, forward +1 % in five bracketed runs, prb +-0.
 * = 2: every shading kernel -- the diffuse kernels retire 4 - 5x the vertices per second and saturate the shard counters (8.2 -> 11.2 ms), as round 1 found.  = 0: block
 * reservation everywhere (rounds 1 - 4).  profiles/r05_pmc_generic_shade_materials1m.txt */
#ifndef HAR_WAVE_RESERVE
#define HAR_WAVE_RESERVE 1
#endif
__device__ __forceinline__ void wave_reserve2(uint32_t *cnt_a, bool pa, uint32_t *cnt_b, bool pb, uint32_t &slot_a, uint32_t &slot_b) {
    const uint64_t ma = __ballot(pa), mb = __ballot(pb);
    uint32_t ba = 0u, bb = 0u;
    if ((threadIdx.x & 63u) == 0u) {
        const uint32_t ta = (uint32_t) __popcll(ma), tb = (uint32_t) __popcll(mb);
        if (ta) ba = atomicAdd(cnt_a, ta);
        if (tb) bb = atomicAdd(cnt_b, tb);
    }
    ba = (uint32_t) __builtin_amdgcn_readfirstlane((int) ba); bb = (uint32_t) __builtin_amdgcn_readfirstlane((int) bb);
    slot_a = ba + wave_rank(ma); slot_b = bb + wave_rank(mb);
}

/* wave-wide float sum with DPP (no LDS): result valid in lane 63 */
__device__ __forceinline__ float wave_sum_to_last(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));  /* quad_perm [1,0,3,2] */
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));  /* quad_perm [2,3,0,1] */
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)); /* row_half_mirror */
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true)); /* row_mirror */
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false)); /* row_bcast:15 */
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false)); /* row_bcast:31 */
    return v;
}

/* wave-wide maximum of NON-NEGATIVE floats with DPP: result valid in lane 63 */
__device__ __forceinline__ float wave_max_to_last(float v) {
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false)));
    return v;
}
/* TexelQueues::gmax: the block's largest |gradient component| among the records it appends (float bits in LDS word `lds_max`; +inf for a non-finite one).
 * All lanes call; the caller publishes lds_max with one global atomicMax per block at the end. */
__device__ __forceinline__ void texel_record_track_max(uint32_t *lds_max, bool has, Vec3 g) {
    if (!__ballot(has)) return;
    float m = has ? fmaxf(fabsf(g.x), fmaxf(fabsf(g.y), fabsf(g.z))) : 0.f;
    if (has && !(m <= 3.4028234e38f)) m = __uint_as_float(0x7f800000u);        /* NaN or inf */
    m = wave_max_to_last(m);
    if ((threadIdx.x & 63u) == 63u && m > 0.f) atomicMax(lds_max, __float_as_uint(m));
}

/* gradient scatter with wave-level pre-reduction: lanes that target the same address are summed
 * (DPP) and committed by one lane, so a constant albedo costs 3 atomics per wave instead of 192 */
#ifndef HAR_TEXEL_ROUNDS
#define HAR_TEXEL_ROUNDS 6
#endif
__device__ __forceinline__ void wave_aggregated_add3(float *dst, Vec3 g, bool active) {
    for (int round = 0; round < HAR_TEXEL_ROUNDS; ++round) {
        uint64_t m = __ballot(active);
        if (m == 0) return;
        const int leader = __ffsll((long long) m) - 1;
        const uint64_t key = __shfl((unsigned long long) (uintptr_t) dst, leader, 64);
        const bool match = active && (uint64_t) (uintptr_t) dst == key;
        float sx = wave_sum_to_last(match ? g.x : 0.f), sy = wave_sum_to_last(match ? g.y : 0.f), sz = wave_sum_to_last(match ? g.z : 0.f);
        if ((threadIdx.x & 63u) == 63u) {
            float *q = (float *) (uintptr_t) key;
            atomicAdd(q, sx); atomicAdd(q + 1, sy); atomicAdd(q + 2, sz);
        }
        active = active && !match;
    }
    if (active) { atomicAdd(dst, g.x); atomicAdd(dst + 1, g.y); atomicAdd(dst + 2, g.z); }
}

/* the same for a per-block LDS accumulator indexed by a small slot number (BSDF / emitter records: 3 floats per slot).  The lanes of a wave mostly share ONE slot,
 * and 64 same-address ds_add_f32 serialise in the CU's single LDS pipeline (>= 64 cycles each, three per vertex); the DPP sum runs on the SIMD instead.  Must be
 * called by all lanes of the wave.  HAR_LDS_PREREDUCE=0: plain LDS atomics (A/B). */
#ifndef HAR_LDS_PREREDUCE
#define HAR_LDS_PREREDUCE 1
#endif
__device__ __forceinline__ void wave_slot_add3(float *acc, uint32_t slot, Vec3 g, bool active) {
#if HAR_LDS_PREREDUCE
    for (int round = 0; round < 3; ++round) {
        const uint64_t m = __ballot(active);
        if (m == 0) return;
        const uint32_t key = __shfl(slot, __ffsll((long long) m) - 1, 64);
        const bool match = active && slot == key;
        const float sx = wave_sum_to_last(match ? g.x : 0.f), sy = wave_sum_to_last(match ? g.y : 0.f), sz = wave_sum_to_last(match ? g.z : 0.f);
        if ((threadIdx.x & 63u) == 63u) { float *q = acc + 3u * key; atomicAdd(q, sx); atomicAdd(q + 1, sy); atomicAdd(q + 2, sz); }
        active = active && !match;
    }
#endif
    if (active) { float *q = acc + 3u * slot; atomicAdd(q, g.x); atomicAdd(q + 1, g.y); atomicAdd(q + 2, g.z); }
}

/* ... and for accumulators too many to keep one LDS slot each (the vertices of a mesh): a direct-mapped CACHE of `slots` (power of two) entries -- tags[slot] = the
 * key that owns the slot (0xffffffff: free), claimed with an integer compare-and-swap; a key that finds its slot taken by another goes to global memory.  The hot
 * keys of a block (the few hundred vertices in view) therefore meet in LDS whatever the size of the mesh.  acc2 / global2: an optional second accumulator under the
 * same keys (vertex normals); pass g2 = 0 / global2 = nullptr without.  All lanes call. */
__device__ __forceinline__ void wave_cached_add3(float *acc, float *acc2, uint32_t *tags, uint32_t slots, uint32_t key, Vec3 g, Vec3 g2, bool active, float *global, float *global2) {
    auto commit = [&](uint32_t k, Vec3 a, Vec3 b) {
        const uint32_t slot = k & (slots - 1u), old = atomicCAS(&tags[slot], 0xffffffffu, k);
        if (old == 0xffffffffu || old == k) {
            float *q = acc + 3u * slot; atomicAdd(q, a.x); atomicAdd(q + 1, a.y); atomicAdd(q + 2, a.z);
            if (acc2) { float *r = acc2 + 3u * slot; atomicAdd(r, b.x); atomicAdd(r + 1, b.y); atomicAdd(r + 2, b.z); }
        } else {
            float *q = global + 3 * (size_t) k; atomicAdd(q, a.x); atomicAdd(q + 1, a.y); atomicAdd(q + 2, a.z);
            if (global2) { float *r = global2 + 3 * (size_t) k; atomicAdd(r, b.x); atomicAdd(r + 1, b.y); atomicAdd(r + 2, b.z); }
        }
    };
    for (int round = 0; round < 4; ++round) {
        const uint64_t m = __ballot(active);
        if (m == 0) return;
        const uint32_t lead = __shfl(key, __ffsll((long long) m) - 1, 64);
        const bool match = active && key == lead;
        const float sx = wave_sum_to_last(match ? g.x : 0.f), sy = wave_sum_to_last(match ? g.y : 0.f), sz = wave_sum_to_last(match ? g.z : 0.f);
        float tx = 0.f, ty = 0.f, tz = 0.f;
        if (acc2) { tx = wave_sum_to_last(match ? g2.x : 0.f); ty = wave_sum_to_last(match ? g2.y : 0.f); tz = wave_sum_to_last(match ? g2.z : 0.f); }
        if ((threadIdx.x & 63u) == 63u) commit(lead, Vec3(sx, sy, sz), Vec3(tx, ty, tz));
        active = active && !match;
    }
    if (active) commit(key, g, g2);
}

/* rank of a lane among the block's lanes that append to the same queue: hist[q] += (lanes of this wave with queue q), one LDS atomic per wave and distinct queue
 * instead of one per lane (a wave's vertices mostly fall into one or two texture row bands, and same-address returning atomics serialise).  All lanes call. */
__device__ __forceinline__ uint32_t wave_ranked_count(uint32_t *hist, uint32_t q, bool active) {
    uint32_t rank = 0;
#if HAR_LDS_PREREDUCE
    const uint32_t lane = threadIdx.x & 63u;
    for (int round = 0; round < 4; ++round) {
        const uint64_t m = __ballot(active);
        if (m == 0) return rank;
        const int leader = __ffsll((long long) m) - 1;
        const uint32_t key = __shfl(q, leader, 64);
        const bool match = active && q == key;
        const uint64_t mm = __ballot(match);
        uint32_t base = 0;
        if (lane == (uint32_t) leader) base = atomicAdd(&hist[key], (uint32_t) __popcll(mm));
        base = __shfl(base, leader, 64);
        if (match) rank = base + (uint32_t) __popcll(mm & ((1ull << lane) - 1ull));
        active = active && !match;
    }
#endif
    if (active) rank = atomicAdd(&hist[q], 1u);
    return rank;
}

__device__ __forceinline__ void store_state(const WaveState &W, uint32_t i, const PathState &s) {
    /* a0.w: maxt of the camera ray (depth 0, eta == 1); bounce rays always have maxt = largest, so the slot carries
     * -eta instead (negative = "unbounded ray", decoded by the traversal kernels and load_state) */
    W.a0[i] = make_float4(s.o.x, s.o.y, s.o.z, (s.flags & 0xffffu) == 0u ? s.maxt : -s.eta);
    W.a1[i] = make_float4(s.d.x, s.d.y, s.d.z, s.prev_bsdf_pdf);
    W.a2[i] = make_float4(s.throughput.x, s.throughput.y, s.throughput.z, __uint_as_float(s.flags));
    W.a3[i] = make_float4(s.prev_p.x, s.prev_p.y, s.prev_p.z, __uint_as_float(s.lane));
    W.a4[i] = make_uint2((uint32_t) s.rng, (uint32_t) (s.rng >> 32));
}
/* ETA_ONE: kernels of scenes whose BSDFs are all `diffuse` (eta of every sample = 1, so the path's eta stays 1) do not read the ray's origin quarter of the state --
 * the shading code needs neither the origin nor maxt, only the accumulated eta that shares their 16 bytes */
template <bool ETA_ONE = false>
__device__ __forceinline__ PathState load_state(const WaveState &W, uint32_t i) {
    float4 a0 = ETA_ONE ? make_float4(0.f, 0.f, 0.f, -1.f) : W.a0[i], a1 = W.a1[i], a2 = W.a2[i], a3 = W.a3[i]; uint2 a4 = W.a4[i];
    PathState s;
    s.o = Vec3(a0.x, a0.y, a0.z); s.maxt = a0.w < 0.f ? HAR_LARGEST : a0.w; s.eta = a0.w < 0.f ? -a0.w : 1.f;
    s.d = Vec3(a1.x, a1.y, a1.z); s.prev_bsdf_pdf = a1.w;
    s.throughput = Vec3(a2.x, a2.y, a2.z); s.flags = __float_as_uint(a2.w);
    s.prev_p = Vec3(a3.x, a3.y, a3.z); s.lane = __float_as_uint(a3.w);
    s.rng = (uint64_t) a4.x | ((uint64_t) a4.y << 32);
    return s;
}

/* ------------------------------------------------------------------ raygen */
/* lane i of the chunk -> (shard, slot): tile t = i / 256 goes to shard t % 8, tile t / 8 of that shard */
__device__ __forceinline__ uint32_t shard_slot(uint32_t i, uint32_t shard_cap) {
    const uint32_t tile = i / kBlock;
    return (tile % HAR_SHARDS) * shard_cap + (tile / HAR_SHARDS) * kBlock + (i % kBlock);
}

template <int MODE, bool LITE = false>      /* LITE: only the ray (a0, a1) is stored -- the first shading kernel rebuilds the rest of the state (ShadeParams::sensor) */
__global__ __launch_bounds__(kBlock) void k_raygen(DSensor C, uint32_t seed, uint32_t spp, uint32_t log_spp, uint32_t lane_base,
                                                   uint32_t n, uint32_t shard_cap, WaveState out, float4 *result, uint32_t *count,
                                                   const float *adj, float4 *dL, PassState ps) {
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < HAR_SHARDS) {            /* lanes dealt to shard i */
        const uint32_t tiles = (n + kBlock - 1) / kBlock, rem = n % kBlock;
        uint32_t t = tiles > i ? (tiles - i + HAR_SHARDS - 1) / HAR_SHARDS : 0u;
        uint32_t c = t * kBlock;
        if (rem && tiles && (tiles - 1) % HAR_SHARDS == i) c -= kBlock - rem;
        count[i * HAR_COUNTER_STRIDE] = c;
    }
    if (i >= n) return;
    LaneSample ls;
    PathState st;
    if (ps.rng) {
        float j[2];
        st = raygen_lane(C, seed, spp, log_spp, lane_base + i, ls, ps.pass ? ps.rng + i : nullptr, j);
        ps.jitter[i] = make_float2(j[0], j[1]);
    } else st = raygen_lane(C, seed, spp, log_spp, lane_base + i, ls);
    if (LITE) { const uint32_t slot = shard_slot(i, shard_cap); out.a0[slot] = make_float4(st.o.x, st.o.y, st.o.z, st.maxt); out.a1[slot] = make_float4(st.d.x, st.d.y, st.d.z, st.prev_bsdf_pdf); }
    else store_state(out, shard_slot(i, shard_cap), st);
    if (MODE != MODE_PRB_ADJOINT) result[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MODE == MODE_PRB_ADJOINT || (MODE == MODE_PRB_PRIMAL && adj)) {      /* PRB_PRIMAL + adj: the primal pass of the record tape needs dL per lane for its emission terms */
        /* adjoint of ImageBlock::put + develop (common.py:696-746): gather grad_in / W over the footprint */
        if (!adj) { dL[i] = make_float4(0.f, 0.f, 0.f, 0.f); return; }      /* forward mode: dL accumulates the lane's differential radiance */
        Footprint F; film_footprint(C, ls, F);
        Vec3 g(0.f);
#pragma unroll
        for (uint32_t ys = 0; ys < HAR_MAX_FILTER_TAPS; ++ys) {        /* static tap indices keep the weights in registers (no scratch) */
            uint32_t y = F.y0 + ys; if (!(ys < F.count && y < C.crop_h)) continue;
#pragma unroll
            for (uint32_t xs = 0; xs < HAR_MAX_FILTER_TAPS; ++xs) {
                uint32_t x = F.x0 + xs; if (!(xs < F.count && x < C.crop_w)) continue;
                float w = F.wx[xs] * F.wy[ys];
                const float *a = adj + 3 * ((size_t) y * C.crop_w + x);
                g = Vec3(fma_(a[0], w, g.x), fma_(a[1], w, g.y), fma_(a[2], w, g.z));
            }
        }
        dL[i] = make_float4(g.x, g.y, g.z, 0.f);
    }
}

/* start of the adjoint pass in tape mode (TapeArrays): lane i of the chunk sits in slot shard_slot(i) of bounce 0's wavefront; its L is the primal pass's
 * result, its dL the adjoint image gathered over its film footprint (the adjoint of ImageBlock::put + develop, common.py:696-746 -- the same gather as
 * k_raygen<MODE_PRB_ADJOINT>, whose path state the tape already holds) */
__global__ __launch_bounds__(kBlock) void k_tape_begin(DSensor C, uint32_t seed, uint32_t spp, uint32_t log_spp, uint32_t lane_base, uint32_t n, uint32_t shard_cap,
                                                       const float4 *result, const float *adj, float4 *la, float2 *lb, float4 *dL_out, const float4 *dL_in) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    Vec3 g(0.f);
    if (dL_in) {            /* record tape, start of the commit pass: dL was gathered before the primal pass (dL_out call), only the slot order is new */
        const float4 d = dL_in[i]; g = Vec3(d.x, d.y, d.z);
        const float4 r = result[i];
        const uint32_t slot = shard_slot(i, shard_cap);
        la[slot] = make_float4(r.x, r.y, r.z, g.x); lb[slot] = make_float2(g.y, g.z);
        return;
    }
    const LaneSample ls = lane_film_pos(C, seed, spp, log_spp, lane_base + i);
    Footprint F; film_footprint(C, ls, F);
#pragma unroll
    for (uint32_t ys = 0; ys < HAR_MAX_FILTER_TAPS; ++ys) {
        uint32_t y = F.y0 + ys; if (!(ys < F.count && y < C.crop_h)) continue;
#pragma unroll
        for (uint32_t xs = 0; xs < HAR_MAX_FILTER_TAPS; ++xs) {
            uint32_t x = F.x0 + xs; if (!(xs < F.count && x < C.crop_w)) continue;
            float w = F.wx[xs] * F.wy[ys];
            const float *a = adj + 3 * ((size_t) y * C.crop_w + x);
            g = Vec3(fma_(a[0], w, g.x), fma_(a[1], w, g.y), fma_(a[2], w, g.z));
        }
    }
    if (dL_out) { dL_out[i] = make_float4(g.x, g.y, g.z, 0.f); return; }      /* record tape, before the primal pass: per-lane dL for the emission terms */
    const float4 r = result[i];
    const uint32_t slot = shard_slot(i, shard_cap);
    la[slot] = make_float4(r.x, r.y, r.z, g.x); lb[slot] = make_float2(g.y, g.z);
}

/* SamplingIntegrator::sample over caller-supplied rays (include/mitsuba/render/integrator.h:432-437): the wavefront starts from the rays
 * instead of the sensor.  Ray i is wavefront lane lane_base + i: its sampler is Sampler::seed's stream of that lane (sampler.cpp:129-148),
 * continued from state[i] when the caller passes its PCG32 states; the loop state is PathIntegrator::sample's initial one (path.cpp:129-147:
 * throughput 1, eta 1, depth 0, prev_bsdf_pdf 1, prev_bsdf_delta true). */
__global__ __launch_bounds__(kBlock) void k_raygen_rays(uint32_t seed, uint32_t lane_base, uint32_t n, uint32_t n_total, uint32_t first, const float *o, const float *d,
                                                        const float *maxt, const uint64_t *state, const uint8_t *active, uint32_t shard_cap, WaveState out, float4 *result, uint32_t *count) {
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < HAR_SHARDS) {
        const uint32_t tiles = (n + kBlock - 1) / kBlock, rem = n % kBlock;
        uint32_t t = tiles > i ? (tiles - i + HAR_SHARDS - 1) / HAR_SHARDS : 0u;
        uint32_t c = t * kBlock;
        if (rem && tiles && (tiles - 1) % HAR_SHARDS == i) c -= kBlock - rem;
        count[i * HAR_COUNTER_STRIDE] = c;
    }
    if (i >= n) return;
    const size_t g = (size_t) first + i;                   /* index into the caller's arrays (SoA, stride n_total) */
    PathState st;
    uint64_t inc;
    sampler_seed(seed, lane_base + i, st.rng, inc);
    if (state) st.rng = state[g];
    st.o = Vec3(o[g], o[n_total + g], o[2 * (size_t) n_total + g]); st.d = Vec3(d[g], d[n_total + g], d[2 * (size_t) n_total + g]); st.maxt = maxt[g];
    /* a masked lane keeps its slot in the dense wavefront but carries a ray of zero length: it ends at once, and k_sample_out discards whatever it gathered
     * and restores its sampler stream (the reference's masked lane never enters the loop) */
    if (active && !active[g]) st.maxt = 0.f;
    st.throughput = Vec3(1.f); st.lane = lane_base + i; st.prev_p = Vec3(0.f); st.prev_bsdf_pdf = 1.f; st.flags = 1u << 16; st.eta = 1.f;
    store_state(out, shard_slot(i, shard_cap), st);
    result[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}
/* radiance + mask of har_integrator_sample: PathIntegrator returns select(valid_ray, result, 0) (path.cpp:341-345), prb returns L and depth != 0 (prb.py:332).
 * A lane the caller masked out (`active`, the Mask argument of integrator.h:432-437) never enters the loop: zero radiance, valid = false, and its sampler stream
 * stays where it was (state_out = the state it came in with). */
__global__ void k_sample_out(uint32_t n, uint32_t n_total, uint32_t first, const float4 *result, const float *valid_lane, int zero_invalid, float *rgb, uint8_t *valid,
                             const uint8_t *active, uint32_t seed, uint32_t lane_base, const uint64_t *state_in, uint64_t *state_out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t g = (size_t) first + i;
    bool v = valid_lane[i] != 0.f;
    float4 r = result[i];
    if (zero_invalid && !v) r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active && !active[g]) {
        r = make_float4(0.f, 0.f, 0.f, 0.f); v = false;
        if (state_out) { uint64_t st, inc; sampler_seed(seed, lane_base + i, st, inc); state_out[g] = state_in ? state_in[g] : st; }
    }
    rgb[g] = r.x; rgb[n_total + g] = r.y; rgb[2 * (size_t) n_total + g] = r.z;
    if (valid) valid[g] = v ? 1 : 0;
}

/* ------------------------------------------------- persistent traversal loop */
/*
 * The traversal kernels are VALU-issue bound (rocprofv3: SQ_ACTIVE_INST_VALU ~ 73 % of SIMD cycles)
 * at ~30 % lane utilisation: incoherent rays need very different numbers of iterations (mean 13,
 * wave maximum 39 on the 1M-triangle scene) and the reference loop runs max-over-lanes triangle
 * tests per iteration.  Two changes fix that (tools/trace_stats.py models both on the CPU):
 *   - Traversal<HAR_TRAV_POLICY>::step() has a fixed shape (<= 1 leaf item, <= 1 node visit, <= 1 pop);
 *   - a wave is PERSISTENT: it draws rays in batches from its shard's cursor (one atomic per
 *     HAR_FETCH_BATCH rays) and refills idle lanes whenever >= HAR_REFILL_IDLE lanes are idle, so
 *     its lanes work on different rays at different stages ("persistent while-while", Aila & Laine,
 *     adapted to 64-lane waves).
 *   take(idx, T) -> bool : load work item idx into T (begin()); false = item has no ray to trace
 *   done(idx, T)         : called by the lane when its ray finishes (divergent; plain stores only)
 *   retire(pred, idx, T) : called by ALL lanes at refill time (wave-uniform; may use wave reductions),
 *                          pred marks lanes whose item finished since the last refill
 */
#ifndef HAR_FETCH_BATCH
#define HAR_FETCH_BATCH 128u
#endif
#ifndef HAR_FETCH_SMALL_FACTOR
#define HAR_FETCH_SMALL_FACTOR 16u    /* shards with at most this many rays per lane of their waves draw 64 rays per fetch instead of HAR_FETCH_BATCH: a shorter ragged end for the small
                                       * launches of late bounces and of a rank's band (1 / 4 / 16: 2 M-lane band 4.69 / 4.55 / 4.56 ms, 8 M 13.06 / 13.04 / 12.90, full frame 77.11 / 77.20 / 77.37) */
#endif
#ifndef HAR_REFILL_IDLE
#define HAR_REFILL_IDLE 12u
#endif
#ifndef HAR_EXTRA_ROUNDS
#define HAR_EXTRA_ROUNDS 0
#endif
#ifndef HAR_EXTRA_MIN
#define HAR_EXTRA_MIN 8u
#endif
#ifndef HAR_RECORD_MIN_WAVES
#define HAR_RECORD_MIN_WAVES 4    /* waves/SIMD the record flavour of the diffuse-only shading kernel (prb's primal pass) is bounded to: 110 - 112 VGPRs at 4 */
#endif
#ifndef HAR_SHADE_MIN_WAVES
#define HAR_SHADE_MIN_WAVES 4     /* __launch_bounds__ waves/SIMD of the shading kernels: caps the generic (all-BSDF) kernel at 128 VGPRs; measured 28.4 -> 24.7 ms on the materials scene */
#endif
#ifndef HAR_MATERIAL_SORT
#define HAR_MATERIAL_SORT 1
#endif
#ifndef HAR_SHAPE_MIN_WAVES
#define HAR_SHAPE_MIN_WAVES 2           /* k_shape_adjoint: 2 = 256 registers per lane + 848 B of scratch, 1 = 438 (accumulation registers as spill space) */
#endif
#ifndef HAR_SORT_WINDOW_MAX
#define HAR_SORT_WINDOW_MAX 8          /* tiles of 256 paths per material-sort window of the generic shading kernels (LDS: 1 KB per tile) */
#endif
#ifndef HAR_SORT_SCHEDULE
#define HAR_SORT_SCHEDULE 0         /* generic shading kernels: 1 = the chunks of a sorted window are shaded in the order of their position in the wavefront (k_shade) */
#endif
#ifndef HAR_DEFER_INST
#define HAR_DEFER_INST 4      /* persistent traversal: instance entries wait for this many lanes (0 = enter at once; host model tools/trace_stats.py HH_DEFER_INST: -3 %; measured 4 / 6 / 8: k_resolve 26.83 -> 26.27 / 26.31 / 26.43 ms, k_trace_closest +-0) */
#endif
#ifndef HAR_REFILL_SELECT
#define HAR_REFILL_SELECT 1     /* branch-free refill of the persistent traversal loop (trace_persistent); 0: the divergent `take` of rounds 1-4 */
#endif
#ifndef HAR_RESOLVE_RETIRE
#define HAR_RESOLVE_RETIRE 1
#endif
#ifndef HAR_TAPE_COMPACT
#define HAR_TAPE_COMPACT 1          /* record tape of diffuse-only scenes: 36-byte records {dLr_drho, tag | rho, u | v} instead of 48 (k_shade<.., RECORD>, k_commit<.., COMPACT>) */
#endif
#ifndef HAR_RESOLVE_LANE_S2
#define HAR_RESOLVE_LANE_S2 0       /* 1: the forward flavours repeat the lane in s2.w, so that k_resolve's commit reads contribution + destination with one 16-byte load (no second look at s1) */
#endif
#ifndef HAR_RESOLVE_ATOMIC
#define HAR_RESOLVE_ATOMIC 0        /* 1: forward k_resolve adds an unoccluded item's contribution with ONE 16-byte load + three no-return float atomics instead of two loads, an add and a
                                     * store.  Same bits.  Measured (round 6): +- 0 on the 1M-triangle scenes, whose shadow-ray kernel is bound by instruction issue -- and 5.3 -> 14.4 ms per
                                     * frame on the Cornell box, whose shadow rays cost next to nothing to trace: 500 M float atomics per frame run at the memory side's fixed rate (~57 G/s,
                                     * see TexelQueues).  Default 0. */
#endif
#ifndef HAR_TRAV_ORDER
#define HAR_TRAV_ORDER 0    /* measured: 0 (node, leaf, pop) 687, 2: 660, 1: 643 Mpaths/s on the 1M-tri scene */
#endif

template <bool ANY, bool RETIRE, typename WaveStack, bool FLAT, typename Take, typename Done, typename Retire>
__device__ __forceinline__ void trace_persistent(const Accel &A, uint32_t *cursor, uint32_t n, WaveStack &stack, int *status,
                                                 Take take, Done done, Retire retire) {
    const uint32_t lane = threadIdx.x & 63u;
    /* rays per fetch: 128 in the bulk; when the shard holds no more rays than its waves have lanes (late bounces, small jobs) every lane gets
     * ONE ray -- such launches are latency chains of single rays, and two rays per lane would double them */
    const uint32_t fetch = n <= (uint32_t) HAR_FETCH_SMALL_FACTOR * (gridDim.x / HAR_SHARDS) * (kBlock / 64u) * 64u ? 64u : (uint32_t) HAR_FETCH_BATCH;
    uint32_t pool_next = 0, pool_end = 0;      /* wave-uniform */
    bool exhausted = false;                    /* wave-uniform */
    bool busy = false, has_result = false;
    uint32_t idx = 0;
    Traversal<HAR_TRAV_POLICY, FLAT> T;
    T.found = false; T.hit.t = HAR_INF;
#if HAR_REFILL_SELECT && !HAR_EXTRA_ROUNDS
    /* BRANCH-FREE REFILL.  The same sequence of operations as the single loop below -- [refill when >= HAR_REFILL_IDLE lanes are idle] [step] ... -- written as two
     * nested loops (refill / step until enough lanes are idle), and in the refill EVERY lane builds a fresh traversal state (the idle ones for their new ray, the
     * others for a ray they drop again: a broadcast load) that is merged into the lane's state field by field with selects.  Written as a divergent
     * `if (idle) take(idx, T)` inside the stepping loop, the compiler keeps the loop-carried state in a SECOND register set around the refill and copies it over,
     * over again and back: 3 x 31 v_mov per refill, ~8 % of the instructions a wave issues (profiles/r04_isa_blocks.txt; 141 -> 63 VALU instructions per refill).
     * Once the shard has no rays left the wave only comes back here when all its lanes are done (the old loop ran `retire` in every step of that phase). */
    if constexpr (WaveStack::kSelectRefill) {
        for (;;) {
            const uint64_t idle = __ballot(!busy);
            const uint32_t n_idle = (uint32_t) __popcll(idle);
            if (RETIRE) { retire(!busy && has_result, idx, T); has_result = false; }
            if (pool_next == pool_end && !exhausted) {
                uint32_t b = 0;
                if (lane == 0) b = atomicAdd(cursor, fetch);
                b = (uint32_t) __builtin_amdgcn_readfirstlane((int) b);
                if (b >= n) exhausted = true;
                else { pool_next = b; pool_end = min(b + fetch, n); }
            }
            const uint32_t avail = pool_end - pool_next;
            if (avail == 0u && n_idle == 64u) break;
            const uint32_t rank = wave_rank(idle);
            const bool want = !busy && rank < avail;
            const uint32_t idx_new = min(pool_next + (want ? rank : 0u), n - 1u);      /* a lane that takes nothing still loads a ray that exists */
            Traversal<HAR_TRAV_POLICY, FLAT> Tn;
            const bool ok = take(idx_new, Tn);                                         /* take() is branch-free: it always begins Tn */
            if (RETIRE) Tn.found = !ok;                                                /* nothing to trace: retire next round */
            T.merge(want, Tn);
            idx = want ? idx_new : idx;
            busy = busy || (want && ok);
            if (RETIRE) has_result = has_result || (want && !ok);
            pool_next += min(n_idle, avail);
            const uint32_t refill_at = (exhausted && pool_next == pool_end) ? 64u : (uint32_t) HAR_REFILL_IDLE;
            uint32_t n_idle_now;
            do {
                bool allow_inst = true;
                if (ANY && HAR_DEFER_INST && !FLAT) {
                    const uint64_t m_inst = __ballot(busy && T.wants_instance_entry());
                    allow_inst = (uint32_t) __popcll(m_inst) >= (uint32_t) HAR_DEFER_INST || __ballot(busy) == m_inst;
                }
                if (busy) {
                    int st = 0;
                    if (T.template step<ANY, WaveStack, NoProbe, HAR_TRAV_ORDER>(A, stack, st, NoProbe(), allow_inst)) {
                        busy = false;
                        if (RETIRE) has_result = true; else done(idx, T);
                    }
                    if (st) atomicMax(status, st);
                }
                n_idle_now = (uint32_t) __popcll(__ballot(!busy));
            } while (n_idle_now < refill_at);
        }
        return;
    }
#endif
    for (;;) {
        const uint64_t idle = __ballot(!busy);
        const uint32_t n_idle = (uint32_t) __popcll(idle);
        if (n_idle >= HAR_REFILL_IDLE) {
            if (RETIRE) { retire(!busy && has_result, idx, T); has_result = false; }
            if (pool_next == pool_end && !exhausted) {
                uint32_t b = 0;
                if (lane == 0) b = atomicAdd(cursor, fetch);
                b = (uint32_t) __builtin_amdgcn_readfirstlane((int) b);
                if (b >= n) exhausted = true;
                else { pool_next = b; pool_end = min(b + fetch, n); }
            }
            const uint32_t avail = pool_end - pool_next;
            if (avail) {
                const uint32_t rank = wave_rank(idle);
                if (!busy && rank < avail) {
                    idx = pool_next + rank;
                    if (take(idx, T)) busy = true;
                    else if (RETIRE) { has_result = true; T.found = true; }     /* nothing to trace: retire next round */
                }
                pool_next += min(n_idle, avail);
            } else if (n_idle == 64u) {
                break;
            }
        }
        /* deferred instance entry (shadow rays only): lanes whose next leaf item enters an instance wait until HAR_DEFER_INST of them do (or no other lane can
         * use the step).  Closest-hit launches do not defer: on a rank's 8.4 M-lane band the waiting lengthens the last rays' chains (k_trace_closest 12.85 ->
         * 13.99 ms) while a full 67 M-lane frame gains nothing; k_resolve gains on both (7.67 -> 7.37 ms, 26.83 -> 26.27 ms) */
        bool allow_inst = true;
        if (ANY && HAR_DEFER_INST && !FLAT) {
            const uint64_t m_inst = __ballot(busy && T.wants_instance_entry());
            allow_inst = (uint32_t) __popcll(m_inst) >= (uint32_t) HAR_DEFER_INST || __ballot(busy) == m_inst;
        }
        if (busy) {
            int st = 0;
            if (T.template step<ANY, WaveStack, NoProbe, HAR_TRAV_ORDER>(A, stack, st, NoProbe(), allow_inst)) {
                busy = false;
                if (RETIRE) has_result = true; else done(idx, T);
            }
            if (st) atomicMax(status, st);
        }
        /* extra leaf rounds: lanes with a pending triangle / instance catch up while the others wait (cheap block),
         * so that more lanes take part in the next node visit (expensive block) */
        for (int r = 0; r < HAR_EXTRA_ROUNDS; ++r) {
            const bool want = busy && T.leaf_pending();
            if ((uint32_t) __popcll(__ballot(want)) < HAR_EXTRA_MIN) break;
            if (want) {
                int st = 0;
                if (T.template leaf_round<ANY>(A, stack, st)) {
                    busy = false;
                    if (RETIRE) has_result = true; else done(idx, T);
                }
                if (st) atomicMax(status, st);
            }
        }
    }
}

/* ----------------------------------------------------------- trace_closest */
#ifndef HAR_TRACE_MIN_WAVES
#define HAR_TRACE_MIN_WAVES 6     /* __launch_bounds__ waves per SIMD of the traversal kernels: the LDS stacks allow 6 blocks per CU, so the registers must too (<= 80; the
                                   * two-level k_trace_closest would take 82 on its own and lose its sixth wave: 871 vs 899 Mpaths/s, profiles/r03_ab_defer_xform.txt).  A/B: 7 with an 11-entry LDS stack */
#endif
/* LIST: the launch walks the rays of the 64-ray packets named in `list` (per shard: list[shard * list_stride + k] = first ray of the packet within the shard) --
 * the packets k_trace_packet gave up on -- instead of the whole wavefront: work item idx is ray list[idx / 64] + idx % 64, `count` holds the number of PACKETS */
template <bool SPILL, bool FLAT = false, bool LIST = false>
__global__ __launch_bounds__(kBlock, HAR_TRACE_MIN_WAVES) void k_trace_closest(Accel A, const uint32_t *count, uint32_t *cursor, uint32_t shard_cap, const float4 *a0,
                                                          const float4 *a1, float4 *h0, uint2 *h1, int *status, uint2 *spill, const uint32_t *list, uint32_t list_stride,
                                                          const uint32_t *shard_count) {
    __shared__ uint2 lds[HAR_LDS_STACK_SMALL * kBlock];
    typedef typename WaveStackOf<SPILL>::type WaveStack;
    typedef Traversal<HAR_TRAV_POLICY, FLAT> Trav;
    WaveStack stack = make_wave_stack<SPILL>(lds, spill);
    const uint32_t shard = blockIdx.x & (HAR_SHARDS - 1), base = shard * shard_cap;
    const uint32_t n = LIST ? 64u * count[shard * HAR_COUNTER_STRIDE] : count[shard * HAR_COUNTER_STRIDE];
    const uint32_t n_rays = LIST ? shard_count[shard * HAR_COUNTER_STRIDE] : n;      /* rays of the shard (the last packet may be partial) */
    if (n == 0) return;
    auto ray_of = [&](uint32_t idx) { return LIST ? list[(size_t) shard * list_stride + (idx >> 6)] + (idx & 63u) : idx; };
    auto take = [&](uint32_t idx, Trav &T) {
        /* branch-free (trace_persistent's refill merges the state with selects): a work item past the shard's last ray begins a copy of that ray and says `false` */
        const uint32_t r0 = ray_of(idx), r = LIST ? min(r0, n_rays - 1u) : r0;
        float4 o = a0[base + r], d = a1[base + r];
        T.begin(A, Vec3(o.x, o.y, o.z), Vec3(d.x, d.y, d.z), o.w < 0.f ? HAR_LARGEST : o.w, (A.top_last & 2u) != 0u);
        return !LIST || r0 < n_rays;
    };
    auto store = [&](uint32_t idx, const Trav &T) {
        const uint32_t r = ray_of(idx);
        if (LIST && r >= n_rays) return;
        h0[HIT0(base + r)] = make_float4(T.hit.t, T.hit.u, T.hit.v, __uint_as_float(T.hit.prim));
#if HAR_HIT_INTERLEAVED      /* the second half as ONE 16-byte store: the ray's 32-byte sector is written completely (no byte-masked partial write) */
#if HAR_HIT_MATINFO
        MeshInfo mi{ 0u, 0u };
        if (T.hit.t != HAR_INF) mi = A.mesh_info[T.hit.shape];
        *reinterpret_cast<uint4 *>(h1 + HIT1(base + r)) = make_uint4(T.hit.shape, T.hit.inst, mi.foff + T.hit.prim, mi.matinfo);
#else
        *reinterpret_cast<uint4 *>(h1 + HIT1(base + r)) = make_uint4(T.hit.shape, T.hit.inst, 0u, 0u);
#endif
#else
        h1[HIT1(base + r)] = make_uint2(T.hit.shape, T.hit.inst);
#endif
    };
#if HAR_CLOSEST_RETIRE
    /* hits are committed at refill time by all lanes that finished since the last refill (one store instruction for >= HAR_REFILL_IDLE lanes)
     * instead of by each lane in the iteration it finishes in (a store instruction + address arithmetic issued for one or two lanes) */
    trace_persistent<false, true, WaveStack, FLAT>(A, cursor + shard * HAR_COUNTER_STRIDE, n, stack, status, take,
        [&](uint32_t, const Trav &) { },
        [&](bool pred, uint32_t idx, const Trav &T) { if (pred) store(idx, T); });
#else
    trace_persistent<false, false, WaveStack, FLAT>(A, cursor + shard * HAR_COUNTER_STRIDE, n, stack, status, take, store,
        [&](bool, uint32_t, const Trav &) { });
#endif
}

/* ------------------------------------------------------------- trace_packet */
/*
 * Wave-shared descent for COHERENT closest-hit launches: the camera rays of a render at >= 64 spp, where the 64 lanes of a wave are 64 samples of ONE pixel
 * (lane = pixel * spp + sample, integrator.cpp:322-334).  The wave walks the BVH once for all its rays: a child box is tested against the PACKET -- interval
 * arithmetic over the rays' origins and reciprocal directions, one child per lane, ~70 instructions per node visit for the wave instead of the 246-instruction
 * per-lane test of eight boxes -- with one traversal stack per wave and wave-uniform control flow.  At the leaves every lane runs the exact Moeller-Trumbore
 * test of its own ray (tri_visit), with the tie rule of the per-lane kernels; the BVH only prunes, and the packet test only keeps MORE boxes than any single
 * ray's test would, so the hit records are those of k_trace_closest (and of the brute-force kernel) bit for bit.
 * A packet that turns out incoherent (a pixel that covers hundreds of triangles, a ray set whose directions straddle an axis and prune little) would cost more
 * than 64 independent walks: after `budget` steps -- or when the wave's stack is full -- the wave drops it and files it in `list`, which a
 * k_trace_closest<.., LIST> launch serves afterwards.  Host model: tools/packet_stats.py (camera rays of the 1M-triangle scene at 512^2: 13.7 node visits +
 * 11.5 leaf tests per packet, ~2 150 VALU instructions per 64 rays against ~6 200 for the per-lane kernel; shadow rays and later bounces are NOT coherent at the
 * scale of the geometry -- a first-bounce shadow packet would visit 9 400 nodes -- and stay with the per-lane kernels).
 */
#ifndef HAR_PACKET_STACK
#define HAR_PACKET_STACK 48        /* stack entries per wave (8 B each, LDS) */
#endif
template <typename F> __device__ __forceinline__ float wave_reduce_f32(float v, F op) {      /* all 64 lanes take part; the result comes back wave-uniform */
#define HAR_DPP_STEP(ctrl, rm) v = op(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), ctrl, rm, 0xF, false)))
    HAR_DPP_STEP(0xB1, 0xF); HAR_DPP_STEP(0x4E, 0xF); HAR_DPP_STEP(0x141, 0xF); HAR_DPP_STEP(0x140, 0xF); HAR_DPP_STEP(0x142, 0xA); HAR_DPP_STEP(0x143, 0xC);
#undef HAR_DPP_STEP
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_min_f32(float v) { return wave_reduce_f32(v, [](float a, float b) { return fminf(a, b); }); }
__device__ __forceinline__ float wave_max_f32(float v) { return wave_reduce_f32(v, [](float a, float b) { return fmaxf(a, b); }); }
/* what a packet knows about its rays: per axis, in the orientation of the rays' travel (an axis the rays run down is mirrored, so that every kept axis has
 * positive directions): on = farthest-back origin bound used for the NEAR plane, of = the one for the FAR plane, id_lo / id_hi the reciprocal directions' range.
 * mixed: the rays' directions have both signs on that axis -- no constraint from its slab. */
struct PacketBounds { float on[3], of[3], id_lo[3], id_hi[3]; uint32_t neg, mixed, octinv; };      /* neg / mixed: bit a = axis a */
__device__ __forceinline__ float uniform_f32(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }
__device__ __forceinline__ PacketBounds packet_bounds(const RaySetup &R) {
    PacketBounds B;
    const float o[3] = { R.o.x, R.o.y, R.o.z }, id[3] = { R.idir.x, R.idir.y, R.idir.z };
    B.octinv = 0u; B.neg = 0u; B.mixed = 0u;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float o_lo = wave_min_f32(o[a]), o_hi = wave_max_f32(o[a]), i_lo = wave_min_f32(id[a]), i_hi = wave_max_f32(id[a]);
        const bool neg = i_hi < 0.f, mixed = !(i_lo > 0.f || i_hi < 0.f);
        if (neg) B.neg |= 1u << a; else B.octinv |= 4u >> a;
        if (mixed) B.mixed |= 1u << a;
        /* mirrored axis: plane' = -plane, o' = -o, idir' = -idir; the values are wave-uniform and are kept in scalar registers */
        B.on[a] = uniform_f32(neg ? -o_lo : o_hi); B.of[a] = uniform_f32(neg ? -o_hi : o_lo);
        B.id_lo[a] = uniform_f32(neg ? -i_hi : i_lo); B.id_hi[a] = uniform_f32(neg ? -i_lo : i_hi);
    }
    return B;
}
/* the packet's test of the eight children of node `index` (wave-uniform): lane l tests child l % 8; returns the node group / triangle group of node_visit */
__device__ __forceinline__ void packet_node_visit(const Accel &A, const PacketBounds &B, float tmax_wave, uint32_t index, uint32_t &ng_x, uint32_t &ng_y, uint32_t &tg_x, uint32_t &tg_y) {
    const uint32_t *np = reinterpret_cast<const uint32_t *>(A.nodes + index);
    const uint4 n0 = reinterpret_cast<const uint4 *>(np)[0];
    const uint2 n1 = reinterpret_cast<const uint2 *>(np)[2];
    const uint32_t c = threadIdx.x & 7u;
    const uint8_t *nb = reinterpret_cast<const uint8_t *>(np);
    const uint32_t imask = n0.w >> 24, lmask = nb[24];            /* the node's inner / leaf child slots (Node8) */
    const float q[6] = { (float) nb[32 + c], (float) nb[40 + c], (float) nb[48 + c], (float) nb[56 + c], (float) nb[64 + c], (float) nb[72 + c] };      /* qlo x y z, qhi x y z */
    const float p[3] = { __uint_as_float(n0.x), __uint_as_float(n0.y), __uint_as_float(n0.z) };
    const float sc[3] = { __uint_as_float((n0.w & 0xffu) << 23), __uint_as_float(((n0.w >> 8) & 0xffu) << 23), __uint_as_float(((n0.w >> 16) & 0xffu) << 23) };
    float lb = 0.f, ub = tmax_wave;
    /* the child planes are rebuilt in the node's coordinates (one rounding of ulp(|plane|) / 2 each) before the rays' origin bounds are subtracted; the per-ray test
     * (node_visit) keeps (origin - o) and q * scale apart.  The difference stays far inside the padding every leaf box carries (pad_box: 2e-5 * |coordinate|, ~170 ulps,
     * inherited by every ancestor through the union), so this test prunes at most what the per-ray test prunes: tests/test_gpu_packet.py at coordinates of 1e4 */
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float lo = fma_(q[a], sc[a], p[a]), hi = fma_(q[3 + a], sc[a], p[a]);
        const bool neg = (B.neg >> a) & 1u;
        const float pn = neg ? -hi : lo, pf = neg ? -lo : hi;                        /* near / far plane along the rays' travel */
        const float dn = pn - B.on[a], df = pf - B.of[a];
        float tn = dn * (dn >= 0.f ? B.id_lo[a] : B.id_hi[a]), tf = df * (df >= 0.f ? B.id_hi[a] : B.id_lo[a]);
        if ((B.mixed >> a) & 1u) { tn = -HAR_INF; tf = HAR_INF; }
        lb = fmaxf(lb, tn); ub = fminf(ub, tf);
    }
    /* this lane's child: an inner one goes to position slot ^ octinv of the node group (node_visit), a leaf to its slot of the leaf group, an empty slot nowhere */
    const uint32_t pos = ((imask >> c) & 1u) ? 1u << (24u + (c ^ B.octinv)) : ((lmask >> c) & 1u) ? 1u << c : 0u;
    uint32_t m = (lb <= ub * 1.000002f) ? pos : 0u;
    /* OR over the eight children = over each aligned group of eight lanes */
    m |= (uint32_t) __builtin_amdgcn_update_dpp((int) m, (int) m, 0xB1, 0xF, 0xF, false);
    m |= (uint32_t) __builtin_amdgcn_update_dpp((int) m, (int) m, 0x4E, 0xF, 0xF, false);
    m |= (uint32_t) __builtin_amdgcn_update_dpp((int) m, (int) m, 0x141, 0xF, 0xF, false);
    const uint32_t hitmask = (uint32_t) __builtin_amdgcn_readfirstlane((int) m);
    ng_x = n1.x; tg_x = n1.y;
    ng_y = (hitmask & 0xff000000u) | imask;
    tg_y = (hitmask & 0xffu) ? ((hitmask & 0xffu) | (lmask << 8)) : 0u;
}
#ifndef HAR_PACKET_MIN_WAVES
#define HAR_PACKET_MIN_WAVES 4      /* 86 registers, no scratch (8: 64 + 60 B of scratch; 4 / 6 / 8 waves measured within 0.5 % of each other, profiles/r04_ab_prb_commit.txt) */
#endif
template <bool FLAT>
__global__ __launch_bounds__(kBlock, HAR_PACKET_MIN_WAVES) void k_trace_packet(Accel A, const uint32_t *count, uint32_t *cursor, uint32_t shard_cap, const float4 *a0, const float4 *a1,
                                                         float4 *h0, uint2 *h1, uint32_t *list, uint32_t list_stride, uint32_t *list_count, uint32_t budget) {
    __shared__ uint2 lds[(kBlock / 64) * HAR_PACKET_STACK];
    const uint32_t lane = threadIdx.x & 63u;
    uint2 *stack = lds + (threadIdx.x >> 6) * HAR_PACKET_STACK;
    const uint32_t shard = blockIdx.x & (HAR_SHARDS - 1), n = count[shard * HAR_COUNTER_STRIDE], base = shard * shard_cap;
    if (n == 0) return;
    for (;;) {
        uint32_t pk = 0;
        if (lane == 0) pk = atomicAdd(cursor + shard * HAR_COUNTER_STRIDE, 64u);
        pk = (uint32_t) __builtin_amdgcn_readfirstlane((int) pk);
        if (pk >= n) break;
        const uint32_t idx = min(pk + lane, n - 1u);                    /* the lanes past the end of the last packet walk a copy of its last ray; their result is dropped */
        const float4 fo = a0[base + idx], fd = a1[base + idx];
        const Vec3 o_w(fo.x, fo.y, fo.z), d_w(fd.x, fd.y, fd.z);
        float tmax = fo.w < 0.f ? HAR_LARGEST : fo.w + 0.f;
        Hit hit; hit.t = HAR_INF; hit.u = 0.f; hit.v = 0.f; hit.prim = 0; hit.shape = 0; hit.inst = 0xffffffffu;
        RaySetup R = ray_setup(o_w, d_w);
        const PacketBounds Bw = packet_bounds(R);
        PacketBounds B = Bw;
        float tmax_wave = wave_max_f32(tmax);
        uint32_t cur_inst = 0xffffffffu, steps = 0;
        bool in_tlas = false;
        /* the levels of a two-level scene in the per-lane kernels' order (Traversal::begin): phase 0 / 1; FLAT: one phase over A.root.
         * Control flow is wave-uniform throughout (every condition below is computed from scalars); it is written without `break` / `continue` out of nested
         * conditionals so that the structurised loops keep all 64 lanes together.  state: 0 walking, 1 this phase is done, 2 given up */
        const bool top_last = !FLAT && (A.top_last & 2u) != 0u, has_top = !FLAT && A.has_tlas && A.top_root != HAR_NO_NODE;
        const int n_phases = (!FLAT && A.has_tlas) ? 2 : 1;
        int state = 0;
        for (int phase = 0; phase < n_phases; ++phase) {
            uint32_t ng_x = A.root, ng_y = 0x80000000u, tg_x = 0u, tg_y = 0u;
            bool skip = state == 2;
            if (FLAT || !A.has_tlas) in_tlas = false;
            else if ((phase == 0) != top_last) { ng_x = A.top_root; in_tlas = false; skip = skip || !has_top; }      /* the top-level meshes' BLAS, world space */
            else in_tlas = true;
            int sp = 0, inst_sp = -1;
            state = skip ? state : 0;
            /* the reference loop of har_accel.h (accel_trace) with wave-uniform state: visit the nearest pending child, run ALL leaf items of the visited node, pop */
            while (!skip && state == 0) {
                ++steps;
                if (steps > budget) state = 2;
                else if (ng_y > 0x00ffffffu) {
                    uint32_t px = ng_x, py = ng_y;
                    const uint32_t child = ng_next_child(px, py, B.octinv);
                    if (py > 0x00ffffffu) {
                        if (sp >= HAR_PACKET_STACK) state = 2;
                        else { if (lane == 0) stack[sp] = make_uint2(px, py); ++sp; }
                    }
                    if (state == 0) packet_node_visit(A, B, tmax_wave, child, ng_x, ng_y, tg_x, tg_y);
                } else { tg_x = ng_x; tg_y = ng_y; ng_x = 0u; ng_y = 0u; }
                bool entered = false;
                while (state == 0 && !entered && tg_y != 0u) {
                    const uint32_t leaf = tg_next_leaf(tg_x, tg_y);
                    ++steps;
                    if (!FLAT && in_tlas) {
                        /* instance entry: the rest of the TLAS node waits on the stack, every lane takes its ray to object space (as Traversal::apply_pending does),
                         * the packet's bounds are rebuilt there; the instance is left when the stack is back at this depth */
                        if (sp + 2 > HAR_PACKET_STACK) state = 2;
                        else {
                            if (ng_y > 0x00ffffffu) { if (lane == 0) stack[sp] = make_uint2(ng_x, ng_y); ++sp; }
                            if (tg_y != 0u) { if (lane == 0) stack[sp] = make_uint2(tg_x, tg_y); ++sp; }
                            const InstRec &I = A.insts[leaf];
                            inst_sp = sp; cur_inst = I.inst_index; in_tlas = false;
                            if (!I.identity) { R = ray_setup(xf_point(I.to_object, o_w), xf_vector(I.to_object, d_w)); B = packet_bounds(R); }
                            ng_x = I.blas_root; ng_y = 0x80000000u; tg_y = 0u;
                            entered = true;
                        }
                    } else {
                        const float t_old = tmax;
                        tri_visit<false>(A, R, tmax, leaf, FLAT ? 0xffffffffu : cur_inst, hit);
                        if (__ballot(tmax != t_old)) tmax_wave = wave_max_f32(tmax);
                    }
                }
                if (state == 0 && ng_y <= 0x00ffffffu) {
                    if (!FLAT && !in_tlas && sp == inst_sp) { R = ray_setup(o_w, d_w); B = Bw; cur_inst = 0xffffffffu; in_tlas = true; inst_sp = -1; }      /* the instance is done: world space again */
                    if (sp == 0) state = 1;
                    else {
                        --sp;
                        const uint2 e = stack[sp];
                        ng_x = (uint32_t) __builtin_amdgcn_readfirstlane((int) e.x); ng_y = (uint32_t) __builtin_amdgcn_readfirstlane((int) e.y);
                    }
                }
            }
        }
        if (state == 2) {
            if (lane == 0) { const uint32_t k = atomicAdd(list_count + shard * HAR_COUNTER_STRIDE, 1u); list[(size_t) shard * list_stride + k] = pk; }
        } else if (pk + lane < n) {
            h0[HIT0(base + idx)] = make_float4(hit.t, hit.u, hit.v, __uint_as_float(hit.prim));
#if HAR_HIT_INTERLEAVED
#if HAR_HIT_MATINFO
            MeshInfo mi{ 0u, 0u };
            if (hit.t != HAR_INF) mi = A.mesh_info[hit.shape];
            *reinterpret_cast<uint4 *>(h1 + HIT1(base + idx)) = make_uint4(hit.shape, hit.inst, mi.foff + hit.prim, mi.matinfo);
#else
            *reinterpret_cast<uint4 *>(h1 + HIT1(base + idx)) = make_uint4(hit.shape, hit.inst, 0u, 0u);
#endif
#else
            h1[HIT1(base + idx)] = make_uint2(hit.shape, hit.inst);
#endif
        }
        /* ONE latch for this loop.  Without a convergent operation here LLVM threads the `state == 2 && lane != 0` path straight to the loop header, which makes two
         * back edges = a nested loop in which lanes 1..63 spin on packet 0 (their copy of `pk`) while lane 0 is parked at the list append: the kernel never ends
         * (seen on the GPU: every launch with a failing packet hung).  A wave barrier may not be duplicated into the two paths, so all lanes meet here. */
        __builtin_amdgcn_wave_barrier();
    }
}

/* adjoint of one NEE / vertex item (wave-uniform call: every lane takes part in the texel pre-reduction):
 * L <- L - Lr_dir; g = dL * (dLr_dir/dslot0 + L * (df/dslot0)/f)  (prb.py:227,288-313).
 * s2 = Lr_dir (or Lr_dir for a unit radiance) + tag, s3 = d Lr_dir / d slot0 + uv.x, s4 = (d f / d slot0) / f + uv.y -- the item layout of k_shade */
/* a texel-gradient record on its way to a queue (see TexelQueues): the bilinear cell, the fractions and the gradient of the interpolated colour */
struct TexelRecord { bool has; uint32_t q, cell, tex; float w1x, w1y; Vec3 g; };

/* the reverse-mode commit of one vertex with the lane's L and dL in REGISTERS (the in-place commit of k_shade keeps them there across the emission
 * term, this commit and the extra-parameter terms; the tape hands them from slot to slot): L <- L - [visible] Lr_dir, `dirty` says whether L changed */
__device__ __forceinline__ void adjoint_commit_regs(const DScene &S, bool pred, bool visible, float4 s2, float4 s3, float4 s4, Vec3 &L, bool &dirty, Vec3 dl,
                                                    float *grad_refl, float *const *grad_tex, float *gacc, const TexelQueues *tq = nullptr, TexelRecord *rec = nullptr) {
    Vec3 g(0.f); float *dst = grad_refl; bool tex = false; TexTaps taps; float *tdst = nullptr;
    bool em_lds = false; uint32_t em_slot = 0; Vec3 ge(0.f);
    if (pred) {
        const uint32_t tag = __float_as_uint(s2.w), bsdf = tag & 0xfffffu, emitter = (tag >> 20) & 0x7ffu;
        if (emitter != HAR_ITEM_NO_EMITTER) {       /* the item carries Lr_dir for a unit radiance: d Lr_dir / d radiance, and Lr_dir = unit * radiance */
            const DEmitter E = S.emitters[emitter];
            if (visible) {
                ge = Vec3(s2.x, s2.y, s2.z) * dl;
                em_slot = S.n_bsdfs + emitter;
                if (em_slot < HAR_LDS_GRAD_BSDFS) em_lds = true;        /* committed below by the whole wave (pre-reduced) */
                else { float *a = grad_refl + 3 * (size_t) em_slot; atomicAdd(a, ge.x); atomicAdd(a + 1, ge.y); atomicAdd(a + 2, ge.z); }
            }
            s2.x *= E.radiance[0]; s2.y *= E.radiance[1]; s2.z *= E.radiance[2];
        }
        if (visible) { L = Vec3(L.x - s2.x, L.y - s2.y, L.z - s2.z); dirty = true; }
        g = visible ? Vec3(s3.x, s3.y, s3.z) : Vec3(0.f);
        if (tag & 0x80000000u) g = g + Vec3(L.x * s4.x, L.y * s4.y, L.z * s4.z);
        g = g * dl;
        const DBsdf B = S.bsdfs[bsdf];
        dst = grad_refl + 3 * (size_t) bsdf;
        if (B.texture >= 0) { tex = true; tex_taps(S.textures[B.texture], s3.w, s4.w, taps); tdst = grad_tex[B.texture]; }
    }
    if (__ballot(em_lds)) wave_slot_add3(gacc, em_slot, ge, em_lds);
    const bool nz = pred && (g.x != 0.f || g.y != 0.f || g.z != 0.f);
    if (tq) {           /* queued textures: hand the record to the caller (its append), no atomics here.  `rec` is always the caller's local (a pointer that
                         * is null on one path keeps the record in scratch memory: 40 B per lane of k_commit until round 4) */
        rec->has = false;
        if (nz && tex) {
            const uint32_t t = (uint32_t) S.bsdfs[(uint32_t) (dst - grad_refl) / 3u].texture;
            const uint2 band = tq->band[t];
            if (band.x != 0xffffffffu) {
                const uint32_t W = S.textures[t].w, y0 = taps.idx[0] / W, x0 = taps.idx[0] - y0 * W;
                rec->has = true; rec->q = band.x + y0 / band.y; rec->cell = x0 | (y0 << 16); rec->tex = t; rec->w1x = taps.w1x; rec->w1y = taps.w1y; rec->g = g;
                tex = false; g = Vec3(0.f);
            }
        }
    }
    const bool nz_direct = nz && (g.x != 0.f || g.y != 0.f || g.z != 0.f || tex);
    {
        const uint32_t bsdf = (uint32_t) (dst - grad_refl) / 3u;
        const bool flat = nz_direct && !tex, lds = flat && bsdf < HAR_LDS_GRAD_BSDFS;
        if (__ballot(lds)) wave_slot_add3(gacc, bsdf, g, lds);
        if (flat && !lds) { atomicAdd(dst, g.x); atomicAdd(dst + 1, g.y); atomicAdd(dst + 2, g.z); }
    }
    if (__ballot(nz_direct && tex)) {
        const float w[4] = { taps.w0x * taps.w0y, taps.w1x * taps.w0y, taps.w0x * taps.w1y, taps.w1x * taps.w1y };
        for (int k = 0; k < 4; ++k)
            wave_aggregated_add3(nz_direct && tex ? tdst + 3 * (size_t) taps.idx[k] : grad_refl, nz_direct && tex ? g * w[k] : Vec3(0.f), nz_direct && tex);
    }
}
/* g * d radiance(u, v) / d texels into the gradient buffer of emitter `index`'s bitmap (type 7): the transpose of the lookup (nearest: four coincident taps of weight 1, 0, 0, 0);
 * called by whole waves */
__device__ __forceinline__ void light_texel_commit(const DScene &S, float *const *grad_tex, bool pred, int32_t index, float u, float v, Vec3 g) {
    pred = pred && (g.x != 0.f || g.y != 0.f || g.z != 0.f);
    if (!__ballot(pred)) return;
    TexTaps taps; float *dst = nullptr;
    taps.idx[0] = taps.idx[1] = taps.idx[2] = taps.idx[3] = 0u; taps.w0x = taps.w1x = taps.w0y = taps.w1y = 0.f;
    if (pred) { const uint32_t t = as_u32(S.emitters[index].radiance[0]); tex_taps(S.textures[t], u, v, taps); dst = grad_tex[t]; }
    const float w[4] = { taps.w0x * taps.w0y, taps.w1x * taps.w0y, taps.w0x * taps.w1y, taps.w1x * taps.w1y };
    for (int k = 0; k < 4; ++k) {
        const bool on = pred && w[k] != 0.f;
        wave_aggregated_add3(on ? dst + 3 * (size_t) taps.idx[k] : nullptr, on ? g * w[k] : Vec3(0.f), on);
    }
}
template <bool FWD = false>
__device__ __forceinline__ void adjoint_commit_values(const DScene &S, bool pred, bool visible, uint32_t lane, float4 s2, float4 s3, float4 s4, float4 *result, const float4 *dL,
                                                      float *grad_refl, float *const *grad_tex, float *gacc, const TexelQueues *tq = nullptr, TexelRecord *rec = nullptr) {
    if (FWD) {
        /* FORWARD mode (RBIntegrator.render_forward, common.py:497-623; prb.py:313 `dL += dr.forward_to(Lo)`): `grad_refl` / `grad_tex` hold the
         * TANGENTS of the parameters (same layout as the gradient buffers: slots of the BSDFs, then of the emitters; one array per bitmap) and are
         * only read; the lane's differential radiance accumulates in dL[lane] (one item per lane and bounce: no race), which raygen zeroed */
        if (pred) {
            float4 L = result[lane];
            const uint32_t tag = __float_as_uint(s2.w), bsdf = tag & 0xfffffu, emitter = (tag >> 20) & 0x7ffu;
            Vec3 acc(0.f);
            if (emitter != HAR_ITEM_NO_EMITTER) {
                const DEmitter E = S.emitters[emitter];
                if (visible) { const float *te = grad_refl + 3 * ((size_t) S.n_bsdfs + emitter); acc = Vec3(s2.x * te[0], s2.y * te[1], s2.z * te[2]); }
                s2.x *= E.radiance[0]; s2.y *= E.radiance[1]; s2.z *= E.radiance[2];
            }
            if (visible) { L = make_float4(L.x - s2.x, L.y - s2.y, L.z - s2.z, 0.f); result[lane] = L; }
            Vec3 g = visible ? Vec3(s3.x, s3.y, s3.z) : Vec3(0.f);
            if (tag & 0x80000000u) g = g + Vec3(L.x * s4.x, L.y * s4.y, L.z * s4.z);
            const DBsdf B = S.bsdfs[bsdf];
            Vec3 tan;
            if (B.texture >= 0) {
                TexTaps taps; tex_taps(S.textures[B.texture], s3.w, s4.w, taps);
                const float *tt = grad_tex[B.texture];
                const float w[4] = { taps.w0x * taps.w0y, taps.w1x * taps.w0y, taps.w0x * taps.w1y, taps.w1x * taps.w1y };
                tan = Vec3(0.f);
                for (int k = 0; k < 4; ++k) { const float *q = tt + 3 * (size_t) taps.idx[k]; tan = Vec3(fma_(q[0], w[k], tan.x), fma_(q[1], w[k], tan.y), fma_(q[2], w[k], tan.z)); }
            } else { const float *q = grad_refl + 3 * (size_t) bsdf; tan = Vec3(q[0], q[1], q[2]); }
            acc = acc + g * tan;
            float4 *acc_out = const_cast<float4 *>(dL);
            const float4 d = acc_out[lane];
            acc_out[lane] = make_float4(d.x + acc.x, d.y + acc.y, d.z + acc.z, 0.f);
        }
        return;
    }
    Vec3 L(0.f), dl(0.f); bool dirty = false;
    if (pred) { const float4 r = result[lane], d4 = dL[lane]; L = Vec3(r.x, r.y, r.z); dl = Vec3(d4.x, d4.y, d4.z); }
    adjoint_commit_regs(S, pred, visible, s2, s3, s4, L, dirty, dl, grad_refl, grad_tex, gacc, tq, rec);
    if (pred && dirty) result[lane] = make_float4(L.x, L.y, L.z, 0.f);
}
/* the four taps of a record committed with direct atomics (queue overflow); wave-uniform call */
__device__ __forceinline__ void texel_record_direct(const DScene &S, float *const *grad_tex, float *dummy, const TexelRecord &r, bool active) {
    if (!__ballot(active)) return;
    uint32_t W = 1u, x0 = 0u, y0 = 0u, x1 = 0u, y1 = 0u; float *dst = dummy;
    if (active) {
        W = S.textures[r.tex].w; const uint32_t H = S.textures[r.tex].h;
        x0 = r.cell & 0xffffu; y0 = r.cell >> 16; x1 = x0 + 1 == W ? 0u : x0 + 1; y1 = y0 + 1 == H ? 0u : y0 + 1;
        dst = grad_tex[r.tex];
    }
    for (int k = 0; k < 4; ++k) {        /* the four bilinear taps (no indexed arrays: they would live in scratch) */
        const uint32_t idx = ((k & 2) ? y1 : y0) * W + ((k & 1) ? x1 : x0);
        const float w = ((k & 1) ? r.w1x : 1.f - r.w1x) * ((k & 2) ? r.w1y : 1.f - r.w1y);
        wave_aggregated_add3(active ? dst + 3 * (size_t) idx : dummy, active ? r.g * w : Vec3(0.f), active);
    }
}
/* append the block's texel records to their band queues (TexelQueues): LDS histogram -> one global atomic per non-empty band -> scattered 32-byte records; a record
 * that finds its queue full is committed with direct atomics by its lane.  Block-wide (three barriers); hist / base: HAR_TQ_MAX words each, gmax: one word of LDS. */
__device__ __forceinline__ void texel_queue_append(const DScene &S, const TexelQueues &tq, uint32_t shard, const TexelRecord &rec, uint32_t *hist, uint32_t *base, uint32_t *gmax,
                                                   float *const *grad_tex, float *grad_slots) {
    if (threadIdx.x < tq.nq) hist[threadIdx.x] = 0u;
    __syncthreads();
    texel_record_track_max(gmax, rec.has, rec.g);
    const uint32_t rank = wave_ranked_count(hist, rec.q, rec.has);
    __syncthreads();
    if (threadIdx.x < tq.nq) { const uint32_t c = hist[threadIdx.x]; base[threadIdx.x] = c ? atomicAdd(tq.count + (size_t) (shard * tq.nq + threadIdx.x) * HAR_COUNTER_STRIDE, c) : 0u; }
    __syncthreads();
    bool overflow = false;
    if (rec.has) {
        const uint32_t slot = base[rec.q] + rank;
        if (slot < tq.cap) {
            float4 *dst = tq.rec + 2 * ((size_t) (shard * tq.nq + rec.q) * tq.cap + slot);
            dst[0] = make_float4(__uint_as_float(rec.cell), __uint_as_float(rec.tex), rec.w1x, rec.w1y);
            dst[1] = make_float4(rec.g.x, rec.g.y, rec.g.z, 0.f);
        } else overflow = true;
    }
    texel_record_direct(S, grad_tex, grad_slots, rec, overflow);
}
template <bool FWD = false>
__device__ __forceinline__ void adjoint_commit(const DScene &S, const ItemArrays &items, uint32_t i, bool pred, bool visible, float4 *result, const float4 *dL,
                                               float *grad_refl, float *const *grad_tex, float *gacc, uint8_t *item_vis, const TexelQueues *tq = nullptr, TexelRecord *rec = nullptr) {
    if (pred && item_vis) item_vis[i] = visible ? 1 : 0;
    float4 s2 = make_float4(0.f, 0.f, 0.f, 0.f), s3 = s2, s4 = s2; uint32_t lane = 0;
    if (pred) { lane = __float_as_uint(items.s1[i].w); s2 = items.s2[i]; s3 = items.s3[i]; s4 = items.s4[i]; }
    adjoint_commit_values<FWD>(S, pred, visible, lane, s2, s3, s4, result, dL, grad_refl, grad_tex, gacc, tq, rec);
}

/* ---------------------------------------------------------------- classify */
/* Material classification of a bounce's closest hits (MaterialQueues): every 256-path tile of a shard is dealt to the class lists in SLOT ORDER (wave
 * ballots give each path its rank within its class; one atomic per class and tile reserves the list range), so a class kernel reads its tile's path
 * state from one 4 KB window per array in increasing order. */
__global__ __launch_bounds__(kBlock) void k_classify(DScene S, uint32_t shard_cap, const uint32_t *count_in, const float4 *h0, const uint2 *h1, MaterialQueues mq) {
    __shared__ uint32_t wave_cnt[kBlock / 64][HAR_MAT_CLASSES + 1], cls_base[HAR_MAT_CLASSES + 1];
    const ShardLoop Q(count_in, shard_cap);
    const uint32_t wave = threadIdx.x >> 6;
    for (uint32_t tile = Q.first_tile(); tile * kBlock < Q.n; tile += Q.tile_step()) {
        const uint32_t local = tile * kBlock + threadIdx.x;
        uint32_t key = HAR_MAT_CLASSES;                                  /* beyond the shard's end */
        if (local < Q.n) {
            const float t = h0[HIT0(Q.base + local)].x;
            if (t == HAR_INF) key = mq.miss_class;
            else key = min(S.meshes[h1[HIT1(Q.base + local)].x].pad1, (uint32_t) HAR_MAT_GENERIC);        /* the mesh's material class (har_scene_create) */
        }
        uint32_t rank = 0;
#pragma unroll
        for (uint32_t c = 0; c < HAR_MAT_CLASSES; ++c) {
            const uint64_t m = __ballot(key == c);
            if (key == c) rank = wave_rank(m);
            if ((threadIdx.x & 63u) == 0u) wave_cnt[wave][c] = (uint32_t) __popcll(m);
        }
        __syncthreads();
        if (threadIdx.x < HAR_MAT_CLASSES) {
            uint32_t total = 0; for (uint32_t w = 0; w < kBlock / 64; ++w) total += wave_cnt[w][threadIdx.x];
            cls_base[threadIdx.x] = total ? atomicAdd(mq.count + (size_t) (threadIdx.x * HAR_SHARDS + Q.shard) * HAR_COUNTER_STRIDE, total) : 0u;
        }
        __syncthreads();
        if (key < HAR_MAT_CLASSES) {
            uint32_t pos = cls_base[key] + rank;
            for (uint32_t w = 0; w < wave; ++w) pos += wave_cnt[w][key];
            mq.idx[(size_t) key * mq.lanes + Q.base + pos] = local;
        }
        __syncthreads();
    }
}

/* ------------------------------------------------------------------- shade */
/* INLINE (adjoint replay of a bounce whose shadow-ray results sit in the replay cache): the visibility of the lane's emitter sample is known here,
 * so the vertex's adjoint is committed on the spot instead of going through an item (80 B written + read) and k_resolve_adjoint_cached */
/* TAB: scenes whose mesh, BSDF and instance tables are small (HAR_TAB_MESHES / _BSDFS / _INSTS records: 17 KB) get a per-block copy of them in LDS.  The shading
 * kernels are bound by the NUMBER of their memory transactions (k_shade at 76 % address-unit busy, docs/rounds/r04.md item 12): a vertex gathers two 16-byte pieces of its
 * mesh record, two or three of its BSDF record and -- inside an instance -- six of the instance's transforms, which ds_read_b128 serves without the texture-address path. */
#define HAR_TAB_MESHES 64
#define HAR_TAB_BSDFS 32
#define HAR_TAB_INSTS 128
template <int MODE, uint32_t TYPES, bool SHAPE = false, bool INLINE = false, bool EXTRA = false, bool QUEUED = false, bool RECORD = false, bool TAB = false, bool FIRST = false>
__global__ __launch_bounds__(kBlock, (RECORD && TYPES == HAR_BSDF_ONLY_DIFFUSE && !SHAPE && !EXTRA) ? HAR_RECORD_MIN_WAVES : HAR_SHADE_MIN_WAVES) void k_shade(DScene S_in, ShadeParams P, uint32_t lane_base, uint32_t shard_cap, const uint32_t *count_in, WaveState in,
                                                  const float4 *h0, const uint2 *h1, WaveState out, uint32_t *count_out,
                                                  ItemArrays items, uint32_t *item_count, float4 *result, ReplayCache rc, uint64_t *pass_rng,
                                                  const float4 *dL, float *grad_slots, ShapeArrays geo, float *const *grad_tex, TexelQueues tq, float *grad_extra,
                                                  MaterialQueues mq, uint32_t mat_class, TapeArrays tape) {
    __shared__ uint32_t lds_r[12];
    __shared__ uint4 tab_mesh[TAB ? HAR_TAB_MESHES * sizeof(DMesh) / 16 : 1], tab_bsdf[TAB ? HAR_TAB_BSDFS * sizeof(DBsdf) / 16 : 1], tab_inst[TAB ? HAR_TAB_INSTS * sizeof(DInst) / 16 : 1];
    DScene S = S_in;
    if (TAB) {
        static_assert(sizeof(DMesh) % 16 == 0 && sizeof(DBsdf) % 16 == 0 && sizeof(DInst) % 16 == 0, "table records are copied as 16-byte words");
        const uint4 *gm = reinterpret_cast<const uint4 *>(S_in.meshes), *gb = reinterpret_cast<const uint4 *>(S_in.bsdfs), *gi = reinterpret_cast<const uint4 *>(S_in.insts);
        for (uint32_t k = threadIdx.x; k < S_in.n_meshes * (uint32_t) (sizeof(DMesh) / 16); k += kBlock) tab_mesh[k] = gm[k];
        for (uint32_t k = threadIdx.x; k < S_in.n_bsdfs * (uint32_t) (sizeof(DBsdf) / 16); k += kBlock) tab_bsdf[k] = gb[k];
        for (uint32_t k = threadIdx.x; k < S_in.n_insts * (uint32_t) (sizeof(DInst) / 16); k += kBlock) tab_inst[k] = gi[k];
        __syncthreads();
        S.meshes = reinterpret_cast<const DMesh *>(tab_mesh); S.bsdfs = reinterpret_cast<const DBsdf *>(tab_bsdf); S.insts = reinterpret_cast<const DInst *>(tab_inst);
    }
    /* EXTRA: gradients w.r.t. alpha_u, alpha_v, eta, k, colour slot 1 of the rough BSDF records (15 floats per record): per-block accumulators for
     * the first HAR_LDS_EXTRA_BSDFS records, global atomics beyond */
    __shared__ float xacc[EXTRA ? 15 * HAR_LDS_EXTRA_BSDFS : 1];
    if (EXTRA) { for (uint32_t k = threadIdx.x; k < 15 * HAR_LDS_EXTRA_BSDFS; k += kBlock) xacc[k] = 0.f; __syncthreads(); }
    __shared__ uint32_t tq_hist[INLINE ? HAR_TQ_MAX : 1], tq_base[INLINE ? HAR_TQ_MAX : 1], tq_gmax;
    __shared__ float gacc[INLINE ? 3 * HAR_LDS_GRAD_BSDFS : 1];
    if (INLINE) { for (uint32_t k = threadIdx.x; k < 3 * HAR_LDS_GRAD_BSDFS; k += kBlock) gacc[k] = 0.f; if (threadIdx.x == 0) tq_gmax = 0u; __syncthreads(); }
    /* adjoint with emitter gradients: per-block accumulators of d L / d radiance from emission hits (slots n_bsdfs + emitter of `grad_slots`) */
    __shared__ float eacc[MODE == MODE_PRB_ADJOINT ? 3 * HAR_LDS_GRAD_EMITTERS : 1];
    const bool fwd = MODE == MODE_PRB_ADJOINT && (P.flags & HAR_SHADE_FORWARD_MODE) != 0u;      /* render_forward: tangents in, dL accumulates (see adjoint_commit_values) */
    const bool emitter_grads = MODE == MODE_PRB_ADJOINT && (P.flags & HAR_SHADE_EMITTER_GRADS) != 0u;
    if (emitter_grads) { for (uint32_t k = threadIdx.x; k < 3 * HAR_LDS_GRAD_EMITTERS; k += kBlock) eacc[k] = 0.f; __syncthreads(); }
    /* sort buckets: the models without microfacet code, escaped paths, the rough models, two-model pairs (HAR_MAT_GENERIC), lanes beyond the wavefront's end */
    constexpr uint32_t kSortKeys = HAR_MAT_CLASSES + 2;
#ifndef HAR_SORT_FIRST
#define HAR_SORT_FIRST 0            /* 1: the first-vertex flavour sorts by material too.  Bounce 0 of a wavefront rebuilt from the lane index: a block's 256 paths are samples of one or a
                                     * few pixels and meet one or two materials -- the sort's key pass (a second read of the hit records) and permuted loads bought nothing:
                                     * generic k_shade 21.95 -> 21.35 ms per frame on materials1m, forward +0.8 %, prb +1.2 %, bracketed (profiles/r05_pmc_generic_shade_materials1m.txt) */
#endif
    constexpr bool kSorted = !QUEUED && TYPES != HAR_BSDF_ONLY_DIFFUSE && HAR_MATERIAL_SORT && (HAR_SORT_FIRST || !FIRST);
    __shared__ uint32_t sort_cnt[kSortKeys];
    __shared__ uint16_t sort_perm[kSorted ? kBlock * HAR_SORT_WINDOW_MAX : 1], sort_tmp[kSorted ? kBlock * HAR_SORT_WINDOW_MAX : 1];
#if HAR_SORT_SCHEDULE
    __shared__ uint8_t sort_chunk[kSorted ? (kBlock / 64) * HAR_SORT_WINDOW_MAX : 1];       /* the window's 64-path chunks in the order they are shaded */
#endif
    ShardLoop Q(count_in, shard_cap);
    /* QUEUED: the paths of ONE material class of this shard, through the index list k_classify built (MaterialQueues) */
    const uint32_t *q_idx = QUEUED ? mq.idx + (size_t) mat_class * mq.lanes + Q.base : nullptr;
    if (QUEUED) Q.n = mq.count[(size_t) (mat_class * HAR_SHARDS + Q.shard) * HAR_COUNTER_STRIDE];
    uint32_t *cnt_alive = count_out + Q.shard * HAR_COUNTER_STRIDE, *cnt_item = item_count + Q.shard * HAR_COUNTER_STRIDE;
    /* MATERIAL SORT over a WINDOW of `win` tiles (generic kernels): the window's 256 * win paths are re-dealt to the lanes by the BSDF model of the surface they hit
     * (block-wide counting sort in LDS), then shaded in `win` rounds of 256, so that a wave runs (mostly) ONE material model of the generic shading code instead of
     * diverging over all of them.  With four models a 256-path window leaves three of four waves with two models (modelled cost 3.2 against 1.7 for perfectly sorted
     * waves, 7.2 unsorted); 2048 paths bring that to 1.9.  The window is a contiguous 4 KB * win range per state array, so the permuted loads still consume whole cache
     * lines (across rounds).  win = 1 for wavefronts too small to give every block a whole window (ShadeParams::sort_window caps it; 1 = the round-3 kernel). */
    const uint32_t win = kSorted ? max(1u, min(min(P.sort_window, (uint32_t) HAR_SORT_WINDOW_MAX), Q.n / (Q.tile_step() * kBlock))) : 1u;
    for (uint32_t tile0 = Q.first_tile() * win; tile0 * kBlock < Q.n; tile0 += Q.tile_step() * win) {
        if (kSorted) {
            __syncthreads();                                             /* the previous window's bucket counts have been read by everyone */
            if (threadIdx.x < kSortKeys) sort_cnt[threadIdx.x] = 0;
            __syncthreads();
            for (uint32_t r = 0; r < win; ++r) {
                const uint32_t l = (tile0 + r) * kBlock + threadIdx.x;
                uint32_t key = kSortKeys - 1u;                             /* out of range */
                if (l < Q.n) {
                    uint2 hs; float t;
                    if (MODE == MODE_PRB_ADJOINT && rc.mode == 2) {
                        const uint32_t cl = __float_as_uint(in.a3[Q.base + l].w) - lane_base;
                        hs = rc.h1[cl]; t = rc.h0[cl].x;
                    } else { hs = h1[HIT1(Q.base + l)]; t = h0[HIT0(Q.base + l)].x; }
                    /* the mesh's material class (DMesh::pad1, har_scene_create): diffuse 0, dielectric 1, rough conductor 2, rough plastic 3, conductor 4, plastic 5, generic 6 */
#if HAR_HIT_MATINFO && HAR_HIT_INTERLEAVED
                    const bool from_record = !(MODE == MODE_PRB_ADJOINT && rc.mode == 2);
                    const uint32_t cls = t == HAR_INF ? 7u : min(from_record ? HAR_MATINFO_CLASS(h1[HIT1(Q.base + l) + 1].y) : S.meshes[hs.x].pad1, (uint32_t) HAR_MAT_GENERIC);
#else
                    const uint32_t cls = t == HAR_INF ? 7u : min(S.meshes[hs.x].pad1, (uint32_t) HAR_MAT_GENERIC);
#endif
                    key = (0x47326510u >> (4u * cls)) & 0xfu;               /* buckets: diffuse, dielectric, conductor, plastic, escaped | rough conductor, rough plastic, generic */
                }
                sort_tmp[r * kBlock + threadIdx.x] = (uint16_t) ((key << 12) | atomicAdd(&sort_cnt[key], 1u));      /* rank within the bucket (< 4096) */
            }
            __syncthreads();
            for (uint32_t r = 0; r < win; ++r) {
                const uint32_t kp = sort_tmp[r * kBlock + threadIdx.x], key = kp >> 12;
                uint32_t off = 0;
                for (uint32_t k = 0; k < key; ++k) off += sort_cnt[k];
                sort_perm[off + (kp & 0xfffu)] = (uint16_t) (r * kBlock + threadIdx.x);
            }
            __syncthreads();
#if HAR_SORT_SCHEDULE
            /* CHUNK SCHEDULE.  Shading the sorted window front to back gives the four waves of a round four consecutive 64-path chunks of ONE class: paths that lie
             * 1 / share apart in the wavefront, so a round's permuted 16-byte loads use one or two entries of every 128-byte line they touch, and the rest of the line
             * is fetched again rounds later by the other classes' chunks -- after the windows of the XCD's other blocks (27 MB) have pushed it out of the 4 MB L2:
             * 405 B read per vertex against 167 unsorted, the kernel at 5.8 TB/s (profiles/r05_pmc_generic_shade_materials1m.txt).  Within a class the sort keeps
             * the wavefront's order, so every chunk covers an interval of the window; taking the chunks in the order of where their intervals START makes the waves
             * of a round walk the same stretch of the window at the same time, each for its own class: a line is fetched once for all of them. */
            if (threadIdx.x < 4u * win) {
                const uint32_t me = threadIdx.x, key = sort_perm[64u * me];
                uint32_t rank = 0;
                for (uint32_t j = 0; j < 4u * win; ++j) { const uint32_t kj = sort_perm[64u * j]; rank += (kj < key || (kj == key && j < me)) ? 1u : 0u; }
                sort_chunk[rank] = (uint8_t) me;
            }
            __syncthreads();
#endif
        }
        /* the sorted window is shaded in rounds of 256 (the tail of the last round holds the out-of-range lanes) */
        const uint32_t rounds = kSorted ? win : 1u;
      for (uint32_t round = 0; round < rounds; ++round) {
#if HAR_SORT_SCHEDULE
        const uint32_t pos = kSorted ? 64u * sort_chunk[round * (kBlock / 64u) + (threadIdx.x >> 6)] + (threadIdx.x & 63u) : round * kBlock + threadIdx.x;
#else
        const uint32_t pos = round * kBlock + threadIdx.x;
#endif
        uint32_t local = kSorted ? tile0 * kBlock + sort_perm[pos] : tile0 * kBlock + threadIdx.x;
        if (QUEUED) local = local < Q.n ? q_idx[local] : 0xffffffffu;
        const bool in_range = QUEUED ? local != 0xffffffffu : local < Q.n;
        const uint32_t i = Q.base + local;
        ShadeResult R; R.alive = false; R.item = false; R.add_emission = false;
        uint32_t lane = 0;
        float4 hh = make_float4(0.f, 0.f, 0.f, 0.f); uint2 hs = make_uint2(0u, 0u); Vec3 d_in(0.f); bool first_vertex = true;
        /* in-place adjoint commit: the lane's L and dL stay in registers from the emission term to the write-back; tape replay (rc.mode == 4): they come
         * from / go to the slot-ordered tape arrays and nothing is compacted (TapeArrays) */
        Vec3 Lr(0.f), dlr(0.f); bool L_dirty = false;
        const bool tape_read = MODE == MODE_PRB_ADJOINT && INLINE && rc.mode == 4;
        bool eg_lds = false; uint32_t eg_slot = 0; Vec3 eg(0.f);          /* d L / d radiance of the emitter met by this lane (emission hit) */
        bool lt_hit = false; Vec3 lt_hit_g(0.f);                            /* ... when that emitter's radiance is a bitmap and its texels are differentiated (HAR_SHADE_LIGHT_TEXELS) */
        if (in_range) {
            PathState st;
            if (FIRST) {         /* slot `local` of shard s holds lane ((local / 256) * HAR_SHARDS + s) * 256 + local % 256 of the chunk (shard_slot) */
                LaneSample ls;
                const uint32_t li = ((local / kBlock) * HAR_SHARDS + Q.shard) * kBlock + (local % kBlock);
                st = raygen_lane(P.sensor, P.seed, P.spp, P.log_spp, lane_base + li, ls, (P.resume && pass_rng) ? pass_rng + li : nullptr);
            } else st = load_state<TYPES == HAR_BSDF_ONLY_DIFFUSE>(in, i);
            d_in = st.d; first_vertex = (st.flags & 0xffffu) == 0u;
            HitExtra hx{ 0u, 0u }; bool has_hx = false;
            if (MODE == MODE_PRB_ADJOINT && rc.mode == 2) { hh = rc.h0[st.lane - lane_base]; hs = rc.h1[st.lane - lane_base]; }      /* replay cache: 24-byte records */
            else {
                hh = h0[HIT0(i)];
#if HAR_HIT_MATINFO && HAR_HIT_INTERLEAVED
                const uint4 q = *reinterpret_cast<const uint4 *>(h1 + HIT1(i));      /* {shape, inst, face in the shading-triangle array, material word} */
                hs = make_uint2(q.x, q.y); hx.gface = q.z; hx.matinfo = q.w; has_hx = true;
#else
                hs = h1[HIT1(i)];
#endif
            }
            if (MODE == MODE_PRB_PRIMAL && rc.mode == 1) { rc.h0[st.lane - lane_base] = hh; rc.h1[st.lane - lane_base] = hs; }
            Hit hit; hit.t = hh.x; hit.u = hh.y; hit.v = hh.z; hit.prim = __float_as_uint(hh.w); hit.shape = hs.x; hit.inst = hs.y;
            shade_lane<MODE, TYPES, EXTRA>(S, P, st, hit, R, has_hx ? &hx : nullptr);
            lane = st.lane - lane_base;
            if (MODE == MODE_PRB_ADJOINT && INLINE) {
                if (tape_read) { const float4 a = tape.la_in[i]; const float2 c2 = tape.lb_in[i]; Lr = Vec3(a.x, a.y, a.z); dlr = Vec3(a.w, c2.x, c2.y); }
                else { const float4 r = result[lane], d4 = dL[lane]; Lr = Vec3(r.x, r.y, r.z); dlr = Vec3(d4.x, d4.y, d4.z); }
            }
            if (MODE == MODE_PATH && pass_rng && !R.alive) {
                /* multi-pass render: the path ends here, its sampler lives on.  A lane that starts a loop iteration draws all six
                 * numbers of that iteration whether or not it survives it (symbolic dr::while_loop: unmasked draws, path.cpp:247,263-264,323) */
                uint64_t r = st.rng; const uint64_t inc = sampler_inc(P.seed, st.lane);
                for (int k = 0; k < 6; ++k) r = r * HAR_PCG32_MULT + inc;
                pass_rng[lane] = r;
            }
            if (R.add_emission && MODE == MODE_PRB_ADJOINT && INLINE) {
                Lr = Vec3(Lr.x - R.em_b.x, Lr.y - R.em_b.y, Lr.z - R.em_b.z); L_dirty = true;
                if ((TYPES & HAR_SCENE_TEXLIGHT) != 0u && R.lt_hit_emitter >= 0) { lt_hit = true; lt_hit_g = R.em_unit * dlr; }
                if (emitter_grads && R.em_index >= 0) {
                    const Vec3 g = R.em_unit * dlr;
                    if ((uint32_t) R.em_index < HAR_LDS_GRAD_EMITTERS) { eg_lds = true; eg_slot = (uint32_t) R.em_index; eg = g; }      /* committed below by the whole wave */
                    else { float *a = grad_slots + 3 * ((size_t) S.n_bsdfs + R.em_index); atomicAdd(a, g.x); atomicAdd(a + 1, g.y); atomicAdd(a + 2, g.z); }
                }
            } else if (R.add_emission) {
                float4 r = result[lane];
                if (MODE == MODE_PATH)            r = make_float4(fma_(R.em_a.x, R.em_b.x, r.x), fma_(R.em_a.y, R.em_b.y, r.y), fma_(R.em_a.z, R.em_b.z, r.z), 0.f);
                else if (MODE == MODE_PRB_PRIMAL || RECORD) r = make_float4(r.x + R.em_b.x, r.y + R.em_b.y, r.z + R.em_b.z, 0.f);      /* RECORD: this IS the primal pass */
                else                              r = make_float4(r.x - R.em_b.x, r.y - R.em_b.y, r.z - R.em_b.z, 0.f);
                result[lane] = r;
                if (RECORD) tape.rec_em[i] = make_float4(R.em_b.x, R.em_b.y, R.em_b.z, 0.f);
                if (MODE == MODE_PRB_ADJOINT && emitter_grads && R.em_index >= 0 && fwd) {
                    const float *te = grad_slots + 3 * ((size_t) S.n_bsdfs + R.em_index);        /* tangent of the emitter's radiance */
                    float4 *acc_out = const_cast<float4 *>(dL);
                    const float4 d = acc_out[lane];
                    acc_out[lane] = make_float4(fma_(R.em_unit.x, te[0], d.x), fma_(R.em_unit.y, te[1], d.y), fma_(R.em_unit.z, te[2], d.z), 0.f);
                } else if (MODE == MODE_PRB_ADJOINT && emitter_grads && R.em_index >= 0) {
                    const float4 dl = dL[lane];
                    const Vec3 g = R.em_unit * Vec3(dl.x, dl.y, dl.z);
                    if ((uint32_t) R.em_index < HAR_LDS_GRAD_EMITTERS) { eg_lds = true; eg_slot = (uint32_t) R.em_index; eg = g; }
                    else { float *a = grad_slots + 3 * ((size_t) S.n_bsdfs + R.em_index); atomicAdd(a, g.x); atomicAdd(a + 1, g.y); atomicAdd(a + 2, g.z); }
                }
            }
        }
        if (MODE == MODE_PRB_ADJOINT && emitter_grads && __ballot(eg_lds)) wave_slot_add3(eacc, eg_slot, eg, eg_lds);
        bool item_pred = in_range && R.item;
        if (SHAPE) item_pred = in_range && (R.item || R.alive);       /* the solid-angle-to-area Jacobian of a continued path moves with the vertex whatever the BSDF value */
        if (INLINE) {
            const bool fact = item_pred && R.nee_emitter >= 0 && (uint32_t) R.nee_emitter < HAR_ITEM_NO_EMITTER;
            const Vec3 c = fact ? R.contrib_unit : R.contrib;
            const uint32_t tag = (R.bsdf & 0xfffffu) | ((fact ? (uint32_t) R.nee_emitter : HAR_ITEM_NO_EMITTER) << 20) | (R.ind_active ? 0x80000000u : 0u);
            const bool visible = item_pred && R.item_ray && rc.vis[tape_read ? i : lane] != 0;      /* the tape's visibility bytes are per vertex slot */
            TexelRecord rec; rec.has = false;
            adjoint_commit_regs(S, item_pred, visible, make_float4(c.x, c.y, c.z, __uint_as_float(tag)), make_float4(R.dLr_drho.x, R.dLr_drho.y, R.dLr_drho.z, R.uv_x),
                                make_float4(R.rel_grad.x, R.rel_grad.y, R.rel_grad.z, R.uv_y), Lr, L_dirty, dlr, grad_slots, grad_tex, gacc,      /* forward mode never commits in place (host) */
                                tq.nq ? &tq : nullptr, &rec);
            if ((TYPES & HAR_SCENE_TEXLIGHT) != 0u && (P.flags & HAR_SHADE_LIGHT_TEXELS)) {
                /* the texels of a bitmap `radiance` (area.cpp:64-70): d Le / d radiance(si.uv) = beta mis at an emitter hit, d Lr_dir / d radiance(ds.uv) = beta mis f / pdf at a
                 * VISIBLE emitter sample -- scattered through the transpose of the bilinear lookup, pre-reduced over the wave (all lanes of a block sample the same few texels) */
                light_texel_commit(S, grad_tex, in_range && lt_hit, R.lt_hit_emitter, R.lt_hit_uv[0], R.lt_hit_uv[1], lt_hit_g);
                const bool lt_nee = visible && R.lt_nee_emitter >= 0;
                light_texel_commit(S, grad_tex, lt_nee, R.lt_nee_emitter, R.lt_nee_uv[0], R.lt_nee_uv[1], R.contrib_unit * dlr);
            }
            if (EXTRA && item_pred) {
                /* the same two terms as for slot 0 (adjoint_commit_regs), for the five other parameter groups: g = dL * ([visible] d Lr_dir / d theta
                 * + [path continues] L * (d f / d theta) / f), with L already reduced by this vertex's Lr_dir */
                const Vec3 L = Lr, dl = dlr;
                for (int g = 0; g < HAR_EXTRA_GROUPS; ++g) {
                    Vec3 v = visible ? R.x_dir[g] : Vec3(0.f);
                    if (R.x_ind) v = v + Vec3(L.x * R.x_rel[g].x, L.y * R.x_rel[g].y, L.z * R.x_rel[g].z);
                    v = v * Vec3(dl.x, dl.y, dl.z);
                    if (v.x == 0.f && v.y == 0.f && v.z == 0.f) continue;
                    if (R.bsdf < HAR_LDS_EXTRA_BSDFS) { float *a = xacc + 15 * R.bsdf + 3 * g; atomicAdd(a, v.x); atomicAdd(a + 1, v.y); atomicAdd(a + 2, v.z); }
                    else { float *a = grad_extra + 15 * (size_t) R.bsdf + 3 * g; atomicAdd(a, v.x); atomicAdd(a + 1, v.y); atomicAdd(a + 2, v.z); }
                }
            }
            /* write-back: replay -> the survivor's slot of the next bounce (the primal pass's compaction, TapeArrays::next); otherwise the lane's result */
            if (in_range) {
                if (tape_read) {
                    const uint32_t nx = R.alive ? tape.next[i] : 0xffffffffu;
                    if (nx != 0xffffffffu) { tape.la_out[nx] = make_float4(Lr.x, Lr.y, Lr.z, dlr.x); tape.lb_out[nx] = make_float2(dlr.y, dlr.z); }
                } else if (L_dirty) result[lane] = make_float4(Lr.x, Lr.y, Lr.z, 0.f);
            }
            item_pred = false;
            if (tq.nq) {
                texel_queue_append(S, tq, Q.shard, rec, tq_hist, tq_base, &tq_gmax, grad_tex, grad_slots);
            }
        }
        if (RECORD && item_pred) {
            /* record tape: what the commit kernel needs of this vertex (the item layout of the adjoint kernels, filed per VERTEX SLOT) */
            const bool fact = R.nee_emitter >= 0 && (uint32_t) R.nee_emitter < HAR_ITEM_NO_EMITTER;
            const Vec3 c = fact ? R.contrib_unit : R.contrib;
            const uint32_t tag = (R.bsdf & 0xfffffu) | ((fact ? (uint32_t) R.nee_emitter : HAR_ITEM_NO_EMITTER) << 20) | (R.ind_active ? 0x80000000u : 0u);
            if (HAR_TAPE_COMPACT && TYPES == HAR_BSDF_ONLY_DIFFUSE && !EXTRA) {
                /* COMPACT record of a diffuse vertex, 36 B instead of 48: Lr_dir = dLr_drho * rho and (df / d rho) / f = 1 / rho, so {dLr_drho (for a unit radiance when the
                 * emitter is differentiated), rho} carries the three vectors -- also at rho = 0 and radiance = 0, where the ratios Lr_dir / rho and Lr_dir / radiance do not
                 * exist; uv.y rides in the first quarter of the rec2 array (k_commit<.., COMPACT> rebuilds the three) */
                const Vec3 d = fact ? R.dLr_drho_unit : R.dLr_drho;
                tape.rec0[i] = make_float4(d.x, d.y, d.z, __uint_as_float(tag));
                tape.rec1[i] = make_float4(R.rho.x, R.rho.y, R.rho.z, R.uv_x);
                reinterpret_cast<float *>(tape.rec2)[i] = R.uv_y;
            } else {
                tape.rec0[i] = make_float4(c.x, c.y, c.z, __uint_as_float(tag));
                tape.rec1[i] = make_float4(R.dLr_drho.x, R.dLr_drho.y, R.dLr_drho.z, R.uv_x);
                tape.rec2[i] = make_float4(R.rel_grad.x, R.rel_grad.y, R.rel_grad.z, R.uv_y);
            }
        }
        const bool has_rec = RECORD && item_pred;
        if (RECORD) item_pred = item_pred && R.item_ray;            /* the queue of the primal pass holds shadow rays only */
        const bool alive = in_range && R.alive, item = item_pred;
        uint32_t slot = 0, islot = 0;
        if (!tape_read) {                                                                          /* tape replay: the primal pass's slots stand, nothing is stored */
            if ((HAR_WAVE_RESERVE == 1 && kSorted) || HAR_WAVE_RESERVE == 2) wave_reserve2(cnt_alive, alive, cnt_item, item, slot, islot);
            else block_reserve2(cnt_alive, alive, cnt_item, item, lds_r, slot, islot);
        }
        if (alive && !tape_read) store_state(out, Q.base + slot, R.next);
        if (MODE == MODE_PRB_PRIMAL && rc.mode == 3 && in_range) tape.next[i] = alive ? Q.base + slot : 0xffffffffu;
        if (RECORD && in_range) tape.next[i] = (alive ? Q.base + slot : HAR_TAPE_DEAD) | (item ? HAR_TAPE_HAS_RAY : 0u) | (has_rec ? HAR_TAPE_HAS_REC : 0u) | (R.add_emission ? HAR_TAPE_HAS_EM : 0u);
        if (item) {
            islot += Q.base;
            items.s0[islot] = make_float4(R.sh_o.x, R.sh_o.y, R.sh_o.z, R.item_ray ? R.sh_maxt : -1.f);
            items.s1[islot] = make_float4(R.sh_d.x, R.sh_d.y, R.sh_d.z, __uint_as_float(lane));
            if (MODE == MODE_PRB_ADJOINT && !RECORD) {
                /* tag = bsdf (20 bits) | emitter (11 bits) | indirect-term flag; with an emitter the item carries the contribution for a unit radiance */
                const bool fact = R.nee_emitter >= 0 && (uint32_t) R.nee_emitter < HAR_ITEM_NO_EMITTER;
                const Vec3 c = fact ? R.contrib_unit : R.contrib;
                const uint32_t tag = (R.bsdf & 0xfffffu) | ((fact ? (uint32_t) R.nee_emitter : HAR_ITEM_NO_EMITTER) << 20) | (R.ind_active ? 0x80000000u : 0u);
                items.s2[islot] = make_float4(c.x, c.y, c.z, __uint_as_float(tag));
            } else items.s2[islot] = make_float4(R.contrib.x, R.contrib.y, R.contrib.z, __uint_as_float(((HAR_RESOLVE_ATOMIC || HAR_RESOLVE_LANE_S2) && (rc.mode == 0 || rc.mode == 1)) ? lane : i));
            /* .w: the vertex slot in tape modes (where k_resolve files the visibility); otherwise the LANE again (HAR_RESOLVE_ATOMIC), so that k_resolve's commit reads
             * the contribution and its destination with ONE 16-byte load */
            if (MODE == MODE_PRB_ADJOINT && !RECORD) {
                items.s3[islot] = make_float4(R.dLr_drho.x, R.dLr_drho.y, R.dLr_drho.z, R.uv_x);
                items.s4[islot] = make_float4(R.rel_grad.x, R.rel_grad.y, R.rel_grad.z, R.uv_y);
            }
            if (SHAPE) {        /* geometry record of the vertex for k_shape_adjoint (vertex-position gradients) */
                /* top-level geometry: its vertex positions are differentiated; instanced geometry: the instance's to_world (index + 1 in bits 8.. of the flags) */
                const uint32_t inst_bits = hs.y == 0xffffffffu ? 0u : (hs.y + 1u) << HAR_SHAPE_INST_SHIFT;
                geo.g0[islot] = make_float4(__uint_as_float(hs.x), hh.w, hh.y, hh.z);
                geo.g1[islot] = make_float4(d_in.x, d_in.y, d_in.z, __uint_as_float(alive ? Q.base + slot : HAR_SHAPE_NO_NEXT));
                geo.g2[islot] = make_float4(R.nee_p.x, R.nee_p.y, R.nee_p.z, __uint_as_float((R.item_ray ? R.nee_flags : (R.nee_flags & HAR_SHAPE_LIT)) | inst_bits));
                geo.g3[islot] = make_float4(R.nee_n.x, R.nee_n.y, R.nee_n.z, R.cos_em);
                geo.g4[islot] = make_float4(R.nee_w.x, R.nee_w.y, R.nee_w.z, 0.f);
                /* the previous vertex of the path (si.wi follows its motion, prb.py:128-140); none for the camera vertex */
                geo.g5[islot] = first_vertex ? make_float4(__uint_as_float(0xffffffffu), 0.f, 0.f, 0.f) : geo.pv0[lane];
                geo.g6[islot] = first_vertex ? make_float4(0.f, 0.f, 0.f, __uint_as_float(0xffffffffu)) : geo.pv1[lane];
            }
        }
        if (SHAPE && in_range && hh.x != HAR_INF) {
            geo.pv0[lane] = make_float4(__uint_as_float(hs.x), hh.w, hh.y, hh.z);
            geo.pv1[lane] = make_float4(d_in.x, d_in.y, d_in.z, __uint_as_float(hs.y));
        }
      }       /* rounds of the window */
    }
    if (emitter_grads && !fwd) {
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < 3 * min(S.n_emitters, (uint32_t) HAR_LDS_GRAD_EMITTERS); k += kBlock) {
            const float v = eacc[k];
            if (v != 0.f) atomicAdd(grad_slots + 3 * (size_t) S.n_bsdfs + k, v);
        }
    }
    if (INLINE) {
        __syncthreads();
        if (threadIdx.x == 0 && tq.nq && tq_gmax) atomicMax(tq.gmax, tq_gmax);
        for (uint32_t k = threadIdx.x; k < 3 * min(S.n_bsdfs + S.n_emitters, (uint32_t) HAR_LDS_GRAD_BSDFS); k += kBlock) {
            const float v = gacc[k];
            if (v != 0.f) atomicAdd(grad_slots + k, v);
        }
        if (EXTRA)
            for (uint32_t k = threadIdx.x; k < 15 * min(S.n_bsdfs, (uint32_t) HAR_LDS_EXTRA_BSDFS); k += kBlock) {
                const float v = xacc[k];
                if (v != 0.f) atomicAdd(grad_extra + k, v);
            }
    }
}

/* ------------------------------------------------------------------ commit */
/* Adjoint pass of the RECORD tape (TapeArrays): bounce b's vertices in the primal pass's slot order.  Per vertex: L <- L - emission met here - [visible] Lr_dir,
 * g = dL * ([visible] dLr_dir / d slot0 + L * (df / d slot0) / f) (prb.py:227,288-313) into the colour slot / the texel-gradient queues / the emitter's slot, then
 * L and dL move on to the survivor's slot of the next bounce.  No geometry, no sampling, no BSDF code: 133 B per vertex streamed. */
template <bool FIRST, bool COMPACT = false>
__global__ __launch_bounds__(kBlock) void k_commit(DScene S, uint32_t shard_cap, const uint32_t *count_in, TapeArrays tape, const uint8_t *vis,
                                                   float *grad_slots, float *const *grad_tex, TexelQueues tq, const float4 *result, const float4 *dL) {
    __shared__ uint32_t tq_hist[HAR_TQ_MAX], tq_base[HAR_TQ_MAX], tq_gmax;
    __shared__ float gacc[3 * HAR_LDS_GRAD_BSDFS];
    for (uint32_t k = threadIdx.x; k < 3 * HAR_LDS_GRAD_BSDFS; k += kBlock) gacc[k] = 0.f;
    if (threadIdx.x == 0) tq_gmax = 0u;
    __syncthreads();
    const ShardLoop Q(count_in, shard_cap);
    for (uint32_t tile = Q.first_tile(); tile * kBlock < Q.n; tile += Q.tile_step()) {
        const uint32_t local = tile * kBlock + threadIdx.x;
        const bool in_range = local < Q.n;
        const uint32_t i = Q.base + local;
        TexelRecord rec; rec.has = false;
        bool pred = false, visible = false, dirty = false;
        float4 s2 = make_float4(0.f, 0.f, 0.f, 0.f), s3 = s2, s4 = s2;
        Vec3 L(0.f), dl(0.f); uint32_t nx = HAR_TAPE_DEAD;
        if (in_range) {
            nx = tape.next[i];
            if (FIRST) {        /* bounce 0: slot i of shard s holds lane ((i / 256) * HAR_SHARDS + s) * 256 + i % 256 of the chunk (shard_slot); L = the primal result */
                const uint32_t lane = ((local / kBlock) * HAR_SHARDS + Q.shard) * kBlock + (local % kBlock);
                const float4 r = result[lane], d4 = dL[lane];
                L = Vec3(r.x, r.y, r.z); dl = Vec3(d4.x, d4.y, d4.z);
            } else {
                const float4 a = tape.la_in[i]; const float2 c2 = tape.lb_in[i];
                L = Vec3(a.x, a.y, a.z); dl = Vec3(a.w, c2.x, c2.y);
            }
            if (nx & HAR_TAPE_HAS_EM) { const float4 e = tape.rec_em[i]; L = Vec3(L.x - e.x, L.y - e.y, L.z - e.z); }
            pred = (nx & HAR_TAPE_HAS_REC) != 0u;
            if (pred) {
                s2 = tape.rec0[i]; s3 = tape.rec1[i];
                if (COMPACT) {
                    /* the 36-byte record of a diffuse vertex (k_shade<.., RECORD>): {d = dLr_dir / d rho [per unit radiance], tag}, {rho, u}, v  ->  the three vectors */
                    const float v = reinterpret_cast<const float *>(tape.rec2)[i];
                    const uint32_t tag = __float_as_uint(s2.w), emitter = (tag >> 20) & 0x7ffu;
                    const Vec3 d(s2.x, s2.y, s2.z), rho(s3.x, s3.y, s3.z);
                    Vec3 dfull = d;
                    if (emitter != HAR_ITEM_NO_EMITTER) { const float *rad = S.emitters[emitter].radiance; dfull = Vec3(d.x * rad[0], d.y * rad[1], d.z * rad[2]); }
                    const float u = s3.w;
                    s2 = make_float4(d.x * rho.x, d.y * rho.y, d.z * rho.z, s2.w);                     /* Lr_dir (per unit radiance when the emitter is differentiated) */
                    s3 = make_float4(dfull.x, dfull.y, dfull.z, u);
                    s4 = make_float4(rho.x != 0.f ? 1.f / rho.x : 0.f, rho.y != 0.f ? 1.f / rho.y : 0.f, rho.z != 0.f ? 1.f / rho.z : 0.f, v);      /* used under the record's indirect-term flag only */
                } else s4 = tape.rec2[i];
                visible = (nx & HAR_TAPE_HAS_RAY) != 0u && vis[i] != 0;
            }
        }
        adjoint_commit_regs(S, pred, visible, s2, s3, s4, L, dirty, dl, grad_slots, grad_tex, gacc, tq.nq ? &tq : nullptr, &rec);
        if (in_range && (nx & HAR_TAPE_DEAD) != HAR_TAPE_DEAD) {
            const uint32_t nslot = nx & HAR_TAPE_DEAD;
            tape.la_out[nslot] = make_float4(L.x, L.y, L.z, dl.x); tape.lb_out[nslot] = make_float2(dl.y, dl.z);
        }
        /* the block's texel records -> their band queues (a per-WAVE append -- ballot ranks, one global atomic per wave and band, no barriers -- was measured 3 % slower
         * on the whole PRB step: four times the returning global atomics; profiles/r04_ab_prb_commit.txt) */
        if (tq.nq) texel_queue_append(S, tq, Q.shard, rec, tq_hist, tq_base, &tq_gmax, grad_tex, grad_slots);
    }
    __syncthreads();
    if (threadIdx.x == 0 && tq.nq && tq_gmax) atomicMax(tq.gmax, tq_gmax);
    for (uint32_t k = threadIdx.x; k < 3 * min(S.n_bsdfs + S.n_emitters, (uint32_t) HAR_LDS_GRAD_BSDFS); k += kBlock) {
        const float v = gacc[k];
        if (v != 0.f) atomicAdd(grad_slots + k, v);
    }
}

/* texel-gradient queues -> gradient textures (see TexelQueues): blockIdx / bpq = shard * nq + band queue; the queue's records are shared by bpq blocks.
 * FIXED = false: the float-atomic version (fallback for launches with a non-finite gradient, and HAR_TQ_FIXED=0 for A/B); its LDS copy has the same footprint. */
template <bool FIXED>
__global__ __launch_bounds__(kBlock) void k_texel_accumulate(TexelQueues tq, float *const *grad_tex, uint32_t bpq, uint32_t spread, uint32_t force_float) {
    extern __shared__ unsigned long long band64[];
    float *band = reinterpret_cast<float *>(band64);
    const uint32_t gbits = *tq.gmax;
    if (FIXED != (!force_float && gbits < 0x7f800000u)) return;            /* both instantiations are launched; exactly one of them runs (no host round trip for gmax) */
    const uint32_t qid = blockIdx.x / bpq, part = blockIdx.x - qid * bpq, q = qid % tq.nq;
    const uint32_t n = min(tq.count[(size_t) qid * HAR_COUNTER_STRIDE], tq.cap);
    if (n == 0 || gbits == 0u) return;
    const uint4 info = tq.qinfo[q];                        /* texture, first row, rows, width */
    const uint32_t W = info.w, row0 = info.y, rows = info.z, H = tq.qinfo[q + tq.nq].x;
    /* the LDS copy holds the band's rows PLUS the row after it (the second bilinear row of cells in the band's last row; row 0 after the
     * texture's last row: repeat wrap), so that every tap of every record is an LDS atomic -- a global atomic in this loop would make each
     * iteration wait for a memory-side round trip */
    const uint32_t nfl = (rows + 1u) * W * 3u;
    float *dst = grad_tex[info.x];
    for (uint32_t k = threadIdx.x; k < nfl; k += kBlock) { if (FIXED) band64[k] = 0ull; else band[k] = 0.f; }
    __syncthreads();
    const uint32_t b = (uint32_t) ((uint64_t) n * part / bpq), e = (uint32_t) ((uint64_t) n * (part + 1) / bpq);
    const float4 *rec = tq.rec + 2 * (size_t) qid * tq.cap;
    /* fixed point: |sum of any accumulator| <= (e - b) * gmax < 2^(en + eg), scaled to stay below 2^62 */
    int eg = 0; (void) frexpf(__uint_as_float(gbits), &eg);
    const int en = 32 - __clz((int) max(e - b, 1u));
    const double scale = FIXED ? ldexp(1.0, 62 - en - eg) : 1.0;
    auto fixed = [&](float v) { return (unsigned long long) __double2ll_rn((double) v * scale); };
    /* spread = 1: every group of four lanes walks its own contiguous segment of the block's share (one 128-byte line per group and step) instead of the wave
     * walking 64 consecutive records; measured neutral for the float atomics (they cost 193 cycles whatever the addresses), kept as a switch (HAR_TQ_SPREAD) */
    const uint32_t len = e - b, per = spread ? (((len + kBlock / 4u - 1u) / (kBlock / 4u) + 3u) & ~3u) : 0u;
    const uint32_t first = spread ? b + (threadIdx.x >> 2) * per + (threadIdx.x & 3u) : b + threadIdx.x;
    const uint32_t last = spread ? min(e, b + ((threadIdx.x >> 2) + 1u) * per) : e, step = spread ? 4u : kBlock;
    const uint32_t trips = spread ? (per + 3u) / 4u : (len + kBlock - 1u) / kBlock;           /* uniform trip count: the wave-level steps below need every lane */
    uint32_t i = first;
    for (uint32_t trip = 0; trip < trips; ++trip, i += step) {
        bool active = i < last;
        float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0;
        if (active) { r0 = rec[2 * (size_t) i]; r1 = rec[2 * (size_t) i + 1]; }
        const uint32_t cell = __float_as_uint(r0.x);
        const float w0x = 1.f - r0.z, w0y = 1.f - r0.w;
        const float w[4] = { w0x * w0y, r0.z * w0y, w0x * r0.w, r0.z * r0.w };
        float p[12];
        for (int t = 0; t < 4; ++t) { p[3 * t] = r1.x * w[t]; p[3 * t + 1] = r1.y * w[t]; p[3 * t + 2] = r1.z * w[t]; }
        auto commit = [&](uint32_t c, const float *v) {
            const uint32_t x0 = c & 0xffffu, ry = (c >> 16) - row0, x1 = x0 + 1 == W ? 0u : x0 + 1;
            const uint32_t a[4] = { 3u * (ry * W + x0), 3u * (ry * W + x1), 3u * (ry * W + x0) + 3u * W, 3u * (ry * W + x1) + 3u * W };
            for (int t = 0; t < 4; ++t)
                for (int ch = 0; ch < 3; ++ch) {
                    if (FIXED) atomicAdd(&band64[a[t] + ch], fixed(v[3 * t + ch]));
                    else atomicAdd(&band[a[t] + ch], v[3 * t + ch]);
                }
        };
        /* the samples of one pixel sit side by side in the queue and fall into the same cell: sum such a run on the SIMD (DPP) and let one lane commit it --
         * 64 same-address LDS atomics serialise (2 cycles per lane even for integers) */
        for (int round = 0; round < 2; ++round) {
            const uint64_t m = __ballot(active);
            if (m == 0) break;
            const uint32_t key = __shfl(cell, __ffsll((long long) m) - 1, 64);
            const bool match = active && cell == key;
            if (__popcll(__ballot(match)) < 16) break;
            float sum[12];
            for (int k = 0; k < 12; ++k) sum[k] = wave_sum_to_last(match ? p[k] : 0.f);
            if ((threadIdx.x & 63u) == 63u) commit(key, sum);
            active = active && !match;
        }
        if (active) commit(cell, p);
    }
    __syncthreads();
    const double inv = 1.0 / scale;
    for (uint32_t k = threadIdx.x; k < nfl; k += kBlock) {
        float v;
        if (FIXED) { const long long acc = (long long) band64[k]; if (acc == 0) continue; v = (float) ((double) acc * inv); }
        else { v = band[k]; if (v == 0.f) continue; }
        const uint32_t r = k / (3u * W), c = k - r * 3u * W;
        uint32_t y = row0 + r; if (y >= H) y -= H;
        atomicAdd(dst + 3 * (size_t) y * W + c, v);
    }
}

/* adjoint resolve of a bounce whose shadow-ray results sit in the replay cache: no traversal, one item per thread */
template <bool FWD>
__global__ __launch_bounds__(kBlock) void k_resolve_adjoint_cached(DScene S, const uint32_t *item_count, uint32_t shard_cap, ItemArrays items, float4 *result,
                                                                   const float4 *dL, float *grad_refl, float *const *grad_tex, ReplayCache rc, uint8_t *item_vis, TexelQueues tq) {
    __shared__ float gacc[3 * HAR_LDS_GRAD_BSDFS];
    /* tq.nq != 0 (reverse mode): the texel gradients go through the band queues like in the in-place commit -- the item path is what vertex-position gradients
     * run on, and with direct atomics a small albedo texture (every path of the chip adds to the same few texels) made this kernel 80 % of such a step */
    __shared__ uint32_t tq_hist[FWD ? 1 : HAR_TQ_MAX], tq_base[FWD ? 1 : HAR_TQ_MAX], tq_gmax;
    for (uint32_t k = threadIdx.x; k < 3 * HAR_LDS_GRAD_BSDFS; k += kBlock) gacc[k] = 0.f;
    if (threadIdx.x == 0) tq_gmax = 0u;
    __syncthreads();
    const ShardLoop Q(item_count, shard_cap);
    for (uint32_t tile = Q.first_tile(); tile * kBlock < Q.n; tile += Q.tile_step()) {
        const uint32_t local = tile * kBlock + threadIdx.x;
        const bool pred = local < Q.n;
        const uint32_t i = Q.base + (pred ? local : 0u);
        bool visible = false;
        if (pred && items.s0[i].w >= 0.f) visible = rc.vis[__float_as_uint(items.s1[i].w)] != 0;
        TexelRecord rec; rec.has = false;
        const bool queued = !FWD && tq.nq != 0;
        adjoint_commit<FWD>(S, items, i, pred, visible, result, dL, grad_refl, grad_tex, gacc, item_vis, queued ? &tq : nullptr, &rec);
        if (queued) texel_queue_append(S, tq, Q.shard, rec, tq_hist, tq_base, &tq_gmax, grad_tex, grad_refl);
    }
    __syncthreads();
    if (FWD) return;
    if (threadIdx.x == 0 && tq.nq && tq_gmax) atomicMax(tq.gmax, tq_gmax);
    for (uint32_t k = threadIdx.x; k < 3 * min(S.n_bsdfs + S.n_emitters, (uint32_t) HAR_LDS_GRAD_BSDFS); k += kBlock) {
        const float v = gacc[k];
        if (v != 0.f) atomicAdd(grad_refl + k, v);
    }
}

/* ------------------------------------------------- resolve (shadow rays + NEE) */
template <int MODE, bool SPILL, bool FWD = false, bool FLAT = false>
__global__ __launch_bounds__(kBlock, (MODE == MODE_PRB_ADJOINT ? 1 : HAR_TRACE_MIN_WAVES)) void k_resolve(    /* the adjoint flavour keeps gradient accumulators in LDS: 5 blocks per CU */
                                                    DScene S, const uint32_t *item_count, uint32_t *cursor, uint32_t shard_cap, ItemArrays items, float4 *result,
                                                    const float4 *dL, float *grad_refl, float *const *grad_tex, int *status, ReplayCache rc, uint2 *spill, uint8_t *item_vis) {
    __shared__ uint2 lds[HAR_LDS_STACK_SMALL * kBlock];
    /* adjoint: per-block accumulators of the constant-albedo gradients.  Every path of the chip adds to the same
     * few floats of grad_refl (one 64 B line): direct global atomics serialise at ~88 atomics/us per line, which
     * made the adjoint 10x slower than the primal pass.  ds_add_f32 here, one global atomic per block and entry. */
    __shared__ float gacc[MODE == MODE_PRB_ADJOINT ? 3 * HAR_LDS_GRAD_BSDFS : 1];
    typedef typename WaveStackOf<SPILL>::type WaveStack;
    typedef Traversal<HAR_TRAV_POLICY, FLAT> Trav;
    WaveStack stack = make_wave_stack<SPILL>(lds, spill);
    const uint32_t shard = blockIdx.x & (HAR_SHARDS - 1), n = item_count[shard * HAR_COUNTER_STRIDE], base = shard * shard_cap;
    if (n == 0) return;
    if (MODE == MODE_PRB_ADJOINT) {
        for (uint32_t k = threadIdx.x; k < 3 * HAR_LDS_GRAD_BSDFS; k += kBlock) gacc[k] = 0.f;
        __syncthreads();
    }
    auto take = [&](uint32_t idx, Trav &T) {
        /* branch-free (trace_persistent's refill merges the state with selects): an item without a shadow ray (maxt < 0) begins a state nobody steps */
        const float4 s0 = items.s0[base + idx], s1 = items.s1[base + idx];
        T.begin(S.accel, Vec3(s0.x, s0.y, s0.z), Vec3(s1.x, s1.y, s1.z), s0.w, (S.accel.top_last & 1u) != 0u);
        return s0.w >= 0.f;
    };
    if (MODE == MODE_PATH || MODE == MODE_PRB_PRIMAL) {
        /* forward: an unoccluded item adds its contribution to its lane's radiance (one item per lane and bounce: no race) */
        auto commit = [&](uint32_t idx, const Trav &T) {
            const uint32_t i = base + idx;
            if (rc.mode == 1) rc.vis[__float_as_uint(items.s1[i].w)] = T.found ? 0 : 1;      /* replay cache: per lane */
            else if (rc.mode == 3 || rc.mode == 5) rc.vis[__float_as_uint(items.s2[i].w)] = T.found ? 0 : 1; /* replay / record tape: per vertex slot */
            if (!T.found) {
#if HAR_RESOLVE_ATOMIC
                /* one item per lane and bounce: the add has a single writer, so a no-return atomic gives the bits of load + add + store -- without the load's round
                 * trip in the middle of the wave's refill (the commit used to be a chain of THREE dependent loads: s1.w -> result[lane], s2) and without reading
                 * `result` into the CU at all (the add happens at the L2) */
                const float4 s2 = items.s2[i];
                const uint32_t lane = (rc.mode == 0 || rc.mode == 1) ? __float_as_uint(s2.w) : __float_as_uint(items.s1[i].w);
                float *r = reinterpret_cast<float *>(result + lane);
                atomicAdd(r, s2.x); atomicAdd(r + 1, s2.y); atomicAdd(r + 2, s2.z);
#else
                const float4 s2 = items.s2[i];
                const uint32_t lane = (HAR_RESOLVE_LANE_S2 && (rc.mode == 0 || rc.mode == 1)) ? __float_as_uint(s2.w) : __float_as_uint(items.s1[i].w);
                const float4 r = result[lane];
                result[lane] = make_float4(r.x + s2.x, r.y + s2.y, r.z + s2.z, 0.f);
#endif
            }
        };
#if HAR_RESOLVE_RETIRE
        /* results are committed at refill time by all the lanes that finished since the last refill -- one store instruction for several lanes instead of one per lane in
         * the step it finishes in (as the closest-hit kernel does, HAR_CLOSEST_RETIRE): the record pass of prb stores a visibility byte for EVERY item */
        trace_persistent<true, true, WaveStack, FLAT>(S.accel, cursor + shard * HAR_COUNTER_STRIDE, n, stack, status, take,
            [&](uint32_t, const Trav &) { },
            [&](bool pred, uint32_t idx, const Trav &T) { if (pred) commit(idx, T); });
#else
        trace_persistent<true, false, WaveStack, FLAT>(S.accel, cursor + shard * HAR_COUNTER_STRIDE, n, stack, status, take, commit,
            [&](bool, uint32_t, const Trav &) { });
#endif
    } else {
        /* adjoint: L <- L - Lr_dir; g = dL * (dLr_dir/drho + [bsdf_val != 0] L / rho)  (prb.py:227,288-313);
         * gradients are committed at refill time by ALL lanes so that the wave pre-reduction can run */
        trace_persistent<true, true, WaveStack, FLAT>(S.accel, cursor + shard * HAR_COUNTER_STRIDE, n, stack, status, take,
            [&](uint32_t, const Trav &) { },
            [&](bool pred, uint32_t idx, const Trav &T) {
                adjoint_commit<FWD>(S, items, base + (pred ? idx : 0u), pred, pred && !T.found, result, dL, grad_refl, grad_tex, gacc, item_vis);
            });
        __syncthreads();
        if (FWD) return;
        for (uint32_t k = threadIdx.x; k < 3 * min(S.n_bsdfs + S.n_emitters, (uint32_t) HAR_LDS_GRAD_BSDFS); k += kBlock) {
            const float v = gacc[k];
            if (v != 0.f) atomicAdd(grad_refl + k, v);
        }
    }
}

/* alpha channel of `rgba` films: 1 for a valid camera sample.  PathIntegrator: valid_ray = the environment is visible or the path met a surface
 * (path.cpp:114-115,307-308,341) -- the camera ray's intersection decides; prb: depth != 0 (prb.py:332).  Runs on the camera rays' hits (after
 * hide_emitters' skipping), one float per lane of the chunk. */
__global__ __launch_bounds__(kBlock) void k_alpha_flags(uint32_t shard_cap, const uint32_t *count_in, const float4 *state_a3, const float4 *h0, uint32_t lane_base,
                                                        float miss_value, float *alpha) {
    const ShardLoop Q(count_in, shard_cap);
    for (uint32_t tile = Q.first_tile(); tile * kBlock < Q.n; tile += Q.tile_step()) {
        const uint32_t local = tile * kBlock + threadIdx.x;
        if (local >= Q.n) continue;
        const uint32_t i = Q.base + local;
        alpha[__float_as_uint(state_a3[i].w) - lane_base] = h0[HIT0(i)].x != HAR_INF ? 1.f : miss_value;
    }
}

/* ------------------------------------------------------- hide_emitters */
/* Integrator::skip_area_emitters (src/render/integrator.cpp:96-124; call sites path.cpp:177-190, prb.py:112-118): a CAMERA ray that hits an area
 * emitter continues through all area emitters along it.  Only the preliminary intersection of the lane is replaced, its ray stays the camera ray.
 * One round: (first = 1) scan the wavefront's camera-ray hits, (first = 0) take the re-traced hits of list `src`, store them as the lanes' hits;
 * every hit that is an emitter again appends a continuation ray (origin offset along the geometric normal, same direction; .w of the direction =
 * the lane's slot) to list `dst`.  The host loops trace(list) -> round until the list is empty. */
__global__ __launch_bounds__(kBlock) void k_skip_emitters(DScene S, int first, uint32_t shard_cap, const uint32_t *count_in, const float4 *ray_d,
                                                          const float4 *hit0, const uint2 *hit1, float4 *h0, uint2 *h1, float4 *dst_o, float4 *dst_d, uint32_t *dst_count) {
    __shared__ uint32_t lds_r[12];
    const ShardLoop Q(count_in, shard_cap);
    uint32_t *cnt = dst_count + Q.shard * HAR_COUNTER_STRIDE;
    for (uint32_t tile = Q.first_tile(); tile * kBlock < Q.n; tile += Q.tile_step()) {
        const uint32_t local = tile * kBlock + threadIdx.x;
        const bool in_range = local < Q.n;
        const uint32_t i = Q.base + local;
        bool again = false; float4 no = make_float4(0.f, 0.f, 0.f, 0.f), nd = no;
        if (in_range) {
            const float4 d = ray_d[i];
            const float4 hh = first ? h0[HIT0(i)] : hit0[HIT0(i)]; const uint2 hs = first ? h1[HIT1(i)] : hit1[HIT1(i)];
            const uint32_t slot = first ? i : __float_as_uint(d.w);
            if (!first) {
                h0[HIT0(slot)] = hh;
#if HAR_HIT_MATINFO && HAR_HIT_INTERLEAVED
                MeshInfo mi{ 0u, 0u };
                if (hh.x != HAR_INF) mi = S.accel.mesh_info[hs.x];
                *reinterpret_cast<uint4 *>(h1 + HIT1(slot)) = make_uint4(hs.x, hs.y, mi.foff + __float_as_uint(hh.w), mi.matinfo);
#else
                h1[HIT1(slot)] = hs;
#endif
            }
            if (hh.x != HAR_INF && S.meshes[hs.x].emitter >= 0) {
                const Vec3 dir(d.x, d.y, d.z);
                const SurfInt si = compute_si(S, dir, hh.x, hh.y, hh.z, __float_as_uint(hh.w), hs.x, hs.y);
                const Vec3 p = offset_p(si, dir);
                no = make_float4(p.x, p.y, p.z, -1.f);                      /* .w < 0: unbounded ray (see store_state) */
                nd = make_float4(d.x, d.y, d.z, __uint_as_float(slot));
                again = true;
            }
        }
        uint32_t slot_a, slot_b;
        block_reserve2(cnt, again, cnt, false, lds_r, slot_a, slot_b);
        if (again) { dst_o[Q.base + slot_a] = no; dst_d[Q.base + slot_a] = nd; }
    }
}

/* ------------------------------------------------------- vertex-position gradients */
/* One thread per adjoint item of a bounce: rebuild the vertex (triangle, barycentrics, incoming direction, the previous vertex), look up the lane's next
 * interaction (detached: prb.py:263-266 computes it outside dr.resume_grad) and apply har_shape_grad.h.  A vertex adds to its own triangle / instance and,
 * through the attached si.wi, to the previous vertex's.  Scenes with few differentiated vertices (a Cornell box has a few dozen) would serialise on a
 * handful of cache lines -- every path of the chip adds to the same vertices -- so those accumulate in LDS and flush once per block; large meshes scatter
 * with global atomics. */
__global__ __launch_bounds__(kBlock, HAR_SHAPE_MIN_WAVES) void k_shape_adjoint(DScene S, const uint32_t *item_count, uint32_t shard_cap, ItemArrays items, ShapeArrays geo, const float4 *result,
                                                          const float4 *dL, int has_next, WaveState next, const float4 *h0, const uint2 *h1, ReplayCache rc, ShapeTargets T) {
    /* vertex gradients: a direct-mapped LDS cache of HAR_LDS_GRAD_VERTS vertices per block (wave_cached_add3) -- every path of the chip adds to the few hundred
     * vertices in view, whatever the size of the mesh; `nacc`: the adjoints of their vertex normals (meshes with vertex normals), under the same tags */
    __shared__ float acc[3 * HAR_LDS_GRAD_VERTS], nacc[3 * HAR_LDS_GRAD_VERTS];
    __shared__ uint32_t vtag[HAR_LDS_GRAD_VERTS];
    /* instance transforms: every path of the chip that meets instance i adds to the same 12 floats -- per-block accumulators for the first
     * HAR_LDS_GRAD_INSTS slots (same-line global atomics serialise at ~88 per microsecond), global atomics beyond */
    __shared__ float iacc[12 * HAR_LDS_GRAD_INSTS];
    const bool lds = T.n_verts != 0;
    if (lds) { for (uint32_t k = threadIdx.x; k < 3 * HAR_LDS_GRAD_VERTS; k += kBlock) { acc[k] = 0.f; nacc[k] = 0.f; } for (uint32_t k = threadIdx.x; k < HAR_LDS_GRAD_VERTS; k += kBlock) vtag[k] = 0xffffffffu; }
    if (T.inst_grad) { for (uint32_t k = threadIdx.x; k < 12 * HAR_LDS_GRAD_INSTS; k += kBlock) iacc[k] = 0.f; }
    __syncthreads();
    /* slot of a vertex's geometry in the gradient buffers, or -1: top-level mesh -> first vertex, instance -> its 12 floats */
    auto target = [&](uint32_t shape, uint32_t inst) -> int32_t {
        if (shape == 0xffffffffu) return -1;
        return inst == 0xffffffffu ? (T.offset ? T.offset[shape] : -1) : (T.inst_slot ? T.inst_slot[inst] : -1);
    };
    /* The scatters below are reached by ALL lanes of a wave (a lane without work passes on = false), because they pre-reduce: the 64 samples of a pixel meet
     * the same triangle at the camera vertex, i.e. the same three vertices -- 64 same-address atomics per float serialise (global: ~88 per microsecond on one line,
     * LDS: one lane per 3 cycles), and a smooth floor seen from above made this kernel 85 % of a shape-gradient step.  Lanes that target the same vertex are summed
     * with DPP and one of them adds (wave_slot_add3 for the block's LDS copy, wave_aggregated_add3 for global memory). */
    auto add_verts = [&](bool on, int32_t off, const uint32_t vid[3], const Vec3 g[3], const Vec3 *gn) {
        if (!__ballot(on)) return;
        const bool normals = T.grad_nrm != nullptr;
        for (int k = 0; k < 3; ++k) {
            const uint32_t v = on ? (uint32_t) off + vid[k] : 0u;
            wave_cached_add3(acc, normals ? nacc : nullptr, vtag, HAR_LDS_GRAD_VERTS, v, on ? g[k] : Vec3(0.f), (on && gn) ? gn[k] : Vec3(0.f), on, T.grad, normals ? T.grad_nrm : nullptr);
        }
    };
    auto add_inst = [&](bool on, int32_t off, const float gM[12]) {
        if (!__ballot(on)) return;
        const bool in_lds = on && (uint32_t) off < HAR_LDS_GRAD_INSTS;
        for (int c = 0; c < 4; ++c) {
            const Vec3 col = on ? Vec3(gM[3 * c], gM[3 * c + 1], gM[3 * c + 2]) : Vec3(0.f);
            if (__ballot(in_lds)) wave_slot_add3(iacc, in_lds ? 4u * (uint32_t) off + (uint32_t) c : 0u, col, in_lds);
            if (__ballot(on && !in_lds)) wave_aggregated_add3(T.inst_grad + (on ? 12 * (size_t) off + 3 * c : 0), col, on && !in_lds);
        }
    };
    const ShardLoop Q(item_count, shard_cap);
    for (uint32_t tile = Q.first_tile(); tile * kBlock < Q.n; tile += Q.tile_step()) {
        const uint32_t local = tile * kBlock + threadIdx.x;
        ShapeGrad G; G.self_mesh = G.self_inst = G.prev_mesh = G.prev_inst = G.self_normals = false;
        int32_t off = -1, poff = -1;
        if (local < Q.n) {
            const uint32_t i = Q.base + local;
            const float4 g0 = geo.g0[i];
            const uint32_t shape = __float_as_uint(g0.x);
            if (shape != 0xffffffffu) {
                const float4 g2 = geo.g2[i], g5 = geo.g5[i], g6 = geo.g6[i];
                ShapeItem it;
                it.shape = shape; it.prim = __float_as_uint(g0.y); it.b1 = g0.z; it.b2 = g0.w;
                it.inst = (__float_as_uint(g2.w) >> HAR_SHAPE_INST_SHIFT) - 1u;       /* 0xffffffff: top-level geometry */
                it.prev_shape = __float_as_uint(g5.x); it.prev_prim = __float_as_uint(g5.y); it.prev_b1 = g5.z; it.prev_b2 = g5.w;
                it.prev_d = Vec3(g6.x, g6.y, g6.z); it.prev_inst = __float_as_uint(g6.w);
                /* a vertex on an instance moves with the instance's to_world (inst_slot) or with the NESTED MESH of its shape group (offset[nested mesh]); never both
                 * (har_integrator_set_grad_positions / _instances refuse the combination, as instance.cpp:162-166 does) */
                auto nested = [&](uint32_t shp, uint32_t inst) -> int32_t { return (shp != 0xffffffffu && inst != 0xffffffffu && T.offset) ? T.offset[shp] : -1; };
                const int32_t noff = nested(it.shape, it.inst), npoff = nested(it.prev_shape, it.prev_inst);
                off = noff >= 0 ? noff : target(it.shape, it.inst); poff = npoff >= 0 ? npoff : target(it.prev_shape, it.prev_inst);
                if (off >= 0 || poff >= 0) {
                    const float4 g1 = geo.g1[i], g3 = geo.g3[i], g4 = geo.g4[i];
                    it.d_in = Vec3(g1.x, g1.y, g1.z); it.next_slot = __float_as_uint(g1.w);
                    it.q = Vec3(g2.x, g2.y, g2.z); it.nee_flags = __float_as_uint(g2.w) & ((1u << HAR_SHAPE_INST_SHIFT) - 1u);
                    it.n_e = Vec3(g3.x, g3.y, g3.z); it.W = Vec3(g4.x, g4.y, g4.z);
                    const uint32_t lane = __float_as_uint(items.s1[i].w);
                    const float4 L4 = result[lane], dl4 = dL[lane];
                    bool nxt = has_next && it.next_slot != HAR_SHAPE_NO_NEXT, next_valid = false;
                    Vec3 np(0.f), nn(0.f), nd(0.f);
                    if (nxt) {
                        const float4 a1 = next.a1[it.next_slot];
                        nd = Vec3(a1.x, a1.y, a1.z);
                        float4 hh; uint2 hs;
                        if (rc.mode == 2) { hh = rc.h0[lane]; hs = rc.h1[lane]; } else { hh = h0[HIT0(it.next_slot)]; hs = h1[HIT1(it.next_slot)]; }
                        next_valid = hh.x != HAR_INF;
                        if (next_valid) { const SurfInt sn = compute_si(S, nd, hh.x, hh.y, hh.z, __float_as_uint(hh.w), hs.x, hs.y); np = sn.p; nn = sn.n; }
                    }
                    /* the emitter sample: w_em = ds.d (surface emitters: normalize(ds.p - si.p), recomputed from the interpolated point) */
                    it.w_em = it.q;
                    if (it.nee_flags & HAR_SHAPE_NEE_AT_POINT) { const SurfInt si = compute_si(S, it.d_in, 0.f, it.b1, it.b2, it.prim, it.shape, it.inst); it.w_em = normalize3(it.q - si.p); }
                    if (!shape_item_adjoint(S, it, off >= 0, poff >= 0, geo.vis[i] != 0, Vec3(L4.x, L4.y, L4.z), Vec3(dl4.x, dl4.y, dl4.z), nxt, next_valid, np, nn, nd, G, noff >= 0, npoff >= 0))
                        G.self_mesh = G.self_inst = G.prev_mesh = G.prev_inst = G.self_normals = false;
                }
            }
        }
        add_verts(G.self_mesh, off, G.vid, G.g, G.self_normals ? G.gn : nullptr);
        add_verts(G.prev_mesh, poff, G.pvid, G.gp, nullptr);
        if (T.inst_grad) { add_inst(G.self_inst, off, G.gM); add_inst(G.prev_inst, poff, G.gpM); }
    }
    __syncthreads();
    if (lds)
        for (uint32_t k = threadIdx.x; k < 3 * HAR_LDS_GRAD_VERTS; k += kBlock) {
            const uint32_t tag = vtag[k / 3u];
            if (tag == 0xffffffffu) continue;
            const float v = acc[k], w = nacc[k];
            if (v != 0.f) atomicAdd(T.grad + 3 * (size_t) tag + k % 3u, v);
            if (w != 0.f && T.grad_nrm) atomicAdd(T.grad_nrm + 3 * (size_t) tag + k % 3u, w);
        }
    if (T.inst_grad) for (uint32_t k = threadIdx.x; k < 12 * min(T.n_insts, (uint32_t) HAR_LDS_GRAD_INSTS); k += kBlock) { const float v = iacc[k]; if (v != 0.f) atomicAdd(T.inst_grad + k, v); }
}

/* regenerated vertex normals, second stage (har_shape_grad.h face_normals_adjoint): one thread per face of the mesh */
__device__ __forceinline__ void normals_face(const DScene &S, const DMesh &M, uint32_t f, uint32_t vid[3], Vec3 P[3]) {
    const uint32_t *fi = S.faces + 4 * (size_t) (M.foff + f);
    for (int k = 0; k < 3; ++k) { vid[k] = fi[k]; const float *r = S.verts + 8 * (size_t) (M.voff + fi[k]); P[k] = Vec3(r[0], r[1], r[2]); }
}
__global__ void k_normals_sums(DScene S, uint32_t mesh, float *acc) {
    const DMesh M = S.meshes[mesh];
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= M.face_count) return;
    uint32_t vid[3]; Vec3 P[3], c[3]; normals_face(S, M, f, vid, P);
    if (!face_corner_normals(P, c)) return;
    for (int k = 0; k < 3; ++k) { float *q = acc + 3 * (size_t) vid[k]; atomicAdd(q, c[k].x); atomicAdd(q + 1, c[k].y); atomicAdd(q + 2, c[k].z); }
}
__global__ void k_normals_adjoint(DScene S, uint32_t mesh, const float *acc, const float *nbar, float *grad) {
    const DMesh M = S.meshes[mesh];
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= M.face_count) return;
    uint32_t vid[3]; Vec3 P[3], ab[3]; normals_face(S, M, f, vid, P);
    bool any = false;
    for (int k = 0; k < 3; ++k) {
        const float *a = acc + 3 * (size_t) vid[k], *b = nbar + 3 * (size_t) vid[k];
        const Vec3 av(a[0], a[1], a[2]), nb(b[0], b[1], b[2]);
        const float l2 = dot3(av, av);
        ab[k] = Vec3(0.f);
        if (!(l2 > 0.f)) continue;
        const float il = rsqrt_(l2); const Vec3 n = av * il;
        ab[k] = (nb - n * dot3(n, nb)) * il;               /* n_v = acc_v / |acc_v| */
        any = any || nb.x != 0.f || nb.y != 0.f || nb.z != 0.f;
    }
    if (!any) return;
    Vec3 g[3] = { Vec3(0.f), Vec3(0.f), Vec3(0.f) };
    face_normals_adjoint(P, ab, g);
    for (int k = 0; k < 3; ++k) { float *q = grad + 3 * (size_t) vid[k]; atomicAdd(q, g[k].x); atomicAdd(q + 1, g[k].y); atomicAdd(q + 2, g[k].z); }
}

/* ------------------------------------------------------------------- splat */
/* ImageBlock::put (coalesced JIT branch, imageblock.cpp:444-520) for the 256 consecutive lanes of a block, as a GATHER in LDS.
 * Lanes are ordered by pixel (lane = pixel * spp + sample), so a block covers a short run of pixels of one image row.  Every lane
 * stages its value and its separable filter weights in LDS; then each thread owns one pixel of the block's tile and sums the lanes
 * whose footprint covers it -- a contiguous lane range, found by binary search on the footprint origins -- with four fma per lane
 * (short ranges are split over several threads).  No per-lane atomics: one LDS add per partial sum, one global atomic per tile
 * float.  The previous formulation (a DPP wave reduction per tap and channel when a wave shares one footprint, per-lane LDS atomics
 * otherwise) cost 5.8 ms per 67 M lanes at >= 64 spp and 45 ms below; this one is spp-independent.
 * Blocks whose lanes span two image rows (only when width * spp is not a multiple of 256) or whose tile exceeds the LDS budget
 * take the per-lane path. */
#define HAR_SPLAT_GATHER_MAX_SPLIT 8
#define HAR_SPLAT_TILE_PIXELS 512           /* LDS tile of the gather: 8 KB; wider tiles are processed in column slabs */
/* WONLY: the weight pass of render_backward (har_render_weights) -- every lane's value is (0, 0, 0, 1), so the gather only multiplies weights */
/* WONLY = 2: the alpha channel of an `rgba` film -- the lane's value is (0, 0, 0, scalar[i]) (har_integrator_set_alpha_film) */
template <int TAPS, int WONLY>
__global__ __launch_bounds__(kBlock) void k_splat(DSensor C, uint32_t seed, uint32_t spp, uint32_t log_spp, uint32_t lane_base, uint32_t n,
                                                  const float4 *result, int weights_only, float *film, const float2 *jitter, const float *scalar) {
    __shared__ float tile[4 * HAR_SPLAT_TILE_PIXELS];
    __shared__ float4 s_val[WONLY ? 1 : kBlock];
    __shared__ float s_sv[WONLY == 2 ? kBlock : 1];
    __shared__ float s_wx[TAPS][kBlock + 1], s_wy[TAPS][kBlock + 1];        /* + 1: threads of a wave read different taps of the same lane -> different banks */
    __shared__ int ext[6];                  /* footprint origin of the first / last active lane of the block, footprint size */
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    const bool act = i < n;
    const uint32_t n_act_u = min((uint32_t) kBlock, n - blockIdx.x * kBlock);
    Footprint F; F.count = 0; F.x0 = 0; F.y0 = 0;
    for (int k = 0; k < HAR_MAX_FILTER_TAPS; ++k) { F.wx[k] = 0.f; F.wy[k] = 0.f; }
    float val[4] = { 0.f, 0.f, 0.f, 1.f };
    /* `shifted`: the footprint does not start at (pixel - n).  pos = ipos + jitter is a float32 addition: for jitter within half an ulp of 1 it rounds
     * up to ipos + 1 (about 1e-5 of the samples of a 512-wide film), and ImageBlock::put (imageblock.cpp:444-470) -- like film_footprint -- takes the
     * footprint from floor(pos).  The gather below knows the lanes of a tile column only through their PIXEL, so such a lane is left out of it
     * (zero weights in LDS) and scatters its taps itself, as the reference's put() does. */
    bool shifted = false;
    if (act) {
        LaneSample ls;
        if (jitter) { const float2 j = jitter[i]; ls = lane_sample(C, lane_base + i, spp, log_spp, j.x, j.y); }
        else ls = lane_film_pos(C, seed, spp, log_spp, lane_base + i);
        film_footprint(C, ls, F);
        const uint32_t nh = (F.count - 1u) / 2u;
        const uint32_t px0 = (uint32_t) ((int32_t) ls.ipos_x - (int32_t) nh - (int32_t) C.crop_x), py0 = (uint32_t) ((int32_t) ls.ipos_y - (int32_t) nh - (int32_t) C.crop_y);
        shifted = F.x0 != px0 || F.y0 != py0;
        if (!weights_only) { float4 r = result[i]; val[0] = r.x; val[1] = r.y; val[2] = r.z; }
        if (WONLY == 2) val[3] = scalar[i];
        /* lanes are ordered by pixel: the block's extent follows from the pixels of its first and last active lane (no LDS min / max atomics) */
        if (threadIdx.x == 0) { ext[0] = (int) px0; ext[1] = (int) py0; ext[4] = (int) F.count; }
        if (threadIdx.x == n_act_u - 1u) { ext[2] = (int) px0; ext[3] = (int) py0; }
    }
    const bool gathered = act && !shifted;
    if (!WONLY) s_val[threadIdx.x] = gathered ? make_float4(val[0], val[1], val[2], val[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (WONLY == 2) s_sv[threadIdx.x] = gathered ? val[3] : 0.f;
#pragma unroll
    for (int k = 0; k < TAPS; ++k) { s_wx[k][threadIdx.x] = gathered ? F.wx[k] : 0.f; s_wy[k][threadIdx.x] = gathered ? F.wy[k] : 0.f; }
    __syncthreads();
    const int ox = ext[0], oy = ext[1], tw = ext[2] + ext[4] - ext[0], th = ext[4];
    const bool one_row = tw > 0 && ext[1] == ext[3] && th <= TAPS;   /* all lanes in one image row (the footprint size is a constant of the filter) */
    if (one_row) {
        /* the lanes of pixel (x, y) are the global lanes [p * spp, (p + 1) * spp), p = y * samp_w + x in sample-grid coordinates (integrator.cpp:322-334), and their
         * footprint starts at x - n: for the tile column `col`, tap t collects exactly the lanes of pixel x = col - t + n */
        const int count = th, nhalf = (count - 1) / 2, y_pix = oy + nhalf;
        const int64_t g0 = (int64_t) lane_base + (int64_t) blockIdx.x * kBlock;
        const int n_act = (int) n_act_u;
        const int slab_w = HAR_SPLAT_TILE_PIXELS / th;
        for (int c0 = 0; c0 < tw; c0 += slab_w) {
            const int sw = min(slab_w, tw - c0), E = sw * th;
            int G = 1; while (G < HAR_SPLAT_GATHER_MAX_SPLIT && E * G * 2 <= kBlock) G *= 2;
            for (int idx = threadIdx.x; idx < E * G; idx += kBlock) {
                const int e = idx % E, g = idx / E, tx = e % sw, ty = e / sw, col = ox + c0 + tx;
                float ax = 0.f, ay = 0.f, az = 0.f, aw = 0.f;
                /* only the taps whose pixel x = col - t + nhalf is one of the block's pixels [x_first, x_last] have lanes here: at spp >= 256 that is ONE tap per
                 * thread (a wave used to walk all `count` taps with mostly idle lanes: 5x the LDS reads) */
                const int x_first = ox + nhalf, x_last = ext[2] + nhalf;
                const int t_lo = max(0, col + nhalf - x_last), t_hi = min(count - 1, col + nhalf - x_first);
                for (int t = t_lo; t <= t_hi; ++t) {
                    const int x = col - t + nhalf;
                    /* pixel (x, y_pix) of the crop window is pixel (x + border, y_pix + border) of the sample grid (Film::sample_border; border = 0 otherwise) */
                    if (x < -(int) C.border || x >= (int) (C.crop_w + C.border)) continue;
                    const int64_t first = ((int64_t) (y_pix + (int) C.border) * C.samp_w + (x + (int) C.border)) * spp - g0;
                    int la = (int) max((int64_t) 0, first), lb = (int) min((int64_t) n_act, first + spp);
                    if (la >= lb) continue;
                    const int len = lb - la; lb = la + (len * (g + 1)) / G; la = la + (len * g) / G;
                    const float *wxp = s_wx[t], *wyp = s_wy[ty];
#pragma unroll 4
                    for (int l = la; l < lb; ++l) {
                        const float w = wxp[l] * wyp[l];
                        if (WONLY == 2) aw += s_sv[l] * w;
                        else if (WONLY) aw += 1.f * w;
                        else { const float4 v = s_val[l]; ax += v.x * w; ay += v.y * w; az += v.z * w; aw += v.w * w; }
                    }
                }
                reinterpret_cast<float4 *>(tile)[idx] = make_float4(ax, ay, az, aw);      /* partial sum g of pixel e (E * G <= 512 slots) */
            }
            __syncthreads();
            for (int k = threadIdx.x; k < E * 4; k += kBlock) {
                float v = tile[k];
                for (int g = 1; g < G; ++g) v += tile[k + 4 * E * g];
                const int px = k >> 2, x = ox + c0 + px % sw, y = oy + px / sw;
                if (v != 0.f && (uint32_t) x < C.crop_w && (uint32_t) y < C.crop_h)
                    atomicAdd(film + 4 * ((size_t) y * C.crop_w + x) + (k & 3), v);
            }
            __syncthreads();
        }
    }
    if (act && (!one_row || shifted)) {           /* lanes of two image rows in one block (width * spp not a multiple of 256), shifted footprints: per-lane scatter */
#pragma unroll
        for (int ys = 0; ys < TAPS; ++ys)
#pragma unroll
            for (int xs = 0; xs < TAPS; ++xs) {
                const uint32_t x = F.x0 + (uint32_t) xs, y = F.y0 + (uint32_t) ys;
                if ((uint32_t) xs < F.count && (uint32_t) ys < F.count && x < C.crop_w && y < C.crop_h) {
                    const float w = F.wx[xs] * F.wy[ys];
                    float *p = film + 4 * ((size_t) y * C.crop_w + x);
                    if (!weights_only) { atomicAdd(p, val[0] * w); atomicAdd(p + 1, val[1] * w); atomicAdd(p + 2, val[2] * w); }
                    atomicAdd(p + 3, val[3] * w);
                }
            }
    }
}

/* pixel jitter of pass `pass` for paths that draw nothing else (max_depth = 0, path.cpp:102-103): numbers 2*pass, 2*pass+1 of the stream */
__global__ void k_pass_jitter(uint32_t seed, uint32_t lane_base, uint32_t n, uint32_t pass, float2 *jitter) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t rng, inc; sampler_seed(seed, lane_base + i, rng, inc);
    for (uint32_t k = 0; k < 2 * pass; ++k) rng = rng * HAR_PCG32_MULT + inc;
    float jx = pcg32_next_float(rng, inc), jy = pcg32_next_float(rng, inc);
    jitter[i] = make_float2(jx, jy);
}

/* --------------------------------------------------------- small utilities */
/* HDRFilm::develop, JIT branch (hdrfilm.cpp:310-395): colour 0 = RGB, 1 = Y (`luminance(rgb)`, spectrum.h:439-442), 2 = XYZ (`srgb_to_xyz`, spectrum.h:402-410: M * rgb
 * as the sum of the matrix columns scaled by the components).  The conversion runs on the WEIGHTED sums, the division by the weight comes last (:390). */
__global__ void k_develop(const float *film, uint32_t npx, float *image, int colour) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npx) return;
    float4 f = reinterpret_cast<const float4 *>(film)[i];
    float w = f.w == 0.f ? 1.f : f.w;
    if (colour == 1) { image[i] = fma_(f.z, 0.072169f, fma_(f.y, 0.715160f, f.x * 0.212671f)) / w; return; }
    if (colour == 2) {
        f = make_float4(fma_(f.z, 0.180423f, fma_(f.y, 0.357580f, f.x * 0.412453f)), fma_(f.z, 0.072169f, fma_(f.y, 0.715160f, f.x * 0.212671f)),
                        fma_(f.z, 0.950227f, fma_(f.y, 0.119193f, f.x * 0.019334f)), f.w);
    }
    image[3 * (size_t) i] = f.x / w; image[3 * (size_t) i + 1] = f.y / w; image[3 * (size_t) i + 2] = f.z / w;
}
__global__ void k_adjoint_image(const float *grad_in, const float *wfilm, uint32_t npx, float *adj) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npx) return;
    float w = wfilm[4 * (size_t) i + 3]; float iw = w == 0.f ? 1.f : w;
    for (int c = 0; c < 3; ++c) adj[3 * (size_t) i + c] = grad_in[3 * (size_t) i + c] / iw;
}
__global__ void k_add(const float *src, float *dst, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && src[i] != 0.f) atomicAdd(dst + i, src[i]);      /* two streams may add to the same destination */
}
__global__ void k_accumulate_stats(const uint32_t *counters, uint32_t n_bounces, unsigned long long *totals, uint32_t paths) {
    if (threadIdx.x || blockIdx.x) return;
    unsigned long long v = 0, sh = 0;
    for (uint32_t b = 0; b < n_bounces; ++b)
        for (uint32_t k = 0; k < HAR_SHARDS; ++k) {
            v += counters[((size_t) b * HAR_SHARDS + k) * HAR_COUNTER_STRIDE];
            sh += counters[((size_t) (HAR_MAX_BOUNCE_SLOTS + b) * HAR_SHARDS + k) * HAR_COUNTER_STRIDE];
        }
    totals[0] += paths; totals[1] += v; totals[2] += v; totals[3] += sh;
}

/* --- array-valued plugin surface (Scene::ray_intersect*, Sampler, BSDF, Sensor, ImageBlock) --- */
template <bool NAIVE>
__global__ __launch_bounds__(kBlock) void k_api_intersect(DScene S, uint32_t n, const float *o, const float *d, const float *maxt, const uint8_t *active,
                                                          float *t, float *u, float *v, uint32_t *prim, uint32_t *shape, uint32_t *inst, int *status) {
    __shared__ uint2 lds[HAR_LDS_STACK_DEPTH * kBlock];
    LdsStack<HAR_LDS_STACK_DEPTH> stack{ lds + threadIdx.x };
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    Vec3 O(o[i], o[n + i], o[2 * (size_t) n + i]), D(d[i], d[n + i], d[2 * (size_t) n + i]);
    Hit hit; int st = 0;
    /* a masked lane traces nothing and reports "no intersection": t = inf, zero-initialised indices (the `active` argument of scene.cpp:216-230) */
    if (active && !active[i]) { hit.t = HAR_INF; hit.u = 0.f; hit.v = 0.f; hit.prim = 0; hit.shape = 0; hit.inst = 0xffffffffu; }
    else if (NAIVE) accel_trace_naive<false>(S.accel, S.blas_tri_ranges, O, D, maxt[i], hit);
    else {      /* the production traversal code (Traversal<HAR_TRAV_POLICY>::step), one ray per lane */
        Traversal<HAR_TRAV_POLICY> T; T.begin(S.accel, O, D, maxt[i], (S.accel.top_last & 2u) != 0u);
        while (!T.template step<false, LdsStack<HAR_LDS_STACK_DEPTH>, NoProbe, 1>(S.accel, stack, st)) { }
        hit = T.hit;
    }
    if (st) atomicMax(status, st);
    t[i] = hit.t; u[i] = hit.u; v[i] = hit.v; prim[i] = hit.prim; shape[i] = hit.shape; inst[i] = hit.inst;
}
template <bool NAIVE>
__global__ __launch_bounds__(kBlock) void k_api_ray_test(DScene S, uint32_t n, const float *o, const float *d, const float *maxt, const uint8_t *active, uint8_t *out, int *status) {
    __shared__ uint2 lds[HAR_LDS_STACK_DEPTH * kBlock];
    LdsStack<HAR_LDS_STACK_DEPTH> stack{ lds + threadIdx.x };
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    Vec3 O(o[i], o[n + i], o[2 * (size_t) n + i]), D(d[i], d[n + i], d[2 * (size_t) n + i]);
    Hit hit; int st = 0; bool r;
    if (active && !active[i]) r = false;                  /* masked lane: `false` (scene.cpp:232-238) */
    else if (NAIVE) r = accel_trace_naive<true>(S.accel, S.blas_tri_ranges, O, D, maxt[i], hit);
    else {
        Traversal<HAR_TRAV_POLICY> T; T.begin(S.accel, O, D, maxt[i], (S.accel.top_last & 1u) != 0u);
        while (!T.template step<true, LdsStack<HAR_LDS_STACK_DEPTH>, NoProbe, 1>(S.accel, stack, st)) { }
        r = T.found;
    }
    if (st) atomicMax(status, st);
    out[i] = r ? 1 : 0;
}
/* out = 33 rows of n floats: p 0-2, n 3-5, sh_frame.n 6-8, sh_frame.s 9-11, sh_frame.t 12-14, wi 15-17, uv 18-19, t 20, dp_du 21-23, dp_dv 24-26, dn_du 27-29,
 * dn_dv 30-32 (SurfaceInteraction3f, interaction.h:345-420).  A lane that is masked or holds no intersection gets what the reference's masked vcall + finalize leave
 * (interaction.h:559-605,804-829): t = inf, zero-initialised fields, the frame coordinate_system builds around a zero normal, wi = -ray.d. */
__global__ void k_api_si(DScene S, uint32_t n, const float *o, const float *d, const float *t, const float *u, const float *v,
                         const uint32_t *prim, const uint32_t *shape, const uint32_t *inst, uint32_t ray_flags, const uint8_t *active, float *out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    (void) o;
    const Vec3 D(d[i], d[n + i], d[2 * (size_t) n + i]);
    const bool shading = (ray_flags & RAY_SHADING) != 0u, valid = !(active && !active[i]) && t[i] != HAR_INF;
    Vec3 vs[6] = { Vec3(0.f), Vec3(0.f), Vec3(0.f), Vec3(0.f), Vec3(0.f), Vec3(0.f) };
    float uvx = 0.f, uvy = 0.f;
    SurfPartials P; P.dp_du = Vec3(0.f); P.dp_dv = Vec3(0.f); P.dn_du = Vec3(0.f); P.dn_dv = Vec3(0.f);
    if (valid) {
        const SurfInt si = compute_si(S, D, t[i], u[i], v[i], prim[i], shape[i], inst[i]);
        vs[0] = si.p; vs[1] = si.n;
        if (shading) {
            vs[2] = si.sn; vs[3] = si.ss; vs[4] = si.st; vs[5] = si.wi; uvx = si.uv_x; uvy = si.uv_y;
            compute_si_partials(S, u[i], v[i], prim[i], shape[i], inst[i], (ray_flags & RAY_NORMAL_PARTIALS) != 0u, P);
        }
    } else if (shading) { coordinate_system(Vec3(0.f), vs[3], vs[4]); vs[5] = -D; }
    for (int k = 0; k < 6; ++k) { out[(3 * k) * (size_t) n + i] = vs[k].x; out[(3 * k + 1) * (size_t) n + i] = vs[k].y; out[(3 * k + 2) * (size_t) n + i] = vs[k].z; }
    out[18 * (size_t) n + i] = uvx; out[19 * (size_t) n + i] = uvy; out[20 * (size_t) n + i] = valid ? t[i] : HAR_INF;
    const Vec3 ps[4] = { P.dp_du, P.dp_dv, P.dn_du, P.dn_dv };
    for (int k = 0; k < 4; ++k) { out[(21 + 3 * k) * (size_t) n + i] = ps[k].x; out[(22 + 3 * k) * (size_t) n + i] = ps[k].y; out[(23 + 3 * k) * (size_t) n + i] = ps[k].z; }
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
/* BSDF::eval_pdf / eval / pdf (bsdf.h:375-465) of scene BSDF `bsdf` under the caller's BSDFContext; `value` / `pdf` may be NULL (eval / pdf alone: for the
 * models of har_bsdf.h the pair is computed in one pass and BSDF::eval, ::pdf return its halves).  A masked lane evaluates to zero. */
__global__ void k_api_bsdf_eval_pdf(DScene S, uint32_t bsdf, BsdfCtx ctx, uint32_t n, const float *wi, const float *uv, const float *wo, const uint8_t *active, float *value, float *pdf) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    BsdfEval e; e.value = Vec3(0.f); e.pdf = 0.f;
    if (!(active && !active[i])) {
        BsdfSide side; const bool ok = bsdf_side(S, bsdf, Vec3(wi[i], wi[n + i], wi[2 * (size_t) n + i]), side);
        TexTaps taps; const BsdfInputs in = bsdf_inputs(S, S.bsdfs[side.index], uv[i], uv[n + i], taps);
        bsdf_eval_pdf<HAR_BSDF_ALL_TYPES, true>(S, side, in, ok, Vec3(wo[i], wo[n + i], wo[2 * (size_t) n + i]), e, bsdf_side_ctx(S, bsdf, side, ctx));
    }
    if (value) { value[i] = e.value.x; value[n + i] = e.value.y; value[2 * (size_t) n + i] = e.value.z; }
    if (pdf) pdf[i] = e.pdf;
}
/* BSDF::sample (bsdf.h:322-373): BSDFSample3f {wo, pdf, eta, sampled_type, sampled_component} + the weight.  A masked lane returns dr::zeros. */
__global__ void k_api_bsdf_sample(DScene S, uint32_t bsdf, BsdfCtx ctx, uint32_t n, const float *wi, const float *uv, const float *s1, const float *s2, const uint8_t *active,
                                  float *wo, float *pdf, float *weight, float *eta, uint32_t *stype, uint32_t *scomp) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    BsdfSample b; b.wo = Vec3(0.f); b.pdf = 0.f; b.weight = Vec3(0.f); b.eta = 0.f; b.delta = false; b.type = 0u; b.comp = 0u;
    if (!(active && !active[i])) {
        BsdfSide side; const bool ok = bsdf_side(S, bsdf, Vec3(wi[i], wi[n + i], wi[2 * (size_t) n + i]), side);
        TexTaps taps; const BsdfInputs in = bsdf_inputs(S, S.bsdfs[side.index], uv[i], uv[n + i], taps);
        bsdf_sample<HAR_BSDF_ALL_TYPES, true>(S, side, in, ok, s1 ? s1[i] : 0.f, s2[i], s2[n + i], b, bsdf_side_ctx(S, bsdf, side, ctx));
    }
    wo[i] = b.wo.x; wo[n + i] = b.wo.y; wo[2 * (size_t) n + i] = b.wo.z; pdf[i] = b.pdf;
    weight[i] = b.weight.x; weight[n + i] = b.weight.y; weight[2 * (size_t) n + i] = b.weight.z;
    if (eta) eta[i] = b.eta;
    if (stype) stype[i] = b.type;
    if (scomp) scomp[i] = b.comp;
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
                   uint32_t shard_cap, const WaveState &out, float4 *result, uint32_t *count, const float *adj, float4 *dL, const PassState &ps, bool lite) {
    dim3 g(blocks_for(n)), b(kBlock);
    if (lite && mode == MODE_PRB_PRIMAL && adj) hipLaunchKernelGGL((k_raygen<MODE_PRB_PRIMAL, true>), g, b, 0, s, C, seed, spp, log_spp, lane_base, n, shard_cap, out, result, count, adj, dL, ps);
    else if (lite && mode == MODE_PATH) hipLaunchKernelGGL((k_raygen<MODE_PATH, true>), g, b, 0, s, C, seed, spp, log_spp, lane_base, n, shard_cap, out, result, count, adj, dL, ps);
    else if (mode == MODE_PRB_ADJOINT) hipLaunchKernelGGL(k_raygen<MODE_PRB_ADJOINT>, g, b, 0, s, C, seed, spp, log_spp, lane_base, n, shard_cap, out, result, count, adj, dL, ps);
    else if (mode == MODE_PRB_PRIMAL && adj) hipLaunchKernelGGL(k_raygen<MODE_PRB_PRIMAL>, g, b, 0, s, C, seed, spp, log_spp, lane_base, n, shard_cap, out, result, count, adj, dL, ps);
    else hipLaunchKernelGGL(k_raygen<MODE_PATH>, g, b, 0, s, C, seed, spp, log_spp, lane_base, n, shard_cap, out, result, count, adj, dL, ps);
}
void launch_raygen_rays(hipStream_t s, uint32_t seed, uint32_t lane_base, uint32_t n, uint32_t n_total, uint32_t first, const float *o, const float *d, const float *maxt,
                        const uint64_t *state, const uint8_t *active, uint32_t shard_cap, const WaveState &out, float4 *result, uint32_t *count) {
    hipLaunchKernelGGL(k_raygen_rays, dim3(blocks_for(n)), dim3(kBlock), 0, s, seed, lane_base, n, n_total, first, o, d, maxt, state, active, shard_cap, out, result, count);
}
void launch_sample_out(hipStream_t s, uint32_t n, uint32_t n_total, uint32_t first, const float4 *result, const float *valid_lane, int zero_invalid, float *rgb, uint8_t *valid,
                       const uint8_t *active, uint32_t seed, uint32_t lane_base, const uint64_t *state_in, uint64_t *state_out) {
    hipLaunchKernelGGL(k_sample_out, dim3(blocks_for(n)), dim3(kBlock), 0, s, n, n_total, first, result, valid_lane, zero_invalid, rgb, valid, active, seed, lane_base, state_in, state_out);
}
/* HAR_FLAT_KERNELS=0: scenes without a TLAS run the generic traversal kernels (A/B switch) */
static bool flat_kernels() { static const bool on = !(getenv("HAR_FLAT_KERNELS") && atoi(getenv("HAR_FLAT_KERNELS")) == 0); return on; }
void launch_trace_closest(hipStream_t s, uint32_t grid, uint2 *spill, const Accel &A, const uint32_t *count, uint32_t *cursor, uint32_t shard_cap,
                          const WaveState &in, float4 *h0, uint2 *h1, int *status, const PacketList *pl) {
    const uint32_t *list = pl ? pl->list : nullptr, *shard_count = pl ? pl->shard_count : nullptr; const uint32_t stride = pl ? pl->stride : 0u;
    /* scenes without a TLAS run the FLAT instantiation of the traversal (har_accel.h): no instance blocks, no world-space ray copy */
#define HAR_LAUNCH_TC(SP, FL, LI) hipLaunchKernelGGL((k_trace_closest<SP, FL, LI>), dim3(grid), dim3(kBlock), 0, s, A, count, cursor, shard_cap, in.a0, in.a1, h0, h1, status, spill, list, stride, shard_count)
    const bool flat = !A.has_tlas && flat_kernels();
    if (pl) { if (spill) HAR_LAUNCH_TC(true, false, true); else if (flat) HAR_LAUNCH_TC(false, true, true); else HAR_LAUNCH_TC(false, false, true); }
    else    { if (spill) HAR_LAUNCH_TC(true, false, false); else if (flat) HAR_LAUNCH_TC(false, true, false); else HAR_LAUNCH_TC(false, false, false); }
#undef HAR_LAUNCH_TC
}
/* the wave-shared descent of a coherent closest-hit launch (k_trace_packet); the packets it gives up on are filed in `pl` for launch_trace_closest(.., &pl) */
void launch_trace_packet(hipStream_t s, uint32_t grid, const Accel &A, const uint32_t *count, uint32_t *cursor, uint32_t shard_cap, const WaveState &in, float4 *h0, uint2 *h1,
                         const PacketList &pl, uint32_t budget) {
    if (!A.has_tlas) hipLaunchKernelGGL(k_trace_packet<true>, dim3(grid), dim3(kBlock), 0, s, A, count, cursor, shard_cap, in.a0, in.a1, h0, h1, pl.list, pl.stride, pl.count, budget);
    else hipLaunchKernelGGL(k_trace_packet<false>, dim3(grid), dim3(kBlock), 0, s, A, count, cursor, shard_cap, in.a0, in.a1, h0, h1, pl.list, pl.stride, pl.count, budget);
}
void launch_shade(int mode, hipStream_t s, uint32_t grid, const DScene &S, const ShadeParams &P, uint32_t lane_base, uint32_t shard_cap, const uint32_t *count_in,
                  const WaveState &in, const float4 *h0, const uint2 *h1, const WaveState &out, uint32_t *count_out, const ItemArrays &items,
                  uint32_t *item_count, float4 *result, const ReplayCache &rc, uint64_t *pass_rng, const float4 *dL, float *grad_slots, const ShapeArrays *geo,
                  float *const *grad_tex, const TexelQueues *tq_in, float *grad_extra, const MaterialQueues *mq_in, uint32_t mat_class, const TapeArrays *tape_in) {
    dim3 g(grid), b(kBlock);
    /* small scenes: mesh / BSDF / instance tables in LDS (k_shade<.., TAB>); HAR_SHADE_TABLES=0 switches it off (A/B) */
    static const bool tab_env = !(getenv("HAR_SHADE_TABLES") && atoi(getenv("HAR_SHADE_TABLES")) == 0);
    const bool tab = tab_env && S.n_meshes <= HAR_TAB_MESHES && S.n_bsdfs <= HAR_TAB_BSDFS && S.n_insts <= HAR_TAB_INSTS;
    const bool first = (P.flags & HAR_SHADE_FIRST_VERTEX) != 0u;       /* bounce 0 after k_raygen<.., LITE>: only the plain forward flavours and the record flavour are instantiated for it (run_chunk decides) */
    const ShapeArrays no_geo{ nullptr, nullptr, nullptr, nullptr, nullptr };
    const TexelQueues no_tq{ nullptr, nullptr, nullptr, nullptr, 0u, 0u };
    const TexelQueues tq = tq_in ? *tq_in : no_tq;
    const MaterialQueues no_mq{ nullptr, nullptr, 0u, 0u };
    const TapeArrays tape = tape_in ? *tape_in : TapeArrays{ nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
    if (rc.mode == 5) {       /* record tape, primal pass: the adjoint flavour of the shading code, primal bookkeeping, one record per vertex (k_shade<.., RECORD>) */
        const bool env = (S.bsdf_types & HAR_SCENE_ENVMAP) != 0u, diffuse = S.bsdf_types == HAR_BSDF_ONLY_DIFFUSE, cls = (S.bsdf_types & 0x7fffffffu & ~HAR_BSDF_CLASSIC_TYPES) == 0u;
#define HAR_LAUNCH_SHADE_RECORD(T) do { if (first && tab) hipLaunchKernelGGL((k_shade<MODE_PRB_ADJOINT, T, false, false, false, false, true, true, true>), g, b, 0, s, S, P, lane_base, shard_cap, count_in, in, h0, h1, out, count_out, items, item_count, result, rc, pass_rng, dL, grad_slots, no_geo, nullptr, no_tq, nullptr, no_mq, 0u, tape); \
        else if (first) hipLaunchKernelGGL((k_shade<MODE_PRB_ADJOINT, T, false, false, false, false, true, false, true>), g, b, 0, s, S, P, lane_base, shard_cap, count_in, in, h0, h1, out, count_out, items, item_count, result, rc, pass_rng, dL, grad_slots, no_geo, nullptr, no_tq, nullptr, no_mq, 0u, tape); \
        else if (tab) hipLaunchKernelGGL((k_shade<MODE_PRB_ADJOINT, T, false, false, false, false, true, true>), g, b, 0, s, S, P, lane_base, shard_cap, count_in, in, h0, h1, out, count_out, items, item_count, result, rc, pass_rng, dL, grad_slots, no_geo, nullptr, no_tq, nullptr, no_mq, 0u, tape); \
        else hipLaunchKernelGGL((k_shade<MODE_PRB_ADJOINT, T, false, false, false, false, true>), g, b, 0, s, S, P, lane_base, shard_cap, count_in, in, h0, h1, out, count_out, items, item_count, result, rc, pass_rng, dL, grad_slots, no_geo, nullptr, no_tq, nullptr, no_mq, 0u, tape); } while (0)
        if (env && (S.bsdf_types & HAR_SCENE_TEXLIGHT)) HAR_LAUNCH_SHADE_RECORD(HAR_BSDF_ALL_TYPES | HAR_SCENE_ENVMAP | HAR_SCENE_TEXLIGHT);
        else if (env) HAR_LAUNCH_SHADE_RECORD(HAR_BSDF_ALL_TYPES | HAR_SCENE_ENVMAP); else if (diffuse) HAR_LAUNCH_SHADE_RECORD(HAR_BSDF_ONLY_DIFFUSE);
        else if (cls) HAR_LAUNCH_SHADE_RECORD(HAR_BSDF_CLASSIC_TYPES); else HAR_LAUNCH_SHADE_RECORD(HAR_BSDF_ALL_TYPES);
#undef HAR_LAUNCH_SHADE_RECORD
        return;
    }
    if (mq_in && mode != MODE_PRB_ADJOINT) {
        /* one material class: the kernel of that BSDF model (TYPES = its bit | HAR_BSDF_QUEUED), the generic classic / all-model kernel for twosided pairs of two models */
        const MaterialQueues mq = *mq_in;
        const bool classic = (S.bsdf_types & 0x7fffffffu & ~HAR_BSDF_CLASSIC_TYPES) == 0u;
#define HAR_LAUNCH_SHADE_Q(M, T) hipLaunchKernelGGL((k_shade<M, T, false, false, false, true>), g, b, 0, s, S, P, lane_base, shard_cap, count_in, in, h0, h1, out, count_out, items, item_count, result, rc, pass_rng, dL, grad_slots, no_geo, nullptr, no_tq, nullptr, mq, mat_class, tape)
#define HAR_LAUNCH_SHADE_Q_MODE(M) do { switch (mat_class) { \
            case BSDF_DIFFUSE:        HAR_LAUNCH_SHADE_Q(M, (1u << BSDF_DIFFUSE) | HAR_BSDF_QUEUED); break; \
            case BSDF_DIELECTRIC:     HAR_LAUNCH_SHADE_Q(M, (1u << BSDF_DIELECTRIC) | HAR_BSDF_QUEUED); break; \
            case BSDF_ROUGHCONDUCTOR: HAR_LAUNCH_SHADE_Q(M, (1u << BSDF_ROUGHCONDUCTOR) | HAR_BSDF_QUEUED); break; \
            case BSDF_ROUGHPLASTIC:   HAR_LAUNCH_SHADE_Q(M, (1u << BSDF_ROUGHPLASTIC) | HAR_BSDF_QUEUED); break; \
            case BSDF_CONDUCTOR:      HAR_LAUNCH_SHADE_Q(M, (1u << BSDF_CONDUCTOR) | HAR_BSDF_QUEUED); break; \
            case BSDF_PLASTIC:        HAR_LAUNCH_SHADE_Q(M, (1u << BSDF_PLASTIC) | HAR_BSDF_QUEUED); break; \
            default: if (classic) HAR_LAUNCH_SHADE_Q(M, HAR_BSDF_CLASSIC_TYPES | HAR_BSDF_QUEUED); else HAR_LAUNCH_SHADE_Q(M, HAR_BSDF_ALL_TYPES | HAR_BSDF_QUEUED); break; } } while (0)
        if (mode == MODE_PATH) HAR_LAUNCH_SHADE_Q_MODE(MODE_PATH); else HAR_LAUNCH_SHADE_Q_MODE(MODE_PRB_PRIMAL);
#undef HAR_LAUNCH_SHADE_Q_MODE
#undef HAR_LAUNCH_SHADE_Q
        return;
    }
    if (geo && mode == MODE_PRB_ADJOINT) {        /* vertex-position / instance gradients (har_integrator_set_grad_positions checks what can be differentiated) */
        /* generic-emitter scenes (mesh / textured / delta lights, environment maps, sampling weights): the emitter sample's kind travels in the geometry record
         * (HAR_SHAPE_NEE_SURFACE / _POINT / _SPOT, har_shape_grad.h) */
        if ((S.bsdf_types & HAR_SCENE_ENVMAP) && (S.bsdf_types & HAR_SCENE_TEXLIGHT))
            hipLaunchKernelGGL((k_shade<MODE_PRB_ADJOINT, HAR_BSDF_ALL_TYPES | HAR_SCENE_ENVMAP | HAR_SCENE_TEXLIGHT, true>), g, b, 0, s, S, P, lane_base, shard_cap, count_in, in, h0, h1, out, count_out, items, item_count,
                               result, rc, pass_rng, dL, grad_slots, *geo, grad_tex, no_tq, nullptr, no_mq, 0u, tape);
        else if (S.bsdf_types & HAR_SCENE_ENVMAP)
            hipLaunchKernelGGL((k_shade<MODE_PRB_ADJOINT, HAR_BSDF_ALL_TYPES | HAR_SCENE_ENVMAP, true>), g, b, 0, s, S, P, lane_base, shard_cap, count_in, in, h0, h1, out, count_out, items, item_count,
                               result, rc, pass_rng, dL, grad_slots, *geo, grad_tex, no_tq, nullptr, no_mq, 0u, tape);
        else if (S.bsdf_types == HAR_BSDF_ONLY_DIFFUSE)
            hipLaunchKernelGGL((k_shade<MODE_PRB_ADJOINT, HAR_BSDF_ONLY_DIFFUSE, true>), g, b, 0, s, S, P, lane_base, shard_cap, count_in, in, h0, h1, out, count_out, items, item_count,
                               result, rc, pass_rng, dL, grad_slots, *geo, grad_tex, no_tq, nullptr, no_mq, 0u, tape);
        else if ((S.bsdf_types & 0x7fffffffu & ~HAR_BSDF_CLASSIC_TYPES) != 0u)      /* `conductor` / `plastic` records somewhere in the scene */
            hipLaunchKernelGGL((k_shade<MODE_PRB_ADJOINT, HAR_BSDF_ALL_TYPES, true>), g, b, 0, s, S, P, lane_base, shard_cap, count_in, in, h0, h1, out, count_out, items, item_count,
                               result, rc, pass_rng, dL, grad_slots, *geo, grad_tex, no_tq, nullptr, no_mq, 0u, tape);
        else
            hipLaunchKernelGGL((k_shade<MODE_PRB_ADJOINT, HAR_BSDF_CLASSIC_TYPES, true>), g, b, 0, s, S, P, lane_base, shard_cap, count_in, in, h0, h1, out, count_out, items, item_count,
                               result, rc, pass_rng, dL, grad_slots, *geo, grad_tex, no_tq, nullptr, no_mq, 0u, tape);
        return;
    }
    if (grad_tex && mode == MODE_PRB_ADJOINT && (rc.mode == 2 || rc.mode == 4)) {       /* cached / taped bounce of the adjoint replay: commit in place (see k_shade) */
        const bool env = (S.bsdf_types & HAR_SCENE_ENVMAP) != 0u, diffuse = S.bsdf_types == HAR_BSDF_ONLY_DIFFUSE, cls = (S.bsdf_types & 0x7fffffffu & ~HAR_BSDF_CLASSIC_TYPES) == 0u;
#define HAR_LAUNCH_SHADE_INLINE(T) hipLaunchKernelGGL((k_shade<MODE_PRB_ADJOINT, T, false, true>), g, b, 0, s, S, P, lane_base, shard_cap, count_in, in, h0, h1, out, count_out, items, item_count, result, rc, pass_rng, dL, grad_slots, no_geo, grad_tex, tq, nullptr, no_mq, 0u, tape)
#define HAR_LAUNCH_SHADE_EXTRA(T) hipLaunchKernelGGL((k_shade<MODE_PRB_ADJOINT, T, false, true, true>), g, b, 0, s, S, P, lane_base, shard_cap, count_in, in, h0, h1, out, count_out, items, item_count, result, rc, pass_rng, dL, grad_slots, no_geo, grad_tex, tq, grad_extra, no_mq, 0u, tape)
        if (grad_extra && !diffuse) {         /* gradients w.r.t. alpha / eta / k / slot 1: the generic shading code with the extra derivative terms */
            if (env && (S.bsdf_types & HAR_SCENE_TEXLIGHT)) HAR_LAUNCH_SHADE_EXTRA(HAR_BSDF_ALL_TYPES | HAR_SCENE_ENVMAP | HAR_SCENE_TEXLIGHT);
            else if (env) HAR_LAUNCH_SHADE_EXTRA(HAR_BSDF_ALL_TYPES | HAR_SCENE_ENVMAP); else HAR_LAUNCH_SHADE_EXTRA(HAR_BSDF_ALL_TYPES);
            return;
        }
        if (env && (S.bsdf_types & HAR_SCENE_TEXLIGHT)) HAR_LAUNCH_SHADE_INLINE(HAR_BSDF_ALL_TYPES | HAR_SCENE_ENVMAP | HAR_SCENE_TEXLIGHT);
        else if (env) HAR_LAUNCH_SHADE_INLINE(HAR_BSDF_ALL_TYPES | HAR_SCENE_ENVMAP); else if (diffuse) HAR_LAUNCH_SHADE_INLINE(HAR_BSDF_ONLY_DIFFUSE);
        else if (cls) HAR_LAUNCH_SHADE_INLINE(HAR_BSDF_CLASSIC_TYPES); else HAR_LAUNCH_SHADE_INLINE(HAR_BSDF_ALL_TYPES);
#undef HAR_LAUNCH_SHADE_INLINE
#undef HAR_LAUNCH_SHADE_EXTRA
        return;
    }
    /* diffuse-only scenes (no twosided wrappers) run kernels in which the other BSDF models are compiled out */
    const bool only_diffuse = S.bsdf_types == HAR_BSDF_ONLY_DIFFUSE;
#define HAR_LAUNCH_SHADE(M, T) do { if (first && M == MODE_PATH) { if (tab) hipLaunchKernelGGL((k_shade<MODE_PATH, T, false, false, false, false, false, true, true>), g, b, 0, s, S, P, lane_base, shard_cap, count_in, in, h0, h1, out, count_out, items, item_count, result, rc, pass_rng, dL, grad_slots, no_geo, nullptr, no_tq, nullptr, no_mq, 0u, tape); \
            else hipLaunchKernelGGL((k_shade<MODE_PATH, T, false, false, false, false, false, false, true>), g, b, 0, s, S, P, lane_base, shard_cap, count_in, in, h0, h1, out, count_out, items, item_count, result, rc, pass_rng, dL, grad_slots, no_geo, nullptr, no_tq, nullptr, no_mq, 0u, tape); } \
        else if (tab && M != MODE_PRB_ADJOINT) hipLaunchKernelGGL((k_shade<M, T, false, false, false, false, false, true>), g, b, 0, s, S, P, lane_base, shard_cap, count_in, in, h0, h1, out, count_out, items, item_count, result, rc, pass_rng, dL, grad_slots, no_geo, nullptr, no_tq, nullptr, no_mq, 0u, tape); \
        else hipLaunchKernelGGL((k_shade<M, T>), g, b, 0, s, S, P, lane_base, shard_cap, count_in, in, h0, h1, out, count_out, items, item_count, result, rc, pass_rng, dL, grad_slots, no_geo, nullptr, no_tq, nullptr, no_mq, 0u, tape); } while (0)
    const bool envmap = (S.bsdf_types & HAR_SCENE_ENVMAP) != 0u;      /* generic BSDF code + environment-map sampling / lookup */
    const bool classic = (S.bsdf_types & 0x7fffffffu & ~HAR_BSDF_CLASSIC_TYPES) == 0u;
#define HAR_LAUNCH_SHADE_MODE(M) do { if (envmap && (S.bsdf_types & HAR_SCENE_TEXLIGHT)) HAR_LAUNCH_SHADE(M, HAR_BSDF_ALL_TYPES | HAR_SCENE_ENVMAP | HAR_SCENE_TEXLIGHT); else if (envmap) HAR_LAUNCH_SHADE(M, HAR_BSDF_ALL_TYPES | HAR_SCENE_ENVMAP); else if (only_diffuse) HAR_LAUNCH_SHADE(M, HAR_BSDF_ONLY_DIFFUSE); else if (classic) HAR_LAUNCH_SHADE(M, HAR_BSDF_CLASSIC_TYPES); else HAR_LAUNCH_SHADE(M, HAR_BSDF_ALL_TYPES); } while (0)
    if (mode == MODE_PATH)            HAR_LAUNCH_SHADE_MODE(MODE_PATH);
    else if (mode == MODE_PRB_PRIMAL) HAR_LAUNCH_SHADE_MODE(MODE_PRB_PRIMAL);
    else                              HAR_LAUNCH_SHADE_MODE(MODE_PRB_ADJOINT);
#undef HAR_LAUNCH_SHADE_MODE
#undef HAR_LAUNCH_SHADE
}
void launch_tape_begin(hipStream_t s, const DSensor &C, uint32_t seed, uint32_t spp, uint32_t log_spp, uint32_t lane_base, uint32_t n, uint32_t shard_cap,
                       const float4 *result, const float *adj, float4 *la, float2 *lb, float4 *dL_out, const float4 *dL_in) {
    hipLaunchKernelGGL(k_tape_begin, dim3(blocks_for(n)), dim3(kBlock), 0, s, C, seed, spp, log_spp, lane_base, n, shard_cap, result, adj, la, lb, dL_out, dL_in);
}
void launch_commit(hipStream_t s, uint32_t grid, const DScene &S, uint32_t shard_cap, const uint32_t *count_in, const TapeArrays &tape, const uint8_t *vis,
                   float *grad_slots, float *const *grad_tex, const TexelQueues *tq, const float4 *result, const float4 *dL) {
    const TexelQueues no_tq{ nullptr, nullptr, nullptr, nullptr, 0u, 0u };
    /* the record layout is the primal pass's choice (launch_shade, rc.mode == 5): diffuse-only scenes write the compact record */
    const bool compact = HAR_TAPE_COMPACT && S.bsdf_types == HAR_BSDF_ONLY_DIFFUSE;
    if (compact) {
        if (result) hipLaunchKernelGGL((k_commit<true, true>), dim3(grid), dim3(kBlock), 0, s, S, shard_cap, count_in, tape, vis, grad_slots, grad_tex, tq ? *tq : no_tq, result, dL);
        else hipLaunchKernelGGL((k_commit<false, true>), dim3(grid), dim3(kBlock), 0, s, S, shard_cap, count_in, tape, vis, grad_slots, grad_tex, tq ? *tq : no_tq, result, dL);
        return;
    }
    if (result) hipLaunchKernelGGL((k_commit<true, false>), dim3(grid), dim3(kBlock), 0, s, S, shard_cap, count_in, tape, vis, grad_slots, grad_tex, tq ? *tq : no_tq, result, dL);
    else hipLaunchKernelGGL((k_commit<false, false>), dim3(grid), dim3(kBlock), 0, s, S, shard_cap, count_in, tape, vis, grad_slots, grad_tex, tq ? *tq : no_tq, result, dL);
}
void launch_classify(hipStream_t s, uint32_t grid, const DScene &S, uint32_t shard_cap, const uint32_t *count_in, const float4 *h0, const uint2 *h1, const MaterialQueues &mq) {
    hipLaunchKernelGGL(k_classify, dim3(grid), dim3(kBlock), 0, s, S, shard_cap, count_in, h0, h1, mq);
}
void launch_texel_accumulate(hipStream_t s, const TexelQueues &tq, float *const *grad_tex, uint32_t bpq, uint32_t lds_bytes) {
    static const uint32_t spread = getenv("HAR_TQ_SPREAD") ? (uint32_t) atoi(getenv("HAR_TQ_SPREAD")) : 0u;      /* 1: groups of four lanes walk separate segments (A/B, see the kernel) */
    static const bool fixed = !(getenv("HAR_TQ_FIXED") && atoi(getenv("HAR_TQ_FIXED")) == 0);       /* 0: float LDS atomics always (A/B) */
    if (fixed) hipLaunchKernelGGL(k_texel_accumulate<true>, dim3(HAR_SHARDS * tq.nq * bpq), dim3(kBlock), lds_bytes, s, tq, grad_tex, bpq, spread, 0u);
    /* launches whose records hold a non-finite gradient (gmax = +inf): the float version, so that NaN / inf reach the texture; an empty launch otherwise */
    hipLaunchKernelGGL(k_texel_accumulate<false>, dim3(HAR_SHARDS * tq.nq * bpq), dim3(kBlock), lds_bytes, s, tq, grad_tex, bpq, spread, fixed ? 0u : 1u);
}
void launch_resolve(int mode, hipStream_t s, uint32_t grid, uint2 *spill, const DScene &S, const uint32_t *item_count, uint32_t *cursor, uint32_t shard_cap, const ItemArrays &items,
                    float4 *result, const float4 *dL, float *grad_refl, float *const *grad_tex, int *status, const ReplayCache &rc, uint8_t *item_vis, int fwd, const TexelQueues *tq) {
    dim3 g(grid), b(kBlock);
    if (mode == MODE_PRB_ADJOINT && rc.mode == 2) {
        const TexelQueues no_tq{ nullptr, nullptr, nullptr, nullptr, 0u, 0u, nullptr };
        if (fwd) hipLaunchKernelGGL(k_resolve_adjoint_cached<true>, g, b, 0, s, S, item_count, shard_cap, items, result, dL, grad_refl, grad_tex, rc, item_vis, no_tq);
        else hipLaunchKernelGGL(k_resolve_adjoint_cached<false>, g, b, 0, s, S, item_count, shard_cap, items, result, dL, grad_refl, grad_tex, rc, item_vis, tq ? *tq : no_tq);
        return;
    }
#define HAR_LAUNCH_RESOLVE(M, SP, FL) hipLaunchKernelGGL((k_resolve<M, SP, false, FL>), g, b, 0, s, S, item_count, cursor, shard_cap, items, result, dL, grad_refl, grad_tex, status, rc, spill, item_vis)
#define HAR_LAUNCH_RESOLVE_FWD(SP) hipLaunchKernelGGL((k_resolve<MODE_PRB_ADJOINT, SP, true>), g, b, 0, s, S, item_count, cursor, shard_cap, items, result, dL, grad_refl, grad_tex, status, rc, spill, item_vis)
    const bool flat = !spill && !S.accel.has_tlas && flat_kernels();
    if (mode == MODE_PRB_ADJOINT && fwd) { if (spill) HAR_LAUNCH_RESOLVE_FWD(true); else HAR_LAUNCH_RESOLVE_FWD(false); }
    else if (mode == MODE_PRB_ADJOINT) { if (spill) HAR_LAUNCH_RESOLVE(MODE_PRB_ADJOINT, true, false); else if (flat) HAR_LAUNCH_RESOLVE(MODE_PRB_ADJOINT, false, true); else HAR_LAUNCH_RESOLVE(MODE_PRB_ADJOINT, false, false); }
    else { if (spill) HAR_LAUNCH_RESOLVE(MODE_PATH, true, false); else if (flat) HAR_LAUNCH_RESOLVE(MODE_PATH, false, true); else HAR_LAUNCH_RESOLVE(MODE_PATH, false, false); }
#undef HAR_LAUNCH_RESOLVE
#undef HAR_LAUNCH_RESOLVE_FWD
}
void launch_alpha_flags(hipStream_t s, uint32_t grid, uint32_t shard_cap, const uint32_t *count_in, const WaveState &in, const float4 *h0, uint32_t lane_base, float miss_value, float *alpha) {
    hipLaunchKernelGGL(k_alpha_flags, dim3(grid), dim3(kBlock), 0, s, shard_cap, count_in, in.a3, h0, lane_base, miss_value, alpha);
}
void launch_skip_emitters(hipStream_t s, uint32_t grid, const DScene &S, int first, uint32_t shard_cap, const uint32_t *count_in, const float4 *ray_o, const float4 *ray_d,
                          const float4 *hit0, const uint2 *hit1, float4 *h0, uint2 *h1, float4 *dst_o, float4 *dst_d, uint32_t *dst_count) {
    (void) ray_o;
    hipLaunchKernelGGL(k_skip_emitters, dim3(grid), dim3(kBlock), 0, s, S, first, shard_cap, count_in, ray_d, hit0, hit1, h0, h1, dst_o, dst_d, dst_count);
}
void launch_shape_adjoint(hipStream_t s, uint32_t grid, const DScene &S, const uint32_t *item_count, uint32_t shard_cap, const ItemArrays &items, const ShapeArrays &geo,
                          const float4 *result, const float4 *dL, int has_next, const WaveState &next, const float4 *h0, const uint2 *h1, const ReplayCache &rc_next,
                          const ShapeTargets &T) {
    hipLaunchKernelGGL(k_shape_adjoint, dim3(grid), dim3(kBlock), 0, s, S, item_count, shard_cap, items, geo, result, dL, has_next, next, h0, h1, rc_next, T);
}
void launch_normals_adjoint(hipStream_t s, const DScene &S, uint32_t mesh, uint32_t face_count, uint32_t vertex_count, float *acc, const float *nbar, float *grad) {
    if (face_count == 0) return;
    (void) hipMemsetAsync(acc, 0, (size_t) 3 * vertex_count * sizeof(float), s);
    hipLaunchKernelGGL(k_normals_sums, dim3(blocks_for(face_count)), dim3(kBlock), 0, s, S, mesh, acc);
    hipLaunchKernelGGL(k_normals_adjoint, dim3(blocks_for(face_count)), dim3(kBlock), 0, s, S, mesh, acc, nbar, grad);
}
void launch_splat(hipStream_t s, const DSensor &C, uint32_t seed, uint32_t spp, uint32_t log_spp, uint32_t lane_base, uint32_t n,
                  const float4 *result, int weights_only, float *film, const float2 *jitter, const float *scalar) {
    /* filter taps per axis: 2 * ceil(radius - 1/2) + 1 (5 for the default gaussian); the kernel is instantiated for <= 5 and <= 9 (LDS budget) */
    const uint32_t taps = C.rfilter == 0 ? 1u : 2u * (uint32_t) ceilf(C.radius - .5f) + 1u;
#define HAR_LAUNCH_SPLAT(T, W) hipLaunchKernelGGL((k_splat<T, W>), dim3(blocks_for(n)), dim3(kBlock), 0, s, C, seed, spp, log_spp, lane_base, n, result, weights_only, film, jitter, scalar)
    const int w = weights_only ? (scalar ? 2 : 1) : 0;
    if (taps <= 5) { if (w == 2) HAR_LAUNCH_SPLAT(5, 2); else if (w == 1) HAR_LAUNCH_SPLAT(5, 1); else HAR_LAUNCH_SPLAT(5, 0); }
    else { if (w == 2) HAR_LAUNCH_SPLAT(HAR_MAX_FILTER_TAPS, 2); else if (w == 1) HAR_LAUNCH_SPLAT(HAR_MAX_FILTER_TAPS, 1); else HAR_LAUNCH_SPLAT(HAR_MAX_FILTER_TAPS, 0); }
#undef HAR_LAUNCH_SPLAT
}
void launch_pass_jitter(hipStream_t s, uint32_t seed, uint32_t lane_base, uint32_t n, uint32_t pass, float2 *jitter) {
    hipLaunchKernelGGL(k_pass_jitter, dim3(blocks_for(n)), dim3(kBlock), 0, s, seed, lane_base, n, pass, jitter);
}
void launch_develop(hipStream_t s, const float *film, uint32_t npx, float *image, int colour) {
    hipLaunchKernelGGL(k_develop, dim3(blocks_for(npx)), dim3(kBlock), 0, s, film, npx, image, colour);
}
void launch_adjoint_image(hipStream_t s, const float *grad_in, const float *wfilm, uint32_t npx, float *adj) {
    hipLaunchKernelGGL(k_adjoint_image, dim3(blocks_for(npx)), dim3(kBlock), 0, s, grad_in, wfilm, npx, adj);
}
void launch_add(hipStream_t s, const float *src, float *dst, uint32_t n) {
    if (n) hipLaunchKernelGGL(k_add, dim3(blocks_for(n)), dim3(kBlock), 0, s, src, dst, n);
}
void launch_accumulate_stats(hipStream_t s, const uint32_t *counters, uint32_t n_bounces, unsigned long long *totals, uint32_t paths) {
    hipLaunchKernelGGL(k_accumulate_stats, dim3(1), dim3(1), 0, s, counters, n_bounces, totals, paths);
}
void launch_api_intersect(hipStream_t s, const DScene &S, uint32_t n, const float *o, const float *d, const float *maxt, const uint8_t *active, int naive,
                          float *t, float *u, float *v, uint32_t *prim, uint32_t *shape, uint32_t *inst, int *status) {
    dim3 g(blocks_for(n)), b(kBlock);
    if (naive) hipLaunchKernelGGL(k_api_intersect<true>, g, b, 0, s, S, n, o, d, maxt, active, t, u, v, prim, shape, inst, status);
    else hipLaunchKernelGGL(k_api_intersect<false>, g, b, 0, s, S, n, o, d, maxt, active, t, u, v, prim, shape, inst, status);
}
void launch_api_ray_test(hipStream_t s, const DScene &S, uint32_t n, const float *o, const float *d, const float *maxt, const uint8_t *active, int naive, uint8_t *out, int *status) {
    dim3 g(blocks_for(n)), b(kBlock);
    if (naive) hipLaunchKernelGGL(k_api_ray_test<true>, g, b, 0, s, S, n, o, d, maxt, active, out, status);
    else hipLaunchKernelGGL(k_api_ray_test<false>, g, b, 0, s, S, n, o, d, maxt, active, out, status);
}
void launch_api_si(hipStream_t s, const DScene &S, uint32_t n, const float *o, const float *d, const float *t, const float *u, const float *v,
                   const uint32_t *prim, const uint32_t *shape, const uint32_t *inst, uint32_t ray_flags, const uint8_t *active, float *out) {
    hipLaunchKernelGGL(k_api_si, dim3(blocks_for(n)), dim3(kBlock), 0, s, S, n, o, d, t, u, v, prim, shape, inst, ray_flags, active, out);
}
void launch_api_sampler_seed(hipStream_t s, uint32_t seed, uint32_t lane_offset, uint32_t n, uint64_t *state, uint64_t *inc) {
    hipLaunchKernelGGL(k_api_sampler_seed, dim3(blocks_for(n)), dim3(kBlock), 0, s, seed, lane_offset, n, state, inc);
}
/* Scene::sample_emitter(index_sample, active) / Scene::pdf_emitter(index, active) (src/render/scene.cpp:248-279), array-valued */
__global__ void k_api_sample_emitter(DScene S, uint32_t n, const float *sample, const uint8_t *active, uint32_t *index, float *weight, float *reused) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (active && !active[i]) { index[i] = 0u; weight[i] = 0.f; reused[i] = 0.f; return; }
    float w, r;
    index[i] = scene_sample_emitter(S, sample[i], true, w, r); weight[i] = w; reused[i] = r;
}
__global__ void k_api_pdf_emitter(DScene S, uint32_t n, const uint32_t *index, const uint8_t *active, float *pdf) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    pdf[i] = ((active && !active[i]) || index[i] >= S.n_emitters) ? 0.f : scene_pdf_emitter(S, index[i]);
}
void launch_api_sample_emitter(hipStream_t s, const DScene &S, uint32_t n, const float *sample, const uint8_t *active, uint32_t *index, float *weight, float *reused) {
    hipLaunchKernelGGL(k_api_sample_emitter, dim3(blocks_for(n)), dim3(kBlock), 0, s, S, n, sample, active, index, weight, reused);
}
void launch_api_pdf_emitter(hipStream_t s, const DScene &S, uint32_t n, const uint32_t *index, const uint8_t *active, float *pdf) {
    hipLaunchKernelGGL(k_api_pdf_emitter, dim3(blocks_for(n)), dim3(kBlock), 0, s, S, n, index, active, pdf);
}
void launch_api_sampler_next(hipStream_t s, uint32_t n, uint64_t *state, const uint64_t *inc, const uint8_t *active, float *out, int dims) {
    hipLaunchKernelGGL(k_api_sampler_next, dim3(blocks_for(n)), dim3(kBlock), 0, s, n, state, inc, active, out, dims);
}
void launch_api_bsdf_eval_pdf(hipStream_t s, const DScene &S, uint32_t bsdf, const BsdfCtx &ctx, uint32_t n, const float *wi, const float *uv, const float *wo, const uint8_t *active,
                              float *value, float *pdf) {
    hipLaunchKernelGGL(k_api_bsdf_eval_pdf, dim3(blocks_for(n)), dim3(kBlock), 0, s, S, bsdf, ctx, n, wi, uv, wo, active, value, pdf);
}
void launch_api_bsdf_sample(hipStream_t s, const DScene &S, uint32_t bsdf, const BsdfCtx &ctx, uint32_t n, const float *wi, const float *uv, const float *s1, const float *s2,
                            const uint8_t *active, float *wo, float *pdf, float *weight, float *eta, uint32_t *stype, uint32_t *scomp) {
    hipLaunchKernelGGL(k_api_bsdf_sample, dim3(blocks_for(n)), dim3(kBlock), 0, s, S, bsdf, ctx, n, wi, uv, s1, s2, active, wo, pdf, weight, eta, stype, scomp);
}
void launch_api_sensor_ray(hipStream_t s, const DSensor &C, uint32_t n, const float *px, const float *py, float *o, float *d, float *maxt) {
    hipLaunchKernelGGL(k_api_sensor_ray, dim3(blocks_for(n)), dim3(kBlock), 0, s, C, n, px, py, o, d, maxt);
}
void launch_api_film_put(hipStream_t s, const DSensor &C, uint32_t n, const float *px, const float *py, const float *values4, float *film) {
    hipLaunchKernelGGL(k_api_film_put, dim3(blocks_for(n)), dim3(kBlock), 0, s, C, n, px, py, values4, film);
}

} // namespace har
