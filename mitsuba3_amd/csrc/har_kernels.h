/* har_kernels.h -- device buffers + launch wrappers of the hip_ad_rgb kernels (see har_kernels.hip). */
#pragma once
#include <hip/hip_runtime.h>
#include "har_path.h"

#define HAR_LDS_STACK_DEPTH 30      /* traversal stack entries per lane of the stand-alone ray-query kernels (all in LDS): 60 KB/block */
#ifndef HAR_LDS_STACK_SMALL         /* LDS entries per lane of the wavefront traversal kernels: 13 x 8 B x 256 = 26 KB/block -> still 6 blocks/CU (160 KB).  Measured (tools/build_variant.sh A/B): 16 entries (5 blocks/CU) is 12 % slower */
#define HAR_LDS_STACK_SMALL 13
#endif
#ifndef HAR_STACK_SPILL             /* deeper entries of a lane's stack live in HBM (per-thread columns of the integrator workspace): LDS bytes set the occupancy of these
                                       VALU-bound kernels, the depth-first bound of a BVH is rarely reached by a ray, and a spilled push/pop costs one 8-byte access */
#define HAR_STACK_SPILL 20
#endif
#ifndef HAR_TRAV_POLICY
#define HAR_TRAV_POLICY 0          /* Traversal<POLICY>, see har_accel.h; measured: POLICY 2 saves 7-15 % iterations in the model but nothing on the GPU */
#endif
#ifndef HAR_STACK_MARGIN            /* stack entries a scene needs beyond HostScene::stack_need() */
#define HAR_STACK_MARGIN (HAR_TRAV_POLICY == 2 ? HAR_MAX_PARKED : 0)
#endif
#define HAR_REPLAY_CACHE_BOUNCES 12 /* PRB replay cache depth (25 B per lane and bounce); deeper bounces are traced twice */
#define HAR_LDS_GRAD_BSDFS 256     /* constant-albedo (and emitter-radiance) gradient slots accumulated per block in LDS (adjoint resolve) */
#define HAR_LDS_EXTRA_BSDFS 16      /* BSDF records whose alpha / eta / k / slot-1 gradients (15 floats each) are accumulated per block in LDS */
#define HAR_LDS_GRAD_EMITTERS 32    /* emitter-radiance gradients of emission hits accumulated per block in LDS (adjoint shade) */
#define HAR_SHARDS 8                /* XCD-private path queues */
#define HAR_MAX_TRAVERSAL_BLOCKS 2048 /* persistent traversal kernels: enough blocks to fill the chip (<= 8 blocks/CU) */
#define HAR_COUNTER_STRIDE 16       /* u32 stride between shard counters: one 64 B line each */
#define HAR_SPLAT_TILE_FLOATS 8192
#define HAR_MAX_BOUNCE_SLOTS 1026

/* Closest-hit records of a wavefront.  HAR_HIT_INTERLEAVED = 1: ONE 32-byte record per ray -- {t, u, v, prim | shape, inst, -, -} -- addressed through
 * two views of the same buffer (h0 = float4 view, h1 = uint2 view starting 16 bytes in).  The lanes of a persistent traversal wave finish at
 * different times, so each record is written alone: as two arrays (16 B + 8 B per ray, HAR_HIT_INTERLEAVED = 0) every ray dirties two 32-byte
 * sectors of HBM (round 1: 64 B written per ray measured, 2.7x the 24 algorithmic bytes); as one aligned record it dirties exactly one. */
#ifndef HAR_HIT_INTERLEAVED
#define HAR_HIT_INTERLEAVED 1
#endif
#if HAR_HIT_INTERLEAVED
#define HIT0(i) ((size_t) 2 * (i))
#define HIT1(i) ((size_t) 4 * (i))
#else
#define HIT0(i) ((size_t) (i))
#define HIT1(i) ((size_t) (i))
#endif
/* HAR_HIT_MATINFO = 1 (A/B build, default OFF): the two spare words of a closest-hit record carry {index of the face in the shading-triangle array, the mesh's
 * material word} (Accel::mesh_info), added by the traversal kernels when a ray retires, so that the shading kernels start their geometry and BSDF loads from the record
 * instead of from a dependent load of the mesh record.  The idea: the generic shading kernel waits on memory 63 % of its wave-cycles at four waves per SIMD, behind a
 * chain hit -> mesh -> face data / mesh -> BSDF record -> tables; take the mesh record out of both chains.  Measured (profiles/r05_ab_hit_matinfo.txt, bracketed):
 * k_shade 23.80 -> 23.59 ms on materials1m, 11.04 -> 10.94 on instanced1m -- and k_trace_closest 31.47 -> 31.96 ms (one 8-byte load per retiring ray costs more in
 * the issue-bound traversal kernel than the shorter chain saves): forward 1 017 -> 1 012 Mpaths/s.  The chain's depth is not what the kernel waits for. */
#ifndef HAR_HIT_MATINFO
#define HAR_HIT_MATINFO 0
#endif
#ifndef HAR_CLOSEST_RETIRE
#define HAR_CLOSEST_RETIRE 1     /* measured on MI355X (instanced 1M scene): 791.7 -> 797.9 Mpaths/s, k_trace_closest 39.16 -> 38.60 ms per frame */
#endif

namespace har {

/* packed SoA path state: 72 B / path (see store_state in har_kernels.hip) */
struct WaveState { float4 *a0, *a1, *a2, *a3; uint2 *a4; };
/* NEE / gradient items written by `shade`, consumed by `resolve` */
struct ItemArrays { float4 *s0, *s1, *s2, *s3, *s4; };
/* PRB replay cache of ONE bounce, indexed by (lane - lane_base): the adjoint pass re-runs sampling and shading with the same random
 * numbers, so its ray queries are bit-identical to the primal pass's; their results (24 B hit record, 1 B visibility) are kept in HBM
 * between the two passes of a chunk instead of being traced twice.  mode 0 = unused, 1 = write (primal pass), 2 = read (adjoint pass) */
struct ReplayCache { float4 *h0; uint2 *h1; uint8_t *vis; int mode; };
/* PRB replay TAPE (modes 3 = record in the primal pass, 4 = replay in the adjoint pass; the default when the adjoint commits in place).  The lane-
 * indexed cache above makes the adjoint pass read hit records, visibility, L and dL through an ever sparser survivor list (1.6x the algorithmic
 * bytes, round-2 PMC passes) and write the path state a second time.  The tape keeps the primal pass's WAVEFRONTS instead: bounce b's compacted
 * path state (72 B per path, written anyway -- each bounce gets its own buffer instead of a ping-pong pair), its hit records (the closest-hit kernel
 * writes them straight into the tape), one visibility byte per vertex slot, and `next` = the slot the survivor took in bounce b + 1.  The adjoint
 * pass then streams every bounce in the primal's slot order: dense reads, NO compaction (no slot reservation, no state store), and L / dL travel in
 * two slot-ordered arrays (la = L.xyz, dL.x; lb = dL.yz) from slot i to slot next[i].  ReplayCache::vis = the bounce's visibility bytes. */
struct TapeArrays { uint32_t *next; float4 *la_in; float2 *lb_in; float4 *la_out; float2 *lb_out; float4 *rec0, *rec1, *rec2, *rec_em; };
/* RECORD tape (modes 5 = record, 6 = commit; the default without alpha / eta / k gradients).  The state tape above still re-runs the whole shading of every
 * vertex in the adjoint pass (surface interaction, emitter sample, two BSDF evaluations) only to rebuild three small vectors: the vertex's direct radiance,
 * its derivative w.r.t. the colour slot and the relative derivative of the BSDF value.  Here the PRIMAL pass shades with the adjoint flavour of shade_lane once
 * and writes those vectors per vertex slot -- rec0 = {Lr_dir (per unit radiance when the emitter is differentiated), tag}, rec1 = {dLr_dir / d slot0, u},
 * rec2 = {(df / d slot0) / f, v}, rec_em = {emitted radiance met at the vertex} -- and the adjoint pass is k_commit: a streaming kernel that walks the
 * wavefronts in slot order, subtracts, multiplies by dL and files the gradients; it touches no geometry.  `next` word of a slot: bits 0..28 the survivor's slot
 * in the next bounce (HAR_TAPE_DEAD: none), bit 29 the vertex has a shadow ray (its visibility byte is valid), bit 30 it has a record, bit 31 an emission record. */
#define HAR_TAPE_DEAD 0x1fffffffu
#define HAR_TAPE_HAS_RAY 0x20000000u
#define HAR_TAPE_HAS_REC 0x40000000u
#define HAR_TAPE_HAS_EM 0x80000000u
/* Multi-pass rendering (integrator.cpp:280-356): the sampler of lane i keeps its PCG32 state from one pass to the next (sampler->advance()
 * does not reseed).  `rng` holds that state per lane of the chunk (pre-offset to the chunk's first lane; nullptr = single pass, streams are
 * seeded in raygen and dropped at path end); `jitter` receives the pass's pixel jitter per chunk lane for the splat kernel, which otherwise
 * recomputes it from the seed; `pass` = 0 seeds the streams instead of reading `rng`. */
struct PassState { uint64_t *rng; float2 *jitter; uint32_t pass; };
/* Vertex-position gradients of the PRB adjoint (har_shape_grad.h).  `ShapeArrays`: per adjoint item, the geometry record written by `shade`
 * (g0 = shape, prim, b1, b2; g1 = d_in, slot of the lane in the next wavefront; g2 = emitter sample point / direction, flags; g3 = its
 * normal, cos theta_o) and the visibility `resolve` found for its shadow ray.  `ShapeTargets`: grad holds 3 floats per vertex of the
 * differentiated meshes, mesh m starting at float3 offset[m] (-1 = not differentiated). */
/* g0..g6: per adjoint item -- {mesh, prim, b1, b2}, {d_in, next slot}, {emitter sample, flags | instance}, {its normal, -}, {beta mis em_weight, -},
 * {previous vertex: mesh, prim, b1, b2}, {the ray the previous vertex lies on, its instance};  pv0 / pv1: per LANE, the last vertex the path met (what the next
 * vertex files as g5 / g6);  vis: the item's shadow-ray result */
struct ShapeArrays { float4 *g0, *g1, *g2, *g3, *g4, *g5, *g6, *pv0, *pv1; uint8_t *vis; };
/* Texel-gradient queues of the PRB adjoint.  Global float atomics run at the memory side of the chip at a fixed rate (measured: ~57 G atomics/s
 * whatever the address distribution -- a 64^2 and a 4096^2 gradient texture cost the same), and a bitmap albedo needs 12 of them per path vertex
 * (4 bilinear taps x RGB): 44 of the 156 ms of a PRB step on the textured 1M-triangle scene.  Instead, the shading kernel APPENDS one 32-byte
 * record per vertex -- {texel cell, bilinear fractions, gradient} -- to the queue of the texture ROW BAND its cell lies in (one queue per band
 * and XCD shard; slots are reserved per 256-thread block with one atomic per non-empty band, like the path compaction), and k_texel_accumulate
 * gives every (band, shard) queue to a few blocks that add the records' taps into an LDS copy of the band (ds_add_f32) and flush it with one
 * global atomic per texel and channel.  ~10x fewer global atomics; the records cost 64 B of HBM traffic per vertex.
 * Queues are bounded (2x the mean band load): a record that finds its queue full is committed with direct atomics by its lane.
 *   band[tex]  = { first queue of the texture (0xffffffff: not queued -- direct atomics), rows per band }
 *   qinfo[q]   = { texture, first row, rows, width } */
#define HAR_TQ_MAX 64                 /* queues (row bands of all queued textures) */
#define HAR_TQ_LDS_BYTES 24576        /* smallest LDS copy of a band: (rows + 1) x width x 3 accumulators of HAR_TQ_ACC_BYTES */
/* The LDS copy of a band accumulates in 64-bit FIXED POINT: on gfx950 a ds_add_f32 wave instruction takes ~193 cycles of the CU's LDS pipeline whatever its
 * addresses (the float atomic unit retires one lane per 3 cycles), ds_add_u64 takes 12 (profiles/r03_lds_atomic_ubench.txt).  `gmax` = the largest |gradient
 * component| of the records of this launch (float bits, atomicMax by the appending kernels, cleared with the counters): the accumulating block scales its
 * share so that n records of that size cannot overflow 2^62.  (Integer sums are order-independent within a block; the blocks' flushes and the overflow
 * records still meet in global float atomics, so the texture gradient as a whole is not bit-reproducible.)
 * A non-finite gradient sets gmax to +inf and the launch falls back to float atomics (NaN / inf reach the texture as before). */
#define HAR_TQ_ACC_BYTES 8
struct TexelQueues { float4 *rec; uint32_t *count; const uint2 *band; const uint4 *qinfo; uint32_t nq, cap; uint32_t *gmax; };
/* Per-material shading queues (north_star: "material-sorted BSDF megakernels"; the reference's dispatch point is the BSDF virtual call of
 * path.cpp:233,266-267).  After the closest-hit launch of a bounce, k_classify deals the shard's paths to HAR_MAT_CLASSES index lists by the BSDF MODEL
 * of the surface they hit (har_bsdf.h: six models + "twosided pair of two models"; escaped paths ride in `miss_class`), in slot order within a
 * 256-path tile, and ONE k_shade launch per non-empty class -- a kernel in which only that model's code exists (registers, no scratch, no divergence
 * over models) -- shades its list.  idx: class c, shard s -> idx[c * lanes + s * shard_cap + k] = slot of the k-th path; count: (c * HAR_SHARDS + s) * stride. */
struct MaterialQueues { uint32_t *idx; uint32_t *count; uint32_t lanes, miss_class; };
#define HAR_SHAPE_INST_SHIFT 8           /* geometry records: bits 8..31 of the flags word = instance index + 1 of a vertex on instanced geometry (0: top-level) */
struct ShapeTargets { const int32_t *offset; float *grad; uint32_t n_verts; const int32_t *inst_slot; float *inst_grad; uint32_t n_insts;
                      float *grad_nrm = nullptr;       /* meshes with vertex normals: adjoints of the vertex normals, laid out like `grad` (first stage, see launch_normals_adjoint) */ /* per instance: slot (12 floats each in inst_grad) or -1; null = no instance is differentiated */ };
#define HAR_LDS_GRAD_INSTS 128        /* instance-transform gradients accumulated per block in LDS (6 KB) */
#define HAR_LDS_GRAD_VERTS 1024       /* entries of the per-block direct-mapped LDS cache of vertex gradients (k_shape_adjoint; power of two) */

void launch_raygen(int mode, hipStream_t s, const DSensor &C, uint32_t seed, uint32_t spp, uint32_t log_spp, uint32_t lane_base, uint32_t n,
                   uint32_t shard_cap, const WaveState &out, float4 *result, uint32_t *count, const float *adj, float4 *dL, const PassState &ps = PassState{ nullptr, nullptr, 0 },
                   bool lite = false);      /* lite: only the rays are stored (k_raygen<.., LITE>; the first shading launch then carries HAR_SHADE_FIRST_VERTEX) */
/* har_integrator_sample: the wavefront of n caller-supplied rays (SoA arrays of n_total rays, this chunk starts at `first`), see k_raygen_rays */
void launch_raygen_rays(hipStream_t s, uint32_t seed, uint32_t lane_base, uint32_t n, uint32_t n_total, uint32_t first, const float *o, const float *d, const float *maxt,
                        const uint64_t *state, const uint8_t *active, uint32_t shard_cap, const WaveState &out, float4 *result, uint32_t *count);
void launch_sample_out(hipStream_t s, uint32_t n, uint32_t n_total, uint32_t first, const float4 *result, const float *valid_lane, int zero_invalid, float *rgb, uint8_t *valid,
                       const uint8_t *active, uint32_t seed, uint32_t lane_base, const uint64_t *state_in, uint64_t *state_out);
/* `spill` = nullptr: the scene's depth-first bound fits the LDS stack and the kernels without the HBM spill path run (3 % faster) */
/* Packets (64 consecutive rays of a shard) the wave-shared descent gave up on: list[shard * stride + k] = first ray of the k-th one, count[shard * HAR_COUNTER_STRIDE]
 * their number, cursor = the work cursor of the per-lane launch that serves them, shard_count = the wavefront's rays per shard (the last packet may be partial) */
struct PacketList { uint32_t *list, *count, *cursor; const uint32_t *shard_count; uint32_t stride; };
/* pl != nullptr: only the rays of the listed packets (count / cursor are then pl->count / pl->cursor) */
void launch_trace_closest(hipStream_t s, uint32_t grid, uint2 *spill, const Accel &A, const uint32_t *count, uint32_t *cursor, uint32_t shard_cap,
                          const WaveState &in, float4 *h0, uint2 *h1, int *status, const PacketList *pl = nullptr);
void launch_trace_packet(hipStream_t s, uint32_t grid, const Accel &A, const uint32_t *count, uint32_t *cursor, uint32_t shard_cap, const WaveState &in, float4 *h0, uint2 *h1,
                         const PacketList &pl, uint32_t budget);
void launch_shade(int mode, hipStream_t s, uint32_t grid, const DScene &S, const ShadeParams &P, uint32_t lane_base, uint32_t shard_cap, const uint32_t *count_in,
                  const WaveState &in, const float4 *h0, const uint2 *h1, const WaveState &out, uint32_t *count_out, const ItemArrays &items,
                  uint32_t *item_count, float4 *result, const ReplayCache &rc, uint64_t *pass_rng = nullptr, const float4 *dL = nullptr, float *grad_slots = nullptr,
                  const ShapeArrays *geo = nullptr, float *const *grad_tex_inline = nullptr,       /* grad_tex_inline (adjoint, cached bounce): commit the vertex adjoint in place, no items */
                  const TexelQueues *tq = nullptr,                                                 /* ... with the texel gradients going through the queues */
                  float *grad_extra = nullptr,                                                     /* ... plus 15 floats per BSDF record: d / d {alpha_u, alpha_v, eta, k, slot 1} */
                  const MaterialQueues *mq = nullptr, uint32_t mat_class = 0,                      /* shade the paths of ONE material class (path / prb primal), see MaterialQueues */
                  const TapeArrays *tape = nullptr);                                               /* rc.mode 3 / 4: the bounce's tape arrays */
/* adjoint pass in tape mode: L (the primal pass's result) and dL (gathered from the adjoint image over the lane's footprint) into bounce 0's slot order */
void launch_tape_begin(hipStream_t s, const DSensor &C, uint32_t seed, uint32_t spp, uint32_t log_spp, uint32_t lane_base, uint32_t n, uint32_t shard_cap,
                       const float4 *result, const float *adj, float4 *la, float2 *lb, float4 *dL_out = nullptr, const float4 *dL_in = nullptr);
/* adjoint pass of the record tape: one bounce's vertices, in slot order (see TapeArrays) */
void launch_commit(hipStream_t s, uint32_t grid, const DScene &S, uint32_t shard_cap, const uint32_t *count_in, const TapeArrays &tape, const uint8_t *vis,
                   float *grad_slots, float *const *grad_tex, const TexelQueues *tq,
                   const float4 *result = nullptr, const float4 *dL = nullptr);       /* bounce 0: L / dL straight from the per-lane arrays (the slot <-> lane map is arithmetic) */
void launch_classify(hipStream_t s, uint32_t grid, const DScene &S, uint32_t shard_cap, const uint32_t *count_in, const float4 *h0, const uint2 *h1, const MaterialQueues &mq);
/* the records of one bounce -> LDS band copies -> grad_tex (see TexelQueues); blocks_per_queue blocks share a queue */
void launch_texel_accumulate(hipStream_t s, const TexelQueues &tq, float *const *grad_tex, uint32_t blocks_per_queue, uint32_t lds_bytes);
void launch_resolve(int mode, hipStream_t s, uint32_t grid, uint2 *spill, const DScene &S, const uint32_t *item_count, uint32_t *cursor, uint32_t shard_cap, const ItemArrays &items,
                    float4 *result, const float4 *dL, float *grad_refl, float *const *grad_tex, int *status, const ReplayCache &rc, uint8_t *item_vis = nullptr,
                    int fwd = 0, const TexelQueues *tq = nullptr);      /* tq: the cached adjoint resolve files texel gradients in the band queues; fwd: forward mode (render_forward) -- grad_refl / grad_tex are the parameters' tangents, dL accumulates */
/* hide_emitters: one round of Integrator::skip_area_emitters over the camera-ray hits (first = 1: scan h0 / h1 of the wavefront `ray_o / ray_d`;
 * first = 0: commit the re-traced hits `hit0 / hit1` of the continuation list `ray_o / ray_d`); continuation rays are appended to `dst_*` */
void launch_skip_emitters(hipStream_t s, uint32_t grid, const DScene &S, int first, uint32_t shard_cap, const uint32_t *count_in, const float4 *ray_o, const float4 *ray_d,
                          const float4 *hit0, const uint2 *hit1, float4 *h0, uint2 *h1, float4 *dst_o, float4 *dst_d, uint32_t *dst_count);
/* adjoint of the geometry-attached terms of the items of one bounce.  `next` / `h0` / `h1` / `rc_next`: wavefront and ray-query results of the
 * FOLLOWING bounce (has_next = 0: the last bounce); runs after that bounce's trace and before its shade, while `result` still holds L of this one */
void launch_shape_adjoint(hipStream_t s, uint32_t grid, const DScene &S, const uint32_t *item_count, uint32_t shard_cap, const ItemArrays &items, const ShapeArrays &geo,
                          const float4 *result, const float4 *dL, int has_next, const WaveState &next, const float4 *h0, const uint2 *h1, const ReplayCache &rc_next,
                          const ShapeTargets &T);
/* second stage of the derivative through regenerated vertex normals (Mesh::compute_normals, mesh.cpp:1216-1267) for mesh `mesh`: acc (3 floats per vertex of the mesh,
 * scratch) <- the angle-weighted normal sums, then every face pushes the adjoints of its three vertices' normals (nbar) to its three positions (grad += ...) */
void launch_normals_adjoint(hipStream_t s, const DScene &S, uint32_t mesh, uint32_t face_count, uint32_t vertex_count, float *acc, const float *nbar, float *grad);
void launch_splat(hipStream_t s, const DSensor &C, uint32_t seed, uint32_t spp, uint32_t log_spp, uint32_t lane_base, uint32_t n,
                  const float4 *result, int weights_only, float *film, const float2 *jitter = nullptr, const float *scalar = nullptr);   /* weights_only + scalar: w * scalar[i] */
/* alpha flags of the camera samples of a wavefront (1 = valid), see k_alpha_flags */
void launch_alpha_flags(hipStream_t s, uint32_t grid, uint32_t shard_cap, const uint32_t *count_in, const WaveState &in, const float4 *h0, uint32_t lane_base, float miss_value, float *alpha);
void launch_pass_jitter(hipStream_t s, uint32_t seed, uint32_t lane_base, uint32_t n, uint32_t pass, float2 *jitter);
void launch_develop(hipStream_t s, const float *film, uint32_t npx, float *image, int colour = 0);
void launch_adjoint_image(hipStream_t s, const float *grad_in, const float *wfilm, uint32_t npx, float *adj);
void launch_add(hipStream_t s, const float *src, float *dst, uint32_t n);      /* dst[i] += src[i] */
void launch_accumulate_stats(hipStream_t s, const uint32_t *counters, uint32_t n_bounces, unsigned long long *totals, uint32_t paths);

/* array-valued plugin surface: `active` = the reference's Mask argument (NULL = all lanes) */
void launch_api_intersect(hipStream_t s, const DScene &S, uint32_t n, const float *o, const float *d, const float *maxt, const uint8_t *active, int naive,
                          float *t, float *u, float *v, uint32_t *prim, uint32_t *shape, uint32_t *inst, int *status);
void launch_api_ray_test(hipStream_t s, const DScene &S, uint32_t n, const float *o, const float *d, const float *maxt, const uint8_t *active, int naive, uint8_t *out, int *status);
void launch_api_si(hipStream_t s, const DScene &S, uint32_t n, const float *o, const float *d, const float *t, const float *u, const float *v,
                   const uint32_t *prim, const uint32_t *shape, const uint32_t *inst, uint32_t ray_flags, const uint8_t *active, float *out);
void launch_api_sample_emitter(hipStream_t s, const DScene &S, uint32_t n, const float *sample, const uint8_t *active, uint32_t *index, float *weight, float *reused);
void launch_api_pdf_emitter(hipStream_t s, const DScene &S, uint32_t n, const uint32_t *index, const uint8_t *active, float *pdf);
void launch_api_sampler_seed(hipStream_t s, uint32_t seed, uint32_t lane_offset, uint32_t n, uint64_t *state, uint64_t *inc);
void launch_api_sampler_next(hipStream_t s, uint32_t n, uint64_t *state, const uint64_t *inc, const uint8_t *active, float *out, int dims);
void launch_api_bsdf_eval_pdf(hipStream_t s, const DScene &S, uint32_t bsdf, const BsdfCtx &ctx, uint32_t n, const float *wi, const float *uv, const float *wo, const uint8_t *active,
                              float *value, float *pdf);
void launch_api_bsdf_sample(hipStream_t s, const DScene &S, uint32_t bsdf, const BsdfCtx &ctx, uint32_t n, const float *wi, const float *uv, const float *s1, const float *s2,
                            const uint8_t *active, float *wo, float *pdf, float *weight, float *eta, uint32_t *stype, uint32_t *scomp);
void launch_api_sensor_ray(hipStream_t s, const DSensor &C, uint32_t n, const float *px, const float *py, float *o, float *d, float *maxt);
void launch_api_film_put(hipStream_t s, const DSensor &C, uint32_t n, const float *px, const float *py, const float *values4, float *film);

} // namespace har
