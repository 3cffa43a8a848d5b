/*
 * har_capi.hip -- implementation of the C ABI declared in include/hip_ad_rgb.h.
 *
 * Host-side driver of the wavefront integrators: SamplingIntegrator::render
 * (src/render/integrator.cpp:151-396, JIT branch) and RBIntegrator.render_backward
 * (src/python/python/ad/integrators/common.py:625-783) re-expressed as an
 * asynchronous sequence of HIP kernel launches on the caller's stream.  There is
 * no CPU fallback anywhere in this file: without a HIP device every entry point
 * that touches the GPU fails with an error.
 */
#include "../../include/hip_ad_rgb.h"
#include "har_kernels.h"
#include "har_scene_host.h"
#include "har_refit_launch.h"
#include "har_vertex_update.h"

#include <algorithm>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

using namespace har;

static thread_local std::string g_error;
static int fail(const std::string &msg) { g_error = msg; return 1; }
int har_set_error(const std::string &msg) { return fail(msg); }       /* used by har_mesh_io.cpp */

#define HIP_TRY(expr)                                                                          \
    do { hipError_t _e = (expr); if (_e != hipSuccess) {                                       \
        return fail(std::string(#expr) + ": " + hipGetErrorString(_e)); } } while (0)

/*
 * Device allocations of the library.  HAR_DEBUG_GUARD = 1 | 2 (debug switch, tests/test_gpu_parity.py::test_guarded_*): every buffer gets a private
 * virtual-address reservation (hipMemAddressReserve / hipMemCreate / hipMemMap) with UNMAPPED ranges on both sides, and sits flush against the
 * end (1) or the start (2) of its mapped pages -- an access past the end (before the start) of any workspace or scene array is then a GPU
 * memory-access fault whatever the neighbouring allocations are, instead of a silent read of another array.  Default: plain hipMalloc.
 */
namespace {
struct GuardedBlock { void *base; size_t reserved; void *mapped; size_t mapped_bytes; hipMemGenericAllocationHandle_t handle; };
std::map<void *, GuardedBlock> g_guarded;
std::mutex g_guarded_mutex;
int guard_mode() { static const int m = getenv("HAR_DEBUG_GUARD") ? atoi(getenv("HAR_DEBUG_GUARD")) : 0; return m; }
/* har_set_allocator: the host's device allocator (the Python host installs PyTorch's caching allocator, so that workspaces and scene arrays show up in -- and are
 * reused through -- the process's one memory pool).  Every block remembers who has to free it, so the hook can be changed while blocks are alive. */
HarAllocFn g_alloc_fn = nullptr; HarFreeFn g_free_fn = nullptr; void *g_alloc_user = nullptr;
struct HostBlock { HarFreeFn free_fn; void *user; };
std::map<void *, HostBlock> g_host_blocks;
std::mutex g_alloc_mutex;
}
int har_set_allocator(HarAllocFn alloc_fn, HarFreeFn free_fn, void *user) {
    if ((alloc_fn == nullptr) != (free_fn == nullptr)) return fail("har_set_allocator: give both functions, or neither (hipMalloc / hipFree)");
    std::lock_guard<std::mutex> lock(g_alloc_mutex);
    g_alloc_fn = alloc_fn; g_free_fn = free_fn; g_alloc_user = user;
    return 0;
}
static hipError_t dev_alloc(void **out, size_t bytes) {
    bytes = std::max<size_t>(bytes, 1);
    if (!guard_mode()) {
        HarAllocFn fn; HarFreeFn ffn; void *user;
        { std::lock_guard<std::mutex> lock(g_alloc_mutex); fn = g_alloc_fn; ffn = g_free_fn; user = g_alloc_user; }
        if (!fn) return hipMalloc(out, bytes);
        void *p = fn(bytes, user);
        if (!p) return hipErrorOutOfMemory;
        std::lock_guard<std::mutex> lock(g_alloc_mutex);
        g_host_blocks[p] = HostBlock{ ffn, user }; *out = p;
        return hipSuccess;
    }
    int dev = 0; hipError_t e = hipGetDevice(&dev); if (e != hipSuccess) return e;
    hipMemAllocationProp prop{}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
    size_t gran = 0; e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum); if (e != hipSuccess) return e;
    gran = std::max<size_t>(gran, 4096);
    GuardedBlock B{};
    B.mapped_bytes = (bytes + gran - 1) / gran * gran;
    const size_t guard = std::max<size_t>(gran, (size_t) 64 << 20);          /* 64 MB of nothing on either side */
    B.reserved = B.mapped_bytes + 2 * guard;
    e = hipMemAddressReserve(&B.base, B.reserved, gran, nullptr, 0); if (e != hipSuccess) return e;
    e = hipMemCreate(&B.handle, B.mapped_bytes, &prop, 0); if (e != hipSuccess) { (void) hipMemAddressFree(B.base, B.reserved); return e; }
    B.mapped = (char *) B.base + guard;
    e = hipMemMap(B.mapped, B.mapped_bytes, 0, B.handle, 0);
    if (e == hipSuccess) {
        hipMemAccessDesc acc{}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
        e = hipMemSetAccess(B.mapped, B.mapped_bytes, &acc, 1);
    }
    if (e != hipSuccess) { (void) hipMemRelease(B.handle); (void) hipMemAddressFree(B.base, B.reserved); return e; }
    /* end-flush placement keeps 256-byte alignment (every array of the library is accessed with <= 16-byte vectors) */
    void *user = guard_mode() == 2 ? B.mapped : (char *) B.mapped + (B.mapped_bytes - bytes) / 256 * 256;
    std::lock_guard<std::mutex> lock(g_guarded_mutex);
    g_guarded[user] = B; *out = user;
    return hipSuccess;
}
static void dev_free(void *p, bool device_is_idle = false) {      /* device_is_idle: the caller has synchronised the device (free_ws: once for all its blocks) */
    if (!p) return;
    if (guard_mode()) {
        std::lock_guard<std::mutex> lock(g_guarded_mutex);
        auto it = g_guarded.find(p);
        if (it != g_guarded.end()) {
            const GuardedBlock B = it->second; g_guarded.erase(it);
            (void) hipDeviceSynchronize();
            (void) hipMemUnmap(B.mapped, B.mapped_bytes); (void) hipMemRelease(B.handle);
            /* the address range stays reserved for the life of the process (quarantine): a stale pointer faults instead of reaching a later allocation */
            return;
        }
    }
    {
        HostBlock B{ nullptr, nullptr };
        {
            std::lock_guard<std::mutex> lock(g_alloc_mutex);
            auto it = g_host_blocks.find(p);
            if (it != g_host_blocks.end()) { B = it->second; g_host_blocks.erase(it); }
        }
        /* hipFree synchronises the device before it releases a block; a pooling allocator hands the block to its next user at once, so do the same here
         * (the library's private streams may still be reading it) */
        if (B.free_fn) { if (!device_is_idle) (void) hipDeviceSynchronize(); B.free_fn(p, B.user); return; }
    }
    (void) hipFree(p);
}

template <typename T> static hipError_t upload(const std::vector<T> &v, const T **dst, std::vector<void *> &owned) {
    *dst = nullptr;
    size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
    void *p = nullptr;
    hipError_t e = dev_alloc(&p, bytes);
    if (e != hipSuccess) return e;
    owned.push_back(p);
    if (!v.empty()) { e = hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice); if (e != hipSuccess) return e; }
    *dst = (const T *) p;
    return hipSuccess;
}

static uint64_t g_scene_serial = 0;
struct HarSceneImpl {
    uint64_t serial = ++g_scene_serial;       /* identifies the scene in per-integrator caches (a freed scene's address may be reused) */
    uint32_t mat_classes = 0, mat_miss_class = 0;   /* material classes (MaterialQueues) the scene's BSDF records fall into: bit mask, and the class escaped paths ride in */
    HostScene hs;
    DScene ds{};
    std::vector<void *> owned;
    std::vector<float *> tex_dev;
    DBsdf *d_bsdfs = nullptr;
    /* har_scene_set_*_device: the host mirrors (hs.textures[k].data, hs.bsdfs, hs.emitters) that no longer hold the device's values */
    std::vector<uint8_t> tex_host_stale; bool bsdf_host_stale = false, emitter_host_stale = false;
    /* incremental accel updates (har_scene_update_instances / har_scene_update_vertices): capacity of the node array (BLAS nodes + the largest TLAS the instances can
     * need), scratch of the device refit -- boxes of the triangle records and of the nodes, the nodes of every BLAS by depth, one surface-area accumulator per BLAS
     * (+ the root box read-back) */
    size_t nodes_cap = 0;
    RefitBox *tri_box = nullptr, *node_box = nullptr; uint32_t *d_refit_order = nullptr; float *d_area = nullptr;
    double last_refit_cost = 0.0, last_refit_ratio = 1.0;
    float *d_emitter_distr = nullptr; DTexture *d_textures = nullptr;
    /* device-resident vertex updates (har_scene_update_vertices_device): per mesh -- the host mirror of its vertex records (hs.verts, hs.shade_tris) is older than
     * the device's; its normals are the ones k_vertex_normals regenerated; its corner list (har_vertex_update.h) on the device.  `pend`: one pinned record the last
     * update's refit writes its figures to (surface-area sum, root box, non-finite flag) behind `pend_ev` -- read by the NEXT call, nothing waits for it */
    std::vector<uint8_t> verts_host_stale, normals_regenerated;
    std::vector<uint32_t *> d_corner_begin, d_corners;
    struct PendingRefit { float area; RefitBox root; uint32_t bad; } *pend = nullptr;
    hipEvent_t pend_ev = nullptr; bool pend_active = false; BlasInfo *pend_blas = nullptr; uint32_t *d_bad = nullptr;
    /* device refit of the instance level (instanced meshes): TLAS nodes by depth, {first vertex, count} of every leaf record's group, the records' boxes; valid for tlas_serial */
    uint32_t *d_tlas_order = nullptr; uint2 *d_inst_vrange = nullptr; RefitBox *d_inst_box = nullptr; uint64_t d_tlas_serial = 0; size_t d_tlas_cap = 0, d_inst_cap = 0;
    /* device-resident instance transforms (har_scene_update_instances_device): TLAS leaf record of every instance, the "singular / not finite" flag of the last update in a
     * pinned word behind an event (read by the next call), and whether hs.insts (the host mirror of the transforms) is older than the device's */
    uint32_t *d_rec_of = nullptr; size_t d_rec_of_cap = 0; uint32_t *pend_inst = nullptr, *d_bad_inst = nullptr; hipEvent_t pend_inst_ev = nullptr; bool pend_inst_active = false;
    bool insts_host_stale = false;
    hipStream_t last_push_stream = nullptr; bool last_push_valid = false;      /* stream of the last har_scene_set_*_device copy: the blocking host read-backs order themselves after it */
    ~HarSceneImpl() {
        if (pend) (void) hipHostFree(pend); if (pend_ev) (void) hipEventDestroy(pend_ev);
        if (pend_inst) (void) hipHostFree(pend_inst); if (pend_inst_ev) (void) hipEventDestroy(pend_inst_ev);
    }
};

struct HarIntegratorImpl {
    int type = HAR_INTEGRATOR_PATH;
    /* lanes per wavefront chunk (multiple of 2048).  Every chunk pays ~3.4 ms of kernel tails (26 launches that each wait for their slowest wave), so
     * chunks are as large as HBM comfortably allows: 2^26 lanes = 15.6 GB of forward workspace, 32 GB with the adjoint items and the replay cache
     * (measured on the 1M-triangle scene, 67 M lanes: 16 M-lane chunks 708, 32 M 758, one 64 M chunk 783 Mpaths/s) */
    uint32_t max_depth = 0, rr_depth = 5, chunk = 1u << 26;
    bool hide_emitters = false;           /* Integrator property (integrator.cpp:29) */
    bool forward_mode = false;            /* har_render_forward in progress: the adjoint kernels read tangents and accumulate differential radiance */
    float *alpha_film = nullptr;          /* user buffer (DEVICE, H x W x 4: channel 3 accumulates w * alpha) of har_integrator_set_alpha_film, or null */
    uint32_t film_row0 = 0, film_rows = 0; /* har_integrator_set_film_window: the film buffers of har_render hold rows [film_row0, film_row0 + film_rows) of the crop window (0 rows = all) */
    float *alpha_lane = nullptr;          /* alpha value per lane of the chunk */
    uint32_t *skip_counters = nullptr;    /* hide_emitters: count + cursor of the two continuation lists of skip_area_emitters */
    uint32_t *pk_list = nullptr, *pk_counters = nullptr;      /* wave-shared descent of the camera rays (k_trace_packet): the packets left to the per-lane kernel, their count + cursor */
    // workspace
    uint32_t ws_lanes = 0; bool ws_adjoint = false; uint32_t shard_cap = 0;
    std::vector<void *> owned;
    WaveState st[2]{};
    float4 *h0 = nullptr; uint2 *h1 = nullptr; float4 *hit_scratch = nullptr;      /* hit records (32 B per lane, two views); scratch records of hide_emitters */
    ItemArrays items{};
    float4 *result = nullptr, *dL = nullptr;
    /* PRB replay cache (see ReplayCache): cache_bounces arrays of ws_lanes entries each */
    float4 *rc_h0 = nullptr; uint2 *rc_h1 = nullptr; uint8_t *rc_vis = nullptr; uint32_t cache_bounces = 0; bool use_cache = true;
    /* PRB replay tape (TapeArrays, har_kernels.h): per bounce the wavefront's path state, its hit records, a visibility byte and the next-slot word per
     * vertex slot; two slot-ordered (L, dL) array pairs that alternate from bounce to bounce */
    int ws_tape = 0 /* 0 none, 1 state tape, 2 record tape */; uint32_t tape_bounces = 0;
    float4 *tape_rec[4] = { nullptr, nullptr, nullptr, nullptr };      /* record tape: rec0, rec1, rec2, rec_em (lanes x bounces each) */
    WaveState tape_st[HAR_REPLAY_CACHE_BOUNCES + 1]{}; float4 *tape_h0 = nullptr; uint8_t *tape_vis = nullptr; uint32_t *tape_next = nullptr;
    float4 *tape_la[2] = { nullptr, nullptr }; float2 *tape_lb[2] = { nullptr, nullptr };
    float *adj = nullptr; size_t adj_floats = 0;
    float *grad_slots = nullptr; size_t grad_slots_cap = 0;   /* adjoint accumulators: (bsdf_count + emitter_count) x 3 */
    float *grad_emitters = nullptr;       /* user buffer (DEVICE, emitter_count x 3) of har_integrator_set_grad_emitters, or null */
    float *grad_bsdf_params = nullptr;    /* user buffer (DEVICE, bsdf_count x 15) of har_integrator_set_grad_bsdf_params, or null */
    bool grad_light_texels = false;       /* har_integrator_set_grad_light_texels: the texels of bitmap `radiance` textures of area lights are differentiated (into their entries of grad_textures) */
    /* vertex-position gradients (har_integrator_set_grad_positions): user buffers per top-level mesh, the flat accumulation buffer + offsets */
    bool shape_on = false; std::vector<float *> pos_user; std::vector<int32_t> pos_offset; std::vector<uint32_t> pos_count;
    int32_t *d_pos_offset = nullptr; float *grad_pos = nullptr; uint32_t pos_verts = 0; ShapeArrays geo{};
    /* differentiated meshes WITH vertex normals: adjoints of the vertex normals and scratch for the normal sums, laid out like grad_pos (har_shape_grad.h) */
    bool pos_smooth = false; float *grad_nrm = nullptr, *nrm_acc = nullptr;
    uint64_t pos_checked_scene = 0; std::vector<uint8_t> pos_checked;      /* meshes of scene `pos_checked_scene` whose normals were found to be the regenerated ones */
    /* instance to_world gradients (har_integrator_set_grad_instances): user buffer (DEVICE, instance_count x 12), per-instance slot table, accumulation buffer */
    float *inst_user = nullptr; uint32_t inst_count = 0; int32_t *d_inst_slot = nullptr; float *grad_inst = nullptr;
    bool material_queues = false;         /* har_integrator_set_material_queues */
    int packet_tracing = -1;              /* har_integrator_set_packet_tracing: -1 automatic, 0 off, 1 every first closest-hit launch */
    int bw_tape_max = 2; uint32_t bw_chunk_max = 0xffffffffu;      /* render_backward: what the last out-of-memory fallback settled on (tape kind, chunk lanes) */
    uint64_t bw_job_key = 0; uint32_t bw_calls_since_stepdown = 0;   /* ... for which job (scene, film, lanes), and how many calls ago */
    uint32_t *mq_idx = nullptr, *mq_count = nullptr;      /* per-material shading queues (MaterialQueues): HAR_MAT_CLASSES index lists of ws_lanes entries, their counters */
    uint2 *stack_spill = nullptr;         /* HBM part of the traversal stacks: HAR_STACK_SPILL entries per thread of the largest traversal grid */
    /* multi-pass rendering: sampler state per lane of the rendered lane range, pixel jitter per chunk lane (see PassState) */
    uint32_t samples_per_pass = 0xffffffffu;
    uint64_t *pass_rng = nullptr; size_t pass_rng_cap = 0; float2 *pass_jitter = nullptr; size_t pass_jitter_cap = 0;
    uint32_t *counters = nullptr;
    unsigned long long *totals = nullptr;
    int *status = nullptr;
    float **d_grad_tex = nullptr; size_t grad_tex_cap = 0;
    /* staging of the per-call pointer table (upload_pointer_table): a ring of pinned slots with one event each */
    void **ptr_ring = nullptr; size_t ptr_ring_cap = 0; hipEvent_t ptr_ring_ev[8]{}; bool ptr_ring_used[8]{}; uint32_t ptr_ring_next = 0;
    /* texel-gradient queues of the adjoint pass (TexelQueues, har_kernels.h): records, counters, band tables; built for `tq_scene` */
    TexelQueues tq{ nullptr, nullptr, nullptr, nullptr, 0u, 0u, nullptr }; uint64_t tq_scene = 0; uint32_t tq_lanes = 0, tq_lds = 0;
    // profiling
    /* Per-launch HIP events of the frames rendered since har_integrator_set_profiling(1).  An event is NEVER re-recorded while an earlier record of it
     * may still be pending: every frame (render_range / backward_range call) takes its own event set from a ring, and a set is only reused after its
     * last event has completed and its durations have been folded into the accumulators -- a render loop that enqueues tens of frames without a
     * synchronisation (bench.py) therefore neither waits nor touches in-flight events. */
    bool profiling = false;
    struct EventSet { std::vector<hipEvent_t> ev; std::vector<int> cls; size_t used = 0; };
    std::vector<EventSet> sets; size_t cur_set = 0;
    double acc_ms[8] = { 0 }; uint64_t acc_launches[8] = { 0 }; uint64_t acc_frames = 0;
    hipStream_t last_stream = nullptr;
    /* two-stream mode: a job of >= HAR_DUAL_MIN_LANES lanes is cut in two halves that run concurrently -- this integrator on the caller's stream, a
     * private twin (own workspace) on `side_stream`.  Every persistent traversal launch ends with a tail of a few hundred microseconds in which
     * the chip waits for the launch's longest rays (a chain of dependent node fetches); with two independent launch sequences in flight the blocks
     * of one fill the CUs the other's tail leaves idle. */
    HarIntegratorImpl *twin = nullptr; bool twin_used = false;
    hipStream_t side_stream = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_stagger = nullptr; bool stagger_record = false;      /* staggered halves (HAR_DUAL_STAGGER): recorded behind the first half's first closest-hit launch, the second half starts there */
    /* shadow-ray overlap (small jobs): bounce b's shadow rays (k_resolve) do not depend on bounce b + 1's closest-hit rays (k_trace_closest) -- both only need
     * bounce b's shading -- so k_resolve runs on `aux_stream` next to the trace launch (and, with the second item set below, next to bounce b + 1's shading too).  One traversal
     * tail per bounce instead of two (see run_chunk). */
    hipStream_t aux_stream = nullptr; hipEvent_t ev_shaded = nullptr, ev_resolved = nullptr, ev_resolved2 = nullptr;
    /* ... with a second set of item arrays and a second radiance accumulator the shadow rays of bounce b only have to be done before bounce b + 2 is SHADED (the item
     * set is free again); `result2` collects what they add and is folded into `result` at the end of the chunk */
    ItemArrays items2{}; float4 *result2 = nullptr;
    void free_ws();
};
/* One device synchronisation for the whole workspace: blocks handed back to a pooling allocator (har_set_allocator) are reused at once, and the library's private
 * streams may still be reading them -- the invariant is "no block of this library is freed while the device runs"; dev_free keeps it per block for single frees. */
void HarIntegratorImpl::free_ws() { if (!owned.empty()) (void) hipDeviceSynchronize(); for (void *p : owned) dev_free(p, true); owned.clear(); ws_lanes = 0; }
#define HAR_DUAL_MIN_LANES (1u << 20)
#define HAR_DUAL_MAX_LANES (1u << 24)
#define HAR_OVERLAP_MAX_LANES (1u << 25)
#ifndef HAR_LATE_OVERLAP_DEFAULT          /* first bounce whose shadow rays run next to the following bounce's closest-hit rays in jobs above HAR_OVERLAP_MAX_LANES (run_chunk) */
#define HAR_LATE_OVERLAP_DEFAULT(rr_depth) 0xffffffffu
#endif

namespace {

void prof_mark(HarIntegratorImpl *I, hipStream_t s, int cls);
enum { CLS_RAYGEN = 0, CLS_TRACE = 1, CLS_SHADE = 2, CLS_RESOLVE = 3, CLS_SPLAT = 4, CLS_OTHER = 6, CLS_START = 7 };

/* set by every failed workspace allocation, cleared by whoever handles it (render_backward's step-down): the condition "out of device memory", as a flag rather
 * than as a substring of the error text */
thread_local bool g_alloc_failed = false;
template <typename T> int ws_alloc(HarIntegratorImpl *I, T **p, size_t count) {
    void *q = nullptr;
    hipError_t e = dev_alloc(&q, std::max<size_t>(count, 1) * sizeof(T));
    if (e != hipSuccess) { g_alloc_failed = true; return fail(std::string("hipMalloc(workspace): ") + hipGetErrorString(e)); }
    I->owned.push_back(q); *p = (T *) q;
    return 0;
}

uint32_t bounce_limit(const HarIntegratorImpl *I) { return std::min<uint32_t>(I->max_depth, HAR_MAX_BOUNCE_SLOTS - 2); }

int ensure_workspace(HarIntegratorImpl *I, uint32_t lanes, bool adjoint, int tape = 0) {
    if (I->ws_lanes >= lanes && (I->ws_adjoint || !adjoint) && (!adjoint || I->ws_tape == tape)) {
        if (I->alpha_film && !I->alpha_lane) return ws_alloc(I, &I->alpha_lane, I->ws_lanes);      /* `rgba` film on an existing workspace */
        return 0;
    }
    I->free_ws();
    I->counters = nullptr; I->totals = nullptr; I->status = nullptr; I->adj = nullptr; I->adj_floats = 0; I->d_grad_tex = nullptr; I->grad_tex_cap = 0;
    I->tq = TexelQueues{ nullptr, nullptr, nullptr, nullptr, 0u, 0u, nullptr }; I->tq_scene = 0; I->tq_lanes = 0;
    I->pass_rng = nullptr; I->pass_rng_cap = 0; I->pass_jitter = nullptr; I->pass_jitter_cap = 0;
    I->grad_slots = nullptr; I->grad_slots_cap = 0; I->mq_idx = nullptr; I->mq_count = nullptr;
    for (int k = 0; k < 2; ++k) {
        if (ws_alloc(I, &I->st[k].a0, lanes) || ws_alloc(I, &I->st[k].a1, lanes) || ws_alloc(I, &I->st[k].a2, lanes) ||
            ws_alloc(I, &I->st[k].a3, lanes) || ws_alloc(I, &I->st[k].a4, lanes)) return 1;
    }
    /* closest-hit records: one 32-byte record per lane, viewed as h0 (float4, stride 2) and h1 (uint2, stride 4) -- see HIT0 / HIT1 in har_kernels.hip */
    if (ws_alloc(I, &I->h0, (size_t) 2 * lanes)) return 1;
    I->h1 = HAR_HIT_INTERLEAVED ? reinterpret_cast<uint2 *>(I->h0 + 1) : reinterpret_cast<uint2 *>(I->h0 + lanes); I->hit_scratch = nullptr;
    if (ws_alloc(I, &I->items.s0, lanes) || ws_alloc(I, &I->items.s1, lanes) || ws_alloc(I, &I->items.s2, lanes)) return 1;
    I->items.s3 = I->items.s4 = nullptr; I->dL = nullptr; I->items2 = ItemArrays{}; I->result2 = nullptr;
    if (adjoint && (ws_alloc(I, &I->items.s3, lanes) || ws_alloc(I, &I->items.s4, lanes) || ws_alloc(I, &I->dL, lanes))) return 1;
    I->geo = ShapeArrays{}; I->d_pos_offset = nullptr; I->grad_pos = nullptr; I->d_inst_slot = nullptr; I->grad_inst = nullptr; I->grad_nrm = nullptr; I->nrm_acc = nullptr;
    if (adjoint && I->shape_on) {
        if (ws_alloc(I, &I->geo.g0, lanes) || ws_alloc(I, &I->geo.g1, lanes) || ws_alloc(I, &I->geo.g2, lanes) || ws_alloc(I, &I->geo.g3, lanes) || ws_alloc(I, &I->geo.g4, lanes) ||
            ws_alloc(I, &I->geo.g5, lanes) || ws_alloc(I, &I->geo.g6, lanes) || ws_alloc(I, &I->geo.pv0, lanes) || ws_alloc(I, &I->geo.pv1, lanes) || ws_alloc(I, &I->geo.vis, lanes)) return 1;
        if (I->pos_verts) {
            if (ws_alloc(I, &I->d_pos_offset, I->pos_offset.size()) || ws_alloc(I, &I->grad_pos, (size_t) 3 * I->pos_verts)) return 1;
            if (I->pos_smooth && (ws_alloc(I, &I->grad_nrm, (size_t) 3 * I->pos_verts) || ws_alloc(I, &I->nrm_acc, (size_t) 3 * I->pos_verts))) return 1;
            HIP_TRY(hipMemcpy(I->d_pos_offset, I->pos_offset.data(), I->pos_offset.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        }
        if (I->inst_count) {
            std::vector<int32_t> slots(I->inst_count); for (uint32_t k = 0; k < I->inst_count; ++k) slots[k] = (int32_t) k;
            if (ws_alloc(I, &I->d_inst_slot, I->inst_count) || ws_alloc(I, &I->grad_inst, (size_t) 12 * I->inst_count)) return 1;
            HIP_TRY(hipMemcpy(I->d_inst_slot, slots.data(), slots.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        }
    }
    I->rc_h0 = nullptr; I->rc_h1 = nullptr; I->rc_vis = nullptr; I->cache_bounces = 0;
    I->ws_tape = 0; I->tape_bounces = 0; I->tape_h0 = nullptr; I->tape_vis = nullptr; I->tape_next = nullptr;
    for (int k = 0; k < 2; ++k) { I->tape_la[k] = nullptr; I->tape_lb[k] = nullptr; }
    for (int k = 0; k < 4; ++k) I->tape_rec[k] = nullptr;
    for (auto &w : I->tape_st) w = WaveState{};
    if (adjoint && tape == 2) {
        /* record tape (TapeArrays): 3 x 16 B of adjoint record + 16 B of emission + 1 B visibility + 4 B next slot per lane and bounce, 2 x 24 B for L / dL:
         * 37 GB for a 2^26-lane chunk at max_depth = 8 */
        const uint32_t nb = bounce_limit(I);
        for (int k = 0; k < 4; ++k) if (ws_alloc(I, &I->tape_rec[k], (size_t) lanes * nb)) return 1;
        if (ws_alloc(I, &I->tape_vis, (size_t) lanes * nb) || ws_alloc(I, &I->tape_next, (size_t) lanes * nb)) return 1;
        for (int k = 0; k < 2; ++k) if (ws_alloc(I, &I->tape_la[k], lanes) || ws_alloc(I, &I->tape_lb[k], lanes)) return 1;
        I->ws_tape = 2; I->tape_bounces = nb;
    } else
    if (adjoint && tape == 1) {
        /* the tape instead of the lane-indexed cache: (nb + 1) x 72 B of path state + nb x (32 B hit + 1 B visibility + 4 B next slot) per lane, 2 x 24 B for
         * L / dL: 62 GB for a 2^26-lane chunk at max_depth = 8 -- what 288 GB of HBM are for (the adjoint shading pass moves 40 % fewer bytes) */
        const uint32_t nb = bounce_limit(I);
        for (uint32_t b = 0; b <= nb; ++b)
            if (ws_alloc(I, &I->tape_st[b].a0, lanes) || ws_alloc(I, &I->tape_st[b].a1, lanes) || ws_alloc(I, &I->tape_st[b].a2, lanes) ||
                ws_alloc(I, &I->tape_st[b].a3, lanes) || ws_alloc(I, &I->tape_st[b].a4, lanes)) return 1;
        if (ws_alloc(I, &I->tape_h0, (size_t) 2 * lanes * nb) || ws_alloc(I, &I->tape_vis, (size_t) lanes * nb) || ws_alloc(I, &I->tape_next, (size_t) lanes * nb)) return 1;
        for (int k = 0; k < 2; ++k) if (ws_alloc(I, &I->tape_la[k], lanes) || ws_alloc(I, &I->tape_lb[k], lanes)) return 1;
        I->ws_tape = 1; I->tape_bounces = nb;
    } else
    if (adjoint && I->use_cache) {
        /* 25 B per lane and cached bounce; bounces beyond the cache are simply traced again */
        const uint32_t nb = std::min<uint32_t>(bounce_limit(I), HAR_REPLAY_CACHE_BOUNCES);
        if (nb && (ws_alloc(I, &I->rc_h0, (size_t) lanes * nb) || ws_alloc(I, &I->rc_h1, (size_t) lanes * nb) || ws_alloc(I, &I->rc_vis, (size_t) lanes * nb))) return 1;
        I->cache_bounces = nb;
    }
    if (ws_alloc(I, &I->result, lanes)) return 1;
    if (ws_alloc(I, &I->stack_spill, (size_t) HAR_STACK_SPILL * HAR_MAX_TRAVERSAL_BLOCKS * 256)) return 1;
    if (ws_alloc(I, &I->skip_counters, (size_t) 4 * HAR_SHARDS * HAR_COUNTER_STRIDE)) return 1;
    if (ws_alloc(I, &I->pk_list, (size_t) lanes / 64 + HAR_SHARDS) || ws_alloc(I, &I->pk_counters, (size_t) 2 * HAR_SHARDS * HAR_COUNTER_STRIDE)) return 1;
    I->alpha_lane = nullptr;
    if (I->alpha_film && ws_alloc(I, &I->alpha_lane, lanes)) return 1;
    if (ws_alloc(I, &I->counters, (size_t) 4 * HAR_MAX_BOUNCE_SLOTS * HAR_SHARDS * HAR_COUNTER_STRIDE) || ws_alloc(I, &I->totals, 4) || ws_alloc(I, &I->status, 1)) return 1;
    /* totals / status are cleared by every render call ON ITS STREAM before use.  (Round 1 also cleared them here with hipMemset: that memset is
     * enqueued on the NULL stream and runs after whatever is queued there -- in two-stream mode after the first half of the frame -- while the twin
     * renders on its non-blocking stream; when the twin finished first, the late memset wiped its counters: half the paths in har_render_stats,
     * seen as an intermittent test failure when scenes of different cost alternate.) */
    I->ws_lanes = lanes; I->ws_adjoint = adjoint; I->shard_cap = lanes / HAR_SHARDS;
    return 0;
}

/* HAR_DEBUG_SYNC=1 (debug switch): name every launch class on stderr and wait for it, so that a GPU fault is attributed to a kernel */
void dbg_sync(hipStream_t s, int cls) {
    static const bool on = getenv("HAR_DEBUG_SYNC") != nullptr;
    if (!on) return;
    static const char *names[8] = { "raygen", "trace_closest", "shade", "resolve", "splat", "?", "other", "start" };
    fprintf(stderr, "[hip_ad_rgb] sync after %s ...", names[cls & 7]); fflush(stderr);
    hipError_t e = hipStreamSynchronize(s);
    fprintf(stderr, " %s\n", hipGetErrorString(e)); fflush(stderr);
}

#define HAR_PROFILE_RING 32       /* event sets (frames in flight) before prof_begin has to wait for the oldest */
/* fold a finished event set into the accumulators (waits for its last event) */
int prof_collect(HarIntegratorImpl *I, HarIntegratorImpl::EventSet &E) {
    if (E.used >= 2) {
        HIP_TRY(hipEventSynchronize(E.ev[E.used - 1]));
        for (size_t k = 1; k < E.used; ++k) {
            float dt = 0.f;
            HIP_TRY(hipEventElapsedTime(&dt, E.ev[k - 1], E.ev[k]));
            int c = E.cls[k]; if (c < 0 || c > 6) c = CLS_OTHER;
            I->acc_ms[c] += dt; I->acc_launches[c]++; I->acc_ms[5] += dt;
        }
        I->acc_frames++;
    }
    E.used = 0;
    return 0;
}
/* start of a frame: take the next event set of the ring */
int prof_begin(HarIntegratorImpl *I, hipStream_t s) {
    if (!I->profiling) return 0;
    if (I->sets.empty()) { I->sets.resize(1); I->cur_set = 0; }
    else {
        const size_t next = (I->cur_set + 1) % HAR_PROFILE_RING;
        if (next >= I->sets.size()) I->sets.resize(next + 1);
        I->cur_set = next;
    }
    if (prof_collect(I, I->sets[I->cur_set])) return 1;
    prof_mark(I, s, CLS_START);
    return 0;
}
void prof_mark(HarIntegratorImpl *I, hipStream_t s, int cls) {
    dbg_sync(s, cls);
    if (!I->profiling || I->sets.empty()) return;
    HarIntegratorImpl::EventSet &E = I->sets[I->cur_set];
    if (E.used == E.ev.size()) {
        hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return;
        E.ev.push_back(e); E.cls.push_back(cls);
    }
    E.cls[E.used] = cls;
    (void) hipEventRecord(E.ev[E.used++], s);
}
void prof_destroy(HarIntegratorImpl *I) {
    for (auto &E : I->sets) for (hipEvent_t e : E.ev) (void) hipEventDestroy(e);
    I->sets.clear();
}

uint32_t log2_exact(uint32_t v) { for (uint32_t k = 0; k < 32; ++k) if ((1u << k) == v) return k; return 0xffffffffu; }



static inline uint32_t *cnt_alive(HarIntegratorImpl *I, uint32_t b) { return I->counters + (size_t) b * HAR_SHARDS * HAR_COUNTER_STRIDE; }
static inline uint32_t *cnt_items(HarIntegratorImpl *I, uint32_t b) { return I->counters + (size_t) (HAR_MAX_BOUNCE_SLOTS + b) * HAR_SHARDS * HAR_COUNTER_STRIDE; }
/* work cursors of the persistent traversal kernels (one per bounce and shard) */
static inline uint32_t *cur_trace(HarIntegratorImpl *I, uint32_t b) { return I->counters + (size_t) (2 * HAR_MAX_BOUNCE_SLOTS + b) * HAR_SHARDS * HAR_COUNTER_STRIDE; }
static inline uint32_t *cur_resolve(HarIntegratorImpl *I, uint32_t b) { return I->counters + (size_t) (3 * HAR_MAX_BOUNCE_SLOTS + b) * HAR_SHARDS * HAR_COUNTER_STRIDE; }

/* texel-gradient queues for the bitmap textures of scene S (see TexelQueues): row bands whose LDS copy fits HAR_TQ_LDS_BYTES, at most HAR_TQ_MAX of
 * them; textures that do not fit keep the direct atomics.  HAR_TEXEL_QUEUES=0 switches the queues off (A/B). */
int ensure_texel_queues(HarSceneImpl *S, HarIntegratorImpl *I) {
    static const bool enabled = !(getenv("HAR_TEXEL_QUEUES") && atoi(getenv("HAR_TEXEL_QUEUES")) == 0);
    if (I->tq_scene == S->serial && I->tq_lanes == I->ws_lanes) return 0;
    /* queues of another scene (params.update() re-creates the scene handle; one integrator may alternate between scenes): give their buffers back
     * first -- the record buffer alone is 64 B per workspace lane */
    {
        void *old[4] = { I->tq.rec, I->tq.count, const_cast<uint2 *>(I->tq.band), const_cast<uint4 *>(I->tq.qinfo) };
        for (void *q : old) {
            if (!q) continue;
            auto it = std::find(I->owned.begin(), I->owned.end(), q);
            if (it != I->owned.end()) { I->owned.erase(it); dev_free(q); }
        }
    }
    I->tq = TexelQueues{ nullptr, nullptr, nullptr, nullptr, 0u, 0u, nullptr }; I->tq_scene = S->serial; I->tq_lanes = I->ws_lanes;
    const size_t nt = S->hs.textures.size();
    if (!enabled || nt == 0) return 0;
    /* LDS copy of a band: the smallest of 24 / 32 / 48 / 64 KB that keeps the texture within HAR_TQ_MAX queues (three 64-bit accumulators per texel: a 256-wide
     * texture gets 4-row bands in 32 KB, five blocks per CU).  Round 2 measured the float version at 24 / 48 / 64 KB: 125.9 / 130.4 / 127.7 ms per PRB step --
     * what matters is that every CU holds several blocks.  HAR_TQ_LDS forces one size (A/B). */
    static const size_t lds_forced = getenv("HAR_TQ_LDS") ? (size_t) atol(getenv("HAR_TQ_LDS")) : 0;
    std::vector<uint2> band(nt); std::vector<uint4> qinfo, heights; uint32_t nq = 0; size_t lds_used = 0;
    for (size_t t = 0; t < nt; ++t) {
        const uint32_t W = S->hs.textures[t].w, H = S->hs.textures[t].h;
        band[t] = make_uint2(0xffffffffu, 1u);
        if (W == 0 || H == 0 || W > 65535u || H > 65535u) continue;
        if (S->hs.textures[t].mode != 0u) continue;          /* the queue records assume the bilinear + repeat neighbourhood (x0 + 1, y0 + 1 wrapped): other modes keep the direct atomics */
        const size_t sizes[4] = { (size_t) HAR_TQ_LDS_BYTES, 32768, 49152, 65536 };
        for (int k = 0; k < 4; ++k) {
            const size_t lds = lds_forced ? lds_forced : sizes[k];
            const size_t row_bytes = (size_t) W * 3 * HAR_TQ_ACC_BYTES;                /* three 64-bit fixed-point accumulators per texel */
            if (row_bytes * 2 > lds) { if (lds_forced) break; continue; }
            /* the LDS copy of a band holds its rows + the row after it (k_texel_accumulate) */
            const uint32_t rows = std::min<uint32_t>(H, (uint32_t) (lds / row_bytes) - 1u), nb = (H + rows - 1) / rows;
            if (nq + nb > HAR_TQ_MAX) { if (lds_forced) break; continue; }
            band[t] = make_uint2(nq, rows);
            for (uint32_t b = 0; b < nb; ++b) { qinfo.push_back(make_uint4((uint32_t) t, b * rows, std::min(rows, H - b * rows), W)); heights.push_back(make_uint4(H, 0u, 0u, 0u)); }
            nq += nb; lds_used = std::max(lds_used, lds);
            break;
        }
    }
    if (nq == 0) return 0;
    qinfo.insert(qinfo.end(), heights.begin(), heights.end());
    uint2 *d_band = nullptr; uint4 *d_qinfo = nullptr; float4 *rec = nullptr; uint32_t *count = nullptr;
    /* every (shard, band) queue holds twice its mean share of a shard's lanes: 2 x lanes records of 32 bytes in total */
    const uint32_t cap = std::max<uint32_t>(1024u, (uint32_t) (2ull * I->shard_cap / nq));
    if (ws_alloc(I, &d_band, nt) || ws_alloc(I, &d_qinfo, qinfo.size()) || ws_alloc(I, &rec, (size_t) 2 * HAR_SHARDS * nq * cap) ||
        ws_alloc(I, &count, (size_t) (HAR_SHARDS * nq + 1) * HAR_COUNTER_STRIDE)) return 1;          /* + the launch's gmax word (cleared with the counters) */
    HIP_TRY(hipMemcpy(d_band, band.data(), nt * sizeof(uint2), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_qinfo, qinfo.data(), qinfo.size() * sizeof(uint4), hipMemcpyHostToDevice));
    I->tq = TexelQueues{ rec, count, d_band, d_qinfo, nq, cap, count + (size_t) HAR_SHARDS * nq * HAR_COUNTER_STRIDE }; I->tq_lds = (uint32_t) lds_used;
    return 0;
}

/* shadow-ray overlap (HarIntegratorImpl::aux_stream) for a job of n lanes?  `path` and the primal passes of `prb`, at most HAR_OVERLAP_MAX_LANES lanes -- the share
 * of a rank when several GPUs split a frame, where a launch is short and its tail (the chip waiting for the launch's longest rays) is a sizeable part of it:
 * measured on the middle bands of the headline frame (tools/band_bench.py, profiles/r03_ab_shadow_overlap.txt), one stream without / with overlap | two streams
 * without / with: 2 M lanes 5.46 / 4.67 | 5.11 / 4.60 ms, 4 M 8.56 / 7.64 | 8.09 / 7.68, 8 M 13.89 / 13.03 | 13.47 / 13.16, 16 M 24.58 / 23.76 | 24.42 / 24.36,
 * 33 M 42.50 / 41.90, 67 M 77.07 / 77.30; with the asynchronous join (second item set + `result2`) and 64-ray fetches 2 M 4.11, 8 M 12.60, 16 M 23.66, 33 M 42.06, 67 M still
 * neutral (78.05 / 78.37).  A single large wavefront keeps one stream and sequential launches (its kernels are timed one by one for the bench
 * line).  Not with the HBM stack spill (both traversal kernels would share it) nor with hide_emitters.  HAR_OVERLAP = 0 / 1 forces it off / on (A/B). */
static bool overlap_applies(const HarSceneImpl *S, const HarIntegratorImpl *I, uint64_t n) {
    static const int overlap_env = getenv("HAR_OVERLAP") ? atoi(getenv("HAR_OVERLAP")) : -1;
    static const bool force_spill = getenv("HAR_FORCE_STACK_SPILL") != nullptr;
    if (force_spill || S->hs.stack_need() + HAR_STACK_MARGIN > HAR_LDS_STACK_SMALL || I->hide_emitters) return false;
    return overlap_env < 0 ? n <= HAR_OVERLAP_MAX_LANES : overlap_env != 0;
}

/* rays of har_integrator_sample: SoA arrays of n_total rays, the chunk covers [first, first + n) */
struct RaySource { const float *o, *d, *maxt; const uint64_t *state; const uint8_t *active; uint32_t n_total, first; };

/* one chunk: raygen + bounce loop.  `mode` selects path / prb primal / prb adjoint kernels; `rays` != nullptr: the wavefront starts from
 * caller-supplied rays (SamplingIntegrator::sample) instead of the sensor; `valid_lane` != nullptr receives the samples' masks */
int run_chunk(HarSceneImpl *S, HarIntegratorImpl *I, const DSensor &C, int mode, uint32_t seed, uint32_t spp, uint32_t log_spp,
              uint32_t lane_base, uint32_t n, float *grad_refl, hipStream_t s, int cache_mode = 0, const PassState &ps = PassState{ nullptr, nullptr, 0 },
              const RaySource *rays = nullptr, float *valid_lane = nullptr) {
    const uint32_t nb = bounce_limit(I);
    const size_t used = (size_t) std::min<uint32_t>(nb + 2, HAR_MAX_BOUNCE_SLOTS) * HAR_SHARDS * HAR_COUNTER_STRIDE * sizeof(uint32_t);
    /* replay tape (cache_mode 3: the primal pass records, 4: the adjoint pass replays; TapeArrays in har_kernels.h) */
    const bool tape_w = cache_mode == 3, tape_r = cache_mode == 4, tape = tape_w || tape_r;
    /* record tape (cache_mode 5: the primal pass shades with the adjoint flavour and writes one record per vertex, 6: the adjoint pass is k_commit) */
    const bool rec_w = cache_mode == 5, rec_r = cache_mode == 6;
    if (tape && (I->ws_tape != 1 || nb > I->tape_bounces || rays)) return fail("internal: tape mode without a tape workspace");
    if ((rec_w || rec_r) && (I->ws_tape != 2 || nb > I->tape_bounces || rays)) return fail("internal: record-tape mode without its workspace");
    if (rec_r) {
        /* the whole adjoint pass: L / dL into bounce 0's slot order, then one streaming commit per bounce (+ the texel queues' accumulation) */

        const uint32_t cgrid = std::max<uint32_t>(HAR_SHARDS, std::min<uint32_t>(((n + 255) / 256 + HAR_SHARDS - 1) / HAR_SHARDS * HAR_SHARDS, 4096u));
        for (uint32_t b = 0; b < nb; ++b) {
            const size_t off = (size_t) b * I->ws_lanes;
            const TapeArrays tp{ I->tape_next + off, I->tape_la[b & 1], I->tape_lb[b & 1], I->tape_la[(b & 1) ^ 1], I->tape_lb[(b & 1) ^ 1],
                                 I->tape_rec[0] + off, I->tape_rec[1] + off, I->tape_rec[2] + off, I->tape_rec[3] + off };
            const bool queued = I->tq.nq != 0;
            if (queued) HIP_TRY(hipMemsetAsync(I->tq.count, 0, (size_t) (HAR_SHARDS * I->tq.nq + 1) * HAR_COUNTER_STRIDE * sizeof(uint32_t), s));
            launch_commit(s, cgrid, S->ds, I->shard_cap, cnt_alive(I, b), tp, I->tape_vis + off, grad_refl, I->d_grad_tex, queued ? &I->tq : nullptr,
                          b == 0 ? I->result : nullptr, b == 0 ? I->dL : nullptr);
            prof_mark(I, s, CLS_SHADE);
            if (queued) {
                static const uint32_t bpq_env = getenv("HAR_TQ_BPQ") ? (uint32_t) atoi(getenv("HAR_TQ_BPQ")) : 0u;
                launch_texel_accumulate(s, I->tq, I->d_grad_tex, bpq_env ? bpq_env : (n > (1u << 22) ? 4u : 1u), I->tq_lds); prof_mark(I, s, CLS_OTHER);
            }
        }
        launch_accumulate_stats(s, I->counters, nb, I->totals, n);
        prof_mark(I, s, CLS_OTHER);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (!tape_r) HIP_TRY(hipMemsetAsync(cnt_alive(I, 0), 0, used, s));      /* the replay reads the primal pass's wavefront sizes */
    HIP_TRY(hipMemsetAsync(cnt_items(I, 0), 0, used, s));
    HIP_TRY(hipMemsetAsync(cur_trace(I, 0), 0, used, s));
    HIP_TRY(hipMemsetAsync(cur_resolve(I, 0), 0, used, s));
    /* FIRST VERTEX: the state of a path at bounce 0 is a function of its lane index -- the ray generation kernel stores only the rays (32 of 72 B per lane) and the first
     * shading launch rebuilds the state instead of reading it (k_raygen<.., LITE>, k_shade<.., FIRST>; ShadeParams::sensor).  Plain forward renders and the recording pass of prb, of one pass, whose
     * bounce-0 wavefront nobody else reads (no alpha / validity flags, no material queues, no tape); the passes of a multi-pass forward render resume their samplers from the pass state in both kernels.  HAR_FIRST_VERTEX=0 switches it off (A/B) */
    static const bool first_env = !(getenv("HAR_FIRST_VERTEX") && atoi(getenv("HAR_FIRST_VERTEX")) == 0);
    static const int mq_env0 = getenv("HAR_MATERIAL_QUEUES") ? atoi(getenv("HAR_MATERIAL_QUEUES")) : -1;
    const bool first_regen = first_env && ((mode == MODE_PATH && cache_mode == 0) || (mode == MODE_PRB_PRIMAL && rec_w && I->adj && !I->forward_mode && !ps.rng)) && !rays && !valid_lane && !(I->alpha_film && I->alpha_lane) &&
                             !(mq_env0 < 0 ? I->material_queues : mq_env0 != 0);
    if (tape_r) launch_tape_begin(s, C, seed, spp, log_spp, lane_base, n, I->shard_cap, I->result, I->adj, I->tape_la[0], I->tape_lb[0]);
    else if (rays) launch_raygen_rays(s, seed, lane_base, n, rays->n_total, rays->first, rays->o, rays->d, rays->maxt, rays->state, rays->active, I->shard_cap, I->st[0], I->result, cnt_alive(I, 0));
    /* forward mode: k_raygen<ADJOINT> takes `adj == nullptr` as "zero dL" -- a workspace that served render_backward before still holds that call's adjoint
     * image in I->adj (possibly of a smaller film), which must not be gathered here */
    /* the adjoint image goes to the adjoint raygen (dL per lane; not in forward mode: dL accumulates there) and to the primal raygen of the record tape (dL for its emission terms) */
    else launch_raygen(mode, s, C, seed, spp, log_spp, lane_base, n, I->shard_cap, tape_w ? I->tape_st[0] : I->st[0], I->result, cnt_alive(I, 0),
                       (I->forward_mode || (mode == MODE_PRB_PRIMAL && !rec_w)) ? nullptr : I->adj, I->dL, ps, first_regen);
    prof_mark(I, s, CLS_RAYGEN);
    const bool fwd = mode == MODE_PRB_ADJOINT && I->forward_mode;
    ShadeParams P{ seed, I->max_depth, I->rr_depth, ((mode != MODE_PATH && I->grad_emitters) ? HAR_SHADE_EMITTER_GRADS : 0u)      /* (the primal pass of a backward step too: it traces the shadow rays of samples that only carry a radiance gradient, shade_lane) */ | (I->hide_emitters ? HAR_SHADE_HIDE_EMITTERS : 0u) |
                   (fwd ? HAR_SHADE_FORWARD_MODE : 0u) | ((mode == MODE_PRB_ADJOINT && I->grad_bsdf_params && !fwd) ? HAR_SHADE_EXTRA_GRADS : 0u) |
                   /* both passes of a backward step: the primal pass traces the shadow rays whose visibility the adjoint pass reads (shade_lane: lt_item) */
                   ((mode != MODE_PATH && I->grad_light_texels && !I->forward_mode && (S->ds.bsdf_types & HAR_SCENE_TEXLIGHT)) ? HAR_SHADE_LIGHT_TEXELS : 0u) };
    /* generic shading kernels: material-sort window in tiles of 256 paths (k_shade; HAR_SORT_WINDOW=1 is the round-3 kernel, A/B) */
    static const uint32_t sort_window_env = getenv("HAR_SORT_WINDOW") ? (uint32_t) std::max(1, atoi(getenv("HAR_SORT_WINDOW"))) : 8u;
    P.sort_window = sort_window_env;
    ShadeParams P0 = P;           /* bounce 0 with first_regen */
    if (first_regen) { P0.flags |= HAR_SHADE_FIRST_VERTEX; P0.spp = spp; P0.log_spp = log_spp; P0.sensor = C; P0.resume = (ps.rng && ps.pass) ? 1u : 0u; }
    /* grid: a multiple of 8 so that block b serves shard b % 8; enough blocks to cover the chunk once */
    const uint32_t grid = std::max<uint32_t>(HAR_SHARDS, std::min<uint32_t>(((n + 255) / 256 + HAR_SHARDS - 1) / HAR_SHARDS * HAR_SHARDS, 4096u));
    /* persistent traversal kernels: enough blocks to fill the chip (<= 8 blocks/CU), never more than the work */
    static const uint32_t tgrid_env = getenv("HAR_TRACE_GRID") ? (uint32_t) std::max(HAR_SHARDS, atoi(getenv("HAR_TRACE_GRID")) / HAR_SHARDS * HAR_SHARDS) : 0u;      /* A/B: blocks of a persistent launch */
    const uint32_t tgrid = std::min<uint32_t>(grid, tgrid_env ? std::min<uint32_t>(tgrid_env, (uint32_t) HAR_MAX_TRAVERSAL_BLOCKS) : (uint32_t) HAR_MAX_TRAVERSAL_BLOCKS);
    /* scenes whose depth-first stack bound fits the LDS entries run the kernels without the HBM spill path */
    static const bool force_spill = getenv("HAR_FORCE_STACK_SPILL") != nullptr;
    uint2 *spill = (force_spill || S->hs.stack_need() + HAR_STACK_MARGIN > HAR_LDS_STACK_SMALL) ? I->stack_spill : nullptr;
    const bool shape = mode == MODE_PRB_ADJOINT && I->shape_on;
    /* adjoint replay of a cached bounce: `shade` commits the vertex adjoint itself (the shadow-ray result is in the cache), no items, no resolve launch */
    static const bool inline_env = getenv("HAR_ADJOINT_INLINE") ? atoi(getenv("HAR_ADJOINT_INLINE")) != 0 : true;
    const bool inline_commit = inline_env && mode == MODE_PRB_ADJOINT && !shape && !I->forward_mode;      /* forward mode commits in the resolve kernels (own instantiation) */
    const ShapeTargets targets{ I->d_pos_offset, I->grad_pos, I->pos_verts, I->d_inst_slot, I->grad_inst, I->inst_count, I->grad_nrm };
    /* per-material shading queues: scenes with more than one BSDF model, `path` and the primal pass of `prb` (the adjoint kernels keep the generic code:
     * their in-place commit is bound by memory traffic, not by the model code).  HAR_MATERIAL_QUEUES=0: the generic kernel with its block-local sort (A/B) */
    static const int mq_env = getenv("HAR_MATERIAL_QUEUES") ? atoi(getenv("HAR_MATERIAL_QUEUES")) : -1;      /* -1: the integrator's setting; 0 / 1 force (A/B) */
    const bool use_mq = (mq_env < 0 ? I->material_queues : mq_env != 0) && mode != MODE_PRB_ADJOINT && cache_mode != 5 && __builtin_popcount(S->mat_classes) >= 2 && !(S->ds.bsdf_types & HAR_SCENE_ENVMAP);
    if (use_mq && !I->mq_idx && (ws_alloc(I, &I->mq_idx, (size_t) HAR_MAT_CLASSES * I->ws_lanes) || ws_alloc(I, &I->mq_count, (size_t) HAR_MAT_CLASSES * HAR_SHARDS * HAR_COUNTER_STRIDE))) return 1;
    const MaterialQueues mq{ I->mq_idx, I->mq_count, I->ws_lanes, S->mat_miss_class };
    bool overlap = mode != MODE_PRB_ADJOINT && !rays && overlap_applies(S, I, n);
    /* LATE OVERLAP (jobs too large for the full overlap above): from bounce `late_from` on, bounce b's shadow rays run on the second stream NEXT TO bounce b + 1's
     * closest-hit rays, and bounce b + 1's shading waits for them -- so both kernels keep writing the one `result` array (no second item set, no result2, no final add:
     * what made the full overlap neutral on a 67 M-lane frame).  The idea was that the launches past the Russian-roulette depth (0.4 - 1 ms each for a few per cent of the
     * frame's rays) would hide each other's tails; measured +0.7 % at best on the headline frame and -0.7 ... +0.4 % elsewhere (profiles/r05_ab_late_overlap.txt): two
     * persistent launches share the same issue slots, there was no idle tail to fill.  DEFAULT OFF; HAR_LATE_OVERLAP=<first bounce> switches it on (A/B) */
    static const int late_env = getenv("HAR_LATE_OVERLAP") ? atoi(getenv("HAR_LATE_OVERLAP")) : -2;
    const bool late_ok = !overlap && mode != MODE_PRB_ADJOINT && !rays && late_env != -1 && overlap_applies(S, I, 0);      /* n = 0: every condition of the full overlap but the job's size */
    const uint32_t late_from = late_env >= 0 ? (uint32_t) late_env : (uint32_t) HAR_LATE_OVERLAP_DEFAULT(I->rr_depth);
    bool late_on = late_ok && late_from < nb, late_pending = false;
    if ((overlap || late_on) && !I->aux_stream) {
        if (hipStreamCreateWithFlags(&I->aux_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&I->ev_shaded, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&I->ev_resolved, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&I->ev_resolved2, hipEventDisableTiming) != hipSuccess) { I->aux_stream = nullptr; overlap = false; late_on = false; }
    }
    if (overlap && !I->result2 && (ws_alloc(I, &I->items2.s0, I->ws_lanes) || ws_alloc(I, &I->items2.s1, I->ws_lanes) || ws_alloc(I, &I->items2.s2, I->ws_lanes) || ws_alloc(I, &I->result2, I->ws_lanes))) {
        I->result2 = nullptr; overlap = false; (void) hipGetLastError();      /* no room for the second item set: one stream (an optimisation, not a requirement) */
    }
    if (overlap) HIP_TRY(hipMemsetAsync(I->result2, 0, (size_t) n * sizeof(float4), s));
    hipEvent_t ev_res[2] = { I->ev_resolved, I->ev_resolved2 };
    bool resolve_pending[2] = { false, false };
    int cur = 0; uint32_t b = 0;
    for (; b < nb; ++b) {
        /* PRB replay cache: the primal pass of render_backward records this bounce's ray-query results per lane, the adjoint pass reads them */
        ReplayCache rc{ nullptr, nullptr, nullptr, 0 };
        if (tape || rec_w) rc = ReplayCache{ nullptr, nullptr, I->tape_vis + (size_t) b * I->ws_lanes, cache_mode };
        else if (cache_mode && b < I->cache_bounces)
            rc = ReplayCache{ I->rc_h0 + (size_t) b * I->ws_lanes, I->rc_h1 + (size_t) b * I->ws_lanes, I->rc_vis + (size_t) b * I->ws_lanes, cache_mode };
        /* the wavefront buffers of this bounce: the ping-pong pair, or the tape's per-bounce buffers (path state in / out, hit records) */
        const WaveState st_in = tape ? I->tape_st[b] : I->st[cur], st_out = tape ? I->tape_st[b + 1] : I->st[cur ^ 1];
        float4 *const h0 = tape ? I->tape_h0 + (size_t) 2 * I->ws_lanes * b : I->h0;
        uint2 *const h1 = tape ? (HAR_HIT_INTERLEAVED ? reinterpret_cast<uint2 *>(h0 + 1) : reinterpret_cast<uint2 *>(h0 + I->ws_lanes)) : I->h1;
        const size_t toff = (size_t) b * I->ws_lanes;
        const TapeArrays tp{ (tape || rec_w) ? I->tape_next + toff : nullptr, I->tape_la[b & 1], I->tape_lb[b & 1], I->tape_la[(b & 1) ^ 1], I->tape_lb[(b & 1) ^ 1],
                             rec_w ? I->tape_rec[0] + toff : nullptr, rec_w ? I->tape_rec[1] + toff : nullptr, rec_w ? I->tape_rec[2] + toff : nullptr, rec_w ? I->tape_rec[3] + toff : nullptr };
        if (rc.mode != 2 && rc.mode != 4) {
            /* camera rays at >= 64 samples per pixel: a wave is one pixel, its 64 rays walk the BVH together (k_trace_packet); what that kernel gives up on
             * -- incoherent packets -- goes to the per-lane kernel through a list.  Same hit records either way.  HAR_PACKET=0 / 1 forces it off / on (A/B). */
            static const int packet_env = getenv("HAR_PACKET") ? atoi(getenv("HAR_PACKET")) : -1;
            static const uint32_t packet_budget = getenv("HAR_PACKET_BUDGET") ? (uint32_t) atoi(getenv("HAR_PACKET_BUDGET")) : 160u;
            const int packet_mode = packet_env >= 0 ? packet_env : I->packet_tracing;
            const bool packet = b == 0 && !tape_r && (packet_mode < 0 ? (!rays && spp >= 64 && spp % 64 == 0 && lane_base % 64 == 0) : packet_mode != 0);
            if (packet) {
                const size_t cs = (size_t) HAR_SHARDS * HAR_COUNTER_STRIDE;
                const PacketList pl{ I->pk_list, I->pk_counters, I->pk_counters + cs, cnt_alive(I, b), I->shard_cap / 64u + 1u };
                HIP_TRY(hipMemsetAsync(I->pk_counters, 0, 2 * cs * sizeof(uint32_t), s));
                launch_trace_packet(s, tgrid, S->ds.accel, cnt_alive(I, b), cur_trace(I, b), I->shard_cap, st_in, h0, h1, pl, packet_budget);
                prof_mark(I, s, CLS_TRACE);             /* two launches of the closest-hit class: the packets, then the rays of the packets that gave up */
                launch_trace_closest(s, tgrid, spill, S->ds.accel, pl.count, pl.cursor, I->shard_cap, st_in, h0, h1, I->status, &pl);
            } else
            launch_trace_closest(s, tgrid, spill, S->ds.accel, cnt_alive(I, b), cur_trace(I, b), I->shard_cap, st_in, h0, h1, I->status);
            prof_mark(I, s, CLS_TRACE);
            if (I->stagger_record && b == 0) { HIP_TRY(hipEventRecord(I->ev_stagger, s)); I->stagger_record = false; }
        }
        if (I->hide_emitters && b == 0 && rc.mode != 2 && rc.mode != 4) {
            /* Integrator::skip_area_emitters (integrator.cpp:96-124) for the camera rays: continuation rays are gathered into a list, traced, and
             * their hits replace the lanes' hits until no lane sits on an area emitter any more.  Scratch: the other wavefront buffer holds the two
             * lists, the (still unused) item arrays the re-traced hits.  One host round trip per round -- `hide_emitters` is not a hot path. */
            const size_t cs = (size_t) HAR_SHARDS * HAR_COUNTER_STRIDE;
            uint32_t *cnt[2] = { I->skip_counters, I->skip_counters + 2 * cs }, *cursor[2] = { I->skip_counters + cs, I->skip_counters + 3 * cs };
            float4 *lo[2] = { st_out.a0, st_out.a2 }, *ld[2] = { st_out.a1, st_out.a3 };
            if (!I->hit_scratch && ws_alloc(I, &I->hit_scratch, (size_t) 2 * I->ws_lanes)) return 1;       /* re-traced hits: records in the layout of h0 / h1 */
            float4 *sh0 = I->hit_scratch; uint2 *sh1 = HAR_HIT_INTERLEAVED ? reinterpret_cast<uint2 *>(I->hit_scratch + 1) : reinterpret_cast<uint2 *>(I->hit_scratch + I->ws_lanes);
            HIP_TRY(hipMemsetAsync(I->skip_counters, 0, 4 * cs * sizeof(uint32_t), s));
            launch_skip_emitters(s, grid, S->ds, 1, I->shard_cap, cnt_alive(I, 0), st_in.a0, st_in.a1, nullptr, nullptr, h0, h1, lo[0], ld[0], cnt[0]);
            for (int round = 0, a = 0; round < 256; ++round, a ^= 1) {
                uint32_t host[HAR_SHARDS * HAR_COUNTER_STRIDE];
                HIP_TRY(hipMemcpyAsync(host, cnt[a], sizeof(host), hipMemcpyDeviceToHost, s));
                HIP_TRY(hipStreamSynchronize(s));
                uint32_t total = 0; for (int k = 0; k < HAR_SHARDS; ++k) total += host[k * HAR_COUNTER_STRIDE];
                if (total == 0) break;
                WaveState list{ lo[a], ld[a], nullptr, nullptr, nullptr };
                launch_trace_closest(s, tgrid, spill, S->ds.accel, cnt[a], cursor[a], I->shard_cap, list, sh0, sh1, I->status);
                HIP_TRY(hipMemsetAsync(cnt[a ^ 1], 0, cs * sizeof(uint32_t), s));
                HIP_TRY(hipMemsetAsync(cursor[a ^ 1], 0, cs * sizeof(uint32_t), s));
                launch_skip_emitters(s, grid, S->ds, 0, I->shard_cap, cnt[a], lo[a], ld[a], sh0, sh1, h0, h1, lo[a ^ 1], ld[a ^ 1], cnt[a ^ 1]);
            }
            prof_mark(I, s, CLS_OTHER);
        }
        if (((I->alpha_film && I->alpha_lane) || valid_lane) && b == 0 && mode != MODE_PRB_ADJOINT) {        /* `rgba` films: is the camera sample valid?  (path.cpp:114-115,307-308; prb.py:332) */
            const float miss = (mode == MODE_PATH && S->ds.env_emitter >= 0 && !I->hide_emitters) ? 1.f : 0.f;
            launch_alpha_flags(s, grid, I->shard_cap, cnt_alive(I, 0), st_in, h0, lane_base, miss, valid_lane ? valid_lane : I->alpha_lane);
        }
        /* vertex-position gradients of the PREVIOUS bounce's vertices: its items are still in place, `result` holds its L, and this bounce's ray
         * queries give the (detached) next interaction of every continued path */
        if (shape && b > 0) {
            launch_shape_adjoint(s, grid, S->ds, cnt_items(I, b - 1), I->shard_cap, I->items, I->geo, I->result, I->dL, 1, st_in, h0, h1, rc, targets);
            prof_mark(I, s, CLS_OTHER);
        }
        /* bounce b - 2's shadow rays used the item set this bounce's shading is about to fill */
        if (resolve_pending[b & 1]) { HIP_TRY(hipStreamWaitEvent(s, ev_res[b & 1], 0)); resolve_pending[b & 1] = false; }
        /* late overlap: bounce b - 1's shadow rays ran next to this bounce's closest-hit rays; shading touches `result` and refills the item set they read */
        if (late_pending) { HIP_TRY(hipStreamWaitEvent(s, I->ev_resolved, 0)); late_pending = false; prof_mark(I, s, CLS_RESOLVE); }
        const ItemArrays &items_b = (overlap && (b & 1)) ? I->items2 : I->items;
        const bool cached = rc.mode == 2 || rc.mode == 4;                             /* adjoint replay of a cached / taped bounce */
        const bool queued = inline_commit && cached && I->tq.nq != 0;                 /* texel gradients of this bounce go through the band queues */
        if (queued) HIP_TRY(hipMemsetAsync(I->tq.count, 0, (size_t) (HAR_SHARDS * I->tq.nq + 1) * HAR_COUNTER_STRIDE * sizeof(uint32_t), s));
        if (use_mq) {
            /* classify the bounce's hits, then one specialised launch per material class of the scene; the launches append their survivors / items to the
             * same compacted queues (slot reservation is per block, so the order of the classes does not matter to any path) */
            HIP_TRY(hipMemsetAsync(I->mq_count, 0, (size_t) HAR_MAT_CLASSES * HAR_SHARDS * HAR_COUNTER_STRIDE * sizeof(uint32_t), s));
            launch_classify(s, grid, S->ds, I->shard_cap, cnt_alive(I, b), h0, h1, mq);
            for (uint32_t c = 0; c < HAR_MAT_CLASSES; ++c)
                if (S->mat_classes & (1u << c))
                    launch_shade(mode, s, grid, S->ds, P, lane_base, I->shard_cap, cnt_alive(I, b), st_in, h0, h1, st_out, cnt_alive(I, b + 1),
                                 items_b, cnt_items(I, b), I->result, rc, ps.rng, I->dL, grad_refl, nullptr, nullptr, nullptr, nullptr, &mq, c, tape ? &tp : nullptr);
        } else
        launch_shade(mode, s, grid, S->ds, (b == 0 && first_regen) ? P0 : P, lane_base, I->shard_cap, cnt_alive(I, b), st_in, h0, h1, st_out, cnt_alive(I, b + 1),
                     items_b, cnt_items(I, b), I->result, rc, ps.rng, I->dL, grad_refl, shape ? &I->geo : nullptr, inline_commit && cached ? I->d_grad_tex : nullptr,
                     queued ? &I->tq : nullptr, (inline_commit && (rc.mode == 2 || rc.mode == 4)) ? I->grad_bsdf_params : nullptr, nullptr, 0, (tape || rec_w) ? &tp : nullptr);
        prof_mark(I, s, CLS_SHADE);
        if (queued) {
            static const uint32_t bpq_env = getenv("HAR_TQ_BPQ") ? (uint32_t) atoi(getenv("HAR_TQ_BPQ")) : 0u;
            launch_texel_accumulate(s, I->tq, I->d_grad_tex, bpq_env ? bpq_env : (n > (1u << 22) ? 4u : 1u), I->tq_lds); prof_mark(I, s, CLS_OTHER);
        }
        if (overlap) {
            HIP_TRY(hipEventRecord(I->ev_shaded, s));
            HIP_TRY(hipStreamWaitEvent(I->aux_stream, I->ev_shaded, 0));
            launch_resolve(mode, I->aux_stream, tgrid, spill, S->ds, cnt_items(I, b), cur_resolve(I, b), I->shard_cap, items_b, I->result2, I->dL, grad_refl, I->d_grad_tex, I->status, rc, nullptr, 0);
            HIP_TRY(hipEventRecord(ev_res[b & 1], I->aux_stream));
            resolve_pending[b & 1] = true;
        } else if (late_on && b >= late_from && b + 1 < nb) {
            HIP_TRY(hipEventRecord(I->ev_shaded, s));
            HIP_TRY(hipStreamWaitEvent(I->aux_stream, I->ev_shaded, 0));
            launch_resolve(mode, I->aux_stream, tgrid, spill, S->ds, cnt_items(I, b), cur_resolve(I, b), I->shard_cap, I->items, I->result, I->dL, grad_refl, I->d_grad_tex, I->status, rc, nullptr, 0);
            HIP_TRY(hipEventRecord(I->ev_resolved, I->aux_stream));
            late_pending = true;
        } else if (!(inline_commit && cached)) {
            /* adjoint items of a cached bounce (the path vertex-position gradients take): texel gradients through the band queues, as in the in-place commit */
            const bool item_queued = mode == MODE_PRB_ADJOINT && rc.mode == 2 && !fwd && I->tq.nq != 0;
            if (item_queued) HIP_TRY(hipMemsetAsync(I->tq.count, 0, (size_t) (HAR_SHARDS * I->tq.nq + 1) * HAR_COUNTER_STRIDE * sizeof(uint32_t), s));
            launch_resolve(mode, s, rc.mode == 2 ? grid : tgrid, spill, S->ds, cnt_items(I, b), cur_resolve(I, b), I->shard_cap, I->items, I->result, I->dL, grad_refl, I->d_grad_tex, I->status, rc,
                           shape ? I->geo.vis : nullptr, fwd ? 1 : 0, item_queued ? &I->tq : nullptr);
            if (item_queued) {
                static const uint32_t bpq_env2 = getenv("HAR_TQ_BPQ") ? (uint32_t) atoi(getenv("HAR_TQ_BPQ")) : 0u;
                launch_texel_accumulate(s, I->tq, I->d_grad_tex, bpq_env2 ? bpq_env2 : (n > (1u << 22) ? 4u : 1u), I->tq_lds);
            }
        }
        prof_mark(I, s, CLS_RESOLVE);
        cur ^= 1;
        if (b >= 15 && (b & 7) == 7) {           /* deep paths are rare: poll so that max_depth = -1 terminates */
            uint32_t alive[HAR_SHARDS * HAR_COUNTER_STRIDE];
            HIP_TRY(hipMemcpyAsync(alive, cnt_alive(I, b + 1), sizeof(alive), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            uint32_t total = 0; for (int k = 0; k < HAR_SHARDS; ++k) total += alive[k * HAR_COUNTER_STRIDE];
            if (total == 0) { ++b; break; }
        }
    }
    for (int k = 0; k < 2; ++k) if (resolve_pending[k]) HIP_TRY(hipStreamWaitEvent(s, ev_res[k], 0));
    if (late_pending) { HIP_TRY(hipStreamWaitEvent(s, I->ev_resolved, 0)); late_pending = false; prof_mark(I, s, CLS_RESOLVE); }
    if (overlap) launch_add(s, reinterpret_cast<const float *>(I->result2), reinterpret_cast<float *>(I->result), 4u * n);      /* the shadow rays' share of the radiance */
    if (shape && b > 0) {            /* the last bounce: no path continues */
        launch_shape_adjoint(s, grid, S->ds, cnt_items(I, b - 1), I->shard_cap, I->items, I->geo, I->result, I->dL, 0, I->st[cur], I->h0, I->h1, ReplayCache{ nullptr, nullptr, nullptr, 0 }, targets);
        prof_mark(I, s, CLS_OTHER);
    }
    if (mode != MODE_PRB_PRIMAL) {
        launch_accumulate_stats(s, I->counters, std::min(b + 1, nb), I->totals, n);
        prof_mark(I, s, CLS_OTHER);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

/* pixels of the sample grid of render(): the crop window, plus the filter border with Film::sample_border (integrator.cpp:162-165) */
int sample_grid(const HarSensor *sensor, uint32_t &w, uint32_t &h) {
    DSensor C; std::string e;
    if (!sensor) return fail("null sensor");
    if (!lower_sensor(*sensor, C, e)) return fail(e);
    w = C.samp_w; h = C.samp_h;
    return 0;
}

/* SamplingIntegrator::render, integrator.cpp:173-183,276-294 + Sampler::set_samples_per_wavefront, sampler.cpp:88-96 */
int pass_layout(const HarIntegratorImpl *I, uint32_t crop_w, uint32_t crop_h, uint32_t spp, uint32_t &spp_per_pass, uint32_t &n_passes) {
    if (spp == 0) return fail("spp must be > 0");
    spp_per_pass = I->samples_per_pass == 0xffffffffu ? spp : std::min(I->samples_per_pass, spp);
    if (spp_per_pass == 0 || spp % spp_per_pass != 0) return fail("sample_count (" + std::to_string(spp) + ") must be a multiple of spp_per_pass (" + std::to_string(spp_per_pass) + ").");
    n_passes = spp / spp_per_pass;
    const uint64_t limit = 0xffffffffull;
    uint64_t wavefront = (uint64_t) crop_w * crop_h * spp_per_pass;
    if (wavefront > limit) {
        spp_per_pass /= (uint32_t) ((wavefront + limit - 1) / limit);
        if (spp_per_pass == 0) return fail("the film alone exceeds the wavefront size limit of 2^32 - 1 lanes");
        n_passes = spp / spp_per_pass;
        /* the reference then calls sampler->set_samples_per_wavefront(spp_per_pass), which throws unless it divides the sample count */
        if (spp % spp_per_pass != 0) return fail("sample_count should be a multiple of samples_per_wavefront! (" + std::to_string(spp) + " samples in passes of " + std::to_string(spp_per_pass) + "; set samples_per_pass)");
    }
    return 0;
}

int check_common(HarSceneImpl *S, HarIntegratorImpl *I, const HarSensor *sensor, uint32_t spp, uint64_t &lb, uint64_t &le, DSensor &C, uint32_t &log_spp) {
    if (!S || !I || !sensor) return fail("null scene / integrator / sensor");
    std::string e;
    if (!lower_sensor(*sensor, C, e)) return fail(e);
    if (C.rfilter != 0 && 2 * (uint32_t) ceilf(C.radius - .5f) + 1 > HAR_MAX_FILTER_TAPS) return fail("reconstruction filter radius too large (max 9 taps)");
    if (spp == 0) return fail("spp must be > 0");
    uint64_t total = (uint64_t) C.samp_w * C.samp_h * spp;
    /* 2^32 wavefront limit of JIT variants (integrator.cpp:276-294, common.py:358-363) */
    if (total > 0xffffffffull) return fail("the rendering task exceeds 2^32 - 1 Monte Carlo samples; render in several passes");
    if (lb == 0 && le == 0) le = total;
    if (lb > le || le > total) return fail("invalid lane range");
    log_spp = log2_exact(spp);
    return 0;
}

} // namespace

extern "C" {

const char *har_last_error(void) { return g_error.c_str(); }

const char *har_device_arch(void) {
    static thread_local std::string arch;
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return nullptr;
    arch = prop.gcnArchName;
    return arch.c_str();
}

int har_scene_create(const HarSceneDesc *desc, HarScene *out) {
    if (!desc || !out) return fail("null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("hip_ad_rgb requires a HIP device (no CPU fallback)");
    HarSceneImpl *S = new HarSceneImpl();
    std::string e;
    if (!lower_scene(*desc, S->hs, e)) { delete S; return fail(e); }
    HostScene &hs = S->hs; DScene &D = S->ds;
    hipError_t err = hipSuccess;
    auto up = [&](auto &vec, auto **dst) { if (err == hipSuccess) err = upload(vec, dst, S->owned); };
    /* the node array keeps room for the largest TLAS the scene's instances can need (a TLAS over n leaves has at most n nodes): an instance update rebuilds the
     * TLAS on the host and rewrites the tail of this array in place */
    S->nodes_cap = hs.nodes.size() + (hs.has_tlas ? hs.insts.size() + 2 : 0);
    {
        void *p = nullptr;
        if (err == hipSuccess) err = dev_alloc(&p, std::max<size_t>(S->nodes_cap, 1) * sizeof(Node8));
        if (err == hipSuccess) { S->owned.push_back(p); if (!hs.nodes.empty()) err = hipMemcpy(p, hs.nodes.data(), hs.nodes.size() * sizeof(Node8), hipMemcpyHostToDevice); }
        D.accel.nodes = (const Node8 *) p;
    }
    up(hs.tris, &D.accel.tris); up(hs.inst_recs, &D.accel.insts);
    up(hs.blas_tri_ranges, &D.blas_tri_ranges); up(hs.verts, &D.verts); up(hs.faces, &D.faces);
#if HAR_SHADING_TRIS
    up(hs.shade_tris, &D.shade_tris);
#endif
    /* material class of every BSDF record and mesh (MaterialQueues; the mesh's class rides in DMesh::pad1 so that k_classify needs ONE dependent load) */
    S->mat_classes = 0;
    {
        std::vector<uint32_t> cls(hs.bsdfs.size());
        for (size_t k = 0; k < hs.bsdfs.size(); ++k) {
            const DBsdf &b = hs.bsdfs[k];
            uint32_t c = std::min<uint32_t>(b.type, BSDF_TYPE_COUNT - 1u);
            if ((b.flags & BF_TWOSIDED) && b.back >= 0 && hs.bsdfs[(size_t) b.back].type != b.type) c = HAR_MAT_GENERIC;
            cls[k] = c;
        }
        for (DMesh &m : hs.meshes) { m.pad1 = m.bsdf < cls.size() ? cls[m.bsdf] : 0u; S->mat_classes |= 1u << m.pad1; }
    }
    S->mat_miss_class = 0; while (S->mat_miss_class < HAR_MAT_CLASSES && !(S->mat_classes & (1u << S->mat_miss_class))) ++S->mat_miss_class;
    up(hs.meshes, &D.meshes); up(hs.bsdfs, &D.bsdfs); up(hs.emitters, &D.emitters); up(hs.insts, &D.insts);
    {   /* Accel::mesh_info: what a retiring closest-hit ray copies into its record (HAR_HIT_MATINFO) */
        std::vector<MeshInfo> info(hs.meshes.size());
        for (size_t k = 0; k < hs.meshes.size(); ++k) {
            const DMesh &m = hs.meshes[k];
            if (m.bsdf > 0xfffffu) { for (void *p : S->owned) dev_free(p); delete S; return fail("more than 2^20 BSDF records"); }
            info[k] = MeshInfo{ m.foff, m.bsdf | ((m.flags & 3u) << 20) | ((m.emitter >= 0 ? 1u : 0u) << 22) | ((m.pad1 & 0xfu) << 24) };
        }
        up(info, &D.accel.mesh_info);
    }
    std::vector<DTexture> dt;
    for (auto &t : hs.textures) {
        const float *p = nullptr; up(t.data, &p);
        S->tex_dev.push_back(const_cast<float *>(p)); dt.push_back(hs.device_texture(dt.size(), p));
    }
    S->tex_host_stale.assign(hs.textures.size(), 0);
    up(dt, &D.textures); S->d_textures = const_cast<DTexture *>(D.textures);
    up(hs.bsdf_tables, &D.bsdf_tables);
    D.envmap = nullptr;
    if (hs.has_envmap) {
        const float *tex = nullptr, *warp = nullptr; up(hs.env_tex, &tex); up(hs.env_warp, &warp);
        hs.envmap.tex = tex; hs.envmap.warp = warp;
        std::vector<DEnvmap> one(1, hs.envmap); up(one, &D.envmap);
    }
    up(hs.emitter_cdf, &D.emitter_cdf);
    {   /* the emitter-selection table always gets room for 2 x emitter_count floats: har_scene_set_emitter_sampling_weights rewrites it in place */
        std::vector<float> table(std::max<size_t>(2 * hs.emitters.size(), 1), 0.f);
        std::copy(hs.emitter_distr.begin(), hs.emitter_distr.end(), table.begin());
        const float *distr = nullptr; up(table, &distr); S->d_emitter_distr = const_cast<float *>(distr); hs.bind_tables(D, distr);
    }
    if (err != hipSuccess) { for (void *p : S->owned) dev_free(p); delete S; return fail(std::string("scene upload: ") + hipGetErrorString(err)); }
    S->d_bsdfs = const_cast<DBsdf *>(D.bsdfs);
    D.accel.root = hs.root; D.accel.has_tlas = hs.has_tlas; D.accel.n_tris = (uint32_t) hs.tris.size(); D.accel.n_insts = (uint32_t) hs.inst_recs.size();
    D.accel.top_root = hs.top_root; D.accel.top_first = hs.top_first; D.accel.top_count = hs.top_count; D.accel.top_last = hs.top_last;
    D.n_emitters = (uint32_t) hs.emitters.size(); D.n_meshes = (uint32_t) hs.meshes.size();
    D.n_bsdfs = (uint32_t) hs.bsdfs.size(); D.n_insts = (uint32_t) hs.insts.size(); D.n_textures = (uint32_t) hs.textures.size();
    D.env_emitter = hs.env_emitter;
    D.bsdf_types = 0; for (const DBsdf &b : hs.bsdfs) D.bsdf_types |= (1u << b.type) | ((b.flags & BF_TWOSIDED) ? 0x80000000u : 0u);
    if (hs.has_envmap || hs.has_mesh_emitters || hs.has_point_emitters || !hs.emitter_distr.empty()) D.bsdf_types |= HAR_SCENE_ENVMAP;
    for (const DEmitter &e : hs.emitters) if (e.type == 7u) D.bsdf_types |= HAR_SCENE_TEXLIGHT;      /* (has_mesh_emitters is set with it: the generic emitter kernels + the texel-distribution code) */
    /* the depth-first bound of the BVH must fit the traversal stacks (LDS entries + HBM spill columns): a deeper scene is refused here instead of
     * rendering with rays that overflow (an overflowing ray is a miss + a status word that only har_render_stats reads) */
    const uint32_t stack_cap = (uint32_t) std::min(HAR_LDS_STACK_DEPTH, HAR_LDS_STACK_SMALL + HAR_STACK_SPILL);
    if (hs.stack_need() + HAR_STACK_MARGIN > stack_cap) {
        const std::string msg = "the scene's BVH needs " + std::to_string(hs.stack_need() + HAR_STACK_MARGIN) + " traversal stack entries per ray, the kernels hold " +
                                std::to_string(stack_cap) + " (HAR_LDS_STACK_DEPTH / HAR_LDS_STACK_SMALL + HAR_STACK_SPILL in har_kernels.h)";
        for (void *p : S->owned) dev_free(p);
        delete S;
        return fail(msg);
    }
    *out = S;
    return 0;
}

int har_scene_destroy(HarScene S) {
    if (!S) return 0;
    (void) hipDeviceSynchronize();     /* accel must outlive in-flight launches (scene_native.inl:44-57) */
    for (void *p : S->owned) dev_free(p);
    delete S;
    return 0;
}

/* device -> host refresh of the mirrors that har_scene_set_*_device left stale (the host setters below rewrite whole records from the mirror) */
/* The values were written by hipMemcpyAsync on the CALLER's stream (har_scene_set_*_device); the blocking copies below run on the null stream, which does not order
 * itself against a non-blocking stream (torch side streams): wait for the last push first, or the mirror picks up the pre-update value and writes it back. */
static void wait_for_device_pushes(HarSceneImpl *S) {
    if (S->last_push_valid) { (void) hipStreamSynchronize(S->last_push_stream); S->last_push_valid = false; }
}
static int sync_host_records(HarSceneImpl *S) {
    if (S->bsdf_host_stale || S->emitter_host_stale) wait_for_device_pushes(S);
    if (S->bsdf_host_stale) { HIP_TRY(hipMemcpy(S->hs.bsdfs.data(), S->d_bsdfs, S->hs.bsdfs.size() * sizeof(DBsdf), hipMemcpyDeviceToHost)); S->bsdf_host_stale = false; }
    if (S->emitter_host_stale) { HIP_TRY(hipMemcpy(S->hs.emitters.data(), S->ds.emitters, S->hs.emitters.size() * sizeof(DEmitter), hipMemcpyDeviceToHost)); S->emitter_host_stale = false; }
    return 0;
}
int har_scene_set_reflectance(HarScene S, uint32_t bsdf, const float rgb[3]) {
    if (!S || bsdf >= S->hs.bsdfs.size()) return fail("invalid bsdf index");
    if (sync_host_records(S)) return 1;
    DBsdf &b = S->hs.bsdfs[bsdf]; b.r = rgb[0]; b.g = rgb[1]; b.b = rgb[2];
    if (b.type == BSDF_ROUGHPLASTIC || b.type == BSDF_PLASTIC) update_roughplastic_sampling_weight(S->hs, bsdf);     /* RoughPlastic::parameters_changed */
    HIP_TRY(hipMemcpy(S->d_bsdfs + bsdf, &b, sizeof(DBsdf), hipMemcpyHostToDevice));
    return 0;
}
static int refresh_host_geometry(HarSceneImpl *S, hipStream_t s);
int har_scene_set_delta_emitter(HarScene S, uint32_t emitter, const HarEmitter *record) {
    if (!S || !record || emitter >= S->hs.emitters.size()) return fail("invalid emitter index");
    if (sync_host_records(S)) return 1;
    if (refresh_host_geometry(S, nullptr)) return 1;            /* a directional light's record follows the scene's bounding sphere: the host's vertices / transforms must be current */
    std::string e;
    if (!scene_set_delta_emitter_host(S->hs, emitter, *record, e)) return fail(e);
    HIP_TRY(hipMemcpy(const_cast<DEmitter *>(S->ds.emitters), S->hs.emitters.data(), S->hs.emitters.size() * sizeof(DEmitter), hipMemcpyHostToDevice));
    if (S->hs.emitters.size() == 1) { S->ds.emitter0 = S->hs.emitters[0]; S->ds.emitter0_valid = 1u; }
    return 0;
}
int har_scene_set_bsdf_params(HarScene S, uint32_t bsdf, const HarBSDF *params) {
    if (!S || !params || bsdf >= S->hs.bsdfs.size()) return fail("invalid bsdf index");
    if (sync_host_records(S)) return 1;
    std::string e;
    if (!scene_set_bsdf_params_host(S->hs, bsdf, *params, e)) return fail(e);
    const DBsdf &b = S->hs.bsdfs[bsdf];
    if (b.type == BSDF_ROUGHPLASTIC && b.table >= 0)
        HIP_TRY(hipMemcpy(const_cast<float *>(S->ds.bsdf_tables) + b.table, S->hs.bsdf_tables.data() + b.table, HAR_ROUGH_TRANSMITTANCE_RES * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(S->d_bsdfs + bsdf, &b, sizeof(DBsdf), hipMemcpyHostToDevice));
    return 0;
}
int har_scene_set_emitter_radiance(HarScene S, uint32_t emitter, const float rgb[3]) {
    if (!S || emitter >= S->hs.emitters.size()) return fail("invalid emitter index");
    if (sync_host_records(S)) return 1;
    DEmitter &e = S->hs.emitters[emitter];
    if (e.type == 2u) return fail("an environment map has no constant radiance");
    if (e.type == 7u) return fail("this area light radiates a bitmap: update the texture (har_scene_set_texture)");
    e.radiance[0] = rgb[0]; e.radiance[1] = rgb[1]; e.radiance[2] = rgb[2];
    HIP_TRY(hipMemcpy(const_cast<DEmitter *>(S->ds.emitters) + emitter, &e, sizeof(DEmitter), hipMemcpyHostToDevice));
    if (S->hs.emitters.size() == 1) { S->ds.emitter0 = e; S->ds.emitter0_valid = 1u; }
    return 0;
}
/* the records whose lobe-selection weight depends on the MEAN of texture `tex` (RoughPlastic / SmoothPlastic::parameters_changed, roughplastic.cpp:204-242,
 * plastic.cpp:188-205: m_specular_sampling_weight from the means of the two reflectances) */
static bool texture_lights_an_emitter(const HostScene &hs, uint32_t tex) {
    for (const DEmitter &e : hs.emitters) if (e.type == 7u && as_u32(e.radiance[0]) == tex) return true;
    return false;
}
static bool texture_feeds_sampling_weight(const HostScene &hs, uint32_t tex) {
    for (const DBsdf &b : hs.bsdfs) if (b.texture == (int32_t) tex && (b.type == BSDF_ROUGHPLASTIC || b.type == BSDF_PLASTIC)) return true;
    return texture_lights_an_emitter(hs, tex);      /* an area light radiates it: its texel distribution is derived on the host */
}
/* the texel distributions of the area lights that radiate bitmap `tex` (BitmapTexture::parameters_changed -> rebuild_internals, bitmap.cpp:484-493): re-derived from the
 * host mirror of the texels and copied over their slice of the device table */
static int refresh_texel_tables(HarSceneImpl *S, uint32_t tex) {
    for (const DEmitter &e : S->hs.emitters) {
        if (e.type != 7u || as_u32(e.radiance[0]) != tex) continue;
        const HostTexture &t = S->hs.textures[tex];
        const uint32_t off = as_u32(e.radiance[1]);
        std::string err;
        if (!texel_table_fill(S->hs, t, off, err)) return fail(err);
        const size_t n = HAR_TEXEL_TABLE_HEADER + (size_t) t.h + (size_t) t.w * t.h;
        HIP_TRY(hipMemcpy(const_cast<float *>(S->ds.emitter_cdf) + off, S->hs.emitter_cdf.data() + off, n * sizeof(float), hipMemcpyHostToDevice));
    }
    return 0;
}
static int refresh_sampling_weights(HarSceneImpl *S, uint32_t tex) {
    for (uint32_t k = 0; k < S->hs.bsdfs.size(); ++k) {
        DBsdf &b = S->hs.bsdfs[k];
        if (b.texture != (int32_t) tex || !(b.type == BSDF_ROUGHPLASTIC || b.type == BSDF_PLASTIC)) continue;
        update_roughplastic_sampling_weight(S->hs, k);
        HIP_TRY(hipMemcpy(S->d_bsdfs + k, &b, sizeof(DBsdf), hipMemcpyHostToDevice));
    }
    return refresh_texel_tables(S, tex);
}
int har_scene_set_texture(HarScene S, uint32_t tex, const float *data) {
    if (!S || tex >= S->hs.textures.size()) return fail("invalid texture index");
    HostTexture &t = S->hs.textures[tex];
    if (texture_lights_an_emitter(S->hs, tex)) { std::string err; if (!texel_table_inputs_ok(t.uvm, data, t.w, t.h, err)) return fail(err); }      /* before anything changes */
    t.data.assign(data, data + t.data.size());
    HIP_TRY(hipMemcpy(S->tex_dev[tex], data, t.data.size() * sizeof(float), hipMemcpyHostToDevice));
    S->tex_host_stale[tex] = 0;
    return refresh_sampling_weights(S, tex);
}
/* The same three updates from DEVICE memory, ordered on `stream`, without a host round trip: what an optimisation loop calls every step (mi.traverse +
 * params.update(), src/python/python/util.py:344-528 -- in the reference the parameters ARE device arrays and update() copies nothing).  The library's host
 * mirror of the value goes stale and is refreshed from the device only when something needs it (har_scene_set_* from the host overwrite it anyway).
 * Exception: a bitmap / colour that feeds the lobe-selection weight of a `plastic` / `roughplastic` record (the mean of the reflectance) -- that weight is
 * computed on the host, so these records take one synchronous device-to-host copy. */
int har_scene_set_texture_device(HarScene S, uint32_t tex, const float *dev, void *stream) {
    if (!S || tex >= S->hs.textures.size()) return fail("invalid texture index");
    if (!dev) return fail("null device pointer");
    HostTexture &t = S->hs.textures[tex];
    hipStream_t s = (hipStream_t) stream;
    if (texture_lights_an_emitter(S->hs, tex)) {          /* its texel distribution is derived on the host; the new texels are checked before anything changes */
        std::vector<float> incoming(t.data.size());
        HIP_TRY(hipMemcpyAsync(incoming.data(), dev, incoming.size() * sizeof(float), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        std::string err;
        if (!texel_table_inputs_ok(t.uvm, incoming.data(), t.w, t.h, err)) return fail(err);
    }
    if (texture_feeds_sampling_weight(S->hs, tex)) {
        HIP_TRY(hipMemcpyAsync(t.data.data(), dev, t.data.size() * sizeof(float), hipMemcpyDeviceToHost, s));
        if (dev != S->tex_dev[tex]) HIP_TRY(hipMemcpyAsync(S->tex_dev[tex], dev, t.data.size() * sizeof(float), hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));
        S->tex_host_stale[tex] = 0;
        return refresh_sampling_weights(S, tex);
    }
    if (dev != S->tex_dev[tex]) HIP_TRY(hipMemcpyAsync(S->tex_dev[tex], dev, t.data.size() * sizeof(float), hipMemcpyDeviceToDevice, s));
    S->tex_host_stale[tex] = 1; S->last_push_stream = s; S->last_push_valid = true;
    return 0;
}
int har_scene_set_reflectance_device(HarScene S, uint32_t bsdf, const float *dev_rgb, void *stream) {
    if (!S || bsdf >= S->hs.bsdfs.size()) return fail("invalid bsdf index");
    if (!dev_rgb) return fail("null device pointer");
    DBsdf &b = S->hs.bsdfs[bsdf];
    hipStream_t s = (hipStream_t) stream;
    if (b.type == BSDF_ROUGHPLASTIC || b.type == BSDF_PLASTIC) {           /* its sampling weight depends on the colour: through the host */
        float rgb[3];
        HIP_TRY(hipMemcpyAsync(rgb, dev_rgb, sizeof(rgb), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        return har_scene_set_reflectance(S, bsdf, rgb);
    }
    static_assert(offsetof(DBsdf, g) == offsetof(DBsdf, r) + 4 && offsetof(DBsdf, b) == offsetof(DBsdf, r) + 8, "slot 0 is three consecutive floats");
    HIP_TRY(hipMemcpyAsync(&S->d_bsdfs[bsdf].r, dev_rgb, 3 * sizeof(float), hipMemcpyDeviceToDevice, s));
    S->bsdf_host_stale = true; S->last_push_stream = (hipStream_t) stream; S->last_push_valid = true;
    return 0;
}
int har_scene_set_emitter_radiance_device(HarScene S, uint32_t emitter, const float *dev_rgb, void *stream) {
    if (!S || emitter >= S->hs.emitters.size()) return fail("invalid emitter index");
    if (!dev_rgb) return fail("null device pointer");
    if (S->hs.emitters[emitter].type == 2u) return fail("an environment map has no constant radiance");
    if (S->hs.emitters[emitter].type == 7u) return fail("this area light radiates a bitmap: update the texture (har_scene_set_texture_device)");
    HIP_TRY(hipMemcpyAsync(const_cast<float *>(S->ds.emitters[0].radiance) + (size_t) emitter * (sizeof(DEmitter) / sizeof(float)), dev_rgb, 3 * sizeof(float),
                           hipMemcpyDeviceToDevice, (hipStream_t) stream));
    S->emitter_host_stale = true; S->last_push_stream = (hipStream_t) stream; S->last_push_valid = true;
    if (S->ds.emitter0_valid) S->ds.emitter0_valid = 2u;      /* the argument copy no longer holds the radiance: the kernels take those three floats from the array, the rest of the record stays in scalar registers */
    return 0;
}
int har_scene_accel_info(HarScene S, uint64_t info[4]) {
    if (!S) return fail("null scene");
    info[0] = S->hs.nodes.size(); info[1] = S->hs.tris.size();
    info[2] = S->hs.nodes.size() * sizeof(Node8) + S->hs.tris.size() * sizeof(TriRec) + S->hs.inst_recs.size() * sizeof(InstRec);
    info[3] = S->hs.stack_need();
    return 0;
}

/* Scene::sample_emitter / pdf_emitter (src/render/scene.cpp:248-279), array-valued */
int har_scene_sample_emitter(HarScene S, uint32_t n, const float *index_sample, const uint8_t *active, uint32_t *index, float *weight, float *reused_sample, void *stream) {
    if (!S) return fail("null scene");
    if (n == 0) return 0;
    if (!index_sample || !index || !weight || !reused_sample) return fail("null sample / output arrays");
    launch_api_sample_emitter((hipStream_t) stream, S->ds, n, index_sample, active, index, weight, reused_sample);
    HIP_TRY(hipGetLastError());
    return 0;
}
int har_scene_pdf_emitter(HarScene S, uint32_t n, const uint32_t *index, const uint8_t *active, float *pdf, void *stream) {
    if (!S) return fail("null scene");
    if (n == 0) return 0;
    if (!index || !pdf) return fail("null index / output arrays");
    launch_api_pdf_emitter((hipStream_t) stream, S->ds, n, index, active, pdf);
    HIP_TRY(hipGetLastError());
    return 0;
}
/* params['<emitter>.sampling_weight'] + update(): Scene::parameters_changed -> update_emitter_sampling_distribution (scene.cpp:120-141, 523-528) */
int har_scene_set_emitter_sampling_weights(HarScene S, const float *weights, uint32_t count) {
    if (!S || !weights) return fail("null argument");
    if (count != S->hs.emitters.size()) return fail("one sampling weight per emitter of the scene");
    std::string e;
    if (!build_emitter_distribution(S->hs, weights, count, e)) return fail(e);
    if (!S->hs.emitter_distr.empty()) HIP_TRY(hipMemcpy(S->d_emitter_distr, S->hs.emitter_distr.data(), S->hs.emitter_distr.size() * sizeof(float), hipMemcpyHostToDevice));
    S->hs.bind_tables(S->ds, S->d_emitter_distr);
    if (S->emitter_host_stale && S->ds.emitter0_valid) S->ds.emitter0_valid = 2u;      /* a radiance pushed device-to-device is newer than the mirror bind_tables copies */
    /* scenes with a distribution run the kernels that carry the generic emitter code */
    const bool generic = S->hs.has_envmap || S->hs.has_mesh_emitters || S->hs.has_point_emitters || !S->hs.emitter_distr.empty();
    S->ds.bsdf_types = generic ? (S->ds.bsdf_types | HAR_SCENE_ENVMAP) : (S->ds.bsdf_types & ~HAR_SCENE_ENVMAP);
    return 0;
}
/* params['<texture>.to_uv'] + update(): BitmapTexture::parameters_changed with a new m_transform (bitmap.cpp:175); six zeros or the identity switch the transform off */
int har_scene_set_texture_to_uv(HarScene S, uint32_t tex, const float to_uv[6]) {
    if (!S || tex >= S->hs.textures.size() || !to_uv) return fail("invalid texture index");
    HostTexture &t = S->hs.textures[tex];
    const float id6[6] = { 1.f, 0.f, 0.f, 0.f, 1.f, 0.f };
    bool zero = true, ident = true;
    for (int k = 0; k < 6; ++k) { if (!std::isfinite(to_uv[k])) return fail("HarTexture::to_uv must be finite"); zero = zero && to_uv[k] == 0.f; ident = ident && to_uv[k] == id6[k]; }
    if (!zero && !ident && to_uv[0] * to_uv[4] - to_uv[1] * to_uv[3] == 0.f) return fail("HarTexture::to_uv is singular");
    if (texture_lights_an_emitter(S->hs, tex)) {          /* the emitter's texel distribution needs a to_uv that keeps the unit square (bitmap.cpp:976-992): checked before anything changes */
        if (S->tex_host_stale[tex]) { wait_for_device_pushes(S); HIP_TRY(hipMemcpy(t.data.data(), S->tex_dev[tex], t.data.size() * sizeof(float), hipMemcpyDeviceToHost)); S->tex_host_stale[tex] = 0; }
        std::string err;
        if (!texel_table_inputs_ok((zero || ident) ? id6 : to_uv, t.data.data(), t.w, t.h, err)) return fail(err);
    }
    t.mode &= ~HAR_TEX_HAS_UV_XF;
    for (int k = 0; k < 6; ++k) t.uvm[k] = id6[k];
    if (!zero && !ident) {
        for (int k = 0; k < 6; ++k) t.uvm[k] = to_uv[k];
        t.mode |= HAR_TEX_HAS_UV_XF;
    }
    const DTexture d = S->hs.device_texture(tex, S->tex_dev[tex]);
    HIP_TRY(hipMemcpy(S->d_textures + tex, &d, sizeof(DTexture), hipMemcpyHostToDevice));
    return refresh_texel_tables(S, tex);
}

/* ---- incremental updates of the acceleration data (Scene::parameters_changed rebuilds only what a dirty shape needs, scene.cpp:517-540; scene_optix.inl:351-372) */
static int stack_fits(const HostScene &hs) {
    const uint32_t stack_cap = (uint32_t) std::min(HAR_LDS_STACK_DEPTH, HAR_LDS_STACK_SMALL + HAR_STACK_SPILL);
    if (hs.stack_need() + HAR_STACK_MARGIN > stack_cap) return fail("the updated scene's BVH needs " + std::to_string(hs.stack_need() + HAR_STACK_MARGIN) + " traversal stack entries per ray, the kernels hold " + std::to_string(stack_cap));
    return 0;
}
/* the arrays build_tlas / update_scene_bounds rewrote on the host -> device, in stream order; the copies read pageable host memory, so the call waits for them */
static int upload_instance_level(HarSceneImpl *S, hipStream_t s) {
    HostScene &hs = S->hs; DScene &D = S->ds;
    if (hs.nodes.size() > S->nodes_cap) return fail("TLAS does not fit the node array");           /* cannot happen: capacity = BLAS nodes + instance count + 2 */
    if (stack_fits(hs)) return 1;
    const size_t tail = hs.nodes.size() - hs.tlas_first;
    if (tail) HIP_TRY(hipMemcpyAsync(const_cast<Node8 *>(D.accel.nodes) + hs.tlas_first, hs.nodes.data() + hs.tlas_first, tail * sizeof(Node8), hipMemcpyHostToDevice, s));
    if (!hs.inst_recs.empty()) HIP_TRY(hipMemcpyAsync(const_cast<InstRec *>(D.accel.insts), hs.inst_recs.data(), hs.inst_recs.size() * sizeof(InstRec), hipMemcpyHostToDevice, s));
    if (!hs.blas_tri_ranges.empty()) HIP_TRY(hipMemcpyAsync(const_cast<uint32_t *>(D.blas_tri_ranges), hs.blas_tri_ranges.data(), hs.blas_tri_ranges.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    if (!hs.insts.empty()) HIP_TRY(hipMemcpyAsync(const_cast<DInst *>(D.insts), hs.insts.data(), hs.insts.size() * sizeof(DInst), hipMemcpyHostToDevice, s));
    D.accel.root = hs.root; D.accel.n_insts = (uint32_t) hs.inst_recs.size();
    D.accel.top_root = hs.top_root; D.accel.top_first = hs.top_first; D.accel.top_count = hs.top_count; D.accel.top_last = hs.top_last;
    return 0;
}
/* the records update_scene_bounds touches: the environment / directional emitters' bounding sphere */
static int upload_scene_bounds(HarSceneImpl *S, hipStream_t s) {
    HostScene &hs = S->hs;
    bool any = hs.env_emitter >= 0; for (const DEmitter &E : hs.emitters) any = any || E.type == 6u;
    if (!any) return 0;
    if (S->emitter_host_stale) {          /* radiances pushed device-to-device are newer than the mirror: fetch them before the records are rewritten */
        std::vector<DEmitter> cur(hs.emitters.size());
        HIP_TRY(hipMemcpyAsync(cur.data(), S->ds.emitters, cur.size() * sizeof(DEmitter), hipMemcpyDeviceToHost, s)); HIP_TRY(hipStreamSynchronize(s));
        for (size_t k = 0; k < cur.size(); ++k) std::memcpy(hs.emitters[k].radiance, cur[k].radiance, 12);
        S->emitter_host_stale = false;
    }
    HIP_TRY(hipMemcpyAsync(const_cast<DEmitter *>(S->ds.emitters), hs.emitters.data(), hs.emitters.size() * sizeof(DEmitter), hipMemcpyHostToDevice, s));
    if (hs.emitters.size() == 1) { S->ds.emitter0 = hs.emitters[0]; S->ds.emitter0_valid = 1u; }
    if (hs.has_envmap && S->ds.envmap) {
        DEnvmap E = hs.envmap;         /* tex / warp already hold the device pointers (har_scene_create) */
        HIP_TRY(hipMemcpyAsync(const_cast<DEnvmap *>(S->ds.envmap), &E, sizeof(DEnvmap), hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    return 0;
}
static int refresh_host_vertices(HarSceneImpl *S, hipStream_t s, int only_mesh);
static int refresh_host_geometry(HarSceneImpl *S, hipStream_t s);
int har_scene_update_instances(HarScene S, uint32_t first, uint32_t count, const float *to_world, const float *to_object, void *stream) {
    if (!S || !to_world || !to_object) return fail("null argument");
    if (count == 0) return 0;
    std::string e;
    hipStream_t s = (hipStream_t) stream;
    if (refresh_host_geometry(S, s)) return 1;              /* the instance boxes and the scene bounds are host builds over the vertex positions and the other instances' transforms */
    if (!scene_set_instances_host(S->hs, first, count, to_world, to_object, e)) return fail(e);
    if (upload_instance_level(S, s) || upload_scene_bounds(S, s)) return 1;
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}
/* refit scratch, on first use; S->tri_box (the "scratch exists" flag) is set last, so a failed allocation leaves no half-made set behind */
static int ensure_refit_scratch(HarSceneImpl *S, hipStream_t s) {
    if (S->tri_box) return 0;
    HostScene &hs = S->hs;
    const size_t n_blas = 1 + hs.blas_groups.size();
    void *p = nullptr;
    HIP_TRY(dev_alloc(&p, std::max<size_t>(hs.tris.size(), 1) * sizeof(RefitBox))); S->owned.push_back(p); RefitBox *tri_box = (RefitBox *) p;
    HIP_TRY(dev_alloc(&p, std::max<size_t>(S->nodes_cap, 1) * sizeof(RefitBox))); S->owned.push_back(p); S->node_box = (RefitBox *) p;
    HIP_TRY(dev_alloc(&p, std::max<size_t>(hs.refit_order.size(), 1) * sizeof(uint32_t))); S->owned.push_back(p); S->d_refit_order = (uint32_t *) p;
    if (!hs.refit_order.empty()) { HIP_TRY(hipMemcpyAsync(S->d_refit_order, hs.refit_order.data(), hs.refit_order.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s)); HIP_TRY(hipStreamSynchronize(s)); }
    HIP_TRY(dev_alloc(&p, (n_blas + 1) * sizeof(float))); S->owned.push_back(p); S->d_area = (float *) p;          /* + 1: the instance level's sum (not watched) */
    S->tri_box = tri_box;
    return 0;
}
static size_t blas_slot(const HostScene &hs, const BlasInfo *B) { return B == &hs.blas_top ? 0 : 1 + (size_t) (B - hs.blas_groups.data()); }
/* the launches of one refit pass of BLAS `B` on the device arrays as they are (triangle records + boxes, then the nodes level by level, deepest first); the node-area
 * sum accumulates in d_area[slot] */
static int enqueue_refit(HarSceneImpl *S, BlasInfo *B, hipStream_t s) {
    const size_t bi = blas_slot(S->hs, B);
    HIP_TRY(hipMemsetAsync(S->d_area + bi, 0, sizeof(float), s));
    launch_refit_triangles(s, S->ds, B->first_tri, B->tri_count, S->tri_box);
    for (size_t l = 0; l + 1 < B->level_begin.size(); ++l)
        launch_refit_nodes(s, S->ds, S->d_refit_order + B->order_first + B->level_begin[l], B->level_begin[l + 1] - B->level_begin[l], S->tri_box, S->node_box, S->d_area + bi);
    HIP_TRY(hipGetLastError());
    return 0;
}
static double refit_cost_figure(float area, const RefitBox &root) {
    const float dx = root.hi[0] - root.lo[0], dy = root.hi[1] - root.lo[1], dz = root.hi[2] - root.lo[2];
    const double root_area = 2.0 * ((double) dx * dy + (double) dy * dz + (double) dz * dx);
    return root_area > 0.0 ? (double) area / root_area : 0.0;           /* sum of the node areas over the root's area (the node term of the SAH) */
}
/* one refit pass + its cost figure, waited for */
static int refit_pass_sync(HarSceneImpl *S, BlasInfo *B, hipStream_t s, double &cost) {
    if (enqueue_refit(S, B, s)) return 1;
    float area = 0.f; RefitBox root{};
    HIP_TRY(hipMemcpyAsync(&area, S->d_area + blas_slot(S->hs, B), sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&root, S->node_box + B->root, sizeof(RefitBox), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    cost = refit_cost_figure(area, root);
    return 0;
}
static int refit_verdict(HarSceneImpl *S, const BlasInfo *B, double cost) {
    static const double max_inflation = getenv("HAR_REFIT_MAX_INFLATION") ? atof(getenv("HAR_REFIT_MAX_INFLATION")) : 1.5;
    static const uint32_t max_refits = getenv("HAR_REFIT_MAX_STEPS") ? (uint32_t) atoi(getenv("HAR_REFIT_MAX_STEPS")) : 0u;
    S->last_refit_cost = cost; S->last_refit_ratio = B->built_area > 0.0 ? cost / B->built_area : 1.0;
    if ((B->built_area > 0.0 && cost > max_inflation * B->built_area) || (max_refits && B->refits >= max_refits)) return HAR_UPDATE_REBUILD_ADVISED;
    return 0;
}
/* device -> host refresh of the vertex records of the meshes a device-resident update left stale on the host (hs.verts, hs.shade_tris): before anything on the host
 * reads positions again (group boxes, scene bounds, har_scene_get_vertices) */
static int refresh_host_vertices(HarSceneImpl *S, hipStream_t s, int only_mesh) {
    HostScene &hs = S->hs; bool any = false;
    for (size_t k = 0; k < S->verts_host_stale.size(); ++k) {
        if (!S->verts_host_stale[k] || (only_mesh >= 0 && (size_t) only_mesh != k)) continue;
        const DMesh &m = hs.meshes[k];
        HIP_TRY(hipMemcpyAsync(hs.verts.data() + 8 * (size_t) m.voff, S->ds.verts + 8 * (size_t) m.voff, 32 * (size_t) m.vertex_count, hipMemcpyDeviceToHost, s));
        any = true;
    }
    if (!any) return 0;
    HIP_TRY(hipStreamSynchronize(s));
    for (size_t k = 0; k < S->verts_host_stale.size(); ++k) {
        if (!S->verts_host_stale[k] || (only_mesh >= 0 && (size_t) only_mesh != k)) continue;
        const DMesh &m = hs.meshes[k];
#if HAR_SHADING_TRIS
        for (uint32_t f = 0; f < m.face_count; ++f)
            for (int c = 0; c < 3; ++c) std::memcpy(hs.shade_tris.data() + 24 * ((size_t) m.foff + f) + 8 * c, hs.verts.data() + 8 * ((size_t) m.voff + hs.faces[4 * ((size_t) m.foff + f) + c]), 32);
#endif
        S->verts_host_stale[k] = 0;
    }
    return 0;
}
/* ... and of the instance transforms a device-resident update left stale on the host (hs.insts; build_tlas re-derives the leaf records from them) */
static int refresh_host_instances(HarSceneImpl *S, hipStream_t s) {
    if (!S->insts_host_stale || S->hs.insts.empty()) { S->insts_host_stale = false; return 0; }
    HIP_TRY(hipMemcpyAsync(S->hs.insts.data(), S->ds.insts, S->hs.insts.size() * sizeof(DInst), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    S->insts_host_stale = false;
    return 0;
}
/* everything a HOST build of the instance level / the scene bounds reads, brought up to date after device-resident updates */
static int refresh_host_geometry(HarSceneImpl *S, hipStream_t s) {
    if (refresh_host_vertices(S, s, -1) || refresh_host_instances(S, s)) return 1;
    recompute_stale_group_boxes(S->hs);
    return 0;
}
/* the figures the LAST device-resident update left in the pinned record: 0, HAR_UPDATE_REBUILD_ADVISED, or 1 (a position was not finite) */
static int collect_pending_refit(HarSceneImpl *S) {
    if (!S->pend_active) return 0;
    HIP_TRY(hipEventSynchronize(S->pend_ev));          /* recorded one update (= at least one frame) ago, or just synchronised by the caller */
    S->pend_active = false;
    if (S->pend->bad) return fail("har_scene_update_vertices_device: a vertex position of the last update was not finite (the scene holds it: create a new scene)");
    return refit_verdict(S, S->pend_blas, refit_cost_figure(S->pend->area, S->pend->root));
}
int har_scene_update_vertices(HarScene S, uint32_t mesh, const float *vertices, void *stream) {
    if (!S || !vertices) return fail("null argument");
    HostScene &hs = S->hs; DScene &D = S->ds;
    hipStream_t s = (hipStream_t) stream;
    std::string e;
    if (!S->verts_host_stale.empty()) {         /* other meshes may have been updated on the device since: the host steps below read their positions */
        if (mesh < S->verts_host_stale.size()) S->verts_host_stale[mesh] = 0;          /* this one is overwritten */
        if (refresh_host_geometry(S, s)) return 1;
        if (collect_pending_refit(S) == 1) return 1;
    }
    if (refresh_host_instances(S, s)) return 1;
    BlasInfo *B = scene_set_vertices_host(hs, mesh, vertices, e);
    if (!B) { (void) fail(e); return HAR_UPDATE_NEEDS_NEW_SCENE; }
    if (mesh < S->normals_regenerated.size()) S->normals_regenerated[mesh] = 0;
    const DMesh &m = hs.meshes[mesh];
    if (ensure_refit_scratch(S, s)) return 1;
    /* the figure of the tree AS BUILT: the first update of a BLAS refits it once on the old vertices (which reproduces the built nodes bit for bit) */
    if (B->built_area == 0.0 && refit_pass_sync(S, B, s, B->built_area)) return 1;
    HIP_TRY(hipMemcpyAsync(const_cast<float *>(D.verts) + 8 * (size_t) m.voff, vertices, 32 * (size_t) m.vertex_count, hipMemcpyHostToDevice, s));
#if HAR_SHADING_TRIS
    HIP_TRY(hipMemcpyAsync(const_cast<float *>(D.shade_tris) + 24 * (size_t) m.foff, hs.shade_tris.data() + 24 * (size_t) m.foff, 96 * (size_t) m.face_count, hipMemcpyHostToDevice, s));
#endif
    double cost = 0.0;
    if (refit_pass_sync(S, B, s, cost)) return 1;
    if (!scene_after_refit_host(hs, B, e)) return fail(e);
    if (B != &hs.blas_top && upload_instance_level(S, s)) return 1;
    if (upload_scene_bounds(S, s)) return 1;
    HIP_TRY(hipStreamSynchronize(s));
    return refit_verdict(S, B, cost);
}
static bool scene_needs_bounds(const HostScene &hs) {
    bool any = hs.env_emitter >= 0; for (const DEmitter &E : hs.emitters) any = any || E.type == 6u;
    return any;
}
/* corner list of a mesh (har_vertex_update.h), built on the host once and kept on the device */
static int ensure_corner_list(HarSceneImpl *S, uint32_t mesh, hipStream_t s) {
    HostScene &hs = S->hs;
    if (S->d_corner_begin.size() != hs.meshes.size()) { S->d_corner_begin.assign(hs.meshes.size(), nullptr); S->d_corners.assign(hs.meshes.size(), nullptr); }
    if (S->d_corner_begin[mesh]) return 0;
    const DMesh &m = hs.meshes[mesh];
    std::vector<uint32_t> begin((size_t) m.vertex_count + 1, 0u), corners(3 * (size_t) m.face_count);
    const uint32_t *F = hs.faces.data() + 4 * (size_t) m.foff;
    for (uint32_t f = 0; f < m.face_count; ++f) for (int k = 0; k < 3; ++k) ++begin[(size_t) F[4 * (size_t) f + k] + 1];
    for (uint32_t v = 0; v < m.vertex_count; ++v) begin[v + 1] += begin[v];
    std::vector<uint32_t> cursor(begin.begin(), begin.end() - 1);
    for (uint32_t f = 0; f < m.face_count; ++f) for (int k = 0; k < 3; ++k) corners[cursor[F[4 * (size_t) f + k]]++] = f | ((uint32_t) k << 30);      /* (face, corner) ascending per vertex */
    void *p = nullptr;
    HIP_TRY(dev_alloc(&p, begin.size() * sizeof(uint32_t))); S->owned.push_back(p); uint32_t *d_begin = (uint32_t *) p;
    HIP_TRY(dev_alloc(&p, std::max<size_t>(corners.size(), 1) * sizeof(uint32_t))); S->owned.push_back(p); uint32_t *d_corners = (uint32_t *) p;
    HIP_TRY(hipMemcpyAsync(d_begin, begin.data(), begin.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    if (!corners.empty()) HIP_TRY(hipMemcpyAsync(d_corners, corners.data(), corners.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));                  /* once per mesh: the vectors die here */
    S->d_corners[mesh] = d_corners; S->d_corner_begin[mesh] = d_begin;
    return 0;
}
/* the device tables of the instance-level refit, for the TLAS as the host last built it (once per build_tlas: a small upload that waits) */
static int ensure_tlas_refit_tables(HarSceneImpl *S, hipStream_t s) {
    HostScene &hs = S->hs;
    if (S->d_tlas_serial == hs.tlas_serial && S->d_tlas_order) return 0;
    const size_t n_nodes = hs.tlas_order.size(), n_rec = hs.inst_recs.size();
    void *p = nullptr;
    if (n_nodes > S->d_tlas_cap) { HIP_TRY(dev_alloc(&p, n_nodes * sizeof(uint32_t))); S->owned.push_back(p); S->d_tlas_order = (uint32_t *) p; S->d_tlas_cap = n_nodes; }
    if (n_rec > S->d_inst_cap) {
        HIP_TRY(dev_alloc(&p, n_rec * sizeof(uint2))); S->owned.push_back(p); S->d_inst_vrange = (uint2 *) p;
        HIP_TRY(dev_alloc(&p, n_rec * sizeof(RefitBox))); S->owned.push_back(p); S->d_inst_box = (RefitBox *) p;
        S->d_inst_cap = n_rec;
    }
    const size_t n_inst = hs.insts.size();
    if (n_inst > S->d_rec_of_cap) { HIP_TRY(dev_alloc(&p, n_inst * sizeof(uint32_t))); S->owned.push_back(p); S->d_rec_of = (uint32_t *) p; S->d_rec_of_cap = n_inst; }
    std::vector<uint32_t> rec_of(n_inst, 0xffffffffu);
    for (size_t r = 0; r < n_rec; ++r) rec_of[hs.inst_recs[r].inst_index] = (uint32_t) r;
    if (n_inst) HIP_TRY(hipMemcpyAsync(S->d_rec_of, rec_of.data(), n_inst * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    std::vector<uint2> vr(n_rec);
    for (size_t r = 0; r < n_rec; ++r) {
        const HarShapeGroup &sg = hs.groups[hs.inst_group[hs.inst_recs[r].inst_index]];
        uint32_t cnt = 0; for (uint32_t m = sg.first_mesh; m < sg.first_mesh + sg.mesh_count; ++m) cnt += hs.meshes[m].vertex_count;
        vr[r] = make_uint2(hs.meshes[sg.first_mesh].voff, cnt);
    }
    if (n_nodes) HIP_TRY(hipMemcpyAsync(S->d_tlas_order, hs.tlas_order.data(), n_nodes * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    if (n_rec) HIP_TRY(hipMemcpyAsync(S->d_inst_vrange, vr.data(), n_rec * sizeof(uint2), hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    S->d_tlas_serial = hs.tlas_serial;
    return 0;
}
int har_scene_update_vertices_device(HarScene S, uint32_t mesh, const float *positions, void *stream) {
    if (!S || !positions) return fail("null argument");
    HostScene &hs = S->hs; DScene &D = S->ds;
    hipStream_t s = (hipStream_t) stream;
    if (mesh >= hs.meshes.size()) { (void) fail("invalid mesh index"); return HAR_UPDATE_NEEDS_NEW_SCENE; }
    const DMesh &m = hs.meshes[mesh];
    if (m.emitter >= 0) { (void) fail("the mesh carries an area emitter (its sampling records are lowered from the positions): create a new scene"); return HAR_UPDATE_NEEDS_NEW_SCENE; }
    BlasInfo *B = nullptr;
    if (mesh < hs.top_mesh_count) B = &hs.blas_top;
    else for (size_t g = 0; g < hs.groups.size(); ++g) if (mesh >= hs.groups[g].first_mesh && mesh < hs.groups[g].first_mesh + hs.groups[g].mesh_count) B = &hs.blas_groups[g];
    if (!B) return fail("mesh belongs to no BLAS");
    /* what the PREVIOUS update's refit reported (its launches finished a frame ago): acted on one step late, so that this call waits for nothing */
    int verdict = collect_pending_refit(S);
    if (verdict == 1) return 1;
    if (S->verts_host_stale.size() != hs.meshes.size()) { S->verts_host_stale.assign(hs.meshes.size(), 0); S->normals_regenerated.assign(hs.meshes.size(), 0); }
    if (ensure_refit_scratch(S, s)) return 1;
    if (!S->pend) {
        HIP_TRY(hipHostMalloc((void **) &S->pend, sizeof(*S->pend), hipHostMallocDefault));
        HIP_TRY(hipEventCreateWithFlags(&S->pend_ev, hipEventDisableTiming));
        void *p = nullptr; HIP_TRY(dev_alloc(&p, sizeof(uint32_t))); S->owned.push_back(p); S->d_bad = (uint32_t *) p;
    }
    if ((m.flags & 1u) && ensure_corner_list(S, mesh, s)) return 1;
    if (B->built_area == 0.0 && refit_pass_sync(S, B, s, B->built_area)) return 1;          /* once per BLAS: the figure of the tree as built */
    HIP_TRY(hipMemsetAsync(S->d_bad, 0, sizeof(uint32_t), s));
    launch_set_positions(s, D, m.voff, m.vertex_count, positions, S->d_bad);
    if (m.flags & 1u) { launch_vertex_normals(s, D, m.voff, m.foff, m.vertex_count, S->d_corner_begin[mesh], S->d_corners[mesh]); S->normals_regenerated[mesh] = 1; }
#if HAR_SHADING_TRIS
    launch_shading_triangles(s, D, m.voff, m.foff, m.face_count);
#endif
    if (enqueue_refit(S, B, s)) return 1;
    HIP_TRY(hipMemcpyAsync(&S->pend->area, S->d_area + blas_slot(hs, B), sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&S->pend->root, S->node_box + B->root, sizeof(RefitBox), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&S->pend->bad, S->d_bad, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipEventRecord(S->pend_ev, s));
    S->pend_active = true; S->pend_blas = B;
    S->verts_host_stale[mesh] = 1;
    if (B == &hs.blas_top && !scene_needs_bounds(hs)) { B->refits++; return verdict; }
    static const bool host_tlas = getenv("HAR_HOST_TLAS_UPDATE") != nullptr;
    if (B != &hs.blas_top && !scene_needs_bounds(hs) && !host_tlas && hs.has_tlas && !hs.tlas_order.empty()) {
        /* an instanced mesh: the boxes of the group's instances move with its vertices.  The instance level keeps its topology and is REFITTED on the device like the
         * BLAS -- every leaf record's exact world-space bound (k_instance_boxes), then the TLAS nodes deepest level first -- so this update waits for nothing either.
         * (The host's cached instance boxes and the group's box are marked stale and re-derived from the refreshed vertices before the next HOST build of the TLAS.) */
        if (ensure_tlas_refit_tables(S, s)) return 1;
        launch_instance_boxes(s, D, (uint32_t) hs.inst_recs.size(), S->d_inst_vrange, S->d_inst_box);
        const size_t n_blas = 1 + hs.blas_groups.size();
        HIP_TRY(hipMemsetAsync(S->d_area + n_blas, 0, sizeof(float), s));
        for (size_t l = 0; l + 1 < hs.tlas_levels.size(); ++l)
            launch_refit_nodes(s, D, S->d_tlas_order + hs.tlas_levels[l], hs.tlas_levels[l + 1] - hs.tlas_levels[l], S->d_inst_box, S->node_box, S->d_area + n_blas);
        HIP_TRY(hipGetLastError());
        const size_t g = (size_t) (B - hs.blas_groups.data());
        if (hs.group_box_stale.size() != hs.groups.size()) hs.group_box_stale.assign(hs.groups.size(), 0);
        hs.group_box_stale[g] = 1;
        for (size_t i = 0; i < hs.insts.size(); ++i) if (hs.inst_group[i] == g) hs.inst_box_valid[i] = 0;
        B->refits++;
        return verdict;
    }
    /* environment / directional emitters follow the scene's bounding sphere (and HAR_HOST_TLAS_UPDATE keeps the instance level a host build): host builds over exact
     * vertex bounds, so these updates read the mesh back (32 B per vertex, device -> host) and wait -- still no host -> device copy of geometry */
    if (refresh_host_geometry(S, s)) return 1;
    const int now = collect_pending_refit(S);
    if (now == 1) return 1;
    std::string e;
    if (!scene_after_refit_host(hs, B, e)) return fail(e);
    if (B != &hs.blas_top && upload_instance_level(S, s)) return 1;
    if (upload_scene_bounds(S, s)) return 1;
    HIP_TRY(hipStreamSynchronize(s));
    return now ? now : verdict;
}
int har_scene_update_instances_device(HarScene S, uint32_t first, uint32_t count, const float *to_world, void *stream) {
    if (!S || !to_world) return fail("null argument");
    if (count == 0) return 0;
    HostScene &hs = S->hs; DScene &D = S->ds;
    hipStream_t s = (hipStream_t) stream;
    if ((uint64_t) first + count > hs.insts.size()) return fail("instance range out of bounds");
    if (!hs.has_tlas) return fail("the scene has no instances");
    /* what the previous device-resident instance update reported (its launches finished a frame ago) */
    if (S->pend_inst_active) {
        HIP_TRY(hipEventSynchronize(S->pend_inst_ev)); S->pend_inst_active = false;
        if (*S->pend_inst) return fail("har_scene_update_instances_device: an instance transform of the last update was singular or not finite (that instance kept its old transform)");
    }
    static const bool host_tlas = getenv("HAR_HOST_TLAS_UPDATE") != nullptr;
    if (scene_needs_bounds(hs) || host_tlas || hs.tlas_order.empty()) {
        /* emitters that follow the scene's bounding sphere: the host path (read the matrices back, invert, rebuild the instance level and the bounds) */
        std::vector<float> tw(12 * (size_t) count), to(12 * (size_t) count);
        HIP_TRY(hipMemcpyAsync(tw.data(), to_world, tw.size() * sizeof(float), hipMemcpyDeviceToHost, s)); HIP_TRY(hipStreamSynchronize(s));
        for (uint32_t k = 0; k < count; ++k) if (!affine_inverse(tw.data() + 12 * (size_t) k, to.data() + 12 * (size_t) k)) return fail("instance transform is singular or not finite");
        return har_scene_update_instances(S, first, count, tw.data(), to.data(), stream);
    }
    if (ensure_refit_scratch(S, s) || ensure_tlas_refit_tables(S, s)) return 1;
    if (!S->pend_inst) {
        HIP_TRY(hipHostMalloc((void **) &S->pend_inst, sizeof(uint32_t), hipHostMallocDefault)); *S->pend_inst = 0u;
        HIP_TRY(hipEventCreateWithFlags(&S->pend_inst_ev, hipEventDisableTiming));
        void *p = nullptr; HIP_TRY(dev_alloc(&p, sizeof(uint32_t))); S->owned.push_back(p); S->d_bad_inst = (uint32_t *) p;
    }
    /* transforms + inverses into the shading records and the TLAS leaf records, the instances' exact world-space bounds, the TLAS nodes deepest level first: the instance
     * level keeps the topology of its last host build (a refit, like the BLAS after a vertex update) */
    HIP_TRY(hipMemsetAsync(S->d_bad_inst, 0, sizeof(uint32_t), s));
    launch_set_instances(s, D, S->d_rec_of, first, count, to_world, S->d_bad_inst);
    launch_instance_boxes(s, D, (uint32_t) hs.inst_recs.size(), S->d_inst_vrange, S->d_inst_box);
    const size_t n_blas = 1 + hs.blas_groups.size();
    HIP_TRY(hipMemsetAsync(S->d_area + n_blas, 0, sizeof(float), s));
    for (size_t l = 0; l + 1 < hs.tlas_levels.size(); ++l)
        launch_refit_nodes(s, D, S->d_tlas_order + hs.tlas_levels[l], hs.tlas_levels[l + 1] - hs.tlas_levels[l], S->d_inst_box, S->node_box, S->d_area + n_blas);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(S->pend_inst, S->d_bad_inst, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipEventRecord(S->pend_inst_ev, s));
    S->pend_inst_active = true; S->insts_host_stale = true;
    for (uint32_t k = 0; k < count; ++k) hs.inst_box_valid[first + k] = 0;
    return 0;
}
int har_scene_get_instances(HarScene S, uint32_t first, uint32_t count, float *to_world, float *to_object, void *stream) {
    if (!S || !to_world || !to_object) return fail("null argument");
    if ((uint64_t) first + count > S->hs.insts.size()) return fail("instance range out of bounds");
    if (refresh_host_instances(S, (hipStream_t) stream)) return 1;
    for (uint32_t k = 0; k < count; ++k) { std::memcpy(to_world + 12 * (size_t) k, S->hs.insts[first + k].to_world, 48); std::memcpy(to_object + 12 * (size_t) k, S->hs.insts[first + k].to_object, 48); }
    return 0;
}
int har_scene_get_vertices(HarScene S, uint32_t mesh, float *vertices, void *stream) {
    if (!S || !vertices) return fail("null argument");
    if (mesh >= S->hs.meshes.size()) return fail("invalid mesh index");
    if (refresh_host_vertices(S, (hipStream_t) stream, (int) mesh)) return 1;
    const DMesh &m = S->hs.meshes[mesh];
    std::memcpy(vertices, S->hs.verts.data() + 8 * (size_t) m.voff, 32 * (size_t) m.vertex_count);
    return 0;
}
int har_scene_refit_info(HarScene S, double info[4]) {
    if (!S || !info) return fail("null argument");
    uint32_t refits = S->hs.blas_top.refits; for (const BlasInfo &b : S->hs.blas_groups) refits += b.refits;
    info[0] = (double) refits; info[1] = S->last_refit_cost; info[2] = S->last_refit_ratio; info[3] = (double) S->hs.nodes.size();
    return 0;
}

static int read_status(int *d_status, hipStream_t s) {
    int st = 0;
    HIP_TRY(hipMemcpyAsync(&st, d_status, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (st == HAR_STACK_OVERFLOW) return fail("BVH traversal stack overflow (scene too deep for the LDS stack)");
    return 0;
}

int har_ray_intersect_preliminary(HarScene S, uint32_t n, const float *o, const float *d, const float *maxt, const uint8_t *active, int naive, float *t, float *u,
                                  float *v, uint32_t *prim, uint32_t *shape, uint32_t *inst, void *stream) {
    if (!S) return fail("null scene");
    if (n == 0) return 0;
    int *st = nullptr; HIP_TRY(dev_alloc((void **) &st, sizeof(int))); HIP_TRY(hipMemsetAsync(st, 0, sizeof(int), (hipStream_t) stream));
    launch_api_intersect((hipStream_t) stream, S->ds, n, o, d, maxt, active, naive, t, u, v, prim, shape, inst, st);
    HIP_TRY(hipGetLastError());
    int rc = read_status(st, (hipStream_t) stream); dev_free(st);
    return rc;
}
int har_ray_test(HarScene S, uint32_t n, const float *o, const float *d, const float *maxt, const uint8_t *active, int naive, uint8_t *hit, void *stream) {
    if (!S) return fail("null scene");
    if (n == 0) return 0;
    int *st = nullptr; HIP_TRY(dev_alloc((void **) &st, sizeof(int))); HIP_TRY(hipMemsetAsync(st, 0, sizeof(int), (hipStream_t) stream));
    launch_api_ray_test((hipStream_t) stream, S->ds, n, o, d, maxt, active, naive, hit, st);
    HIP_TRY(hipGetLastError());
    int rc = read_status(st, (hipStream_t) stream); dev_free(st);
    return rc;
}
/* RayFlags the entry points accept (interaction.h:19-87): unknown bits and FollowShape together with DetachShape are refused, never ignored */
static int check_ray_flags(uint32_t ray_flags) {
    if (ray_flags & ~(uint32_t) RAY_KNOWN_FLAGS) return fail("ray_flags: unknown RayFlags bits (known: Minimal 0, Shading 1, NormalPartials 2, FollowShape 4, DetachShape 8)");
    if ((ray_flags & RAY_FOLLOW_SHAPE) && (ray_flags & RAY_DETACH_SHAPE)) return fail("ray_flags: at most one of FollowShape and DetachShape can be specified");
    return 0;
}
int har_compute_surface_interaction(HarScene S, uint32_t n, const float *o, const float *d, const float *t, const float *u, const float *v,
                                    const uint32_t *prim, const uint32_t *shape, const uint32_t *inst, uint32_t ray_flags, const uint8_t *active, float *out, void *stream) {
    if (!S) return fail("null scene");
    if (check_ray_flags(ray_flags)) return 1;
    if (n == 0) return 0;
    launch_api_si((hipStream_t) stream, S->ds, n, o, d, t, u, v, prim, shape, inst, ray_flags, active, out);
    HIP_TRY(hipGetLastError());
    return 0;
}
int har_ray_intersect(HarScene S, uint32_t n, const float *o, const float *d, const float *maxt, uint32_t ray_flags, const uint8_t *active, int naive, float *t, float *u,
                      float *v, uint32_t *prim, uint32_t *shape, uint32_t *inst, float *si, void *stream) {
    if (!S) return fail("null scene");
    if (check_ray_flags(ray_flags)) return 1;
    if (har_ray_intersect_preliminary(S, n, o, d, maxt, active, naive, t, u, v, prim, shape, inst, stream)) return 1;
    return har_compute_surface_interaction(S, n, o, d, t, u, v, prim, shape, inst, ray_flags, active, si, stream);
}
int har_sampler_seed(uint32_t seed, uint32_t lane_offset, uint32_t n, uint64_t *state, uint64_t *inc, void *stream) {
    if (n == 0) return 0;
    launch_api_sampler_seed((hipStream_t) stream, seed, lane_offset, n, state, inc);
    HIP_TRY(hipGetLastError());
    return 0;
}
int har_sampler_next_1d(uint32_t n, uint64_t *state, const uint64_t *inc, const uint8_t *active, float *out, void *stream) {
    if (n == 0) return 0;
    launch_api_sampler_next((hipStream_t) stream, n, state, inc, active, out, 1);
    HIP_TRY(hipGetLastError());
    return 0;
}
int har_sampler_next_2d(uint32_t n, uint64_t *state, const uint64_t *inc, const uint8_t *active, float *out, void *stream) {
    if (n == 0) return 0;
    launch_api_sampler_next((hipStream_t) stream, n, state, inc, active, out, 2);
    HIP_TRY(hipGetLastError());
    return 0;
}
/* HarBSDFContext -> BsdfCtx; NULL = BSDFContext() (Radiance, all lobes, all components) */
static int lower_ctx(const HarBSDFContext *ctx, BsdfCtx &out) {
    out = BsdfCtx();
    if (!ctx) return 0;
    if (ctx->mode > 1u) return fail("HarBSDFContext::mode must be 0 (TransportMode::Radiance) or 1 (TransportMode::Importance)");
    out.mode = ctx->mode; out.type_mask = ctx->type_mask; out.component = ctx->component;
    return 0;
}
static int bsdf_eval_common(HarScene S, uint32_t bsdf, const HarBSDFContext *ctx, uint32_t n, const float *wi, const float *uv, const float *wo, const uint8_t *active,
                            float *value, float *pdf, void *stream) {
    if (!S || bsdf >= S->hs.bsdfs.size()) return fail("invalid bsdf index");
    BsdfCtx c; if (lower_ctx(ctx, c)) return 1;
    if (n == 0) return 0;
    if (!wi || !uv || !wo) return fail("null wi / uv / wo");
    launch_api_bsdf_eval_pdf((hipStream_t) stream, S->ds, bsdf, c, n, wi, uv, wo, active, value, pdf);
    HIP_TRY(hipGetLastError());
    return 0;
}
int har_bsdf_eval_pdf(HarScene S, uint32_t bsdf, const HarBSDFContext *ctx, uint32_t n, const float *wi, const float *uv, const float *wo, const uint8_t *active,
                      float *value, float *pdf, void *stream) {
    if (!value || !pdf) return fail("null value / pdf");
    return bsdf_eval_common(S, bsdf, ctx, n, wi, uv, wo, active, value, pdf, stream);
}
int har_bsdf_eval(HarScene S, uint32_t bsdf, const HarBSDFContext *ctx, uint32_t n, const float *wi, const float *uv, const float *wo, const uint8_t *active, float *value, void *stream) {
    if (!value) return fail("null value");
    return bsdf_eval_common(S, bsdf, ctx, n, wi, uv, wo, active, value, nullptr, stream);
}
int har_bsdf_pdf(HarScene S, uint32_t bsdf, const HarBSDFContext *ctx, uint32_t n, const float *wi, const float *uv, const float *wo, const uint8_t *active, float *pdf, void *stream) {
    if (!pdf) return fail("null pdf");
    return bsdf_eval_common(S, bsdf, ctx, n, wi, uv, wo, active, nullptr, pdf, stream);
}
int har_bsdf_sample(HarScene S, uint32_t bsdf, const HarBSDFContext *ctx, uint32_t n, const float *wi, const float *uv, const float *sample1, const float *sample2,
                    const uint8_t *active, float *wo, float *pdf, float *weight, float *eta, uint32_t *sampled_type, uint32_t *sampled_component, void *stream) {
    if (!S || bsdf >= S->hs.bsdfs.size()) return fail("invalid bsdf index");
    BsdfCtx c; if (lower_ctx(ctx, c)) return 1;
    if (n == 0) return 0;
    if (!wi || !uv || !sample2 || !wo || !pdf || !weight) return fail("null input / output arrays");
    launch_api_bsdf_sample((hipStream_t) stream, S->ds, bsdf, c, n, wi, uv, sample1, sample2, active, wo, pdf, weight, eta, sampled_type, sampled_component);
    HIP_TRY(hipGetLastError());
    return 0;
}
int har_sensor_sample_ray(const HarSensor *sensor, uint32_t n, const float *px, const float *py, float *o, float *d, float *maxt, void *stream) {
    DSensor C; std::string e;
    if (!sensor || !lower_sensor(*sensor, C, e)) return fail(e.empty() ? "null sensor" : e);
    if (n == 0) return 0;
    launch_api_sensor_ray((hipStream_t) stream, C, n, px, py, o, d, maxt);
    HIP_TRY(hipGetLastError());
    return 0;
}
int har_film_put(const HarSensor *sensor, uint32_t n, const float *px, const float *py, const float *values4, float *film, void *stream) {
    DSensor C; std::string e;
    if (!sensor || !lower_sensor(*sensor, C, e)) return fail(e.empty() ? "null sensor" : e);
    if (n == 0) return 0;
    launch_api_film_put((hipStream_t) stream, C, n, px, py, values4, film);
    HIP_TRY(hipGetLastError());
    return 0;
}
int har_film_develop(const float *film, uint32_t width, uint32_t height, float *image, void *stream) {
    return har_film_develop_format(film, width, height, HAR_PIXEL_RGB, image, stream);
}
int har_film_develop_format(const float *film, uint32_t width, uint32_t height, int pixel_format, float *image, void *stream) {
    if (!film || !image) return fail("null film / image");
    if (pixel_format != HAR_PIXEL_RGB && pixel_format != HAR_PIXEL_Y && pixel_format != HAR_PIXEL_XYZ) return fail("har_film_develop_format: pixel_format must be HAR_PIXEL_RGB, _Y or _XYZ");
    launch_develop((hipStream_t) stream, film, width * height, image, pixel_format);
    HIP_TRY(hipGetLastError());
    return 0;
}

int har_integrator_create(int type, int32_t max_depth, int32_t rr_depth, uint32_t chunk_lanes, HarIntegrator *out) {
    if (!out) return fail("null argument");
    if (type != HAR_INTEGRATOR_PATH && type != HAR_INTEGRATOR_PRB) return fail("unknown integrator type");
    /* MonteCarloIntegrator ctor, src/render/integrator.cpp:539-550 */
    if (max_depth < 0 && max_depth != -1) return fail("\"max_depth\" must be set to -1 (infinite) or a value >= 0");
    if (rr_depth <= 0) return fail("\"rr_depth\" must be set to a value greater than zero!");
    HarIntegratorImpl *I = new HarIntegratorImpl();
    I->type = type; I->max_depth = (uint32_t) max_depth; I->rr_depth = (uint32_t) rr_depth;
    if (chunk_lanes) I->chunk = std::max<uint32_t>(2048u, (chunk_lanes + 2047u) / 2048u * 2048u);
    *out = I;
    return 0;
}
int har_integrator_destroy(HarIntegrator I) {
    if (!I) return 0;
    (void) hipDeviceSynchronize();
    I->free_ws();
    prof_destroy(I);
    if (I->twin) { I->twin->free_ws(); prof_destroy(I->twin); }
    if (I->ev_fork) (void) hipEventDestroy(I->ev_fork);
    if (I->ev_join) (void) hipEventDestroy(I->ev_join);
    if (I->ev_stagger) (void) hipEventDestroy(I->ev_stagger);
    if (I->side_stream) (void) hipStreamDestroy(I->side_stream);
    for (HarIntegratorImpl *J : { I->twin, I })
        if (J) {
            if (J->ptr_ring) (void) hipHostFree(J->ptr_ring);
            for (int k = 0; k < 8; ++k) if (J->ptr_ring_ev[k]) (void) hipEventDestroy(J->ptr_ring_ev[k]);
            if (J->ev_shaded) (void) hipEventDestroy(J->ev_shaded);
            if (J->ev_resolved) (void) hipEventDestroy(J->ev_resolved);
            if (J->ev_resolved2) (void) hipEventDestroy(J->ev_resolved2);
            if (J->aux_stream) (void) hipStreamDestroy(J->aux_stream);
        }
    if (I->twin) delete I->twin;
    delete I;
    return 0;
}

static int render_range(HarScene S, HarIntegrator I, const HarSensor *sensor, uint32_t seed, uint32_t spp, uint64_t lb, uint64_t le, float *film, void *stream) {
    DSensor C; uint32_t log_spp;
    if (!S || !I || !sensor) return fail("null scene / integrator / sensor");
    /* multi-pass layout; `path` only: the Python AD integrators render one wavefront or refuse (common.py:358-363) */
    uint32_t spp_pass = spp, n_passes = 1, grid_w = 0, grid_h = 0;
    if (sample_grid(sensor, grid_w, grid_h)) return 1;
    if (I->type == HAR_INTEGRATOR_PATH && pass_layout(I, grid_w, grid_h, spp, spp_pass, n_passes)) return 1;
    if (check_common(S, I, sensor, spp_pass, lb, le, C, log_spp)) return 1;
    if (!film) return fail("null film");
    hipStream_t s = (hipStream_t) stream;
    uint32_t chunk = (uint32_t) std::min<uint64_t>(I->chunk, (std::max<uint64_t>(le - lb, 2048) + 2047) / 2048 * 2048);
    if (ensure_workspace(I, chunk, false)) return 1;
    const bool multi = n_passes > 1;
    if (multi) {
        if (I->pass_rng_cap < le - lb && I->max_depth != 0) { if (ws_alloc(I, &I->pass_rng, (size_t) (le - lb))) return 1; I->pass_rng_cap = (size_t) (le - lb); }
        if (I->pass_jitter_cap < chunk) { if (ws_alloc(I, &I->pass_jitter, chunk)) return 1; I->pass_jitter_cap = chunk; }
    }
    HIP_TRY(hipMemsetAsync(I->totals, 0, 4 * sizeof(unsigned long long), s));
    HIP_TRY(hipMemsetAsync(I->status, 0, sizeof(int), s));
    I->last_stream = s;
    if (prof_begin(I, s)) return 1;
    const int mode = I->type == HAR_INTEGRATOR_PATH ? MODE_PATH : MODE_PRB_PRIMAL;
    for (uint32_t pass = 0; pass < n_passes; ++pass) {
        if (I->max_depth == 0) {      /* path.cpp:102-103: nothing but the weight channel */
            for (uint64_t base = lb; base < le; base += chunk) {
                uint32_t n = (uint32_t) std::min<uint64_t>(chunk, le - base);
                if (multi) launch_pass_jitter(s, seed, (uint32_t) base, n, pass, I->pass_jitter);
                launch_splat(s, C, seed, spp_pass, log_spp, (uint32_t) base, n, nullptr, 1, film, multi ? I->pass_jitter : nullptr);
            }
            continue;
        }
        for (uint64_t base = lb; base < le; base += chunk) {
            uint32_t n = (uint32_t) std::min<uint64_t>(chunk, le - base);
            const PassState ps{ multi ? I->pass_rng + (base - lb) : nullptr, multi ? I->pass_jitter : nullptr, pass };
            if (run_chunk(S, I, C, mode, seed, spp_pass, log_spp, (uint32_t) base, n, nullptr, s, 0, ps)) return 1;
            if (mode == MODE_PRB_PRIMAL) { launch_accumulate_stats(s, I->counters, bounce_limit(I), I->totals, n); prof_mark(I, s, CLS_OTHER); }
            launch_splat(s, C, seed, spp_pass, log_spp, (uint32_t) base, n, I->result, 0, film, ps.jitter);
            if (I->alpha_film && I->alpha_lane) launch_splat(s, C, seed, spp_pass, log_spp, (uint32_t) base, n, nullptr, 1, I->alpha_film, ps.jitter, I->alpha_lane);
            prof_mark(I, s, CLS_SPLAT);
        }
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

/* The table of texture gradient (tangent) buffers is a HOST array of the caller that only lives for the call, while the kernels read it from the device.  It is
 * staged through a ring of eight PINNED slots, so that the copy is asynchronous and the call returns without waiting for the stream (round 4 synchronised the
 * stream here, once per optimisation step: the GPU then idled while the host enqueued the whole adjoint pass).  A slot is reused only after the copy that read it
 * has run (its event); the device table itself is rewritten in stream order. */
static int upload_pointer_table(HarIntegratorImpl *I, const void *const *host, size_t n, hipStream_t s) {
    if (I->ptr_ring_cap < n) {
        (void) hipDeviceSynchronize();
        if (I->ptr_ring) (void) hipHostFree(I->ptr_ring);
        I->ptr_ring = nullptr; I->ptr_ring_cap = 0;
        HIP_TRY(hipHostMalloc((void **) &I->ptr_ring, 8 * n * sizeof(void *), hipHostMallocDefault));
        I->ptr_ring_cap = n;
        for (int k = 0; k < 8; ++k) { if (!I->ptr_ring_ev[k]) HIP_TRY(hipEventCreateWithFlags(&I->ptr_ring_ev[k], hipEventDisableTiming)); I->ptr_ring_used[k] = false; }
    }
    const uint32_t k = I->ptr_ring_next; I->ptr_ring_next = (k + 1u) & 7u;
    if (I->ptr_ring_used[k]) HIP_TRY(hipEventSynchronize(I->ptr_ring_ev[k]));
    void **slot = I->ptr_ring + (size_t) k * I->ptr_ring_cap;
    std::memcpy(slot, host, n * sizeof(void *));
    HIP_TRY(hipMemcpyAsync(I->d_grad_tex, slot, n * sizeof(void *), hipMemcpyHostToDevice, s));
    HIP_TRY(hipEventRecord(I->ptr_ring_ev[k], s)); I->ptr_ring_used[k] = true;
    return 0;
}

static int backward_range(HarScene S, HarIntegrator I, const HarSensor *sensor, const float *grad_in, const float *weight_film, uint32_t seed,
                          uint32_t spp, uint64_t lb, uint64_t le, float *grad_reflectance, float *const *grad_textures, void *stream);

/* two-stream driver (see HarIntegratorImpl::twin): returns the split point, or `le` when the job runs on one stream */
static uint64_t dual_split(HarIntegrator I, uint64_t lb, uint64_t le, hipStream_t s) {
    /* measured on MI355X (1M-triangle scene): 2 M lanes 5.84 -> 5.40 ms, 8 M 13.6 -> 13.0, 16 M 24.0 -> 23.5, 32 M +-0, 67 M 85.4 -> 86.5 ms: the two
     * launch sequences run in lock-step, so only part of the tails is hidden -- worth it for the small jobs a rank sees when N GPUs share a
     * frame, not for a single large wavefront.  HAR_STREAMS = 1 / 2 forces one / two streams. */
    static const int forced = getenv("HAR_STREAMS") ? atoi(getenv("HAR_STREAMS")) : 0;
    I->twin_used = false;
    if (forced == 1 || le - lb < HAR_DUAL_MIN_LANES || (forced != 2 && le - lb > HAR_DUAL_MAX_LANES)) return le;
    if (!I->twin) {
        if (hipStreamCreateWithFlags(&I->side_stream, hipStreamNonBlocking) != hipSuccess) return le;
        if (hipEventCreateWithFlags(&I->ev_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&I->ev_join, hipEventDisableTiming) != hipSuccess) return le;
        I->twin = new HarIntegratorImpl();
    }
    HarIntegratorImpl *T = I->twin;
    T->type = I->type; T->max_depth = I->max_depth; T->rr_depth = I->rr_depth; T->chunk = I->chunk; T->samples_per_pass = I->samples_per_pass;
    T->grad_emitters = I->grad_emitters; T->grad_bsdf_params = I->grad_bsdf_params; T->grad_light_texels = I->grad_light_texels; T->profiling = I->profiling; T->hide_emitters = I->hide_emitters;
    T->alpha_film = I->alpha_film;
    if (T->use_cache != I->use_cache) { (void) hipDeviceSynchronize(); T->free_ws(); T->use_cache = I->use_cache; }
    if (hipEventRecord(I->ev_fork, s) != hipSuccess || hipStreamWaitEvent(I->side_stream, I->ev_fork, 0) != hipSuccess) return le;
    I->twin_used = true;
    /* HAR_DUAL_FRAC (percent, default 50): the share of the first half -- an uneven cut de-synchronises the two launch sequences (A/B) */
    static const uint64_t frac = getenv("HAR_DUAL_FRAC") ? (uint64_t) std::min(90, std::max(10, atoi(getenv("HAR_DUAL_FRAC")))) : 50u;
    return lb + ((le - lb) * frac / 100 + 2047) / 2048 * 2048;
}
static int dual_join(HarIntegrator I, hipStream_t s) {
    HIP_TRY(hipEventRecord(I->ev_join, I->side_stream));
    HIP_TRY(hipStreamWaitEvent(s, I->ev_join, 0));
    return 0;
}

/* film window (har_integrator_set_film_window): the rows the lanes [lb, le) can splat into -- their pixel rows in the sample grid, moved by the sample border, widened by
 * the reconstruction filter's footprint (film_footprint, har_path.h) -- must lie inside the window; the kernels then get the address row 0 WOULD have */
static int apply_film_window(HarIntegrator I, const HarSensor *sensor, uint32_t spp, uint64_t lb, uint64_t le, float *&film) {
    if (!I->film_rows) return 0;
    DSensor C; std::string e;
    if (!lower_sensor(*sensor, C, e)) return fail(e);
    uint32_t spp_pass = spp, n_passes = 1;
    if (I->type == HAR_INTEGRATOR_PATH && pass_layout(I, C.samp_w, C.samp_h, spp, spp_pass, n_passes)) return 1;
    const uint64_t per_row = (uint64_t) C.samp_w * spp_pass;
    if (lb == 0 && le == 0) le = per_row * C.samp_h;
    if (le <= lb || per_row == 0) return 0;
    const int64_t taps = C.rfilter == 0 ? 0 : (int64_t) ceilf(C.radius - .5f);
    int64_t y0 = (int64_t) (lb / per_row) - (int64_t) C.border - taps, y1 = (int64_t) ((le - 1) / per_row) - (int64_t) C.border + taps;
    y0 = std::max<int64_t>(y0, 0); y1 = std::min<int64_t>(y1, (int64_t) C.crop_h - 1);
    if (y0 <= y1 && (y0 < (int64_t) I->film_row0 || y1 >= (int64_t) I->film_row0 + I->film_rows))
        return fail("har_integrator_set_film_window: the lanes splat into film rows [" + std::to_string(y0) + ", " + std::to_string(y1) + "], the window holds [" +
                    std::to_string(I->film_row0) + ", " + std::to_string(I->film_row0 + I->film_rows - 1) + "]");
    film -= (size_t) I->film_row0 * C.crop_w * 4;
    return 0;
}

int har_integrator_set_film_window(HarIntegrator I, uint32_t row_begin, uint32_t row_count) {
    if (!I) return fail("null integrator");
    I->film_row0 = row_count ? row_begin : 0; I->film_rows = row_count;
    return 0;
}

int har_render(HarScene S, HarIntegrator I, const HarSensor *sensor, uint32_t seed, uint32_t spp, uint64_t lb, uint64_t le, float *film, void *stream) {
    if (!S || !I || !sensor) return fail("null scene / integrator / sensor");
    if (!film) return fail("null film");
    float *const alpha_saved = I->alpha_film;
    if (I->film_rows) {
        if (apply_film_window(I, sensor, spp, lb, le, film)) return 1;
        if (I->alpha_film) { float *a = I->alpha_film; if (apply_film_window(I, sensor, spp, lb, le, a)) return 1; I->alpha_film = a; }
    }
    struct Restore { HarIntegrator I; float *a; ~Restore() { I->alpha_film = a; } } restore{ I, alpha_saved };
    uint64_t total_lb = lb, total_le = le;
    if (lb == 0 && le == 0) {                           /* "all lanes": resolve the range here so that it can be cut */
        uint32_t spp_pass = spp, n_passes = 1, grid_w = 0, grid_h = 0;
        if (sample_grid(sensor, grid_w, grid_h)) return 1;
        if (I->type == HAR_INTEGRATOR_PATH && pass_layout(I, grid_w, grid_h, spp, spp_pass, n_passes)) return 1;
        total_le = (uint64_t) grid_w * grid_h * spp_pass;
        if (total_le == 0 || total_le > 0xffffffffull) return render_range(S, I, sensor, seed, spp, lb, le, film, stream);      /* reports the error */
    }
    /* jobs small enough for the shadow-ray overlap run on one stream: that beats two half-jobs with or without overlap (overlap_applies) */
    static const int streams_env = getenv("HAR_STREAMS") ? atoi(getenv("HAR_STREAMS")) : 0;
    const bool single = streams_env != 2 && overlap_applies(S, I, std::min<uint64_t>(total_le - total_lb, I->chunk));
    const uint64_t mid = (total_le > total_lb && !single) ? dual_split(I, total_lb, total_le, (hipStream_t) stream) : total_le;
    if (mid >= total_le) { I->twin_used = false; return render_range(S, I, sensor, seed, spp, lb, le, film, stream); }
    /* HAR_DUAL_STAGGER=1 (A/B): the second half starts behind the first half's first closest-hit launch, so that one half's memory-bound shading launches
     * run next to the other half's issue-bound traversal launches instead of next to their own kind */
    static const bool stagger_env = getenv("HAR_DUAL_STAGGER") && atoi(getenv("HAR_DUAL_STAGGER")) != 0;
    if (stagger_env && !I->ev_stagger && hipEventCreateWithFlags(&I->ev_stagger, hipEventDisableTiming) != hipSuccess) I->ev_stagger = nullptr;
    I->stagger_record = stagger_env && I->ev_stagger;
    int rc = render_range(S, I, sensor, seed, spp, total_lb, mid, film, stream);
    if (stagger_env && I->ev_stagger && !I->stagger_record) (void) hipStreamWaitEvent(I->side_stream, I->ev_stagger, 0);
    I->stagger_record = false;
    rc |= render_range(S, I->twin, sensor, seed, spp, mid, total_le, film, (void *) I->side_stream);
    rc |= dual_join(I, (hipStream_t) stream);
    return rc;
}

int har_render_backward(HarScene S, HarIntegrator I, const HarSensor *sensor, const float *grad_in, const float *weight_film, uint32_t seed,
                        uint32_t spp, uint64_t lb, uint64_t le, float *grad_reflectance, float *const *grad_textures, void *stream) {
    if (!S || !I || !sensor) return fail("null scene / integrator / sensor");
    uint64_t total_lb = lb, total_le = le;
    if (lb == 0 && le == 0) {
        uint32_t grid_w = 0, grid_h = 0;
        if (sample_grid(sensor, grid_w, grid_h)) return 1;
        total_le = (uint64_t) grid_w * grid_h * spp;
        if (total_le == 0 || total_le > 0xffffffffull) return backward_range(S, I, sensor, grad_in, weight_film, seed, spp, lb, le, grad_reflectance, grad_textures, stream);
    }
    /* as in har_render: one stream when the primal pass overlaps its shadow rays (PRB bands of 8 / 16 / 33 M lanes: 16.46 / 30.19 / 54.32 ms against 17.01 / 30.90 /
     * 55.34 ms on two streams, profiles/r03_ab_shadow_overlap.txt) */
    static const int streams_env = getenv("HAR_STREAMS") ? atoi(getenv("HAR_STREAMS")) : 0;
    const bool single = streams_env != 2 && overlap_applies(S, I, std::min<uint64_t>(total_le - total_lb, I->chunk));
    const uint64_t mid = (total_le > total_lb && I->type == HAR_INTEGRATOR_PRB && !I->shape_on && !single) ? dual_split(I, total_lb, total_le, (hipStream_t) stream) : total_le;
    if (mid >= total_le) I->twin_used = false;
    if (mid >= total_le) return backward_range(S, I, sensor, grad_in, weight_film, seed, spp, lb, le, grad_reflectance, grad_textures, stream);
    int rc = backward_range(S, I, sensor, grad_in, weight_film, seed, spp, total_lb, mid, grad_reflectance, grad_textures, stream);
    rc |= backward_range(S, I->twin, sensor, grad_in, weight_film, seed, spp, mid, total_le, grad_reflectance, grad_textures, (void *) I->side_stream);
    rc |= dual_join(I, (hipStream_t) stream);
    return rc;
}

int har_integrator_sample(HarScene S, HarIntegrator I, uint32_t seed, uint32_t lane_offset, uint32_t n, const float *o, const float *d, const float *maxt,
                          const uint64_t *state, const uint8_t *active, float *rgb, uint8_t *valid, uint64_t *state_out, void *stream) {
    if (!S || !I) return fail("null scene / integrator");
    if (n == 0) return 0;
    if (!o || !d || !maxt || !rgb) return fail("null ray / output arrays");
    if ((uint64_t) lane_offset + n > 0xffffffffull) return fail("lane_offset + n exceeds the 2^32 - 1 lanes of a wavefront");
    if (state_out && I->type != HAR_INTEGRATOR_PATH) return fail("state_out: only `path` reports the sampler state after sample()");
    hipStream_t s = (hipStream_t) stream;
    const uint32_t chunk = (uint32_t) std::min<uint64_t>(I->chunk, ((uint64_t) std::max<uint32_t>(n, 2048) + 2047) / 2048 * 2048);
    if (ensure_workspace(I, chunk, false)) return 1;
    if (!I->alpha_lane && ws_alloc(I, &I->alpha_lane, I->ws_lanes)) return 1;      /* the samples' masks use the per-lane alpha array of `rgba` films */
    HIP_TRY(hipMemsetAsync(I->totals, 0, 4 * sizeof(unsigned long long), s));
    HIP_TRY(hipMemsetAsync(I->status, 0, sizeof(int), s));
    I->last_stream = s; I->twin_used = false;
    if (prof_begin(I, s)) return 1;
    const int mode = I->type == HAR_INTEGRATOR_PATH ? MODE_PATH : MODE_PRB_PRIMAL;
    const DSensor C{};                                     /* no sensor on this entry point */
    for (uint64_t base = 0; base < n; base += chunk) {
        const uint32_t m = (uint32_t) std::min<uint64_t>(chunk, n - base);
        const RaySource rays{ o, d, maxt, state, active, n, (uint32_t) base };
        if (I->max_depth == 0) {                           /* path.cpp:102-103 / prb.py: no interaction at all */
            HIP_TRY(hipMemsetAsync(I->result, 0, (size_t) m * sizeof(float4), s));
            HIP_TRY(hipMemsetAsync(I->alpha_lane, 0, (size_t) m * sizeof(float), s));
            if (state_out) { if (!state) return fail("state_out with max_depth = 0 needs `state` (the sampler is not touched, path.cpp:102-103)"); HIP_TRY(hipMemcpyAsync(state_out + base, state + base, (size_t) m * sizeof(uint64_t), hipMemcpyDeviceToDevice, s)); }
        } else {
            /* the lanes' final sampler states come back through the multi-pass mechanism (PassState::rng is indexed by lane - lane_base) */
            const PassState ps{ state_out ? state_out + base : nullptr, nullptr, 1u };
            if (run_chunk(S, I, C, mode, seed, 1, 0, lane_offset + (uint32_t) base, m, nullptr, s, 0, ps, &rays, I->alpha_lane)) return 1;
            if (mode == MODE_PRB_PRIMAL) launch_accumulate_stats(s, I->counters, bounce_limit(I), I->totals, m);
        }
        launch_sample_out(s, m, n, (uint32_t) base, I->result, I->alpha_lane, mode == MODE_PATH ? 1 : 0, rgb, valid, active, seed, lane_offset + (uint32_t) base, state, state_out);
        prof_mark(I, s, CLS_OTHER);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int har_sampler_clone(uint32_t n, const uint64_t *state, const uint64_t *inc, uint64_t *state_dst, uint64_t *inc_dst, void *stream) {
    if (n == 0) return 0;
    if (!state || !inc || !state_dst || !inc_dst) return fail("null sampler state");
    HIP_TRY(hipMemcpyAsync(state_dst, state, (size_t) n * sizeof(uint64_t), hipMemcpyDeviceToDevice, (hipStream_t) stream));
    HIP_TRY(hipMemcpyAsync(inc_dst, inc, (size_t) n * sizeof(uint64_t), hipMemcpyDeviceToDevice, (hipStream_t) stream));
    return 0;
}
int har_sampler_advance(uint32_t n, uint64_t *state, const uint64_t *inc, void *stream) {
    /* IndependentSampler::advance (src/samplers/independent.cpp:69-72 -> Sampler::advance, src/render/sampler.cpp:69-72) moves the sample index and
     * resets the dimension index; the PCG32 streams are untouched (no reseed), so the device state does not change */
    (void) n; (void) state; (void) inc; (void) stream;
    return 0;
}

int har_integrator_set_alpha_film(HarIntegrator I, float *alpha_film) {
    if (!I) return fail("null integrator");
    I->alpha_film = alpha_film;          /* the per-lane alpha values are allocated with the next render's workspace and kept */
    return 0;
}
int har_integrator_set_hide_emitters(HarIntegrator I, int hide) {
    if (!I) return fail("null integrator");
    I->hide_emitters = hide != 0;
    return 0;
}
int har_integrator_set_samples_per_pass(HarIntegrator I, uint32_t samples_per_pass) {
    if (!I) return fail("null integrator");
    if (I->type != HAR_INTEGRATOR_PATH) return fail("samples_per_pass is a property of SamplingIntegrator (`path`); the AD integrators render a single wavefront");
    I->samples_per_pass = samples_per_pass ? samples_per_pass : 0xffffffffu;
    return 0;
}
int har_render_pass_layout(HarIntegrator I, const HarSensor *sensor, uint32_t spp, uint32_t *spp_per_pass, uint32_t *n_passes) {
    if (!I || !sensor || !spp_per_pass || !n_passes) return fail("null argument");
    *spp_per_pass = spp; *n_passes = 1;
    if (I->type != HAR_INTEGRATOR_PATH) return spp ? 0 : fail("spp must be > 0");
    uint32_t grid_w = 0, grid_h = 0;
    if (sample_grid(sensor, grid_w, grid_h)) return 1;
    return pass_layout(I, grid_w, grid_h, spp, *spp_per_pass, *n_passes);
}

int har_render_weights(const HarSensor *sensor, uint32_t seed, uint32_t spp, uint64_t lb, uint64_t le, float *film, void *stream) {
    if (!sensor || !film) return fail("null sensor / film");
    DSensor C; std::string e;
    if (!lower_sensor(*sensor, C, e)) return fail(e);
    uint64_t total = (uint64_t) C.samp_w * C.samp_h * spp;
    if (spp == 0 || total > 0xffffffffull) return fail("invalid sample count");
    if (lb == 0 && le == 0) le = total;
    if (lb > le || le > total) return fail("invalid lane range");
    uint32_t log_spp = log2_exact(spp);
    const uint64_t chunk = 1u << 24;
    for (uint64_t base = lb; base < le; base += chunk) {
        uint32_t n = (uint32_t) std::min<uint64_t>(chunk, le - base);
        launch_splat((hipStream_t) stream, C, seed, spp, log_spp, (uint32_t) base, n, nullptr, 1, film);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

static int backward_range(HarScene S, HarIntegrator I, const HarSensor *sensor, const float *grad_in, const float *weight_film, uint32_t seed,
                          uint32_t spp, uint64_t lb, uint64_t le, float *grad_reflectance, float *const *grad_textures, void *stream) {
    DSensor C; uint32_t log_spp;
    if (check_common(S, I, sensor, spp, lb, le, C, log_spp)) return 1;
    if (I->type != HAR_INTEGRATOR_PRB) return fail("render_backward is implemented by the `prb` integrator");
    if (!grad_in || !weight_film || !grad_reflectance) return fail("null gradient buffers");
    if (I->max_depth == 0) return 0;
    hipStream_t s = (hipStream_t) stream;
    uint32_t chunk = (uint32_t) std::min<uint64_t>(I->chunk, (std::max<uint64_t>(le - lb, 2048) + 2047) / 2048 * 2048);
    /* replay TAPE instead of the lane-indexed replay cache (TapeArrays, har_kernels.h) whenever the adjoint commits in place: the default.  Not with
     * vertex-position / instance gradients (their adjoint goes through items), hide_emitters (its device round trip re-traces into the ping-pong
     * buffers), the replay cache switched off, or more bounces than the tape holds.  HAR_PRB_TAPE=0: the round-2 cache (A/B). */
    static const bool tape_env = getenv("HAR_PRB_TAPE") ? atoi(getenv("HAR_PRB_TAPE")) != 0 : true;
    static const bool inline_env0 = getenv("HAR_ADJOINT_INLINE") ? atoi(getenv("HAR_ADJOINT_INLINE")) != 0 : true;
    /* ... and the RECORD tape (primal pass writes one adjoint record per vertex, the adjoint pass is a streaming commit) unless alpha / eta / k gradients are
     * asked for (their fifteen extra vectors per vertex stay with the re-shading replay).  HAR_PRB_TAPE=1: the state tape (A/B) */
    static const int tape_kind_env = getenv("HAR_PRB_TAPE") ? atoi(getenv("HAR_PRB_TAPE")) : 2;
    const bool tape_ok = tape_env && inline_env0 && I->use_cache && !I->shape_on && !I->hide_emitters && bounce_limit(I) <= HAR_REPLAY_CACHE_BOUNCES;
    /* what an earlier out-of-memory step-down settled on applies to THAT job (scene, film, lane count): another job starts from the full configuration again, and the
     * same job retries it every 16th call -- the failure may have been a transient state of the allocator pool the library shares with the caller's tensors */
    const uint64_t job_key = S->serial * 0x9e3779b97f4a7c15ull ^ ((uint64_t) C.crop_w << 40) ^ ((uint64_t) C.crop_h << 20) ^ (le - lb);
    if (I->bw_job_key != job_key || (++I->bw_calls_since_stepdown & 15u) == 0u) { I->bw_tape_max = 2; I->bw_chunk_max = 0xffffffffu; I->bw_job_key = job_key; }
    chunk = std::min(chunk, I->bw_chunk_max);
    /* texels of a light's bitmap radiance: committed in place by the re-shading replay as well (the record tape holds neither the sampled uv nor the unit weight) */
    const bool light_texels = I->grad_light_texels && (S->ds.bsdf_types & HAR_SCENE_TEXLIGHT) != 0u;
    int tape = !tape_ok ? 0 : (tape_kind_env >= 2 && !I->grad_bsdf_params && !light_texels && chunk <= (1u << 29)) ? 2 : 1;
    tape = std::min(tape, I->bw_tape_max);
    /* The tapes are the large workspaces (record tape 69 B, state tape 115 B per lane and bounce against 25 B for the lane-indexed cache: 28 - 90 GB for a 2^26-lane chunk
     * at max_depth 6 - 12).  When the device -- or the host's allocator pool, shared with the caller's tensors -- cannot hold one, step down instead of failing: record
     * tape -> lane-indexed replay cache (same gradients, more traffic), then halve the chunk (more launch tails) down to 2^20 lanes. */
    const size_t npx = (size_t) C.crop_w * C.crop_h, nt = S->hs.textures.size();
    /* the kernels accumulate into (bsdf_count + emitter_count) x 3 slots; the two halves are added to the caller's buffers at the end */
    const size_t nb3 = 3 * S->hs.bsdfs.size(), ne3 = 3 * S->hs.emitters.size();
    /* everything render_backward allocates, so that the step-down below sees every refusal (the texel queues are sized by the workspace's lanes) */
    auto allocate = [&]() -> int {
        if (ensure_workspace(I, chunk, true, tape)) return 1;
        if (I->adj_floats < 3 * npx) { if (ws_alloc(I, &I->adj, 3 * npx)) return 1; I->adj_floats = 3 * npx; }
        if (I->grad_tex_cap < std::max<size_t>(nt, 1)) { if (ws_alloc(I, &I->d_grad_tex, std::max<size_t>(nt, 1))) return 1; I->grad_tex_cap = std::max<size_t>(nt, 1); }
        if (ensure_texel_queues(S, I)) return 1;
        if (I->grad_slots_cap < nb3 + ne3 + 3) { if (ws_alloc(I, &I->grad_slots, nb3 + ne3 + 3)) return 1; I->grad_slots_cap = nb3 + ne3 + 3; }
        return 0;
    };
    for (;;) {
        g_alloc_failed = false;
        if (allocate() == 0) break;
        const std::string why = g_error;
        if (!g_alloc_failed) return 1;                                         /* not an allocation failure: the error stands */
        I->bw_calls_since_stepdown = 0;
        (void) hipDeviceSynchronize(); I->free_ws(); (void) hipGetLastError();
        if (tape != 0) { tape = 0; I->bw_tape_max = 0; }
        else if (chunk > (1u << 20)) { chunk = std::max<uint32_t>(1u << 20, (chunk / 2 + 2047) / 2048 * 2048); I->bw_chunk_max = chunk; }
        else return fail("render_backward: no workspace fits the device (" + why + ")");
        static const bool verbose = getenv("HAR_VERBOSE") != nullptr;
        if (verbose) fprintf(stderr, "[hip_ad_rgb] render_backward: %s -- retrying with tape %d, chunk %u lanes\n", why.c_str(), tape, chunk);
    }
    if (nt) {
        if (!grad_textures) return fail("grad_textures is null but the scene has bitmap textures");
        for (size_t k = 0; k < nt; ++k) if (!grad_textures[k]) return fail("null texture gradient buffer");
        if (upload_pointer_table(I, (const void *const *) grad_textures, nt, s)) return 1;
    }
    if (light_texels) {
        if (!I->use_cache || I->shape_on) return fail("gradients of a light's texels need the replay cache and cannot be combined with vertex-position gradients");
        if (bounce_limit(I) > HAR_REPLAY_CACHE_BOUNCES) return fail("gradients of a light's texels: max_depth must not exceed " + std::to_string(HAR_REPLAY_CACHE_BOUNCES) + " (the cached bounces)");
        static const bool inline_env = getenv("HAR_ADJOINT_INLINE") ? atoi(getenv("HAR_ADJOINT_INLINE")) != 0 : true;
        if (!inline_env) return fail("gradients of a light's texels need the in-place commit (HAR_ADJOINT_INLINE=0 is set)");
    }
    if (I->grad_bsdf_params) {
        /* the alpha / eta / k / slot-1 terms are committed in place by the cached-bounce shading kernel only */
        if (!I->use_cache || I->shape_on) return fail("gradients of alpha / eta / k / specular colours need the replay cache and cannot be combined with vertex-position gradients");
        if (bounce_limit(I) > HAR_REPLAY_CACHE_BOUNCES) return fail("gradients of alpha / eta / k / specular colours: max_depth must not exceed " + std::to_string(HAR_REPLAY_CACHE_BOUNCES) + " (the cached bounces)");
        static const bool inline_env = getenv("HAR_ADJOINT_INLINE") ? atoi(getenv("HAR_ADJOINT_INLINE")) != 0 : true;
        if (!inline_env) return fail("gradients of alpha / eta / k / specular colours need the in-place commit (HAR_ADJOINT_INLINE=0 is set)");
    }
    HIP_TRY(hipMemsetAsync(I->grad_slots, 0, (nb3 + ne3 + 3) * sizeof(float), s));
    if (I->shape_on) {
        if ((I->pos_verts && I->pos_offset.size() != S->hs.meshes.size()) || (I->inst_count && I->inst_count != S->hs.insts.size()))
            return fail("har_integrator_set_grad_positions / har_integrator_set_grad_instances was called for a different scene");
        if (I->pos_verts) HIP_TRY(hipMemsetAsync(I->grad_pos, 0, (size_t) 3 * I->pos_verts * sizeof(float), s));
        if (I->pos_verts && I->grad_nrm) HIP_TRY(hipMemsetAsync(I->grad_nrm, 0, (size_t) 3 * I->pos_verts * sizeof(float), s));
        if (I->inst_count) HIP_TRY(hipMemsetAsync(I->grad_inst, 0, (size_t) 12 * I->inst_count * sizeof(float), s));
    }
    HIP_TRY(hipMemsetAsync(I->totals, 0, 4 * sizeof(unsigned long long), s));
    HIP_TRY(hipMemsetAsync(I->status, 0, sizeof(int), s));
    I->last_stream = s;
    if (prof_begin(I, s)) return 1;
    launch_adjoint_image(s, grad_in, weight_film, (uint32_t) npx, I->adj);
    prof_mark(I, s, CLS_OTHER);
    for (uint64_t base = lb; base < le; base += chunk) {
        uint32_t n = (uint32_t) std::min<uint64_t>(chunk, le - base);
        /* pass 1: primal, keeps L per lane in `result` (common.py:752-762) */
        if (run_chunk(S, I, C, MODE_PRB_PRIMAL, seed, spp, log_spp, (uint32_t) base, n, I->grad_slots, s, tape == 2 ? 5 : tape == 1 ? 3 : I->cache_bounces ? 1 : 0)) return 1;
        /* pass 2: adjoint replay with the identical sample stream (common.py:765-775) */
        if (run_chunk(S, I, C, MODE_PRB_ADJOINT, seed, spp, log_spp, (uint32_t) base, n, I->grad_slots, s, tape == 2 ? 6 : tape == 1 ? 4 : I->cache_bounces ? 2 : 0)) return 1;
    }
    launch_add(s, I->grad_slots, grad_reflectance, (uint32_t) nb3);
    if (I->grad_emitters && ne3) launch_add(s, I->grad_slots + nb3, I->grad_emitters, (uint32_t) ne3);
    if (I->shape_on)
        for (size_t m = 0; m < I->pos_user.size(); ++m) {
            if (!I->pos_user[m]) continue;
            /* meshes with vertex normals: the summed adjoints of the vertex normals go through compute_normals once per face (second stage) */
            if (I->grad_nrm && (S->hs.meshes[m].flags & 1u))
                launch_normals_adjoint(s, S->ds, (uint32_t) m, S->hs.meshes[m].face_count, S->hs.meshes[m].vertex_count, I->nrm_acc + 3 * (size_t) I->pos_offset[m],
                                       I->grad_nrm + 3 * (size_t) I->pos_offset[m], I->grad_pos + 3 * (size_t) I->pos_offset[m]);
            launch_add(s, I->grad_pos + 3 * (size_t) I->pos_offset[m], I->pos_user[m], 3 * I->pos_count[m]);
        }
    if (I->shape_on && I->inst_count) launch_add(s, I->grad_inst, I->inst_user, 12 * I->inst_count);
    HIP_TRY(hipGetLastError());
    return 0;
}

/* RBIntegrator.render_forward (src/python/python/ad/integrators/common.py:497-623): primal pass (L per lane), then the differential pass in
 * FORWARD mode -- the adjoint kernels read the parameters' tangents where render_backward accumulates gradients, and every lane sums
 * <d Lo / d theta, tangent> over its vertices (prb.py:313) -- then the lanes' differential radiance is splatted like an image. */
int har_render_forward(HarScene S, HarIntegrator I, const HarSensor *sensor, uint32_t seed, uint32_t spp, uint64_t lb, uint64_t le,
                       const float *tangent_reflectance, const float *const *tangent_textures, const float *tangent_emitters, float *film, void *stream) {
    DSensor C; uint32_t log_spp;
    if (check_common(S, I, sensor, spp, lb, le, C, log_spp)) return 1;
    if (I->type != HAR_INTEGRATOR_PRB) return fail("render_forward is implemented by the `prb` integrator");
    if (!film || !tangent_reflectance) return fail("null film / tangent buffers");
    if (I->shape_on) return fail("render_forward: tangents of vertex positions are not implemented (use render_backward for shape gradients)");
    hipStream_t s = (hipStream_t) stream;
    if (I->max_depth == 0) {        /* no interaction, no derivative: a zero image with the filter weights */
        return har_render_weights(sensor, seed, spp, lb, le, film, stream);
    }
    uint32_t chunk = (uint32_t) std::min<uint64_t>(I->chunk, (std::max<uint64_t>(le - lb, 2048) + 2047) / 2048 * 2048);
    if (ensure_workspace(I, chunk, true)) return 1;
    const size_t nt = S->hs.textures.size();
    if (I->grad_tex_cap < std::max<size_t>(nt, 1)) { if (ws_alloc(I, &I->d_grad_tex, std::max<size_t>(nt, 1))) return 1; I->grad_tex_cap = std::max<size_t>(nt, 1); }
    if (nt) {
        if (!tangent_textures) return fail("tangent_textures is null but the scene has bitmap textures");
        for (size_t k = 0; k < nt; ++k) if (!tangent_textures[k]) return fail("null texture tangent buffer");
        if (upload_pointer_table(I, (const void *const *) tangent_textures, nt, s)) return 1;
    }
    /* tangent slots in the layout of the gradient slots: (bsdf_count + emitter_count) x 3 */
    const size_t nb3 = 3 * S->hs.bsdfs.size(), ne3 = 3 * S->hs.emitters.size();
    if (I->grad_slots_cap < nb3 + ne3 + 3) { if (ws_alloc(I, &I->grad_slots, nb3 + ne3 + 3)) return 1; I->grad_slots_cap = nb3 + ne3 + 3; }
    HIP_TRY(hipMemsetAsync(I->grad_slots, 0, (nb3 + ne3 + 3) * sizeof(float), s));
    if (nb3) HIP_TRY(hipMemcpyAsync(I->grad_slots, tangent_reflectance, nb3 * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (tangent_emitters && ne3) HIP_TRY(hipMemcpyAsync(I->grad_slots + nb3, tangent_emitters, ne3 * sizeof(float), hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemsetAsync(I->totals, 0, 4 * sizeof(unsigned long long), s));
    HIP_TRY(hipMemsetAsync(I->status, 0, sizeof(int), s));
    I->last_stream = s; I->twin_used = false;
    if (prof_begin(I, s)) return 1;
    float *saved_emitters = I->grad_emitters;
    I->grad_emitters = tangent_emitters ? I->grad_slots + nb3 : nullptr;       /* only its non-null-ness is read (HAR_SHADE_EMITTER_GRADS) */
    int rc = 0;
    for (uint64_t base = lb; base < le && !rc; base += chunk) {
        uint32_t n = (uint32_t) std::min<uint64_t>(chunk, le - base);
        rc = run_chunk(S, I, C, MODE_PRB_PRIMAL, seed, spp, log_spp, (uint32_t) base, n, nullptr, s, I->cache_bounces ? 1 : 0);
        I->forward_mode = true;
        if (!rc) rc = run_chunk(S, I, C, MODE_PRB_ADJOINT, seed, spp, log_spp, (uint32_t) base, n, I->grad_slots, s, I->cache_bounces ? 2 : 0);
        I->forward_mode = false;
        if (!rc) { launch_splat(s, C, seed, spp, log_spp, (uint32_t) base, n, I->dL, 0, film); prof_mark(I, s, CLS_SPLAT); }
    }
    I->grad_emitters = saved_emitters;
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return 0;
}

/* BSDFs with only delta lobes on MOVING geometry: bsdf.eval() is zero there, so prb.py:288 forms relative_grad(0) -- a 0 / 0 whose value the tree does not
 * pin -- and the sampled direction is a reflection / refraction of wi, not a direction the solid-angle-to-area Jacobian could hold fixed.  Refused. */
static bool record_has_smooth_lobe(const HostScene &hs, int32_t index) {
    if (index < 0 || (size_t) index >= hs.bsdfs.size()) return false;
    const DBsdf &b = hs.bsdfs[(size_t) index];
    return bsdf_is_smooth(b) && (b.back < 0 || bsdf_is_smooth(hs.bsdfs[(size_t) b.back]));
}

int har_integrator_set_grad_positions(HarIntegrator I, HarScene S, float *const *grad_positions) {
    if (!I) return fail("null integrator");
    if (I->type != HAR_INTEGRATOR_PRB) return fail("vertex-position gradients are computed by the `prb` integrator");
    std::vector<float *> user; std::vector<int32_t> offset; std::vector<uint32_t> count; uint32_t verts = 0; bool smooth = false;
    if (grad_positions) {
        if (!S) return fail("null scene");
        /* the hand-derived adjoint of har_shape_grad.h covers top-level meshes, flat-shaded or with (regenerated) vertex normals, carrying any BSDF with a non-delta lobe (the directional derivatives
         * of the models come from har_bsdf_dir.h); the rest of the scene may carry any model -- a vertex next to moving geometry contributes through its
         * attached si.wi (prb.py:128-140) */
        const size_t nm = S->hs.meshes.size();
        offset.assign(nm, -1); user.assign(nm, nullptr); count.assign(nm, 0);
        for (size_t m = 0; m < nm; ++m) {          /* top-level meshes, then the meshes of the shape groups (vertex positions shared by all their instances) */
            if (!grad_positions[m]) continue;
            const DMesh &M = S->hs.meshes[m];
            if (m >= S->hs.top_mesh_count && I->inst_count) return fail("Cannot differentiate instance parameters and shapegroup internal parameters at the same time!");      /* instance.cpp:162-166 */
            const bool known = (I->pos_checked_scene == S->serial && m < I->pos_checked.size() && I->pos_checked[m]) ||      /* an optimisation loop calls this every step */
                               (m < S->normals_regenerated.size() && S->normals_regenerated[m]);                             /* k_vertex_normals wrote them (har_scene_update_vertices_device) */
            if ((M.flags & 1u) && known) smooth = true;
            else if (M.flags & 1u) {
                if (refresh_host_vertices(S, nullptr, (int) m)) return 1;
                /* a position update regenerates the vertex normals (mesh.cpp:876-878 -> compute_normals): the gradient is that of the REGENERATED normals, so the
                 * mesh must carry them -- stored normals of another origin (file, analytic) would render one surface and differentiate another */
                std::vector<float> copy(S->hs.verts.begin() + 8 * (size_t) M.voff, S->hs.verts.begin() + 8 * (size_t) (M.voff + M.vertex_count));
                if (har_mesh_compute_normals(M.vertex_count, copy.data(), M.face_count, S->hs.faces.data() + 4 * (size_t) M.foff)) return 1;
                float worst = 0.f;
                for (size_t v = 0; v < M.vertex_count; ++v) for (int c = 3; c < 6; ++c) worst = std::max(worst, std::fabs(copy[8 * v + c] - S->hs.verts[8 * ((size_t) M.voff + v) + c]));
                if (worst > 1e-4f) return fail("vertex-position gradients of a mesh with vertex normals: its normals are not the ones a position update regenerates (Mesh::compute_normals, mesh.cpp:876-878); write the positions once (params.update()) or regenerate the normals first");
                smooth = true;
                if (I->pos_checked_scene != S->serial) { I->pos_checked_scene = S->serial; I->pos_checked.assign(nm, 0); }
                I->pos_checked[m] = 1;
            }
            if (!record_has_smooth_lobe(S->hs, M.bsdf)) return fail("vertex-position gradients: a differentiated mesh cannot carry a BSDF made of delta lobes only (`dielectric`, `conductor`); other meshes of the scene may");
            offset[m] = (int32_t) verts; user[m] = grad_positions[m]; count[m] = M.vertex_count; verts += M.vertex_count;
        }
        if (verts == 0) { offset.clear(); user.clear(); count.clear(); }
    }
    if (offset != I->pos_offset || verts != I->pos_verts || smooth != I->pos_smooth) {      /* the geometry records and the offset table are part of the adjoint workspace */
        (void) hipDeviceSynchronize();
        I->free_ws();
    }
    I->pos_user = user; I->pos_offset = offset; I->pos_count = count; I->pos_verts = verts; I->pos_smooth = smooth; I->shape_on = verts != 0 || I->inst_count != 0;
    return 0;
}

int har_integrator_set_grad_instances(HarIntegrator I, HarScene S, float *grad_to_world) {
    if (!I) return fail("null integrator");
    if (I->type != HAR_INTEGRATOR_PRB) return fail("instance to_world gradients are computed by the `prb` integrator");
    uint32_t n = 0;
    if (grad_to_world) {
        if (!S) return fail("null scene");
        /* as for the vertex positions: any BSDF with a non-delta lobe on the moving geometry, i.e. on the meshes of the shape groups */
        for (size_t m = S->hs.top_mesh_count; m < S->hs.meshes.size(); ++m)
            if (!record_has_smooth_lobe(S->hs, S->hs.meshes[m].bsdf)) return fail("instance to_world gradients: an instanced mesh cannot carry a BSDF made of delta lobes only (`dielectric`, `conductor`); top-level meshes may");
        n = (uint32_t) S->hs.insts.size();
        if (n == 0) return fail("the scene has no instances");
        for (size_t m = S->hs.top_mesh_count; m < I->pos_offset.size(); ++m)
            if (I->pos_offset[m] >= 0) return fail("Cannot differentiate instance parameters and shapegroup internal parameters at the same time!");      /* instance.cpp:162-166 */
        if (n >= (1u << (32 - HAR_SHAPE_INST_SHIFT)) - 1u) return fail("too many instances for the adjoint's geometry records");
    }
    if (n != I->inst_count) { (void) hipDeviceSynchronize(); I->free_ws(); }      /* the geometry records and the slot table are part of the adjoint workspace */
    I->inst_user = n ? grad_to_world : nullptr; I->inst_count = n; I->shape_on = I->pos_verts != 0 || n != 0;
    return 0;
}

int har_integrator_set_grad_bsdf_params(HarIntegrator I, float *grad) {
    if (!I) return fail("null integrator");
    if (I->type != HAR_INTEGRATOR_PRB) return fail("BSDF parameter gradients are computed by the `prb` integrator");
    I->grad_bsdf_params = grad;
    return 0;
}

int har_integrator_set_grad_light_texels(HarIntegrator I, int on) {
    if (!I) return fail("null integrator");
    if (I->type != HAR_INTEGRATOR_PRB) return fail("gradients of a light's texels are computed by the `prb` integrator");
    I->grad_light_texels = on != 0;
    return 0;
}

int har_integrator_set_grad_emitters(HarIntegrator I, float *grad_emitters) {
    if (!I) return fail("null integrator");
    if (I->type != HAR_INTEGRATOR_PRB) return fail("emitter gradients are computed by the `prb` integrator");
    I->grad_emitters = grad_emitters;
    return 0;
}

int har_render_stats(HarIntegrator I, HarStats *out) {
    if (!I || !out) return fail("null argument");
    memset(out, 0, sizeof(*out));
    if (!I->totals) return 0;
    hipStream_t s = I->last_stream;
    unsigned long long t[4] = { 0, 0, 0, 0 };
    HIP_TRY(hipMemcpyAsync(t, I->totals, sizeof(t), hipMemcpyDeviceToHost, s));
    if (read_status(I->status, s)) return 1;
    out->paths = t[0]; out->vertices = t[1]; out->closest_rays = t[2]; out->shadow_rays = t[3];
    if (I->twin_used && I->twin && I->twin->totals) {      /* the caller's stream has joined the side stream: one more copy on it sees the twin's counters */
        HIP_TRY(hipMemcpyAsync(t, I->twin->totals, sizeof(t), hipMemcpyDeviceToHost, s));
        if (read_status(I->twin->status, s)) return 1;
        out->paths += t[0]; out->vertices += t[1]; out->closest_rays += t[2]; out->shadow_rays += t[3];
    }
    return 0;
}

int har_integrator_set_packet_tracing(HarIntegrator I, int mode) {
    if (!I) return fail("null integrator");
    if (mode < -1 || mode > 1) return fail("har_integrator_set_packet_tracing: mode must be -1 (automatic), 0 (off) or 1 (on)");
    I->packet_tracing = mode;
    if (I->twin) I->twin->packet_tracing = mode;
    return 0;
}

int har_integrator_set_material_queues(HarIntegrator I, int enable) {
    if (!I) return fail("null integrator");
    I->material_queues = enable != 0;
    if (I->twin) I->twin->material_queues = I->material_queues;
    return 0;
}

int har_integrator_set_replay_cache(HarIntegrator I, int enable) {
    if (!I) return fail("null integrator");
    if (I->use_cache != (enable != 0)) { (void) hipDeviceSynchronize(); I->free_ws(); I->use_cache = enable != 0; }
    return 0;
}

int har_integrator_set_profiling(HarIntegrator I, int enable) {
    if (!I) return fail("null integrator");
    /* (re)start: frames enqueued so far are dropped from the statistics (their events complete on their own and are reused later) */
    for (HarIntegratorImpl *J : { I, I->twin }) {
        if (!J) continue;
        J->profiling = enable != 0;
        if (!J->sets.empty()) { (void) hipDeviceSynchronize(); for (auto &E : J->sets) E.used = 0; }
        for (int k = 0; k < 8; ++k) { J->acc_ms[k] = 0.0; J->acc_launches[k] = 0; }
        J->acc_frames = 0;
    }
    return 0;
}

static int add_timing(HarIntegratorImpl *I, double ms[8], double launches[8], uint64_t &frames) {
    for (auto &E : I->sets) if (prof_collect(I, E)) return 1;
    for (int k = 0; k < 8; ++k) { ms[k] += I->acc_ms[k]; launches[k] += (double) I->acc_launches[k]; }
    frames = std::max(frames, I->acc_frames);
    return 0;
}
/* AVERAGE per frame over the frames rendered since har_integrator_set_profiling(1): ms[c] = device time between consecutive events of class c,
 * launches[c] = launches of class c per frame (rounded); ms[7] = number of frames averaged over.  In two-stream mode the launches of both
 * halves are summed: kernels of the two streams overlap, so the class totals exceed the wall time. */
int har_render_timing(HarIntegrator I, float ms[8], uint32_t launches[8]) {
    if (!I) return fail("null integrator");
    double m[8] = { 0 }, l[8] = { 0 }; uint64_t frames = 0;
    if (add_timing(I, m, l, frames)) return 1;
    if (I->twin && add_timing(I->twin, m, l, frames)) return 1;
    const double f = frames ? (double) frames : 1.0;
    for (int k = 0; k < 8; ++k) { ms[k] = (float) (m[k] / f); launches[k] = (uint32_t) (l[k] / f + 0.5); }
    ms[7] = (float) frames;
    return 0;
}

} // extern "C"
