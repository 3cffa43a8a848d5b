/*
 * har_multi.hip -- ONE host thread, N GPUs: the multi-GPU partitioning of SURVEY.md section 8(e) behind the C ABI.
 *
 * The reference's contract is one Integrator::render call from one host thread (include/mitsuba/render/integrator.h:74-79); it has no multi-GPU path, so the
 * partitioning is this repository's design (mitsuba3_amd/distributed.py is the same design as one PROCESS per GPU under torch.distributed -- what bench.py --gpus N
 * runs; this file is the route a C++ host takes, INTEGRATION.md Route A).  Per device: a replica of the scene (its own BVH, textures, records), an integrator with
 * its own workspace, a stream.  A frame:
 *   1. the sample grid's pixel rows are dealt to the devices as contiguous BANDS with all their samples; a device renders the lanes of its band with the GLOBAL lane
 *      index (har_render's lane range), so the union of the bands draws exactly the samples of a single-GPU render;
 *   2. every device splats into a private full-size film; ONE collective adds them on device 0: ncclReduce inside one ncclGroupStart / End over the communicators
 *      of ncclCommInitAll (RCCL, resolved at run time -- librccl is looked up only when a group of more than one DISTINCT device is created; a group that names one
 *      physical device several times, and a process without RCCL, add the films with peer copies + one add kernel per film instead);
 *   3. device 0 develops the film (har_film_develop_format) on the caller's stream.
 * Bands are re-cut from the measured device times of the previous frames (HIP events per device, read one frame late: nothing waits), like distributed.py's
 * BandBalancer, and frozen after HAR_MULTI_ADAPT_FRAMES frames.
 * Nothing here synchronises the host with a device: the call returns when everything is enqueued, the image is valid in `stream` order on device 0.
 */
#include "../../include/hip_ad_rgb.h"
#include "har_kernels.h"
#include "har_scene_host.h"

#include <dlfcn.h>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <string>
#include <vector>

extern int har_set_error(const std::string &msg);

namespace {

using namespace har;

#define MULTI_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return har_set_error(std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

/* the six RCCL entry points the reduce needs, looked up at run time (rccl.h:236,448-466,550: ncclCommInitAll, ncclReduce, ncclFloat = 7, ncclSum = 0) */
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(void **comms, int ndev, const int *devlist) = nullptr;
    int (*CommDestroy)(void *comm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Reduce)(const void *send, void *recv, size_t count, int datatype, int op, int root, void *comm, hipStream_t stream) = nullptr;
    int (*AllReduce)(const void *send, void *recv, size_t count, int datatype, int op, void *comm, hipStream_t stream) = nullptr;
    int (*Broadcast)(const void *send, void *recv, size_t count, int datatype, int root, void *comm, hipStream_t stream) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool load(std::string &why) {
        if (lib) return true;
        const char *names[] = { "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1" };
        lib = dlopen(names[0], RTLD_NOW | RTLD_NOLOAD);                         /* a copy the process already holds (PyTorch's) comes first: two RCCLs in one process clash */
        for (int k = 0; !lib && k < 3; ++k) lib = dlopen(names[k], RTLD_NOW | RTLD_LOCAL);
        if (!lib) { why = std::string("librccl not found: ") + dlerror(); return false; }
        CommInitAll = (decltype(CommInitAll)) dlsym(lib, "ncclCommInitAll"); CommDestroy = (decltype(CommDestroy)) dlsym(lib, "ncclCommDestroy");
        GroupStart = (decltype(GroupStart)) dlsym(lib, "ncclGroupStart"); GroupEnd = (decltype(GroupEnd)) dlsym(lib, "ncclGroupEnd");
        Reduce = (decltype(Reduce)) dlsym(lib, "ncclReduce"); GetErrorString = (decltype(GetErrorString)) dlsym(lib, "ncclGetErrorString");
        AllReduce = (decltype(AllReduce)) dlsym(lib, "ncclAllReduce"); Broadcast = (decltype(Broadcast)) dlsym(lib, "ncclBroadcast");
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Reduce || !AllReduce || !Broadcast) { why = "librccl lacks ncclCommInitAll / ncclReduce / ncclGroupStart"; lib = nullptr; return false; }
        return true;
    }
};
Rccl g_rccl;

struct Replica {
    int device = 0;
    HarScene scene = nullptr; HarIntegrator integ = nullptr;
    hipStream_t stream = nullptr;            /* private stream (device 0 renders on the caller's stream) */
    float *film = nullptr, *staging = nullptr;   /* H x W x 4; staging: on device 0, the peer copy of this replica's film (copy reduce) */
    hipEvent_t done = nullptr;
    hipEvent_t t0[2][5] = {}, t1[2][5] = {};   /* [0 = render, 1 = render_backward][slot of the BandState ring]: this replica's band, start / end */
    void *comm = nullptr;
    /* render_backward: the band's weight film (H x W x 4, then the complete one), the adjoint image, ONE flat gradient buffer {bsdf slots, emitter slots, texture 0, 1, ...} and
     * its staging copy on device 0 (copy reduce), the texture pointer table into the flat buffer, this band's events */
    float *wfilm = nullptr, *grad_in = nullptr, *grads = nullptr, *grads_staging = nullptr;
    std::vector<float *> tex_ptrs;
    hipEvent_t bdone = nullptr, wdone = nullptr;
};

} // namespace

struct HarMultiImpl {
    std::vector<Replica> rep;
    bool use_rccl = false; std::string reduce_note;
    hipEvent_t start = nullptr;              /* recorded on the caller's stream: the other devices' streams begin after it */
    size_t film_floats = 0;
    /* bands (rows of the sample grid), distributed.py BandBalancer */
    /* A host loop runs AHEAD of the devices: when frame f + 1 is enqueued, frame f has usually not finished, so "the previous frame's times" do not exist yet.  Every
     * frame therefore records its bands' events into one of four slots together with the bounds it used; a call looks, newest first, for a slot whose events have all
     * completed and re-cuts the bands from THAT frame's bounds and times (nothing waits; a synchronous caller simply finds the last frame).  After `adapt_frames`
     * re-cuts the bands are frozen. */
    struct BandState {
        std::vector<uint32_t> bounds; uint32_t rows = 0, frames = 0; std::vector<float> last_ms;
        struct Slot { std::vector<uint32_t> bounds; bool valid = false; } slot[4]; uint32_t next = 0;
        uint32_t skip = 0;                     /* frames not to measure: the first frame on a new sample grid pays workspace allocation and code-object loads on every device */
    } bands[2];                                /* 0 = render, 1 = render_backward (the adjoint's cost profile is not the forward render's) */
    uint32_t adapt_frames = 3;
    /* render_backward: its own bands (the adjoint's cost profile differs from the forward render's), the layout of the flat gradient buffer, events of device 0 */
    uint32_t bsdf_count = 0, emitter_count = 0; std::vector<size_t> tex_floats; size_t grad_floats = 0, bfilm_floats = 0, bimg_floats = 0;
    hipEvent_t wsum = nullptr, gin = nullptr;
};

namespace {

void cut_bands(HarMultiImpl *M, int kind, uint32_t rows) {
    const uint32_t n = (uint32_t) M->rep.size();
    HarMultiImpl::BandState &B = M->bands[kind];
    B.rows = rows; B.frames = 0; B.bounds.resize(n + 1);
    for (uint32_t r = 0; r <= n; ++r) B.bounds[r] = (uint32_t) ((uint64_t) rows * r / n);
    for (auto &sl : B.slot) sl.valid = false;
    B.skip = 1;
}
/* BandBalancer.update (mitsuba3_amd/distributed.py): boundaries that equalise the integral of the piecewise-constant cost per row measured on the last frame;
 * false = the bands stay (a time that is not positive, fewer rows than devices) */
bool rebalance_bounds(uint32_t rows, const std::vector<uint32_t> &b, const std::vector<double> &ms, std::vector<uint32_t> &out) {
    const uint32_t n = (uint32_t) ms.size();
    if (n < 2 || rows < n || b.size() != n + 1) return false;
    for (double t : ms) if (!(t > 0.0)) return false;
    std::vector<double> dens(n); double total = 0.0;
    for (uint32_t r = 0; r < n; ++r) { dens[r] = ms[r] / std::max<uint32_t>(b[r + 1] - b[r], 1u); total += ms[r]; }
    std::vector<uint32_t> nb{ 0u }; uint32_t r = 0; double acc = 0.0;
    for (uint32_t k = 1; k < n; ++k) {
        const double target = total * k / n;
        while (r < n - 1 && acc + ms[r] < target) { acc += ms[r]; ++r; }
        const double y = dens[r] > 0.0 ? b[r] + (target - acc) / dens[r] : b[r + 1];
        const long yi = (long) std::nearbyint(y);                                                /* Python's round(): half to even */
        nb.push_back((uint32_t) std::min<long>(std::max<long>(yi, (long) nb.back() + 1), (long) rows - (long) (n - k)));      /* every device keeps at least one row */
    }
    nb.push_back(rows);
    out = nb;
    return true;
}
/* the bands of this call: (re)cut for a new sample grid, else re-cut from the newest measured frame that has finished; returns the ring slot this call records into */
uint32_t begin_bands(HarMultiImpl *M, int kind, uint32_t rows) {
    const uint32_t n = (uint32_t) M->rep.size();
    HarMultiImpl::BandState &B = M->bands[kind];
    if (B.rows != rows || B.bounds.size() != n + 1) cut_bands(M, kind, rows);
    else if (n > 1 && B.frames < M->adapt_frames) {
        for (uint32_t age = 1; age <= 4; ++age) {
            const uint32_t idx = (B.next + 4u - age) & 3u;
            if (!B.slot[idx].valid) continue;
            std::vector<float> ms(n, 0.f); bool ready = true;
            for (uint32_t k = 0; k < n && ready; ++k) {
                Replica &R = M->rep[k];
                (void) hipSetDevice(R.device);
                ready = hipEventQuery(R.t1[kind][idx]) == hipSuccess && hipEventElapsedTime(&ms[k], R.t0[kind][idx], R.t1[kind][idx]) == hipSuccess;
            }
            (void) hipGetLastError();
            if (!ready) continue;                                  /* still in flight: try an older frame */
            std::vector<double> t(ms.begin(), ms.end()); std::vector<uint32_t> nb;
            if (rebalance_bounds(B.rows, B.slot[idx].bounds, t, nb)) B.bounds = nb;
            B.last_ms = ms; B.frames++;
            for (auto &sl : B.slot) sl.valid = false;              /* frames measured with older bounds are obsolete */
            break;
        }
    }
    /* this frame's slot: a slot whose frame is still in flight is NOT recycled (a host far ahead of the devices would otherwise overwrite every measurement before it
     * completes): when all four are pending the frame records into the spare slot 4, which is never read */
    uint32_t idx = 4u;
    const bool measure = n > 1 && B.skip == 0 && B.frames < M->adapt_frames;
    if (B.skip) --B.skip;
    if (measure) for (uint32_t q = 0; q < 4u; ++q) { const uint32_t c = (B.next + q) & 3u; if (!B.slot[c].valid) { idx = c; break; } }
    if (idx < 4u) { B.next = (idx + 1u) & 3u; B.slot[idx].bounds = B.bounds; B.slot[idx].valid = true; }
    return idx;
}

int destroy(HarMultiImpl *M) {
    if (!M) return 0;
    for (Replica &R : M->rep) {
        (void) hipSetDevice(R.device);
        (void) hipDeviceSynchronize();
        if (R.comm && g_rccl.CommDestroy) (void) g_rccl.CommDestroy(R.comm);
        if (R.integ) (void) har_integrator_destroy(R.integ);
        if (R.scene) (void) har_scene_destroy(R.scene);
        if (R.film) (void) hipFree(R.film);
        if (R.wfilm) (void) hipFree(R.wfilm); if (R.grad_in) (void) hipFree(R.grad_in); if (R.grads) (void) hipFree(R.grads);
        for (hipEvent_t ev : { R.bdone, R.wdone, R.done }) if (ev) (void) hipEventDestroy(ev);
        for (int kind = 0; kind < 2; ++kind) for (int q = 0; q < 5; ++q) { if (R.t0[kind][q]) (void) hipEventDestroy(R.t0[kind][q]); if (R.t1[kind][q]) (void) hipEventDestroy(R.t1[kind][q]); }
        if (R.stream) (void) hipStreamDestroy(R.stream);
    }
    if (!M->rep.empty()) {
        (void) hipSetDevice(M->rep[0].device);
        for (Replica &R : M->rep) { if (R.staging) (void) hipFree(R.staging); if (R.grads_staging) (void) hipFree(R.grads_staging); }
        if (M->start) (void) hipEventDestroy(M->start);
        if (M->wsum) (void) hipEventDestroy(M->wsum); if (M->gin) (void) hipEventDestroy(M->gin);
    }
    delete M;
    return 0;
}

} // namespace

extern "C" {

int har_multi_create(const HarSceneDesc *desc, int integrator_type, int32_t max_depth, int32_t rr_depth, uint32_t chunk_lanes, const int *devices, uint32_t n_devices,
                     HarMulti *out) {
    if (!desc || !out || !devices || n_devices == 0) return har_set_error("har_multi_create: null argument / no device");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return har_set_error("hip_ad_rgb requires a HIP device (no CPU fallback)");
    bool distinct = true;
    for (uint32_t a = 0; a < n_devices; ++a) {
        if (devices[a] < 0 || devices[a] >= ndev) return har_set_error("har_multi_create: device " + std::to_string(devices[a]) + " does not exist (" + std::to_string(ndev) + " visible)");
        for (uint32_t b = 0; b < a; ++b) distinct = distinct && devices[a] != devices[b];
    }
    int caller_device = 0; (void) hipGetDevice(&caller_device);
    HarMultiImpl *M = new HarMultiImpl();
    M->rep.resize(n_devices);
    M->bsdf_count = desc->bsdf_count; M->emitter_count = desc->emitter_count;
    for (uint32_t t = 0; t < desc->texture_count; ++t) M->tex_floats.push_back((size_t) desc->textures[t].width * desc->textures[t].height * 3);
    M->grad_floats = 3 * ((size_t) M->bsdf_count + M->emitter_count);
    for (size_t f : M->tex_floats) M->grad_floats += f;
    if (getenv("HAR_MULTI_ADAPT_FRAMES")) M->adapt_frames = (uint32_t) atoi(getenv("HAR_MULTI_ADAPT_FRAMES"));
    int rc = 0;
    for (uint32_t k = 0; k < n_devices && !rc; ++k) {
        Replica &R = M->rep[k]; R.device = devices[k];
        if (hipSetDevice(R.device) != hipSuccess) { rc = har_set_error("hipSetDevice failed"); break; }
        rc = har_scene_create(desc, &R.scene);                                   /* the replica: BVH build + upload on THIS device */
        if (!rc) rc = har_integrator_create(integrator_type, max_depth, rr_depth, chunk_lanes, &R.integ);
        if (!rc && k > 0 && hipStreamCreateWithFlags(&R.stream, hipStreamNonBlocking) != hipSuccess) rc = har_set_error("hipStreamCreate failed");
        if (!rc && hipEventCreateWithFlags(&R.done, hipEventDisableTiming) != hipSuccess) rc = har_set_error("hipEventCreate failed");
        for (int kind = 0; kind < 2 && !rc; ++kind) for (int q = 0; q < 5 && !rc; ++q)
            if (hipEventCreate(&R.t0[kind][q]) != hipSuccess || hipEventCreate(&R.t1[kind][q]) != hipSuccess) rc = har_set_error("hipEventCreate failed");
    }
    if (!rc) { (void) hipSetDevice(devices[0]); if (hipEventCreateWithFlags(&M->start, hipEventDisableTiming) != hipSuccess) rc = har_set_error("hipEventCreate failed"); }
    /* the collective: RCCL for a group of distinct devices; peer copies + adds otherwise (one physical device named several times cannot form a communicator) */
    const char *force = getenv("HAR_MULTI_REDUCE");
    if (!rc && n_devices > 1) {
        std::string why;
        if (force && std::string(force) == "copy") M->reduce_note = "peer copies + add (HAR_MULTI_REDUCE=copy)";
        else if (!distinct) M->reduce_note = "peer copies + add (a device is named more than once: no communicator)";
        else if (!g_rccl.load(why)) M->reduce_note = "peer copies + add (" + why + ")";
        else {
            std::vector<void *> comms(n_devices, nullptr);
            const int e = g_rccl.CommInitAll(comms.data(), (int) n_devices, devices);
            if (e != 0) M->reduce_note = std::string("peer copies + add (ncclCommInitAll: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "error") + ")";
            else { for (uint32_t k = 0; k < n_devices; ++k) M->rep[k].comm = comms[k]; M->use_rccl = true; M->reduce_note = "ncclReduce (RCCL), one group call"; }
        }
    } else if (!rc) M->reduce_note = "single device: no collective";
    (void) hipSetDevice(caller_device);
    if (rc) { destroy(M); return rc; }
    *out = M;
    return 0;
}

int har_band_rebalance(uint32_t rows, uint32_t n, const uint32_t *bounds, const double *seconds, uint32_t *out) {
    if (!bounds || !seconds || !out || n == 0) return har_set_error("har_band_rebalance: null argument");
    std::vector<uint32_t> b(bounds, bounds + n + 1), nb; std::vector<double> t(seconds, seconds + n);
    if (!rebalance_bounds(rows, b, t, nb)) nb = b;
    for (uint32_t k = 0; k <= n; ++k) out[k] = nb[k];
    return 0;
}

int har_multi_destroy(HarMulti M) { int dev = 0; (void) hipGetDevice(&dev); const int rc = destroy(M); (void) hipSetDevice(dev); return rc; }

int har_multi_replica(HarMulti M, uint32_t k, HarScene *scene, HarIntegrator *integrator, int *device) {
    if (!M || k >= M->rep.size()) return har_set_error("har_multi_replica: invalid index");
    if (scene) *scene = M->rep[k].scene;
    if (integrator) *integrator = M->rep[k].integ;
    if (device) *device = M->rep[k].device;
    return 0;
}

int har_multi_info(HarMulti M, uint32_t *n_devices, uint32_t *band_rows, float *band_ms, char *reduce, uint32_t reduce_len) {
    if (!M) return har_set_error("null group");
    const uint32_t n = (uint32_t) M->rep.size();
    if (n_devices) *n_devices = n;
    const HarMultiImpl::BandState &B = M->bands[0];
    if (band_rows) for (uint32_t r = 0; r <= n; ++r) band_rows[r] = r < B.bounds.size() ? B.bounds[r] : 0u;
    if (band_ms) for (uint32_t r = 0; r < n; ++r) band_ms[r] = r < B.last_ms.size() ? B.last_ms[r] : 0.f;
    if (reduce && reduce_len) snprintf(reduce, reduce_len, "%s", M->reduce_note.c_str());
    return 0;
}

int har_multi_render(HarMulti M, const HarSensor *sensor, uint32_t seed, uint32_t spp, int pixel_format, float *image, float *film_out, void *stream) {
    if (!M || !sensor) return har_set_error("har_multi_render: null argument");
    if (!image && !film_out) return har_set_error("har_multi_render: neither an image nor a film buffer");
    const uint32_t n = (uint32_t) M->rep.size();
    DSensor C; std::string e;
    if (!lower_sensor(*sensor, C, e)) return har_set_error(e);
    uint32_t spp_pass = spp, n_passes = 1;
    if (har_render_pass_layout(M->rep[0].integ, sensor, spp, &spp_pass, &n_passes)) return 1;
    const uint64_t row_lanes = (uint64_t) C.samp_w * spp_pass;                 /* lanes of one pixel row of the (per-pass) wavefront: bands are whole rows */
    if (row_lanes * C.samp_h > 0xffffffffull) return har_set_error("the per-pass wavefront exceeds 2^32 - 1 lanes");
    const size_t film_floats = (size_t) C.crop_w * C.crop_h * 4;
    int caller_device = 0; (void) hipGetDevice(&caller_device);
    struct Back { int d; ~Back() { (void) hipSetDevice(d); } } back{ caller_device };
    hipStream_t s0 = (hipStream_t) stream;
    /* (re)allocate the films when the sensor's crop window changed */
    if (film_floats != M->film_floats) {
        for (Replica &R : M->rep) {
            MULTI_TRY(hipSetDevice(R.device));
            MULTI_TRY(hipDeviceSynchronize());
            if (R.film) { (void) hipFree(R.film); R.film = nullptr; }
            MULTI_TRY(hipMalloc((void **) &R.film, film_floats * sizeof(float)));
        }
        MULTI_TRY(hipSetDevice(M->rep[0].device));
        for (uint32_t k = 1; k < n; ++k) {
            Replica &R = M->rep[k];
            if (R.staging) { (void) hipFree(R.staging); R.staging = nullptr; }
            if (!M->use_rccl) MULTI_TRY(hipMalloc((void **) &R.staging, film_floats * sizeof(float)));
        }
        M->film_floats = film_floats;
    }
    const uint32_t slot = begin_bands(M, 0, C.samp_h);
    const std::vector<uint32_t> &bounds = M->bands[0].bounds;
    /* 1. every device renders its band */
    MULTI_TRY(hipSetDevice(M->rep[0].device));
    MULTI_TRY(hipEventRecord(M->start, s0));
    int rc = 0;
    for (uint32_t k = 0; k < n; ++k) {
        Replica &R = M->rep[k];
        MULTI_TRY(hipSetDevice(R.device));
        hipStream_t s = k == 0 ? s0 : R.stream;
        if (k > 0) MULTI_TRY(hipStreamWaitEvent(s, M->start, 0));                /* after whatever the caller enqueued before the call (parameter updates) */
        MULTI_TRY(hipMemsetAsync(R.film, 0, film_floats * sizeof(float), s));
        MULTI_TRY(hipEventRecord(R.t0[0][slot], s));
        const uint64_t lb = (uint64_t) bounds[k] * row_lanes, le = (uint64_t) bounds[k + 1] * row_lanes;
        if (n == 1) rc = har_render(R.scene, R.integ, sensor, seed, spp, 0, 0, R.film, (void *) s);
        else if (le > lb) rc = har_render(R.scene, R.integ, sensor, seed, spp, lb, le, R.film, (void *) s);
        if (rc) return rc;
        MULTI_TRY(hipEventRecord(R.t1[0][slot], s));
        if (k > 0) MULTI_TRY(hipEventRecord(R.done, s));
    }
    /* 2. ONE collective: the films meet on device 0 */
    if (n > 1 && M->use_rccl) {
        int e2 = g_rccl.GroupStart();
        for (uint32_t k = 0; k < n && !e2; ++k) {
            Replica &R = M->rep[k];
            (void) hipSetDevice(R.device);
            e2 = g_rccl.Reduce(R.film, R.film, film_floats, 7 /* ncclFloat */, 0 /* ncclSum */, 0, R.comm, k == 0 ? s0 : R.stream);
        }
        const int e3 = g_rccl.GroupEnd();
        if (e2 || e3) return har_set_error(std::string("ncclReduce: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(e2 ? e2 : e3) : "error"));
    } else if (n > 1) {
        MULTI_TRY(hipSetDevice(M->rep[0].device));
        for (uint32_t k = 1; k < n; ++k) {
            Replica &R = M->rep[k];
            MULTI_TRY(hipStreamWaitEvent(s0, R.done, 0));
            MULTI_TRY(hipMemcpyPeerAsync(R.staging, M->rep[0].device, R.film, R.device, film_floats * sizeof(float), s0));
            launch_add(s0, R.staging, M->rep[0].film, (uint32_t) film_floats);
        }
        MULTI_TRY(hipGetLastError());
    }
    /* 3. device 0: the accumulated film and / or the developed image, in the caller's stream order */
    MULTI_TRY(hipSetDevice(M->rep[0].device));
    if (film_out) MULTI_TRY(hipMemcpyAsync(film_out, M->rep[0].film, film_floats * sizeof(float), hipMemcpyDeviceToDevice, s0));
    if (image) rc = har_film_develop_format(M->rep[0].film, C.crop_w, C.crop_h, pixel_format, image, (void *) s0);
    return rc;
}

} // extern "C"

/* the buffers of every device summed: on device 0 (root_only) or everywhere.  RCCL: one group call; otherwise peer copies + adds on device 0 (and copies back).
 * `buf(k)` / `staging(k)`: replica k's buffer and its landing area on device 0; `ready(k)`: the event after which replica k's buffer is complete; `count` floats */
template <typename Buf, typename Stage, typename Ready>
static int sum_over_devices(HarMultiImpl *M, hipStream_t s0, size_t count, bool root_only, Buf buf, Stage staging, Ready ready, hipEvent_t done0) {
    const uint32_t n = (uint32_t) M->rep.size();
    if (n < 2 || count == 0) return 0;
    if (M->use_rccl) {
        int e2 = g_rccl.GroupStart();
        for (uint32_t k = 0; k < n && !e2; ++k) {
            Replica &R = M->rep[k];
            (void) hipSetDevice(R.device);
            hipStream_t s = k == 0 ? s0 : R.stream;
            e2 = root_only ? g_rccl.Reduce(buf(k), buf(k), count, 7, 0, 0, R.comm, s) : g_rccl.AllReduce(buf(k), buf(k), count, 7, 0, R.comm, s);
        }
        const int e3 = g_rccl.GroupEnd();
        if (e2 || e3) return har_set_error(std::string("RCCL reduce: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(e2 ? e2 : e3) : "error"));
        return 0;
    }
    MULTI_TRY(hipSetDevice(M->rep[0].device));
    for (uint32_t k = 1; k < n; ++k) {
        Replica &R = M->rep[k];
        MULTI_TRY(hipStreamWaitEvent(s0, ready(k), 0));
        MULTI_TRY(hipMemcpyPeerAsync(staging(k), M->rep[0].device, buf(k), R.device, count * sizeof(float), s0));
        launch_add(s0, staging(k), buf(0), (uint32_t) count);
    }
    MULTI_TRY(hipGetLastError());
    if (root_only) return 0;
    MULTI_TRY(hipEventRecord(done0, s0));
    for (uint32_t k = 1; k < n; ++k) {
        Replica &R = M->rep[k];
        MULTI_TRY(hipSetDevice(R.device));
        MULTI_TRY(hipStreamWaitEvent(R.stream, done0, 0));
        MULTI_TRY(hipMemcpyPeerAsync(buf(k), R.device, buf(0), M->rep[0].device, count * sizeof(float), R.stream));
    }
    MULTI_TRY(hipSetDevice(M->rep[0].device));
    return 0;
}

extern "C" {

int har_multi_render_backward(HarMulti M, const HarSensor *sensor, const float *grad_in, uint32_t seed, uint32_t spp, float *grad_reflectance, float *const *grad_textures,
                              float *grad_emitters, void *stream) {
    if (!M || !sensor || !grad_in) return har_set_error("har_multi_render_backward: null argument");
    const uint32_t n = (uint32_t) M->rep.size();
    DSensor C; std::string e;
    if (!lower_sensor(*sensor, C, e)) return har_set_error(e);
    const uint64_t row_lanes = (uint64_t) C.samp_w * spp;
    if (row_lanes * C.samp_h > 0xffffffffull) return har_set_error("render_backward: the wavefront exceeds 2^32 - 1 lanes");       /* common.py:358-363 */
    const size_t film_floats = (size_t) C.crop_w * C.crop_h * 4, img_floats = (size_t) C.crop_w * C.crop_h * 3;
    int caller_device = 0; (void) hipGetDevice(&caller_device);
    struct Back { int d; ~Back() { (void) hipSetDevice(d); } } back{ caller_device };
    hipStream_t s0 = (hipStream_t) stream;
    /* buffers, on first use / when the film changed */
    if (film_floats != M->bfilm_floats || !M->rep[0].grads) {
        for (uint32_t k = 0; k < n; ++k) {
            Replica &R = M->rep[k];
            MULTI_TRY(hipSetDevice(R.device));
            MULTI_TRY(hipDeviceSynchronize());
            for (float **p : { &R.wfilm, &R.grad_in, &R.grads }) if (*p) { (void) hipFree(*p); *p = nullptr; }
            MULTI_TRY(hipMalloc((void **) &R.wfilm, film_floats * sizeof(float)));
            if (k > 0) MULTI_TRY(hipMalloc((void **) &R.grad_in, img_floats * sizeof(float)));
            MULTI_TRY(hipMalloc((void **) &R.grads, std::max<size_t>(M->grad_floats, 1) * sizeof(float)));
            R.tex_ptrs.clear();
            size_t off = 3 * ((size_t) M->bsdf_count + M->emitter_count);
            for (size_t f : M->tex_floats) { R.tex_ptrs.push_back(R.grads + off); off += f; }
            if (!R.bdone) { MULTI_TRY(hipEventCreateWithFlags(&R.bdone, hipEventDisableTiming)); MULTI_TRY(hipEventCreateWithFlags(&R.wdone, hipEventDisableTiming)); }
        }
        MULTI_TRY(hipSetDevice(M->rep[0].device));
        for (uint32_t k = 1; k < n; ++k) {
            Replica &R = M->rep[k];
            if (R.grads_staging) { (void) hipFree(R.grads_staging); R.grads_staging = nullptr; }
            if (!M->use_rccl) MULTI_TRY(hipMalloc((void **) &R.grads_staging, std::max(std::max<size_t>(M->grad_floats, 1), film_floats) * sizeof(float)));
        }
        if (!M->wsum) { MULTI_TRY(hipEventCreateWithFlags(&M->wsum, hipEventDisableTiming)); MULTI_TRY(hipEventCreateWithFlags(&M->gin, hipEventDisableTiming)); }
        M->bfilm_floats = film_floats;
    }
    /* bands of the adjoint (their own: its cost profile is not the forward render's), re-cut from the previous call's device times */
    const uint32_t slot = begin_bands(M, 1, C.samp_h);
    const std::vector<uint32_t> &bbounds = M->bands[1].bounds;
    MULTI_TRY(hipSetDevice(M->rep[0].device));
    MULTI_TRY(hipEventRecord(M->start, s0));
    /* 1. the filter weights of every band's samples; their sum W[px] is needed by every device (the adjoint of develop divides by it: common.py:696-746) */
    for (uint32_t k = 0; k < n; ++k) {
        Replica &R = M->rep[k];
        MULTI_TRY(hipSetDevice(R.device));
        hipStream_t s = k == 0 ? s0 : R.stream;
        if (k > 0) MULTI_TRY(hipStreamWaitEvent(s, M->start, 0));
        MULTI_TRY(hipMemsetAsync(R.wfilm, 0, film_floats * sizeof(float), s));
        const uint64_t lb = (uint64_t) bbounds[k] * row_lanes, le = (uint64_t) bbounds[k + 1] * row_lanes;
        if (n == 1) { if (har_render_weights(sensor, seed, spp, 0, 0, R.wfilm, (void *) s)) return 1; }
        else if (le > lb && har_render_weights(sensor, seed, spp, lb, le, R.wfilm, (void *) s)) return 1;
        if (k > 0) { MULTI_TRY(hipEventRecord(R.wdone, s)); MULTI_TRY(hipMemcpyPeerAsync(R.grad_in, R.device, grad_in, M->rep[0].device, img_floats * sizeof(float), s)); }
    }
    if (sum_over_devices(M, s0, film_floats, false, [&](uint32_t k) { return M->rep[k].wfilm; }, [&](uint32_t k) { return M->rep[k].grads_staging; },
                         [&](uint32_t k) { return M->rep[k].wdone; }, M->wsum)) return 1;
    /* 2. every device replays its band: primal pass + adjoint pass, gradients into its own flat buffer */
    for (uint32_t k = 0; k < n; ++k) {
        Replica &R = M->rep[k];
        MULTI_TRY(hipSetDevice(R.device));
        hipStream_t s = k == 0 ? s0 : R.stream;
        MULTI_TRY(hipEventRecord(R.t0[1][slot], s));                 /* the band's own work: after the wait for every device's weights */
        MULTI_TRY(hipMemsetAsync(R.grads, 0, std::max<size_t>(M->grad_floats, 1) * sizeof(float), s));
        if (har_integrator_set_grad_emitters(R.integ, grad_emitters ? R.grads + 3 * (size_t) M->bsdf_count : nullptr)) return 1;
        const uint64_t lb = (uint64_t) bbounds[k] * row_lanes, le = (uint64_t) bbounds[k + 1] * row_lanes;
        const float *gi = k == 0 ? grad_in : R.grad_in;
        int rc = 0;
        if (n == 1) rc = har_render_backward(R.scene, R.integ, sensor, gi, R.wfilm, seed, spp, 0, 0, R.grads, R.tex_ptrs.empty() ? nullptr : R.tex_ptrs.data(), (void *) s);
        else if (le > lb) rc = har_render_backward(R.scene, R.integ, sensor, gi, R.wfilm, seed, spp, lb, le, R.grads, R.tex_ptrs.empty() ? nullptr : R.tex_ptrs.data(), (void *) s);
        if (rc) return rc;
        MULTI_TRY(hipEventRecord(R.t1[1][slot], s));
        if (k > 0) MULTI_TRY(hipEventRecord(R.bdone, s));
    }
    /* 3. ONE collective for all gradient buffers (they are one flat array per device), then device 0 adds the pieces to the caller's buffers */
    if (sum_over_devices(M, s0, M->grad_floats, true, [&](uint32_t k) { return M->rep[k].grads; }, [&](uint32_t k) { return M->rep[k].grads_staging; },
                         [&](uint32_t k) { return M->rep[k].bdone; }, M->gin)) return 1;
    MULTI_TRY(hipSetDevice(M->rep[0].device));
    Replica &R0 = M->rep[0];
    if (grad_reflectance && M->bsdf_count) launch_add(s0, R0.grads, grad_reflectance, 3u * M->bsdf_count);
    if (grad_emitters && M->emitter_count) launch_add(s0, R0.grads + 3 * (size_t) M->bsdf_count, grad_emitters, 3u * M->emitter_count);
    if (grad_textures) for (size_t t = 0; t < M->tex_floats.size(); ++t) if (grad_textures[t]) launch_add(s0, R0.tex_ptrs[t], grad_textures[t], (uint32_t) M->tex_floats[t]);
    MULTI_TRY(hipGetLastError());
    return 0;
}

} // extern "C"
