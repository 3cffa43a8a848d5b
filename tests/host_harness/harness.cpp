/*
 * harness.cpp -- HOST TEST HARNESS (test tool, not a product path).
 *
 * Compiles the product's HAR_HD device functions (mitsuba3_amd/csrc/*.h: BVH8
 * traversal, surface interaction, shading stages, film footprint) and its host
 * scene lowering with g++ and drives them lane by lane, so that the logic the
 * HIP kernels execute can be checked against the oracle on a machine without
 * a GPU.  The shipped library (libhip_ad_rgb.so) never contains or calls this.
 */
#include "../../mitsuba3_amd/csrc/har_path.h"
#include "../../mitsuba3_amd/csrc/har_scene_host.h"
#include <cstdio>
#include <string>
#include <vector>

using namespace har;

struct HostStack {
    static constexpr int Capacity = 24;
    uint32_t x[Capacity], y[Capacity];
    void push(int l, uint32_t a, uint32_t b) { x[l] = a; y[l] = b; }
    void pop(int l, uint32_t &a, uint32_t &b) { a = x[l]; b = y[l]; }
};

struct HScene {
    HostScene hs;
    std::vector<DTexture> dtex;
    DScene ds;
};

static void bind(HScene &H) {
    HostScene &hs = H.hs;
    H.dtex.clear();
    for (auto &t : hs.textures) H.dtex.push_back(DTexture{ t.data.data(), t.w, t.h });
    DScene &S = H.ds;
    S.accel.nodes = hs.nodes.data(); S.accel.tris = hs.tris.data(); S.accel.insts = hs.inst_recs.data();
    S.accel.root = hs.root; S.accel.has_tlas = hs.has_tlas; S.accel.n_tris = (uint32_t) hs.tris.size(); S.accel.n_insts = (uint32_t) hs.inst_recs.size();
    S.blas_tri_ranges = hs.blas_tri_ranges.data();
    S.verts = hs.verts.data(); S.faces = hs.faces.data(); S.meshes = hs.meshes.data(); S.bsdfs = hs.bsdfs.data();
    S.textures = H.dtex.data(); S.emitters = hs.emitters.data(); S.insts = hs.insts.data();
    S.n_emitters = (uint32_t) hs.emitters.size(); S.n_meshes = (uint32_t) hs.meshes.size();
    S.n_bsdfs = (uint32_t) hs.bsdfs.size(); S.n_textures = (uint32_t) hs.textures.size();
}

extern "C" {

void *hh_scene_create(const HarSceneDesc *d, char *err, int errlen) {
    HScene *H = new HScene();
    std::string e;
    if (!lower_scene(*d, H->hs, e)) { snprintf(err, errlen, "%s", e.c_str()); delete H; return nullptr; }
    bind(*H);
    return H;
}
void hh_scene_destroy(void *h) { delete (HScene *) h; }
void hh_scene_info(void *h, uint64_t info[4]) {
    HScene *H = (HScene *) h; info[0] = H->hs.nodes.size(); info[1] = H->hs.tris.size(); info[2] = H->hs.stats.max_depth; info[3] = H->hs.inst_recs.size();
}

int hh_trace(void *h, uint32_t n, const float *o, const float *d, const float *maxt, int naive, int anyhit,
             float *t, float *u, float *v, uint32_t *prim, uint32_t *shape, uint32_t *inst, uint8_t *hitflag) {
    HScene *H = (HScene *) h; int status = 0;
    for (uint32_t i = 0; i < n; ++i) {
        Vec3 O(o[i], o[n + i], o[2 * (size_t) n + i]), D(d[i], d[n + i], d[2 * (size_t) n + i]);
        Hit hit; HostStack st; bool r;
        if (naive) r = anyhit ? accel_trace_naive<true>(H->ds.accel, H->ds.blas_tri_ranges, O, D, maxt[i], hit)
                              : accel_trace_naive<false>(H->ds.accel, H->ds.blas_tri_ranges, O, D, maxt[i], hit);
        else       r = anyhit ? accel_trace<true>(H->ds.accel, O, D, maxt[i], hit, st, status)
                              : accel_trace<false>(H->ds.accel, O, D, maxt[i], hit, st, status);
        if (anyhit) hitflag[i] = r;
        else { t[i] = hit.t; u[i] = hit.u; v[i] = hit.v; prim[i] = hit.prim; shape[i] = hit.shape; inst[i] = hit.inst; }
    }
    return status;
}

/* lane-by-lane emulation of the wavefront pipeline (raygen -> {trace, shade, shadow}* -> splat) */
int hh_render(void *h, const HarSensor *sensor, int mode, uint32_t seed, uint32_t spp, int32_t max_depth, int32_t rr_depth,
              uint64_t lane_begin, uint64_t lane_end, float *film) {
    HScene *H = (HScene *) h; const DScene &S = H->ds;
    DSensor C; std::string e; if (!lower_sensor(*sensor, C, e)) return -1;
    uint64_t total = (uint64_t) C.crop_w * C.crop_h * spp;
    if (lane_begin == 0 && lane_end == 0) lane_end = total;
    uint32_t log_spp = 0xffffffffu; for (uint32_t k = 0; k < 32; ++k) if ((1u << k) == spp) log_spp = k;
    ShadeParams P{ seed, (uint32_t) max_depth, (uint32_t) rr_depth };
    int status = 0;
    for (uint64_t lane = lane_begin; lane < lane_end; ++lane) {
        LaneSample ls; PathState st = raygen_lane(C, seed, spp, log_spp, (uint32_t) lane, ls);
        Vec3 result(0.f);
        bool alive = P.max_depth != 0;
        while (alive) {
            Hit hit; HostStack stack;
            accel_trace<false>(S.accel, st.o, st.d, st.maxt, hit, stack, status);
            ShadeResult R;
            if (mode == MODE_PATH) shade_lane<MODE_PATH>(S, P, st, hit, R); else shade_lane<MODE_PRB_PRIMAL>(S, P, st, hit, R);
            if (R.add_emission) result = mode == MODE_PATH ? fma3(R.em_a, R.em_b, result) : result + R.em_b;
            if (R.item && R.item_ray) {
                Hit sh; HostStack s2;
                if (!accel_trace<true>(S.accel, R.sh_o, R.sh_d, R.sh_maxt, sh, s2, status)) result = result + R.contrib;
            }
            alive = R.alive; st = R.next;
        }
        LaneSample fp = lane_film_pos(C, seed, spp, log_spp, (uint32_t) lane);
        Footprint F; film_footprint(C, fp, F);
        const float val[4] = { result.x, result.y, result.z, 1.f };
        for (uint32_t ys = 0; ys < F.count; ++ys) for (uint32_t xs = 0; xs < F.count; ++xs) {
            uint32_t x = F.x0 + xs, y = F.y0 + ys;
            if (x < C.crop_w && y < C.crop_h) { float w = F.wx[xs] * F.wy[ys]; float *p = film + 4 * ((size_t) y * C.crop_w + x); for (int k = 0; k < 4; ++k) p[k] += val[k] * w; }
        }
    }
    return status;
}

} // extern "C"
