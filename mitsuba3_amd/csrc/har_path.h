/*
 * har_path.h -- per-lane stages of the wavefront integrators (HAR_HD).
 *
 * The HIP kernels in har_kernels.hip are thin wrappers (load SoA state ->
 * call the stage -> ballot-compact -> store); all path logic lives here so
 * that it is identical on the GPU and in the host test harness.
 *
 *  MODE_PATH        PathIntegrator::sample, src/integrators/path.cpp:94-346
 *  MODE_PRB_PRIMAL  PRBIntegrator.sample(mode=Primal), ad/integrators/prb.py:68-339
 *  MODE_PRB_ADJOINT PRBIntegrator.sample(mode=Backward), same file, with the
 *                   hand-derived diffuse/albedo gradients of SURVEY.md App. B
 */
#pragma once
#include "har_scene.h"

namespace har {

enum { MODE_PATH = 0, MODE_PRB_PRIMAL = 1, MODE_PRB_ADJOINT = 2 };

#define HAR_MAX_FILTER_TAPS 9

/* Loop state of one path between two bounces (LoopState, path.cpp:129-159) */
struct PathState {
    Vec3 o, d; float maxt;
    Vec3 throughput;
    uint64_t rng;
    uint32_t lane;
    Vec3 prev_p; float prev_bsdf_pdf;
    uint32_t flags;            /* bits 0..15 depth, bit 16 prev_bsdf_delta */
    float eta;                 /* product of the sampled relative IORs (path.cpp:300, prb.py:229) */
};

struct ShadeParams {
    uint32_t seed, max_depth, rr_depth;
    uint32_t flags = 0;        /* bit 0 (adjoint): also accumulate the gradient w.r.t. the radiance of `area` / `constant` emitters; bit 1: hide_emitters */
    uint32_t sort_window = 1;  /* generic shading kernels: tiles of 256 paths per material-sort window (k_shade) */
    /* HAR_SHADE_FIRST_VERTEX (bounce 0 of a wavefront that starts at the sensor): the path state of a slot is a function of its lane index alone -- the shading kernel
     * rebuilds it (raygen_lane) instead of reading 72 bytes that the ray generation kernel would have to write first */
    uint32_t spp = 0, log_spp = 0, resume = 0;      /* resume: pass > 0 of a multi-pass render -- the lane's sampler continues from its stored state (k_shade's pass_rng) */
    DSensor sensor{};
};
#define HAR_SHADE_FIRST_VERTEX 32u
#define HAR_SHADE_EMITTER_GRADS 1u
#define HAR_SHADE_FORWARD_MODE 4u    /* adjoint kernels in FORWARD mode (RBIntegrator.render_forward): parameter tangents in, differential radiance out */
#define HAR_SHADE_EXTRA_GRADS 16u    /* adjoint: also differentiate w.r.t. alpha / eta / k / colour slot 1 of the rough BSDF models (ShadeResult::x_dir / x_rel) */
#define HAR_SHADE_SCALAR_DRAWS 8u    /* scalar variants: the two emitter samples are only drawn where the BSDF has a smooth lobe (path.cpp:244-249); JIT variants draw them on every lane */
#define HAR_SHADE_LIGHT_TEXELS 64u    /* adjoint: also differentiate w.r.t. the texels of bitmap `radiance` textures of area lights (ShadeResult::lt_*; committed in place by the re-shading replay) */
#define HAR_SHADE_HIDE_EMITTERS 2u   /* Integrator property `hide_emitters`: the environment is not seen by camera rays (path.cpp:114-115, prb.py:146-148) */
#define HAR_ITEM_NO_EMITTER 0x7ffu   /* emitter field of an adjoint item's tag: contribution stored as is (no emitter gradient) */

struct ShadeResult {
    bool alive;
    PathState next;
    bool add_emission; Vec3 em_a, em_b;   /* PATH: result = fma(em_a, em_b, result); PRB: L +/-= em_b */
    bool item;                            /* enqueue an NEE (+ gradient) item */
    bool item_ray;                        /* item carries a shadow ray to test */
    Vec3 sh_o, sh_d; float sh_maxt;
    Vec3 contrib;                         /* PATH: throughput*bsdf*em*mis; PRB: Lr_dir */
    /* MODE_PRB_ADJOINT only: d Lr_dir / d slot0, and (d f / d slot0) / f at the sampled direction */
    Vec3 dLr_drho, rel_grad; bool ind_active; uint32_t bsdf; float uv_x, uv_y;
    /* ... for the COMPACT record of diffuse-only scenes (HAR_TAPE_COMPACT, k_shade<.., RECORD> / k_commit<.., COMPACT>): d Lr_dir / d slot0 for a UNIT radiance of
     * `nee_emitter`, and the colour of slot 0 at the vertex -- for a diffuse vertex Lr_dir = dLr_drho * rho and (df / d rho) / f = 1 / rho */
    Vec3 dLr_drho_unit, rho;
    /* ... with HAR_SHADE_EMITTER_GRADS: d em_b / d radiance of emitter `em_index` (emission hit; -1 = none), and the NEE contribution for a UNIT
     * radiance of emitter `nee_emitter` (contrib = contrib_unit * radiance; -1 = not factorable, e.g. an environment map) */
    Vec3 em_unit; int32_t em_index; Vec3 contrib_unit; int32_t nee_emitter;
    /* ... with vertex-position gradients (har_shape_grad.h): the detached emitter sample (position or, for an environment, direction; normal),
     * cos(theta_o) towards it and the HAR_SHAPE_* flags of the vertex */
    Vec3 nee_p, nee_n; float cos_em; uint32_t nee_flags;
    Vec3 nee_w;                           /* beta * mis * em_weight: Lr_dir = nee_w * f(wi, wo_em) */
    /* ... with HAR_SHADE_EXTRA_GRADS: the non-slot-0 parameters of the vertex's BSDF record, groups 0 alpha_u, 1 alpha_v, 2 eta, 3 k (roughconductor's
     * complex IOR, per channel), 4 colour slot 1: x_dir[g] = d Lr_dir / d theta_g (per channel), x_rel[g] = (d f / d theta_g) / f at the sampled
     * direction -- what dLr_drho / rel_grad are for slot 0 (prb.py:288-313) */
    Vec3 x_dir[5], x_rel[5]; bool x_ind;
    /* ... with HAR_SHADE_LIGHT_TEXELS (kernels of scenes with a bitmap-radiance area light only): the emitter met at the vertex / sampled from it when its radiance is a bitmap
     * (-1: none), and the uv its bitmap is looked up at -- si.uv of the hit (area.cpp:83-90), ds.uv of the sample (:139-146).  d em_b / d radiance(uv) = em_unit, d contrib / d
     * radiance(uv) = contrib_unit, as for the colour of a uniform light */
    int32_t lt_hit_emitter, lt_nee_emitter; float lt_hit_uv[2], lt_nee_uv[2];
};
#define HAR_EXTRA_GROUPS 5

/* raygen: SamplingIntegrator::render_sample up to the camera ray (integrator.cpp:448-483) */
HAR_HD PathState raygen_lane(const DSensor &C, uint32_t seed, uint32_t spp, uint32_t log_spp, uint32_t lane, LaneSample &ls,
                             const uint64_t *resume = nullptr, float *jitter = nullptr) {
    PathState st;
    uint64_t inc;
    sampler_seed(seed, lane, st.rng, inc);
    if (resume) st.rng = *resume;                 /* pass > 0 of a multi-pass render: the stream continues (integrator.cpp:349-356) */
    float jx = pcg32_next_float(st.rng, inc), jy = pcg32_next_float(st.rng, inc);
    if (jitter) { jitter[0] = jx; jitter[1] = jy; }
    ls = lane_sample(C, lane, spp, log_spp, jx, jy);
    lane_camera_ray(C, ls, st.o, st.d, st.maxt);
    st.throughput = Vec3(1.f); st.lane = lane; st.prev_p = Vec3(0.f); st.prev_bsdf_pdf = 1.f; st.flags = 1u << 16; st.eta = 1.f;
    return st;
}

/* film position of a lane, recomputed from its RNG stream (saves 8 B/lane of state) */
HAR_HD LaneSample lane_film_pos(const DSensor &C, uint32_t seed, uint32_t spp, uint32_t log_spp, uint32_t lane) {
    uint64_t rng, inc;
    sampler_seed(seed, lane, rng, inc);
    float jx = pcg32_next_float(rng, inc), jy = pcg32_next_float(rng, inc);
    return lane_sample(C, lane, spp, log_spp, jx, jy);
}

/* ImageBlock::put footprint, coalesced JIT branch (imageblock.cpp:444-470): top-left pixel
 * (crop-relative, may be "negative" = huge unsigned), per-axis weights */
struct Footprint { uint32_t x0, y0, count; float wx[HAR_MAX_FILTER_TAPS], wy[HAR_MAX_FILTER_TAPS]; };
HAR_HD void film_footprint(const DSensor &C, const LaneSample &L, Footprint &F) {
    if (C.rfilter == 0) {
        F.count = 1; F.wx[0] = 1.f; F.wy[0] = 1.f;
        F.x0 = (uint32_t) ((int32_t) floorf(L.ipos_x) - (int32_t) C.crop_x);
        F.y0 = (uint32_t) ((int32_t) floorf(L.ipos_y) - (int32_t) C.crop_y);
        return;
    }
    uint32_t n = (uint32_t) ceilf(C.radius - .5f);
    F.count = 2 * n + 1;
    int32_t ix = (int32_t) floorf(L.pos_x) - (int32_t) n, iy = (int32_t) floorf(L.pos_y) - (int32_t) n;
    F.x0 = (uint32_t) (ix - (int32_t) C.crop_x); F.y0 = (uint32_t) (iy - (int32_t) C.crop_y);
    float relx = ((float) ix + .5f) - L.pos_x, rely = ((float) iy + .5f) - L.pos_y;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (uint32_t k = 0; k < HAR_MAX_FILTER_TAPS; ++k) {      /* static indices: the weights stay in registers */
        if (k < F.count) { F.wx[k] = rfilter_eval(C, relx + (float) k); F.wy[k] = rfilter_eval(C, rely + (float) k); }
    }
}

/* what a closest-hit record carries beyond the preliminary intersection (Accel::mesh_info; HAR_HIT_MATINFO): the face's index in the shading-triangle array, the mesh's material word */
struct HitExtra { uint32_t gface, matinfo; };

template <int MODE, uint32_t TYPES = HAR_BSDF_ALL_TYPES, bool EXTRA = false>
HAR_HD void shade_lane(const DScene &S, const ShadeParams &P, const PathState &st, const Hit &hit, ShadeResult &R, const HitExtra *hx = nullptr) {
    R.alive = false; R.add_emission = false; R.item = false; R.item_ray = false;
    if (MODE == MODE_PRB_ADJOINT && EXTRA) { for (int g = 0; g < 5; ++g) { R.x_dir[g] = Vec3(0.f); R.x_rel[g] = Vec3(0.f); } R.x_ind = false; }
    if (MODE == MODE_PRB_ADJOINT) { R.lt_hit_emitter = -1; R.lt_nee_emitter = -1; R.lt_hit_uv[0] = R.lt_hit_uv[1] = R.lt_nee_uv[0] = R.lt_nee_uv[1] = 0.f; }
    if (MODE == MODE_PRB_ADJOINT) { R.em_index = -1; R.nee_emitter = -1; R.em_unit = Vec3(0.f); R.contrib_unit = Vec3(0.f); R.nee_flags = 0u; R.cos_em = 0.f; R.nee_p = Vec3(0.f); R.nee_n = Vec3(0.f); R.nee_w = Vec3(0.f); }
    uint64_t rng = st.rng;
    const uint64_t inc = sampler_inc(P.seed, st.lane);
    const uint32_t depth = st.flags & 0xffffu;
    const bool prev_delta = (st.flags >> 16) & 1u;

    /* with a hit record that carries the face index and the material word, nothing below waits for a load of the mesh record (only an emitter hit reads it) */
    SurfInt si = hx ? compute_si_record(S, st.d, hit.t, hit.u, hit.v, hx->gface, HAR_MATINFO_FLAGS(hx->matinfo), hit.shape, hit.inst)
                    : compute_si(S, st.d, hit.t, hit.u, hit.v, hit.prim, hit.shape, hit.inst);
    const bool valid = si.valid();
    DMesh M{};
    if (valid && hx) { M.bsdf = HAR_MATINFO_BSDF(hx->matinfo); M.emitter = HAR_MATINFO_EMITTER(hx->matinfo) ? S.meshes[si.mesh].emitter : -1; }
    else if (valid) M = S.meshes[si.mesh];
    const int emitter = valid ? M.emitter : S.env_emitter;          /* si.emitter(scene): the environment for a miss (scene.h:822-832) */
    const float pmf = S.n_emitters ? 1.f / (float) S.n_emitters : 0.f;   /* scene.cpp:139 */
    /* Scene::m_emitter_distr (scene.cpp:120-141): non-uniform emitter selection; only in the kernels of scenes that carry the generic emitter code */
    const bool distr = (TYPES & HAR_SCENE_ENVMAP) != 0u && S.emitter_distr != nullptr;

    /* ---- direct emission + MIS with the previous BSDF sample (path.cpp:206-221, prb.py:148-161).  hide_emitters: a camera ray that escapes does
     * not see the environment (path.cpp:114-115 keeps valid_ray false, so the sample's result is dropped; prb.py:146-148 masks the eval) */
    if (emitter >= 0 && !((P.flags & HAR_SHADE_HIDE_EMITTERS) && depth == 0u && !valid)) {
        const DEmitter E = S.emitters[emitter];
        const bool env = E.type == 1u || E.type == 2u;
        const bool envmap = (TYPES & HAR_SCENE_ENVMAP) != 0u && E.type == 2u;      /* kernels of scenes without an environment map compile this out */
        Vec3 rel = si.p - st.prev_p;
        float dist = norm3(rel);
        Vec3 dd = div3(rel, dist);
        /* pdf_direction: area light (area.cpp:170-197) or uniform sphere (constant.cpp:155-160) */
        /* Scene::pdf_emitter_direction (scene.cpp:378-388): emitter_pmf = m_emitter_pmf, or sampling_weight * normalization with a distribution */
        const float hit_pmf = distr ? S.emitter_distr[emitter] * S.emitter_norm : pmf;
        const bool textured = (TYPES & HAR_SCENE_TEXLIGHT) != 0u && E.type == 7u;     /* bitmap radiance (area.cpp:83-90, 185-191): value and density depend on si.uv */
        float em_pdf = prev_delta ? 0.f : (envmap ? envmap_pdf_direction(*S.envmap, st.d) : env ? HAR_INV_FOUR_PI :
                                           textured ? textured_area_pdf_direction(S, E, dd, si.sn, dist, si.uv_x, si.uv_y) : emitter_pdf_direction(E, dd, si.sn, dist)) * hit_pmf;
        float mis = mis_weight(st.prev_bsdf_pdf, em_pdf);
        Vec3 rad(E.radiance[0], E.radiance[1], E.radiance[2]);
        if (envmap) rad = envmap_eval(*S.envmap, st.d);                              /* envmap.cpp:228-236: v = to_world^-1 * (-si.wi) */
        if (textured) rad = texture_eval_uv(S.textures[as_u32(E.radiance[0])], si.uv_x, si.uv_y);
        const bool facing = env || si.wi.z > 0.f;                                      /* area.cpp:83-90 / constant.cpp:90-94 */
        if (MODE == MODE_PATH) {
            Vec3 Le = (facing && st.prev_bsdf_pdf > 0.f) ? rad : Vec3(0.f);
            R.add_emission = true; R.em_a = st.throughput; R.em_b = Le * mis;
        } else {
            Vec3 ev = facing ? rad : Vec3(0.f);
            R.add_emission = true; R.em_a = Vec3(0.f); R.em_b = (st.throughput * mis) * ev;
            if (MODE == MODE_PRB_ADJOINT) {          /* prb.py:160-161 with the emitter's eval attached: d Le / d radiance = beta * mis */
                R.em_index = (E.type != 2u && E.type != 7u && facing) ? emitter : -1;      /* (no colour parameter behind an environment map or a bitmap radiance) */
                R.em_unit = st.throughput * mis;
                if (textured && facing && (P.flags & HAR_SHADE_LIGHT_TEXELS)) { R.lt_hit_emitter = emitter; R.lt_hit_uv[0] = si.uv_x; R.lt_hit_uv[1] = si.uv_y; }
            }
        }
    }

    bool active_next = (depth + 1u < P.max_depth) && valid;
    if (MODE == MODE_PATH && !active_next) return;
    if (MODE != MODE_PATH && !valid) return;      /* prb: nothing below contributes for a miss */

    /* BSDF record serving this side (twosided.cpp) and its evaluated colour parameters */
    BsdfSide side; bool side_ok = true;
    if (TYPES == HAR_BSDF_ONLY_DIFFUSE) { side.index = M.bsdf; side.wi = si.wi; side.wo_sign = 1.f; }      /* no twosided records either */
    else side_ok = bsdf_side(S, M.bsdf, si.wi, side);
    const DBsdf B = S.bsdfs[side.index];
    TexTaps taps;
    const BsdfInputs bin = bsdf_inputs(S, B, si.uv_x, si.uv_y, taps);

    /* ---- emitter sampling (path.cpp:238-258, prb.py:163-175; scene.cpp:316-366).  The two samples are drawn by
     * every active lane; only BSDFs with a Smooth lobe use them (path.cpp:237, prb.py:169) */
    float ex = 0.f, ey = 0.f;
    /* BSDFFlags::Smooth of the record: a compile-time constant in the kernels of a single-model material queue */
    const bool smooth = HAR_BSDF_SINGLE(TYPES) ? (HAR_BSDF_SINGLE_TYPE(TYPES) != (uint32_t) BSDF_DIELECTRIC && HAR_BSDF_SINGLE_TYPE(TYPES) != (uint32_t) BSDF_CONDUCTOR) : bsdf_is_smooth(B);
    if (!(P.flags & HAR_SHADE_SCALAR_DRAWS) || smooth) { ex = pcg32_next_float(rng, inc); ey = pcg32_next_float(rng, inc); }
    DirSample ds; ds.pdf = 0.f; ds.d = Vec3(0.f); ds.p = Vec3(0.f); ds.n = Vec3(0.f); ds.dist = 0.f;
    Vec3 em_weight(0.f); float em_unit = 0.f; uint32_t em_sampled = 0;
    bool lt_sampled = false;                                     /* HAR_SHADE_LIGHT_TEXELS: the sample lies on a bitmap-radiance light */
    bool em_delta = false;                                       /* DirectionSample::delta: a point light's sample carries MIS weight 1 (path.cpp:274, prb.py:211) */
    bool active_em = active_next && S.n_emitters > 0 && smooth;
    if (active_em) {
        uint32_t index = 0; float wgt = 1.f, sel_pmf = pmf;
        if (S.n_emitters > 1 && distr) {                         /* sample_emitter with m_emitter_distr (scene.cpp:258-261): sample_reuse_pmf, weight = rcp(pmf) */
            float reused, p;
            index = discrete_sample_reuse_pmf(S.emitter_distr, S.emitter_distr + S.n_emitters, S.emitter_valid_lo, S.emitter_valid_hi, S.emitter_sum, S.emitter_norm, ex,
                                              !(P.flags & HAR_SHADE_SCALAR_DRAWS), reused, p);
            wgt = rcp_(p); ex = reused;
        } else if (S.n_emitters > 1) {                           /* sample_emitter, scene.cpp:248-271 */
            float scaled = ex * (float) S.n_emitters;
            index = (uint32_t) scaled; if (index > S.n_emitters - 1u) index = S.n_emitters - 1u;
            wgt = (float) S.n_emitters; ex = scaled - (float) index;
        }
        /* pdf_emitter(index) = eval_pmf_normalized (scene.cpp:273-279, applied at :338).  JIT variants take that branch for a single emitter too (:326); the scalar
         * variants sample their only emitter directly (:351-354) and never multiply by its pmf */
        if (distr && (S.n_emitters > 1 || !(P.flags & HAR_SHADE_SCALAR_DRAWS))) sel_pmf = S.emitter_distr[index] * S.emitter_norm;
        else if (distr) sel_pmf = 1.f;
        if ((TYPES & HAR_SCENE_ENVMAP) != 0u && S.emitters[index].type == 2u) envmap_sample_direction(*S.envmap, si.p, ex, ey, ds, em_weight);
        else if ((TYPES & HAR_SCENE_ENVMAP) != 0u && S.emitters[index].type == 3u)
            mesh_emitter_sample_direction(S, S.emitters[index], si.p, ex, ey, ds, em_weight, MODE != MODE_PATH ? &em_unit : nullptr);
        else if ((TYPES & HAR_SCENE_ENVMAP) != 0u && S.emitters[index].type == 4u) {
            point_sample_direction(S.emitters[index], si.p, ds, em_weight, MODE != MODE_PATH ? &em_unit : nullptr); em_delta = true;
        } else if ((TYPES & HAR_SCENE_ENVMAP) != 0u && S.emitters[index].type == 5u) {
            spot_sample_direction(S.emitters[index], si.p, ds, em_weight, MODE != MODE_PATH ? &em_unit : nullptr); em_delta = true;
        } else if ((TYPES & HAR_SCENE_ENVMAP) != 0u && S.emitters[index].type == 6u) {
            directional_sample_direction(S.emitters[index], si.p, ds, em_weight, MODE != MODE_PATH ? &em_unit : nullptr); em_delta = true;
        } else if ((TYPES & HAR_SCENE_TEXLIGHT) != 0u && S.emitters[index].type == 7u) {
            /* (the PRIMAL pass of a texel-gradient step asks for the unit weight too: a sample over texels that are all ZERO contributes nothing and still has a derivative,
             * so its shadow ray must be traced by the pass that fills the visibility bytes -- lt_item below) */
            const bool lt = MODE != MODE_PATH && (P.flags & HAR_SHADE_LIGHT_TEXELS) != 0u;
            float lt_uv[2] = { 0.f, 0.f };
            textured_area_sample_direction(S, S.emitters[index], si.p, ex, ey, ds, em_weight, lt ? &em_unit : nullptr, lt ? lt_uv : nullptr);
            if (lt) lt_sampled = true;
            if (lt && MODE == MODE_PRB_ADJOINT) { R.lt_nee_emitter = (int32_t) index; R.lt_nee_uv[0] = lt_uv[0]; R.lt_nee_uv[1] = lt_uv[1]; }
        }
        /* a scene with ONE emitter (the bench scenes): its record travels with the kernel arguments (DScene::emitter0), i.e. in scalar registers, instead of being gathered
         * by every lane -- six vector loads less in a kernel that is bound by the number of its memory transactions (k_shade, docs/rounds/r05.md item 11) */
        else if (S.n_emitters == 1u && S.emitter0_valid) {
            if (S.emitter0_valid == 2u) {      /* its radiance was pushed device-to-device since (har_scene_set_emitter_radiance_device): three floats from the array, the rest from the arguments */
                DEmitter E0 = S.emitter0;
                const float *rad = S.emitters[0].radiance;
                E0.radiance[0] = rad[0]; E0.radiance[1] = rad[1]; E0.radiance[2] = rad[2];
                emitter_sample_direction(E0, si.p, ex, ey, ds, em_weight, MODE != MODE_PATH ? &em_unit : nullptr);
            } else emitter_sample_direction(S.emitter0, si.p, ex, ey, ds, em_weight, MODE != MODE_PATH ? &em_unit : nullptr);
        }
        else emitter_sample_direction(S.emitters[index], si.p, ex, ey, ds, em_weight, MODE != MODE_PATH ? &em_unit : nullptr);
        ds.pdf *= sel_pmf; em_weight = em_weight * wgt; em_unit *= wgt; em_sampled = index;
        active_em = ds.pdf != 0.f;
    }
    Vec3 wo_em = active_em ? si.to_local(ds.d) : Vec3(0.f);

    /* ---- BSDF (bsdf.cpp:21-31 eval_pdf_sample) */
    float s1 = pcg32_next_float(rng, inc);
    float s2x = pcg32_next_float(rng, inc), s2y = pcg32_next_float(rng, inc);
    BsdfEval ev; bsdf_eval_pdf<TYPES>(S, side, bin, side_ok, wo_em, ev);
    BsdfSample bs; bs.wo = Vec3(0.f); bs.pdf = 0.f; bs.weight = Vec3(0.f); bs.eta = 0.f; bs.delta = false; bs.type = 0u; bs.comp = 0u;
    if (MODE == MODE_PATH || active_next) bsdf_sample<TYPES>(S, side, bin, side_ok, s1, s2x, s2y, bs);

    /* ---- NEE contribution (path.cpp:271-281, prb.py:210-216); visibility is resolved by the shadow kernel */
    R.contrib = Vec3(0.f); R.dLr_drho = Vec3(0.f); R.dLr_drho_unit = Vec3(0.f); R.rho = bin.slot0;
    if (active_em) {
        float mis_em = ((TYPES & HAR_SCENE_ENVMAP) != 0u && em_delta) ? 1.f : mis_weight(ds.pdf, ev.pdf);
        if (MODE == MODE_PATH) R.contrib = st.throughput * ((ev.value * em_weight) * mis_em);
        else {
            R.contrib = ((st.throughput * mis_em) * ev.value) * em_weight;
            if (MODE == MODE_PRB_ADJOINT) {
                R.dLr_drho = ((st.throughput * mis_em) * ev.d_slot0) * em_weight;
                R.dLr_drho_unit = ((st.throughput * mis_em) * ev.d_slot0) * em_unit;
                /* em_weight = radiance * em_unit for `area` / `constant` emitters (prb.py:198-206, attached eval_emitter_direction) */
                R.nee_emitter = ((P.flags & HAR_SHADE_EMITTER_GRADS) && S.emitters[em_sampled].type != 2u && S.emitters[em_sampled].type != 7u) ? (int32_t) em_sampled : -1;
                R.contrib_unit = ((st.throughput * mis_em) * ev.value) * em_unit;
                if (EXTRA && TYPES != HAR_BSDF_ONLY_DIFFUSE) {
                    BsdfEvalExtra x; bsdf_eval_extra(S, side, bin, side_ok, wo_em, x);
                    const Vec3 w = (st.throughput * mis_em) * em_weight;
                    R.x_dir[0] = w * x.d_alpha_u; R.x_dir[1] = w * x.d_alpha_v; R.x_dir[2] = w * x.d_eta; R.x_dir[3] = w * x.d_k; R.x_dir[4] = w * ev.d_slot1;
                }
            }
        }
        /* prb: a sample whose contribution is ZERO can still have a derivative -- a black albedo (d Lr_dir / d rho = beta mis (cos / pi) em_weight at rho = 0), an emitter whose
         * radiance is zero (d Lr_dir / d radiance = contrib_unit), zero texels of a light's bitmap.  The reference tests the visibility of every sample with a density
         * (scene.cpp:338-346), so those gradients exist there; here their shadow ray is traced too -- by BOTH passes of a backward step, whose conditions must agree (the
         * adjoint pass reads the visibility bytes the primal pass filled) */
        bool grad_item = false;
        if (MODE != MODE_PATH) {
            const Vec3 d0 = ((st.throughput * mis_em) * ev.d_slot0) * em_weight;
            grad_item = d0.x != 0.f || d0.y != 0.f || d0.z != 0.f;
            if ((P.flags & HAR_SHADE_EMITTER_GRADS) || ((TYPES & HAR_SCENE_TEXLIGHT) != 0u && lt_sampled)) {
                const Vec3 cu = ((st.throughput * mis_em) * ev.value) * em_unit;
                grad_item = grad_item || cu.x != 0.f || cu.y != 0.f || cu.z != 0.f;
            }
        }
        if (R.contrib.x != 0.f || R.contrib.y != 0.f || R.contrib.z != 0.f || grad_item) {
            R.item = true; R.item_ray = true;
            spawn_ray_to(si, ds.p, R.sh_o, R.sh_d, R.sh_maxt);
            if (MODE == MODE_PRB_ADJOINT) {
                const uint32_t et = S.emitters[em_sampled].type;
                const bool surface = et == 0u || et == 3u || et == 7u;           /* EmitterFlags::Surface (prb.py:178) */
                /* point / spot lights: neither a surface nor infinite -- prb.py:191-192 re-attaches ds.d = normalize(ds.p - si.p) for them as well (HAR_SHAPE_NEE_POINT /
                 * _SPOT, har_shape_grad.h); the spot's record index rides in the (zero) normal */
                const bool point = (TYPES & HAR_SCENE_ENVMAP) != 0u && et == 4u, spot = (TYPES & HAR_SCENE_ENVMAP) != 0u && et == 5u;
                R.nee_flags = 1u | (surface ? 2u : 0u) | (point ? 16u : 0u) | (spot ? 32u : 0u);
                R.nee_p = (surface || point || spot) ? ds.p : ds.d; R.nee_n = spot ? Vec3(as_f32(em_sampled), 0.f, 0.f) : ds.n; R.cos_em = wo_em.z * side.wo_sign;
                R.nee_w = (st.throughput * mis_em) * em_weight;
            }
        }
    }
    if (MODE == MODE_PRB_ADJOINT && side_ok && side.wi.z > 0.f) R.nee_flags |= 4u | (side.wo_sign < 0.f ? 8u : 0u);      /* HAR_SHAPE_LIT, HAR_SHAPE_FLIPPED */

    /* ---- continue the path (path.cpp:287-331, prb.py:227-252) */
    PathState &N = R.next;
    Vec3 wo_world = si.to_world(bs.wo);
    N.o = offset_p(si, wo_world); N.d = wo_world; N.maxt = HAR_LARGEST;
    N.throughput = st.throughput * bs.weight;
    N.eta = st.eta * bs.eta;
    N.lane = st.lane; N.prev_p = si.p; N.prev_bsdf_pdf = bs.pdf;
    const uint32_t delta_bit = bs.delta ? 1u << 16 : 0u;
    float tmax = hmax3(N.throughput);
    float rr_prob = fminf(tmax * (N.eta * N.eta), .95f);
    bool rr_active, alive;
    if (MODE == MODE_PATH) {
        uint32_t nd = depth + 1u;                                  /* path.cpp:317: depth++ BEFORE the test */
        rr_active = nd >= P.rr_depth;
        bool rr_continue = pcg32_next_float(rng, inc) < rr_prob;
        if (rr_active) N.throughput = N.throughput * rcp_(rr_prob);
        alive = active_next && (!rr_active || rr_continue) && (tmax != 0.f);
        N.flags = nd | delta_bit;
    } else {
        active_next = active_next && (tmax != 0.f);
        rr_active = depth >= P.rr_depth;                           /* prb.py:249: depth not yet incremented */
        if (rr_active) N.throughput = N.throughput * rcp_(rr_prob);
        bool rr_continue = pcg32_next_float(rng, inc) < rr_prob;
        alive = active_next && (!rr_active || rr_continue);
        N.flags = (depth + 1u) | delta_bit;
    }
    N.rng = rng;
    R.alive = alive;

    if (MODE == MODE_PRB_ADJOINT) {
        /* Lr_ind = L * relative_grad(bsdf.eval(si, wo, active_next)) (prb.py:288-297): d/d slot0 = L * (df/dslot0) / f */
        Vec3 wo = si.to_local(wo_world);
        BsdfEval e2; bsdf_eval_pdf<TYPES>(S, side, bin, side_ok && alive, wo, e2);
        R.rel_grad = Vec3(e2.value.x != 0.f ? e2.d_slot0.x / e2.value.x : 0.f, e2.value.y != 0.f ? e2.d_slot0.y / e2.value.y : 0.f,
                          e2.value.z != 0.f ? e2.d_slot0.z / e2.value.z : 0.f);
        R.ind_active = alive && (R.rel_grad.x != 0.f || R.rel_grad.y != 0.f || R.rel_grad.z != 0.f);
        R.bsdf = side.index; R.uv_x = si.uv_x; R.uv_y = si.uv_y;
        if (EXTRA && TYPES != HAR_BSDF_ONLY_DIFFUSE) {
            BsdfEvalExtra x; bsdf_eval_extra(S, side, bin, side_ok && alive, wo, x);
            const Vec3 dv[5] = { x.d_alpha_u, x.d_alpha_v, x.d_eta, x.d_k, e2.d_slot1 };
            R.x_ind = false;
            for (int g = 0; g < 5; ++g) {
                R.x_rel[g] = Vec3(e2.value.x != 0.f ? dv[g].x / e2.value.x : 0.f, e2.value.y != 0.f ? dv[g].y / e2.value.y : 0.f, e2.value.z != 0.f ? dv[g].z / e2.value.z : 0.f);
                R.x_ind = R.x_ind || (alive && (R.x_rel[g].x != 0.f || R.x_rel[g].y != 0.f || R.x_rel[g].z != 0.f));
            }
        }
        R.item = R.item_ray || R.ind_active || (EXTRA && TYPES != HAR_BSDF_ONLY_DIFFUSE && R.x_ind);   /* otherwise the vertex has a zero gradient */
    }
}

} // namespace har
