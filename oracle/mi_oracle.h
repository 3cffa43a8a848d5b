/*
 * mi_oracle.h -- C ABI of the CPU ORACLE for the hip_ad_rgb hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is a CPU restatement of the
 * reference's (mitsuba3 v3.9.1, variant llvm_ad_rgb) wavefront path tracing
 * path; it exists so that tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg can check / time the HIP product against it.  Nothing in
 * mitsuba3_amd/ may include, link, import or call it.
 *
 * PARITY STATUS: the real reference cannot be built or imported in this
 * environment (ext/drjit, ext/embree, ext/nanobind are empty; no wheel, no
 * network -- SURVEY.md 8c).  The oracle is therefore pinned ONLY by the
 * asset-free known-answer tests that the reference's own test-suite holds for
 * this path (TEA KATs, diffuse closed form, Cornell pixel (124,36), stairs
 * analytic depth, brute-force == accelerated, AD linearity / finite
 * differences, ImageBlock-vs-NumPy) and by the published PCG32 test vector.
 * Arithmetic that lives in Dr.Jit / Embree (NOT IN TREE: PCG32, sincos
 * polynomial, rcp/rsqrt lowering, dot/cross fma order, bilinear texture
 * fetch, Embree's triangle test) is restated from the published algorithms:
 * for those pieces the header says "parity unpinned".
 *
 * All struct layouts below are plain C, 8-byte aligned, and intentionally
 * identical (field for field) to include/hip_ad_rgb.h so that one set of
 * host arrays can be fed to both the oracle and the product in a test.
 */
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Triangle mesh in the reference's packed layout
 * (include/mitsuba/render/mesh_utils.h:19-34, mesh.h:137-156):
 * 8 x f32 per vertex {pos[3], normal[3], uv[2]}, 4 x u32 per face
 * {v0, v1, v2, flags}. */
typedef struct OrcMesh {
    const float    *vertex_ptr;
    const uint32_t *index_ptr;
    uint32_t vertex_count, face_count;
    uint32_t bsdf;      /* index into OrcSceneDesc::bsdfs */
    int32_t  emitter;   /* index into OrcSceneDesc::emitters or -1 */
    uint32_t flags;     /* bit0: has vertex normals, bit1: has texcoords */
    uint32_t reserved;
} OrcMesh;

typedef struct OrcShapeGroup { uint32_t first_mesh, mesh_count; } OrcShapeGroup;

/* column-major 3x4 affine, like ShapeIR::to_world (scene_ir.h:103-105) */
typedef struct OrcInstance {
    uint32_t group;
    float to_world[12];
    float to_object[12];
} OrcInstance;

/* type 0 diffuse, 1 dielectric, 2 roughconductor, 3 roughplastic (src/bsdfs); two colour slots
 * (reflectance = slot 0, may be a bitmap; reflectance2 = slot 1); flags: bit0 twosided, bit1 GGX,
 * bit2 sample_visible, bit3 roughplastic nonlinear; back = BSDF of the back side of a twosided pair or -1.
 * Field for field the layout of HarBSDF (include/hip_ad_rgb.h). */
typedef struct OrcBSDF {
    uint32_t type;
    int32_t  texture;     /* -1: constant slot 0, else index of bitmap */
    float    reflectance[3];
    uint32_t flags;
    float    reflectance2[3];
    float    alpha_u, alpha_v;
    float    eta;
    float    eta_c[3], k_c[3];
    int32_t  back;
} OrcBSDF;

/* bitmap texture, H x W x 3 f32 (src/textures/bitmap.cpp:175-206); mode: bit 0 filter_type nearest (else bilinear), bits 1-2 wrap_mode 0 repeat / 2 mirror / 4 clamp
 * (field for field HarTexture) */
/* to_uv: BitmapTexture's `to_uv` (src/textures/bitmap.cpp:175), the top two rows of the 3 x 3 homogeneous matrix of a ScalarAffineTransform3f, row-major;
 * all zero = unset (identity) */
typedef struct OrcTexture { const float *data; uint32_t width, height; uint32_t mode, reserved; float to_uv[6]; } OrcTexture;

typedef struct OrcEmitter {
    uint32_t type;        /* 0 = area light on a rectangle, 1 = constant environment (src/emitters/constant.cpp; radiance only),
                             2 = environment map (src/emitters/envmap.cpp): mesh = index of the H x W x 3 image in `textures`,
                             radiance[0] = scale, radiance[1] = mis_compensation (0 / 1), to_world / to_local = emitter transform,
                             3 = area light on a top-level triangle mesh (`mesh`; Mesh::sample_position, src/render/mesh.cpp:1662-1712),
                             4 = point light (src/emitters/point.cpp): radiance = radiant intensity, to_world[9..11] = position,
                             5 = spot light (src/emitters/spot.cpp): radiance = intensity, to_world / to_local = transform and inverse,
                             normal[0] = cutoff_angle, normal[1] = beam_width in degrees (no `texture`),
                             6 = directional light (src/emitters/directional.cpp): radiance = irradiance, to_world = the emitter's transform (light travels along its +z) */
    uint32_t mesh;        /* mesh that carries the emitter */
    float radiance[3];
    float to_world[12];   /* rectangle to_world, column-major 3x4 */
    float normal[3];      /* m_frame.n (rectangle.cpp:118) */
    float inv_area;       /* m_inv_surface_area (rectangle.cpp:123) */
    float to_local[12];   /* inverse of to_world as the reference's Transform tracks it (type 2 only) */
    float sampling_weight; /* Emitter::m_sampling_weight (src/render/emitter.cpp:9) */
    uint32_t radiance_texture; /* type 7 = area light on a rectangle whose `radiance` is a bitmap (area.cpp:74,133-165,185-191): index of the bitmap in `textures`;
                                  `radiance` is not read, to_world / normal as for type 0 */
} OrcEmitter;

typedef struct OrcSceneDesc {
    const OrcMesh       *meshes;    uint32_t mesh_count, top_mesh_count;
    const OrcShapeGroup *groups;    uint32_t group_count, pad0;
    const OrcInstance   *instances; uint32_t instance_count, pad1;
    const OrcBSDF       *bsdfs;     uint32_t bsdf_count, pad2;
    const OrcTexture    *textures;  uint32_t texture_count, pad3;
    const OrcEmitter    *emitters;  uint32_t emitter_count, pad4;
} OrcSceneDesc;

/* perspective sensor + hdrfilm + rfilter, everything already lowered to the
 * matrices the reference keeps (perspective.cpp:174-198) */
typedef struct OrcSensor {
    float sample_to_camera[16];   /* row-major 4x4 */
    float to_world[16];           /* row-major 4x4 (camera -> world) */
    float near_clip, far_clip;
    uint32_t film_width, film_height;
    uint32_t crop_offset_x, crop_offset_y, crop_width, crop_height;
    uint32_t rfilter;             /* 0 box, 1 gaussian, 2 tent, 3 mitchell, 4 catmullrom, 5 lanczos (src/rfilters/ *.cpp) */
    float    rfilter_stddev;      /* parameter 0: gaussian stddev, tent radius, mitchell B, lanczos lobes */
    float    rfilter_param1;      /* parameter 1: mitchell C */
    uint32_t sample_border;       /* Film::sample_border (film.cpp:29-32): render() samples crop_size + 2 * rfilter->border_size() pixels (integrator.cpp:162-165) */
    float    principal_point_offset_x, principal_point_offset_y;   /* perspective.cpp:147-150,213-221 */
    uint32_t projection;          /* 0 = PerspectiveCamera, 1 = OrthographicCamera (src/sensors/orthographic.cpp): sample_to_camera = orthographic_projection(...)^-1 */
} OrcSensor;

typedef struct OrcStats {
    uint64_t paths;
    uint64_t vertices;     /* loop iterations summed over paths (K-bar = vertices/paths) */
    uint64_t closest_rays;
    uint64_t shadow_rays;
} OrcStats;

/* ---- scene ---- */
void *orc_scene_create(const OrcSceneDesc *desc);
void  orc_scene_destroy(void *scene);
/* update a constant reflectance / a texture in place (for finite differences) */
void  orc_scene_set_reflectance(void *scene, uint32_t bsdf, const float rgb[3]);
void  orc_scene_sample_emitter(void *scene, uint32_t n, const float *sample, uint32_t *index, float *weight, float *reused);
void  orc_scene_pdf_emitter(void *scene, uint32_t n, const uint32_t *index, float *pdf);
int   orc_scene_set_emitter_weights(void *scene, const float *weights, uint32_t n);
void  orc_scene_set_texture_to_uv(void *scene, uint32_t texture, const float to_uv[6]);
void  orc_scene_set_texture(void *scene, uint32_t texture, const float *data);

/* ---- Hierarchical2D<Float, 0> (distr_2d.h:370-860) and the environment-map emitter (src/emitters/envmap.cpp) as free functions ---- */
void *orc_hier2d_create(const float *data, uint32_t width, uint32_t height, int normalize);
void  orc_hier2d_destroy(void *h);
void  orc_hier2d_sample(void *h, uint32_t n, const float *sample /*[n][2]*/, float *pos /*[n][2]*/, float *pdf);
void  orc_hier2d_invert(void *h, uint32_t n, const float *pos, float *sample, float *pdf);
void  orc_hier2d_eval(void *h, uint32_t n, const float *pos, float *pdf);
uint32_t orc_hier2d_data(void *h, float *out /* nullable */, uint32_t *level_table /* [n_levels][3]: width, size, offset; nullable */, uint32_t *n_levels);
void *orc_envmap_create(const float *rgb, uint32_t width, uint32_t height, float scale, int mis_compensation, const float to_world[12], const float to_local[12]);
void  orc_envmap_destroy(void *e);
void  orc_envmap_set_bsphere(void *e, const float center[3], float radius);
void  orc_envmap_eval(void *e, uint32_t n, const float *d_world /*[n][3]*/, float *rgb);
void  orc_envmap_sample_direction(void *e, uint32_t n, const float *ref_p, const float *sample, float *d, float *dist, float *pdf, float *weight);
void  orc_envmap_pdf_direction(void *e, uint32_t n, const float *d_world, float *pdf);

/* ---- multi-pass JIT render (integrator.cpp:173-183,276-356): spp / spp_per_pass wavefronts of W*H*spp_per_pass lanes whose
 *      sampler streams continue across passes; [lane_begin, lane_end) index the per-pass wavefront ---- */
int   orc_render_path_passes(void *scene, const OrcSensor *sensor, uint32_t seed, uint32_t spp, uint32_t spp_per_pass, int32_t max_depth,
                             int32_t rr_depth, uint64_t lane_begin, uint64_t lane_end, float *film, OrcStats *stats, int threads);

/* ---- scalar-variant driver (BASELINE config 1, `scalar_rgb`): SamplingIntegrator::render CPU branch
 *      (integrator.cpp:190-274,398-446), Spiral (spiral.cpp:27-73), ImageBlock::put with the discretised filter.
 *      n_threads = pool_size() + 1 of the emulated run (it decides the block size, integrator.cpp:203-214). ---- */
int   orc_render_path_scalar(void *scene, const OrcSensor *sensor, uint32_t seed, uint32_t spp, int32_t max_depth, int32_t rr_depth,
                             uint32_t n_threads, float *film /* H x W x 4, accumulated */, OrcStats *stats, uint32_t *block_size);
void  orc_morton_decode(uint32_t m, uint32_t out[2]);
uint32_t orc_spiral(uint32_t size_x, uint32_t size_y, uint32_t block_size, uint32_t max_blocks, int32_t *out);

/* ---- BSDF / microfacet building blocks (golden-vector checks; src/render/tests/test_microfacet.py,
 *      src/bsdfs/tests/test_dielectric.py, test_twosided.py) ---- */
/* MicrofacetDistribution(type 0 beckmann / 1 ggx, alpha_u, alpha_v, sample_visible): out = {eval(m), pdf(wi, m), smith_g1(wi, m)} */
void  orc_microfacet_eval(int type, float alpha_u, float alpha_v, int sample_visible, const float wi[3], const float m[3], float out[3]);
/* sample(wi, sample) -> m[3], pdf */
void  orc_microfacet_sample(int type, float alpha_u, float alpha_v, int sample_visible, const float wi[3], const float sample[2], float m[3], float *pdf);
/* fresnel(cos_theta_i, eta) -> {r, cos_theta_t, eta_it, eta_ti};  fresnel_conductor(cos, eta, k) */
void  orc_fresnel(float cos_theta_i, float eta, float out[4]);
float orc_fresnel_conductor(float cos_theta_i, float eta, float k);
/* BSDF::eval_pdf / BSDF::sample of scene BSDF `bsdf` (twosided handled): wi, wo local; uv[2] */
void  orc_bsdf_eval_pdf(void *scene, uint32_t bsdf, const float wi[3], const float uv[2], const float wo[3], float value[3], float *pdf);
void  orc_bsdf_sample(void *scene, uint32_t bsdf, const float wi[3], const float uv[2], float sample1, const float sample2[2],
                      float wo[3], float *pdf, float weight[3], float *eta, int *delta);
/* ... with the BSDFContext (mode 0 Radiance / 1 Importance, type_mask, component; include/mitsuba/render/bsdf.h:140-186) and the Mask argument (`active`) of
 * BSDF::eval (which = 0) / pdf (1) / eval_pdf (2) and BSDF::sample, restated per plugin in orc_bsdf_ctx.h.  sampled_type is the BSDFFlags lobe bit. */
void  orc_bsdf_evaluate_ctx(void *scene, uint32_t bsdf, uint32_t mode, uint32_t type_mask, uint32_t component, int which, int active, const float wi[3], const float uv[2],
                            const float wo[3], float value[3], float *pdf);
void  orc_bsdf_sample_ctx(void *scene, uint32_t bsdf, uint32_t mode, uint32_t type_mask, uint32_t component, int active, const float wi[3], const float uv[2], float sample1,
                          const float sample2[2], float wo[3], float *pdf, float weight[3], float *eta, uint32_t *sampled_type, uint32_t *sampled_component);
/* roughplastic precomputation of scene BSDF `bsdf`: out[0..63] external transmittance, out[64] internal reflectance, out[65] specular sampling weight */
void  orc_gauss_legendre(int n, float *nodes, float *weights);      /* quad.h:27-90 */
void  orc_roughplastic_tables(void *scene, uint32_t bsdf, float out[66]);

/* ---- ray queries: Scene::ray_intersect_preliminary / ray_test / _naive ---- */
/* mode: 0 = BVH, 1 = brute force.  rays SoA: o[3][n], d[3][n], maxt[n]. */
void orc_ray_intersect(void *scene, uint32_t n, const float *o, const float *d,
                       const float *maxt, int mode, float *t, float *u, float *v,
                       uint32_t *prim, uint32_t *shape, uint32_t *inst);
void orc_ray_test(void *scene, uint32_t n, const float *o, const float *d,
                  const float *maxt, int mode, uint8_t *hit);

/* ... with the Mask argument (`active`, n bytes, nullable): masked lanes report t = inf / hit = 0 */
void orc_ray_intersect_masked(void *scene, uint32_t n, const float *o, const float *d, const float *maxt, const uint8_t *active, int mode,
                              float *t, float *u, float *v, uint32_t *prim, uint32_t *shape, uint32_t *inst);
void orc_ray_test_masked(void *scene, uint32_t n, const float *o, const float *d, const float *maxt, const uint8_t *active, int mode, uint8_t *hit);

/* ---- integrators ----
 * lanes [lane_begin, lane_end) of the reference's wavefront ordering are
 * rendered; film is the raw H x W x 4 {R,G,B,W} accumulation buffer (added
 * to, not cleared).  Pass 0,0 for "all lanes". */
int orc_render_path(void *scene, const OrcSensor *s, uint32_t seed, uint32_t spp,
                    int32_t max_depth, int32_t rr_depth, uint64_t lane_begin,
                    uint64_t lane_end, float *film, OrcStats *stats, int threads);
/* prb primal (prb.py:68 mode=Primal) */
int orc_render_prb(void *scene, const OrcSensor *s, uint32_t seed, uint32_t spp,
                   int32_t max_depth, int32_t rr_depth, uint64_t lane_begin,
                   uint64_t lane_end, float *film, OrcStats *stats, int threads);
/* RBIntegrator.render_backward (common.py:625-783): grad_in is H x W x 3;
 * grad_reflectance is bsdf_count x 3 (constant albedos), grad_textures[i] is
 * H_i x W_i x 3 per bitmap (may be NULL when texture_count == 0). All added to. */
int orc_render_prb_backward(void *scene, const OrcSensor *s, const float *grad_in,
                            uint32_t seed, uint32_t spp, int32_t max_depth,
                            int32_t rr_depth, float *grad_reflectance,
                            float *const *grad_textures, OrcStats *stats, int threads);
/* ... plus grad_emitters (emitter_count x 3, may be NULL): gradient w.r.t. the radiance of `area` / `constant` emitters
 * (prb.py:160-161 attached emitter.eval, :198-206 attached eval_emitter_direction) */
/* textured area lights (emitter type 7): BitmapTexture::sample_position / pdf_position of the emitter's bitmap, and DiscreteDistribution2D::sample on its own */
int orc_emitter_texture_sample_position(void *scene, uint32_t emitter, const float *sample, uint32_t n, float *uv, float *pdf, int pdf_only);
int orc_discrete_distribution_2d_sample(const float *values, uint32_t w, uint32_t h, const float *sample, uint32_t n, uint32_t *pos, float *pmf, float *reused);
int orc_render_prb_backward_ex(void *scene, const OrcSensor *s, const float *grad_in, uint32_t seed, uint32_t spp, int32_t max_depth,
                               int32_t rr_depth, float *grad_reflectance, float *const *grad_textures, float *grad_emitters,
                               OrcStats *stats, int threads);
/* SamplingIntegrator::sample (integrator.h:432-437) over n caller-supplied rays (SoA 3 x n): path (prb = 0) or the primal prb sample (prb = 1);
 * ray i draws from the stream of wavefront lane lane_offset + i, continued from state[i] if given; rgb 3 x n, valid n, state_out n (nullable) */
int orc_integrator_sample(void *scene, int prb, uint32_t n, const float *o, const float *d, const float *maxt, uint32_t seed, uint32_t lane_offset,
                          const uint64_t *state, int32_t max_depth, int32_t rr_depth, float *rgb, uint8_t *valid, uint64_t *state_out, int threads);
int orc_integrator_sample_masked(void *scene, int prb, uint32_t n, const float *o, const float *d, const float *maxt, uint32_t seed, uint32_t lane_offset,
                                 const uint64_t *state, const uint8_t *active, int32_t max_depth, int32_t rr_depth, float *rgb, uint8_t *valid, uint64_t *state_out, int threads);
/* render_backward plus the gradients of the rough models' `alpha` / `alpha_u` / `alpha_v`, `eta`, `k` (roughconductor.cpp:226-520) and `alpha`,
 * `specular_reflectance` (roughplastic.cpp:244-420): grad_bsdf_params = bsdf_count x 15 floats {alpha_u[3], alpha_v[3], eta[3], k[3], slot1[3]} (per-channel
 * contributions; sum the three for a scalar alpha), added to.  The derivatives of the BSDF value are central differences of a double-precision
 * restatement of the models -- PARITY UNPINNED against the reference (its AD is Dr.Jit's). */
int orc_render_prb_backward_bsdf_params(void *scene, const OrcSensor *s, const float *grad_in, uint32_t seed, uint32_t spp, int32_t max_depth, int32_t rr_depth,
                                        float *grad_reflectance, float *const *grad_textures, float *grad_bsdf_params, OrcStats *stats, int threads);
/* RBIntegrator.render_forward (src/python/python/ad/integrators/common.py:497-623): the forward-mode derivative image of `prb`.  Tangents in
 * the layout of orc_render_prb_backward_ex's gradient buffers (tangent_emitters may be NULL); film = raw H x W x 4 accumulation of the lanes'
 * differential radiance dL = sum over vertices <d Lo / d theta, tangent> (prb.py:313), orc_film_develop(film) is the gradient image */
int orc_render_prb_forward(void *scene, const OrcSensor *s, uint32_t seed, uint32_t spp, int32_t max_depth, int32_t rr_depth, const float *tangent_reflectance,
                           const float *const *tangent_textures, const float *tangent_emitters, float *film, int threads);
/* the two pieces a rank of a multi-GPU job runs (mitsuba3_amd/distributed.py render_backward_distributed): the weight-only splat of its lane
 * band (film H x W x 4, added to), and the backward pass of lanes [lane_begin, lane_end) (0, 0 = all) against the all-reduced weight film
 * (NULL: computed here over the whole wavefront) */
int orc_render_weights(const OrcSensor *s, uint32_t seed, uint32_t spp, uint64_t lane_begin, uint64_t lane_end, float *film, int threads);
int orc_render_prb_backward_lanes(void *scene, const OrcSensor *s, const float *grad_in, const float *weight_film, uint32_t seed, uint32_t spp,
                                  int32_t max_depth, int32_t rr_depth, uint64_t lane_begin, uint64_t lane_end, float *grad_reflectance,
                                  float *const *grad_textures, float *grad_emitters, OrcStats *stats, int threads);
void orc_scene_set_emitter_radiance(void *scene, uint32_t emitter, const float rgb[3]);
/* Integrator property `hide_emitters` (src/render/integrator.cpp:29; path.cpp:114-115,177-190; prb.py:112-118,146-148) for every later render */
void orc_scene_set_hide_emitters(void *scene, int hide);
/* forward renders return the alpha channel of an `rgba` film in all three colour channels: 1 where the sample's ray is valid
 * (path.cpp:114-115,307-308,341; prb.py:332 `depth != 0`), 0 elsewhere, filtered like the radiance */
void orc_scene_set_alpha_only(void *scene, int on);
/* ... plus the gradient w.r.t. the VERTEX POSITIONS of the meshes with pos_mask[m] != 0 (prb.py:124-141 attached surface interaction,
 * :176-216 emitter sampling from the attached point, :261-297 attached wo and solid-angle-to-area Jacobian): grad_positions[m] holds 3
 * doubles per vertex and is added to.  Restated over forward-mode dual numbers (orc_dual.h) for `diffuse` BSDFs (plain or inside `twosided`) and flat-shaded
 * top-level meshes; returns -2 / -3 outside that domain.  PARITY UNPINNED: the reference's own shape-gradient tests
 * (src/python/python/tests/test_ad_integrators.py) need Dr.Jit and are not runnable here. */
int orc_render_prb_backward_shape(void *scene, const OrcSensor *s, const float *grad_in, uint32_t seed, uint32_t spp, int32_t max_depth,
                                  int32_t rr_depth, float *grad_reflectance, float *const *grad_textures, const uint8_t *pos_mask,
                                  double *const *grad_positions, OrcStats *stats, int threads);
/* + gradients w.r.t. the `to_world` of instances (Instance::compute_surface_interaction with an attached transform, src/shapes/instance.cpp:150-266:
 * the hit point follows to_world and is re-intersected with the moving tangent plane, normals and uv stay detached): 12 doubles per instance,
 * column-major 3x4, accumulated into; instances with inst_mask[i] == 0 are not differentiated.  `diffuse` BSDFs only (-2 otherwise). */
int orc_render_prb_backward_instances(void *scene, const OrcSensor *s, const float *grad_in, uint32_t seed, uint32_t spp, int32_t max_depth,
                                      int32_t rr_depth, float *grad_reflectance, float *const *grad_textures, const uint8_t *inst_mask,
                                      double *grad_to_world, OrcStats *stats, int threads);
/* params['mesh.vertex_positions'] = ...; params.update(): new positions (3 floats per vertex) of a top-level mesh + acceleration rebuild */
void orc_scene_set_vertex_positions(void *scene, uint32_t mesh, const float *positions);
/* HDRFilm::develop (hdrfilm.cpp:398-399): image[h][w][3] = RGB / (W==0?1:W) */
void orc_film_develop(const float *film, uint32_t width, uint32_t height, float *image);

/* ---- unit-level entry points (KATs) ---- */
void     orc_sample_tea_32(uint32_t v0, uint32_t v1, int rounds, uint32_t out[2]);
float    orc_sample_tea_float32(uint32_t v0, uint32_t v1, int rounds);
double   orc_sample_tea_float64(uint32_t v0, uint32_t v1, int rounds);
void     orc_pcg32_seed(uint64_t initstate, uint64_t initseq, uint64_t state_inc[2]);
uint32_t orc_pcg32_next_uint32(uint64_t state_inc[2]);
float    orc_pcg32_next_float32(uint64_t state_inc[2]);
/* the first n floats lane `lane` of a wavefront sampler seeded with `seed` draws
 * (sampler.cpp:129-148 + independent.cpp:77-97) */
void  orc_sampler_stream(uint32_t seed, uint32_t lane, uint32_t n, float *out);
float orc_rfilter_eval(uint32_t rfilter, float stddev, float x);
float orc_rfilter_eval2(uint32_t rfilter, float param0, float param1, float x);      /* any filter type, with its radius cut-off */
/* ImageBlock::put, coalesced JIT branch (imageblock.cpp:444-540) */
void  orc_film_put(const OrcSensor *s, uint32_t n, const float *pos_x,
                   const float *pos_y, const float *values4, float *film);
/* PerspectiveCamera::sample_ray (perspective.cpp:200-237) on adjusted positions */
void  orc_sensor_sample_ray(const OrcSensor *s, uint32_t n, const float *px,
                            const float *py, float *o, float *d, float *maxt);
/* SmoothDiffuse on a canonical frame (wi given in local coords) */
void  orc_diffuse_eval_pdf(const float refl[3], const float wi[3], const float wo[3],
                           float value[3], float *pdf);
void  orc_diffuse_sample(const float refl[3], const float wi[3], float s1,
                         const float s2[2], float wo[3], float *pdf, float weight[3]);
void  orc_square_to_cosine_hemisphere(const float s[2], float out[3]);
void  orc_square_to_uniform_sphere(const float s[2], float out[3]);              /* warp.h:250-255 */
void  orc_square_to_uniform_disk_concentric(const float s[2], float out[2]);    /* warp.h:54-90 */
void  orc_coordinate_system(const float n[3], float s[3], float t[3]);
void  orc_mesh_compute_normals(uint32_t nv, float *vertices, uint32_t nf, const uint32_t *faces);
float orc_sincos(float x, float *c); /* returns sin */
int orc_default_threads(void);        /* worker threads used for `threads <= 0`: affinity mask capped by the container's CPU quota */
float orc_math_fn(int fn, float x, float y); /* 0 exp, 1 log, 2 erf, 3 atan2(x, y), 4 acos, 5 tan, 6 erfinv, 7 sin, 8 cos (orc_math.h / orc_bsdf.h) */
void  orc_math_fn_array(int fn, uint32_t n, const float *x, const float *y /* nullable */, float *out);
/* full SurfaceInteraction for one hit (tests of Mesh::compute_surface_interaction) */
void  orc_surface_interaction(void *scene, const float o[3], const float d[3],
                              float t, float u, float v, uint32_t prim, uint32_t shape,
                              uint32_t inst, float out[24]);

/* ... with RayFlags (Shading 1, NormalPartials 2, FollowShape 4, DetachShape 8) and the mask: out[33] = p, n, sh_frame.n, .s, .t, wi, uv, t, dp_du, dp_dv, dn_du, dn_dv */
void  orc_surface_interaction_flags(void *scene, const float o[3], const float d[3], float t, float u, float v, uint32_t prim, uint32_t shape, uint32_t inst,
                                    uint32_t ray_flags, int active, float out[33]);

/* ---- independent scene builders (restating util.py:569-703 etc.) ---- */
/* A transform is 32 floats: row-major 4x4 `matrix` followed by the row-major
 * 4x4 `inverse_transpose` (the pair include/mitsuba/core/transform.h keeps). */
void orc_look_at(const float origin[3], const float target[3], const float up[3], float out[32]);
void orc_translate(const float v[3], float out[32]);
void orc_scale(const float v[3], float out[32]);
void orc_rotate(const float axis[3], float angle_deg, float out[32]);
void orc_matmul(const float a[32], const float b[32], float out[32]);   /* affine a * b */
void orc_affine_inverse(const float m[32], float out[32]);
void orc_perspective_sensor(const float to_world[32], double fov_deg, const char *fov_axis,
                            float near_clip, float far_clip, uint32_t width, uint32_t height,
                            uint32_t crop_x, uint32_t crop_y, uint32_t crop_w, uint32_t crop_h,
                            uint32_t rfilter, float stddev, OrcSensor *out);
/* Rectangle / Cube mesh records baked with to_world (rectangle.cpp:131-156,
 * cube.cpp:58-113, mesh.cpp:1160-1215).  vertices: 4*8 / 24*8 floats,
 * faces: 2*4 / 12*4 u32. */
void orc_rectangle(const float to_world[32], float *vertices, uint32_t *faces,
                   float normal[3], float *inv_area);
void orc_cube(const float to_world[32], float *vertices, uint32_t *faces);
void orc_bake_mesh(const float to_world[32], float *vertices, uint32_t vertex_count, uint32_t *faces, uint32_t face_count);   /* mesh_utils.cpp:33-88 */

#ifdef __cplusplus
}
#endif
