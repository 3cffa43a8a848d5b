/*
 * hip_ad_rgb.h -- C ABI of the MI355X-native `hip_ad_rgb` hot path
 * (forward `path` + `prb` adjoint on triangle scenes) for Mitsuba 3.
 *
 * This is the drop-in boundary: plain C, `extern "C"`, raw
 * pointers + sizes, no torch / Dr.Jit types.  Every entry point names the
 * reference interface it replaces (paths relative to the mitsuba3 tree).
 * All array arguments marked DEVICE are device pointers in HBM (the caller
 * owns them -- e.g. torch tensors); HOST pointers are ordinary host memory.
 * `stream` is a hipStream_t passed as void* (NULL = default stream).
 *
 * Error convention: every function returns 0 on success, non-zero on failure;
 * har_last_error() returns a thread-local message (the reference throws C++
 * exceptions, src/render/integrator.cpp:40,178).
 *
 * Wavefront data layout: SoA.  A "Ray3f" wavefront of width n is three
 * arrays o[3][n], d[3][n] (component-major) and maxt[n]; a
 * "PreliminaryIntersection3f" is t[n], u[n], v[n], prim_index[n],
 * shape_index[n], inst_index[n] (include/mitsuba/render/interaction.h:717-836).
 */
#pragma once
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------
 *  Scene IR  (replaces SceneIR / ShapeIR, include/mitsuba/render/scene_ir.h:14-176)
 * ---------------------------------------------------------------------- */

/* ShapeIR::Kind::Triangles in the packed Mesh layout
 * (include/mitsuba/render/mesh_utils.h:19-34; Mesh::describe, src/render/mesh.cpp:2797-2813):
 * 8 x f32 per vertex {pos[3], normal[3], uv[2]}, 4 x u32 per face {v0,v1,v2,flags}. HOST pointers. */
typedef struct HarMesh {
    const float    *vertex_ptr;
    const uint32_t *index_ptr;
    uint32_t vertex_count, face_count;
    uint32_t bsdf;      /* index into HarSceneDesc::bsdfs */
    int32_t  emitter;   /* index into HarSceneDesc::emitters, or -1 */
    uint32_t flags;     /* bit0: has vertex normals, bit1: has texcoords */
    uint32_t reserved;
} HarMesh;

/* ShapeGroup: meshes [first_mesh, first_mesh + mesh_count) (src/render/shapegroup.cpp) */
typedef struct HarShapeGroup { uint32_t first_mesh, mesh_count; } HarShapeGroup;

/* InstanceEntry (scene_ir.h:128-138): column-major 3x4 affine + its inverse */
typedef struct HarInstance {
    uint32_t group;
    float to_world[12];
    float to_object[12];
} HarInstance;

/* BSDF record.  Colour parameters live in two "slots":
 *   type 0 diffuse         (src/bsdfs/diffuse.cpp)         slot0 = reflectance
 *   type 1 dielectric      (src/bsdfs/dielectric.cpp)      slot0 = specular_reflectance, slot1 = specular_transmittance, eta
 *   type 2 roughconductor  (src/bsdfs/roughconductor.cpp)  slot0 = specular_reflectance, eta_c / k_c (RGB), alpha_u / alpha_v
 *   type 3 roughplastic    (src/bsdfs/roughplastic.cpp)    slot0 = diffuse_reflectance, slot1 = specular_reflectance, eta, alpha_u
 *   type 4 conductor       (src/bsdfs/conductor.cpp)       slot0 = specular_reflectance, eta_c / k_c (RGB)
 *   type 5 plastic         (src/bsdfs/plastic.cpp)         slot0 = diffuse_reflectance, slot1 = specular_reflectance, eta
 * slot0 may be a bitmap texture (`texture` >= 0), slot1 is constant.
 * flags: bit0 twosided (src/bsdfs/twosided.cpp; `back` = record used for the back side, -1 = the same one),
 *        bit1 GGX distribution (else Beckmann), bit2 sample_visible, bit3 plastic / roughplastic `nonlinear`. */
#define HAR_BSDF_DIFFUSE        0
#define HAR_BSDF_DIELECTRIC     1
#define HAR_BSDF_ROUGHCONDUCTOR 2
#define HAR_BSDF_ROUGHPLASTIC   3
#define HAR_BSDF_CONDUCTOR      4
#define HAR_BSDF_PLASTIC        5
#define HAR_BSDF_TWOSIDED       1u
#define HAR_BSDF_GGX            2u
#define HAR_BSDF_SAMPLE_VISIBLE 4u
#define HAR_BSDF_NONLINEAR      8u
typedef struct HarBSDF {
    uint32_t type;
    int32_t  texture;     /* -1: constant slot0 (srgb), else bitmap index */
    float    reflectance[3];          /* slot0 */
    uint32_t flags;
    float    reflectance2[3];         /* slot1 */
    float    alpha_u, alpha_v;
    float    eta;                     /* int_ior / ext_ior */
    float    eta_c[3], k_c[3];        /* conductor: real and imaginary part of the relative IOR */
    int32_t  back;
} HarBSDF;

/* BitmapTexture, raw H x W x 3 f32 (src/textures/bitmap.cpp:175-206). HOST pointer.  mode = filter_type | wrap_mode (bitmap.cpp:182-206):
 * HAR_TEX_BILINEAR / HAR_TEX_NEAREST, HAR_TEX_REPEAT / HAR_TEX_MIRROR / HAR_TEX_CLAMP; 0 = the reference's defaults (bilinear, repeat). */
#define HAR_TEX_BILINEAR 0u
#define HAR_TEX_NEAREST  1u
#define HAR_TEX_REPEAT   0u
#define HAR_TEX_MIRROR   2u
#define HAR_TEX_CLAMP    4u
/* to_uv: the `to_uv` property of BitmapTexture (bitmap.cpp:175), a 2-D affine map applied to the surface's uv before every lookup (`uv = m_transform * si.uv`,
 * bitmap.cpp:565,792,831,847): row-major 2 x 3, { m00, m01, m02, m10, m11, m12 } -> u' = fma(m01, v, fma(m00, u, m02)), v' likewise (AffineTransform * Point,
 * include/mitsuba/core/transform.h:322-335).  Six zeros (a zero-initialised record) mean the identity. */
typedef struct HarTexture { const float *data; uint32_t width, height; uint32_t mode, reserved; float to_uv[6]; } HarTexture;

/* type 0: AreaLight on a Rectangle (src/emitters/area.cpp, src/shapes/rectangle.cpp:108-179);
 * type 1: ConstantBackgroundEmitter (src/emitters/constant.cpp): only `radiance` is read, at most one per scene.
 * The order of the array is the order of Scene::emitters() (children in declaration order, scene.cpp:40-70). */
typedef struct HarEmitter {
    uint32_t type;        /* 0 = area (src/emitters/area.cpp on a rectangle), 1 = constant (src/emitters/constant.cpp),
                             2 = envmap (src/emitters/envmap.cpp): mesh = index of the H x W x 3 lat-long radiance image in `textures`,
                             radiance[0] = scale, radiance[1] = mis_compensation (0 / 1), to_world / to_local = emitter transform,
                             3 = area on any top-level triangle mesh `mesh` (Mesh::sample_position, src/render/mesh.cpp:1662-1712): radiance only,
                             4 = point (src/emitters/point.cpp): radiance = the radiant intensity, to_world[9..11] = the position (m_position,
                             point.cpp:62-77: `position` or the translation of `to_world`); a delta emitter -- sampled with MIS weight 1, never hit,
                             5 = spot (src/emitters/spot.cpp, without `texture`): radiance = the intensity along the axis, to_world / to_local = the emitter's
                             transform and its inverse, normal[0] = cutoff_angle, normal[1] = beam_width in degrees (update(), spot.cpp:300-312),
                             6 = directional (src/emitters/directional.cpp): radiance = the irradiance, to_world = the emitter's transform -- light travels along
                             its +z axis (`direction` is lowered to look_at(0, direction, up) by the host, directional.cpp:69-78); delta direction, infinite,
                             not an environment emitter (escaping rays do not see it),
                             7 = area on a rectangle whose `radiance` is a BITMAP (area.cpp:74, the spatially varying branches :133-165 and :185-191): `radiance_texture` =
                             index of the bitmap in `textures`, `mesh` / to_world / normal as for type 0; `radiance` is not read.  The texture is importance-sampled
                             (BitmapTexture::sample_position, bitmap.cpp:622-660, over a DiscreteDistribution2D of the texels' luminance) and the uv mapped onto the shape by
                             Rectangle::eval_parameterization (rectangle.cpp:215-237).  The bitmap's to_uv must map the unit square onto itself (bitmap.cpp:976-992).  Its
                             texels are parameters of the scene (har_scene_set_texture re-derives the distribution) but no gradient is produced for them */
    uint32_t mesh;
    float radiance[3];
    float to_world[12];   /* column-major 3x4 */
    float normal[3];
    float inv_area;
    float to_local[12];   /* inverse of to_world as the reference's Transform tracks it (type 2 only) */
    float sampling_weight; /* Emitter property `sampling_weight` (src/render/emitter.cpp:9; default 1 -- set it, a zero-initialised record has weight 0): as soon as one emitter's
                            * weight differs from 1 the scene picks emitters from a DiscreteDistribution over the weights instead of uniformly
                            * (Scene::update_emitter_sampling_distribution, src/render/scene.cpp:120-141; sample_emitter :248-271, pdf_emitter :273-279,
                            * pdf_emitter_direction :378-388).  Weights are non-negative and not all zero. */
    uint32_t radiance_texture; /* type 7 only */
} HarEmitter;

typedef struct HarSceneDesc {
    const HarMesh       *meshes;    uint32_t mesh_count, top_mesh_count;
    const HarShapeGroup *groups;    uint32_t group_count, pad0;
    const HarInstance   *instances; uint32_t instance_count, pad1;
    const HarBSDF       *bsdfs;     uint32_t bsdf_count, pad2;
    const HarTexture    *textures;  uint32_t texture_count, pad3;
    const HarEmitter    *emitters;  uint32_t emitter_count, pad4;
} HarSceneDesc;

/* PerspectiveCamera + HDRFilm + ReconstructionFilter, lowered
 * (src/sensors/perspective.cpp:174-198, src/films/hdrfilm.cpp:241-288) */
typedef struct HarSensor {
    float sample_to_camera[16];   /* row-major 4x4 */
    float to_world[16];           /* row-major 4x4 */
    float near_clip, far_clip;
    uint32_t film_width, film_height;
    uint32_t crop_offset_x, crop_offset_y, crop_width, crop_height;
    uint32_t rfilter;             /* 0 box, 1 gaussian, 2 tent, 3 mitchell, 4 catmullrom, 5 lanczos (src/rfilters/ *.cpp) */
    float    rfilter_stddev;      /* parameter 0: gaussian `stddev`, tent `radius`, mitchell `B`, lanczos `lobes` */
    float    rfilter_param1;      /* parameter 1: mitchell `C` */
    uint32_t sample_border;       /* Film::sample_border (src/render/film.cpp:29-32): != 0 -> the lane -> pixel map of render() runs over the crop window
                                   * enlarged by rfilter->border_size() = ceil(radius - 1/2 - 2 RayEpsilon) pixels on every side
                                   * (src/render/integrator.cpp:162-165, 322-339); the film itself keeps the crop size, splats are clipped to it */
    float    principal_point_offset_x, principal_point_offset_y;   /* PerspectiveCamera `principal_point_offset_x / _y` (src/sensors/perspective.cpp:147-150): sample_ray adds
                                   * film_size * offset / crop_size to the film position before it is taken to the near plane (:213-221) */
    uint32_t projection;          /* 0 = PerspectiveCamera (src/sensors/perspective.cpp), 1 = OrthographicCamera (src/sensors/orthographic.cpp:131-157): sample_to_camera is
                                   * the inverse of orthographic_projection (sensor.h:272-307), an affine map; rays start on the near plane and run along to_world's +z */
} HarSensor;

/* counters of one render call (all lanes), read back with har_render_stats */
typedef struct HarStats {
    uint64_t paths;
    uint64_t vertices;      /* loop iterations (path vertices shaded) */
    uint64_t closest_rays;
    uint64_t shadow_rays;
} HarStats;

typedef struct HarSceneImpl      *HarScene;
typedef struct HarIntegratorImpl *HarIntegrator;

const char *har_last_error(void);
/* returns the gfx arch string of the current device ("gfx950"), or NULL without a GPU */
const char *har_device_arch(void);

/* ------------------------------------------------------------------------
 *  Device memory (the reference allocates through Dr.Jit's caching allocator, jit_malloc: every array of the renderer lives in the process's one pool).
 *  By default the library calls hipMalloc / hipFree.  A host that owns a device allocator installs it here: alloc_fn(bytes, user) returns device memory
 *  usable on any stream of the current device (NULL = out of memory), free_fn(ptr, user) releases a block alloc_fn returned.  Blocks remember the
 *  function that frees them, so the hook may change while blocks are alive; NULL, NULL restores hipMalloc.  INVARIANT: no block is freed while the device runs --
 *  the library synchronises the device before it frees (once per workspace), because a pooling allocator hands a freed block to its next user at once and blocks are
 *  not bound to a stream here.  alloc_fn returning NULL means "out of memory": har_render_backward then steps down to a smaller workspace (record tape -> lane-indexed
 *  replay cache -> smaller chunks) instead of failing.  (mitsuba3_amd installs PyTorch's caching allocator: torch.cuda.caching_allocator_alloc / _delete.)
 * ------------------------------------------------------------------------ */
typedef void *(*HarAllocFn)(size_t bytes, void *user);
typedef void (*HarFreeFn)(void *ptr, void *user);
int har_set_allocator(HarAllocFn alloc_fn, HarFreeFn free_fn, void *user);

/* ------------------------------------------------------------------------
 *  Scene + acceleration structure
 *  replaces Scene::Scene / SceneAccel::{init, rebuild, release}
 *  (src/render/scene.cpp:26-144, include/mitsuba/render/accel.h:36-45,
 *   accel_native.h:26-44; closest analogue: build_metal_accel / release_metal_accel,
 *   src/render/metal/accel.h:27-33)
 * ---------------------------------------------------------------------- */
int har_scene_create(const HarSceneDesc *desc, HarScene *out);
int har_scene_destroy(HarScene scene);
/* SceneParameters update of `<bsdf>.reflectance.value` / `<bsdf>.reflectance.data`
 * (mi.traverse + params.update(), src/python/python/util.py). HOST data. */
int har_scene_set_reflectance(HarScene scene, uint32_t bsdf, const float rgb[3]);
/* ... of the NON-colour parameters of a BSDF record -- `<bsdf>.alpha.value / .alpha_u / .alpha_v`, `.eta.value`, `.k.value`, `.specular_reflectance.value`
 * (RoughConductor::traverse, src/bsdfs/roughconductor.cpp:213-225; RoughPlastic::traverse + parameters_changed, roughplastic.cpp:204-242; SmoothPlastic, plastic.cpp:188-205):
 * alpha_u, alpha_v, eta, eta_c, k_c and reflectance2 of `params` replace the record's (type, flags, texture, slot 0 and back side stay), the derived quantities follow
 * (roughplastic's transmittance table is rewritten in place, internal reflectance, lobe-selection weight).  The scene handle, the acceleration data and every workspace
 * survive: an optimisation loop over a roughness does not rebuild a scene per step.  HOST data, the record is uploaded synchronously. */
int har_scene_set_bsdf_params(HarScene scene, uint32_t bsdf, const HarBSDF *params);
/* ... of the placement / cone of a DELTA emitter -- `<emitter>.position` (PointLight::traverse, src/emitters/point.cpp:84-88), `<emitter>.to_world`, `.cutoff_angle`,
 * `.beam_width` (SpotLight::traverse / parameters_changed -> update, spot.cpp:111-118, 300-312), `<emitter>.to_world` of a directional light (directional.cpp:93-109): the
 * record of emitter `emitter` (HarEmitter type 4, 5 or 6, the type it had) is re-lowered in place from `record`; the scene handle and the acceleration data survive.  Area,
 * environment and mesh lights are part of the scene's geometry / tables and need a new scene.  HOST data. */
int har_scene_set_delta_emitter(HarScene scene, uint32_t emitter, const HarEmitter *record);
int har_scene_set_emitter_radiance(HarScene scene, uint32_t emitter, const float rgb[3]);   /* `area` / `constant` emitters */
int har_scene_set_texture(HarScene scene, uint32_t texture, const float *data);
/* The same updates from DEVICE memory, ordered on `stream` (NULL = default stream), with no host round trip and no synchronisation: the optimisation loop of
 * BASELINE config 4 (render -> loss -> backward -> optimiser step -> params.update(), src/python/python/util.py:344-528; in the reference the scene parameters ARE
 * device arrays).  data / rgb: DEVICE pointers (H x W x 3 floats / 3 floats) that must stay valid until the copy has run on `stream`.  Exception: the colour or bitmap
 * of a `plastic` / `roughplastic` record also sets its lobe-selection weight (the MEAN of the reflectance, RoughPlastic::parameters_changed,
 * roughplastic.cpp:204-242), which is computed on the host: those records take one synchronous device-to-host copy. */
int har_scene_set_texture_device(HarScene scene, uint32_t texture, const float *data, void *stream);
int har_scene_set_reflectance_device(HarScene scene, uint32_t bsdf, const float *rgb, void *stream);
int har_scene_set_emitter_radiance_device(HarScene scene, uint32_t emitter, const float *rgb, void *stream);
/* Scene::sample_emitter(index_sample, active) -> (index, emitter_weight, reused sample) and Scene::pdf_emitter(index, active) (src/render/scene.cpp:248-279), array-valued,
 * DEVICE arrays of n entries: uniform selection with sample re-use, or DiscreteDistribution::sample_reuse_pmf over the emitters' `sampling_weight`s when one of them
 * differs from 1 (:120-141, :258-261).  A scene without emitters returns index 0xffffffff and weight 0 (:251-256); a masked lane zeros. */
int har_scene_sample_emitter(HarScene scene, uint32_t n, const float *index_sample, const uint8_t *active, uint32_t *index, float *weight, float *reused_sample, void *stream);
int har_scene_pdf_emitter(HarScene scene, uint32_t n, const uint32_t *index, const uint8_t *active, float *pdf, void *stream);
/* params['<emitter>.sampling_weight'] + update() (Emitter::traverse, src/render/emitter.cpp:13; Scene::parameters_changed rebuilds the distribution, scene.cpp:523-528):
 * `weights` = HOST array, one per emitter of the scene */
int har_scene_set_emitter_sampling_weights(HarScene scene, const float *weights, uint32_t count);
/* params['<texture>.to_uv'] + update() (BitmapTexture::traverse, src/textures/bitmap.cpp): the 2 x 3 rows of HarTexture::to_uv, HOST */
int har_scene_set_texture_to_uv(HarScene scene, uint32_t texture, const float to_uv[6]);
/* ------------------------------------------------------------------------
 *  Incremental updates of the acceleration data.  The reference rebuilds what a changed shape needs, not the scene: Scene::parameters_changed calls
 *  m_accel.rebuild only when a shape is dirty (src/render/scene.cpp:517-540); the OptiX backend re-builds the dirty geometry and refreshes the instance level
 *  with stable handles (src/render/scene_optix.inl:351-372).  Both calls keep the HarScene handle, every device array and every integrator workspace.
 *   - har_scene_update_instances: new `to_world` / `to_object` (column-major 3 x 4 each, HOST) for instances [first, first + count): the instance level (TLAS,
 *     a host build over <= instance_count boxes) is rebuilt and rewritten in place, the bottom-level BVHs are not touched;
 *   - har_scene_update_vertices: new packed vertex records (HOST, vertex_count x 8 floats as in HarMesh::vertex_ptr; the caller has regenerated the normals,
 *     Mesh::parameters_changed, mesh.cpp:876-878) of mesh `mesh`: the BLAS that holds it is REFITTED on the device -- triangle records rewritten, node boxes
 *     re-quantised bottom-up with the builder's own arithmetic -- and the instance level rebuilt if the mesh is instanced.  Ray queries stay exact (the boxes
 *     only prune); what a refit cannot do is re-sort triangles that moved far, so the call watches the tree's cost (sum of node areas / root area):
 *       0                           done
 *       HAR_UPDATE_REBUILD_ADVISED  done -- the scene is valid -- but the cost has grown by more than HAR_REFIT_MAX_INFLATION (default 1.5x) against the tree as built at the last
 *                                   build (or HAR_REFIT_MAX_STEPS refits have passed): a new scene would trace faster
 *       HAR_UPDATE_NEEDS_NEW_SCENE  not done: the mesh carries an area emitter (har_last_error says so); create a new scene
 *       1                           error
 *  Both synchronise `stream` before they return (their sources are pageable host memory). */
#define HAR_UPDATE_NEEDS_NEW_SCENE 2
#define HAR_UPDATE_REBUILD_ADVISED 3
int har_scene_update_instances(HarScene scene, uint32_t first, uint32_t count, const float *to_world, const float *to_object, void *stream);
int har_scene_update_vertices(HarScene scene, uint32_t mesh, const float *vertices, void *stream);
/* The same update with the positions ALREADY ON THE DEVICE -- what Mesh::parameters_changed does in the reference's JIT variants, where a position update never leaves
 * the GPU (src/render/mesh.cpp:848-899: pack(regenerate_normals) -> compute_normals :1216-1267; Scene::parameters_changed hands the accel its new vertices,
 * src/render/scene.cpp:517-540).  `positions` = DEVICE, vertex_count x 3 floats (the layout of '<shape>.positions', Mesh::traverse).  Enqueued on `stream`:
 * positions -> packed vertex records, vertex normals regenerated if the mesh carries normals (Mesh::compute_normals as a deterministic per-vertex gather,
 * har_vertex_update.h), the 96-byte shading triangles rewritten, the BLAS refitted.  For a top-level mesh in a scene without environment / directional emitters the
 * call copies nothing between host and device and waits for nothing: the refit's cost figure and a "position not finite" flag land in a pinned record that the NEXT
 * update call reads, so HAR_UPDATE_REBUILD_ADVISED (and the error for a non-finite position) arrive ONE CALL LATE.  A mesh inside a shape group also has the instance
 * level REFITTED on the device (the exact world-space bounds of its instances, then the TLAS nodes; topology kept) -- no copy, no wait either.  Only a scene whose
 * environment / directional emitters follow the scene's bounding sphere additionally reads the mesh's vertex records back (device -> host) for the host and waits.
 * The host mirror of the mesh is refreshed lazily (har_scene_get_vertices, or any later call that needs it).  Return codes as above. */
int har_scene_update_vertices_device(HarScene scene, uint32_t mesh, const float *positions, void *stream);
/* New `to_world` of instances [first, first + count) ALREADY ON THE DEVICE (Instance::parameters_changed of a JIT variant, src/shapes/instance.cpp:79-91): `to_world` = DEVICE,
 * count x 12 floats (column-major 3 x 4 each, the layout of HarInstance::to_world).  Enqueued on `stream`: the inverses (formed in double, rounded once), the shading and TLAS leaf
 * records, the instances' exact world-space bounds, a REFIT of the instance level (its topology stays that of the last host build) -- no copy, no wait; a singular / non-finite
 * matrix leaves its instance as it was and is reported by the NEXT call.  Scenes whose environment / directional emitters follow the scene's bounding sphere take the host path
 * (har_scene_update_instances) after reading the matrices back.  har_scene_get_instances: the transforms as the device holds them (HOST out, 12 floats each). */
int har_scene_update_instances_device(HarScene scene, uint32_t first, uint32_t count, const float *to_world, void *stream);
int har_scene_get_instances(HarScene scene, uint32_t first, uint32_t count, float *to_world, float *to_object, void *stream);
/* the packed vertex records (HOST out, vertex_count x 8 floats) of `mesh` as the device holds them -- after device-resident updates the only current copy */
int har_scene_get_vertices(HarScene scene, uint32_t mesh, float *vertices, void *stream);
/* info[0] = refits since the scene was created, info[1] = cost figure of the last refitted BLAS, info[2] = its ratio to the figure at the first refit, info[3] = nodes */
int har_scene_refit_info(HarScene scene, double info[4]);
/* accel statistics: node count, triangle count, bytes */
int har_scene_accel_info(HarScene scene, uint64_t info[4]);

/* Every array-valued call below takes the reference's `Mask active` as `const uint8_t *active` (DEVICE, n bytes; NULL = all lanes active).
 * A masked lane does no work and returns what the reference's masked lane returns: "no intersection" (t = inf) / false / zero value and weight;
 * it never advances a sampler stream. */

/* Scene::ray_intersect_preliminary (src/render/scene.cpp:216-230) /
 * Scene::ray_intersect_naive (:240-244, naive != 0: brute-force kernel).  DEVICE arrays.
 * (`coherent`, `reorder`, `reorder_hint`, `reorder_hint_bits` of the reference are scheduling hints for its backends, DRJIT_MARK_USED only.) */
int har_ray_intersect_preliminary(HarScene scene, uint32_t n, const float *o, const float *d,
                                  const float *maxt, const uint8_t *active, int naive, float *t, float *u, float *v,
                                  uint32_t *prim_index, uint32_t *shape_index,
                                  uint32_t *inst_index, void *stream);
/* Scene::ray_test (src/render/scene.cpp:232-238).  hit: u8[n]. DEVICE arrays. */
int har_ray_test(HarScene scene, uint32_t n, const float *o, const float *d, const float *maxt,
                 const uint8_t *active, int naive, uint8_t *hit, void *stream);
/* RayFlags (include/mitsuba/render/interaction.h:19-87).  Minimal: t, p, n only; Shading (= Default): + uv, dp_du, dp_dv, sh_frame, wi;
 * NormalPartials (needs Shading): + dn_du, dn_dv.  FollowShape / DetachShape choose which AD dependence the reference attaches to the hit
 * ("no effect in non-differentiable variants"): accepted, the values are the same; both at once are refused like every unknown bit. */
#define HAR_RAY_MINIMAL         0u
#define HAR_RAY_SHADING         1u
#define HAR_RAY_NORMAL_PARTIALS 2u
#define HAR_RAY_FOLLOW_SHAPE    4u
#define HAR_RAY_DETACH_SHAPE    8u
#define HAR_RAY_DEFAULT         HAR_RAY_SHADING
/* rows of a SurfaceInteraction3f wavefront (SoA, n floats per row; interaction.h:345-420) */
#define HAR_SI_ROWS 33   /* p 0-2, n 3-5, sh_frame.n 6-8, sh_frame.s 9-11, sh_frame.t 12-14, wi 15-17, uv 18-19, t 20, dp_du 21-23, dp_dv 24-26, dn_du 27-29, dn_dv 30-32 */
/* PreliminaryIntersection::compute_surface_interaction(ray, ray_flags, active) (interaction.h:804-829,
 * src/render/mesh.cpp:2255-2437, src/shapes/instance.cpp:150-266, finalize_surface_interaction interaction.h:559-605).
 * out: HAR_SI_ROWS x f32 SoA [33][n]; fields the flags do not ask for are zero.  A lane that is masked or holds no
 * intersection: t = inf, zero fields, wi = -d (with Shading).  DEVICE arrays. */
int har_compute_surface_interaction(HarScene scene, uint32_t n, const float *o, const float *d,
                                    const float *t, const float *u, const float *v,
                                    const uint32_t *prim_index, const uint32_t *shape_index,
                                    const uint32_t *inst_index, uint32_t ray_flags, const uint8_t *active, float *out, void *stream);
/* Scene::ray_intersect(ray, ray_flags, coherent, ..., active) (src/render/scene.cpp:197-214): the preliminary intersection (t .. inst_index, also
 * outputs) expanded into the SurfaceInteraction `si` ([33][n]); naive != 0: Scene::ray_intersect_naive (:240-244) */
int har_ray_intersect(HarScene scene, uint32_t n, const float *o, const float *d, const float *maxt, uint32_t ray_flags, const uint8_t *active, int naive,
                      float *t, float *u, float *v, uint32_t *prim_index, uint32_t *shape_index, uint32_t *inst_index, float *si, void *stream);

/* ------------------------------------------------------------------------
 *  Sampler  (src/render/sampler.cpp:129-148, src/samplers/independent.cpp:77-97)
 * ---------------------------------------------------------------------- */
/* PCG32Sampler::seed(seed, wavefront_size): state/inc are u64[n] DEVICE arrays;
 * lane i of the wavefront gets the stream of global lane `lane_offset + i`. */
int har_sampler_seed(uint32_t seed, uint32_t lane_offset, uint32_t n, uint64_t *state,
                     uint64_t *inc, void *stream);
/* IndependentSampler::next_1d(active): advances lanes with active[i] != 0 (active may be NULL) */
int har_sampler_next_1d(uint32_t n, uint64_t *state, const uint64_t *inc, const uint8_t *active,
                        float *out, void *stream);
/* IndependentSampler::next_2d: out is [2][n] */
int har_sampler_next_2d(uint32_t n, uint64_t *state, const uint64_t *inc, const uint8_t *active,
                        float *out, void *stream);

/* ------------------------------------------------------------------------
 *  BSDF  (include/mitsuba/render/bsdf.h:322-465, src/bsdfs/ *.cpp)
 *  wi/wo in the local shading frame, SoA [3][n]; uv [2][n]; DEVICE arrays.
 * ---------------------------------------------------------------------- */
/* BSDFContext (include/mitsuba/render/bsdf.h:140-186).  NULL = BSDFContext(): Radiance, every lobe type, every component.
 * mode: dielectric's transmitted weight carries eta_ti^2 only under Radiance (dielectric.cpp:362-367); type_mask / component select lobes through
 * BSDFContext::is_enabled (:177-181) -- component indices: dielectric 0 reflection, 1 transmission; roughplastic 0 glossy, 1 diffuse; plastic 0 delta
 * reflection, 1 diffuse; the one-lobe models 0; `twosided` lists the front BSDF's components first, then the back's (twosided.cpp:86-99,129-146). */
#define HAR_TRANSPORT_RADIANCE   0u
#define HAR_TRANSPORT_IMPORTANCE 1u
#define HAR_LOBE_DIFFUSE_REFLECTION 0x02u   /* BSDFFlags (bsdf.h:31-80) */
#define HAR_LOBE_GLOSSY_REFLECTION  0x08u
#define HAR_LOBE_DELTA_REFLECTION   0x20u
#define HAR_LOBE_DELTA_TRANSMISSION 0x40u
#define HAR_LOBE_ALL                0x1ffu
typedef struct HarBSDFContext { uint32_t mode, type_mask, component; } HarBSDFContext;
/* BSDF::eval_pdf / eval / pdf(ctx, si, wo, active) (bsdf.h:375-465) */
int har_bsdf_eval_pdf(HarScene scene, uint32_t bsdf, const HarBSDFContext *ctx, uint32_t n, const float *wi, const float *uv,
                      const float *wo, const uint8_t *active, float *value /*[3][n]*/, float *pdf, void *stream);
int har_bsdf_eval(HarScene scene, uint32_t bsdf, const HarBSDFContext *ctx, uint32_t n, const float *wi, const float *uv,
                  const float *wo, const uint8_t *active, float *value /*[3][n]*/, void *stream);
int har_bsdf_pdf(HarScene scene, uint32_t bsdf, const HarBSDFContext *ctx, uint32_t n, const float *wi, const float *uv,
                 const float *wo, const uint8_t *active, float *pdf, void *stream);
/* BSDF::sample(ctx, si, sample1, sample2, active) -> (BSDFSample3f, weight) (bsdf.h:322-373): wo, pdf, weight, and -- each may be NULL --
 * BSDFSample3f::eta, ::sampled_type (a BSDFFlags lobe bit; Delta lobes are HAR_LOBE_DELTA_*), ::sampled_component.  sample1 may be NULL (zeros). */
int har_bsdf_sample(HarScene scene, uint32_t bsdf, const HarBSDFContext *ctx, uint32_t n, const float *wi, const float *uv,
                    const float *sample1, const float *sample2 /*[2][n]*/, const uint8_t *active, float *wo /*[3][n]*/,
                    float *pdf, float *weight /*[3][n]*/, float *eta, uint32_t *sampled_type, uint32_t *sampled_component, void *stream);

/* ------------------------------------------------------------------------
 *  Sensor / film
 * ---------------------------------------------------------------------- */
/* PerspectiveCamera::sample_ray (src/sensors/perspective.cpp:200-237) */
int har_sensor_sample_ray(const HarSensor *sensor, uint32_t n, const float *pos_x,
                          const float *pos_y, float *o, float *d, float *maxt, void *stream);
/* ImageBlock::put (src/render/imageblock.cpp:187-540): film is H x W x 4 {R,G,B,W} */
int har_film_put(const HarSensor *sensor, uint32_t n, const float *pos_x, const float *pos_y,
                 const float *values4 /*[n][4]*/, float *film, void *stream);
/* HDRFilm::develop (src/films/hdrfilm.cpp:301-404): image H x W x 3 = RGB / (W==0 ? 1 : W) */
int har_film_develop(const float *film, uint32_t width, uint32_t height, float *image, void *stream);
/* ... with the film's `pixel_format` (hdrfilm.cpp:149-176, develop :326-395): HAR_PIXEL_RGB -> H x W x 3; HAR_PIXEL_Y -> H x W x 1, luminance(rgb)
 * (include/mitsuba/core/spectrum.h:439-442); HAR_PIXEL_XYZ -> H x W x 3, srgb_to_xyz(rgb) (spectrum.h:402-410).  The conversion is applied to the weighted
 * sums, the division by the weight comes last, as in the reference.  The alpha channel of `rgba` / `luminance_alpha` / `xyza` films is a second film
 * (har_integrator_set_alpha_film) developed as A / W by the caller. */
#define HAR_PIXEL_RGB 0
#define HAR_PIXEL_Y   1
#define HAR_PIXEL_XYZ 2
int har_film_develop_format(const float *film, uint32_t width, uint32_t height, int pixel_format, float *image, void *stream);

/* ------------------------------------------------------------------------
 *  Integrators
 * ---------------------------------------------------------------------- */
#define HAR_INTEGRATOR_PATH 0   /* src/integrators/path.cpp */
#define HAR_INTEGRATOR_PRB  1   /* src/python/python/ad/integrators/prb.py */
/* MonteCarloIntegrator ctor (src/render/integrator.cpp:539-550): max_depth -1 = infinite,
 * rr_depth > 0.  chunk_lanes = wavefront chunk size (0 = default). */
int har_integrator_create(int type, int32_t max_depth, int32_t rr_depth, uint32_t chunk_lanes,
                          HarIntegrator *out);
int har_integrator_destroy(HarIntegrator integrator);

/* SamplingIntegrator::render (src/render/integrator.cpp:151-396, JIT branch) /
 * ADIntegrator.render (common.py:46-110) restricted to lanes [lane_begin, lane_end)
 * of the W*H*spp wavefront (0,0 = all): splats into `film` (DEVICE, H x W x 4,
 * accumulated, not cleared, not developed) so that tiles rendered by several
 * GPUs can be summed with one reduce before har_film_develop.
 * Multi-pass rendering (integrator.cpp:173-183,276-356; `path` only): when `samples_per_pass` is set, or W*H*spp exceeds
 * 2^32 - 1, the job runs as n_passes wavefronts of W*H*spp_per_pass lanes whose sampler streams CONTINUE from pass to
 * pass; lane_begin/lane_end then index the per-pass wavefront (har_render_pass_layout tells its size). */
int har_render(HarScene scene, HarIntegrator integrator, const HarSensor *sensor, uint32_t seed,
               uint32_t spp, uint64_t lane_begin, uint64_t lane_end, float *film, void *stream);

/* `samples_per_pass` property of SamplingIntegrator (integrator.cpp:140-147); 0 = unset */
int har_integrator_set_samples_per_pass(HarIntegrator integrator, uint32_t samples_per_pass);
/* Integrator property `hide_emitters` (src/render/integrator.cpp:29): camera rays pass through area emitters (Integrator::skip_area_emitters,
 * integrator.cpp:96-124; path.cpp:177-190, prb.py:112-118) and do not see the environment (path.cpp:114-115, prb.py:146-148) */
int har_integrator_set_hide_emitters(HarIntegrator integrator, int hide);
/* Alpha channel of `rgba` films (HDRFilm pixel_format, hdrfilm.cpp:135-160): when a DEVICE buffer of H x W x 4 floats (crop window) is set,
 * har_render also accumulates  w * alpha  of every sample into its channel 3 -- alpha = 1 for a valid camera sample: PathIntegrator's valid_ray
 * (path.cpp:114-115,307-308,341), `depth != 0` for prb (prb.py:332) -- with the same reconstruction-filter weights w as the radiance, so that
 * A = channel 3 / W of the film (hdrfilm.cpp:398-399).  NULL switches it off. */
int har_integrator_set_alpha_film(HarIntegrator integrator, float *alpha_film);
/* Film WINDOW of har_render (multi-GPU bands): with row_count > 0 the `film` (and alpha film) buffers of the following har_render calls hold only rows
 * [row_begin, row_begin + row_count) of the crop window -- row_count x crop_width x 4 floats -- instead of the whole film.  A rank that renders a band of pixel rows
 * needs its band plus the reconstruction filter's reach on either side (and the sample border); har_render checks that the lanes it is given cannot splat outside the
 * window and fails otherwise.  row_count = 0: the whole film again (default).  A 4096^2 film is 256 MiB; an eighth of it plus the halo is what a rank of eight owns. */
int har_integrator_set_film_window(HarIntegrator integrator, uint32_t row_begin, uint32_t row_count);
/* ------------------------------------------------------------------------
 *  Multi-GPU render from ONE host thread (SURVEY.md section 8e behind the C ABI).  The reference's contract is a single Integrator::render call from one host
 *  thread (include/mitsuba/render/integrator.h:74-79) and it has no multi-GPU path; a C++ host that binds this library as a variant (INTEGRATION.md Route A) reaches
 *  N GPUs through a GROUP: per device a replica of the scene, an integrator with its workspace and a stream.  har_multi_render deals the pixel rows of the sample
 *  grid to the devices as contiguous bands with all their samples (global lane indices: the union of the bands draws the samples of a single-GPU render), adds the
 *  private films on devices[0] with ONE collective -- ncclReduce (RCCL, looked up at run time) inside one ncclGroupStart / End over ncclCommInitAll's communicators;
 *  peer copies + add kernels when a device is named more than once or RCCL is not loadable -- and develops the film there.  Bands are re-cut from the measured device
 *  times of the first frames (read one frame late: nothing waits).  The call returns when everything is enqueued; `image` / `film` are valid in `stream` order on
 *  devices[0].  (mitsuba3_amd/distributed.py is the same partitioning as one process per GPU under torch.distributed: what bench.py --gpus N runs.)
 *    har_multi_create   builds the replicas (scene description as for har_scene_create, integrator as for har_integrator_create) on devices[0 .. n_devices)
 *    har_multi_replica  the k-th replica's handles, for parameter updates (har_scene_set_* / har_scene_update_* with that device current) and statistics
 *    har_multi_render   image: DEVICE (devices[0]), H x W x 3 (x 1 for HAR_PIXEL_Y), or NULL; film: DEVICE (devices[0]), H x W x 4 accumulated RGBW, or NULL
 *    har_multi_render_backward  the prb adjoint over the group (below)
 *    har_multi_info     band_rows[n_devices + 1] boundaries of the current bands, band_ms[n_devices] device time of the last measured frame, which collective is in use */
typedef struct HarMultiImpl *HarMulti;
int har_multi_create(const HarSceneDesc *desc, int integrator_type, int32_t max_depth, int32_t rr_depth, uint32_t chunk_lanes, const int *devices, uint32_t n_devices,
                     HarMulti *out);
int har_multi_destroy(HarMulti group);
int har_multi_replica(HarMulti group, uint32_t k, HarScene *scene, HarIntegrator *integrator, int *device);
int har_multi_render(HarMulti group, const HarSensor *sensor, uint32_t seed, uint32_t spp, int pixel_format, float *image, float *film, void *stream);
/* RBIntegrator.render_backward (src/python/python/ad/integrators/common.py:625-783) over the group (integrator_type HAR_INTEGRATOR_PRB): every device splats the filter weights
 * of its band, ONE all-reduce makes W[px] complete everywhere (the adjoint of develop divides by it), every device replays its band -- primal + adjoint pass -- into ONE flat gradient
 * buffer, ONE reduce brings the buffers to devices[0], which adds them to the caller's: grad_in = DEVICE (devices[0]), H x W x 3; grad_reflectance (bsdf_count x 3), grad_textures
 * (HOST array of texture_count DEVICE pointers, H_t x W_t x 3 each, NULL entries skipped), grad_emitters (emitter_count x 3, or NULL: no emitter gradients) = DEVICE (devices[0])
 * buffers the call ACCUMULATES into, as har_render_backward does.  Its bands are cut separately from har_multi_render's (the adjoint's cost profile is not the forward render's).
 * Vertex-position / instance / BSDF-parameter gradients are not routed through the group. */
int har_multi_render_backward(HarMulti group, const HarSensor *sensor, const float *grad_in, uint32_t seed, uint32_t spp, float *grad_reflectance, float *const *grad_textures,
                              float *grad_emitters, void *stream);
int har_multi_info(HarMulti group, uint32_t *n_devices, uint32_t *band_rows, float *band_ms, char *reduce, uint32_t reduce_len);
/* the band arithmetic on its own (host only, no device): bounds[n + 1] = the current row boundaries of n bands over `rows` rows, seconds[n] = what each band cost ->
 * out[n + 1] = boundaries of equal measured cost (every band keeps at least one row; unchanged when a time is not positive).  The same numbers as BandBalancer.update of
 * mitsuba3_amd/distributed.py, which the multi-process route uses: tests/test_distributed_cpu.py holds the two to each other. */
int har_band_rebalance(uint32_t rows, uint32_t n, const uint32_t *bounds, const double *seconds, uint32_t *out);

/* the pass split har_render will use for `spp` samples per pixel of this sensor's crop window; fails like the reference
 * when spp is not a multiple of the pass size (integrator.cpp:177-179, sampler.cpp:93-94) */
int har_render_pass_layout(HarIntegrator integrator, const HarSensor *sensor, uint32_t spp, uint32_t *spp_per_pass, uint32_t *n_passes);

/* RBIntegrator.render_backward (common.py:625-783), split so that the weight
 * image can be reduced across GPUs between the two calls:
 *  1. har_render_weights: splat W=1 of lanes [begin,end) into film (channel 3 only);
 *  2. har_render_backward: given the (reduced) weight film and grad_in (H x W x 3),
 *     run the primal pass + adjoint replay of lanes [begin,end) and accumulate
 *     gradients into grad_reflectance (bsdf_count x 3) and grad_textures[i]
 *     (H_i x W_i x 3; HOST array of DEVICE pointers, entries may be NULL). */
int har_render_weights(const HarSensor *sensor, uint32_t seed, uint32_t spp, uint64_t lane_begin,
                       uint64_t lane_end, float *film, void *stream);
int har_render_backward(HarScene scene, HarIntegrator integrator, const HarSensor *sensor,
                        const float *grad_in, const float *weight_film, uint32_t seed,
                        uint32_t spp, uint64_t lane_begin, uint64_t lane_end,
                        float *grad_reflectance, float *const *grad_textures, void *stream);

/* Gradient w.r.t. the radiance of `area` / `constant` emitters (prb.py:160-161,198-206 with the emitter attached): when a DEVICE buffer of
 * emitter_count x 3 floats is set, har_render_backward also accumulates into it; NULL switches it off again */
int har_integrator_set_grad_emitters(HarIntegrator integrator, float *grad_emitters);

/* Gradient w.r.t. VERTEX POSITIONS (params['<mesh>.positions'] of mi.traverse): the geometry-attached part of PRBIntegrator.sample
 * (src/python/python/ad/integrators/prb.py:124-141 attached surface interaction -- Mesh::compute_surface_interaction with AD-attached
 * vertices, src/render/mesh.cpp:2286-2323, and SurfaceInteraction::attach_motion, include/mitsuba/render/interaction.h:525-545 --,
 * :176-216 emitter sampling from the attached point, :261-297 attached outgoing direction and solid_angle_to_area_jacobian,
 * ad/integrators/common.py:1355-1384).  `grad_positions` = HOST array of mesh_count DEVICE pointers (HarSceneDesc::meshes: the top-level meshes,
 * then the meshes of the shape groups); entry m (vertex_count x 3 floats) makes mesh m differentiable, NULL entries do not; har_render_backward then also accumulates into those buffers.  A NULL array switches the
 * feature off.  Like `prb` itself this has no visibility-boundary term (that is prb_reparam / the projective integrators).
 * A mesh INSIDE a shape group moves all its instances at once (object-space positions; Instance::compute_surface_interaction with a detached to_world,
 * src/shapes/instance.cpp:150-204); like the reference (:162-166) this cannot be combined with har_integrator_set_grad_instances.
 * The differentiated meshes are flat-shaded, or carry the vertex normals a position update regenerates (Mesh::compute_normals, mesh.cpp:876-878,
 * 1216-1267: the gradient then runs through the interpolated normal and the angle-weighted normal sums of the whole one-ring) -- and carry a BSDF with a non-delta lobe (diffuse, roughconductor, roughplastic, plastic; plain
 * or inside `twosided`) -- the attached si.wi / wo reach the BSDF value (prb.py:128-140, 276-288); the other meshes of the scene may carry any BSDF.  Fails otherwise.
 * The scene may be lit by ANY emitter of this variant (prb.py:176-216 is one code path for all of them): `area` on rectangles / triangle meshes / with a bitmap radiance
 * (EmitterFlags::Surface: re-attached ds.d + Jacobian), `point` and `spot` (neither surface nor infinite: ds.d = normalize(ds.p - si.p) re-attached, :191-192; the point
 * light's squared_norm(ds.p - it.p) follows it.p, point.cpp:155-165; the spot's falloff follows ds.d while its rcp(ds.dist) is detached, spot.cpp:252-274), `directional`,
 * `constant`, `envmap` (EmitterFlags::Infinite: nothing re-attached), with or without sampling weights.
 * New vertex positions are installed in place: har_scene_update_vertices (host records) / har_scene_update_vertices_device (positions already on the device); the BLAS is refitted. */
int har_integrator_set_grad_positions(HarIntegrator integrator, HarScene scene, float *const *grad_positions);

/* Gradient w.r.t. the `to_world` of INSTANCES (params['<instance>.to_world']): Instance::compute_surface_interaction with an attached transform
 * (src/shapes/instance.cpp:150-266) inside PRBIntegrator.sample -- the nested interaction is detached (:181-189), the hit point follows
 * to_world (:191-193) and is put back onto the ray through the moving tangent plane (:240-249); normals and uv stay detached (:194-224, 250-251).
 * `grad_to_world` = DEVICE buffer of instance_count x 12 floats (column-major 3x4 like HarInstance::to_world; the reference's 4x4 has a constant
 * fourth row); har_render_backward then also accumulates into it.  NULL switches the feature off.  Can be combined with
 * har_integrator_set_grad_positions (as in the reference, NOT for the meshes of the instanced shape groups themselves, instance.cpp:162-166).
 * Like `prb` itself: no visibility-boundary term.  The instanced meshes carry BSDFs with a non-delta lobe (as above); fails otherwise. */
int har_integrator_set_grad_instances(HarIntegrator integrator, HarScene scene, float *grad_to_world);

/* RBIntegrator.render_forward (src/python/python/ad/integrators/common.py:497-623): the forward-mode derivative image of the `prb` integrator.
 * tangent_reflectance (DEVICE, bsdf_count x 3), tangent_textures (HOST array of texture_count DEVICE pointers, H_i x W_i x 3 each; NULL when the
 * scene has no bitmaps) and tangent_emitters (DEVICE, emitter_count x 3, may be NULL) are the dr.set_grad() values of the scene parameters, in
 * the layout of har_render_backward's gradient buffers.  Lanes [lane_begin, lane_end) (0, 0 = all) splat their differential radiance
 * dL = sum over vertices <d Lo / d theta, tangent> (prb.py:313) into `film` (DEVICE, H x W x 4, accumulated like har_render's);
 * har_film_develop(film) is the gradient image.  Parameters: colour slot 0 of every BSDF (constant or bitmap), radiance of `area` / `constant`
 * emitters.  Vertex positions: use har_render_backward. */
int har_render_forward(HarScene scene, HarIntegrator integrator, const HarSensor *sensor, uint32_t seed, uint32_t spp, uint64_t lane_begin,
                       uint64_t lane_end, const float *tangent_reflectance, const float *const *tangent_textures, const float *tangent_emitters,
                       float *film, void *stream);

/* BASELINE config 1 (`scalar_rgb`, "plumbing, no GPU"): SamplingIntegrator::render's NON-JIT branch on the host
 * (src/render/integrator.cpp:190-274 block size + seed scaling, src/render/spiral.cpp:27-73, integrator.cpp:398-446 render_block: Morton pixel
 * order, per-pixel reseed; :448-520 render_sample; ImageBlock::put's scalar branch with the discretised filter, src/render/imageblock.cpp:228-375,
 * include/mitsuba/core/rfilter.h:70-79; put_block :160-186) for the `path` integrator.  A separate, explicitly named CPU entry point that runs the
 * same path code as the kernels (host compilation of the HAR_HD headers) with the scalar variants' sampler semantics; the hip_ad_rgb entry points
 * never call it and never fall back to it.  desc / sensor as for har_scene_create / har_render; film = HOST memory, H x W x 4 {R, G, B, W},
 * accumulated; block_size 0 = the reference's choice for n_threads workers (0 = all host cores), returned in *block_size_used (may be NULL).
 * `hide_emitters`, `rgba` films and multi-pass rendering are not part of this entry point. */
int har_render_scalar(const HarSceneDesc *desc, const HarSensor *sensor, uint32_t seed, uint32_t spp, int32_t max_depth, int32_t rr_depth,
                      uint32_t block_size, uint32_t n_threads, float *film, uint32_t *block_size_used);

/* SamplingIntegrator::sample(scene, sampler, ray, medium, aovs, active) -> (Spectrum, Mask), array-valued
 * (include/mitsuba/render/integrator.h:432-437; PathIntegrator::sample src/integrators/path.cpp:94-346, PRBIntegrator.sample(mode=Primal)
 * src/python/python/ad/integrators/prb.py:68-339): n rays in (DEVICE, SoA: o, d = 3 x n floats, maxt = n floats), radiance out (rgb = 3 x n) and
 * the returned mask (valid, n bytes, may be NULL).  The `sampler` argument is the PCG32 wavefront sampler of Sampler::seed(seed, .) restricted to
 * lanes [lane_offset, lane_offset + n): `state` = NULL starts the freshly seeded streams, otherwise ray i continues from state[i] (as produced
 * by har_sampler_seed / har_sampler_next_* for the same seed and lane); `state_out` (may be NULL, `path` only) receives the states after the
 * call -- a lane that starts a loop iteration draws all of that iteration's numbers (JIT loop semantics, path.cpp:247,263-264,323).
 * `path` returns select(valid, L, 0) (path.cpp:341-345), `prb` returns L and valid = depth != 0 (prb.py:332).  No medium, no AOVs.
 * `active` (n bytes, NULL = all): a masked ray never enters the loop -- zero radiance, valid = 0, and state_out = the state it came in with. */
int har_integrator_sample(HarScene scene, HarIntegrator integrator, uint32_t seed, uint32_t lane_offset, uint32_t n, const float *o, const float *d,
                          const float *maxt, const uint64_t *state, const uint8_t *active, float *rgb, uint8_t *valid, uint64_t *state_out, void *stream);
/* Sampler::clone (include/mitsuba/render/sampler.h:89-99): copies the n PCG32 streams (DEVICE arrays) -- both samplers then produce the same
 * numbers (RBIntegrator.render_backward's `sampler.clone()`, common.py:755).  Sampler::fork (sampler.h:78-87) has no device state: a forked
 * sampler is a new host object that is seeded later.  Sampler::advance (sampler.h:109-115; independent.cpp:69-72) only moves the host-side
 * sample / dimension indices of the independent sampler: har_sampler_advance leaves the streams as they are and exists so that a binding can
 * forward the virtual call. */
int har_sampler_clone(uint32_t n, const uint64_t *state, const uint64_t *inc, uint64_t *state_dst, uint64_t *inc_dst, void *stream);
int har_sampler_advance(uint32_t n, uint64_t *state, const uint64_t *inc, void *stream);

/* `prb`: gradients of the NON-colour-slot-0 parameters of the rough BSDF models (src/bsdfs/roughconductor.cpp:226-520 `alpha` / `alpha_u` /
 * `alpha_v`, `eta`, `k`; src/bsdfs/roughplastic.cpp:244-420 `alpha`, `specular_reflectance`) for the following har_render_backward calls.
 * grad = DEVICE buffer of bsdf_count x 15 floats, ACCUMULATED into: per BSDF record five groups of three channel contributions --
 * [0..2] alpha_u (sum the three for the scalar; roughplastic's single `alpha` is reported here), [3..5] alpha_v, [6..8] eta (RGB), [9..11] k (RGB),
 * [12..14] colour slot 1 (roughplastic.specular_reflectance).  NULL switches it off (default).  Needs the replay cache (default) and
 * max_depth <= 12; delta lobes and the `eta` of plastic / rough plastic carry no gradient (as in the reference). */
int har_integrator_set_grad_bsdf_params(HarIntegrator integrator, float *grad);

/* `prb`: gradients w.r.t. the TEXELS of a bitmap `radiance` of area lights (src/emitters/area.cpp:64-70: `radiance` is a differentiable traverse entry; :83-90 eval at si.uv,
 * :133-165 sample_direction with the bitmap evaluated at the sampled ds.uv) for the following har_render_backward calls: d Le / d radiance(si.uv) at emitter hits (prb.py:160-161),
 * d Lr_dir / d radiance(ds.uv) at visible emitter samples with the sampling density detached (prb.py:174-175, 203-206), both through the transpose of the bitmap lookup, ACCUMULATED
 * into the light's bitmap's entry of `grad_textures` (which har_render_backward takes for every texture of the scene).  0 switches it off (default).  Needs the replay cache
 * (default) and max_depth <= 12; not combined with vertex-position gradients; reverse mode only. */
int har_integrator_set_grad_light_texels(HarIntegrator integrator, int on);
/* counters of the last har_render / har_render_backward on this integrator (synchronises) */
int har_render_stats(HarIntegrator integrator, HarStats *out);
/* HIP-event timing of the render calls ("frames") issued since har_integrator_set_profiling(.., 1): one event per kernel launch, recorded on
 * the stream of the launch.  Every frame records into its own event set (a ring of 32 sets; an event is never re-recorded while an earlier
 * record may be pending), so frames may be enqueued back-to-back without synchronisation.  har_render_timing waits for the recorded frames and
 * returns the AVERAGE PER FRAME: ms[0]=raygen ms[1]=trace_closest ms[2]=shade ms[3]=resolve (shadow rays) ms[4]=splat ms[5]=total ms[6]=other,
 * launches[i] = launches of that class per frame, ms[7] = number of frames averaged.  set_profiling(.., 1) restarts the statistics. */
int har_integrator_set_profiling(HarIntegrator integrator, int enable);
/* `prb` only: keep the primal pass's ray-query results (24 B hit + 1 B visibility per lane and bounce, first 12 bounces) in
 * HBM and reuse them in the adjoint replay of the same chunk instead of tracing every ray twice (default: enabled).
 * The gradients are identical either way: the replayed rays are bit-identical to the primal ones. */
int har_integrator_set_replay_cache(HarIntegrator integrator, int enable);
/* Per-material shading queues for `path` and the primal pass of `prb` (the reference dispatches per BSDF through the virtual calls of
 * src/integrators/path.cpp:233,266-267): after the closest-hit launch of a bounce the paths are dealt to one index list per BSDF model and every
 * model is shaded by its own kernel.  Results are identical to the default (one kernel with a block-local material sort).  Default: OFF -- measured
 * slower on MI355X (the class kernels gather path state through sparse index lists). */
int har_integrator_set_material_queues(HarIntegrator integrator, int enable);
/* Wave-shared BVH descent for the FIRST closest-hit launch of a render (the camera rays): at >= 64 samples per pixel the 64 lanes of a wave are samples of one
 * pixel (lane = pixel * spp + sample, integrator.cpp:322-334) and walk the acceleration structure together -- one conservative box test per child for the whole
 * wave, exact per-ray triangle tests at the leaves (mesh.h:1130-1155), so the intersections are those of the per-ray kernels bit for bit; packets that turn out
 * incoherent fall back to the per-ray kernel.  mode -1 (default): automatic (renders with spp a multiple of 64); 0: off; 1: on for every first launch, also for
 * the rays of har_integrator_sample (testing). */
int har_integrator_set_packet_tracing(HarIntegrator integrator, int mode);
int har_render_timing(HarIntegrator integrator, float ms[8], uint32_t launches[8]);


/* ------------------------------------------------------------------------
 *  Host-side plugin lowering (no GPU involved).  A Transform4f is 32 floats:
 *  row-major 4x4 `matrix` followed by row-major 4x4 `inverse_transpose`, the pair
 *  include/mitsuba/core/transform.h keeps.
 * ---------------------------------------------------------------------- */
/* Transform4f::translate / scale / rotate / look_at (transform.h:132-203) */
int har_transform_translate(const float v[3], float out[32]);
int har_transform_scale(const float v[3], float out[32]);
int har_transform_rotate(const float axis[3], float angle_deg, float out[32]);
int har_transform_look_at(const float origin[3], const float target[3], const float up[3], float out[32]);
/* Transform4f::operator* (affine branch, transform.h:364-400) and inverse() (:81-84) */
int har_transform_mul(const float a[32], const float b[32], float out[32]);
int har_transform_inverse(const float a[32], float out[32]);
/* PerspectiveCamera ctor + update_camera_transforms (src/sensors/perspective.cpp:137-198),
 * parse_fov (src/render/sensor.cpp:142-190), HDRFilm crop window, rfilter type + its first parameter (rfilter_param1 is set to 1/3, mitchell's default C) */
/* OrthographicCamera (src/sensors/orthographic.cpp:104-121 update_camera_transforms; sensor.h:272-307 orthographic_projection): fills `out` like
 * har_perspective_sensor (to_world may carry a scale -- it sets the size of the view) */
int har_orthographic_sensor(const float to_world[32], float near_clip, float far_clip, uint32_t width, uint32_t height,
                            uint32_t crop_offset_x, uint32_t crop_offset_y, uint32_t crop_width, uint32_t crop_height,
                            uint32_t rfilter, float rfilter_stddev, HarSensor *out);
int har_perspective_sensor(const float to_world[32], double fov, const char *fov_axis, float near_clip,
                           float far_clip, uint32_t width, uint32_t height, uint32_t crop_x,
                           uint32_t crop_y, uint32_t crop_w, uint32_t crop_h, uint32_t rfilter,
                           float stddev, HarSensor *out);
/* Rectangle::initialize (src/shapes/rectangle.cpp:108-156): 4 vertex + 2 face records baked with
 * to_world, plus m_frame.n and m_inv_surface_area for area-light sampling */
int har_shape_rectangle(const float to_world[32], int flip_normals, float vertices[32], uint32_t faces[8],
                        float normal[3], float *inv_area);
/* Cube ctor (src/shapes/cube.cpp:58-113): 24 vertex + 12 face records baked with to_world */
int har_shape_cube(const float to_world[32], float vertices[192], uint32_t faces[48]);
/* Mesh::transform + flip_winding (src/render/mesh.cpp:1160-1215) on packed records, in place */
int har_mesh_transform(const float to_world[32], uint32_t vertex_count, float *vertices,
                       uint32_t face_count, uint32_t *faces, int has_normals);

/* ------------------------------------------------------------------------
 *  Mesh files (host side).  PLYMesh ctor (src/shapes/ply.cpp:113-345): ASCII / binary little- / big-endian PLY with
 *  triangle faces -> packed records (8 f32 per vertex, 4 u32 per face, mesh_utils.h:19-34) allocated with malloc;
 *  flags bit0 = has vertex normals (stored or regenerated unless face_normals), bit1 = has texcoords.
 *  Like the reference's loaders (PackedMesh::set_transform, src/render/mesh_utils.cpp:33-44,101-133) they bake `to_world`
 *  (har_transform_* layout: matrix + inverse, NULL = identity) and `flip_normals` into the records BEFORE normals are regenerated.
 *  har_mesh_compute_normals = Mesh::compute_normals (src/render/mesh.cpp:1218-1267), in place.
 * ---------------------------------------------------------------------- */
typedef struct HarMeshData { float *vertices; uint32_t *faces; uint32_t vertex_count, face_count, flags, reserved; } HarMeshData;
int  har_mesh_load_ply(const char *filename, int face_normals, int flip_tex_coords, const float *to_world /*[32] or NULL*/, int flip_normals, HarMeshData *out);
/* OBJMesh ctor (src/shapes/obj.cpp:98-296) + Mesh::from_corners (src/render/mesh_utils.cpp:210-560): polygons are fan-triangulated,
 * corners of a point weld when normal / texcoord / UV-orientation agree; missing normals are regenerated per surface point */
int  har_mesh_load_obj(const char *filename, int face_normals, int flip_tex_coords, const float *to_world, int flip_normals, HarMeshData *out);
/* SerializedMesh ctor + load_legacy / load_v5 (src/shapes/serialized.cpp:225-450): container versions 3, 4 and 5 (the packed records as the reference writes them
 * today, Mesh::write_serialized), sub-mesh `shape_index`.  face_normals: 0 / 1, or -1 = the property is unset (a version-5 file's stored FaceNormals flag then applies) */
int  har_mesh_load_serialized(const char *filename, int shape_index, int face_normals, const float *to_world, int flip_normals, HarMeshData *out);
int  har_mesh_compute_normals(uint32_t vertex_count, float *vertices, uint32_t face_count, const uint32_t *faces);
void har_mesh_free(HarMeshData *mesh);

/* ------------------------------------------------------------------------
 *  Image files (host side): HDRFilm::write (src/films/hdrfilm.cpp:414-560) -> Bitmap::write.  image = H x W x C float32.
 *  EXR: OpenEXR 2 scanline, uncompressed, FLOAT channels R, G, B [, A] or Y; PFM: "PF"/"Pf", little endian.
 * ---------------------------------------------------------------------- */
int  har_image_write_exr(const char *filename, const float *image, uint32_t width, uint32_t height, uint32_t channels);
int  har_image_write_pfm(const char *filename, const float *image, uint32_t width, uint32_t height, uint32_t channels);
/* Bitmap(filename) (src/core/bitmap.cpp read_exr / read_pfm) for the environment-map emitter: scanline OpenEXR (NO / ZIPS / ZIP compression,
 * HALF / FLOAT / UINT channels R G B [A] or Y) and PFM -> malloc'ed H x W x C float32, C = 1, 3 or 4 */
typedef struct HarImage { float *data; uint32_t width, height, channels, reserved; } HarImage;
int  har_image_read(const char *filename, HarImage *out);
void har_image_free(HarImage *image);

#ifdef __cplusplus
}
#endif
