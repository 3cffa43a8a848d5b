/* har_scene_host.h -- host-side lowering of a HarSceneDesc into flat arrays + accel
 * (the work of Scene::Scene + SceneAccel::init, src/render/scene.cpp:26-144). */
#pragma once
#include "../../include/hip_ad_rgb.h"
#include "har_scene.h"
#include "har_accel_build.h"
#include <string>
#include <vector>

namespace har {

struct HostTexture { std::vector<float> data; uint32_t w, h, mode = 0; };

struct HostScene {
    std::vector<float> verts;
    std::vector<float> shade_tris;      /* HAR_SHADING_TRIS builds: three vertex records per face (empty otherwise) */
    std::vector<uint32_t> faces;
    std::vector<DMesh> meshes; uint32_t top_mesh_count = 0;      /* meshes [0, top_mesh_count) are the scene's own shapes, the rest belong to shape groups */
    std::vector<DBsdf> bsdfs;
    std::vector<float> bsdf_tables;          /* roughplastic external transmittance, 64 floats per table */
    std::vector<HostTexture> textures;
    std::vector<DEmitter> emitters;
    std::vector<DInst> insts;
    std::vector<Node8> nodes;
    std::vector<TriRec> tris;
    std::vector<InstRec> inst_recs;
    std::vector<uint32_t> blas_tri_ranges;
    int32_t env_emitter = -1;
    /* environment map (emitter type 2): halo'ed radiance storage, hierarchical warp storage, record without device pointers */
    std::vector<float> env_tex, env_warp; DEnvmap envmap{}; bool has_envmap = false;
    std::vector<float> emitter_cdf; bool has_mesh_emitters = false;      /* face-area tables of the mesh area lights (emitter type 3) */
    bool has_point_emitters = false;                                     /* emitter type 4 (delta position): shaded by the kernels with the generic emitter code */
    uint32_t root = 0;
    bool has_tlas = false;
    Bvh8Stats stats;
    uint32_t blas_depth = 0, tlas_depth = 0;   /* traversal stack need = blas_depth + (has_tlas ? tlas_depth + 1 : 0) */
    uint32_t top_last = 0;     /* Accel::top_last */
    uint32_t top_root = 0xffffffffu, top_first = 0, top_count = 0;   /* two-level scenes: BLAS of the top-level geometry (Accel::top_root) */
    uint32_t stack_need() const { return blas_depth + (has_tlas ? tlas_depth + 1 : 0); }
};

/* RoughPlastic::parameters_changed (src/bsdfs/roughplastic.cpp:204-242): m_specular_sampling_weight from the means of the colour slots */
void update_roughplastic_sampling_weight(HostScene &hs, uint32_t bsdf);
/* quad::gauss_legendre (include/mitsuba/core/quad.h:27-90): nodes / weights on [-1, 1] */
void quad_gauss_legendre(int n, std::vector<float> &nodes, std::vector<float> &weights);

/* returns false and fills `err` on invalid input */
bool lower_scene(const HarSceneDesc &desc, HostScene &out, std::string &err);

/* GaussianFilter ctor (src/rfilters/gaussian.cpp:48-93) + sensor repack */
bool lower_sensor(const HarSensor &in, DSensor &out, std::string &err);

} // namespace har
