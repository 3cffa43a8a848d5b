/* har_scene_host.h -- host-side lowering of a HarSceneDesc into flat arrays + accel
 * (the work of Scene::Scene + SceneAccel::init, src/render/scene.cpp:26-144). */
#pragma once
#include "../../include/hip_ad_rgb.h"
#include "har_scene.h"
#include "har_accel_build.h"
#include <string>
#include <vector>

namespace har {

struct HostTexture { std::vector<float> data; uint32_t w, h, mode = 0; float uvm[6] = { 1.f, 0.f, 0.f, 0.f, 1.f, 0.f }; };      /* mode carries HAR_TEX_HAS_UV_XF when uvm is not the identity */

/* one bottom-level BVH: its nodes are the range [root, root + node_count) of HostScene::nodes (the root first, children behind their parents), its triangle records
 * [first_tri, first_tri + tri_count); refit order = its nodes sorted by depth, deepest first (level_begin[l] .. level_begin[l + 1] of HostScene::refit_order) */
struct BlasInfo {
    uint32_t root = 0, node_count = 0, first_tri = 0, tri_count = 0; float lo[3], hi[3]; bool empty = true;
    uint32_t order_first = 0; std::vector<uint32_t> level_begin;      /* into refit_order, relative to order_first; levels from the deepest to the root */
    uint32_t refits = 0; double built_area = 0.0;                     /* refits since the last build; sum of the node surface areas right after the build (0 = not measured yet) */
};

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
    /* Scene::m_emitter_distr (scene.cpp:120-141): empty when every sampling_weight is 1; else pmf[n] then cdf[n] (DScene::emitter_distr) */
    std::vector<float> emitter_distr; float emitter_sum = 0.f, emitter_norm = 0.f; uint32_t emitter_valid_lo = 0, emitter_valid_hi = 0;
    /* fills the texture / emitter-distribution fields of a DScene whose array pointers the caller has set (device or host copies) */
    void bind_tables(DScene &D, const float *distr_ptr) const {
        D.emitter_distr = emitter_distr.empty() ? nullptr : distr_ptr; D.emitter_sum = emitter_sum; D.emitter_norm = emitter_norm;
        D.emitter_valid_lo = emitter_valid_lo; D.emitter_valid_hi = emitter_valid_hi;
        D.emitter0_valid = emitters.size() == 1 ? 1u : 0u;
        if (D.emitter0_valid) D.emitter0 = emitters[0];
    }
    DTexture device_texture(size_t i, const float *data_ptr) const {
        const HostTexture &t = textures[i]; DTexture d{ data_ptr, t.w, t.h, t.mode, 0u, { t.uvm[0], t.uvm[1], t.uvm[2], t.uvm[3], t.uvm[4], t.uvm[5] } };
        return d;
    }
    uint32_t root = 0;
    bool has_tlas = false;
    Bvh8Stats stats;
    uint32_t blas_depth = 0, tlas_depth = 0;   /* traversal stack need = blas_depth + (has_tlas ? tlas_depth + 1 : 0) */
    uint32_t top_last = 0;     /* Accel::top_last */
    uint32_t top_root = 0xffffffffu, top_first = 0, top_count = 0;   /* two-level scenes: BLAS of the top-level geometry (Accel::top_root) */
    uint32_t stack_need() const { return blas_depth + (has_tlas ? tlas_depth + 1 : 0); }
    /* ---- what the incremental updates need (har_scene_update_instances / _vertices; Scene::parameters_changed -> m_accel.rebuild for the dirty shapes only,
     * src/render/scene.cpp:517-540, scene_optix.inl:351-372) */
    std::vector<HarShapeGroup> groups; std::vector<uint32_t> inst_group;      /* the SceneIR split: shape groups, the group of every instance */
    BlasInfo blas_top; std::vector<BlasInfo> blas_groups;                     /* the bottom-level BVHs */
    std::vector<uint32_t> refit_order;                                        /* node indices of every BLAS by depth (BlasInfo::level_begin) */
    std::vector<PrimBox> inst_boxes; std::vector<uint8_t> inst_box_valid;     /* world-space box of every instance (TLAS leaves), recomputed for the instances that moved */
    uint32_t tlas_first = 0;                                                  /* the TLAS nodes are the tail [tlas_first, nodes.size()) of `nodes` */
    /* device refit of the instance level after a device-resident vertex update of an instanced mesh (har_scene_update_vertices_device): the TLAS nodes by depth, deepest
     * first (tlas_levels[l] .. tlas_levels[l + 1] of tlas_order), re-derived by every build_tlas (tlas_serial counts them); the groups whose BlasInfo::lo / hi no longer
     * bound their vertices because the positions changed on the device only (recompute_stale_group_boxes before the next host build) */
    std::vector<uint32_t> tlas_order, tlas_levels; uint64_t tlas_serial = 0; std::vector<uint8_t> group_box_stale;
};

/* RoughPlastic::parameters_changed (src/bsdfs/roughplastic.cpp:204-242): m_specular_sampling_weight from the means of the colour slots */
void update_roughplastic_sampling_weight(HostScene &hs, uint32_t bsdf);
/* quad::gauss_legendre (include/mitsuba/core/quad.h:27-90): nodes / weights on [-1, 1] */
void quad_gauss_legendre(int n, std::vector<float> &nodes, std::vector<float> &weights);

/* Scene::update_emitter_sampling_distribution (src/render/scene.cpp:120-141) from `weights` (one per emitter): fills emitter_distr / _sum / _norm / _valid_* (empty table =
 * uniform selection: every weight is 1) */
bool build_emitter_distribution(HostScene &hs, const float *weights, uint32_t n, std::string &err);
/* (re)derive the texel distribution of a bitmap radiated by an area light (emitter type 7) into hs.emitter_cdf[off ..]; false: unusable to_uv / no luminance */
bool texel_table_fill(HostScene &hs, const HostTexture &t, uint32_t off, std::string &err);
/* what texel_table_fill needs of a to_uv (2 x 3) and of the texels (w * h * 3), checked BEFORE a setter changes anything */
bool texel_table_inputs_ok(const float uvm[6], const float *texels, uint32_t w, uint32_t h, std::string &err);
/* returns false and fills `err` on invalid input */
bool lower_scene(const HarSceneDesc &desc, HostScene &out, std::string &err);
/* the pieces of lower_scene an update re-runs: bounding sphere of the scene for the environment / directional emitters (constant.cpp:72-87), world boxes of the
 * instances that are not valid + TLAS over them (the BLAS arrays are untouched: hs.nodes is cut back to tlas_first and the new TLAS appended) */
void update_scene_bounds(HostScene &hs);
bool build_tlas(HostScene &hs, std::string &err);
/* host half of an instance update: new transforms for instances [first, first + count) (column-major 3 x 4 each, with their inverses), their boxes invalidated,
 * the TLAS rebuilt, the scene bounds refreshed */
bool scene_set_instances_host(HostScene &hs, uint32_t first, uint32_t count, const float *to_world, const float *to_object, std::string &err);
/* host half of a vertex update of mesh `mesh` (vertex_count x 8 packed records): the vertex buffer is overwritten and the BLAS that holds the mesh returned
 * (NULL + err: the update needs a new scene -- the mesh carries an emitter, whose sampling tables are lowered from the positions).  The caller refits that BLAS
 * (device: har_refit.hip; host harness: refit_blas_host), then calls scene_after_refit_host, which recomputes what hangs on it: the BLAS box, the boxes of the
 * instances of its group + the TLAS, the scene bounds */
BlasInfo *scene_set_vertices_host(HostScene &hs, uint32_t mesh, const float *vertices, std::string &err);
/* alpha_u / alpha_v / eta / eta_c / k_c / reflectance2 of `in` into record `index`, derived quantities (roughplastic table -- rewritten in place --, internal reflectance, sampling weight) with them */
/* the record of delta emitter `index` (point / spot / directional, same type as before) re-lowered in place from `e`; the scene bounds of directional lights follow */
bool scene_set_delta_emitter_host(HostScene &hs, uint32_t index, const HarEmitter &e, std::string &err);
bool lower_plain_emitter(const HarEmitter &e, DEmitter &de, std::string &err);
bool scene_set_bsdf_params_host(HostScene &hs, uint32_t index, const HarBSDF &in, std::string &err);
bool scene_after_refit_host(HostScene &hs, BlasInfo *blas, std::string &err);
/* BlasInfo::lo / hi of the groups marked in group_box_stale from the (refreshed) host vertices; their instances' cached boxes are dropped */
void recompute_stale_group_boxes(HostScene &hs);
/* the instance level refitted on the host arrays with the code of the device path (harness): instance boxes from the vertices, then the TLAS nodes deepest level first */
void refit_tlas_host(HostScene &hs);
/* the refit itself on the host arrays (sequential; what the kernels of har_refit.hip do): returns the sum of the node surface areas */
double refit_blas_host(HostScene &hs, BlasInfo &blas);

/* GaussianFilter ctor (src/rfilters/gaussian.cpp:48-93) + sensor repack */
bool lower_sensor(const HarSensor &in, DSensor &out, std::string &err);

} // namespace har
