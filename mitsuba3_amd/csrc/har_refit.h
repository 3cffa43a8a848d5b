/*
 * har_refit.h -- refit of a bottom-level BVH8 after its vertices moved (HAR_HD: the HIP kernels of har_refit.hip and the host harness run the same code).
 *
 * The reference rebuilds the acceleration data of the shapes that changed, not the scene (Scene::parameters_changed -> m_accel.rebuild only when a shape is
 * dirty, src/render/scene.cpp:517-540; the OptiX backend re-builds the dirty GAS and refreshes the IAS with stable handles, src/render/scene_optix.inl:351-372).
 * Here a vertex update keeps the TOPOLOGY of the tree (which triangle hangs where) and recomputes everything that depends on positions:
 *   1. refit_triangle: the 48-byte triangle record {p0, e1, e2} of every leaf from the packed vertex buffer, exactly as the builder writes it
 *      (build_blas, har_scene_host.cpp), and the padded box of the triangle;
 *   2. refit_node, deepest level first: the box of a node = union of its children's boxes (leaf slots: the triangle boxes; inner slots: the boxes the
 *      previous level wrote), its frame and the 8-bit child planes through the builder's own node_set_frame / node_quantise_child (har_accel.h).
 * Ray queries stay exact by construction -- the triangle test reads the rewritten records, the boxes only prune and are conservative -- and a refit of
 * geometry that did NOT move reproduces the built arrays bit for bit (tests/test_accel_update_cpu.py).  What degrades when triangles move far is the
 * quality of the tree; the sum of the node surface areas is accumulated per refit so that the caller can decide to rebuild (HarSceneImpl::refit_policy).
 */
#pragma once
#include "har_scene.h"

namespace har {

struct RefitBox { float lo[3], hi[3]; };

/* leaf `i` of the triangle-record array: rewrite the record from the vertex buffer, return the padded box of the triangle (build_blas) */
HAR_HD RefitBox refit_triangle(const DMesh *meshes, const float *verts, const uint32_t *faces, TriRec *tris, uint32_t i) {
    TriRec t = tris[i];
    const DMesh M = meshes[t.shape];
    const uint32_t *f = faces + 4 * ((size_t) M.foff + t.prim);
    float p[3][3];
    for (int k = 0; k < 3; ++k) { const float *v = verts + 8 * ((size_t) M.voff + f[k]); p[k][0] = v[0]; p[k][1] = v[1]; p[k][2] = v[2]; }
    t.p0x = p[0][0]; t.p0y = p[0][1]; t.p0z = p[0][2];
    t.e1x = p[1][0] - p[0][0]; t.e1y = p[1][1] - p[0][1]; t.e1z = p[1][2] - p[0][2];
    t.e2x = p[2][0] - p[0][0]; t.e2y = p[2][1] - p[0][1]; t.e2z = p[2][2] - p[0][2];
    tris[i] = t;
    RefitBox b;
    for (int a = 0; a < 3; ++a) { b.lo[a] = fminf(p[0][a], fminf(p[1][a], p[2][a])); b.hi[a] = fmaxf(p[0][a], fmaxf(p[1][a], p[2][a])); }
    pad_box(b.lo, b.hi);
    return b;
}

/* node `index`: children's boxes -> own box (written to node_box[index]), frame, child planes; returns the surface area of the node's box */
HAR_HD float refit_node(Node8 *nodes, uint32_t index, const RefitBox *tri_box, RefitBox *node_box) {
    Node8 n = nodes[index];
    RefitBox cb[8]; RefitBox nb;
    for (int a = 0; a < 3; ++a) { nb.lo[a] = HAR_INF; nb.hi[a] = -HAR_INF; }
    uint32_t n_inner = 0, n_leaf = 0;
    const uint32_t present = (uint32_t) n.imask | (uint32_t) n.lmask;
    for (int s = 0; s < 8; ++s) {
        if (!(present & (1u << s))) continue;
        cb[s] = (n.lmask & (1u << s)) ? tri_box[n.tri_base + n_leaf++] : node_box[n.child_base + n_inner++];
        for (int a = 0; a < 3; ++a) { nb.lo[a] = fminf(nb.lo[a], cb[s].lo[a]); nb.hi[a] = fmaxf(nb.hi[a], cb[s].hi[a]); }
    }
    if (!present) { node_box[index] = nb; return 0.f; }          /* the node of an empty BLAS */
    node_set_frame(n, nb.lo, nb.hi);
    for (int s = 0; s < 8; ++s) if (present & (1u << s)) node_quantise_child(n, s, cb[s].lo, cb[s].hi);
    nodes[index] = n;
    node_box[index] = nb;
    const float dx = nb.hi[0] - nb.lo[0], dy = nb.hi[1] - nb.lo[1], dz = nb.hi[2] - nb.lo[2];
    return 2.f * (dx * dy + dy * dz + dz * dx);
}

} // namespace har
