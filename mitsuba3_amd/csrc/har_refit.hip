/*
 * har_refit.hip -- device refit of a bottom-level BVH8 after a vertex update (har_refit.h has the per-leaf / per-node code and the design notes).
 * Both kernels are streaming passes over small arrays (48 B per triangle record, 80 B per node): HBM-bound, one thread per record, no LDS.  A 1M-triangle
 * BLAS is 48 MB of records + 24 MB of boxes + 15.6 MB of nodes, i.e. tens of microseconds of traffic; the cost of a refit is its 1 + depth launches.
 */
#include "har_refit_launch.h"

namespace har {

namespace {

__global__ void k_refit_triangles(const DMesh *meshes, const float *verts, const uint32_t *faces, TriRec *tris, uint32_t first, uint32_t count, RefitBox *tri_box) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    tri_box[first + i] = refit_triangle(meshes, verts, faces, tris, first + i);
}

/* one level of one BLAS: `order` lists its nodes; the surface areas of the level are summed per wave (DPP-free: a plain shuffle reduction, once per 64 nodes)
 * and added to `area` with one atomic per wave */
__global__ void k_refit_nodes(Node8 *nodes, const uint32_t *order, uint32_t count, const RefitBox *tri_box, RefitBox *node_box, float *area) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    float a = 0.f;
    if (i < count) a = refit_node(nodes, order[i], tri_box, node_box);
    for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
    if ((threadIdx.x & 63u) == 0u && a != 0.f) atomicAdd(area, a);
}

} // namespace

void launch_refit_triangles(hipStream_t s, const DScene &S, uint32_t first, uint32_t count, RefitBox *tri_box) {
    if (!count) return;
    hipLaunchKernelGGL(k_refit_triangles, dim3((count + 255u) / 256u), dim3(256), 0, s, S.meshes, S.verts, S.faces, const_cast<TriRec *>(S.accel.tris), first, count, tri_box);
}
void launch_refit_nodes(hipStream_t s, const DScene &S, const uint32_t *order, uint32_t count, const RefitBox *tri_box, RefitBox *node_box, float *area) {
    if (!count) return;
    hipLaunchKernelGGL(k_refit_nodes, dim3((count + 255u) / 256u), dim3(256), 0, s, const_cast<Node8 *>(S.accel.nodes), order, count, tri_box, node_box, area);
}

} // namespace har
