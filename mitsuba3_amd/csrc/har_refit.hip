/*
 * har_refit.hip -- device refit of a bottom-level BVH8 after a vertex update (har_refit.h has the per-leaf / per-node code and the design notes).
 * Both kernels are streaming passes over small arrays (48 B per triangle record, 80 B per node): HBM-bound, one thread per record, no LDS.  A 1M-triangle
 * BLAS is 48 MB of records + 24 MB of boxes + 15.6 MB of nodes, i.e. tens of microseconds of traffic; the cost of a refit is its 1 + depth launches.
 */
#include "har_refit_launch.h"
#include "har_vertex_update.h"

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

/* ---- device-resident vertex update (har_vertex_update.h): streaming passes, one thread per vertex / per corner record */
__global__ void k_set_positions(float *verts, const float *positions, uint32_t n, uint32_t *bad) {
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const float x = positions[3 * (size_t) v], y = positions[3 * (size_t) v + 1], z = positions[3 * (size_t) v + 2];
    float *o = verts + 8 * (size_t) v; o[0] = x; o[1] = y; o[2] = z;
    if (!(isfinite(x) && isfinite(y) && isfinite(z))) atomicOr(bad, 1u);          /* reported by the NEXT update call (nothing waits for this launch) */
}
__global__ void k_vertex_normals(float *verts, const uint32_t *faces, const uint32_t *corner_begin, const uint32_t *corners, uint32_t n) {
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < n) vertex_normal(verts, faces, corner_begin, corners, v);
}
/* one thread per 16 bytes of a shading triangle: 6 float4 per face, coalesced stores */
__global__ void k_shading_triangles(const float *verts, const uint32_t *faces, float *shade_tris, uint32_t face_count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 6u * face_count) return;
    const uint32_t f = i / 6u, q = i % 6u;
    const float4 *src = reinterpret_cast<const float4 *>(verts + 8 * (size_t) faces[4 * (size_t) f + (q >> 1)]) + (q & 1u);
    reinterpret_cast<float4 *>(shade_tris)[6 * (size_t) f + q] = *src;
}

/* one block per TLAS leaf record: the exact world-space bound of the record's group vertices (vrange[r] = first vertex, count) under its to_world */
__global__ void k_instance_boxes(const InstRec *recs, const uint2 *vrange, const float *verts, RefitBox *out) {
    __shared__ float red[6][256 / 64];
    const uint32_t r = blockIdx.x;
    const InstRec &I = recs[r];
    const uint2 vr = vrange[r];
    RefitBox b = instance_box_empty();
    for (uint32_t v = threadIdx.x; v < vr.y; v += blockDim.x) instance_box_grow(b, I.to_world, verts + 8 * ((size_t) vr.x + v));
    float m[6] = { b.lo[0], b.lo[1], b.lo[2], -b.hi[0], -b.hi[1], -b.hi[2] };        /* six minima */
    for (int k = 0; k < 6; ++k) {
        for (int off = 32; off > 0; off >>= 1) m[k] = fminf(m[k], __shfl_down(m[k], off, 64));
        if ((threadIdx.x & 63u) == 0u) red[k][threadIdx.x >> 6] = m[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        RefitBox o;
        for (int k = 0; k < 3; ++k) {
            float lo = red[k][0], hi = red[3 + k][0];
            for (uint32_t w = 1; w < blockDim.x / 64u; ++w) { lo = fminf(lo, red[k][w]); hi = fminf(hi, red[3 + k][w]); }
            o.lo[k] = lo; o.hi[k] = -hi;
        }
        instance_box_finish(o);
        out[r] = o;
    }
}

/* new instance transforms (DEVICE, column-major 3 x 4 each) -> the shading records (DInst, by instance) and the TLAS leaf records (InstRec, rec_of[instance] or 0xffffffff) */
__global__ void k_set_instances(DInst *insts, InstRec *recs, const uint32_t *rec_of, uint32_t first, uint32_t count, const float *to_world, uint32_t *bad) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    float m[12], inv[12];
    for (int j = 0; j < 12; ++j) m[j] = to_world[12 * (size_t) k + j];
    if (!affine_inverse(m, inv)) { atomicOr(bad, 2u); return; }          /* reported by the next update call; the old transform stays */
    DInst &D = insts[first + k];
    for (int j = 0; j < 12; ++j) { D.to_world[j] = m[j]; D.to_object[j] = inv[j]; }
    const uint32_t r = rec_of[first + k];
    if (r != 0xffffffffu) for (int j = 0; j < 12; ++j) { recs[r].to_world[j] = m[j]; recs[r].to_object[j] = inv[j]; }
}

} // namespace

void launch_set_instances(hipStream_t s, const DScene &S, const uint32_t *rec_of, uint32_t first, uint32_t count, const float *to_world, uint32_t *bad) {
    if (!count) return;
    hipLaunchKernelGGL(k_set_instances, dim3((count + 63u) / 64u), dim3(64), 0, s, const_cast<DInst *>(S.insts), const_cast<InstRec *>(S.accel.insts), rec_of, first, count, to_world, bad);
}
void launch_instance_boxes(hipStream_t s, const DScene &S, uint32_t n_records, const uint2 *vrange, RefitBox *out) {
    if (!n_records) return;
    hipLaunchKernelGGL(k_instance_boxes, dim3(n_records), dim3(256), 0, s, S.accel.insts, vrange, S.verts, out);
}
void launch_set_positions(hipStream_t s, const DScene &S, uint32_t voff, uint32_t vertex_count, const float *positions, uint32_t *bad) {
    if (!vertex_count) return;
    hipLaunchKernelGGL(k_set_positions, dim3((vertex_count + 255u) / 256u), dim3(256), 0, s, const_cast<float *>(S.verts) + 8 * (size_t) voff, positions, vertex_count, bad);
}
void launch_vertex_normals(hipStream_t s, const DScene &S, uint32_t voff, uint32_t foff, uint32_t vertex_count, const uint32_t *corner_begin, const uint32_t *corners) {
    if (!vertex_count) return;
    hipLaunchKernelGGL(k_vertex_normals, dim3((vertex_count + 255u) / 256u), dim3(256), 0, s, const_cast<float *>(S.verts) + 8 * (size_t) voff, S.faces + 4 * (size_t) foff, corner_begin, corners, vertex_count);
}
void launch_shading_triangles(hipStream_t s, const DScene &S, uint32_t voff, uint32_t foff, uint32_t face_count) {
    if (!face_count || !S.shade_tris) return;
    hipLaunchKernelGGL(k_shading_triangles, dim3((6u * face_count + 255u) / 256u), dim3(256), 0, s, S.verts + 8 * (size_t) voff, S.faces + 4 * (size_t) foff, const_cast<float *>(S.shade_tris) + 24 * (size_t) foff, face_count);
}

void launch_refit_triangles(hipStream_t s, const DScene &S, uint32_t first, uint32_t count, RefitBox *tri_box) {
    if (!count) return;
    hipLaunchKernelGGL(k_refit_triangles, dim3((count + 255u) / 256u), dim3(256), 0, s, S.meshes, S.verts, S.faces, const_cast<TriRec *>(S.accel.tris), first, count, tri_box);
}
void launch_refit_nodes(hipStream_t s, const DScene &S, const uint32_t *order, uint32_t count, const RefitBox *tri_box, RefitBox *node_box, float *area) {
    if (!count) return;
    hipLaunchKernelGGL(k_refit_nodes, dim3((count + 255u) / 256u), dim3(256), 0, s, const_cast<Node8 *>(S.accel.nodes), order, count, tri_box, node_box, area);
}

} // namespace har
