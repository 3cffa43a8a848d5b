/* har_refit_launch.h -- launch wrappers of har_refit.hip */
#pragma once
#include <hip/hip_runtime.h>
#include "har_refit.h"

namespace har {

/* triangle records [first, first + count) of S.accel.tris rewritten from S.verts / S.faces; their padded boxes -> tri_box[first ..] */
void launch_refit_triangles(hipStream_t s, const DScene &S, uint32_t first, uint32_t count, RefitBox *tri_box);
/* the nodes order[0 .. count) (one depth level of one BLAS, deepest level first): boxes, frames, child planes; their surface areas are added to *area */
void launch_refit_nodes(hipStream_t s, const DScene &S, const uint32_t *order, uint32_t count, const RefitBox *tri_box, RefitBox *node_box, float *area);

/* device-resident vertex update of one mesh (har_vertex_update.h): positions (DEVICE, 3 floats per vertex) into the packed vertex records (`bad` is OR-ed with 1 when a
 * position is not finite), regenerated vertex normals from the mesh's corner list, the 96-byte shading triangles of its faces */
void launch_set_positions(hipStream_t s, const DScene &S, uint32_t voff, uint32_t vertex_count, const float *positions, uint32_t *bad);
void launch_vertex_normals(hipStream_t s, const DScene &S, uint32_t voff, uint32_t foff, uint32_t vertex_count, const uint32_t *corner_begin, const uint32_t *corners);
void launch_shading_triangles(hipStream_t s, const DScene &S, uint32_t voff, uint32_t foff, uint32_t face_count);
/* world-space boxes of the TLAS leaf records (S.accel.insts[0 .. n_records)): exact bound of the record's group vertices vrange[r] = {first vertex, count} under its to_world */
/* instances [first, first + count): to_world (DEVICE, column-major 3 x 4 each) and its inverse into the shading records and the TLAS leaf records (rec_of[instance]); a singular /
 * non-finite matrix ORs 2 into *bad and leaves its instance as it was */
void launch_set_instances(hipStream_t s, const DScene &S, const uint32_t *rec_of, uint32_t first, uint32_t count, const float *to_world, uint32_t *bad);
void launch_instance_boxes(hipStream_t s, const DScene &S, uint32_t n_records, const uint2 *vrange, RefitBox *out);

} // namespace har
