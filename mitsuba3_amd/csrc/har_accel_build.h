/* har_accel_build.h -- host builder interface of the compressed 8-wide BVH (see har_accel_build.cpp). */
#pragma once
#include "har_accel.h"
#include <vector>

namespace har {

struct PrimBox { float lo[3], hi[3]; };
struct Bvh8Stats { uint32_t max_depth = 0; };

/* conservative padding applied to every primitive box before the build */
void pad_prim_box(PrimBox &b);

/* Appends a BVH8 over `prims` to `nodes`; leaves reference positions
 * leaf_base + k of the primitive sequence appended to `leaf_order` (indices into
 * `prims`).  `max_leaf` (1..3) = primitives per leaf: every primitive of a hit leaf is processed without a box
 * test of its own, so the TLAS (primitive = an instance entry, expensive) is built with 1.  `prim_cost` = cost of processing one
 * primitive relative to one node visit, for the SAH-optimal collapse of the binary tree into 8-wide nodes, which is used for sets of at least
 * `dp_min_prims` primitives (smaller sets: greedy largest-area-first collapse).  Returns the root node index. */
uint32_t build_bvh8(const std::vector<PrimBox> &prims, std::vector<Node8> &nodes, uint32_t leaf_base,
                    std::vector<uint32_t> &leaf_order, Bvh8Stats *stats, uint32_t max_leaf = 3, float prim_cost = 0.3f, uint32_t dp_min_prims = 0);

} // namespace har
