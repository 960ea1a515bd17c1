// sah_build.h -- optional device-side topology: a binned-SAH binary tree over the same primitives.
//
// The reference's tree (random-axis median split, bvh_build.h) is always built bit-exactly and is what the ABI
// returns; it also defines the tie order (earlier DFS leaf wins, reference pbrt_engine.cu:421-438).  But DESIGN.md
// section 3 shows that the closest hit does not depend on which conservative hierarchy is traversed, so the
// device may walk a better one.  This builder produces a full binary tree (one primitive per leaf, 2n-1 nodes,
// post-order, root last) in the same rt_bvh_node32 form, ready for build_wide_bvh.
#pragma once
#include <vector>

#include "../../include/rt_abi.h"

namespace rtamd {

void build_sah_bvh(const rt_triangle148 *tris, int n, std::vector<rt_bvh_node32> &out);

} // namespace rtamd
