// wide_bvh.h -- turns the reference's binary BVH (post-order rt_bvh_node32 array, root last) into the
// device layout of rt_types.h: 4-wide, 128-byte nodes in breadth-first order, primitives re-ordered
// into DFS leaf order.  The TOPOLOGY is the reference's (each wide node is a 2-level cut of the
// binary tree, chosen largest-surface-area-first); only boxes and packing are device-specific.
#pragma once
#include <string>
#include <vector>

#include "../../include/rt_abi.h"
#include "rt_types.h"

namespace rtamd {

struct WideBvh {
    std::vector<WideNode> nodes; // nodes[0] is the root
    std::vector<QuantNode> qnodes; // the same nodes, quantized (rt_types.h); same indices
    std::vector<DevPrim> prims;  // DFS leaf order of the binary tree
    std::vector<DevShade> shade; // same order
    std::vector<DevUV> uv;       // same order; empty unless some triangle has texIdx != -1
    int depth = 0;               // wide levels below the root node (root = 1)
    int stack_entries = 0;       // worst-case traversal stack entries per ray
    float pad = 0.0f;            // box pad actually applied (box_mode 0)
};

// leaf_size: 1..kMaxLeafPrims primitives per leaf.  box_mode: 0 = re-fitted + one pad, 1 = reference boxes.
// Returns false and fills err if the node array is not a tree over exactly the n primitives.
// ref_rank: per primitive, its DFS-leaf rank in the reference tree (tie order); nullptr = the rank in `nodes` itself.
// collapse: how binary subtrees become 4-wide nodes.  0 = expand the child with the largest half-area until four
// slots are taken.  1 = the partition that minimises the surface-area cost of the wide tree (dynamic programme over the
// binary tree, after Ylitie, Karras, Laine 2017, sec. 4.1).  Either way the result is a conservative hierarchy over the
// same primitives in the same DFS order, so traced results cannot differ.
bool build_wide_bvh(const rt_triangle148 *tris, int n, const rt_bvh_node32 *nodes, int m, int leaf_size, int box_mode,
                    const int32_t *ref_rank, WideBvh &out, std::string &err, int collapse = 0);

// The packed scene of RT_OPT_NODE_FORMAT 2 (rt_types.h, PackNode): wide nodes and primitives in ONE array of 48-byte
// records, every node's children in one block.  Built from a finished WideBvh: the same boxes (the quantised values of
// qnodes, bit for bit), the same primitives, the same tie-break ranks, so traced results cannot differ.
struct PackRecord {
    uint32_t w[12];
};
static_assert(sizeof(PackRecord) == 48 && sizeof(PackRecord) == sizeof(PackNode) && sizeof(PackRecord) == sizeof(DevPrim), "one record size");
struct PackScene {
    std::vector<PackRecord> recs;      // record 0 = the root node; qnodes.size() + prims.size() records
    std::vector<uint32_t> rec_of_prim; // rec_of_prim[p] = record that holds WideBvh::prims[p]
};
// false, with the reason in `why` and `out` empty, when the tree does not fit the format (a leaf of more than two
// primitives, 2^28 units or more, a grid exponent beyond int8): the caller then runs format 1.
bool build_pack_scene(const WideBvh &w, PackScene &out, std::string &why);

// DFS (left first) leaf order of a binary tree: order[rank] = primitive.  Validates that `nodes` is a tree over
// exactly the n primitives.
bool dfs_leaf_order(const rt_bvh_node32 *nodes, int m, int n, std::vector<int32_t> &order, std::string &err);

} // namespace rtamd
