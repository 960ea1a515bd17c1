// rt_types.h -- device data layout shared by the host builders and the HIP kernels.
#pragma once
#include <cstdint>

namespace rtamd {

// ---- wide BVH node: 4 children, 128 bytes, one L2 line, 128-byte aligned ----
// SoA inside the node so that one 16-byte load brings the same plane of all 4 children:
//   plane[0..2] = min x/y/z of children 0..3, plane[3..5] = max x/y/z,
//   child[4]    = reference per child (see below), aux[4] unused (keeps the node one line).
// A ray picks its near/far plane vectors by direction sign, so no per-child select is needed.
struct WideNode {
    float plane[6][4];
    uint32_t child[4];
    uint32_t aux[4];
};
static_assert(sizeof(WideNode) == 128, "WideNode must be one 128-byte line");

// ---- quantized wide BVH node: 4 children, 64 bytes (two per 128-byte line) ----
// Child boxes are stored as 8-bit offsets from the node's own origin on a power-of-two grid per axis
// (conservative: lo rounded down, hi rounded up), so a node visit moves 4 x 16 bytes through the
// vector memory path instead of 7 x 16:
//   16 B: origin x, y, z (float), grid step of x (float, a power of two)
//   16 B: qlo_x[4], qhi_x[4], qlo_y[4], qhi_y[4]   (one byte per child, child c in byte c)
//   16 B: qlo_z[4], qhi_z[4], grid step of y, grid step of z (floats: the kernel multiplies, it decodes nothing)
//   16 B: child[4]   (same encoding as WideNode::child)
struct QuantNode {
    float origin[3];
    float step_x;
    uint32_t qx_lo, qx_hi, qy_lo, qy_hi;
    uint32_t qz_lo, qz_hi;
    float step_y, step_z;
    uint32_t child[4];
};
static_assert(sizeof(QuantNode) == 64, "QuantNode must be 64 bytes");

// ---- packed scene (RT_OPT_NODE_FORMAT 2): ONE array of 48-byte records, wide nodes and primitives together ----
// A node visit is THREE 16-byte loads instead of four.  What makes the room: the grid steps are exponent bytes, and the
// four child references are not stored but rebuilt from one base and a byte per child.  That needs one address space
// for both kinds of children, so nodes (PackNode) and primitives (DevPrim, unchanged) live in the same array, addressed
// in UNITS of 16 bytes (record r starts at unit 3 r; the root is record 0): the children of a node -- one record per
// inner child, one per primitive of a leaf child, in slot order -- follow each other in one block, so a node's children
// are also its neighbours in memory.
//   16 B: origin x, y, z (float), child bytes (child c in byte c)
//   16 B: qlo_x[4], qhi_x[4], qlo_y[4], qhi_y[4]      (as QuantNode: the same boxes, bit for bit)
//   16 B: qlo_z[4], qhi_z[4], base4 = 4 * (first unit of the child block), grid exponents x | y << 8 | z << 16 (int8 each)
// child byte = (units from the block's start) << 2 | leaf << 1 | (primitives in the leaf - 1); kPackEmpty = unused slot
// reference  = base4 + child byte = unit << 2 | leaf << 1 | (count - 1): two instructions (v_bfe_u32, v_add_u32) per child.
// Limits, checked by the host (build_pack_scene), which otherwise leaves the packed scene empty and format 1 runs:
// leaves of at most two primitives (RT_OPT_LEAF_SIZE <= 2); fewer than 2^28 units = 4 GiB (32-bit byte offsets).
struct PackNode {
    float origin[3];
    uint32_t child_bytes;
    uint32_t qx_lo, qx_hi, qy_lo, qy_hi;
    uint32_t qz_lo, qz_hi;
    uint32_t base4;
    uint32_t exps;
};
static_assert(sizeof(PackNode) == 48, "PackNode must be 48 bytes");
constexpr uint32_t kPackLeafBit = 2u;    // in a packed reference
constexpr uint32_t kPackEmpty = 0xFFu;   // child byte of an unused slot (offset 63 is beyond any block: at most 8 records)
constexpr uint64_t kPackMaxUnits = (uint64_t)1 << 28;

// child reference encoding (formats 0 and 1)
constexpr uint32_t kLeafBit = 0x80000000u;  // leaf: bits 0..27 first primitive (DFS rank), bits 28..30 count-1
constexpr uint32_t kEmptyRef = 0xFFFFFFFFu; // unused slot (its box is inverted and can never be hit)
constexpr uint32_t kLeafStartMask = 0x0FFFFFFFu;
constexpr int kMaxLeafPrims = 8;

// ---- primitive record used by the intersection test: 48 bytes, in DFS leaf order of the DEVICE tree ----
// ("earlier DFS leaf wins a tie", reference pbrt_engine.cu:421-438, is "lower ref_rank wins")
//   a = (v0.x, v0.y, v0.z, e1.x)      sphere: centre, radius
//   b = (e1.y, e1.z, e2.x, e2.y)
//   c = (e2.z, shape (0 tri / 1 sphere), original index, DFS-leaf rank in the REFERENCE tree = tie-break order)
struct DevPrim {
    float a[4];
    float b[4];
    float e2z;
    int32_t shape;
    int32_t orig;
    int32_t ref_rank;
};
static_assert(sizeof(DevPrim) == 48, "DevPrim must be 48 bytes");

// ---- shading record fetched once per hit: 16 bytes, same order ----
struct DevShade {
    float normal[3];
    int32_t mat;
};

// ---- texture coordinates of a triangle, only allocated when the scene has a textured triangle: 32 bytes, same order.
//      u[k], v[k] = texture coordinate of vertex k (Triangle::u / ::v, include/RT/triangle.h:345-354) ----
struct DevUV {
    float u[3];
    float v0;
    float v12[2];
    int32_t tex; // texIdx; -1 = none
    int32_t pad;
};
// one texture: RGBA8 texels in device memory, row 0 first
struct DevTex {
    const unsigned char *texels;
    int32_t w, h;
};

// ---- material, camera: small PODs kept in kernel-argument / constant space ----
struct DevMat {
    int32_t kind;
    float albedo[3];
    float param;
};

struct DevCamera { // reference camera object, include/RT/rt_camera.h:62-69
    float origin[3], lower_left[3], horizontal[3], vertical[3];
    float u[3], v[3], w[3];
    float lens_radius, time0, time1;
};

} // namespace rtamd
