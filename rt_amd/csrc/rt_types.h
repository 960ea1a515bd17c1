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

// child reference encoding
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
