// rt_abi.cpp -- the C ABI of include/rt_abi.h over the host builders and the HIP kernels.
// Host code, compiled by hipcc with -ffp-contract=off (the camera basis is float arithmetic that must
// match the reference's camera::update operation for operation).
#include "../../include/rt_abi.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "bvh_build.h"
#include "kernels.h"
#include "rt_shared_math.h"
#include "rt_types.h"
#include "sah_build.h"
#include "wide_bvh.h"

#pragma clang fp contract(off)

using namespace rtamd;

struct rt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // NX/NY/NS, include/RT/pbrt_engine.h:24-26 ; ITER_NUM, pbrt_engine.cu:458
    int nx = 640, ny = 360, ns = 10, max_depth = 50;
    int honor_vfov = 0, leaf_size = 2, box_mode = 0, shard_rank = 0, shard_world = 1, kernel_variant = 2;
    int shade_batch = 32, swap_batch = 8, tail_split = 0, tail_taper = 150; // tail_split 0: by frame size
    uint32_t *d_ring_state = nullptr; // round records, kRingWords words per pixel slot
    size_t ring_items = 0;
    uint64_t ring_generation = 0;     // frame-id generation (next_frame_tag) the records were last zeroed in
    unsigned long long seed_salt = 0;
    int raw_sum = 0;
    int queue_threshold = 48; // k_trace_queue: hand out rays when fewer lanes than this still traverse
    int queue_waves = 24;     // k_trace_queue: persistent waves per CU, at most
    int primary_waves = 16;   // primary rays: persistent waves per CU, at most
    int primary_chunk = 256;  // primary rays: pixel slots per band-counter atomic
    int collapse = 1;         // how binary subtrees become wide nodes (wide_bvh.h)
    int deterministic = 1;    // RT_OPT_DETERMINISTIC
    // analysis builds: rays recorded by the last render (rt_test_dump_rays)
    float *d_ray_dump = nullptr;
    unsigned long long *d_ray_dump_count = nullptr;
    unsigned long long ray_dump_cap = 0;
    unsigned long long *d_wave_times = nullptr; // analysis builds: rt_test_wave_times
    size_t wave_times_cap = 0;                  // waves
    // textures (lambertian albedo of triangles with texIdx != -1)
    void *d_uv = nullptr;
    DevTex *d_tex = nullptr;               // RT_TEX_NUM descriptors
    unsigned char *d_texels[RT_TEX_NUM] = {}; // device copies
    int tex_w[RT_TEX_NUM] = {}, tex_h[RT_TEX_NUM] = {};
    bool tex_dirty = true;
    int refill_threshold = 24, waves_per_cu = 24, lds_stack = 16, node_format = 1, leaf_batch = 16, occupancy = 6, topology = 1;
    int n_cus = 256;
    unsigned char *static_area = nullptr; // this device's static scratch (kernels.h), looked up at rt_create
    unsigned int *d_bands = nullptr;
    uint32_t *d_spill = nullptr;
    size_t spill_words = 0;

    std::vector<rt_triangle148> tris;
    std::vector<rt_bvh_node32> bnodes;
    WideBvh wide;
    bool scene_on_device = false;
    bool imported = false;       // the device scene came from rt_import_scene: no triangle records, no host layout
    size_t n_wide = 0, n_prims = 0; // sizes of the device scene (c->wide is empty for an imported one)

    void *d_nodes = nullptr, *d_qnodes = nullptr, *d_prims = nullptr, *d_shade = nullptr;
    // the packed scene (RT_OPT_NODE_FORMAT 2, rt_types.h): nodes and primitives in one record array, and the shading /
    // texture-coordinate records indexed by the same record numbers (the entries of node records are unused)
    void *d_pack = nullptr, *d_pshade = nullptr, *d_puv = nullptr;
    size_t n_pack = 0;          // records in d_pack
    std::string pack_why;       // why the scene has no packed form (then format 1 runs)
    DevMat *d_mats = nullptr;
    float *d_fb = nullptr;
    size_t fb_pixels = 0;
    unsigned long long *d_counter = nullptr;

    rt_material mats[RT_MAT_NUM];
    bool mats_dirty = true;

    bool cam_set = false;
    float lookfrom[3], lookat[3], vfov = 40.0f, focus = 10.0f, aperture = 0.0f;

    std::string err;
};

namespace {

// eight band counters (64 bytes with padding), then the kTravStats analysis counters
constexpr size_t kBandBytes = 64 + sizeof(unsigned long long) * kTravStats;

int fail(rt_ctx *c, int code, const std::string &msg) {
    if (c) c->err = msg;
    return code;
}

#define HIPCHK(c, call)                                                                                                \
    do {                                                                                                               \
        hipError_t e_ = (call);                                                                                        \
        if (e_ != hipSuccess) {                                                                                        \
            char buf_[512];                                                                                            \
            snprintf(buf_, sizeof(buf_), "HIP error %d (%s) at %s:%d '%s'", (int)e_, hipGetErrorString(e_), __FILE__, \
                     __LINE__, #call);                                                                                 \
            return fail(c, RT_E_HIP, buf_);                                                                            \
        }                                                                                                              \
    } while (0)

// Nothing may throw across the C boundary (rt_abi.h): the host builders allocate vectors of many GB (bad_alloc) and
// every entry point that reaches them runs between these two.
#define RT_TRY try {
#define RT_CATCH(c)                                                                                                    \
    }                                                                                                                  \
    catch (const std::bad_alloc &) { return fail(c, RT_E_LIMIT, "out of host memory"); }                               \
    catch (const std::exception &e_) { return fail(c, RT_E_ARG, std::string("exception: ") + e_.what()); }             \
    catch (...) { return fail(c, RT_E_ARG, "unknown exception"); }

struct F3 {
    float x, y, z;
};
inline F3 f3(const float *p) { return F3{p[0], p[1], p[2]}; }
inline F3 sub(F3 a, F3 b) { return F3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline F3 scl(float t, F3 v) { return F3{t * v.x, t * v.y, t * v.z}; }
inline F3 cross(F3 a, F3 b) { return F3{a.y * b.z - a.z * b.y, -(a.x * b.z - a.z * b.x), a.x * b.y - a.y * b.x}; }
inline float length(F3 v) { return std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z); }
inline F3 unit(F3 v) {
    float l = length(v);
    return F3{v.x / l, v.y / l, v.z / l};
}
inline void put(float *p, F3 v) {
    p[0] = v.x;
    p[1] = v.y;
    p[2] = v.z;
}

// camera::update, include/RT/rt_camera.h:37-52, with vup = (0,1,0), t0 = 0, t1 = 1 (pbrt_engine.cu:629)
void camera_basis(DevCamera &c, const float lookfrom[3], const float lookat[3], float vfov, float aspect, float aperture,
                  float focus) {
    const F3 vup{0.0f, 1.0f, 0.0f};
    c.time0 = 0.0f;
    c.time1 = 1.0f;
    c.lens_radius = aperture / 2.0f;
    float theta = (vfov / 180.f) * (float)3.1415926; // M_PI of vec3.h:16
    float half_height = tan_sh(theta / 2.0f);
    float half_width = aspect * half_height;
    F3 org = f3(lookfrom);
    F3 w = unit(sub(org, f3(lookat)));
    F3 u = unit(cross(vup, w));
    F3 v = cross(w, u);
    F3 ll = sub(sub(sub(org, scl(half_width * focus, u)), scl(half_height * focus, v)), scl(focus, w));
    put(c.origin, org);
    put(c.lower_left, ll);
    put(c.horizontal, scl(2.0f * half_width * focus, u));
    put(c.vertical, scl(2.0f * half_height * focus, v));
    put(c.u, u);
    put(c.v, v);
    put(c.w, w);
}

void free_scene(rt_ctx *c) {
    if (c->d_nodes) (void)hipFree(c->d_nodes);
    if (c->d_qnodes) (void)hipFree(c->d_qnodes);
    if (c->d_prims) (void)hipFree(c->d_prims);
    if (c->d_shade) (void)hipFree(c->d_shade);
    if (c->d_uv) (void)hipFree(c->d_uv);
    if (c->d_pack) (void)hipFree(c->d_pack);
    if (c->d_pshade) (void)hipFree(c->d_pshade);
    if (c->d_puv) (void)hipFree(c->d_puv);
    c->d_nodes = c->d_qnodes = c->d_prims = c->d_shade = c->d_uv = nullptr;
    c->d_pack = c->d_pshade = c->d_puv = nullptr;
    c->n_pack = 0;
    c->scene_on_device = false;
}

// device layout from the reference tree (topology 0) or from a re-built SAH tree over the same primitives
// (topology 1); either way ties are broken by the DFS-leaf rank in the REFERENCE tree.
bool make_device_layout(const rt_triangle148 *tris, int n, const rt_bvh_node32 *ref_nodes, int m, int leaf_size,
                        int box_mode, int topology, WideBvh &wide, std::string &err, int collapse = 0) {
    if (topology == 0) return build_wide_bvh(tris, n, ref_nodes, m, leaf_size, box_mode, nullptr, wide, err, collapse);
    std::vector<int32_t> order;
    if (!dfs_leaf_order(ref_nodes, m, n, order, err)) return false;
    std::vector<int32_t> ref_rank(n);
    for (int r = 0; r < n; r++) ref_rank[order[r]] = r;
    std::vector<rt_bvh_node32> sah;
    build_sah_bvh(tris, n, sah);
    // the SAH nodes carry un-padded boxes: always re-fit + pad (box mode 0)
    return build_wide_bvh(tris, n, sah.data(), (int)sah.size(), leaf_size, 0, ref_rank.data(), wide, err, collapse);
}

// The packed form of the scene that c->wide holds (format 2).  A tree that does not fit the format is not an error:
// pack_why says why and format 1 runs.
int upload_pack_scene(rt_ctx *c) {
    PackScene ps;
    c->pack_why.clear();
    if (!build_pack_scene(c->wide, ps, c->pack_why)) return RT_OK;
    const size_t n = ps.recs.size();
    HIPCHK(c, hipMalloc(&c->d_pack, n * sizeof(PackRecord)));
    HIPCHK(c, hipMemcpy(c->d_pack, ps.recs.data(), n * sizeof(PackRecord), hipMemcpyHostToDevice));
    std::vector<PackRecord>().swap(ps.recs);
    {
        std::vector<DevShade> sh(n);
        std::memset(sh.data(), 0, n * sizeof(DevShade));
        for (size_t p = 0; p < c->wide.shade.size(); p++) sh[ps.rec_of_prim[p]] = c->wide.shade[p];
        HIPCHK(c, hipMalloc(&c->d_pshade, n * sizeof(DevShade)));
        HIPCHK(c, hipMemcpy(c->d_pshade, sh.data(), n * sizeof(DevShade), hipMemcpyHostToDevice));
    }
    if (!c->wide.uv.empty()) {
        std::vector<DevUV> uv(n);
        std::memset(uv.data(), 0, n * sizeof(DevUV));
        for (size_t p = 0; p < c->wide.uv.size(); p++) uv[ps.rec_of_prim[p]] = c->wide.uv[p];
        HIPCHK(c, hipMalloc(&c->d_puv, n * sizeof(DevUV)));
        HIPCHK(c, hipMemcpy(c->d_puv, uv.data(), n * sizeof(DevUV), hipMemcpyHostToDevice));
    }
    c->n_pack = n;
    return RT_OK;
}

int upload_scene(rt_ctx *c) {
    std::string err;
    if (!make_device_layout(c->tris.data(), (int)c->tris.size(), c->bnodes.data(), (int)c->bnodes.size(), c->leaf_size,
                            c->box_mode, c->topology, c->wide, err, c->collapse))
        return fail(c, RT_E_ARG, "wide BVH: " + err);
    if ((size_t)c->wide.stack_entries * 64 * 4 > 64 * 1024) return fail(c, RT_E_LIMIT, "tree too deep for the LDS stack");
    if (c->wide.qnodes.size() > ((size_t)1 << 26)) // the kernels address quantised nodes by a 32-bit byte offset
        return fail(c, RT_E_LIMIT, "more than 2^26 wide nodes");
    free_scene(c);
    size_t nb = c->wide.nodes.size() * sizeof(WideNode), pb = c->wide.prims.size() * sizeof(DevPrim),
           sb = c->wide.shade.size() * sizeof(DevShade);
    HIPCHK(c, hipMalloc(&c->d_nodes, nb));
    size_t qb = c->wide.qnodes.size() * sizeof(QuantNode);
    HIPCHK(c, hipMalloc(&c->d_qnodes, qb));
    HIPCHK(c, hipMemcpyAsync(c->d_qnodes, c->wide.qnodes.data(), qb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMalloc(&c->d_prims, pb));
    HIPCHK(c, hipMalloc(&c->d_shade, sb));
    HIPCHK(c, hipMemcpyAsync(c->d_nodes, c->wide.nodes.data(), nb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_prims, c->wide.prims.data(), pb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_shade, c->wide.shade.data(), sb, hipMemcpyHostToDevice, c->stream));
    if (!c->wide.uv.empty()) {
        const size_t ub = c->wide.uv.size() * sizeof(DevUV);
        HIPCHK(c, hipMalloc(&c->d_uv, ub));
        HIPCHK(c, hipMemcpyAsync(c->d_uv, c->wide.uv.data(), ub, hipMemcpyHostToDevice, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->node_format == 2) {
        int rc = upload_pack_scene(c);
        if (rc) return rc;
    }
    c->scene_on_device = true;
    c->imported = false;
    c->n_wide = c->wide.qnodes.size();
    c->n_prims = c->wide.prims.size();
    return RT_OK;
}

// ---- the built device scene as one relocatable block of bytes (rt_export_scene / rt_import_scene) ----
// header (256 bytes), then the sections in this order, each starting on a multiple of 256 bytes:
// quantised nodes, primitives, shading records, texture coordinates, [float nodes], [reference binary nodes]
struct SceneBlobHeader {
    char magic[8]; // "RTAMDSC2"
    uint64_t n_wide, n_prims, n_uv, n_float_nodes, n_bnodes, total_bytes;
    int32_t depth, stack_entries;
    float pad;
    int32_t reserved;
    // what the writing build laid out: a block from another build of the library (other record sizes, another node
    // encoding) is refused instead of being traversed
    uint32_t layout_version;
    uint32_t sz_qnode, sz_prim, sz_shade, sz_uv, sz_fnode, sz_bnode;
    uint32_t reserved2;
    uint64_t payload_hash; // blob_hash over bytes [kBlobAlign, total_bytes)
};
static_assert(sizeof(SceneBlobHeader) <= 256, "scene blob header");
constexpr uint32_t kBlobLayoutVersion = 2; // bump whenever QuantNode / DevPrim / DevShade / DevUV or the child-reference encoding change
constexpr uint64_t kBlobAlign = 256;
inline uint64_t blob_round(uint64_t b) { return (b + kBlobAlign - 1) / kBlobAlign * kBlobAlign; }
struct SceneBlobLayout {
    uint64_t qnodes, prims, shade, uv, fnodes, bnodes, total;
};
SceneBlobLayout blob_layout(const SceneBlobHeader &h) {
    SceneBlobLayout L;
    uint64_t at = kBlobAlign;
    L.qnodes = at, at += blob_round(h.n_wide * sizeof(QuantNode));
    L.prims = at, at += blob_round(h.n_prims * sizeof(DevPrim));
    L.shade = at, at += blob_round(h.n_prims * sizeof(DevShade));
    L.uv = at, at += blob_round(h.n_uv * sizeof(DevUV));
    L.fnodes = at, at += blob_round(h.n_float_nodes * sizeof(WideNode));
    L.bnodes = at, at += blob_round(h.n_bnodes * sizeof(rt_bvh_node32));
    L.total = at;
    return L;
}
// Hash of a block's payload (every byte is a multiple of 256 from the block's start, so 8-byte words): a multiply-xor
// chain per 4 MiB chunk, chunks hashed on up to eight threads (the 100 M-triangle block is 8 GB) and folded in order.
// Not cryptographic: it catches truncated-then-padded, stale and torn blocks, which is what a file in /dev/shm can be.
uint64_t blob_hash_words(const unsigned char *p, uint64_t bytes, uint64_t seed) {
    uint64_t h = seed ^ 0x9E3779B97F4A7C15ull;
    for (uint64_t i = 0; i + 8 <= bytes; i += 8) {
        uint64_t w;
        std::memcpy(&w, p + i, 8);
        h = (h ^ w) * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29;
    }
    return h;
}
uint64_t blob_hash(const void *payload, uint64_t bytes) {
    constexpr uint64_t kChunk = (uint64_t)4 << 20;
    const unsigned char *p = static_cast<const unsigned char *>(payload);
    const uint64_t n_chunks = (bytes + kChunk - 1) / kChunk;
    std::vector<uint64_t> part(n_chunks);
    const unsigned n_threads = (unsigned)std::min<uint64_t>(8, std::max<uint64_t>(1, n_chunks / 4));
    auto work = [&](unsigned t) {
        for (uint64_t c = t; c < n_chunks; c += n_threads) part[c] = blob_hash_words(p + c * kChunk, std::min(kChunk, bytes - c * kChunk), c);
    };
    if (n_threads <= 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < n_threads; t++) th.emplace_back(work, t);
        for (auto &x : th) x.join();
    }
    return blob_hash_words(reinterpret_cast<const unsigned char *>(part.data()), n_chunks * 8, bytes);
}
// flags: bit 0 = with the 128-byte float nodes (RT_OPT_NODE_FORMAT 0 and kernel variant 0 need them), bit 1 = with the
// reference's binary node array (rt_get_bvh_nodes on the importing side)
int write_scene_blob(const WideBvh &w, const std::vector<rt_bvh_node32> &bnodes, int flags, void *out, uint64_t cap, uint64_t *bytes) {
    SceneBlobHeader h;
    std::memset(&h, 0, sizeof(h));
    std::memcpy(h.magic, "RTAMDSC2", 8);
    h.n_wide = w.qnodes.size();
    h.n_prims = w.prims.size();
    h.n_uv = w.uv.size();
    h.n_float_nodes = (flags & 1) ? w.nodes.size() : 0;
    h.n_bnodes = (flags & 2) ? bnodes.size() : 0;
    h.depth = w.depth;
    h.stack_entries = w.stack_entries;
    h.pad = w.pad;
    h.layout_version = kBlobLayoutVersion;
    h.sz_qnode = sizeof(QuantNode), h.sz_prim = sizeof(DevPrim), h.sz_shade = sizeof(DevShade), h.sz_uv = sizeof(DevUV);
    h.sz_fnode = sizeof(WideNode), h.sz_bnode = sizeof(rt_bvh_node32);
    const SceneBlobLayout L = blob_layout(h);
    h.total_bytes = L.total;
    if (bytes) *bytes = L.total;
    if (!out) return RT_OK; // size query
    if (cap < L.total) return RT_E_ARG;
    char *o = static_cast<char *>(out);
    // a section is copied and the rest of its 256-byte-aligned slot zeroed: the hash covers every byte of the payload
    auto put_section = [&](uint64_t at, uint64_t next, const void *src, uint64_t n) {
        if (n) std::memcpy(o + at, src, n);
        if (next > at + n) std::memset(o + at + n, 0, next - at - n);
    };
    put_section(L.qnodes, L.prims, w.qnodes.data(), h.n_wide * sizeof(QuantNode));
    put_section(L.prims, L.shade, w.prims.data(), h.n_prims * sizeof(DevPrim));
    put_section(L.shade, L.uv, w.shade.data(), h.n_prims * sizeof(DevShade));
    put_section(L.uv, L.fnodes, w.uv.data(), h.n_uv * sizeof(DevUV));
    put_section(L.fnodes, L.bnodes, w.nodes.data(), h.n_float_nodes * sizeof(WideNode));
    put_section(L.bnodes, L.total, bnodes.data(), h.n_bnodes * sizeof(rt_bvh_node32));
    h.payload_hash = blob_hash(o + kBlobAlign, L.total - kBlobAlign);
    std::memset(o, 0, kBlobAlign);
    std::memcpy(o, &h, sizeof(h));
    return RT_OK;
}
// checks a blob's header against its length and against this build's record layout; err names what is wrong
bool read_scene_blob_header(const void *blob, uint64_t bytes, SceneBlobHeader &h, SceneBlobLayout &L, std::string &err) {
    if (!blob || bytes < kBlobAlign) return err = "scene blob too short", false;
    std::memcpy(&h, blob, sizeof(h));
    if (std::memcmp(h.magic, "RTAMDSC", 7) == 0 && h.magic[7] != '2') return err = "scene blob of another layout version (magic)", false;
    if (std::memcmp(h.magic, "RTAMDSC2", 8) != 0) return err = "not a scene blob (bad magic)", false;
    if (h.layout_version != kBlobLayoutVersion || h.sz_qnode != sizeof(QuantNode) || h.sz_prim != sizeof(DevPrim) ||
        h.sz_shade != sizeof(DevShade) || h.sz_uv != sizeof(DevUV) || h.sz_fnode != sizeof(WideNode) || h.sz_bnode != sizeof(rt_bvh_node32))
        return err = "scene blob was written by another build of the library (layout version / record sizes differ)", false;
    if (h.n_wide == 0 || h.n_prims == 0 || h.n_wide > ((uint64_t)1 << 26) || h.n_prims > 0x0FFFFFFFull ||
        (h.n_uv != 0 && h.n_uv != h.n_prims) || (h.n_float_nodes != 0 && h.n_float_nodes != h.n_wide) ||
        h.n_bnodes > 0x7FFFFFFFull || h.stack_entries < 1 || (uint64_t)h.stack_entries * 256 > 64 * 1024)
        return err = "scene blob header out of range", false;
    L = blob_layout(h);
    if (L.total != h.total_bytes || bytes < L.total) return err = "scene blob truncated", false;
    return true;
}
// The sections themselves, before any of it reaches the GPU: the payload hash, then one pass over the quantised nodes
// and the per-primitive records.  Every child reference must name a later node (the layout is breadth-first: that also
// rules out cycles) or a primitive range inside the array; the depth the references imply must fit the stack the
// header announces (3 entries per level + 2: wide_bvh.cpp); material and texture indices must be in range.  A corrupt
// block is an RT_E_ARG here instead of an out-of-bounds read or a stack overflow inside the traversal kernel.
bool validate_scene_blob(const void *blob, const SceneBlobHeader &h, const SceneBlobLayout &L, std::string &err) {
    const char *b = static_cast<const char *>(blob);
    if (blob_hash(b + kBlobAlign, L.total - kBlobAlign) != h.payload_hash) return err = "scene blob payload does not match its checksum (stale, torn or padded block)", false;
    const QuantNode *q = reinterpret_cast<const QuantNode *>(b + L.qnodes);
    std::vector<int32_t> level(h.n_wide, 0);
    std::vector<uint8_t> covered; // primitives under some leaf
    level[0] = 1;
    int32_t max_level = 1;
    uint64_t prims_in_leaves = 0;
    for (uint64_t i = 0; i < h.n_wide; i++) {
        if (level[i] == 0) return err = "scene blob: wide node " + std::to_string(i) + " is not referenced by an earlier node", false;
        max_level = std::max(max_level, level[i]);
        for (int s = 0; s < 4; s++) {
            const uint32_t r = q[i].child[s];
            if (r == kEmptyRef) continue;
            if (r & kLeafBit) {
                const uint64_t start = r & kLeafStartMask, cnt = ((r >> 28) & 7u) + 1u;
                if (start + cnt > h.n_prims) return err = "scene blob: leaf of wide node " + std::to_string(i) + " reaches beyond the primitive array", false;
                prims_in_leaves += cnt;
            } else {
                if ((uint64_t)r >= h.n_wide || (uint64_t)r <= i) return err = "scene blob: child reference of wide node " + std::to_string(i) + " out of range", false;
                if (level[r] != 0) return err = "scene blob: wide node " + std::to_string(r) + " is referenced twice", false;
                level[r] = level[i] + 1;
            }
        }
    }
    if (prims_in_leaves != h.n_prims) return err = "scene blob: leaves do not cover the primitive array exactly once", false;
    if (3 * (int64_t)max_level + 2 > (int64_t)h.stack_entries || max_level > h.depth)
        return err = "scene blob: tree is deeper than the header says (traversal stack would overflow)", false;
    const DevShade *sh = reinterpret_cast<const DevShade *>(b + L.shade);
    for (uint64_t i = 0; i < h.n_prims; i++)
        if (sh[i].mat < 0 || sh[i].mat >= RT_MAT_NUM) return err = "scene blob: material index out of range", false;
    const DevPrim *pr = reinterpret_cast<const DevPrim *>(b + L.prims);
    for (uint64_t i = 0; i < h.n_prims; i++)
        if (pr[i].shape != 0 && pr[i].shape != 1) return err = "scene blob: primitive shape must be 0 or 1", false;
    const DevUV *uv = reinterpret_cast<const DevUV *>(b + L.uv);
    for (uint64_t i = 0; i < h.n_uv; i++)
        if (uv[i].tex < -1 || uv[i].tex >= RT_TEX_NUM) return err = "scene blob names a texture index out of range", false;
    if (h.n_float_nodes) { // the float copy carries the same references
        const WideNode *w = reinterpret_cast<const WideNode *>(b + L.fnodes);
        for (uint64_t i = 0; i < h.n_wide; i++)
            if (std::memcmp(w[i].child, q[i].child, sizeof(q[i].child)) != 0) return err = "scene blob: float and quantised nodes disagree", false;
    }
    if (h.n_bnodes && h.n_bnodes != 2 * h.n_prims - 1) return err = "scene blob: binary node count must be 2n-1", false;
    return true;
}

// the descriptor table follows every rt_set_texture at once (d_tex is allocated at rt_create), so that a render has
// nothing left to upload
int upload_textures(rt_ctx *c) {
    if (!c->tex_dirty) return RT_OK;
    DevTex tmp[RT_TEX_NUM];
    for (int i = 0; i < RT_TEX_NUM; i++) {
        tmp[i].texels = c->d_texels[i];
        tmp[i].w = c->tex_w[i];
        tmp[i].h = c->tex_h[i];
    }
    HIPCHK(c, hipMemcpyAsync(c->d_tex, tmp, sizeof(tmp), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->tex_dirty = false;
    return RT_OK;
}
// every texture a triangle names must have been set (host-side check, touches no device memory)
int check_textures(rt_ctx *c) {
    for (const DevUV &q : c->wide.uv)
        if (q.tex != -1 && !c->d_texels[q.tex])
            return fail(c, RT_E_ARG, "a triangle names texture " + std::to_string(q.tex) + ", which was never set (rt_set_texture)");
    return RT_OK;
}

int upload_mats(rt_ctx *c) {
    if (!c->mats_dirty) return RT_OK;
    DevMat tmp[RT_MAT_NUM];
    for (int i = 0; i < RT_MAT_NUM; i++) {
        tmp[i].kind = c->mats[i].kind;
        for (int k = 0; k < 3; k++) tmp[i].albedo[k] = c->mats[i].albedo[k];
        tmp[i].param = c->mats[i].param;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_mats, tmp, sizeof(tmp), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->mats_dirty = false;
    return RT_OK;
}

int ensure_fb(rt_ctx *c) {
    size_t px = (size_t)c->nx * c->ny;
    if (c->d_fb && c->fb_pixels == px) return RT_OK;
    if (c->d_fb) (void)hipFree(c->d_fb);
    c->d_fb = nullptr;
    c->fb_pixels = 0;
    HIPCHK(c, hipMalloc((void **)&c->d_fb, px * 3 * sizeof(float)));
    c->fb_pixels = px;
    return RT_OK;
}

int ready_to_trace(rt_ctx *c) {
    if (c->imported) {
        if (!c->scene_on_device)
            return fail(c, RT_E_ARG, "imported scene: a device-layout option changed and there are no triangle records to rebuild from; "
                                     "import again or set the scene");
        if (!c->d_nodes && (c->node_format == 0 || c->kernel_variant == 0))
            return fail(c, RT_E_ARG, "imported scene carries no float nodes (export flag 1): RT_OPT_NODE_FORMAT 0 / RT_OPT_KERNEL 0 cannot run");
    } else {
        if (c->tris.empty()) return fail(c, RT_E_ARG, "no scene: call rt_set_scene first");
        if (c->bnodes.empty()) return fail(c, RT_E_ARG, "no BVH: call rt_build_bvh or rt_set_bvh_nodes first");
    }
    if (!c->scene_on_device) {
        int rc = upload_scene(c);
        if (rc) return rc;
    }
    int rc = check_textures(c);
    if (rc) return rc;
    rc = upload_textures(c);
    if (rc) return rc;
    return upload_mats(c);
}

void fill_frame(rt_ctx *c, KernelFrame &F) {
    float aspect = float(c->nx) / float(c->ny);
    camera_basis(F.cam, c->lookfrom, c->lookat, c->honor_vfov ? c->vfov : 40.0f, aspect, c->aperture, c->focus);
    F.nx = c->nx;
    F.ny = c->ny;
    F.ns = c->ns;
    F.max_depth = c->max_depth;
    F.tiles_x = (c->nx + 7) / 8;
    F.tiles_y = (c->ny + 7) / 8;
    F.div_tiles_x = fastdiv_make((uint32_t)F.tiles_x);
    long total = (long)F.tiles_x * F.tiles_y;
    F.shard_rank = c->shard_rank;
    F.shard_world = c->shard_world;
    F.n_local_tiles = (int)((total - c->shard_rank + c->shard_world - 1) / c->shard_world);
    F.ray_counter = c->d_counter;
    F.seed_salt = c->seed_salt;
    F.raw_sum = c->raw_sum;
}

// The node format a launch runs with: the packed scene needs its device arrays (a tree that does not fit the format has
// none: pack_why) and exists for the speculative persistent kernels only; otherwise format 2 means format 1.
int effective_format(const rt_ctx *c) {
    if (c->node_format != 2) return c->node_format;
    return c->d_pack && c->kernel_variant == 2 ? 2 : 1;
}

KernelScene scene_of(rt_ctx *c) {
    KernelScene S;
    S.nodes = (const float4 *)c->d_nodes;
    S.mats = c->d_mats;
    S.tex = c->d_tex;
    if (effective_format(c) == 2) { // one record array is both the node and the primitive array
        S.qnodes = (const uint4 *)c->d_pack;
        S.prims = (const float4 *)c->d_pack;
        S.shade = (const float4 *)c->d_pshade;
        S.uv = (const float4 *)c->d_puv;
        return S;
    }
    S.qnodes = (const uint4 *)c->d_qnodes;
    S.prims = (const float4 *)c->d_prims;
    S.shade = (const float4 *)c->d_shade;
    S.uv = (const float4 *)c->d_uv;
    return S;
}

void fill_stats(rt_ctx *c, rt_stats *st, uint64_t rays, float ms, double total_ms) {
    if (!st) return;
    st->rays = rays;
    st->kernel_ms = ms;
    st->total_ms = total_ms;
    st->wide_nodes = c->n_wide;
    st->prims = c->n_prims;
    st->stack_entries = c->wide.stack_entries;
    st->handoff_torn = 0;
}

// The tag of a round record is frame_id * 64 + round, and a record array is never cleared between frames: that needs
// frame ids that no array sees twice.  They are drawn per DEVICE, not per context, so that arrays shared by the
// contexts of a device can rely on them too: ids run 1 .. 2^26 - 1 and then start over in a new "generation"; an
// array that was last zeroed in another generation is zeroed again before it is used.
constexpr int kMaxDevices = 64;
constexpr uint64_t kFrameIds = (1ull << 26) - 1;
std::mutex g_frame_mutex;
uint64_t g_frames_issued[kMaxDevices] = {};
void next_frame_tag(int device, uint32_t &id, uint64_t &generation) {
    std::lock_guard<std::mutex> lock(g_frame_mutex);
    const uint64_t n = g_frames_issued[device]++;
    id = (uint32_t)(n % kFrameIds) + 1u;
    generation = n / kFrameIds + 1u; // never 0: a fresh array (generation 0) is always zeroed first
}

// kernel variant 1: zero the band counters, size the spill area, fill the launch record
uint64_t g_static_ring_generation[kMaxDevices] = {}; // as rt_ctx::ring_generation, for a device's static record area

enum LaunchKind { kLaunchRender, kLaunchPrimary, kLaunchRays };
// no_alloc: every scratch buffer comes out of the device's static area (kernels.h) instead of the context's own
// allocations; what does not fit is refused (RT_E_LIMIT), nothing is allocated or freed.
int prepare_persistent(rt_ctx *c, PersistLaunch &L, LaunchKind kind, size_t n_rays = 0, bool no_alloc = false) {
    const bool render = kind == kLaunchRender;
    // Render: waves_per_cu.  Textured scenes run the 128-VGPR instances (4 waves/SIMD): no point in launching more
    // waves than are resident.  Primary rays are short and coherent, and every wave launched costs a ramp of atomics
    // and first misses: with 256 slots per atomic, 8 waves per CU measured best at 1080p (1 M triangles: 0.52 ms vs
    // 0.57 at 12 and 0.62 at 16) and 16 at 2160p (1.03 ms vs 1.25 at 8).  Ray arrays can be anything: the 288 M rays of a bench frame, replayed, take 82 / 49 / 39 ms
    // at 8 / 16 / 24 waves per CU; so a ray array gets enough waves for ~16 rays per lane, between 4 per CU and
    // RT_OPT_QUEUE_WAVES.
    int per_cu = c->waves_per_cu;
    if (kind == kLaunchPrimary) { // ~16 pixels per lane, between 8 waves per CU and RT_OPT_PRIMARY_WAVES
        const size_t want = ((size_t)c->nx * c->ny / (size_t)std::max(c->shard_world, 1) + (size_t)c->n_cus * 1024 - 1) / ((size_t)c->n_cus * 1024);
        per_cu = std::min(per_cu, (int)std::min<size_t>((size_t)c->primary_waves, std::max<size_t>(want, 8)));
    } else if (kind == kLaunchRays) {
        const size_t want = (n_rays + (size_t)c->n_cus * 1024 - 1) / ((size_t)c->n_cus * 1024);
        per_cu = std::min(per_cu, (int)std::min<size_t>((size_t)c->queue_waves, std::max<size_t>(want, 4)));
    }
    else if (!c->wide.uv.empty())
        per_cu = std::min(per_cu, 16);
    L.grid_waves = c->n_cus * per_cu;
    L.lds_entries = c->lds_stack < c->wide.stack_entries ? c->lds_stack : c->wide.stack_entries;
    L.threshold = c->refill_threshold;
    L.node_format = effective_format(c);
    L.speculative = c->kernel_variant >= 2 ? 1 : 0;
    L.deterministic = c->deterministic;
    L.leaf_batch = c->leaf_batch;
    L.primary_chunk = c->primary_chunk;
    L.occupancy = c->occupancy;
    int spill_entries = c->wide.stack_entries - L.lds_entries;
    size_t words = (size_t)(spill_entries > 0 ? spill_entries : 1) * (size_t)L.grid_waves * 64;
    if (no_alloc) {
        if (words * sizeof(uint32_t) > kStaticSpillBytes)
            return fail(c, RT_E_LIMIT, "the scene's tree is too deep for the static scratch: its traversal stacks spill " +
                                           std::to_string(words * sizeof(uint32_t)) + " bytes, the limit is " + std::to_string(kStaticSpillBytes));
    } else if (words > c->spill_words) {
        if (c->d_spill) (void)hipFree(c->d_spill);
        c->d_spill = nullptr;
        c->spill_words = 0;
        HIPCHK(c, hipMalloc((void **)&c->d_spill, words * sizeof(uint32_t)));
        c->spill_words = words;
    }
    L.spill = no_alloc ? reinterpret_cast<uint32_t *>(c->static_area + kStaticSpillOff) : c->d_spill;
    L.band_counters = no_alloc ? reinterpret_cast<unsigned int *>(c->static_area + kStaticBandsOff) : c->d_bands;
    L.rounds = 1;
    L.round_state = nullptr;
    L.frame_id = 0;
    const long tiles = (long)((c->nx + 7) / 8) * ((c->ny + 7) / 8);
    const size_t slots = (size_t)((tiles - c->shard_rank + c->shard_world - 1) / c->shard_world) * 64;
    // Rounds cost a hand-off each (2160p, 64 spp: 146.6 ms with 2 or 4 rounds, 150 with 8, 155.5 with 16) and buy a
    // short end of the frame, which matters when a lane sees few pixels (1080p: 5 per lane; 54 / 47.5 / 45.7 / 44.9 ms
    // with 1 / 2 / 4 / 8 rounds).  RT_OPT_TAIL_SPLIT 0: about 42 pieces per lane, between 2 and 8 rounds.
    int want_rounds = c->tail_split;
    if (want_rounds == 0) {
        const double per_lane = (double)slots / ((double)L.grid_waves * 64.0);
        want_rounds = per_lane > 0.0 ? (int)std::lround(42.0 / per_lane) : 8;
        want_rounds = std::max(2, std::min(want_rounds, 8));
    }
    const int rounds = std::min(std::min(want_rounds, c->ns), 63);
    // slot | round << 26 in one register; the record array is addressed through a buffer descriptor (32-bit byte offsets)
    if (render && rounds > 1 && slots > 0 && slots < ((size_t)1 << 26) &&
        slots * kRingWords * sizeof(uint32_t) <= 0xFFFFFFFFull) {
        uint32_t *records;
        size_t record_bytes;
        uint64_t *zeroed_in;
        if (no_alloc) {
            if (slots * kRingWords * sizeof(uint32_t) > kStaticRingBytes)
                return fail(c, RT_E_LIMIT, "frame too large for the static scratch: " + std::to_string(slots) + " pixel slots, the limit is " +
                                               std::to_string(kStaticRingBytes / (kRingWords * sizeof(uint32_t))));
            records = reinterpret_cast<uint32_t *>(c->static_area + kStaticRingOff);
            record_bytes = kStaticRingBytes;
            zeroed_in = &g_static_ring_generation[c->device];
        } else {
            if (slots > c->ring_items) {
                if (c->d_ring_state) (void)hipFree(c->d_ring_state);
                c->d_ring_state = nullptr;
                c->ring_items = 0;
                HIPCHK(c, hipMalloc((void **)&c->d_ring_state, slots * kRingWords * sizeof(uint32_t)));
                c->ring_items = slots;
                c->ring_generation = 0;
            }
            records = c->d_ring_state;
            record_bytes = c->ring_items * kRingWords * sizeof(uint32_t);
            zeroed_in = &c->ring_generation;
        }
        uint64_t generation;
        next_frame_tag(c->device, L.frame_id, generation);
        if (generation != *zeroed_in) { // fresh array, or the ids have started over since it was last zeroed
            HIPCHK(c, hipMemsetAsync(records, 0, record_bytes, c->stream));
            *zeroed_in = generation;
        }
        L.round_state = records;
        L.rounds = rounds;
        // Shares of the rounds: boundary r at ns * (1 - (1 - r/R)^p), p = RT_OPT_TAIL_TAPER / 100.  p = 1: equal
        // shares; p > 1: the last rounds are short, and what is still running when the work has been handed out is
        // a short piece per lane.  Kept strictly increasing (a round runs at least one sample).
        {
            const double p = c->tail_taper / 100.0;
            int prev = 0;
            for (int r = 0; r < rounds; r++) {
                int k = r + 1 == rounds ? c->ns : (int)std::floor(c->ns * (1.0 - std::pow(1.0 - double(r + 1) / rounds, p)));
                k = std::max(k, prev + 1);
                k = std::min(k, c->ns - (rounds - 1 - r)); // leave one sample for every later round
                L.round_end[r] = prev = k;
            }
        }
    }
    L.stats = reinterpret_cast<unsigned long long *>(L.band_counters + 16);
    if (!no_alloc && kind != kLaunchRays && c->d_wave_times && (size_t)L.grid_waves <= c->wave_times_cap) L.wave_times = c->d_wave_times;
    if (!no_alloc && render && c->d_ray_dump) {
        L.ray_dump = c->d_ray_dump;
        L.ray_dump_count = c->d_ray_dump_count;
        L.ray_dump_cap = c->ray_dump_cap;
        HIPCHK(c, hipMemsetAsync(c->d_ray_dump_count, 0, sizeof(unsigned long long), c->stream));
    }
    HIPCHK(c, hipMemsetAsync(L.band_counters, 0, kBandBytes, c->stream));
    return RT_OK;
}

double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

} // namespace

extern "C" {

int rt_abi_version(void) { return RT_ABI_VERSION; }

int rt_set_texture(rt_ctx *c, int index, const uint8_t *rgba8, int w, int h) {
    if (!c) return RT_E_ARG;
    if (index < 0 || index >= RT_TEX_NUM) return fail(c, RT_E_ARG, "texture index out of range");
    if (!rgba8 || w <= 0 || h <= 0 || w > 32768 || h > 32768) return fail(c, RT_E_ARG, "texture must be 1..32768 texels per side");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->d_texels[index]) (void)hipFree(c->d_texels[index]);
    c->d_texels[index] = nullptr;
    const size_t bytes = (size_t)4 * w * h;
    HIPCHK(c, hipMalloc((void **)&c->d_texels[index], bytes));
    HIPCHK(c, hipMemcpy(c->d_texels[index], rgba8, bytes, hipMemcpyHostToDevice));
    c->tex_w[index] = w;
    c->tex_h[index] = h;
    c->tex_dirty = true;
    return upload_textures(c);
}

int rt_default_materials(rt_material out[RT_MAT_NUM]) {
    if (!out) return RT_E_ARG;
    // material_init, pbrt_engine.cu:25-47; the random entries come from ONE stream seeded 1995 (:599),
    // drawn left to right: albedo x, y, z (then fuzz).
    auto set = [&](int i, int kind, float r, float g, float b, float p) {
        out[i].kind = kind;
        out[i].albedo[0] = r;
        out[i].albedo[1] = g;
        out[i].albedo[2] = b;
        out[i].param = p;
    };
    auto fuzz = [](float f) { return (double)f < 1.0 ? f : 1.0f; }; // metal ctor, material.h:160-162
    set(0, RT_MAT_LAMBERTIAN, 0.5f, 0.5f, 0.5f, 0.0f);
    set(1, RT_MAT_LAMBERTIAN, 0.2f, 0.4f, 0.1f, 0.0f);
    set(2, RT_MAT_DIELECTRIC, 1.0f, 1.0f, 1.0f, 0.4f);
    set(3, RT_MAT_LIGHT, 20.0f, 20.0f, 20.0f, 0.0f);
    set(4, RT_MAT_METAL, 1.0f, (float)(182.0 / 255.0), (float)(193.0 / 255.0), fuzz((float)0.01));
    set(5, RT_MAT_METAL, 0.8f, 0.85f, 0.88f, fuzz(0.0f));
    set(6, RT_MAT_METAL, 0.7f, 0.6f, 0.5f, fuzz(0.0f));
    set(7, RT_MAT_DIELECTRIC, 1.0f, 1.0f, 1.0f, 1.5f);
    set(8, RT_MAT_LAMBERTIAN, 0.65f, 0.05f, 0.05f, 0.0f);
    set(9, RT_MAT_LAMBERTIAN, 0.12f, 0.45f, 0.15f, 0.0f);
    Rng g;
    rng_init(g, 1995);
    for (int i = 10; i < 30; i++) {
        float r = rng_uniform(g), gg = rng_uniform(g), b = rng_uniform(g);
        set(i, RT_MAT_LAMBERTIAN, r, gg, b, 0.0f);
    }
    for (int i = 30; i < 50; i++) {
        float r = rng_uniform(g), gg = rng_uniform(g), b = rng_uniform(g);
        float f = rng_uniform(g) / 50;
        set(i, RT_MAT_METAL, r, gg, b, fuzz(f));
    }
    return RT_OK;
}

int rt_create(rt_ctx **out, int device_id) {
    if (!out) return RT_E_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return RT_E_HIP; // no CPU fallback
    if (device_id < 0 || device_id >= count || device_id >= kMaxDevices) return RT_E_ARG;
    rt_ctx *c = new (std::nothrow) rt_ctx();
    if (!c) return RT_E_LIMIT;
    c->device = device_id;
    auto bail = [&](hipError_t e) { // every member is null-initialised and rt_destroy null-checks each one
        (void)e;
        (void)rt_destroy(c);
        return RT_E_HIP;
    };
    hipError_t e;
    if ((e = hipSetDevice(device_id)) != hipSuccess) return bail(e);
    if ((e = hipStreamCreate(&c->stream)) != hipSuccess) return bail(e);
    if ((e = hipEventCreate(&c->ev0)) != hipSuccess) return bail(e);
    if ((e = hipEventCreate(&c->ev1)) != hipSuccess) return bail(e);
    if ((e = hipMalloc((void **)&c->d_counter, sizeof(unsigned long long))) != hipSuccess) return bail(e);
    if ((e = hipMalloc((void **)&c->d_mats, sizeof(DevMat) * RT_MAT_NUM)) != hipSuccess) return bail(e);
    if ((e = hipMalloc((void **)&c->d_bands, kBandBytes)) != hipSuccess) return bail(e);
    {
        hipDeviceProp_t prop;
        if ((e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess) return bail(e);
        c->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    if ((e = hipMalloc((void **)&c->d_tex, sizeof(DevTex) * RT_TEX_NUM)) != hipSuccess) return bail(e);
    {
        // the library's set-up call: this device's copy of the static scratch array is looked up (and zeroed) here, once
        // per device, with the device current; entry points only ever see the pointer kept from this call
        void *p = nullptr;
        if ((e = static_scratch(device_id, &p)) != hipSuccess) return bail(e);
        c->static_area = static_cast<unsigned char *>(p);
    }
    rt_default_materials(c->mats);
    if (upload_mats(c) != RT_OK || upload_textures(c) != RT_OK) return bail(hipErrorUnknown);
    *out = c;
    return RT_OK;
}

int rt_destroy(rt_ctx *c) {
    if (!c) return RT_E_ARG;
    (void)hipSetDevice(c->device);
    free_scene(c);
    if (c->d_fb) (void)hipFree(c->d_fb);
    if (c->d_mats) (void)hipFree(c->d_mats);
    if (c->d_counter) (void)hipFree(c->d_counter);
    if (c->d_bands) (void)hipFree(c->d_bands);
    if (c->d_ray_dump) (void)hipFree(c->d_ray_dump);
    if (c->d_ray_dump_count) (void)hipFree(c->d_ray_dump_count);
    if (c->d_wave_times) (void)hipFree(c->d_wave_times);
    if (c->d_tex) (void)hipFree(c->d_tex);
    for (int i = 0; i < RT_TEX_NUM; i++)
        if (c->d_texels[i]) (void)hipFree(c->d_texels[i]);
    if (c->d_ring_state) (void)hipFree(c->d_ring_state);
    if (c->d_spill) (void)hipFree(c->d_spill);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return RT_OK;
}

const char *rt_last_error(rt_ctx *c) { return c ? c->err.c_str() : "null context"; }

int rt_resize(rt_ctx *c, int nx, int ny, int ns) {
    if (!c) return RT_E_ARG;
    if (nx <= 0 || ny <= 0 || ns <= 0) return fail(c, RT_E_ARG, "nx, ny, ns must be positive");
    if ((long)nx * ny > (1L << 30)) return fail(c, RT_E_LIMIT, "frame too large (pixel index must fit int32)");
    if (nx > 65535 || ny > 65535) return fail(c, RT_E_LIMIT, "frame too large (at most 65535 pixels per side)");
    HIPCHK(c, hipSetDevice(c->device));
    c->nx = nx;
    c->ny = ny;
    c->ns = ns;
    return ensure_fb(c);
}

int rt_set_scene(rt_ctx *c, const rt_triangle148 *tris, int n) {
    RT_TRY
    if (!c) return RT_E_ARG;
    if (!tris || n <= 0) return fail(c, RT_E_ARG, "empty scene");
    if (tris[0].shapeType != 0)
        return fail(c, RT_E_UNSUPPORTED,
                    "tri[0] must be a triangle: it is the hard-coded light of pbrt_engine.cu:464,502 and a sphere there "
                    "can make the zero-weighted pdf term NaN");
    for (int i = 0; i < n; i++) {
        if (tris[i].shapeType != 0 && tris[i].shapeType != 1)
            return fail(c, RT_E_UNSUPPORTED, "shapeType must be 0 (triangle) or 1 (sphere)");
        if (tris[i].shapeType == 0 && (tris[i].texIdx < -1 || tris[i].texIdx >= RT_TEX_NUM))
            return fail(c, RT_E_ARG, "texIdx must be -1 (no texture) or an index below RT_TEX_NUM");
        if (tris[i].matType < 0 || tris[i].matType >= RT_MAT_NUM) return fail(c, RT_E_ARG, "matType out of range");
        // non-finite coordinates would feed NaN keys to the builders' sort comparators (no strict weak order)
        bool finite = true;
        for (int k = 0; k < 3; k++) finite = finite && std::isfinite(tris[i].v0[k]);
        if (tris[i].shapeType == 0)
            for (int k = 0; k < 3; k++) finite = finite && std::isfinite(tris[i].v1[k]) && std::isfinite(tris[i].v2[k]);
        else
            finite = finite && std::isfinite(tris[i].v1[0]);
        if (!finite) return fail(c, RT_E_ARG, "non-finite vertex or radius in primitive " + std::to_string(i));
    }
    HIPCHK(c, hipSetDevice(c->device));
    c->tris.assign(tris, tris + n);
    c->bnodes.clear();
    free_scene(c);
    c->imported = false;
    return RT_OK;
    RT_CATCH(c)
}

int rt_build_bvh(rt_ctx *c, int64_t rand_calls_before) {
    RT_TRY
    if (!c) return RT_E_ARG;
    if (c->tris.empty()) return fail(c, RT_E_ARG, "no scene");
    if (rand_calls_before < 0) return fail(c, RT_E_ARG, "rand_calls_before < 0");
    HIPCHK(c, hipSetDevice(c->device));
    build_binary_bvh(c->tris.data(), (int)c->tris.size(), rand_calls_before, c->bnodes);
    return upload_scene(c);
    RT_CATCH(c)
}

int rt_set_bvh_nodes(rt_ctx *c, const rt_bvh_node32 *nodes, int m) {
    RT_TRY
    if (!c) return RT_E_ARG;
    if (c->tris.empty()) return fail(c, RT_E_ARG, "no scene");
    if (!nodes || m != 2 * (int)c->tris.size() - 1) return fail(c, RT_E_ARG, "node count must be 2*n-1");
    HIPCHK(c, hipSetDevice(c->device));
    c->bnodes.assign(nodes, nodes + m);
    int rc = upload_scene(c);
    if (rc) c->bnodes.clear();
    return rc;
    RT_CATCH(c)
}

int rt_get_bvh_nodes(rt_ctx *c, rt_bvh_node32 *out, int cap, int *m) {
    RT_TRY
    if (!c || !m) return RT_E_ARG;
    *m = (int)c->bnodes.size();
    if (out) {
        if (cap < *m) return fail(c, RT_E_ARG, "output too small");
        std::memcpy(out, c->bnodes.data(), c->bnodes.size() * sizeof(rt_bvh_node32));
    }
    return RT_OK;
    RT_CATCH(c)
}

int rt_set_materials(rt_ctx *c, const rt_material *mats, int n) {
    if (!c) return RT_E_ARG;
    if (!mats || n <= 0 || n > RT_MAT_NUM) return fail(c, RT_E_ARG, "1..50 materials");
    for (int i = 0; i < n; i++)
        if (mats[i].kind < RT_MAT_LAMBERTIAN || mats[i].kind > RT_MAT_LIGHT) return fail(c, RT_E_ARG, "bad material kind");
    std::memcpy(c->mats, mats, sizeof(rt_material) * (size_t)n);
    c->mats_dirty = true;
    HIPCHK(c, hipSetDevice(c->device));
    return upload_mats(c); // at once: a render finds the table resident
}

int rt_set_camera(rt_ctx *c, const float lookfrom[3], const float lookat[3], float vfov, float dist_to_focus,
                  float aperture) {
    if (!c || !lookfrom || !lookat) return RT_E_ARG;
    // A non-finite camera makes every ray NaN; the reference then reports NaN-t "hits" that depend on which boxes it
    // happened to visit, which no other hierarchy reproduces: refuse instead of diverging silently.
    bool finite = std::isfinite(vfov) && std::isfinite(dist_to_focus) && std::isfinite(aperture);
    for (int k = 0; k < 3; k++) finite = finite && std::isfinite(lookfrom[k]) && std::isfinite(lookat[k]);
    if (!finite) return fail(c, RT_E_ARG, "camera parameters must be finite");
    for (int k = 0; k < 3; k++) {
        c->lookfrom[k] = lookfrom[k];
        c->lookat[k] = lookat[k];
    }
    c->vfov = vfov;
    c->focus = dist_to_focus;
    c->aperture = aperture;
    c->cam_set = true;
    return RT_OK;
}

int rt_set_option(rt_ctx *c, int option, int64_t value) {
    if (!c) return RT_E_ARG;
    switch (option) {
    case RT_OPT_MAX_DEPTH:
        if (value < 1 || value > 1000000) return fail(c, RT_E_ARG, "max depth out of range");
        c->max_depth = (int)value;
        return RT_OK;
    case RT_OPT_HONOR_VFOV:
        c->honor_vfov = value ? 1 : 0;
        return RT_OK;
    case RT_OPT_LEAF_SIZE:
        if (value < 1 || value > kMaxLeafPrims) return fail(c, RT_E_ARG, "leaf size 1..8");
        if (c->leaf_size != (int)value) c->scene_on_device = false;
        c->leaf_size = (int)value;
        return RT_OK;
    case RT_OPT_BOX_MODE:
        if (value != 0 && value != 1) return fail(c, RT_E_ARG, "box mode 0 or 1");
        if (c->box_mode != (int)value) c->scene_on_device = false;
        c->box_mode = (int)value;
        return RT_OK;
    case RT_OPT_SHARD_RANK:
        if (value < 0 || value >= c->shard_world) return fail(c, RT_E_ARG, "rank must be in [0, world)");
        c->shard_rank = (int)value;
        return RT_OK;
    case RT_OPT_SHARD_WORLD:
        if (value < 1 || value > 4096) return fail(c, RT_E_ARG, "world out of range");
        c->shard_world = (int)value;
        if (c->shard_rank >= c->shard_world) c->shard_rank = 0;
        return RT_OK;
    case RT_OPT_KERNEL:
        if (value == 3) return fail(c, RT_E_ARG, "kernel variant 3 (two paths per lane) was a measured negative and left the library (branch exp/k-dual): 0..2");
        if (value < 0 || value > 2) return fail(c, RT_E_ARG, "kernel variant 0..2");
        c->kernel_variant = (int)value;
        return RT_OK;
    case RT_OPT_REFILL_THRESHOLD:
        if (value < 1 || value > 64) return fail(c, RT_E_ARG, "refill threshold 1..64");
        c->refill_threshold = (int)value;
        return RT_OK;
    case RT_OPT_WAVES_PER_CU:
        if (value < 1 || value > 32) return fail(c, RT_E_ARG, "waves per CU 1..32");
        c->waves_per_cu = (int)value;
        return RT_OK;
    case RT_OPT_LDS_STACK:
        if (value < 1 || value > 248) return fail(c, RT_E_ARG, "LDS stack entries 1..248");
        c->lds_stack = (int)value;
        return RT_OK;
    case RT_OPT_NODE_FORMAT:
        if (value < 0 || value > 2) return fail(c, RT_E_ARG, "node format 0, 1 or 2");
        // the packed scene is laid out and uploaded on demand; an imported scene has no host layout to pack and keeps
        // running format 1 (effective_format)
        if (value == 2 && !c->d_pack && !c->imported) c->scene_on_device = false;
        c->node_format = (int)value;
        return RT_OK;
    case RT_OPT_LEAF_BATCH:
        if (value < 1 || value > 64) return fail(c, RT_E_ARG, "leaf batch 1..64");
        c->leaf_batch = (int)value;
        return RT_OK;
    case RT_OPT_OCCUPANCY:
        if (value != 4 && value != 6 && value != 8) return fail(c, RT_E_ARG, "occupancy 4, 6 or 8");
        c->occupancy = (int)value;
        return RT_OK;
    case RT_OPT_SHADE_BATCH:
        if (value < 1 || value > 64) return fail(c, RT_E_ARG, "shade batch 1..64");
        c->shade_batch = (int)value;
        return RT_OK;
    case RT_OPT_SWAP_BATCH:
        if (value < 1 || value > 64) return fail(c, RT_E_ARG, "swap batch 1..64");
        c->swap_batch = (int)value;
        return RT_OK;
    case RT_OPT_QUEUE_THRESHOLD:
        if (value < 1 || value > 64) return fail(c, RT_E_ARG, "queue threshold 1..64");
        c->queue_threshold = (int)value;
        return RT_OK;
    case RT_OPT_PRIMARY_WAVES:
        if (value < 1 || value > 32) return fail(c, RT_E_ARG, "primary waves per CU 1..32");
        c->primary_waves = (int)value;
        return RT_OK;
    case RT_OPT_PRIMARY_CHUNK:
        if (value < 64 || value > 65536 || value % 64) return fail(c, RT_E_ARG, "primary chunk: a multiple of 64, 64..65536");
        c->primary_chunk = (int)value;
        return RT_OK;
    case RT_OPT_QUEUE_WAVES:
        if (value < 1 || value > 32) return fail(c, RT_E_ARG, "queue waves per CU 1..32");
        c->queue_waves = (int)value;
        return RT_OK;
    case RT_OPT_SEED_SALT:
        if (value < 0) return fail(c, RT_E_ARG, "seed salt must not be negative");
        c->seed_salt = (unsigned long long)value;
        return RT_OK;
    case RT_OPT_RAW_SUM:
        c->raw_sum = value ? 1 : 0;
        return RT_OK;
    case RT_OPT_TAIL_SPLIT:
        if (value < 0 || value > 63) return fail(c, RT_E_ARG, "rounds 0 (by frame size) or 1..63");
        c->tail_split = (int)value;
        return RT_OK;
    case RT_OPT_COLLAPSE:
        if (value != 0 && value != 1) return fail(c, RT_E_ARG, "collapse 0 or 1");
        if (c->collapse != (int)value) c->scene_on_device = false;
        c->collapse = (int)value;
        return RT_OK;
    case RT_OPT_TAIL_TAPER:
        if (value < 100 || value > 400) return fail(c, RT_E_ARG, "taper 100..400 (percent)");
        c->tail_taper = (int)value;
        return RT_OK;
    case RT_OPT_DETERMINISTIC:
        if (value != 0 && value != 1) return fail(c, RT_E_ARG, "deterministic 0 or 1");
        c->deterministic = (int)value;
        return RT_OK;
    case RT_OPT_TOPOLOGY:
        if (value != 0 && value != 1) return fail(c, RT_E_ARG, "topology 0 or 1");
        if (c->topology != (int)value) c->scene_on_device = false;
        c->topology = (int)value;
        return RT_OK;
    default:
        return fail(c, RT_E_ARG, "unknown option");
    }
}

int rt_get_option(rt_ctx *c, int option, int64_t *value) {
    if (!c || !value) return RT_E_ARG;
    switch (option) {
    case RT_OPT_MAX_DEPTH: *value = c->max_depth; return RT_OK;
    case RT_OPT_HONOR_VFOV: *value = c->honor_vfov; return RT_OK;
    case RT_OPT_LEAF_SIZE: *value = c->leaf_size; return RT_OK;
    case RT_OPT_BOX_MODE: *value = c->box_mode; return RT_OK;
    case RT_OPT_SHARD_RANK: *value = c->shard_rank; return RT_OK;
    case RT_OPT_SHARD_WORLD: *value = c->shard_world; return RT_OK;
    case RT_OPT_KERNEL: *value = c->kernel_variant; return RT_OK;
    case RT_OPT_REFILL_THRESHOLD: *value = c->refill_threshold; return RT_OK;
    case RT_OPT_WAVES_PER_CU: *value = c->waves_per_cu; return RT_OK;
    case RT_OPT_LDS_STACK: *value = c->lds_stack; return RT_OK;
    case RT_OPT_NODE_FORMAT: *value = c->node_format; return RT_OK;
    case RT_OPT_LEAF_BATCH: *value = c->leaf_batch; return RT_OK;
    case RT_OPT_OCCUPANCY: *value = c->occupancy; return RT_OK;
    case RT_OPT_TOPOLOGY: *value = c->topology; return RT_OK;
    case RT_OPT_SHADE_BATCH: *value = c->shade_batch; return RT_OK;
    case RT_OPT_SWAP_BATCH: *value = c->swap_batch; return RT_OK;
    case RT_OPT_TAIL_SPLIT: *value = c->tail_split; return RT_OK;
    case RT_OPT_TAIL_TAPER: *value = c->tail_taper; return RT_OK;
    case RT_OPT_COLLAPSE: *value = c->collapse; return RT_OK;
    case RT_OPT_DETERMINISTIC: *value = c->deterministic; return RT_OK;
    case RT_OPT_SEED_SALT: *value = (int64_t)c->seed_salt; return RT_OK;
    case RT_OPT_RAW_SUM: *value = c->raw_sum; return RT_OK;
    case RT_OPT_QUEUE_THRESHOLD: *value = c->queue_threshold; return RT_OK;
    case RT_OPT_QUEUE_WAVES: *value = c->queue_waves; return RT_OK;
    case RT_OPT_PRIMARY_WAVES: *value = c->primary_waves; return RT_OK;
    case RT_OPT_PRIMARY_CHUNK: *value = c->primary_chunk; return RT_OK;
    default: return fail(c, RT_E_ARG, "unknown option");
    }
}

static int render_into(rt_ctx *c, float *d_fb, rt_stats *st, double t0, bool no_alloc = false) {
    KernelFrame F;
    fill_frame(c, F);
    if (no_alloc) F.ray_counter = reinterpret_cast<unsigned long long *>(c->static_area + kStaticCounterOff);
    HIPCHK(c, hipMemsetAsync(F.ray_counter, 0, sizeof(unsigned long long), c->stream));
    if (c->shard_world > 1) // pixels of other ranks' tiles read as 0, so that a sum over ranks assembles the frame
        HIPCHK(c, hipMemsetAsync(d_fb, 0, (size_t)c->nx * c->ny * 3 * sizeof(float), c->stream));
    PersistLaunch L;
    if (c->kernel_variant >= 1) {
        int rc = prepare_persistent(c, L, kLaunchRender, 0, no_alloc);
        if (rc) return rc;
    }
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    if (c->kernel_variant >= 1)
        HIPCHK(c, launch_persistent(0, scene_of(c), F, L, d_fb, nullptr, nullptr, c->stream));
    else
        HIPCHK(c, launch_render(scene_of(c), F, c->wide.stack_entries, d_fb, c->stream));
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float ms = 0.0f;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    unsigned long long rays = 0;
    HIPCHK(c, hipMemcpy(&rays, F.ray_counter, sizeof(rays), hipMemcpyDeviceToHost));
    fill_stats(c, st, rays, ms, now_ms() - t0);
    if (st && c->kernel_variant >= 1 && L.rounds > 1) { // round records seen torn (detected, polled again): expected 0
        unsigned int torn = 0;
        HIPCHK(c, hipMemcpy(&torn, L.band_counters + kTornCounter, sizeof(torn), hipMemcpyDeviceToHost));
        st->handoff_torn = (int32_t)std::min<unsigned int>(torn, 0x7FFFFFFFu);
    }
    return RT_OK;
}

int rt_render_device_noalloc(rt_ctx *c, float *fb_rgb_device, rt_stats *st) {
    if (!c || !fb_rgb_device) return RT_E_ARG;
    RT_TRY
    double t0 = now_ms();
    if (!c->cam_set) return fail(c, RT_E_ARG, "no camera: call rt_set_camera first");
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->imported && (c->tris.empty() || c->bnodes.empty()))
        return fail(c, RT_E_ARG, "no scene / no BVH: call rt_set_scene and rt_build_bvh first");
    if (c->imported && !c->d_nodes && (c->node_format == 0 || c->kernel_variant == 0))
        return fail(c, RT_E_ARG, "imported scene carries no float nodes");
    // everything a render reads must be resident already: this call uploads and allocates nothing
    if (!c->scene_on_device)
        return fail(c, RT_E_ARG, "the device scene is not resident (a layout option changed since the build): call rt_build_bvh, "
                                 "rt_set_bvh_nodes or rt_render_device first");
    if (c->mats_dirty || c->tex_dirty) return fail(c, RT_E_ARG, "material or texture tables are not resident");
    int rc = check_textures(c);
    if (rc) return rc;
    return render_into(c, fb_rgb_device, st, t0, true);
    RT_CATCH(c)
}

int rt_render_device(rt_ctx *c, float *fb_rgb_device, rt_stats *st) {
    RT_TRY
    if (!c || !fb_rgb_device) return RT_E_ARG;
    double t0 = now_ms();
    if (!c->cam_set) return fail(c, RT_E_ARG, "no camera: call rt_set_camera first");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ready_to_trace(c);
    if (rc) return rc;
    return render_into(c, fb_rgb_device, st, t0);
    RT_CATCH(c)
}

int rt_render(rt_ctx *c, float *fb_rgb_host, rt_stats *st) {
    RT_TRY
    if (!c || !fb_rgb_host) return RT_E_ARG;
    double t0 = now_ms();
    if (!c->cam_set) return fail(c, RT_E_ARG, "no camera: call rt_set_camera first");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ready_to_trace(c);
    if (rc) return rc;
    rc = ensure_fb(c);
    if (rc) return rc;
    rc = render_into(c, c->d_fb, st, t0);
    if (rc) return rc;
    HIPCHK(c, hipMemcpy(fb_rgb_host, c->d_fb, (size_t)c->nx * c->ny * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (st) st->total_ms = now_ms() - t0;
    return RT_OK;
    RT_CATCH(c)
}

int rt_trace_primary(rt_ctx *c, int32_t *tri_id_host, float *t_host, rt_stats *st) {
    RT_TRY
    if (!c || !tri_id_host || !t_host) return RT_E_ARG;
    double t0 = now_ms();
    if (!c->cam_set) return fail(c, RT_E_ARG, "no camera: call rt_set_camera first");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ready_to_trace(c);
    if (rc) return rc;
    size_t px = (size_t)c->nx * c->ny;
    int32_t *d_ids = nullptr;
    float *d_ts = nullptr;
    HIPCHK(c, hipMalloc((void **)&d_ids, px * sizeof(int32_t)));
    if (hipMalloc((void **)&d_ts, px * sizeof(float)) != hipSuccess) {
        (void)hipFree(d_ids);
        return fail(c, RT_E_HIP, "hipMalloc failed");
    }
    auto cleanup = [&]() {
        (void)hipFree(d_ids);
        (void)hipFree(d_ts);
    };
    KernelFrame F;
    fill_frame(c, F);
    PersistLaunch L;
    if (c->kernel_variant >= 1) {
        int rc2 = prepare_persistent(c, L, kLaunchPrimary);
        if (rc2) {
            cleanup();
            return rc2;
        }
    }
    hipError_t e = hipSuccess;
    float ms = 0.0f;
    unsigned long long rays = 0;
    do {
        if ((e = hipMemsetAsync(c->d_counter, 0, sizeof(unsigned long long), c->stream)) != hipSuccess) break;
        if ((e = hipMemsetAsync(d_ids, 0xFF, px * sizeof(int32_t), c->stream)) != hipSuccess) break;
        if ((e = hipMemsetAsync(d_ts, 0, px * sizeof(float), c->stream)) != hipSuccess) break;
        if ((e = hipEventRecord(c->ev0, c->stream)) != hipSuccess) break;
        if (c->kernel_variant >= 1)
            e = launch_persistent(1, scene_of(c), F, L, nullptr, d_ids, d_ts, c->stream);
        else
            e = launch_trace_primary(scene_of(c), F, c->wide.stack_entries, d_ids, d_ts, c->stream);
        if (e != hipSuccess) break;
        if ((e = hipEventRecord(c->ev1, c->stream)) != hipSuccess) break;
        if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) break;
        if ((e = hipEventElapsedTime(&ms, c->ev0, c->ev1)) != hipSuccess) break;
        if ((e = hipMemcpy(&rays, c->d_counter, sizeof(rays), hipMemcpyDeviceToHost)) != hipSuccess) break;
        if ((e = hipMemcpy(tri_id_host, d_ids, px * sizeof(int32_t), hipMemcpyDeviceToHost)) != hipSuccess) break;
        if ((e = hipMemcpy(t_host, d_ts, px * sizeof(float), hipMemcpyDeviceToHost)) != hipSuccess) break;
    } while (0);
    cleanup();
    if (e != hipSuccess) return fail(c, RT_E_HIP, std::string("HIP error in rt_trace_primary: ") + hipGetErrorString(e));
    fill_stats(c, st, rays, ms, now_ms() - t0);
    return RT_OK;
    RT_CATCH(c)
}

int rt_trace_rays(rt_ctx *c, const float *org_dir6, int n_rays, int32_t *tri_id_host, float *t_host, rt_stats *st) {
    RT_TRY
    if (!c || !org_dir6 || !tri_id_host || !t_host || n_rays <= 0) return RT_E_ARG;
    double t0 = now_ms();
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ready_to_trace(c);
    if (rc) return rc;
    float *d_rays = nullptr, *d_ts = nullptr;
    int32_t *d_ids = nullptr;
    hipError_t e = hipSuccess;
    float ms = 0.0f;
    unsigned long long rays = 0;
    do {
        if ((e = hipMalloc((void **)&d_rays, (size_t)n_rays * 6 * sizeof(float))) != hipSuccess) break;
        if ((e = hipMalloc((void **)&d_ids, (size_t)n_rays * sizeof(int32_t))) != hipSuccess) break;
        if ((e = hipMalloc((void **)&d_ts, (size_t)n_rays * sizeof(float))) != hipSuccess) break;
        if ((e = hipMemcpyAsync(d_rays, org_dir6, (size_t)n_rays * 6 * sizeof(float), hipMemcpyHostToDevice, c->stream)) !=
            hipSuccess)
            break;
        if ((e = hipMemsetAsync(c->d_counter, 0, sizeof(unsigned long long), c->stream)) != hipSuccess) break;
        if ((e = hipEventRecord(c->ev0, c->stream)) != hipSuccess) break;
        if (c->kernel_variant >= 1) {
            PersistLaunch L;
            if (prepare_persistent(c, L, kLaunchRays, (size_t)n_rays) != RT_OK) {
                e = hipErrorOutOfMemory;
                break;
            }
            L.threshold = c->queue_threshold;
            if ((e = hipEventRecord(c->ev0, c->stream)) != hipSuccess) break; // after the set-up memsets
            if ((e = launch_trace_queue(scene_of(c), L, d_rays, (uint32_t)n_rays, c->d_bands, d_ids, d_ts, c->d_counter,
                                        c->stream)) != hipSuccess)
                break;
        } else if ((e = launch_trace_rays(scene_of(c), c->wide.stack_entries, d_rays, n_rays, d_ids, d_ts, c->d_counter,
                                          c->stream)) != hipSuccess)
            break;
        if ((e = hipEventRecord(c->ev1, c->stream)) != hipSuccess) break;
        if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) break;
        if ((e = hipEventElapsedTime(&ms, c->ev0, c->ev1)) != hipSuccess) break;
        if ((e = hipMemcpy(&rays, c->d_counter, sizeof(rays), hipMemcpyDeviceToHost)) != hipSuccess) break;
        if ((e = hipMemcpy(tri_id_host, d_ids, (size_t)n_rays * sizeof(int32_t), hipMemcpyDeviceToHost)) != hipSuccess) break;
        if ((e = hipMemcpy(t_host, d_ts, (size_t)n_rays * sizeof(float), hipMemcpyDeviceToHost)) != hipSuccess) break;
    } while (0);
    if (d_rays) (void)hipFree(d_rays);
    if (d_ids) (void)hipFree(d_ids);
    if (d_ts) (void)hipFree(d_ts);
    if (e != hipSuccess) return fail(c, RT_E_HIP, std::string("HIP error in rt_trace_rays: ") + hipGetErrorString(e));
    fill_stats(c, st, rays, ms, now_ms() - t0);
    return RT_OK;
    RT_CATCH(c)
}

int rt_write_ppm(const char *path, const float *fb, int nx, int ny) {
    if (!path || !fb || nx <= 0 || ny <= 0) return RT_E_ARG;
    FILE *f = std::fopen(path, "w");
    if (!f) return RT_E_ARG;
    std::fprintf(f, "P3\n%d %d\n255\n", nx, ny);
    for (int j = ny - 1; j >= 0; j--) {
        for (int i = 0; i < nx; i++) {
            size_t p = 3 * ((size_t)j * nx + i);
            int ir = int(255.99 * fb[p + 0]);
            int ig = int(255.99 * fb[p + 1]);
            int ib = int(255.99 * fb[p + 2]);
            std::fprintf(f, "%d %d %d\n", ir, ig, ib);
        }
    }
    return std::fclose(f) == 0 ? RT_OK : RT_E_ARG;
}

int rt_test_glibc_rand(uint32_t seed, int n, int32_t *out) {
    if (!out || n < 0) return RT_E_ARG;
    GlibcRand g(seed);
    for (int i = 0; i < n; i++) out[i] = g.next();
    return RT_OK;
}

int rt_test_introsort(const float *key, int32_t *idx, int n) {
    if (!key || !idx || n < 0) return RT_E_ARG;
    introsort_by_key(key, idx, n);
    return RT_OK;
}

int rt_test_build_bvh(const rt_triangle148 *tris, int n, int64_t rand_calls_before, rt_bvh_node32 *out) {
    RT_TRY
    if (!tris || !out || n <= 0 || rand_calls_before < 0) return RT_E_ARG;
    std::vector<rt_bvh_node32> nodes;
    build_binary_bvh(tris, n, rand_calls_before, nodes);
    std::memcpy(out, nodes.data(), nodes.size() * sizeof(rt_bvh_node32));
    return RT_OK;
    RT_CATCH(nullptr)
}

int rt_test_wide_layout(const rt_triangle148 *tris, int n, const rt_bvh_node32 *nodes, int m, int leaf_size,
                        int box_mode, int topology, void *wide_nodes, void *quant_nodes, void *prims, int32_t info[4]) {
    RT_TRY
    if (!tris || !nodes || !info) return RT_E_ARG;
    WideBvh w;
    std::string err;
    if (topology < 0 || topology > 3) return RT_E_ARG; // bit 0: RT_OPT_TOPOLOGY, bit 1: RT_OPT_COLLAPSE
    if (!make_device_layout(tris, n, nodes, m, leaf_size, box_mode, topology & 1, w, err, topology >> 1)) return RT_E_ARG;
    info[0] = (int32_t)w.nodes.size();
    info[1] = (int32_t)w.prims.size();
    info[2] = w.depth;
    info[3] = w.stack_entries;
    if (wide_nodes) std::memcpy(wide_nodes, w.nodes.data(), w.nodes.size() * sizeof(WideNode));
    if (quant_nodes) std::memcpy(quant_nodes, w.qnodes.data(), w.qnodes.size() * sizeof(QuantNode));
    if (prims) std::memcpy(prims, w.prims.data(), w.prims.size() * sizeof(DevPrim));
    return RT_OK;
    RT_CATCH(nullptr)
}

int rt_test_pack_layout(const rt_triangle148 *tris, int n, const rt_bvh_node32 *nodes, int m, int leaf_size, int topology,
                        void *records, uint32_t *rec_of_prim, int64_t info[2]) {
    RT_TRY
    if (!tris || !nodes || !info) return RT_E_ARG;
    WideBvh w;
    std::string err;
    if (topology < 0 || topology > 3) return RT_E_ARG;
    if (!make_device_layout(tris, n, nodes, m, leaf_size, 0, topology & 1, w, err, topology >> 1)) return RT_E_ARG;
    PackScene ps;
    std::string why;
    info[0] = info[1] = 0;
    if (!build_pack_scene(w, ps, why)) return RT_OK; // does not fit the format: zero records
    info[0] = (int64_t)ps.recs.size();
    info[1] = (int64_t)w.qnodes.size();
    if (records) std::memcpy(records, ps.recs.data(), ps.recs.size() * sizeof(PackRecord));
    if (rec_of_prim) std::memcpy(rec_of_prim, ps.rec_of_prim.data(), ps.rec_of_prim.size() * sizeof(uint32_t));
    return RT_OK;
    RT_CATCH(nullptr)
}

int rt_export_scene(rt_ctx *c, int flags, void *out, uint64_t cap, uint64_t *bytes) {
    if (!c || !bytes) return RT_E_ARG;
    RT_TRY
    if (flags < 0 || flags > 3) return fail(c, RT_E_ARG, "export flags 0..3");
    if (c->imported) return fail(c, RT_E_ARG, "an imported scene keeps no host copy to export: pass the block it came from on");
    if (c->tris.empty() || c->bnodes.empty()) return fail(c, RT_E_ARG, "no scene / no BVH to export");
    if (c->wide.qnodes.empty() || !c->scene_on_device) { // a layout option changed since the build: lay out again
        HIPCHK(c, hipSetDevice(c->device));
        int rc = upload_scene(c);
        if (rc) return rc;
    }
    if (write_scene_blob(c->wide, c->bnodes, flags, out, cap, bytes) != RT_OK) return fail(c, RT_E_ARG, "export buffer too small");
    return RT_OK;
    RT_CATCH(c)
}

int rt_import_scene(rt_ctx *c, const void *blob, uint64_t bytes) {
    if (!c) return RT_E_ARG;
    RT_TRY
    SceneBlobHeader h;
    SceneBlobLayout L;
    std::string err;
    if (!read_scene_blob_header(blob, bytes, h, L, err) || !validate_scene_blob(blob, h, L, err)) return fail(c, RT_E_ARG, err);
    HIPCHK(c, hipSetDevice(c->device));
    const char *b = static_cast<const char *>(blob);
    free_scene(c);
    c->tris.clear();
    c->tris.shrink_to_fit();
    c->bnodes.clear();
    c->wide = WideBvh();
    c->imported = false;
    HIPCHK(c, hipMalloc(&c->d_qnodes, h.n_wide * sizeof(QuantNode)));
    HIPCHK(c, hipMalloc(&c->d_prims, h.n_prims * sizeof(DevPrim)));
    HIPCHK(c, hipMalloc(&c->d_shade, h.n_prims * sizeof(DevShade)));
    HIPCHK(c, hipMemcpy(c->d_qnodes, b + L.qnodes, h.n_wide * sizeof(QuantNode), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_prims, b + L.prims, h.n_prims * sizeof(DevPrim), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_shade, b + L.shade, h.n_prims * sizeof(DevShade), hipMemcpyHostToDevice));
    if (h.n_uv) {
        HIPCHK(c, hipMalloc(&c->d_uv, h.n_uv * sizeof(DevUV)));
        HIPCHK(c, hipMemcpy(c->d_uv, b + L.uv, h.n_uv * sizeof(DevUV), hipMemcpyHostToDevice));
        c->wide.uv.resize(h.n_uv); // kept on the host: which textures the scene names (check_textures)
        std::memcpy(c->wide.uv.data(), b + L.uv, h.n_uv * sizeof(DevUV));
    }
    if (h.n_float_nodes) {
        HIPCHK(c, hipMalloc(&c->d_nodes, h.n_float_nodes * sizeof(WideNode)));
        HIPCHK(c, hipMemcpy(c->d_nodes, b + L.fnodes, h.n_float_nodes * sizeof(WideNode), hipMemcpyHostToDevice));
    }
    if (h.n_bnodes) c->bnodes.assign(reinterpret_cast<const rt_bvh_node32 *>(b + L.bnodes), reinterpret_cast<const rt_bvh_node32 *>(b + L.bnodes) + h.n_bnodes);
    c->wide.depth = h.depth;
    c->wide.stack_entries = h.stack_entries;
    c->wide.pad = h.pad;
    c->n_wide = h.n_wide;
    c->n_prims = h.n_prims;
    c->imported = true;
    c->scene_on_device = true;
    return RT_OK;
    RT_CATCH(c)
}

int rt_test_scene_blob(const rt_triangle148 *tris, int n, const rt_bvh_node32 *nodes, int m, int leaf_size, int box_mode,
                       int topology, int flags, void *out, uint64_t cap, uint64_t *bytes) {
    if (!tris || !nodes || !bytes || flags < 0 || flags > 3 || topology < 0 || topology > 3) return RT_E_ARG;
    RT_TRY
    WideBvh w;
    std::string err;
    if (!make_device_layout(tris, n, nodes, m, leaf_size, box_mode, topology & 1, w, err, topology >> 1)) return RT_E_ARG;
    std::vector<rt_bvh_node32> bn(nodes, nodes + m);
    return write_scene_blob(w, bn, flags, out, cap, bytes);
    RT_CATCH(nullptr)
}

int rt_test_scene_blob_check(void *blob, uint64_t bytes, int rehash, char *err_out, int err_cap) {
    RT_TRY
    SceneBlobHeader h;
    SceneBlobLayout L;
    std::string err;
    bool ok = read_scene_blob_header(blob, bytes, h, L, err);
    if (ok && rehash) { // tests: make an edited payload pass the checksum, so that the structural checks are what refuses it
        h.payload_hash = blob_hash(static_cast<const char *>(blob) + kBlobAlign, L.total - kBlobAlign);
        std::memcpy(blob, &h, sizeof(h));
    }
    ok = ok && validate_scene_blob(blob, h, L, err);
    if (err_out && err_cap > 0) std::snprintf(err_out, (size_t)err_cap, "%s", ok ? "" : err.c_str());
    return ok ? RT_OK : RT_E_ARG;
    RT_CATCH(nullptr)
}

int rt_test_scene_blob_info(const void *blob, uint64_t bytes, uint64_t info[16]) {
    if (!info) return RT_E_ARG;
    RT_TRY
    SceneBlobHeader h;
    SceneBlobLayout L;
    std::string err;
    if (!read_scene_blob_header(blob, bytes, h, L, err) || !validate_scene_blob(blob, h, L, err)) return RT_E_ARG;
    const uint64_t v[16] = {h.n_wide, h.n_prims, h.n_uv, h.n_float_nodes, h.n_bnodes, h.total_bytes, (uint64_t)h.depth,
                            (uint64_t)h.stack_entries, L.qnodes, L.prims, L.shade, L.uv, L.fnodes, L.bnodes, h.payload_hash, h.layout_version};
    std::memcpy(info, v, sizeof(v));
    return RT_OK;
    RT_CATCH(nullptr)
}

int rt_test_trav_stats(rt_ctx *c, uint64_t out[40]) {
    if (!c || !out) return RT_E_ARG;
#ifdef RT_TRAV_STATS
    if (hipSetDevice(c->device) != hipSuccess) return RT_E_HIP;
    HIPCHK(c, hipMemcpy(out, c->d_bands + 16, sizeof(uint64_t) * kTravStats, hipMemcpyDeviceToHost));
    return RT_OK;
#else
    return fail(c, RT_E_UNSUPPORTED, "built without RT_TRAV_STATS (make STATS=1 builds librt_amd_stats.so)");
#endif
}

int rt_trace_rays_device(rt_ctx *c, const float *org_dir6_device, int n_rays, int32_t *tri_id_device, float *t_device,
                         rt_stats *st) {
    RT_TRY
    if (!c || !org_dir6_device || !tri_id_device || !t_device || n_rays <= 0) return RT_E_ARG;
    double t0 = now_ms();
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ready_to_trace(c);
    if (rc) return rc;
    HIPCHK(c, hipMemsetAsync(c->d_counter, 0, sizeof(unsigned long long), c->stream));
    PersistLaunch L;
    rc = prepare_persistent(c, L, kLaunchRays, (size_t)n_rays);
    if (rc) return rc;
    L.threshold = c->queue_threshold;
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    HIPCHK(c, launch_trace_queue(scene_of(c), L, org_dir6_device, (uint32_t)n_rays, c->d_bands, tri_id_device, t_device,
                                 c->d_counter, c->stream));
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float ms = 0.0f;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    unsigned long long rays = 0;
    HIPCHK(c, hipMemcpy(&rays, c->d_counter, sizeof(rays), hipMemcpyDeviceToHost));
    fill_stats(c, st, rays, ms, now_ms() - t0);
    return RT_OK;
    RT_CATCH(c)
}

int rt_test_dump_rays(rt_ctx *c, uint64_t capacity) {
    if (!c) return RT_E_ARG;
#ifdef RT_TRAV_STATS
    HIPCHK(c, hipSetDevice(c->device));
    if (c->d_ray_dump) (void)hipFree(c->d_ray_dump);
    c->d_ray_dump = nullptr;
    c->ray_dump_cap = 0;
    if (capacity == 0) return RT_OK;
    if (!c->d_ray_dump_count) HIPCHK(c, hipMalloc((void **)&c->d_ray_dump_count, sizeof(unsigned long long)));
    HIPCHK(c, hipMalloc((void **)&c->d_ray_dump, capacity * 6 * sizeof(float)));
    c->ray_dump_cap = capacity;
    return RT_OK;
#else
    (void)capacity;
    return fail(c, RT_E_UNSUPPORTED, "built without RT_TRAV_STATS (make STATS=1 builds librt_amd_stats.so)");
#endif
}

int rt_test_wave_times(rt_ctx *c, uint64_t *out, uint64_t n_waves) {
    if (!c) return RT_E_ARG;
#ifdef RT_TRAV_STATS
    if (!out || n_waves == 0 || n_waves > (1u << 20)) return fail(c, RT_E_ARG, "wave times: out and 1..2^20 waves");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->wave_times_cap < n_waves) {
        if (c->d_wave_times) (void)hipFree(c->d_wave_times);
        c->d_wave_times = nullptr;
        c->wave_times_cap = 0;
        HIPCHK(c, hipMalloc((void **)&c->d_wave_times, n_waves * 3 * sizeof(unsigned long long)));
        HIPCHK(c, hipMemset(c->d_wave_times, 0, n_waves * 3 * sizeof(unsigned long long)));
        c->wave_times_cap = n_waves;
    }
    HIPCHK(c, hipMemcpy(out, c->d_wave_times, n_waves * 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return RT_OK;
#else
    (void)out, (void)n_waves;
    return fail(c, RT_E_UNSUPPORTED, "built without RT_TRAV_STATS (make STATS=1 builds librt_amd_stats.so)");
#endif
}

int rt_test_trace_dumped(rt_ctx *c, rt_stats *st) {
    if (!c) return RT_E_ARG;
    if (!c->d_ray_dump) return fail(c, RT_E_ARG, "no recorded rays: rt_test_dump_rays, then render");
    HIPCHK(c, hipSetDevice(c->device));
    unsigned long long n = 0;
    HIPCHK(c, hipMemcpy(&n, c->d_ray_dump_count, sizeof(n), hipMemcpyDeviceToHost));
    if (n > c->ray_dump_cap) n = c->ray_dump_cap;
    if (n == 0 || n > 0x7FFFFFFFull) return fail(c, RT_E_LIMIT, "recorded ray count out of range");
    int32_t *d_ids = nullptr;
    float *d_ts = nullptr;
    HIPCHK(c, hipMalloc((void **)&d_ids, n * sizeof(int32_t)));
    if (hipMalloc((void **)&d_ts, n * sizeof(float)) != hipSuccess) {
        (void)hipFree(d_ids);
        return fail(c, RT_E_HIP, "out of device memory");
    }
    int rc = rt_trace_rays_device(c, c->d_ray_dump, (int)n, d_ids, d_ts, st);
    (void)hipFree(d_ids);
    (void)hipFree(d_ts);
    return rc;
}

int rt_test_sort_dumped(rt_ctx *c, int mode) {
    if (!c) return RT_E_ARG;
#ifdef RT_TRAV_STATS
    RT_TRY
    if (!c->d_ray_dump) return fail(c, RT_E_ARG, "no recorded rays: rt_test_dump_rays, then render");
    if (mode < 1 || mode > 2) return fail(c, RT_E_ARG, "sort mode 1 (origin cell, then octant) or 2 (octant, then origin cell)");
    HIPCHK(c, hipSetDevice(c->device));
    unsigned long long n = 0;
    HIPCHK(c, hipMemcpy(&n, c->d_ray_dump_count, sizeof(n), hipMemcpyDeviceToHost));
    if (n > c->ray_dump_cap) n = c->ray_dump_cap;
    if (n == 0 || n > 0x7FFFFFFFull) return fail(c, RT_E_LIMIT, "recorded ray count out of range");
    std::vector<float> rays(6 * (size_t)n), sorted(6 * (size_t)n);
    HIPCHK(c, hipMemcpy(rays.data(), c->d_ray_dump, rays.size() * sizeof(float), hipMemcpyDeviceToHost));
    float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
    for (size_t i = 0; i < n; i++)
        for (int a = 0; a < 3; a++) {
            const float v = rays[6 * i + a];
            if (v < lo[a]) lo[a] = v;
            if (v > hi[a]) hi[a] = v;
        }
    auto spread = [](uint32_t v) { // 7 bits -> every third bit
        uint32_t r = 0;
        for (int b = 0; b < 7; b++) r |= ((v >> b) & 1u) << (3 * b);
        return r;
    };
    std::vector<uint64_t> keys(n);
    for (size_t i = 0; i < n; i++) {
        uint32_t cell = 0, oct = 0;
        for (int a = 0; a < 3; a++) {
            const float ext = hi[a] - lo[a];
            int q = ext > 0.0f ? (int)((rays[6 * i + a] - lo[a]) / ext * 128.0f) : 0;
            q = q < 0 ? 0 : (q > 127 ? 127 : q);
            cell |= spread((uint32_t)q) << a;
            oct |= (rays[6 * i + 3 + a] < 0.0f ? 1u : 0u) << a;
        }
        const uint32_t key = mode == 1 ? (cell << 3) | oct : (oct << 21) | cell;
        keys[i] = ((uint64_t)key << 32) | (uint64_t)i;
    }
    std::sort(keys.begin(), keys.end());
    for (size_t i = 0; i < n; i++) std::memcpy(&sorted[6 * i], &rays[6 * (size_t)(uint32_t)keys[i]], 6 * sizeof(float));
    HIPCHK(c, hipMemcpy(c->d_ray_dump, sorted.data(), sorted.size() * sizeof(float), hipMemcpyHostToDevice));
    return RT_OK;
    RT_CATCH(c)
#else
    (void)mode;
    return fail(c, RT_E_UNSUPPORTED, "built without RT_TRAV_STATS (make STATS=1 builds librt_amd_stats.so)");
#endif
}

int rt_test_sincosf(float x, float *s, float *c) {
    if (!s || !c) return RT_E_ARG;
    double sd, cd;
    sincos_sh((double)x, sd, cd);
    *s = (float)sd;
    *c = (float)cd;
    return RT_OK;
}

int rt_test_fastdiv(uint32_t d, const uint32_t *n, int count, uint32_t *q) {
    if (d == 0 || !n || !q || count < 0) return RT_E_ARG;
    const FastDiv f = fastdiv_make(d);
    for (int i = 0; i < count; i++) q[i] = fastdiv_host(f, n[i]);
    return RT_OK;
}

int rt_test_tile_owner(int nx, int ny, int world, int32_t *owner) {
    if (nx <= 0 || ny <= 0 || world <= 0 || !owner) return RT_E_ARG;
    // the rule of tile_xy() in kernels.hip: 8x8 tiles, row-major tile index g, owner = g % world
    const int tiles_x = (nx + 7) / 8;
    for (int j = 0; j < ny; j++)
        for (int i = 0; i < nx; i++) owner[(size_t)j * nx + i] = (int32_t)((((long)(j / 8)) * tiles_x + i / 8) % world);
    return RT_OK;
}

} // extern "C"
