/*
 * rt_abi.h -- C ABI of the MI355X trace-loop library (librt_amd.so).
 *
 * This is the drop-in boundary for ONE hot path of Chang-Liu-TAMU/RT: the per-pixel trace
 * loop (camera ray generation -> BVH traversal -> ray/triangle + ray/sphere hit -> material
 * scatter -> sample accumulation).  Each entry point names the reference interface it replaces
 * (paths relative to the reference root).  Plain C: pointers and sizes only, no C++ or torch
 * types.  Every call returns 0 on success and a non-zero RT_E_* code on failure; nothing exits
 * or throws across the boundary (the reference prints and exit(99)s, include/RT/common.h:26-36).
 * rt_last_error() returns the text of the last failure on a context.
 *
 * One context drives one GPU.  Calls on one context must not overlap; create / resize / render
 * may arrive from different host threads (the reference calls them from its main, console and
 * worker threads: main.cpp:195-220,712; src/pbrt/pbrt_engine.cu:148), so every call makes its
 * device current first.
 *
 * There is NO CPU fallback: if no HIP device is usable, rt_create fails.
 */
#ifndef RT_ABI_H
#define RT_ABI_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 1

enum {
    RT_OK = 0,
    RT_E_ARG = 1,      /* bad argument / call order */
    RT_E_HIP = 2,      /* HIP runtime error (text in rt_last_error) */
    RT_E_UNSUPPORTED = 3, /* input outside the supported subset (textures, sphere as tri[0], ...) */
    RT_E_LIMIT = 4     /* a size limit was exceeded */
};

/* The reference's Triangle record, taken unchanged: include/RT/triangle.h:345-354
 * (v0,v1,v2,normal,edge1,edge2,u,v,n0,n1,n2 as float[3]; calNormal,matType,shapeType,texIdx).
 * shapeType 0 = triangle, 1 = sphere (centre v0, radius v1[0]).  148 bytes. */
typedef struct rt_triangle148 {
    float v0[3], v1[3], v2[3];
    float normal[3];
    float edge1[3], edge2[3];
    float u[3], v[3];
    float n0[3], n1[3], n2[3];
    int32_t calNormal, matType, shapeType, texIdx;
} rt_triangle148;

/* The reference's bvh_node: include/RT/triangle.h:364-391.  Post-order array, root last;
 * leaf iff left == right, and then left is the triangle index.  32 bytes. */
typedef struct rt_bvh_node32 {
    float box_min[3], box_max[3];
    int32_t left, right;
} rt_bvh_node32;

/* POD form of one entry of the reference's device-heap material table
 * (include/RT/material.h:86-218; table built by material_init, pbrt_engine.cu:25-47). */
enum { RT_MAT_LAMBERTIAN = 0, RT_MAT_METAL = 1, RT_MAT_DIELECTRIC = 2, RT_MAT_LIGHT = 3 };
typedef struct rt_material {
    int32_t kind;
    float albedo[3];
    float param; /* metal: fuzz (already clamped to <= 1); dielectric: ref_idx */
} rt_material;
#define RT_MAT_NUM 50 /* MAT_NUM, include/RT/pbrt_engine.h:28 */
#define RT_TEX_NUM 16 /* textures a context can hold (the reference keeps a vector of texture objects, texture.cpp:45-86) */

typedef struct rt_stats {
    uint64_t rays;        /* closest-hit queries (= hit() calls, pbrt_engine.cu:371) of the last launch */
    double kernel_ms;     /* duration of the trace kernel, HIP events on the context's stream */
    double total_ms;      /* whole call, host clock */
    uint64_t wide_nodes;  /* device BVH size */
    uint64_t prims;
    int32_t stack_entries; /* LDS stack entries per lane */
    int32_t handoff_torn; /* render in rounds: polls of a pixel-state record that found an 8-byte granule whose two words
                             were not from one store (detected by the check word, read again); 0 is what gfx950 has shown */
} rt_stats;

typedef struct rt_ctx rt_ctx;

/* PbrtEngine::getInstance / ctor (include/RT/pbrt_engine.h:53-58, pbrt_engine.cu:65-67).
 * device_id: HIP device ordinal.  Defaults as the reference: 640x360, 10 spp, depth 50. */
int rt_create(rt_ctx **out, int device_id);
/* ~PbrtEngine / __releaseMemory (pbrt_engine.h:60-62,164-182) */
int rt_destroy(rt_ctx *ctx);
const char *rt_last_error(rt_ctx *ctx);
int rt_abi_version(void);

/* setNx/setNy/setNs + uponPixelNumChange (pbrt_engine.h:65-84; refreshPixelNum pbrt_engine.cu:70-86).
 * (Re)allocates the device framebuffer.  The reference's per-pixel global scratch
 * (__allocateCudaStackMemory, pbrt_engine.cu:219-236) has no counterpart: stacks live in LDS. */
int rt_resize(rt_ctx *ctx, int nx, int ny, int ns);

/* m_vecTriangles after __genTrianglesFromScene (pbrt_engine.cu:167-178): n records, the mesh
 * named "light" first.  Copied during the call.  tri[0] must be a triangle (it is the hard-coded
 * light of pbrt_engine.cu:464,502).  A triangle's texIdx is -1 (no texture) or names a texture set with
 * rt_set_texture before the next render; its u[] / v[] fields are then the vertices' texture coordinates. */
int rt_set_scene(rt_ctx *ctx, const rt_triangle148 *tris, int n);

/* __genBvhNodes (pbrt_engine.cu:181-202): bit-exact host builder (make_bvh_nodes, :335-368, with
 * glibc rand() in its default state and libstdc++ std::sort, both restated), then wide-ify and
 * upload.  rand_calls_before: how many rand() calls preceded the build in the caller's process
 * (the reference shares rand() with scene setup, main.cpp:449-474); 0 = default state. */
int rt_build_bvh(rt_ctx *ctx, int64_t rand_calls_before);
/* Inject a node array built elsewhere (e.g. by the reference's own make_bvh_nodes) instead. */
int rt_set_bvh_nodes(rt_ctx *ctx, const rt_bvh_node32 *nodes, int m);
/* Read back the binary node array (m = 2n-1). cap = capacity of out in nodes. */
int rt_get_bvh_nodes(rt_ctx *ctx, rt_bvh_node32 *out, int cap, int *m);

/* The built device scene (wide nodes, re-ordered primitives, shading records, texture coordinates) as ONE relocatable
 * block of bytes, so that a scene is built once per node and handed to the other GPUs' contexts instead of every rank
 * flattening, building (pbrt_engine.cu:181-202 on every rank) and laying it out again: BASELINE config 5's 100 M
 * triangles are 14.8 GB of records and a 47 s all-core build per rank otherwise.  No reference counterpart.
 * rt_export_scene: out == NULL returns the size in *bytes; flags bit 0 = include the 128-byte float nodes
 * (RT_OPT_NODE_FORMAT 0 / RT_OPT_KERNEL 0 need them), bit 1 = include the reference's binary node array (so that
 * rt_get_bvh_nodes works on the importing side).  rt_import_scene replaces the context's scene with the block's
 * (copied to the device during the call; the block may be a read-only shared mapping) after checking it on the host:
 * layout version and record sizes of the writing build, a checksum of the payload, every child reference and primitive
 * range, the depth against the announced stack (a stale, torn or foreign block is RT_E_ARG, not a fault in the kernel); materials, textures, camera
 * and options are not part of it.  An imported scene cannot be re-laid-out: changing RT_OPT_LEAF_SIZE / BOX_MODE /
 * TOPOLOGY / COLLAPSE afterwards is an error at the next render. */
int rt_export_scene(rt_ctx *ctx, int flags, void *out, uint64_t cap, uint64_t *bytes);
int rt_import_scene(rt_ctx *ctx, const void *blob, uint64_t bytes);

/* One texture object (Texture::getTexture -> cudaCreateTextureObject, src/cxx/texture.cpp:45-86): RGBA8 texels, w*h*4
 * bytes, row 0 first, copied during the call.  Sampled as the reference's descriptor says: normalised coordinates,
 * wrap addressing, linear filtering, channels as byte/255 (utilities.cu:4-9); a lambertian or a diffusive_light
 * triangle with texIdx != -1 reads it (material.h:100-106, 140-146; metal and dielectric ignore it).  Image decoding
 * stays with the caller (the reference uses stb). */
int rt_set_texture(rt_ctx *ctx, int index, const uint8_t *rgba8, int w, int h);

/* material_init + rand_init (pbrt_engine.cu:25-47,597-601) as a host-side table. */
int rt_default_materials(rt_material out[RT_MAT_NUM]);
int rt_set_materials(rt_ctx *ctx, const rt_material *mats, int n);

/* camera_update (pbrt_engine.cu:626-632) -> camera::update (include/RT/rt_camera.h:37-52).
 * vfov is accepted and, like the reference (:629), ignored: the field of view is 40 degrees
 * unless RT_OPT_HONOR_VFOV is set.  aspect = nx/ny of the last rt_resize.  Non-finite parameters are refused
 * (RT_E_ARG): the reference would trace NaN rays, whose result is outside the parity claim. */
int rt_set_camera(rt_ctx *ctx, const float lookfrom[3], const float lookat[3], float vfov, float dist_to_focus,
                  float aperture);

enum {
    RT_OPT_MAX_DEPTH = 1,   /* ITER_NUM, pbrt_engine.cu:458 (default 50) */
    RT_OPT_HONOR_VFOV = 2,  /* default 0 */
    RT_OPT_LEAF_SIZE = 3,   /* primitives per device leaf, 1..8 (default 2); applies at next build */
    RT_OPT_BOX_MODE = 4,    /* 0 = re-fitted boxes + one small pad (default), 1 = the reference's
                               cumulatively padded boxes (triangle.h:28-31,76-85) */
    RT_OPT_SHARD_RANK = 5,  /* multi-GPU tile sharding: this context renders the 8x8 tiles with */
    RT_OPT_SHARD_WORLD = 6, /*   tile_index % world == rank; other pixels are written as 0 */
    RT_OPT_KERNEL = 7,      /* kernel variant, see DESIGN.md: 0 = one wave per tile, 1 = persistent waves
                               with per-lane refill, 2 = 1 + speculative traversal (postponed leaves; default).
                               (3, two paths per lane, was a measured negative and is refused: branch exp/k-dual) */
    RT_OPT_REFILL_THRESHOLD = 8, /* kernel 1: refill when fewer than this many of 64 lanes still traverse */
    RT_OPT_WAVES_PER_CU = 9,     /* kernel 1: persistent waves launched per CU */
    RT_OPT_LDS_STACK = 10,       /* kernel 1: stack entries per lane kept in LDS (the rest spill to HBM) */
    RT_OPT_NODE_FORMAT = 11,     /* kernel 1/2: 0 = 128-byte float nodes, 1 = 64-byte quantized nodes (default),
                                  * 2 (kernel 2 only) = the packed scene: 48-byte nodes and the primitives in one record
                                  * array, three loads per node visit instead of four; a tree that does not fit it
                                  * (leaves of more than two primitives, 4 GiB or more) and an imported scene run 1 */
    RT_OPT_LEAF_BATCH = 12,      /* kernel 2: primitive tests run when this many lanes have a leaf pending */
    RT_OPT_OCCUPANCY = 13,       /* kernel 1/2, quantized nodes: waves per SIMD the registers are held to: 4, 6, 8 */
    RT_OPT_TOPOLOGY = 14,        /* device hierarchy: 0 = the reference's tree, wide-ified; 1 = a binned-SAH tree
                                    re-built over the same primitives (default).  Results are identical;
                                    rt_get_bvh_nodes always returns the reference's tree. */
    RT_OPT_SHADE_BATCH = 15,     /* accepted and ignored: they tuned kernel variant 3, which left the library */
    RT_OPT_SWAP_BATCH = 16,
    RT_OPT_TAIL_SPLIT = 17,      /* kernel 1/2 render: the frame is walked in this many rounds (1..63, at most ns; default 0 = 2..8 by pixels per lane),
                                    each running ns/rounds samples of every pixel and handing the pixel's state on to
                                    whichever lane takes it next round, so that the frame ends on fractions of a pixel
                                    instead of whole pixels; 1 = off.  Same image either way. */
    /* Sample-pass sharding (NOT a parity path -- the reference's per-pixel stream is serial over samples,
     * so splitting a pixel's samples over GPUs needs other streams): */
    RT_OPT_SEED_SALT = 18,       /* added to the per-pixel seed 1995 + pixel_index (default 0 = the reference's streams) */
    RT_OPT_RAW_SUM = 19,         /* 1: the frame holds per-pixel SUMS of the (per-sample clamped) samples instead of the
                                    clamped mean, so that ranks' frames can be added and divided by the total count */
    RT_OPT_QUEUE_THRESHOLD = 20, /* rt_trace_rays*: a lane takes its next ray when fewer than this many of 64 still traverse */
    RT_OPT_QUEUE_WAVES = 21,     /* rt_trace_rays*: most persistent waves per CU, 1..32 (default 24; a short array gets fewer) */
    RT_OPT_PRIMARY_WAVES = 22,   /* rt_trace_primary*: most persistent waves per CU, 1..32 (default 16; a 1080p frame gets 8) */
    RT_OPT_PRIMARY_CHUNK = 23,   /* rt_trace_primary*: pixel slots a wave takes per atomic, a multiple of 64 (default 256) */
    RT_OPT_TAIL_TAPER = 24,      /* rounds: sample boundary r at ns (1 - (1 - r/R)^(value/100)); 100 = equal shares, more = short last rounds (default 150) */
    RT_OPT_COLLAPSE = 25,        /* device tree: 0 = wide nodes by largest-area expansion, 1 = by least surface-area cost (dynamic programme; default) */
    RT_OPT_DETERMINISTIC = 26    /* 1 (default): a ray's traversal is a function of the ray and the tree alone, so a frame is the same bytes
                                    launch after launch, whatever the rounds, waves and thresholds.  0: box tests cull against the freshest
                                    hit, 5 % faster on the bench frame, and about one ray in 10^10 -- a grazing ray whose epsilon-free
                                    Moeller-Trumbore t (triangle.h:137-156) lies outside its own triangle's box -- may resolve either way
                                    depending on which lanes share its wave.  Either way those rays can differ from the
                                    reference, which culls nothing. */
};
int rt_set_option(rt_ctx *ctx, int option, int64_t value);
int rt_get_option(rt_ctx *ctx, int option, int64_t *value);

/* render<<<>>> + cudaDeviceSynchronize + D2H copy (pbrt_engine.cu:110-122).  fb_rgb: nx*ny*3
 * floats, pixel_index = j*nx + i, row j = 0 at the BOTTOM (the PPM writer flips).  Synchronous. */
int rt_render(rt_ctx *ctx, float *fb_rgb_host, rt_stats *stats);
/* Same, but the frame stays in device memory the caller owns (nx*ny*3 floats on this context's
 * device); the call returns after the kernel has completed. */
int rt_render_device(rt_ctx *ctx, float *fb_rgb_device, rt_stats *stats);

/* rt_render_device for callers that cannot allocate device memory: tight serving loops, harnesses that forbid
 * hipMalloc, processes that must keep their device-memory footprint fixed.  No counterpart in the reference.  It is
 * NOT usable under HIP stream capture: like rt_render_device it runs on the context's own stream, issues
 * hipMemsetAsync for its counters (and, once per 2^26 frames of a device, for the record area), synchronises that
 * stream, reads the event timer and copies the ray counter back -- all illegal or unrecorded while a capture is open.  Same arguments, same frame (bit for bit), same
 * kernels as rt_render_device, but every scratch buffer of the launch -- band counters, ray counter, the traversal
 * stacks' spill area, the round records -- comes out of ONE static device array of 64 MiB that the library holds on
 * every GPU it runs on until the process ends, and the call itself makes no hipMalloc / hipFree and uploads nothing:
 * scene, BVH, materials and textures must be resident (rt_build_bvh / rt_set_bvh_nodes, rt_set_materials and
 * rt_set_texture upload when they are called; after a layout option changed, build again or call rt_render_device
 * once).  Limits, refused with RT_E_LIMIT and a message: frames of more than 314 572 pixel slots (8x8 tiles x 64;
 * the reference's default 640x360 fits, 1280x720 does not) when the frame is rendered in rounds (RT_OPT_TAIL_SPLIT
 * != 1; with 1 any frame size runs), and scenes whose traversal stacks spill more than 40 MiB - 4 KiB (stack entries
 * beyond RT_OPT_LDS_STACK x RT_OPT_WAVES_PER_CU x CUs x 256 bytes; the 1 M-triangle bench scene spills 37.7 MB).  All contexts of one GPU share that GPU's array: calls of
 * this entry point that overlap in time on one GPU race; calls in sequence need no clearing by the caller. */
int rt_render_device_noalloc(rt_ctx *ctx, float *fb_rgb_device, rt_stats *stats);

/* Parity hook (no reference counterpart): the primary ray of sample 0 of every pixel
 * (pbrt_engine.cu:584-586), closest hit via hit() (:371).  tri_id = index into the rt_set_scene
 * array, -1 on a miss; t = rec.t (0 on a miss).  Host buffers of nx*ny. */
int rt_trace_primary(rt_ctx *ctx, int32_t *tri_id_host, float *t_host, rt_stats *stats);
/* Closest hit for caller-supplied rays (origin[3], direction[3] per ray, host memory). */
int rt_trace_rays(rt_ctx *ctx, const float *org_dir6, int n_rays, int32_t *tri_id_host, float *t_host,
                  rt_stats *stats);

/* The same for rays and results that live in this context's device memory (no copies; n_rays * 6 floats in,
 * n_rays ids and n_rays t out).  Persistent waves with per-lane refill from the ray array. */
int rt_trace_rays_device(rt_ctx *ctx, const float *org_dir6_device, int n_rays, int32_t *tri_id_device, float *t_device,
                         rt_stats *stats);

/* PPM writer of __renderScene (pbrt_engine.cu:123-133): "P3", rows top-down from j = ny-1,
 * int(255.99*c), one pixel per line, no gamma. */
int rt_write_ppm(const char *path, const float *fb_rgb, int nx, int ny);

/* Test hooks for the restated toolchain pieces the builder depends on. */
int rt_test_glibc_rand(uint32_t seed, int n, int32_t *out);
int rt_test_introsort(const float *key, int32_t *idx, int n);
/* The host builder alone (no GPU needed): out must hold 2n-1 nodes. */
int rt_test_build_bvh(const rt_triangle148 *tris, int n, int64_t rand_calls_before, rt_bvh_node32 *out);
/* The device layout alone (no GPU needed).  Sizes first (pass NULL buffers), then the copies:
 * wide_nodes: 128 bytes each; quant_nodes: 64 bytes each (same indices); prims: 48 bytes each (DFS leaf order); info[4] = {n_wide, n_prims,
 * depth, stack_entries}.  topology: bit 0 = RT_OPT_TOPOLOGY, bit 1 = RT_OPT_COLLAPSE. */
int rt_test_wide_layout(const rt_triangle148 *tris, int n, const rt_bvh_node32 *nodes, int m, int leaf_size,
                        int box_mode, int topology, void *wide_nodes, void *quant_nodes, void *prims, int32_t info[4]);

/* The packed scene of RT_OPT_NODE_FORMAT 2 alone (no GPU needed; box mode 0).  Sizes first (NULL buffers): info[0] =
 * records (0: the tree does not fit the format, e.g. leaves of more than two primitives), info[1] = wide nodes among
 * them.  records: 48 bytes each; rec_of_prim[p] = record of primitive p of rt_test_wide_layout's prims. */
int rt_test_pack_layout(const rt_triangle148 *tris, int n, const rt_bvh_node32 *nodes, int m, int leaf_size, int topology,
                        void *records, uint32_t *rec_of_prim, int64_t info[2]);

/* rt_export_scene's block from records and nodes alone, and a block's header (no GPU needed): info = {wide nodes,
 * primitives, uv records, float nodes, binary nodes, total bytes, depth, stack entries, then the byte offsets of the
 * six sections}; RT_E_ARG for a block that rt_import_scene would refuse. */
int rt_test_scene_blob(const rt_triangle148 *tris, int n, const rt_bvh_node32 *nodes, int m, int leaf_size, int box_mode,
                       int topology, int flags, void *out, uint64_t cap, uint64_t *bytes);
int rt_test_scene_blob_info(const void *blob, uint64_t bytes, uint64_t info[16]);
/* What rt_import_scene checks before anything reaches the GPU, alone (no GPU needed): the header (magic, layout version,
 * record sizes of the writing build, counts against the length), the payload's checksum, and one pass over the sections
 * (every child reference names a later node or a primitive range inside the array, the implied depth fits the announced
 * stack, material / texture indices in range).  RT_OK, or RT_E_ARG with the reason in err_out.  rehash != 0 first
 * rewrites the header's checksum from the payload as it is (tests: an edited block that still passes the checksum). */
int rt_test_scene_blob_check(void *blob, uint64_t bytes, int rehash, char *err_out, int err_cap);

/* The tile-sharding rule alone (no GPU needed): owner[j*nx+i] = rank that renders pixel (i, j) when the frame
 * is split over `world` contexts (RT_OPT_SHARD_RANK / RT_OPT_SHARD_WORLD). */
int rt_test_tile_owner(int nx, int ny, int world, int32_t *owner);
/* The kernels' division of a tile index by tiles_x (multiply-high and shifts, no GPU needed): q[i] = n[i] / d. */
int rt_test_fastdiv(uint32_t d, const uint32_t *n, int count, uint32_t *q);

/* The library's sine / cosine (binary64 range reduction + Taylor, rounded once to float): what the
 * host-side scene code (rt_scene.hpp, scene_graph.py) uses for glm::rotate so that both produce the same matrices. */
int rt_test_sincosf(float x, float *s, float *c);

/* Lane-occupancy counters of the last persistent-kernel launch (analysis builds only: `make STATS=1` builds
 * librt_amd_stats.so; the shipped library returns RT_E_UNSUPPORTED).  Layout: rt_amd/csrc/kernels.h kStat*. */
int rt_test_trav_stats(rt_ctx *ctx, uint64_t out[40]);

/* Analysis builds only (make STATS=1): record every ray of the next renders in device memory (capacity rays; 0 frees
 * the buffer), and trace the recorded rays again with rt_trace_rays_device's kernel -- how long the traversal alone
 * takes when taking a ray costs nothing but its loads (tools/trav_stats.py --replay). */
int rt_test_dump_rays(rt_ctx *ctx, uint64_t capacity);
int rt_test_trace_dumped(rt_ctx *ctx, rt_stats *stats);
/* Analysis builds only: re-order the recorded rays on the host -- mode 1: by origin cell (128^3 Morton code), then
 * direction octant; mode 2: by octant, then cell -- the most coherent order any ray-sorting scheme could hand the
 * traversal kernel; rt_test_trace_dumped then prices it (tools/trav_stats.py --replay ... --sort). */
int rt_test_sort_dumped(rt_ctx *ctx, int mode);

/* Analysis builds only: for every wave of the last render, on the constant 100 MHz device clock: out[3 w] = start,
 * out[3 w + 1] = end, out[3 w + 2] = when it first found every band handed out; for the first n_waves waves; the first
 * call switches the recording on.  Shows how
 * long the end of a frame runs on thinning waves (tools/trav_stats.py --wave-times). */
int rt_test_wave_times(rt_ctx *ctx, uint64_t *out, uint64_t n_waves);

#ifdef __cplusplus
}
#endif
#endif
