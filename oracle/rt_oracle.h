/*
 * rt_oracle.h -- C interface of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (rt_amd/, include/) may
 * include, link, import or execute anything under oracle/.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as
 * the checker / the reported CPU baseline.
 *
 * PARITY UNPINNED.  The reference (Chang-Liu-TAMU/RT) has no tests, no golden
 * vectors and no recorded outputs for this path, and its
 * sources cannot be compiled in this image without writing stand-in headers for
 * CUDA, cuRAND, glm and GL, which the build rules forbid.  This oracle is
 * therefore a from-scratch restatement of the reference's algorithm, written
 * from reading the source; every function cites the file:line it follows.
 * Third-party arithmetic that is absent from the reference checkout:
 *   - cuRAND XORWOW (CUDA Toolkit, version unpinned by CMakeLists.txt:5):
 *     restated from the published algorithm (Marsaglia xorwow + cuRAND's
 *     seeding constants); pinned only by the known-answer vector recorded in
 *     SURVEY.md section 0.
 *   - CUDA __cosf/__sinf/tanf/powf device intrinsics: not reproducible off
 *     NVIDIA hardware; replaced by the deterministic double-precision
 *     routines specified in DESIGN.md ("shared transcendentals").
 *   - glibc rand() and libstdc++ std::sort: used directly here (the BVH's
 *     shape is defined by them, pbrt_engine.cu:335-368, triangle.h:12-15).
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference Triangle, include/RT/triangle.h:345-354 (148 bytes on x86-64) */
typedef struct {
    float v0[3], v1[3], v2[3];
    float normal[3];
    float edge1[3], edge2[3];
    float u[3], v[3];
    float n0[3], n1[3], n2[3];
    int32_t calNormal, matType, shapeType, texIdx;
} o_tri148;

/* reference bvh_node, include/RT/triangle.h:364-391 (32 bytes) */
typedef struct {
    float mn[3], mx[3];
    int32_t left, right;
} o_node32;

/* POD form of the reference's polymorphic material table, material.h:86-218 */
enum { O_LAMBERTIAN = 0, O_METAL = 1, O_DIELECTRIC = 2, O_LIGHT = 3 };
typedef struct {
    int32_t kind;
    float albedo[3];
    float param; /* metal: fuzz; dielectric: ref_idx */
} o_mat;

/* reference camera, include/RT/rt_camera.h:62-69 (96 bytes) */
typedef struct {
    float origin[3], lower_left[3], horizontal[3], vertical[3];
    float u[3], v[3], w[3];
    float lens_radius, time0, time1;
} o_camera;

typedef struct {
    uint64_t rays;      /* hit() calls */
    uint64_t box_tests; /* aabb::hit calls */
    uint64_t tri_tests; /* Triangle::hit calls */
} o_counts;

/* -- RNG: cuRAND XORWOW restatement (call sites pbrt_engine.cu:580,599) -- */
void o_xorwow_init(uint64_t seed, uint32_t state[6]);
uint32_t o_xorwow_next(uint32_t state[6]);
float o_xorwow_uniform(uint32_t state[6]);

/* -- shared transcendentals -- */
float o_cosf(float x);
float o_sinf(float x);
float o_tanf(float x);
float o_pow5f(float x);

/* -- host-side Triangle constructors, triangle.h:90-111 -- */
void o_make_triangle(const float v0[3], const float v1[3], const float v2[3], int matType, int texIdx,
                     o_tri148 *out);
void o_make_sphere(const float c[3], float r, int matType, o_tri148 *out);

/* -- material_init, pbrt_engine.cu:25-47, stream seeded 1995 (:599) -- */
void o_material_table(o_mat out[50]);

/* -- make_bvh_nodes, pbrt_engine.cu:335-368; srand(1) (= glibc default state) first.
 *    out must hold 2n-1 nodes.  returns node count. -- */
/* -- textures (lambertian albedo when a triangle's texIdx != -1; material.h:100-106, utilities.cu:4-9): RGBA8,
 *    w*h texels, row 0 first.  A process-wide table: the oracle is a test tool. -- */
#define O_TEX_NUM 16
int o_set_texture(int idx, const unsigned char *rgba, int w, int h);
void o_tex_fetch(int idx, float u, float v, float out[3]);

int o_build_bvh(const o_tri148 *tris, int n, o_node32 *out);
/* same, with rand() advanced by rand_calls_before draws first (the reference shares the unseeded rand() between
 * scene setup, main.cpp:449-474, and every build of a session) */
int o_build_bvh_after(const o_tri148 *tris, int n, long long rand_calls_before, o_node32 *out);

/* -- camera::update, rt_camera.h:37-52 (vfov is what the caller passes; the
 *    reference's camera_update kernel always passes 40, pbrt_engine.cu:629) -- */
void o_camera_update(o_camera *cam, const float lookfrom[3], const float lookat[3], const float vup[3], float vfov,
                     float aspect, float aperture, float focus_dist, float t0, float t1);

/* -- hit(), pbrt_engine.cu:371-456, reference visit pattern.  returns 1 on hit.
 *    out_id = index into tris, out_t = rec.t -- */
int o_hit(const o_tri148 *tris, const o_node32 *nodes, int n_nodes, const float org[3], const float dir[3],
          int32_t *out_id, float *out_t, o_counts *cnt);

/* -- DFS (left-first) leaf rank of every triangle in the tree; rank[n_tris] -- */
void o_dfs_rank(const o_node32 *nodes, int n_nodes, int n_tris, int32_t *rank);

/* -- brute force over all triangles (the reference's LINEAR_ITER idea, pbrt_engine.cu:372-380),
 *    closest t, ties to the lowest DFS-leaf rank given in rank[] (NULL: lowest index). -- */
int o_hit_linear(const o_tri148 *tris, int n, const int32_t *rank, const float org[3], const float dir[3],
                 int32_t *out_id, float *out_t);

/* -- ordered, t-shrinking closest hit on the same binary tree; counts the box and
 *    triangle tests that define the algorithmic bytes per ray. -- */
int o_hit_ordered(const o_tri148 *tris, const o_node32 *nodes, int n_nodes, const float org[3], const float dir[3],
                  int32_t *out_id, float *out_t, o_counts *cnt);

/* -- render(), pbrt_engine.cu:569-595, over the pixel lattice
 *      i = x0, x0+sx, ... < nx ; j = y0, y0+sy, ... < ny
 *    (sx = sy = 1, x0 = y0 = 0 is the whole frame).  fb is nx*ny*3 floats, row j=0 at the
 *    bottom, untouched outside the lattice.  mode: 0 = reference visit pattern,
 *    1 = ordered traversal (same image, fewer tests).  threads >= 1. -- */
void o_render(const o_tri148 *tris, int n_tris, const o_node32 *nodes, int n_nodes, const o_mat *mats,
              const o_camera *cam, int nx, int ny, int ns, int max_depth, int x0, int y0, int sx, int sy, int mode,
              int threads, float *fb, o_counts *cnt);

/* -- primary ray of sample 0 of every lattice pixel: hit id (-1 = miss) and t.
 *    ids/ts are nx*ny, untouched outside the lattice. -- */
void o_trace_primary(const o_tri148 *tris, int n_tris, const o_node32 *nodes, int n_nodes, const o_camera *cam, int nx,
                     int ny, int x0, int y0, int sx, int sy, int mode, int threads, int32_t *ids, float *ts,
                     o_counts *cnt);

/* -- analysis: log every ray the following single-threaded o_render calls trace into buf (8 floats per ray: origin,
 *    direction, hit id as int32 bits (-1 = miss), t; at most cap rays are stored).  Returns the number of rays seen
 *    since the previous call; buf = NULL switches the log off. -- */
uint64_t o_set_ray_log(float *buf, uint64_t cap);

/* -- PPM writer, pbrt_engine.cu:123-133 -- */
int o_write_ppm(const char *path, const float *fb, int nx, int ny);

/* -- glibc rand() passthrough for pinning the product's restatement -- */
void o_libc_rand_seq(unsigned seed, int n, int32_t *out);
/* -- std::sort passthrough (sorts idx[] by key[idx] with operator<) -- */
void o_std_sort_by_key(const float *key, int32_t *idx, int n);

#ifdef __cplusplus
}
#endif
#endif
