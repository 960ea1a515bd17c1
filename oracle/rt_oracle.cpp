/*
 * rt_oracle.cpp -- CPU restatement of the per-pixel trace loop of Chang-Liu-TAMU/RT.
 *
 * TEST INFRASTRUCTURE ONLY (see rt_oracle.h).  PARITY UNPINNED (see rt_oracle.h).
 *
 * Written from scratch from a reading of the reference; no reference text is
 * copied.  Every block cites the reference file:line it follows (paths relative
 * to the reference root).  Arithmetic is IEEE binary32 unless a line promotes to
 * double exactly where the reference's C++ expression does.  Build with
 * -ffp-contract=off (see Makefile): the reference's host results are defined
 * without fused multiply-add.
 *
 * Evaluation-order rule: wherever the reference draws several random numbers
 * inside one argument list (pbrt_engine.cu:41,44; material.h:25; rt_camera.h:12)
 * the order is unspecified by C++; this restatement draws left to right
 * (x, then y, then z, then fuzz), see SURVEY.md section 0.
 */
#include "rt_oracle.h"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <climits>
#include <thread>
#include <vector>

namespace {

/* ------------------------------------------------------------------ vec3 --
 * include/RT/vec3.h:17-169.  Operation order is part of the contract. */
struct V {
    float x, y, z;
};
inline V mk(float a, float b, float c) { return V{a, b, c}; }
inline V ld(const float *p) { return V{p[0], p[1], p[2]}; }
inline void st(float *p, V v) {
    p[0] = v.x;
    p[1] = v.y;
    p[2] = v.z;
}
inline V add(V a, V b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }   /* vec3.h:81 */
inline V sub(V a, V b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }   /* vec3.h:85 */
inline V mul(V a, V b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }   /* vec3.h:89 */
inline V scl(float t, V v) { return mk(t * v.x, t * v.y, t * v.z); }     /* vec3.h:97,105 */
inline V divs(V v, float t) { return mk(v.x / t, v.y / t, v.z / t); }    /* vec3.h:101: three divides */
inline V neg(V v) { return mk(-v.x, -v.y, -v.z); }                       /* vec3.h:33 */
inline float dot(V a, V b) { return a.x * b.x + a.y * b.y + a.z * b.z; } /* vec3.h:109 */
inline V cross(V a, V b) {                                               /* vec3.h:113-117 */
    return mk(a.y * b.z - a.z * b.y, -(a.x * b.z - a.z * b.x), a.x * b.y - a.y * b.x);
}
inline float sqlen(V v) { return v.x * v.x + v.y * v.y + v.z * v.z; } /* vec3.h:45 */
inline float len(V v) { return std::sqrt(sqlen(v)); }                 /* vec3.h:44 (float sqrt) */
inline V unit(V v) { return divs(v, len(v)); }                        /* vec3.h:167 */
inline float fabs_ref(float f) { return (double)f > 0.0 ? f : -f; }   /* vec3.h:58 cuAbs */
inline void clip_color(V &v) {                                        /* vec3.h:52-56 */
    v.x = (double)v.x > 1.0 ? 1.0f : v.x;
    v.y = (double)v.y > 1.0 ? 1.0f : v.y;
    v.z = (double)v.z > 1.0 ? 1.0f : v.z;
}

const double REF_PI = 3.1415926; /* vec3.h:16 overrides math.h's M_PI */
const float SHIFT_F = (float)1e-2; /* macros.h:5-7; converted to float in the vec3 ctor (triangle.h:29-30) */

/* ------------------------------------------------------------------ RNG --
 * cuRAND XORWOW (third party, absent from the checkout): Marsaglia's xorwow with
 * cuRAND's seeding for curand_init(seed, 0, 0, &s).  Call sites:
 * pbrt_engine.cu:580 (per pixel, seed 1995+pixel_index) and :599 (table, 1995). */
struct Rng {
    uint32_t v[5];
    uint32_t d;
};
inline void rng_init(Rng &r, uint64_t seed) {
    uint32_t s0 = (uint32_t)seed ^ 0xaad26b49u;
    uint32_t s1 = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
    uint32_t t0 = 1099087573u * s0;
    uint32_t t1 = 2591861531u * s1;
    r.d = 6615241u + t1 + t0;
    r.v[0] = 123456789u + t0;
    r.v[1] = 362436069u ^ t0;
    r.v[2] = 521288629u + t1;
    r.v[3] = 88675123u ^ t1;
    r.v[4] = 5783321u + t0;
}
inline uint32_t rng_next(Rng &r) {
    uint32_t t = r.v[0] ^ (r.v[0] >> 2);
    r.v[0] = r.v[1];
    r.v[1] = r.v[2];
    r.v[2] = r.v[3];
    r.v[3] = r.v[4];
    r.v[4] = (r.v[4] ^ (r.v[4] << 4)) ^ (t ^ (t << 1));
    r.d += 362437u;
    return r.v[4] + r.d;
}
/* curand_uniform: (0,1] */
inline float rng_uniform(Rng &r) { return (float)rng_next(r) * 2.3283064e-10f + (2.3283064e-10f / 2.0f); }

/* ---------------------------------------------------- shared transcendentals --
 * Stand in for CUDA's __cosf/__sinf (utilities.cu:23-24), tan (rt_camera.h:43) and
 * pow (material.h:10), none of which is reproducible off NVIDIA hardware.
 * Specification (DESIGN.md "shared transcendentals"): evaluate in binary64 without
 * FMA: k = floor(x*2/pi + 1/2); r = (x - k*PIO2_HI) - k*PIO2_LO; Taylor polynomials
 * of degree 13 (sin) and 14 (cos) in Horner form on r*r; quadrant fix-up; round to
 * binary32 once. */
inline void sincos_d(double x, double &s, double &c) {
    const double TWO_OVER_PI = 0.63661977236758134308;
    const double PIO2_HI = 1.57079632679489655800e+00;
    const double PIO2_LO = 6.12323399573676603587e-17;
    double kd = std::floor(x * TWO_OVER_PI + 0.5);
    double r = (x - kd * PIO2_HI) - kd * PIO2_LO;
    double z = r * r;
    double ps = -1.0 / 6.0 +
                z * (1.0 / 120.0 +
                     z * (-1.0 / 5040.0 + z * (1.0 / 362880.0 + z * (-1.0 / 39916800.0 + z * (1.0 / 6227020800.0)))));
    double sr = r + r * z * ps;
    double pc = -0.5 + z * (1.0 / 24.0 +
                            z * (-1.0 / 720.0 +
                                 z * (1.0 / 40320.0 +
                                      z * (-1.0 / 3628800.0 + z * (1.0 / 479001600.0 + z * (-1.0 / 87178291200.0))))));
    double cr = 1.0 + z * pc;
    long long k = (long long)kd;
    switch (k & 3) {
    case 0: s = sr; c = cr; break;
    case 1: s = cr; c = -sr; break;
    case 2: s = -sr; c = -cr; break;
    default: s = -cr; c = sr; break;
    }
}
inline float cos_sh(float x) {
    double s, c;
    sincos_d((double)x, s, c);
    return (float)c;
}
inline float sin_sh(float x) {
    double s, c;
    sincos_d((double)x, s, c);
    return (float)s;
}
inline float tan_sh(float x) {
    double s, c;
    sincos_d((double)x, s, c);
    return (float)(s / c);
}
inline float pow5_sh(float x) {
    double d = (double)x;
    double d2 = d * d;
    double d4 = d2 * d2;
    return (float)(d4 * d);
}

/* ------------------------------------------------------------- geometry -- */
struct Box {
    V mn, mx;
};
inline Box padded_box(V a, V b) { /* aabb ctor, triangle.h:28-31 */
    Box r;
    r.mn = add(a, mk(-SHIFT_F, -SHIFT_F, -SHIFT_F));
    r.mx = add(b, mk(SHIFT_F, SHIFT_F, SHIFT_F));
    return r;
}
inline float fmin_ref(float a, float b) { return a < b ? a : b; } /* triangle.h:17 */
inline float fmax_ref(float a, float b) { return a > b ? a : b; } /* triangle.h:21 */

inline Box tri_box(const o_tri148 &t) { /* Triangle::bounding_box, triangle.h:250-273 */
    if (t.shapeType == 0) {
        float x1 = fmin_ref(fmin_ref(t.v0[0], t.v1[0]), t.v2[0]);
        float x2 = fmax_ref(fmax_ref(t.v0[0], t.v1[0]), t.v2[0]);
        float y1 = fmin_ref(fmin_ref(t.v0[1], t.v1[1]), t.v2[1]);
        float y2 = fmax_ref(fmax_ref(t.v0[1], t.v1[1]), t.v2[1]);
        float z1 = fmin_ref(fmin_ref(t.v0[2], t.v1[2]), t.v2[2]);
        float z2 = fmax_ref(fmax_ref(t.v0[2], t.v1[2]), t.v2[2]);
        return padded_box(mk(x1, y1, z1), mk(x2, y2, z2));
    }
    V c = ld(t.v0);
    float r = t.v1[0];
    return padded_box(sub(c, mk(r, r, r)), add(c, mk(r, r, r)));
}
inline Box union_box(const Box &a, const Box &b) { /* surrounding_box, triangle.h:76-85 (re-padded) */
    V small = mk(std::fmin(a.mn.x, b.mn.x), std::fmin(a.mn.y, b.mn.y), std::fmin(a.mn.z, b.mn.z));
    V big = mk(std::fmax(a.mx.x, b.mx.x), std::fmax(a.mx.y, b.mx.y), std::fmax(a.mx.z, b.mx.z));
    return padded_box(small, big);
}

struct Ray {
    V o, d;
    float time;
};

/* aabb::hit, triangle.h:52-70, including the line-63 quirk (compares t0, assigns t1) */
inline bool box_hit_ref(const o_node32 &nd, const Ray &r, float tmin, float tmax) {
    const float o[3] = {r.o.x, r.o.y, r.o.z};
    const float d[3] = {r.d.x, r.d.y, r.d.z};
    for (int a = 0; a < 3; a++) {
        float inv = 1.0f / d[a];
        float t0 = (nd.mn[a] - o[a]) * inv;
        float t1 = (nd.mx[a] - o[a]) * inv;
        if (inv < 0.0f) {
            float s = t0;
            t0 = t1;
            t1 = s;
        }
        tmin = t0 > tmin ? t0 : tmin;
        tmax = t0 < tmax ? t1 : tmax;
        if (tmax <= tmin) return false;
    }
    return true;
}

/* the textbook slab test on the same t0/t1 arithmetic; returns entry distance */
inline bool box_hit_std(const o_node32 &nd, const Ray &r, float tmin, float tmax, float &tnear) {
    const float o[3] = {r.o.x, r.o.y, r.o.z};
    const float d[3] = {r.d.x, r.d.y, r.d.z};
    for (int a = 0; a < 3; a++) {
        float inv = 1.0f / d[a];
        float t0 = (nd.mn[a] - o[a]) * inv;
        float t1 = (nd.mx[a] - o[a]) * inv;
        if (inv < 0.0f) {
            float s = t0;
            t0 = t1;
            t1 = s;
        }
        tmin = t0 > tmin ? t0 : tmin;
        tmax = t1 < tmax ? t1 : tmax;
        if (tmax < tmin) return false;
    }
    tnear = tmin;
    return true;
}

/* hit_record, material.h:39-59 */
struct Rec {
    float t;
    V p, n;
    int mat;
    int state; /* 0 valid, 1 invalid */
    float u, v;
    int tex;
    bool front;
    int tri; /* oracle bookkeeping: which primitive */
};

/* Triangle::hit, triangle.h:132-218 */
inline bool prim_hit(const o_tri148 &T, int tri_index, const Ray &r, float tmin, float tmax, Rec &rec) {
    rec.state = 1; /* rec.reset(), triangle.h:134 */
    if (T.shapeType == 0) {
        V v0 = ld(T.v0), e1 = ld(T.edge1), e2 = ld(T.edge2);
        V tvec = sub(r.o, v0);
        V pvec = cross(r.d, e2);
        float det = dot(e1, pvec);
        det = 1.0f / det; /* no zero check, triangle.h:139-141 */
        float u = dot(tvec, pvec) * det;
        if (u < 0.0f || u > 1.0f) return false;
        V qvec = cross(tvec, e1);
        float v = dot(r.d, qvec) * det;
        if (v < 0.0f || (u + v) > 1.0f) return false;
        float t = dot(e2, qvec) * det;
        if (t <= tmin || t >= tmax) return false;
        V nrm = ld(T.normal); /* smooth normal is computed into a shadowed variable, triangle.h:158-161 */
        rec.t = t;
        rec.p = add(r.o, scl(t, r.d)); /* ray.h:21-23 */
        if (dot(r.d, nrm) < 0.0f) {
            rec.n = nrm;
            rec.front = true;
        } else {
            rec.n = neg(nrm);
            rec.front = false;
        }
        rec.mat = T.matType;
        rec.state = 0;
        { /* setUV, triangle.h:113-118: 1.0 - x - y in double */
            float z = (float)(1.0 - (double)u - (double)v);
            rec.u = z * T.u[0] + u * T.u[1] + v * T.u[2];
            rec.v = z * T.v[0] + u * T.v[1] + v * T.v[2];
        }
        rec.tex = T.texIdx;
        rec.tri = tri_index;
        return true;
    } else if (T.shapeType == 1) { /* triangle.h:180-213 */
        V c = ld(T.v0);
        float radius = T.v1[0];
        V oc = sub(r.o, c);
        float a = dot(r.d, r.d);
        float b = (float)(2.0 * (double)dot(oc, r.d));
        float cc = dot(oc, oc) - radius * radius;
        float disc = b * b - 4 * a * cc;
        if (disc > 0) {
            /* both "roots" use the minus sign (triangle.h:189,200): the far root is never tried */
            for (int k = 0; k < 2; k++) {
                float tem = (float)((double)(-b - std::sqrt(disc)) / (2.0 * (double)a));
                if (tem < tmax && tem > tmin) {
                    rec.t = tem;
                    rec.p = add(r.o, scl(tem, r.d));
                    rec.n = divs(sub(rec.p, c), radius);
                    rec.mat = T.matType;
                    rec.state = 0;
                    rec.front = true;
                    rec.tex = -1;
                    rec.tri = tri_index;
                    return true;
                }
            }
        }
        return false;
    }
    return false;
}

struct Scene {
    const o_tri148 *tris;
    int n_tris;
    const o_node32 *nodes;
    int n_nodes;
    std::vector<int32_t> rank; /* DFS leaf rank per triangle (ordered mode) */
};

void dfs_rank(const o_node32 *nodes, int n_nodes, int n_tris, int32_t *rank) {
    for (int i = 0; i < n_tris; i++) rank[i] = INT32_MAX;
    if (n_nodes <= 0) return;
    std::vector<int> stk;
    stk.push_back(n_nodes - 1);
    int next = 0;
    while (!stk.empty()) {
        int c = stk.back();
        stk.pop_back();
        const o_node32 &nd = nodes[c];
        if (nd.left == nd.right) {
            if (nd.left >= 0 && nd.left < n_tris) rank[nd.left] = next++;
        } else {
            stk.push_back(nd.right);
            stk.push_back(nd.left);
        }
    }
}

const int ORACLE_STACK = 96; /* private scratch; the reference's 20/25 overflow above 2^19 triangles */

/* hit(), pbrt_engine.cu:371-456: explicit-stack DFS from root = n_nodes-1, every reached leaf
 * tested against the FIXED interval, pairwise merge on unwind (ties -> earlier DFS leaf). */
bool hit_reference(const Scene &sc, const Ray &r, float tmin, float tmax, Rec &out, o_counts &cnt) {
    int node_stack[ORACLE_STACK];
    bool came_as_left[ORACLE_STACK];
    Rec rec_stack[ORACLE_STACK];
    int sp = 0, rp = 0, fp = 0;
    int curr = sc.n_nodes - 1;
    bool at_leaf = false;
    bool branch = true;
    cnt.rays++;
    while (curr != -1 || sp) {
        at_leaf = false;
        while (curr != -1) {
            const o_node32 &nd = sc.nodes[curr];
            node_stack[sp++] = curr;
            came_as_left[fp++] = branch;
            cnt.box_tests++;
            if (!box_hit_ref(nd, r, tmin, tmax)) {
                rec_stack[rp++].state = 1;
                curr = -1;
                at_leaf = true;
                continue;
            }
            if (nd.left == nd.right) {
                cnt.tri_tests++;
                prim_hit(sc.tris[nd.left], nd.left, r, tmin, tmax, rec_stack[rp++]);
                at_leaf = true;
                curr = -1;
            } else {
                curr = nd.left;
                branch = true;
            }
        }
        --sp;
        bool flag = came_as_left[--fp];
        if (!at_leaf) {
            /* the reference's variable named "left" is the most recently pushed record,
             * i.e. the RIGHT child's; "right" is the LEFT child's and is also the output slot */
            Rec &second = rec_stack[--rp];
            Rec &first = rec_stack[--rp];
            rp++;
            if (second.state == 0 && first.state == 0) {
                if (second.t < first.t) first = second;
            } else if (first.state == 0) {
                /* keep */
            } else if (second.state == 0) {
                first = second;
            }
        }
        if (sp) {
            int parent = node_stack[sp - 1];
            if (flag) {
                curr = sc.nodes[parent].right;
                branch = false;
            }
        }
    }
    out = rec_stack[0];
    return rec_stack[0].state == 0;
}

/* ordered, t-shrinking closest hit on the same tree; same answer (min t, ties to the lower
 * DFS leaf rank), fewer tests.  Defines the algorithmic bytes of SURVEY.md 8(d). */
bool hit_ordered(const Scene &sc, const Ray &r, float tmin, float tmax, Rec &out, o_counts &cnt) {
    struct Ent {
        int node;
        float tnear;
    };
    Ent stk[ORACLE_STACK * 2];
    int sp = 0;
    cnt.rays++;
    out.state = 1;
    if (sc.n_nodes <= 0) return false;
    float best_t = tmax;
    int best_rank = INT32_MAX;
    bool found = false;
    float tn;
    cnt.box_tests++;
    if (!box_hit_std(sc.nodes[sc.n_nodes - 1], r, tmin, best_t, tn)) return false;
    stk[sp++] = Ent{sc.n_nodes - 1, tn};
    Rec tmp;
    while (sp) {
        Ent e = stk[--sp];
        if (e.tnear > best_t) continue;
        const o_node32 &nd = sc.nodes[e.node];
        if (nd.left == nd.right) {
            cnt.tri_tests++;
            if (prim_hit(sc.tris[nd.left], nd.left, r, tmin, tmax, tmp)) {
                int rk = sc.rank[nd.left];
                if (!found || tmp.t < best_t || (tmp.t == best_t && rk < best_rank)) {
                    found = true;
                    best_t = tmp.t;
                    best_rank = rk;
                    out = tmp;
                }
            }
            continue;
        }
        float tl, tr;
        cnt.box_tests += 2;
        bool hl = box_hit_std(sc.nodes[nd.left], r, tmin, best_t, tl);
        bool hr = box_hit_std(sc.nodes[nd.right], r, tmin, best_t, tr);
        if (hl && hr) {
            if (tr < tl) {
                stk[sp++] = Ent{nd.left, tl};
                stk[sp++] = Ent{nd.right, tr};
            } else {
                stk[sp++] = Ent{nd.right, tr};
                stk[sp++] = Ent{nd.left, tl};
            }
        } else if (hl) {
            stk[sp++] = Ent{nd.left, tl};
        } else if (hr) {
            stk[sp++] = Ent{nd.right, tr};
        }
    }
    return found;
}

inline bool scene_hit(const Scene &sc, int mode, const Ray &r, float tmin, float tmax, Rec &out, o_counts &cnt) {
    return mode == 0 ? hit_reference(sc, r, tmin, tmax, out, cnt) : hit_ordered(sc, r, tmin, tmax, out, cnt);
}

/* ---------------------------------------------------------------- camera --
 * camera::update, rt_camera.h:37-52 */
void camera_update(o_camera &c, V lookfrom, V lookat, V vup, float vfov, float aspect, float aperture, float focus,
                   float t0, float t1) {
    c.time0 = t0;
    c.time1 = t1;
    c.lens_radius = aperture / 2.0f;
    float theta = (vfov / 180.f) * (float)REF_PI;
    float half_height = tan_sh(theta / 2.0f);
    float half_width = aspect * half_height;
    V w = unit(sub(lookfrom, lookat));
    V u = unit(cross(vup, w));
    V v = cross(w, u);
    V ll = sub(sub(sub(lookfrom, scl(half_width * focus, u)), scl(half_height * focus, v)), scl(focus, w));
    V hor = scl(2.0f * half_width * focus, u);
    V ver = scl(2.0f * half_height * focus, v);
    st(c.origin, lookfrom);
    st(c.lower_left, ll);
    st(c.horizontal, hor);
    st(c.vertical, ver);
    st(c.u, u);
    st(c.v, v);
    st(c.w, w);
}

/* random_in_unit_disk_d, rt_camera.h:9-15 (x drawn first) */
inline V disk_sample(Rng &g) {
    V p;
    do {
        float a = rng_uniform(g);
        float b = rng_uniform(g);
        p = sub(scl(2.0f, mk(a, b, 0.0f)), mk(1.0f, 1.0f, 0.0f));
    } while (dot(p, p) >= 1.0f);
    return p;
}

/* camera::get_ray, rt_camera.h:54-60 */
inline Ray camera_ray(const o_camera &c, float s, float t, Rng &g) {
    V rd = scl(c.lens_radius, disk_sample(g));
    V cu = ld(c.u), cv = ld(c.v), org = ld(c.origin);
    V offset = add(scl(rd.x, cu), scl(rd.y, cv));
    float random_t = c.time0 + (c.time1 - c.time0) * rng_uniform(g);
    Ray r;
    r.o = add(org, offset);
    r.d = sub(sub(add(add(ld(c.lower_left), scl(s, ld(c.horizontal))), scl(t, ld(c.vertical))), org), offset);
    r.time = random_t;
    return r;
}

/* ------------------------------------------------------------- sampling --
 * onb::build_from_w, utilities.h:46-57 */
struct Onb {
    V u, v, w;
};
inline Onb onb_from_w(V n) {
    Onb b;
    b.w = unit(n);
    V a = ((double)fabs_ref(b.w.x) > 0.9) ? mk(0, 1, 0) : mk(1, 0, 0);
    b.v = unit(cross(b.w, a));
    b.u = cross(b.w, b.v);
    return b;
}
/* random_cosine_direction, utilities.cu:18-26 (the factor 2 on x and y is the reference's) */
inline V cosine_direction(Rng &g) {
    float r1 = rng_uniform(g);
    float r2 = rng_uniform(g);
    float z = std::sqrt(1 - r2);
    float phi = (float)(2 * REF_PI * (double)r1);
    float x = (float)((double)cos_sh(phi) * 2.0 * (double)std::sqrt(r2));
    float y = (float)((double)sin_sh(phi) * 2.0 * (double)std::sqrt(r2));
    return mk(x, y, z);
}
/* cosine_pdf::generate / value, utilities.cu:52-59; onb::local utilities.h:42-44 */
inline V cosine_generate(const Onb &b, Rng &g) {
    V a = cosine_direction(g);
    return add(add(scl(a.x, b.u), scl(a.y, b.v)), scl(a.z, b.w));
}
inline float cosine_value(const Onb &b, V dir) {
    float c = dot(unit(dir), b.w);
    return (float)(c > 0 ? (double)c / REF_PI : 0.0);
}
/* random_in_unit_sphere, material.h:27-33 (x, y, z drawn in that order) */
inline V ball_sample(Rng &g) {
    V p;
    do {
        float a = rng_uniform(g);
        float b = rng_uniform(g);
        float c = rng_uniform(g);
        p = sub(scl(2.0f, mk(a, b, c)), mk(1, 1, 1));
    } while (sqlen(p) >= 1.0f);
    return p;
}
inline V reflect_ref(V v, V n) { return sub(v, scl(2.0f * dot(v, n), n)); } /* material.h:35-37 */
inline bool refract_ref(V v, V n, float ni_over_nt, V &out) {               /* material.h:13-23 */
    V uv = unit(v);
    float dt = dot(uv, n);
    float disc = 1.0f - ni_over_nt * ni_over_nt * (1 - dt * dt);
    if (disc > 0) {
        out = sub(scl(ni_over_nt, sub(uv, scl(dt, n))), scl(std::sqrt(disc), n));
        return true;
    }
    return false;
}
inline float schlick_ref(float cosine, float ref_idx) { /* material.h:7-11 */
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * pow5_sh(1.0f - cosine);
}

/* Triangle::pdf_value on tri[0], triangle.h:275-318 (evaluated at pbrt_engine.cu:518, weight 0) */
float light_pdf_value(const Scene &sc, V o, V v) {
    const o_tri148 &T = sc.tris[0];
    if (T.shapeType == 0) {
        float x0 = 213, z0 = 227, x1 = 343, z1 = 332;
        float area = (x1 - x0) * (z1 - z0);
        float distance2 = sqlen(v);
        float cosine = fabs_ref(unit(v).y);
        if ((double)cosine < 0.000001) return 0;
        return distance2 / (cosine * area);
    } else if (T.shapeType == 1) {
        Rec tmp;
        Ray r{o, v, 0.0f};
        if (prim_hit(T, 0, r, (float)0.001, FLT_MAX, tmp)) {
            float rr = T.v1[0];
            float cos_theta_max = std::sqrt(1 - rr * rr / sqlen(sub(ld(T.v0), o)));
            float solid_angle = (float)(2 * REF_PI * (double)(1 - cos_theta_max));
            return 1 / solid_angle;
        }
        return 0;
    }
    return 0.0f;
}

/* tex2D<float4>(tex, u, v) of getAttenuationFromTexture (src/pbrt/utilities.cu:4-9) on the texture objects that
 * src/cxx/texture.cpp:74-85 creates: RGBA8 read as normalised float, normalised coordinates, wrap addressing, linear
 * filtering.  CUDA's filter is hardware (weights in 1.8 fixed point; the rest unspecified): like __cosf it cannot be
 * reproduced off NVIDIA hardware, so ONE deterministic restatement of the published formula (CUDA C Programming
 * Guide, "Texture Fetching": xB = x - 0.5, i = floor(xB), alpha = frac(xB) in 8 fractional bits, the four-texel blend)
 * is shared by the oracle and the product, each implementing it independently.  Parity unpinned. */
struct OTexture {
    std::vector<unsigned char> rgba;
    int w = 0, h = 0;
};
static OTexture g_tex[O_TEX_NUM];

static inline float tex_coord(float u, int n, int &i0, int &i1) { /* returns the quantised weight of texel i1 */
    if (!(std::fabs(u) <= 3.0e38f)) u = 0.0f;      /* NaN / inf: defined as 0 */
    float x = (u - std::floor(u)) * (float)n;      /* wrap, then scale to texels */
    float xb = x - 0.5f;
    float fl = std::floor(xb);
    float a = xb - fl;
    a = std::floor(a * 256.0f + 0.5f) / 256.0f;    /* 1.8 fixed point, round to nearest */
    int i = (int)fl;                                /* -1 .. n-1 */
    i0 = ((i % n) + n) % n;
    i1 = (i0 + 1) % n;
    return a;
}
static V tex_fetch(int idx, float u, float v) {
    const OTexture &T = g_tex[idx];
    int i0, i1, j0, j1;
    float a = tex_coord(u, T.w, i0, i1), b = tex_coord(v, T.h, j0, j1);
    float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    float out[3];
    for (int c = 0; c < 3; c++) {
        auto tx = [&](int i, int j) { return (float)T.rgba[4 * ((size_t)j * T.w + i) + c] / 255.0f; };
        out[c] = ((w00 * tx(i0, j0) + w10 * tx(i1, j0)) + w01 * tx(i0, j1)) + w11 * tx(i1, j1);
    }
    return mk(out[0], out[1], out[2]);
}

/* analysis hook (o_set_ray_log): every ray path_color traces, with its result, 8 floats each:
 * origin, direction, hit id (as float bits, -1 = miss), t.  Single-threaded renders only. */
float *g_ray_log = nullptr;
uint64_t g_ray_log_cap = 0, g_ray_log_n = 0;
inline void log_ray(const Ray &r, bool hit, const Rec &rec) {
    if (!g_ray_log) return;
    if (g_ray_log_n < g_ray_log_cap) {
        float *o = g_ray_log + 8 * g_ray_log_n;
        o[0] = r.o.x, o[1] = r.o.y, o[2] = r.o.z, o[3] = r.d.x, o[4] = r.d.y, o[5] = r.d.z;
        int32_t id = hit ? rec.tri : -1;
        std::memcpy(&o[6], &id, 4);
        o[7] = hit ? rec.t : 0.0f;
    }
    g_ray_log_n++;
}

/* color(), pbrt_engine.cu:459-566 */
V path_color(const Scene &sc, const o_mat *mats, int max_depth, int mode, const Ray &primary, Rng &g, o_counts &cnt) {
    Ray cur = primary;
    V thr = mk(1.0f, 1.0f, 1.0f);
    for (int i = 0; i < max_depth; i++) {
        Rec rec;
        const bool was_hit = scene_hit(sc, mode, cur, 0.001f, FLT_MAX, rec, cnt);
        log_ray(cur, was_hit, rec);
        if (was_hit) {
            const o_mat &m = mats[rec.mat];
            V alb = ld(m.albedo);
            if (m.kind == O_LAMBERTIAN) { /* material.h:95-116 */
                Onb b = onb_from_w(rec.n);
                V atten = rec.tex == -1 ? alb : tex_fetch(rec.tex, rec.u, rec.v); /* :100-106 */
                (void)rng_uniform(g); /* pbrt_engine.cu:505, thres = 1 */
                Ray sc_ray;
                sc_ray.o = rec.p;
                sc_ray.d = cosine_generate(b, g);
                sc_ray.time = primary.time; /* :506 takes the ORIGINAL ray's time */
                float thres = 1.0f;
                float pdf = thres * cosine_value(b, sc_ray.d) + (1 - thres) * light_pdf_value(sc, rec.p, sc_ray.d);
                if ((double)pdf < 1e-6) return mk(0, 0, 0); /* :519-521, unclipped */
                float cosine = dot(rec.n, unit(sc_ray.d));  /* lambertian::scatter_pdf, material.h:111-116 */
                if ((double)cosine < 0.0) cosine = 0.0f;
                float spdf = (float)((double)cosine / 3.1415926);
                thr = mul(thr, divs(scl(spdf, atten), pdf)); /* :523 */
                cur = sc_ray;
            } else if (m.kind == O_METAL) { /* material.h:163-171 */
                V reflected = reflect_ref(unit(cur.d), rec.n);
                V fuzzv = ball_sample(g);
                Ray sp;
                sp.o = rec.p;
                sp.d = add(reflected, scl(m.param, fuzzv));
                sp.time = cur.time;
                bool ok = dot(sp.d, rec.n) > 0.0f;
                thr = mul(thr, alb); /* :485 and :544 both multiply by the attenuation */
                if (!ok) break;
                cur = sp;
            } else if (m.kind == O_DIELECTRIC) { /* material.h:180-215 */
                float ref_idx = m.param;
                V outward;
                V reflected = reflect_ref(cur.d, rec.n);
                float ni_over_nt;
                V refracted = mk(0, 0, 0);
                float reflect_prob;
                float cosine;
                float dn = dot(cur.d, rec.n);
                if (dn > 0.0f) {
                    outward = neg(rec.n);
                    ni_over_nt = ref_idx;
                    cosine = dn / len(cur.d);
                    cosine = std::sqrt(1.0f - ref_idx * ref_idx * (1 - cosine * cosine));
                } else {
                    outward = rec.n;
                    ni_over_nt = 1.0f / ref_idx;
                    cosine = -dn / len(cur.d);
                }
                if (refract_ref(cur.d, outward, ni_over_nt, refracted))
                    reflect_prob = schlick_ref(cosine, ref_idx);
                else
                    reflect_prob = 1.0f;
                Ray sp;
                sp.o = rec.p;
                sp.time = cur.time;
                if (rng_uniform(g) < reflect_prob)
                    sp.d = reflected;
                else
                    sp.d = refracted;
                thr = mul(thr, mk(1.0f, 1.0f, 1.0f));
                cur = sp;
            } else { /* diffusive_light, material.h:137-148: scatter() false; attenuation = albedo, or the texture
                      * when the hit triangle names one (:140-146), multiplied in at pbrt_engine.cu:544 */
                V atten = rec.tex == -1 ? alb : tex_fetch(rec.tex, rec.u, rec.v);
                thr = mul(thr, atten);
                break;
            }
        } else { /* sky, pbrt_engine.cu:554-558 */
            V ud = unit(cur.d);
            float t = 0.5f * (ud.y + 1.0f);
            V c = add(scl(1.0f - t, mk(1.0f, 1.0f, 1.0f)), scl(t, mk(0.5f, 0.7f, 1.0f)));
            thr = mul(thr, c);
            break;
        }
    }
    clip_color(thr); /* per-SAMPLE clamp, :564 */
    return thr;
}

inline Ray primary_ray(const o_camera &cam, int i, int j, int nx, int ny, Rng &g) { /* pbrt_engine.cu:584-586 */
    float u = ((float)i + rng_uniform(g)) / (float)nx;
    float v = ((float)j + rng_uniform(g)) / (float)ny;
    return camera_ray(cam, u, v, g);
}

template <class F> void for_rows(int ny, int y0, int sy, int threads, F f) {
    std::vector<int> rows;
    for (int j = y0; j < ny; j += sy) rows.push_back(j);
    if (threads <= 1) {
        for (size_t k = 0; k < rows.size(); k++) f(rows[k], 0);
        return;
    }
    std::atomic<size_t> next(0);
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++)
        pool.emplace_back([&, t]() {
            for (;;) {
                size_t k = next.fetch_add(1);
                if (k >= rows.size()) break;
                f(rows[k], t);
            }
        });
    for (auto &th : pool) th.join();
}

} // namespace

/* ========================================================== C interface == */
extern "C" {

void o_xorwow_init(uint64_t seed, uint32_t state[6]) {
    Rng r;
    rng_init(r, seed);
    std::memcpy(state, &r, sizeof(r));
}
uint32_t o_xorwow_next(uint32_t state[6]) {
    Rng r;
    std::memcpy(&r, state, sizeof(r));
    uint32_t x = rng_next(r);
    std::memcpy(state, &r, sizeof(r));
    return x;
}
float o_xorwow_uniform(uint32_t state[6]) {
    Rng r;
    std::memcpy(&r, state, sizeof(r));
    float x = rng_uniform(r);
    std::memcpy(state, &r, sizeof(r));
    return x;
}

float o_cosf(float x) { return cos_sh(x); }
float o_sinf(float x) { return sin_sh(x); }
float o_tanf(float x) { return tan_sh(x); }
float o_pow5f(float x) { return pow5_sh(x); }

void o_make_triangle(const float v0[3], const float v1[3], const float v2[3], int matType, int texIdx, o_tri148 *out) {
    /* Triangle(v0,v1,v2,mat,tex,u,v) -> setTriangle, triangle.h:90-111; member defaults :345-354 */
    std::memset(out, 0, sizeof(*out));
    V a = ld(v0), b = ld(v1), c = ld(v2);
    st(out->v0, a);
    st(out->v1, b);
    st(out->v2, c);
    V e1 = sub(b, a), e2 = sub(c, a);
    st(out->edge1, e1);
    st(out->edge2, e2);
    st(out->normal, unit(cross(e1, e2)));
    out->calNormal = -1;
    out->matType = matType;
    out->shapeType = 0;
    out->texIdx = texIdx;
}
void o_make_sphere(const float c[3], float r, int matType, o_tri148 *out) { /* triangle.h:95-100 */
    std::memset(out, 0, sizeof(*out));
    st(out->v0, ld(c));
    out->v1[0] = r;
    out->calNormal = -1;
    out->matType = matType;
    out->shapeType = 1;
    out->texIdx = 0; /* member default; Triangle::hit reports -1 for spheres regardless */
}

void o_material_table(o_mat out[50]) { /* material_init, pbrt_engine.cu:25-47 */
    auto set = [&](int i, int kind, float r, float g, float b, float p) {
        out[i].kind = kind;
        out[i].albedo[0] = r;
        out[i].albedo[1] = g;
        out[i].albedo[2] = b;
        out[i].param = p;
    };
    auto fuzz = [](float f) { return (double)f < 1.0 ? f : 1.0f; }; /* metal ctor, material.h:160-162 */
    set(0, O_LAMBERTIAN, 0.5f, 0.5f, 0.5f, 0);
    set(1, O_LAMBERTIAN, 0.2f, 0.4f, 0.1f, 0);
    set(2, O_DIELECTRIC, 1, 1, 1, 0.4f);
    set(3, O_LIGHT, 20.0f, 20.0f, 20.0f, 0);
    set(4, O_METAL, 1.0f, (float)(182.0 / 255.0), (float)(193.0 / 255.0), fuzz((float)0.01));
    set(5, O_METAL, 0.8f, 0.85f, 0.88f, fuzz(0.0f));
    set(6, O_METAL, 0.7f, 0.6f, 0.5f, fuzz(0.0f));
    set(7, O_DIELECTRIC, 1, 1, 1, 1.5f);
    set(8, O_LAMBERTIAN, 0.65f, 0.05f, 0.05f, 0);
    set(9, O_LAMBERTIAN, 0.12f, 0.45f, 0.15f, 0);
    Rng g;
    rng_init(g, 1995); /* rand_init, pbrt_engine.cu:597-601 */
    for (int i = 10; i < 30; i++) {
        float r = rng_uniform(g), gg = rng_uniform(g), b = rng_uniform(g);
        set(i, O_LAMBERTIAN, r, gg, b, 0);
    }
    for (int i = 30; i < 50; i++) {
        float r = rng_uniform(g), gg = rng_uniform(g), b = rng_uniform(g);
        float f = rng_uniform(g) / 50;
        set(i, O_METAL, r, gg, b, fuzz(f));
    }
}

int o_set_texture(int idx, const unsigned char *rgba, int w, int h) {
    if (idx < 0 || idx >= O_TEX_NUM || w <= 0 || h <= 0 || !rgba) return 1;
    g_tex[idx].rgba.assign(rgba, rgba + (size_t)4 * w * h);
    g_tex[idx].w = w;
    g_tex[idx].h = h;
    return 0;
}
void o_tex_fetch(int idx, float u, float v, float out[3]) {
    V c = tex_fetch(idx, u, v);
    st(out, c);
}

int o_build_bvh(const o_tri148 *tris, int n, o_node32 *out) { return o_build_bvh_after(tris, n, 0, out); }

int o_build_bvh_after(const o_tri148 *tris, int n, long long rand_calls_before, o_node32 *out) {
    /* __genBvhNodes + make_bvh_nodes(vector version), pbrt_engine.cu:181-197,335-368.
     * rand() in its default state == srand(1) (triangle.h:12-15 never seeds it). */
    struct Item {
        const o_tri148 *p;
        int idx;
    };
    if (n <= 0) return 0;
    std::vector<Item> items;
    items.reserve(n);
    for (int i = 0; i < n; i++) items.push_back(Item{&tris[i], i});
    std::vector<o_node32> nodes;
    nodes.reserve(2 * (size_t)n);
    srand(1);
    /* the caller's process had already drawn this many numbers (scene setup, main.cpp:449-474; earlier builds) */
    for (long long i = 0; i < rand_calls_before; i++) (void)rand();
    auto leaf = [&](const Item &it) {
        Box b = tri_box(*it.p);
        o_node32 nd;
        st(nd.mn, b.mn);
        st(nd.mx, b.mx);
        nd.left = it.idx;
        nd.right = it.idx;
        return nd;
    };
    auto parent = [&](int l, int r) {
        Box a{ld(nodes[l].mn), ld(nodes[l].mx)}, b{ld(nodes[r].mn), ld(nodes[r].mx)};
        Box u = union_box(a, b);
        o_node32 nd;
        st(nd.mn, u.mn);
        st(nd.mx, u.mx);
        nd.left = l;
        nd.right = r;
        return nd;
    };
    struct Rec2 {
        static void go(std::vector<Item> &it, int m, int nn, std::vector<o_node32> &nodes,
                       const std::function<o_node32(const Item &)> &leaf,
                       const std::function<o_node32(int, int)> &parent) {
            float dr = rand() / (float)RAND_MAX; /* drand48 wrapper, triangle.h:12-15 */
            int axis = int(3 * dr);
            /* X_CMP / Y_CMP / Z_CMP, triangle.h:394-434: padded box min, recomputed per comparison */
            auto key = [axis](const Item &a) {
                Box b = tri_box(*a.p);
                return axis == 0 ? b.mn.x : (axis == 1 ? b.mn.y : b.mn.z);
            };
            std::sort(it.begin() + m, it.begin() + nn, [&](Item a, Item b) { return key(a) < key(b); });
            int cnt = nn - m;
            if (cnt == 1) {
                nodes.push_back(leaf(it[m]));
            } else if (cnt == 2) {
                int i1 = (int)nodes.size();
                int i2 = i1 + 1;
                nodes.push_back(leaf(it[m]));
                nodes.push_back(leaf(it[m + 1]));
                nodes.push_back(parent(i1, i2));
            } else if (cnt != 0) {
                go(it, m, m + cnt / 2, nodes, leaf, parent);
                int i1 = (int)nodes.size() - 1;
                go(it, m + cnt / 2, nn, nodes, leaf, parent);
                int i2 = (int)nodes.size() - 1;
                nodes.push_back(parent(i1, i2));
            }
        }
    };
    std::function<o_node32(const Item &)> lf = leaf;
    std::function<o_node32(int, int)> pf = parent;
    Rec2::go(items, 0, n, nodes, lf, pf);
    std::memcpy(out, nodes.data(), nodes.size() * sizeof(o_node32));
    return (int)nodes.size();
}

void o_camera_update(o_camera *cam, const float lookfrom[3], const float lookat[3], const float vup[3], float vfov,
                     float aspect, float aperture, float focus_dist, float t0, float t1) {
    camera_update(*cam, ld(lookfrom), ld(lookat), ld(vup), vfov, aspect, aperture, focus_dist, t0, t1);
}

void o_dfs_rank(const o_node32 *nodes, int n_nodes, int n_tris, int32_t *rank) { dfs_rank(nodes, n_nodes, n_tris, rank); }

int o_hit(const o_tri148 *tris, const o_node32 *nodes, int n_nodes, const float org[3], const float dir[3],
          int32_t *out_id, float *out_t, o_counts *cnt) {
    Scene sc{tris, 0, nodes, n_nodes, {}};
    o_counts local{0, 0, 0};
    Rec rec;
    Ray r{ld(org), ld(dir), 0.0f};
    bool h = hit_reference(sc, r, 0.001f, FLT_MAX, rec, cnt ? *cnt : local);
    *out_id = h ? rec.tri : -1;
    *out_t = h ? rec.t : 0.0f;
    return h ? 1 : 0;
}

int o_hit_linear(const o_tri148 *tris, int n, const int32_t *rank, const float org[3], const float dir[3],
                 int32_t *out_id, float *out_t) {
    Ray r{ld(org), ld(dir), 0.0f};
    bool found = false;
    float best_t = 0;
    int best_rank = 0, best = -1;
    Rec tmp;
    for (int i = 0; i < n; i++) {
        if (prim_hit(tris[i], i, r, 0.001f, FLT_MAX, tmp)) {
            int rk = rank ? rank[i] : i;
            if (!found || tmp.t < best_t || (tmp.t == best_t && rk < best_rank)) {
                found = true;
                best_t = tmp.t;
                best_rank = rk;
                best = i;
            }
        }
    }
    *out_id = best;
    *out_t = found ? best_t : 0.0f;
    return found ? 1 : 0;
}

int o_hit_ordered(const o_tri148 *tris, const o_node32 *nodes, int n_nodes, const float org[3], const float dir[3],
                  int32_t *out_id, float *out_t, o_counts *cnt) {
    int n_tris = (n_nodes + 1) / 2;
    Scene sc{tris, n_tris, nodes, n_nodes, {}};
    sc.rank.resize(n_tris);
    dfs_rank(nodes, n_nodes, n_tris, sc.rank.data());
    o_counts local{0, 0, 0};
    Rec rec;
    Ray r{ld(org), ld(dir), 0.0f};
    bool h = hit_ordered(sc, r, 0.001f, FLT_MAX, rec, cnt ? *cnt : local);
    *out_id = h ? rec.tri : -1;
    *out_t = h ? rec.t : 0.0f;
    return h ? 1 : 0;
}

void o_render(const o_tri148 *tris, int n_tris, const o_node32 *nodes, int n_nodes, const o_mat *mats,
              const o_camera *cam, int nx, int ny, int ns, int max_depth, int x0, int y0, int sx, int sy, int mode,
              int threads, float *fb, o_counts *cnt) {
    Scene sc{tris, n_tris, nodes, n_nodes, {}};
    if (mode != 0) {
        sc.rank.resize(n_tris);
        dfs_rank(nodes, n_nodes, n_tris, sc.rank.data());
    }
    if (threads < 1) threads = 1;
    std::vector<o_counts> per(threads, o_counts{0, 0, 0});
    for_rows(ny, y0, sy, threads, [&](int j, int tid) {
        o_counts &c = per[tid];
        for (int i = x0; i < nx; i += sx) {
            int pixel_index = j * nx + i;
            Rng g;
            rng_init(g, (uint64_t)(1995 + pixel_index)); /* pbrt_engine.cu:580 */
            V col = mk(0, 0, 0);
            for (int k = 0; k < ns; k++) {
                Ray r = primary_ray(*cam, i, j, nx, ny, g);
                V s = path_color(sc, mats, max_depth, mode, r, g, c);
                col = add(col, s);
            }
            { /* col /= float(ns): multiply by float(1.0/t), vec3.h:158-165 */
                float k = (float)(1.0 / (double)(float)ns);
                col = mk(col.x * k, col.y * k, col.z * k);
            }
            clip_color(col); /* :591-593 */
            st(fb + 3 * (size_t)pixel_index, col);
        }
    });
    if (cnt) {
        cnt->rays = cnt->box_tests = cnt->tri_tests = 0;
        for (auto &c : per) {
            cnt->rays += c.rays;
            cnt->box_tests += c.box_tests;
            cnt->tri_tests += c.tri_tests;
        }
    }
}

void o_trace_primary(const o_tri148 *tris, int n_tris, const o_node32 *nodes, int n_nodes, const o_camera *cam, int nx,
                     int ny, int x0, int y0, int sx, int sy, int mode, int threads, int32_t *ids, float *ts,
                     o_counts *cnt) {
    Scene sc{tris, n_tris, nodes, n_nodes, {}};
    if (mode != 0) {
        sc.rank.resize(n_tris);
        dfs_rank(nodes, n_nodes, n_tris, sc.rank.data());
    }
    if (threads < 1) threads = 1;
    std::vector<o_counts> per(threads, o_counts{0, 0, 0});
    for_rows(ny, y0, sy, threads, [&](int j, int tid) {
        o_counts &c = per[tid];
        for (int i = x0; i < nx; i += sx) {
            int pixel_index = j * nx + i;
            Rng g;
            rng_init(g, (uint64_t)(1995 + pixel_index));
            Ray r = primary_ray(*cam, i, j, nx, ny, g);
            Rec rec;
            bool h = scene_hit(sc, mode, r, 0.001f, FLT_MAX, rec, c);
            ids[pixel_index] = h ? rec.tri : -1;
            ts[pixel_index] = h ? rec.t : 0.0f;
        }
    });
    if (cnt) {
        cnt->rays = cnt->box_tests = cnt->tri_tests = 0;
        for (auto &c : per) {
            cnt->rays += c.rays;
            cnt->box_tests += c.box_tests;
            cnt->tri_tests += c.tri_tests;
        }
    }
}

uint64_t o_set_ray_log(float *buf, uint64_t cap) {
    const uint64_t n = g_ray_log_n;
    g_ray_log = buf;
    g_ray_log_cap = buf ? cap : 0;
    g_ray_log_n = 0;
    return n;
}

int o_write_ppm(const char *path, const float *fb, int nx, int ny) { /* pbrt_engine.cu:123-133 */
    std::ofstream ofs(path);
    if (!ofs) return 1;
    ofs << "P3\n" << nx << " " << ny << "\n255\n";
    for (int j = ny - 1; j >= 0; j--) {
        for (int i = 0; i < nx; i++) {
            size_t p = 3 * ((size_t)j * nx + i);
            int ir = int(255.99 * fb[p + 0]);
            int ig = int(255.99 * fb[p + 1]);
            int ib = int(255.99 * fb[p + 2]);
            ofs << ir << " " << ig << " " << ib << "\n";
        }
    }
    return ofs.good() ? 0 : 1;
}

void o_libc_rand_seq(unsigned seed, int n, int32_t *out) {
    srand(seed);
    for (int i = 0; i < n; i++) out[i] = rand();
}
void o_std_sort_by_key(const float *key, int32_t *idx, int n) {
    std::sort(idx, idx + n, [key](int32_t a, int32_t b) { return key[a] < key[b]; });
}

} /* extern "C" */
