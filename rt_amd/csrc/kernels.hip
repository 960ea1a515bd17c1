// kernels.hip -- hand-written HIP for gfx950 (MI355X): the per-pixel trace loop.
//
// What is parity-critical and what is not:
//   * Primitive intersection (ray/triangle, ray/sphere), camera rays, RNG, material scatter and
//     sample accumulation follow the reference's float/double operation order exactly
//     (file:line cited per function) and are compiled WITHOUT fused multiply-add
//     (-ffp-contract=off on the command line and the pragma below).  Results must be bit-identical
//     to the CPU oracle.
//   * Box tests are NOT parity-critical: any conservative culling returns the same closest hit
//     (min t, ties to the lower DFS-leaf rank == lower index in the re-ordered primitive array).
//     They use explicit v_fma_f32 (__builtin_fmaf), approximate reciprocals and re-fitted boxes.
//
// Execution shape: one 64-lane wavefront per workgroup = one 8x8 pixel tile; per-lane traversal
// stack in LDS at [entry][lane] (bank-conflict-free: consecutive lanes, consecutive banks); block ->
// tile mapping is XCD-aware (blocks b and b+8 share an XCD's L2, so each XCD gets one contiguous
// band of tiles).
#include <hip/hip_runtime.h>

#include <algorithm>

#include <cfloat>
#include <cstdint>
#include <mutex>

#include "kernels.h"
#include "rt_shared_math.h"
#include "rt_types.h"

#pragma clang fp contract(off)

namespace rtamd {

// ------------------------------------------------------------------ vec3 (reference include/RT/vec3.h)
struct V3 {
    float x, y, z;
};
__device__ __forceinline__ V3 mk(float a, float b, float c) { return V3{a, b, c}; }
__device__ __forceinline__ V3 ldv(const float *p) { return V3{p[0], p[1], p[2]}; }
__device__ __forceinline__ V3 vadd(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 vsub(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 vmul(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 vscl(float t, V3 v) { return mk(t * v.x, t * v.y, t * v.z); }
__device__ __forceinline__ V3 vdiv(V3 v, float t) { return mk(v.x / t, v.y / t, v.z / t); } // three divides, vec3.h:101
__device__ __forceinline__ V3 vneg(V3 v) { return mk(-v.x, -v.y, -v.z); }
__device__ __forceinline__ float vdot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 vcross(V3 a, V3 b) { // vec3.h:113-117
    return mk(a.y * b.z - a.z * b.y, -(a.x * b.z - a.z * b.x), a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float vsqlen(V3 v) { return v.x * v.x + v.y * v.y + v.z * v.z; }
__device__ __forceinline__ float vlen(V3 v) { return __builtin_sqrtf(vsqlen(v)); }
__device__ __forceinline__ V3 vunit(V3 v) { return vdiv(v, vlen(v)); }
__device__ __forceinline__ float abs_ref(float f) { return (double)f > 0.0 ? f : -f; } // cuAbs, vec3.h:58
__device__ __forceinline__ void clip_color(V3 &v) {                                    // vec3.h:52-56
    v.x = (double)v.x > 1.0 ? 1.0f : v.x;
    v.y = (double)v.y > 1.0 ? 1.0f : v.y;
    v.z = (double)v.z > 1.0 ? 1.0f : v.z;
}
#define RT_REF_PI 3.1415926 /* vec3.h:16 */

// (float)((double)c / 3.1415926), bit for bit, mostly without the division.  q' = RN(c * RN(1 / P)) lies within 3 ulp
// (double) of Q = RN(c / P): relative errors 2^-53 (reciprocal) + 2^-53 (product) against 2^-53 (quotient).  Rounded to
// float the two differ only if a float rounding midpoint lies between them, i.e. if the 29 mantissa bits of q' below
// float precision are within 3 of 0x10000000; then, and for c too small for a normal float result (or not positive,
// or NaN), the division runs.  One lane in 2^25 takes it.
__device__ __forceinline__ float div_ref_pi(float c) {
    const double q = (double)c * (1.0 / RT_REF_PI);
    const uint32_t low = (uint32_t)(unsigned long long)__double_as_longlong(q) & 0x1FFFFFFFu;
    const bool near_midpoint = (low - 0x0FFFFFF8u) <= 16u; // |low - 0x10000000| <= 8
    if (__builtin_expect(near_midpoint || !(c >= 1e-30f), 0)) return (float)((double)c / RT_REF_PI);
    return (float)q;
}

// ------------------------------------------------------------------ closest hit
struct Ray {
    V3 o, d;
};
struct Hit {
    float t;
    int rank; // index into the DFS-ordered primitive array; 0x7fffffff = miss
};
#define RT_NO_HIT 0x7fffffff

// Primitive test.  Triangle: Triangle::hit, include/RT/triangle.h:136-179 (Moeller-Trumbore, no
// determinant epsilon, two-sided).  Sphere: triangle.h:180-213 (near root only; the "second root"
// repeats the minus sign).  t interval is the reference's fixed (0.001, FLT_MAX).
struct PrimRegs {
    float4 A, B, C;
};
// PSTR: float4s per index step.  3 = `rank` counts 48-byte records (formats 0 and 1, and everything outside the
// traversal); 1 = `rank` is a unit address of the packed scene (format 2, rt_types.h), below 2^28: a 32-bit byte offset.
template <int PSTR = 3>
__device__ __forceinline__ PrimRegs prim_load(const float4 *__restrict__ prims, uint32_t rank) {
    const float4 *p = PSTR == 1 ? reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(prims) + (size_t)(uint32_t)(rank << 4))
                                : prims + 3 * (size_t)rank;
    PrimRegs R;
    R.A = p[0], R.B = p[1], R.C = p[2];
    // The test reads C.x, C.y and C.w, and hipcc narrows the third load to a dwordx2 plus a dword: FOUR vector-memory
    // instructions per primitive, each one L1 access per lane, where three fetch the record.  Naming all of C keeps
    // it one dwordx4.
    asm volatile("" ::"v"(R.C.x), "v"(R.C.y), "v"(R.C.z), "v"(R.C.w));
    return R;
}
__device__ __forceinline__ bool prim_test(const PrimRegs &R, const Ray &r, float &t_out, int &ref_rank) {
    const float4 A = R.A, B = R.B, C = R.C;
    ref_rank = __float_as_int(C.w);
    if (__float_as_int(C.y) == 0) {
        V3 v0 = mk(A.x, A.y, A.z), e1 = mk(A.w, B.x, B.y), e2 = mk(B.z, B.w, C.x);
        V3 tvec = vsub(r.o, v0);
        V3 pvec = vcross(r.d, e2);
        float det = vdot(e1, pvec);
        det = 1.0f / det;
        float u = vdot(tvec, pvec) * det;
        if (u < 0.0f || u > 1.0f) return false;
        V3 qvec = vcross(tvec, e1);
        float v = vdot(r.d, qvec) * det;
        if (v < 0.0f || (u + v) > 1.0f) return false;
        float t = vdot(e2, qvec) * det;
        if (t <= 0.001f || t >= FLT_MAX) return false;
        t_out = t;
        return true;
    } else {
        V3 c = mk(A.x, A.y, A.z);
        float radius = A.w;
        V3 oc = vsub(r.o, c);
        float a = vdot(r.d, r.d);
        float b = (float)(2.0 * (double)vdot(oc, r.d));
        float cc = vdot(oc, oc) - radius * radius;
        float disc = b * b - 4 * a * cc;
        if (disc > 0) {
            float tem = (float)((double)(-b - __builtin_sqrtf(disc)) / (2.0 * (double)a));
            if (tem < FLT_MAX && tem > 0.001f) {
                t_out = tem;
                return true;
            }
        }
        return false;
    }
}

// min t; ties go to the earlier DFS leaf of the REFERENCE tree (pbrt_engine.cu:421-438).  The tie branch is rare:
// it re-reads the incumbent's reference rank instead of carrying it in a register.
template <int PSTR = 3>
__device__ __forceinline__ bool better_hit(const float4 *__restrict__ prims, float t, int ref_rank, float best_t, int best_idx) {
    if (t < best_t) return true;
    if (t == best_t && best_idx != RT_NO_HIT) return ref_rank < __float_as_int(prims[PSTR * (size_t)best_idx + 2].w);
    return false;
}

template <int PSTR = 3>
__device__ __forceinline__ bool prim_test(const float4 *__restrict__ prims, uint32_t rank, const Ray &r, float &t_out,
                                          int &ref_rank) {
    return prim_test(prim_load<PSTR>(prims, rank), r, t_out, ref_rank);
}

__device__ __forceinline__ float safe_inv(float d) {
    float inv = __builtin_amdgcn_rcpf(d);
    if (!(__builtin_fabsf(inv) <= 1e18f)) inv = __builtin_copysignf(1e18f, d);
    return inv;
}

// Box tests for rays of ANY direction length (rt_trace_rays takes what the caller hands it; the reference does not
// normalise directions either).  The reciprocal direction is capped at 1e18 per component (safe_inv), which is harmless
// for a component that is tiny beside the others and wrong when ALL of them are tiny: the three slabs' intervals are
// then scaled by different factors and stop overlapping.  So the box tests of the ray-array kernels run on the
// direction scaled by a power of two that brings its largest component into [0.5, 1): d' = d * 2^-e, t' = t * 2^e.
// Exact (powers of two), and it touches nothing parity-critical: primitive tests take the ray as it came.
// Returns 2^e (e clamped to +-100, 0 for a zero or non-finite direction); `to_unit` = 2^-e.
__device__ __forceinline__ float dir_scale(const V3 &d, float &to_unit) {
    const float m = fmaxf(fmaxf(__builtin_fabsf(d.x), __builtin_fabsf(d.y)), __builtin_fabsf(d.z));
    int e = 0;
    if (m > 0.0f && m < __builtin_inff()) e = (int)((__float_as_uint(m) >> 23) & 0xFFu) - 126; // m = f * 2^e, f in [0.5, 1)
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    to_unit = __uint_as_float((uint32_t)(127 - e) << 23);
    return __uint_as_float((uint32_t)(127 + e) << 23);
}

__device__ __forceinline__ uint32_t sel4(uint4 c, uint32_t s) {
    uint32_t lo = (s & 1) ? c.y : c.x;
    uint32_t hi = (s & 1) ? c.w : c.z;
    return (s & 2) ? hi : lo;
}

#define RT_CSWAP(a, b)                                                                                                \
    {                                                                                                                  \
        uint32_t lo_ = min(a, b);                                                                                      \
        uint32_t hi_ = max(a, b);                                                                                      \
        a = lo_;                                                                                                       \
        b = hi_;                                                                                                       \
    }

// while-while style traversal of the 4-wide BVH with a per-lane LDS stack.
// stack: this lane's column; entry k lives at stack[k * 64].
__device__ __forceinline__ Hit closest_hit(const float4 *__restrict__ nodes, const float4 *__restrict__ prims,
                                           const Ray &r, uint32_t *stack) {
    float to_unit;
    const float ts = dir_scale(r.d, to_unit); // box distances below are t * ts (see dir_scale)
    const float tmin_s = 0.001f * ts;
    const float idx = safe_inv(r.d.x * to_unit), idy = safe_inv(r.d.y * to_unit), idz = safe_inv(r.d.z * to_unit);
    const float nox = -r.o.x * idx, noy = -r.o.y * idy, noz = -r.o.z * idz;
    // near / far plane vectors by direction sign (float4 index inside the node)
    // (by the sign of the reciprocal: -0 counts as negative, as aabb::hit's `invD < 0` has it, triangle.h:52-70)
    const int nxi = idx < 0.0f ? 3 : 0, fxi = 3 - nxi;
    const int nyi = idy < 0.0f ? 4 : 1, fyi = 5 - nyi;
    const int nzi = idz < 0.0f ? 5 : 2, fzi = 7 - nzi;

    float best_t = FLT_MAX;
    int best_rank = RT_NO_HIT;
    uint32_t cur = 0; // root node
    int sp = 0;
    for (;;) {
        if (!(cur & kLeafBit)) {
            const float4 *n = nodes + 8 * (size_t)cur;
            const float4 nx = n[nxi], ny = n[nyi], nz = n[nzi];
            const float4 fx = n[fxi], fy = n[fyi], fz = n[fzi];
            const uint4 ch = *reinterpret_cast<const uint4 *>(n + 6);
            uint32_t k0, k1, k2, k3;
#define RT_SLAB(K, C, I)                                                                                               \
    {                                                                                                                  \
        float tn = fmaxf(fmaxf(__builtin_fmaf(nx.C, idx, nox), __builtin_fmaf(ny.C, idy, noy)),                        \
                         fmaxf(__builtin_fmaf(nz.C, idz, noz), tmin_s));                                               \
        float tf = fminf(fminf(__builtin_fmaf(fx.C, idx, nox), __builtin_fmaf(fy.C, idy, noy)),                        \
                         fminf(__builtin_fmaf(fz.C, idz, noz), best_t * ts));                                          \
        K = (tn <= tf) ? ((__float_as_uint(tn) & ~3u) | I) : 0xFFFFFFFFu;                                              \
    }
            RT_SLAB(k0, x, 0u)
            RT_SLAB(k1, y, 1u)
            RT_SLAB(k2, z, 2u)
            RT_SLAB(k3, w, 3u)
#undef RT_SLAB
            RT_CSWAP(k0, k1)
            RT_CSWAP(k2, k3)
            RT_CSWAP(k0, k2)
            RT_CSWAP(k1, k3)
            RT_CSWAP(k1, k2)
            if (k0 != 0xFFFFFFFFu) {
                cur = sel4(ch, k0 & 3u);
                if (k1 != 0xFFFFFFFFu) {
                    if (k2 != 0xFFFFFFFFu) {
                        if (k3 != 0xFFFFFFFFu) {
                            stack[sp * 64] = sel4(ch, k3 & 3u);
                            sp++;
                        }
                        stack[sp * 64] = sel4(ch, k2 & 3u);
                        sp++;
                    }
                    stack[sp * 64] = sel4(ch, k1 & 3u);
                    sp++;
                }
                continue;
            }
        } else {
            const uint32_t start = cur & kLeafStartMask;
            const uint32_t cnt = ((cur >> 28) & 7u) + 1u;
            for (uint32_t i = 0; i < cnt; i++) {
                float t;
                int rr;
                if (prim_test(prims, start + i, r, t, rr)) {
                    if (better_hit(prims, t, rr, best_t, best_rank)) {
                        best_t = t;
                        best_rank = (int)(start + i);
                    }
                }
            }
        }
        if (sp == 0) break;
        sp--;
        cur = stack[sp * 64];
    }
    Hit h;
    h.t = best_t;
    h.rank = best_rank;
    return h;
}

// ------------------------------------------------------------------ camera (reference rt_camera.h:9-60)
__device__ __forceinline__ V3 disk_sample(Rng &g) { // random_in_unit_disk_d, x drawn first
    V3 p;
    do {
        float a = rng_uniform(g);
        float b = rng_uniform(g);
        p = vsub(vscl(2.0f, mk(a, b, 0.0f)), mk(1.0f, 1.0f, 0.0f));
    } while (vdot(p, p) >= 1.0f);
    return p;
}
__device__ __forceinline__ Ray camera_ray(const DevCamera &c, float s, float t, Rng &g) { // camera::get_ray
    V3 rd = vscl(c.lens_radius, disk_sample(g));
    V3 cu = ldv(c.u), cv = ldv(c.v), org = ldv(c.origin);
    V3 offset = vadd(vscl(rd.x, cu), vscl(rd.y, cv));
    (void)rng_uniform(g); // ray time, rt_camera.h:58: consumed, never read by hit()
    Ray r;
    r.o = vadd(org, offset);
    r.d = vsub(vsub(vadd(vadd(ldv(c.lower_left), vscl(s, ldv(c.horizontal))), vscl(t, ldv(c.vertical))), org), offset);
    return r;
}
__device__ __forceinline__ Ray primary_ray(const DevCamera &c, int i, int j, int nx, int ny, Rng &g) {
    float u = ((float)i + rng_uniform(g)) / (float)nx; // pbrt_engine.cu:584-585
    float v = ((float)j + rng_uniform(g)) / (float)ny;
    return camera_ray(c, u, v, g);
}

// ------------------------------------------------------------------ sampling (utilities.h:25-59, utilities.cu:18-59)
struct Onb {
    V3 u, v, w;
};
__device__ __forceinline__ Onb onb_from_w(V3 n) {
    Onb b;
    b.w = vunit(n);
    V3 a = ((double)abs_ref(b.w.x) > 0.9) ? mk(0, 1, 0) : mk(1, 0, 0);
    b.v = vunit(vcross(b.w, a));
    b.u = vcross(b.w, b.v);
    return b;
}
__device__ __forceinline__ V3 cosine_generate(const Onb &b, Rng &g) {
    float r1 = rng_uniform(g);
    float r2 = rng_uniform(g);
    float z = __builtin_sqrtf(1 - r2);
    float phi = (float)(2 * RT_REF_PI * (double)r1);
    double sn, cs;
    sincos_sh((double)phi, sn, cs);
    float sr2 = __builtin_sqrtf(r2);
    float x = (float)((double)(float)cs * 2.0 * (double)sr2); // the factor 2 is the reference's (utilities.cu:23-24)
    float y = (float)((double)(float)sn * 2.0 * (double)sr2);
    return vadd(vadd(vscl(x, b.u), vscl(y, b.v)), vscl(z, b.w));
}
__device__ __forceinline__ float cosine_value(const Onb &b, V3 dir) {
    float c = vdot(vunit(dir), b.w);
    return c > 0 ? div_ref_pi(c) : 0.0f;
}
__device__ __forceinline__ V3 ball_sample(Rng &g) { // random_in_unit_sphere, material.h:27-33
    V3 p;
    do {
        float a = rng_uniform(g);
        float b = rng_uniform(g);
        float c = rng_uniform(g);
        p = vsub(vscl(2.0f, mk(a, b, c)), mk(1, 1, 1));
    } while (vsqlen(p) >= 1.0f);
    return p;
}
__device__ __forceinline__ V3 reflect_ref(V3 v, V3 n) { return vsub(v, vscl(2.0f * vdot(v, n), n)); }
__device__ __forceinline__ bool refract_ref(V3 v, V3 n, float ni_over_nt, V3 &out) { // material.h:13-23
    V3 uv = vunit(v);
    float dt = vdot(uv, n);
    float disc = 1.0f - ni_over_nt * ni_over_nt * (1 - dt * dt);
    if (disc > 0) {
        out = vsub(vscl(ni_over_nt, vsub(uv, vscl(dt, n))), vscl(__builtin_sqrtf(disc), n));
        return true;
    }
    return false;
}
__device__ __forceinline__ float schlick_ref(float cosine, float ref_idx) { // material.h:7-11
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * pow5_sh(1.0f - cosine);
}

// tex2D<float4>(tex, u, v) of getAttenuationFromTexture (src/pbrt/utilities.cu:4-9): RGBA8 as normalised float,
// normalised coordinates, wrap addressing, linear filtering (src/cxx/texture.cpp:74-85).  CUDA's filter is hardware;
// this is the shared deterministic restatement of its published formula, written independently of
// the oracle's: weights in 8 fractional bits, float blend without FMA in a fixed order.
__device__ __forceinline__ float tex_coord(float u, int n, int &i0, int &i1) {
    if (!(__builtin_fabsf(u) <= 3.0e38f)) u = 0.0f;
    const float x = (u - __builtin_floorf(u)) * (float)n;
    const float xb = x - 0.5f;
    const float fl = __builtin_floorf(xb);
    float a = xb - fl;
    a = __builtin_floorf(a * 256.0f + 0.5f) / 256.0f;
    const int i = (int)fl;
    i0 = ((i % n) + n) % n;
    i1 = (i0 + 1) % n;
    return a;
}
__device__ __forceinline__ V3 tex_fetch(const DevTex &T, float u, float v) {
    int i0, i1, j0, j1;
    const float a = tex_coord(u, T.w, i0, i1), b = tex_coord(v, T.h, j0, j1);
    const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    const uchar4 *tx = reinterpret_cast<const uchar4 *>(T.texels);
    const uchar4 t00 = tx[(size_t)j0 * T.w + i0], t10 = tx[(size_t)j0 * T.w + i1];
    const uchar4 t01 = tx[(size_t)j1 * T.w + i0], t11 = tx[(size_t)j1 * T.w + i1];
#define RT_TEXC(C) (((w00 * ((float)t00.C / 255.0f) + w10 * ((float)t10.C / 255.0f)) + w01 * ((float)t01.C / 255.0f)) + w11 * ((float)t11.C / 255.0f))
    const V3 c = mk(RT_TEXC(x), RT_TEXC(y), RT_TEXC(z));
#undef RT_TEXC
    return c;
}

// lambertian::scatter's and diffusive_light::scatter's attenuation for a textured triangle (material.h:100-106,
// 140-146; metal and dielectric never read the texture): the hit record's (u, v) are the
// barycentric blend of the vertices' texture coordinates (setUV, triangle.h:113-118, with 1.0 - x - y in double).
// The barycentrics are recomputed here with the arithmetic of prim_test: same inputs, same bits.
__device__ __forceinline__ V3 textured_albedo(const KernelScene &S, int rank, const Ray &r, V3 alb) {
    const float4 q0 = S.uv[2 * (size_t)rank], q1 = S.uv[2 * (size_t)rank + 1];
    const int tex = __float_as_int(q1.z);
    if (tex == -1) return alb;
    const float4 *p = S.prims + 3 * (size_t)rank;
    const float4 A = p[0], B = p[1], C = p[2];
    const V3 v0 = mk(A.x, A.y, A.z), e1 = mk(A.w, B.x, B.y), e2 = mk(B.z, B.w, C.x);
    const V3 tvec = vsub(r.o, v0);
    const V3 pvec = vcross(r.d, e2);
    float det = vdot(e1, pvec);
    det = 1.0f / det;
    const float x = vdot(tvec, pvec) * det;
    const V3 qvec = vcross(tvec, e1);
    const float y = vdot(r.d, qvec) * det;
    const float z = (float)(1.0 - (double)x - (double)y);
    const float u = z * q0.x + x * q0.y + y * q0.z;
    const float v = z * q0.w + x * q1.x + y * q1.y;
    return tex_fetch(S.tex[tex], u, v);
}

// One bounce of color() (pbrt_engine.cu:467-560) after hit() has returned.
// Returns true if the path continues with `cur` replaced; false if the sample is finished, in which
// case `thr` is the sample's value BEFORE the per-sample clamp, and `clamp` says whether to apply it
// (the pdf < 1e-6 exit at :519-521 returns black unclamped).
// TEX: compiled with the texture path (taken when the scene has texture coordinates).  The product kernel keeps an
// instance without it: inlined, the path costs 40 spilled registers at 80 VGPRs even when no triangle is textured.
template <bool TEX>
__device__ __forceinline__ bool shade_bounce(const KernelScene &S, const Hit &h, Ray &cur, V3 &thr, Rng &g, bool &clamp) {
    clamp = true;
    if (h.rank == RT_NO_HIT) { // sky, :554-558
        V3 ud = vunit(cur.d);
        float t = 0.5f * (ud.y + 1.0f);
        V3 c = vadd(vscl(1.0f - t, mk(1.0f, 1.0f, 1.0f)), vscl(t, mk(0.5f, 0.7f, 1.0f)));
        thr = vmul(thr, c);
        return false;
    }
    // hit record: p, face-forwarded normal (triangle.h:163-173) or sphere normal (:193)
    const float4 pc = S.prims[3 * (size_t)h.rank + 2];
    const float4 sh = S.shade[h.rank];
    const int shape = __float_as_int(pc.y);
    const int mat_id = __float_as_int(sh.w);
    V3 p = vadd(cur.o, vscl(h.t, cur.d));
    V3 n;
    if (shape == 0) {
        V3 gn = mk(sh.x, sh.y, sh.z);
        n = (vdot(cur.d, gn) < 0.0f) ? gn : vneg(gn);
    } else {
        const float4 pa = S.prims[3 * (size_t)h.rank];
        n = vdiv(vsub(p, mk(pa.x, pa.y, pa.z)), pa.w);
    }
    const DevMat m = S.mats[mat_id];
    const V3 alb = mk(m.albedo[0], m.albedo[1], m.albedo[2]);
    if (m.kind == RT_MAT_LAMBERTIAN) { // material.h:95-116 + pbrt_engine.cu:500-525
        V3 atten = alb;
        if (TEX && S.uv != nullptr && shape == 0) atten = textured_albedo(S, h.rank, cur, alb);
        Onb b = onb_from_w(n);
        (void)rng_uniform(g); // :505
        V3 dir = cosine_generate(b, g);
        // pdf = 1*p1.value + 0*p2.value (:518).  p2 = tri[0].pdf_value is finite for a triangle tri[0]
        // (rt_set_scene enforces that), so it adds exactly +0; a NaN direction poisons p1 the same way.
        float pdf = cosine_value(b, dir);
        if ((double)pdf < 1e-6) {
            thr = mk(0, 0, 0);
            clamp = false;
            return false;
        }
        float cosine = vdot(n, vunit(dir)); // lambertian::scatter_pdf
        if ((double)cosine < 0.0) cosine = 0.0f;
        float spdf = div_ref_pi(cosine);
        thr = vmul(thr, vdiv(vscl(spdf, atten), pdf));
        cur.o = p;
        cur.d = dir;
        return true;
    } else if (m.kind == RT_MAT_METAL) { // material.h:163-171
        V3 reflected = reflect_ref(vunit(cur.d), n);
        V3 fz = ball_sample(g);
        V3 dir = vadd(reflected, vscl(m.param, fz));
        thr = vmul(thr, alb);
        if (!(vdot(dir, n) > 0.0f)) return false;
        cur.o = p;
        cur.d = dir;
        return true;
    } else if (m.kind == RT_MAT_DIELECTRIC) { // material.h:180-215
        float ref_idx = m.param;
        V3 outward;
        V3 reflected = reflect_ref(cur.d, n);
        float ni_over_nt;
        V3 refracted = mk(0, 0, 0);
        float reflect_prob;
        float cosine;
        float dn = vdot(cur.d, n);
        if (dn > 0.0f) {
            outward = vneg(n);
            ni_over_nt = ref_idx;
            cosine = dn / vlen(cur.d);
            cosine = __builtin_sqrtf(1.0f - ref_idx * ref_idx * (1 - cosine * cosine));
        } else {
            outward = n;
            ni_over_nt = 1.0f / ref_idx;
            cosine = -dn / vlen(cur.d);
        }
        if (refract_ref(cur.d, outward, ni_over_nt, refracted))
            reflect_prob = schlick_ref(cosine, ref_idx);
        else
            reflect_prob = 1.0f;
        V3 dir = (rng_uniform(g) < reflect_prob) ? reflected : refracted;
        thr = vmul(thr, mk(1.0f, 1.0f, 1.0f));
        cur.o = p;
        cur.d = dir;
        return true;
    } else { // diffusive_light, material.h:137-148 -> pbrt_engine.cu:544-545; a textured light triangle emits its texture
        V3 atten = alb;
        if (TEX && S.uv != nullptr && shape == 0) atten = textured_albedo(S, h.rank, cur, alb);
        thr = vmul(thr, atten);
        return false;
    }
}

// ------------------------------------------------------------------ block -> tile mapping
// Blocks b and b+8 run on the same XCD (round-robin dispatch); give each XCD one contiguous run of
// this context's tiles so that its 4 MiB L2 sees one band of the image.  Bijective for any count.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t n) {
    uint32_t q = n >> 3, r = n & 7u;
    uint32_t xcd = b & 7u;
    uint32_t base = xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    return base + (b >> 3);
}

// Tile numbering: row-major (g = ty * tiles_x + tx); the eight XCD bands are contiguous ranges of g: horizontal strips.
// (Column-major numbering -- vertical strips, the same mix of sky and ground in every band -- was measured: primary
// rays 0.49 instead of 0.54 ms, the 64 spp frame 46.6 instead of 44.7 ms; with stealing spread over all bands
// 0.50 instead of 0.55 ms and 42.2 instead of 41.7 ms.)
__device__ __forceinline__ void tile_xy(const KernelFrame &F, uint32_t g, int &tx, int &ty) {
    uint32_t q = g;
    if (F.div_tiles_x.shift != ~0u) {
        const uint32_t t = __umulhi(F.div_tiles_x.mul, g);
        q = (t + ((g - t) >> 1)) >> F.div_tiles_x.shift;
    }
    ty = (int)q;
    tx = (int)(g - q * (uint32_t)F.tiles_x);
}

__device__ __forceinline__ bool tile_pixel(const KernelFrame &F, int &i, int &j) {
    uint32_t local = xcd_remap(blockIdx.x, F.n_local_tiles);
    uint32_t g = local * (uint32_t)F.shard_world + (uint32_t)F.shard_rank;
    int tx, ty;
    tile_xy(F, g, tx, ty);
    int lane = threadIdx.x;
    i = tx * 8 + (lane & 7);
    j = ty * 8 + (lane >> 3);
    return i < F.nx && j < F.ny;
}

__device__ __forceinline__ void add_rays(unsigned long long *counter, uint32_t n) {
    // wave-level sum, then one atomic per wave
    for (int off = 32; off > 0; off >>= 1) n += __shfl_xor(n, off);
    if (threadIdx.x == 0 && n) atomicAdd(counter, (unsigned long long)n);
}

extern __shared__ uint32_t s_stack[];

// ------------------------------------------------------------------ kernels
// Primary ray of sample 0 of every pixel: parity hook (hit ids and t).
__global__ void __launch_bounds__(64) k_trace_primary(KernelScene S, KernelFrame F, int32_t *ids, float *ts) {
    int i, j;
    bool in = tile_pixel(F, i, j);
    uint32_t nr = 0;
    if (in) {
        int pixel_index = j * F.nx + i;
        Rng g;
        rng_init(g, (uint64_t)(1995 + pixel_index) + F.seed_salt); // pbrt_engine.cu:580
        Ray r = primary_ray(F.cam, i, j, F.nx, F.ny, g);
        Hit h = closest_hit(S.nodes, S.prims, r, s_stack + threadIdx.x);
        nr = 1;
        if (h.rank != RT_NO_HIT) {
            ids[pixel_index] = __float_as_int(S.prims[3 * (size_t)h.rank + 2].z);
            ts[pixel_index] = h.t;
        } else {
            ids[pixel_index] = -1;
            ts[pixel_index] = 0.0f;
        }
    }
    add_rays(F.ray_counter, nr);
}

// Closest hit for caller-supplied rays.
__global__ void __launch_bounds__(64) k_trace_rays(KernelScene S, const float *org_dir6, int n, int32_t *ids, float *ts,
                                                   unsigned long long *counter) {
    int k = blockIdx.x * 64 + threadIdx.x;
    uint32_t nr = 0;
    if (k < n) {
        Ray r;
        r.o = mk(org_dir6[6 * (size_t)k + 0], org_dir6[6 * (size_t)k + 1], org_dir6[6 * (size_t)k + 2]);
        r.d = mk(org_dir6[6 * (size_t)k + 3], org_dir6[6 * (size_t)k + 4], org_dir6[6 * (size_t)k + 5]);
        Hit h = closest_hit(S.nodes, S.prims, r, s_stack + threadIdx.x);
        nr = 1;
        if (h.rank != RT_NO_HIT) {
            ids[k] = __float_as_int(S.prims[3 * (size_t)h.rank + 2].z);
            ts[k] = h.t;
        } else {
            ids[k] = -1;
            ts[k] = 0.0f;
        }
    }
    add_rays(counter, nr);
}

// render(), pbrt_engine.cu:569-595, as a per-lane state machine: every loop trip traces ONE ray of the
// lane's current path (primary or bounce), so lanes whose paths end early start their next sample at
// once instead of idling until the longest path of the wave ends.  RNG order per pixel is the
// reference's: samples serial, draws in path order.
__global__ void __launch_bounds__(64) k_render(KernelScene S, KernelFrame F, float *fb) {
    int i, j;
    bool in = tile_pixel(F, i, j);
    uint32_t nr = 0;
    if (in) {
        const int pixel_index = j * F.nx + i;
        Rng g;
        rng_init(g, (uint64_t)(1995 + pixel_index) + F.seed_salt);
        V3 col = mk(0, 0, 0);
        int k = 0;
        int bounce = 0;
        Ray cur;
        cur.o = mk(0, 0, 0);
        cur.d = mk(0, 0, 1);
        V3 thr = mk(1, 1, 1);
        bool fresh = true;
        while (k < F.ns) {
            if (fresh) {
                cur = primary_ray(F.cam, i, j, F.nx, F.ny, g);
                thr = mk(1.0f, 1.0f, 1.0f);
                bounce = 0;
                fresh = false;
            }
            Hit h = closest_hit(S.nodes, S.prims, cur, s_stack + threadIdx.x);
            nr++;
            bool clamp;
            bool go_on = shade_bounce<true>(S, h, cur, thr, g, clamp);
            bounce++;
            if (!go_on || bounce >= F.max_depth) { // ITER_NUM exhausted: falls to RETURN_COLOR with the throughput
                if (clamp) clip_color(thr);       // per-SAMPLE clamp, :564
                col = vadd(col, thr);
                k++;
                fresh = true;
            }
        }
        if (!F.raw_sum) { // col /= float(ns): multiply by float(1.0/ns), vec3.h:158-165
            float kk = (float)(1.0 / (double)(float)F.ns);
            col = mk(col.x * kk, col.y * kk, col.z * kk);
            clip_color(col); // :591-593
        }
        fb[3 * (size_t)pixel_index + 0] = col.x;
        fb[3 * (size_t)pixel_index + 1] = col.y;
        fb[3 * (size_t)pixel_index + 2] = col.z;
    }
    add_rays(F.ray_counter, nr);
}

// ------------------------------------------------------------------ persistent kernel (variant 1)
// Persistent waves with per-lane refill (after Aila & Laine's persistent while-while tracer): the
// traversal state of every lane lives in registers + its LDS stack column ACROSS rays, so a lane whose
// ray has finished does not wait for the slowest ray of the wave: when fewer than `threshold` lanes
// are still traversing, the wave leaves the traversal loop, the finished lanes shade / start their
// next sample / fetch a new pixel, and all lanes re-enter the traversal loop together.  Pixels are
// handed out from eight band counters (one band of tiles per XCD, so each XCD's L2 sees one region of
// the image; a wave that finds its band empty steals from the next).  Every pixel carries its own RNG
// stream (seed 1995 + pixel_index), so which lane renders it does not change the result.
//
// Stack: the first `lds_entries` entries of a lane's traversal stack are in LDS ([entry][lane]), the
// rest spill to a global array ([entry][global lane]); the spill path is only taken by unusually deep
// pops, which keeps LDS per wave small and occupancy high.
// hipcc sinks a load into the branch that first uses it; for a node that turns one memory round trip per
// traversal step into two (boxes, then child references).  Naming the values in an empty asm statement pins
// every load of the node before the first use of any of them.
#define RT_PIN4(V) asm volatile("" ::"v"((V).x), "v"((V).y), "v"((V).z), "v"((V).w))

#define RT_REF_NONE 0xFFFFFFFFu /* "no node": has the leaf bit set but is never a real leaf reference */

// Deterministic culling.  A node step drops children whose box the ray enters beyond the closest hit so far.  With
// speculative traversal WHEN a lane's parked leaf is tested depends on the other lanes of its wave (the batch fires
// when enough of them hold one), so "so far" would depend on the company a ray keeps -- harmless while every
// accepted hit lies inside its own leaf's box, because then no culled box can hold a closer one.  The reference's
// Moeller-Trumbore (triangle.h:137-156, no determinant epsilon) breaks that for grazing rays: with |det| tiny the
// computed t can lie BEFORE the ray even enters the triangle's box (measured on the 1 M-triangle scene: t = 2.06286
// against a box entered at 2.06311; about one ray in 10^10).  The reference, which culls nothing, takes such a hit;
// here it is taken if its leaf is reached before a hit in between is known -- a matter of timing.  So the value
// culled against is refreshed only at a point of the lane's OWN sequence: when it parks the next leaf, by which time
// every earlier leaf has been tested.  A ray's node visits, leaf tests and result are then a function of the ray and
// the tree alone: frames are reproducible launch to launch and whatever the rounds, waves or thresholds.  (They still
// differ from the reference's on those ~10^-10 of rays whose inconsistent hit a box test culls: DESIGN.md 3.)
// (a template flag, DET, on the traversal code: RT_OPT_DETERMINISTIC 0 runs the instances without it, which cull against
// the freshest value and take 5 % less time on the bench frame -- 37.0 against 39.0 ms -- at the price of that timing
// dependence.  The non-speculative and float-node variants always cull deterministically.)
// Child references by node format (rt_types.h).  Formats 0 and 1: leaf bit 31, count - 1 in bits 30:28, first
// primitive (a record index) below.  Format 2 (packed scene): unit << 2 | leaf << 1 | (count - 1); a leaf's primitives
// are the records at units first, first + 3, ...; T.best_rank holds a UNIT address while a ray is traced and becomes a
// record index (unit / 3) when the ray is finished (hit_record_index).  RT_REF_NONE has the leaf bit of either.
template <int FMT> __device__ __forceinline__ bool ref_is_leaf(uint32_t r) { return FMT == 2 ? (r & kPackLeafBit) != 0u : (r & kLeafBit) != 0u; }
template <int FMT> __device__ __forceinline__ uint32_t leaf_first(uint32_t r) { return FMT == 2 ? r >> 2 : r & kLeafStartMask; }
template <int FMT> __device__ __forceinline__ uint32_t leaf_count(uint32_t r) { return FMT == 2 ? (r & 1u) + 1u : ((r >> 28) & 7u) + 1u; }
template <int FMT> __device__ __forceinline__ bool leaf_has_second(uint32_t r) { return FMT == 2 ? (r & 1u) != 0u : (r & 0x70000000u) != 0u; }
template <int FMT> struct PrimIndex {
    static constexpr int stride = FMT == 2 ? 1 : 3; // float4s per index step (prim_load's PSTR)
    static constexpr uint32_t next = FMT == 2 ? 3u : 1u; // index step from one primitive of a leaf to the next
};
template <int FMT> __device__ __forceinline__ int hit_record_index(int best_rank) {
    if (FMT != 2 || best_rank == RT_NO_HIT) return best_rank;
    return (int)(__umulhi((uint32_t)best_rank, 0xAAAAAAABu) >> 1); // unit / 3, exact below 2^31
}

// SCALED (ray-array kernels): box distances are t * T.ts (dir_scale above); the render kernels' rays are of ordinary
// length by construction (camera, scatter) and keep the plain constants.
#define RT_SCALED_T(T, V) (SCALED ? (V) * (T).ts : (V))
#define RT_CULL_T(T) (DET ? (T).cull_t : RT_SCALED_T(T, (T).best_t))
#define RT_TMIN(T) (SCALED ? (T).tmin : 0.001f)
#define RT_CULL_SYNC(T)                                                                                                \
    do {                                                                                                               \
        if (DET) (T).cull_t = RT_SCALED_T(T, (T).best_t);                                                              \
    } while (0)
struct Trav {
    float idx, idy, idz, nox, noy, noz;
    int nxi, nyi, nzi;
    float best_t;
    float cull_t; // what node steps cull against: the closest hit among the leaves parked BEFORE the pending one (see RT_DET_CULL)
    int best_rank;
    uint32_t cur;
    uint32_t pend; // speculative traversal: one postponed leaf
    uint32_t sp;   // LDS address of the next free stack entry of this lane: stack base + entry * 256 + lane * 4
    float ts, tmin; // SCALED only: box distances are t * ts; tmin = 0.001 * ts
};

template <bool SCALED = false>
__device__ __forceinline__ void trav_begin(Trav &T, const Ray &r, uint32_t sp0) {
    float to_unit = 1.0f;
    T.ts = SCALED ? dir_scale(r.d, to_unit) : 1.0f;
    T.tmin = 0.001f * T.ts;
    T.idx = safe_inv(SCALED ? r.d.x * to_unit : r.d.x);
    T.idy = safe_inv(SCALED ? r.d.y * to_unit : r.d.y);
    T.idz = safe_inv(SCALED ? r.d.z * to_unit : r.d.z);
    T.nox = -r.o.x * T.idx;
    T.noy = -r.o.y * T.idy;
    T.noz = -r.o.z * T.idz;
    // float nodes only (the other formats take the sign of idx / idy / idz in the node step): near-plane vector by the
    // sign of the RECIPROCAL direction, so that -0 goes with the negative components, as aabb::hit's `invD < 0` has it
    T.nxi = T.idx < 0.0f ? 3 : 0;
    T.nyi = T.idy < 0.0f ? 4 : 1;
    T.nzi = T.idz < 0.0f ? 5 : 2;
    T.best_t = FLT_MAX;
    T.cull_t = FLT_MAX;
    T.best_rank = RT_NO_HIT;
    T.cur = 0;
    T.pend = RT_REF_NONE;
    T.sp = sp0;
}

// Accesses that must name global memory (a volatile or atomic access through a generic pointer becomes flat_*, which
// waits on both counters and is not among the hand-off forms measured in MI355X_MICROARCH.md, "Valid forms").
typedef __attribute__((address_space(1))) uint32_t gu32;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Round record of one pixel slot (see "Rounds" at k_persistent): ten 8-byte granules {value, check} in five 16-byte
// pieces, kRingWords words per slot.  Every piece is written by ONE buffer_store_dwordx4 ... sc1 (write-through: it
// bypasses the writer's L1 and is dropped from its XCD's L2) and read by ONE buffer_load_dwordx4 ... sc1 (bypasses the
// reader's L1): the data IS the flag, recipe R2 of cdna_hip_programming.md, Guideline 16 -- no ordering between the
// pieces, no drain on the writer, one round trip for the reader.
// A granule's check word is tag ^ rotl(value, 13), tag = frame_id * 64 + round: it names the record's (frame, round)
// AND is bound to the value beside it.  A reader accepts a record when check ^ rotl(value, 13) == tag in all ten
// granules.  What that catches, and what is left to assume:
//   * a record that is part old, part new: the old granules yield an older tag -- polled again;
//   * a granule whose two words are not from the same store (a 4-byte tear inside an 8-byte granule, which
//     MI355X_MICROARCH.md reports as never observed on gfx950 / ROCm 7.2 "also for 16-B sc1 halves" but does not
//     promise): new check with old value yields tag ^ rotl(new ^ old), which is the tag only if the value did not
//     change (rotl is a bijection) -- DETECTED, exactly, per granule; polled again, and counted when the tag it yields
//     is one no earlier launch can have written (band_counters[8] -> rt_stats.handoff_torn);
//   * left: an aligned 4-byte word of an sc1 store is never seen half-written, and an agent-scope sc1 load eventually
//     observes an sc1 store -- the forms LLVM's gfx942 memory model itself emits for agent-scope relaxed atomics.
// One writer and one reader per record and round, in turn.  (Two check words in a sixth piece were built first: 2 %
// of the frame for the extra store, load and registers; binding the tag to the value costs ten rotates a side.)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ring_rsrc(uint32_t *state, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(state, 0, (int)bytes, 0x00020000);
}
#define RT_AUX_SC1 16 /* cache-policy bits of the raw buffer builtins on gfx950: 16 = sc1 */
__device__ __forceinline__ uint32_t ring_chk(uint32_t v, uint32_t tag) { return tag ^ __builtin_amdgcn_alignbit(v, v, 19); } // tag ^ rotl(v, 13)
// (The pixel state goes through scalars, not through a reference to the Rng struct: with the struct's address taken
// here hipcc kept the whole RNG state in scratch memory for the rest of the kernel.)
__device__ __forceinline__ void ring_publish(uint32_t *state, uint32_t bytes, uint32_t slot, uint32_t tag, uint32_t v0,
                                             uint32_t v1, uint32_t v2, uint32_t v3, uint32_t v4, uint32_t d, V3 col, int k) {
    const __amdgpu_buffer_rsrc_t rs = ring_rsrc(state, bytes);
    const uint32_t off = slot * (uint32_t)(kRingWords * 4);
    const uint32_t cx = __float_as_uint(col.x), cy = __float_as_uint(col.y), cz = __float_as_uint(col.z), kk = (uint32_t)k;
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{v0, ring_chk(v0, tag), v1, ring_chk(v1, tag)}, rs, off, 0, RT_AUX_SC1);
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{v2, ring_chk(v2, tag), v3, ring_chk(v3, tag)}, rs, off + 16u, 0, RT_AUX_SC1);
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{v4, ring_chk(v4, tag), d, ring_chk(d, tag)}, rs, off + 32u, 0, RT_AUX_SC1);
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{cx, ring_chk(cx, tag), cy, ring_chk(cy, tag)}, rs, off + 48u, 0, RT_AUX_SC1);
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{cz, ring_chk(cz, tag), kk, ring_chk(kk, tag)}, rs, off + 64u, 0, RT_AUX_SC1);
}
// one poll: five loads in flight together, one round trip.  true when every granule yields `tag`, and then w[0..9] =
// the ten values (RNG v0..v4, d; sum x, y, z; sample count).  `beyond` = some granule yielded a tag above `tag`: no
// launch before this one can have written that (frame ids only grow on a record array), so it is a torn granule.
__device__ __forceinline__ bool ring_poll(uint32_t *state, uint32_t bytes, uint32_t slot, uint32_t tag, uint32_t (&w)[10], bool &beyond) {
    const __amdgpu_buffer_rsrc_t rs = ring_rsrc(state, bytes);
    const uint32_t off = slot * (uint32_t)(kRingWords * 4);
    asm volatile("" ::: "memory"); // a poll is a fresh read every time: nothing of an earlier one may be reused
    const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, RT_AUX_SC1);
    const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16u, 0, RT_AUX_SC1);
    const u32x4 c = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 32u, 0, RT_AUX_SC1);
    const u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 48u, 0, RT_AUX_SC1);
    const u32x4 e = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 64u, 0, RT_AUX_SC1);
    asm volatile("" ::: "memory");
    w[0] = a.x, w[1] = a.z, w[2] = b.x, w[3] = b.z, w[4] = c.x, w[5] = c.z, w[6] = d.x, w[7] = d.z, w[8] = e.x, w[9] = e.z;
    // the tag each granule yields: check ^ rotl(value, 13)
    const uint32_t t0 = ring_chk(a.x, a.y), t1 = ring_chk(a.z, a.w), t2 = ring_chk(b.x, b.y), t3 = ring_chk(b.z, b.w),
                   t4 = ring_chk(c.x, c.y), t5 = ring_chk(c.z, c.w), t6 = ring_chk(d.x, d.y), t7 = ring_chk(d.z, d.w),
                   t8 = ring_chk(e.x, e.y), t9 = ring_chk(e.z, e.w);
    const uint32_t lo = min(min(min(t0, t1), min(t2, t3)), min(min(t4, t5), min(min(t6, t7), min(t8, t9))));
    const uint32_t hi = max(max(max(t0, t1), max(t2, t3)), max(max(t4, t5), max(max(t6, t7), max(t8, t9))));
    beyond = hi > tag;
    return lo == tag && hi == tag;
}

// LDS through address-space-3 pointers and byte offsets: ds_read / ds_write with the offset in one register
typedef __attribute__((address_space(3))) uint32_t lu32;
typedef __attribute__((address_space(3))) char lchar;
typedef __attribute__((address_space(3))) u32x4 lu128;
// child references of the node a lane is working on: [lane][slot], written with one ds_write_b128 and read back by
// slot.  Picking one of four registers by a per-lane index costs seven vector instructions (two tests, three selects),
// four times per node; one v_and_or_b32 and a ds_read_b32 do the same and the LDS pipe has the room.
// The same kilobyte is the first four of the seven fields of the leaf exchange ([field][slot], see leaf_fire): the two
// uses never overlap in time.
__shared__ u32x4 s_child[7 * 16];

struct StackRef {
    uint32_t *spill;    // this lane's column, stride spill_stride
    uint32_t sp0;       // Trav::sp of an empty stack (LDS address of s_stack + lane * 4)
    uint32_t lds_limit; // sp0 of lane 0 + lds_entries * 256: a Trav::sp below it names an entry in LDS
    uint32_t fast_limit; // lds_limit - 512: a stack below it stays in LDS through one node step and the leaf parking
    uint32_t child16;   // LDS address of s_child[lane]
    int lds_entries;
    uint32_t spill_stride;
};
__device__ __forceinline__ uint32_t lds_addr(const void *p) { return (uint32_t)(uintptr_t)(lchar *)p; }
__device__ __forceinline__ void stack_ref_init(StackRef &K, uint32_t lane, uint32_t *spill, int lds_entries, uint32_t total_lanes) {
    K.spill = spill + (size_t)blockIdx.x * 64u + lane;
    K.sp0 = lds_addr(s_stack) + lane * 4u;
    K.lds_limit = lds_addr(s_stack) + (uint32_t)lds_entries * 256u;
    K.fast_limit = lds_entries >= 2 ? K.lds_limit - 512u : 0u;
    K.child16 = lds_addr(s_child) + lane * 16u;
    K.lds_entries = lds_entries;
    K.spill_stride = total_lanes;
}
__device__ __forceinline__ lu32 *stk_lds(uint32_t sp) { return (lu32 *)(uintptr_t)sp; }
// The spill accesses are volatile so that hipcc cannot fold "LDS or global" into one flat_load / flat_store
// (a flat access waits on both counters and is slower than ds_read even when it lands in LDS).
__device__ __forceinline__ void stk_push(const StackRef &K, uint32_t &sp, uint32_t v) {
    if (__builtin_expect(sp < K.lds_limit, 1))
        *stk_lds(sp) = v;
    else
        *(volatile gu32 *)(K.spill + (size_t)((sp - K.lds_limit) >> 8) * K.spill_stride) = v;
    sp += 256u;
}
__device__ __forceinline__ uint32_t stk_pop(const StackRef &K, uint32_t &sp) {
    sp -= 256u;
    uint32_t v;
    if (__builtin_expect(sp < K.lds_limit, 1))
        v = *stk_lds(sp);
    else
        v = *(volatile gu32 *)(K.spill + (size_t)((sp - K.lds_limit) >> 8) * K.spill_stride);
    return v;
}

// slot of a child in its sort key: bits 3:2, so that key & 12 is the byte offset of its reference in s_child[lane]
#define RT_KEY(TN, I) ((__float_as_uint(TN) & ~15u) | ((I) << 2))
__device__ __forceinline__ uint32_t child_ref(uint32_t key, uint32_t child16) {
    uint32_t a; // hipcc turns (key & 12) | child16 into an and and an add
    asm("v_and_or_b32 %0, %1, 12, %2" : "=v"(a) : "v"(key), "v"(child16));
    return *(lu32 *)(uintptr_t)a;
}

// packed scene: reference of the child whose slot * 8 sits in the key's low five bits
__device__ __forceinline__ uint32_t pack_ref(uint32_t child_bytes, uint32_t base4, uint32_t key) {
    uint32_t b;
    asm("v_bfe_u32 %0, %1, %2, 8" : "=v"(b) : "v"(child_bytes), "v"(key));
    return base4 + b;
}

// sort the (up to four) hit children by entry distance, continue with the nearest, push the others far-first
// PREP: statement that makes the references available; REF(key): the reference of the child a sort key names
#define RT_REFS_TO_LDS                                                                                                 \
    const uint32_t lane16 = K.child16;                                                                                 \
    *(lu128 *)(uintptr_t)lane16 = u32x4{ch.x, ch.y, ch.z, ch.w};
#define RT_REF_FROM_LDS(KEY) child_ref(KEY, lane16)
#define RT_SORT_AND_PUSH(PREP, REF)                                                                                    \
    RT_CSWAP(k0, k1)                                                                                                   \
    RT_CSWAP(k2, k3)                                                                                                   \
    RT_CSWAP(k0, k2)                                                                                                   \
    RT_CSWAP(k1, k3)                                                                                                   \
    RT_CSWAP(k1, k2)                                                                                                   \
    if (k0 != 0xFFFFFFFFu) {                                                                                           \
        PREP                                                                                                           \
        T.cur = REF(k0);                                                                                               \
        if (k1 != 0xFFFFFFFFu) {                                                                                       \
            const uint32_t c1 = REF(k1), c2 = REF(k2), c3 = REF(k3);                                                   \
            if (__builtin_expect(fast, 1)) { /* all three entries would be in LDS */                                   \
                if (k2 != 0xFFFFFFFFu) {                                                                               \
                    if (k3 != 0xFFFFFFFFu) {                                                                           \
                        *stk_lds(T.sp) = c3;                                                                           \
                        T.sp += 256u;                                                                                  \
                    }                                                                                                  \
                    *stk_lds(T.sp) = c2;                                                                               \
                    T.sp += 256u;                                                                                      \
                }                                                                                                      \
                *stk_lds(T.sp) = c1;                                                                                   \
                T.sp += 256u;                                                                                          \
            } else {                                                                                                   \
                if (k2 != 0xFFFFFFFFu) {                                                                               \
                    if (k3 != 0xFFFFFFFFu) stk_push(K, T.sp, c3);                                                      \
                    stk_push(K, T.sp, c2);                                                                             \
                }                                                                                                      \
                stk_push(K, T.sp, c1);                                                                                 \
            }                                                                                                          \
        }                                                                                                              \
        return;                                                                                                        \
    }

__device__ __forceinline__ uint32_t pop_or_none(const StackRef &K, uint32_t &sp) { return sp == K.sp0 ? RT_REF_NONE : stk_pop(K, sp); }

// one wide node: tests its four child boxes, sets T.cur to the nearest hit child (pushing the others) or to the
// next stack entry (RT_REF_NONE when the stack is empty).
// FMT 0: 128-byte float nodes (WideNode).  FMT 1: 64-byte quantized nodes (QuantNode).
template <int FMT, bool DET, bool SCALED = false>
__device__ __forceinline__ void node_step(Trav &T, const KernelScene &S, const StackRef &K) {
    // every stack entry this step can touch is in LDS (entries [e - 1, e + 2] for a stack of e): the common case
    const bool fast = T.sp < K.fast_limit;
    {
        uint32_t k0, k1, k2, k3;
        if (FMT == 0) {
            const float4 *n = S.nodes + 8 * (size_t)T.cur;
            const float4 nx = n[T.nxi], ny = n[T.nyi], nz = n[T.nzi];
            const float4 fx = n[3 - T.nxi], fy = n[5 - T.nyi], fz = n[7 - T.nzi];
            uint4 ch = *reinterpret_cast<const uint4 *>(n + 6);
            RT_PIN4(ch);
#define RT_SLAB(K_, C, I)                                                                                              \
    {                                                                                                                  \
        float tn = fmaxf(fmaxf(__builtin_fmaf(nx.C, T.idx, T.nox), __builtin_fmaf(ny.C, T.idy, T.noy)),                \
                         fmaxf(__builtin_fmaf(nz.C, T.idz, T.noz), RT_TMIN(T)));                                       \
        float tf = fminf(fminf(__builtin_fmaf(fx.C, T.idx, T.nox), __builtin_fmaf(fy.C, T.idy, T.noy)),                \
                         fminf(__builtin_fmaf(fz.C, T.idz, T.noz), RT_CULL_T(T)));                                         \
        K_ = (tn <= tf) ? RT_KEY(tn, I) : 0xFFFFFFFFu;                                                                 \
    }
            RT_SLAB(k0, x, 0u)
            RT_SLAB(k1, y, 1u)
            RT_SLAB(k2, z, 2u)
            RT_SLAB(k3, w, 3u)
#undef RT_SLAB
            RT_SORT_AND_PUSH(RT_REFS_TO_LDS, RT_REF_FROM_LDS)
        } else if (FMT == 1) {
            // byte offset in 32 bits (rt_abi.cpp refuses more than 2^26 wide nodes): one shift and the scalar-base form
            // of global_load instead of a 64-bit shift and a 64-bit add
            const uint4 *n = reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(S.qnodes) + (size_t)(uint32_t)(T.cur << 6));
            uint4 h = n[0], qa = n[1], qb = n[2], ch = n[3];
            RT_PIN4(h);
            RT_PIN4(qa);
            RT_PIN4(qb);
            RT_PIN4(ch);
            // grid step per axis (a power of two) folded into the ray's reciprocal direction
            const float sx = __uint_as_float(h.w) * T.idx;
            const float sy = __uint_as_float(qb.z) * T.idy;
            const float sz = __uint_as_float(qb.w) * T.idz;
            const float ax = __builtin_fmaf(__uint_as_float(h.x), T.idx, T.nox);
            const float ay = __builtin_fmaf(__uint_as_float(h.y), T.idy, T.noy);
            const float az = __builtin_fmaf(__uint_as_float(h.z), T.idz, T.noz);
            // near / far by the sign of the reciprocal direction (-0 counts as negative, as aabb::hit's `invD < 0` does):
            // three registers fewer than carrying the plane indices of the float-node format
            const bool ngx = T.idx < 0.0f, ngy = T.idy < 0.0f, ngz = T.idz < 0.0f;
            const uint32_t nqx = ngx ? qa.y : qa.x, fqx = ngx ? qa.x : qa.y;
            const uint32_t nqy = ngy ? qa.w : qa.z, fqy = ngy ? qa.z : qa.w;
            const uint32_t nqz = ngz ? qb.y : qb.x, fqz = ngz ? qb.x : qb.y;
#define RT_QB(W, SH) ((float)(((W) >> (SH)) & 0xFFu))
            // (three v_pk_fma_f32 per child instead of six v_fma_f32 were tried twice.  First the register pairs cost
            // spills: 56.2 vs 53.2 ms.  With the child references in LDS the pairs fit -- 84 instead of 96 vector
            // instructions in this block, no spill in the loop -- and the frame still took 47.0 vs 45.25 ms: a packed
            // FMA does not issue at the rate of a plain one here.)
#define RT_SLABQ(K_, SH, I, KEYF)                                                                                         \
    {                                                                                                                  \
        float tn = fmaxf(fmaxf(__builtin_fmaf(RT_QB(nqx, SH), sx, ax), __builtin_fmaf(RT_QB(nqy, SH), sy, ay)),        \
                         fmaxf(__builtin_fmaf(RT_QB(nqz, SH), sz, az), RT_TMIN(T)));                                   \
        float tf = fminf(fminf(__builtin_fmaf(RT_QB(fqx, SH), sx, ax), __builtin_fmaf(RT_QB(fqy, SH), sy, ay)),        \
                         fminf(__builtin_fmaf(RT_QB(fqz, SH), sz, az), RT_CULL_T(T)));                                     \
        K_ = (tn <= tf) ? KEYF(tn, I) : 0xFFFFFFFFu;                                                                   \
    }
            RT_SLABQ(k0, 0, 0u, RT_KEY)
            RT_SLABQ(k1, 8, 1u, RT_KEY)
            RT_SLABQ(k2, 16, 2u, RT_KEY)
            RT_SLABQ(k3, 24, 3u, RT_KEY)
            RT_SORT_AND_PUSH(RT_REFS_TO_LDS, RT_REF_FROM_LDS)
        } else {
            // packed scene: three loads; a reference is the byte offset of its record / 4 (below 2^30)
            const uint4 *n = reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(S.qnodes) + (size_t)(uint32_t)(T.cur << 2));
            uint4 h = n[0], qa = n[1], qb = n[2];
            RT_PIN4(h);
            RT_PIN4(qa);
            RT_PIN4(qb);
            // grid step per axis = 2^(exponent byte), folded into the ray's reciprocal direction
            const float sx = __builtin_ldexpf(T.idx, __builtin_amdgcn_sbfe((int)qb.w, 0, 8));
            const float sy = __builtin_ldexpf(T.idy, __builtin_amdgcn_sbfe((int)qb.w, 8, 8));
            const float sz = __builtin_ldexpf(T.idz, __builtin_amdgcn_sbfe((int)qb.w, 16, 8));
            const float ax = __builtin_fmaf(__uint_as_float(h.x), T.idx, T.nox);
            const float ay = __builtin_fmaf(__uint_as_float(h.y), T.idy, T.noy);
            const float az = __builtin_fmaf(__uint_as_float(h.z), T.idz, T.noz);
            // near / far by the sign of the reciprocal direction (-0 counts as negative, as aabb::hit's `invD < 0` does):
            // three registers fewer than carrying the plane indices of the float-node format
            const bool ngx = T.idx < 0.0f, ngy = T.idy < 0.0f, ngz = T.idz < 0.0f;
            const uint32_t nqx = ngx ? qa.y : qa.x, fqx = ngx ? qa.x : qa.y;
            const uint32_t nqy = ngy ? qa.w : qa.z, fqy = ngy ? qa.z : qa.w;
            const uint32_t nqz = ngz ? qb.y : qb.x, fqz = ngz ? qb.x : qb.y;
            // the sort key carries 8 * slot in its low five bits: v_bfe_u32 takes the child's byte by it (the
            // instruction reads bits 4:0 of the offset operand), one add makes the reference
#define RT_KEY_PACKED(TN, I) ((__float_as_uint(TN) & ~31u) | ((I) << 3))
            RT_SLABQ(k0, 0, 0u, RT_KEY_PACKED)
            RT_SLABQ(k1, 8, 1u, RT_KEY_PACKED)
            RT_SLABQ(k2, 16, 2u, RT_KEY_PACKED)
            RT_SLABQ(k3, 24, 3u, RT_KEY_PACKED)
#undef RT_KEY_PACKED
#define RT_REF_PACKED(KEY) pack_ref(h.w, qb.z, KEY)
            RT_SORT_AND_PUSH(, RT_REF_PACKED)
#undef RT_REF_PACKED
        }
#undef RT_SLABQ
#undef RT_QB
    }
    if (__builtin_expect(fast, 1)) {
        if (T.sp == K.sp0) {
            T.cur = RT_REF_NONE;
        } else {
            T.sp -= 256u;
            T.cur = *stk_lds(T.sp);
        }
    } else {
        T.cur = pop_or_none(K, T.sp);
    }
}

// one leaf: tests its primitives; min t, ties to the earlier DFS leaf (pbrt_engine.cu:421-438)
template <int FMT>
__device__ __forceinline__ void leaf_test(Trav &T, const KernelScene &S, const Ray &r, uint32_t ref) {
    constexpr int PS = PrimIndex<FMT>::stride;
    const uint32_t start = leaf_first<FMT>(ref);
    const uint32_t end = start + leaf_count<FMT>(ref) * PrimIndex<FMT>::next;
    // (loading two primitives per trip before testing the first hides the second round trip of a two-primitive leaf,
    // and costs 26 more spilled registers in the refill phase: 48.4 vs 48.2 ms, dropped)
    for (uint32_t p = start; p != end; p += PrimIndex<FMT>::next) {
        float t;
        int rr;
        if (prim_test<PS>(S.prims, p, r, t, rr)) {
            if (better_hit<PS>(S.prims, t, rr, T.best_t, T.best_rank)) {
                T.best_t = t;
                T.best_rank = (int)p;
            }
        }
    }
}

// __ballot() takes an int and compares it with zero in a vector register (v_cndmask + v_cmp per call); the builtin
// takes the predicate where it already is, in a scalar mask
__device__ __forceinline__ unsigned long long ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

__device__ __forceinline__ uint32_t mbcnt64(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// Speculative traversal's leaf step for the whole wave.  A leaf holds two primitives as a rule and the step runs with
// about 20 of 64 lanes, so the second primitive of a lane's leaf is handed to a lane that has no leaf to test: the
// owner writes its ray and the primitive's index to LDS slot `rank among owners`, the helper with the same rank
// among the free lanes reads it, every lane runs ONE primitive test (owners their first primitive, helpers the
// borrowed one), helpers write (t, reference rank) back and owners merge it after their own result -- the order the
// one-lane loop had, so the result is the same.  The primitive test is ~75 vector instructions; the exchange is
// LDS traffic and ~25.  Owners without a helper, and primitives beyond the second, take the loop.
// Must be called by all 64 lanes.  xchg = LDS address of s_child (the leaf exchange's fields).
// leaf_mask = ballot(want_leaf), which the caller has.
template <int FMT>
__device__ __forceinline__ void leaf_fire(Trav &T, const KernelScene &S, const Ray &cur, bool want_leaf,
                                          unsigned long long leaf_mask, uint32_t xchg) {
    constexpr int PS = PrimIndex<FMT>::stride;
    constexpr uint32_t PN = PrimIndex<FMT>::next;
    const uint32_t ref = T.pend;
    const uint32_t start = leaf_first<FMT>(ref);
    // masks stay scalar: owners = lanes whose leaf holds a second primitive, free = lanes without a leaf
    const bool has2 = leaf_has_second<FMT>(ref), two = has2 && want_leaf;
    const unsigned long long own_m = ballot(has2) & leaf_mask, free_m = ~leaf_mask;
    const uint32_t n_own = (uint32_t)__popcll(own_m), n_free = (uint32_t)__popcll(free_m);
    const uint32_t pairs = n_own < n_free ? n_own : n_free; // wave-uniform
    const uint32_t rank = want_leaf ? mbcnt64(own_m) : mbcnt64(free_m);
    const bool exporting = two && rank < pairs, helping = !want_leaf && rank < pairs;
    const uint32_t slot = xchg + rank * 4u;
    {
        if (exporting) {
            *(lu32 *)(uintptr_t)(slot + 0u * 256u) = __float_as_uint(cur.o.x);
            *(lu32 *)(uintptr_t)(slot + 1u * 256u) = __float_as_uint(cur.o.y);
            *(lu32 *)(uintptr_t)(slot + 2u * 256u) = __float_as_uint(cur.o.z);
            *(lu32 *)(uintptr_t)(slot + 3u * 256u) = __float_as_uint(cur.d.x);
            *(lu32 *)(uintptr_t)(slot + 4u * 256u) = __float_as_uint(cur.d.y);
            *(lu32 *)(uintptr_t)(slot + 5u * 256u) = __float_as_uint(cur.d.z);
            *(lu32 *)(uintptr_t)(slot + 6u * 256u) = start + PN;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    Ray lr = cur;
    uint32_t lp = start;
    {
        if (helping) {
            lr.o.x = __uint_as_float(*(lu32 *)(uintptr_t)(slot + 0u * 256u));
            lr.o.y = __uint_as_float(*(lu32 *)(uintptr_t)(slot + 1u * 256u));
            lr.o.z = __uint_as_float(*(lu32 *)(uintptr_t)(slot + 2u * 256u));
            lr.d.x = __uint_as_float(*(lu32 *)(uintptr_t)(slot + 3u * 256u));
            lr.d.y = __uint_as_float(*(lu32 *)(uintptr_t)(slot + 4u * 256u));
            lr.d.z = __uint_as_float(*(lu32 *)(uintptr_t)(slot + 5u * 256u));
            lp = *(lu32 *)(uintptr_t)(slot + 6u * 256u);
        }
    }
    float t = FLT_MAX;
    int rr = 0;
    bool hit = false;
    if (want_leaf || helping) hit = prim_test<PS>(S.prims, lp, lr, t, rr);
    if (pairs != 0u) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (helping) { // a hit has t < FLT_MAX (prim_test): FLT_MAX says "missed"
            *(lu32 *)(uintptr_t)(slot + 0u * 256u) = __float_as_uint(hit ? t : FLT_MAX);
            *(lu32 *)(uintptr_t)(slot + 1u * 256u) = (uint32_t)rr;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (want_leaf) {
        if (hit && better_hit<PS>(S.prims, t, rr, T.best_t, T.best_rank)) {
            T.best_t = t;
            T.best_rank = (int)start;
        }
        const uint32_t end = start + leaf_count<FMT>(ref) * PN;
        uint32_t p = start + PN;
        if (exporting) {
            const float t1 = __uint_as_float(*(lu32 *)(uintptr_t)(slot + 0u * 256u));
            const int rr1 = (int)*(lu32 *)(uintptr_t)(slot + 1u * 256u);
            if (t1 < FLT_MAX && better_hit<PS>(S.prims, t1, rr1, T.best_t, T.best_rank)) {
                T.best_t = t1;
                T.best_rank = (int)(start + PN);
            }
            p += PN;
        }
        for (; p != end; p += PN) {
            if (prim_test<PS>(S.prims, p, cur, t, rr)) {
                if (better_hit<PS>(S.prims, t, rr, T.best_t, T.best_rank)) {
                    T.best_t = t;
                    T.best_rank = (int)p;
                }
            }
        }
        T.pend = RT_REF_NONE;
    }
}

// one traversal step: one wide node, or one leaf.  returns false when the ray is finished.
template <int FMT>
__device__ __forceinline__ bool trav_step(Trav &T, const KernelScene &S, const Ray &r, const StackRef &K) {
    constexpr bool DET = true; // leaves are tested where they are reached: the value culled against is always current
    constexpr bool SCALED = false;
    if (!ref_is_leaf<FMT>(T.cur)) {
        node_step<FMT, DET>(T, S, K);
    } else {
        leaf_test<FMT>(T, S, r, T.cur);
        RT_CULL_SYNC(T);
        T.cur = pop_or_none(K, T.sp);
    }
    return T.cur != RT_REF_NONE;
}

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// which XCD this wave runs on (HW_REG_XCC_ID, bits 3:0): used for L2 affinity only, never for correctness
__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 7u; }

__device__ __forceinline__ uint32_t band_begin(uint32_t n_tiles, uint32_t b) { return (uint32_t)(((uint64_t)n_tiles * b) >> 3); }

// Wave-cooperative pixel fetch.  Must be called by all 64 lanes together.  The wave keeps a private run
// of slots [q.next, q.end) taken from its XCD's band with ONE atomic per CHUNK slots (a returning atomic
// costs about a microsecond: per-ray fetches would dominate short primary rays); `need` lanes take
// consecutive slots from it.  When its band is empty the wave takes from the other bands in turn, a different one
// per chunk and starting at a band of its own, so that the waves of a finished band spread over all bands that still
// have work instead of piling onto the next one; after all eight are known to be empty the remaining `need` lanes
// get false.  slot = local tile index * 64 + lane-in-tile.
// With rounds > 1 (render only) every band is walked `rounds` times, round-major: a band's counter runs over
// rounds * band_slots, and the slot's round comes back in `round` (see "Rounds" at k_persistent).
struct SlotQueue { // wave-uniform
    uint32_t next, end;   // the wave's run, in the coordinates of its band's counter (round * band_slots + offset)
    uint32_t round;       // round of q.next
    uint32_t band_slots;  // slots per round of the band the run comes from
    uint32_t base_slot;   // first slot of that band
    uint32_t empty, cursor; // empty: bit b = band b was found exhausted; cursor: the band last taken from
};
__device__ __forceinline__ bool fetch_slot(const uint32_t CHUNK, unsigned int *band_counters, uint32_t n_tiles, uint32_t home,
                                           SlotQueue &q, bool need, uint32_t &slot, uint32_t rounds, uint32_t &round) {
    const uint32_t lane = lane_id();
    // (After the last slot has been handed out a wave still holds what is left of its own run, which only its own
    // lanes can take.  Runs of 16 instead of 64 slots in the last round were measured: 41.8 vs 41.7 ms, no gain -- what
    // is left then is one piece per lane, 5 % of all iterations at 12 lanes per wave.)
    for (int guard = 0; guard < 64; guard++) {
        const unsigned long long mask = ballot(need);
        if (mask == 0ull) break;
        const uint32_t avail = q.end - q.next;
        if (avail == 0u) {
            if (q.empty == 0xFFu) break;
            uint32_t b = home;
            if (q.empty & (1u << home)) {
                b = q.cursor;
                do b = (b + 1u) & 7u;
                while (q.empty & (1u << b));
                q.cursor = b;
            }
            const uint32_t begin = band_begin(n_tiles, b), end = band_begin(n_tiles, b + 1u);
            const uint32_t band_slots = (end - begin) * 64u;
            uint32_t base = 0;
            if (lane == 0u) base = atomicAdd(&band_counters[b], CHUNK);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (base >= band_slots * rounds) { // also true for an empty band (band_slots == 0)
                q.empty |= 1u << b;
            } else {
                // the run stays in counter coordinates: it may cross from one round into the next (runs of different
                // lengths share a counter), and every slot it covers must be taken
                q.round = rounds > 1u ? base / band_slots : 0u;
                q.band_slots = band_slots;
                q.base_slot = begin * 64u;
                q.next = base;
                q.end = base + CHUNK < band_slots * rounds ? base + CHUNK : band_slots * rounds;
            }
            continue;
        }
        const uint32_t cnt = (uint32_t)__popcll(mask);
        const uint32_t before = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        if (need && before < avail) {
            uint32_t r = q.round;
            uint32_t off = q.next + before - r * q.band_slots; // a run is shorter than a round: at most one boundary
            if (off >= q.band_slots) {
                off -= q.band_slots;
                r++;
            }
            slot = q.base_slot + off;
            round = r;
            need = false;
        }
        q.next += cnt < avail ? cnt : avail;
        if (q.next - q.round * q.band_slots >= q.band_slots) q.round++;
    }
    return !need;
}

struct PersistArgs {
    unsigned int *band_counters; // 8, zeroed before the launch
    uint32_t *spill;             // [spill_entries][total lanes]
    int lds_entries;
    int threshold;               // leave the traversal loop when fewer lanes than this are traversing
    int leaf_batch;              // SPEC: run the primitive tests when this many lanes have a leaf pending
    uint32_t total_lanes;
    unsigned long long *stats;
    float *ray_dump;                     // analysis builds: where k_persistent records the rays it traces (or nullptr)
    unsigned long long *ray_dump_count;
    unsigned long long ray_dump_cap;
    unsigned long long *wave_times;      // analysis builds: [3 * wave] = start, end, all bands dry (s_memrealtime), or nullptr
    int rounds;            // 1 = off
    int round_end[64];     // sample count at which round r hands the pixel on (the last one: ns)
    uint32_t chunk;        // slots a wave takes per atomic on its band counter (64 when rounds > 1)
    uint32_t *round_state; // kRingWords words per slot: ten {value, check} granules, check = tag ^ rotl(value, 13), tag = frame_id * 64 + the round the record starts
    uint32_t round_bytes;  // size of that array (the buffer descriptor's range)
    uint32_t frame_id;
};

// Rounds.  A pixel is the unit of work the reference allows (one RNG stream per pixel, its samples in sequence),
// and one pixel of the bench frame is ~140 rays: when the pixel queue runs dry every lane is somewhere inside its
// last pixel, waves thin out lane by lane, and 15 % of all lane-iterations of the frame belong to lanes that have
// nothing left to do (tools/trav_stats.py).  The pixels started last set the length of that tail, so no pixel may
// start late: the frame is walked in R rounds.  In round r a lane takes a slot exactly as before (same bands, same
// chunks, same XCD affinity), runs samples [r*ns/R, (r+1)*ns/R) of that pixel, writes the pixel's state (RNG, sum,
// sample count: 10 words) to the slot's record and goes on to the next slot; the lane that takes the slot in round
// r+1 -- any lane of any wave -- picks the state up.  All pixels advance together and the frame ends on units of
// ns/R samples.  A record has one writer and one reader per round, in turn, so one record per slot is enough; its
// tags name the round it holds (frame_id * 64 + round).  A lane whose record is not there yet keeps the slot and
// polls; the writer never waits for anything, so this cannot deadlock.  Who runs which samples cannot change a value.
// Hand-off between lanes of different CUs / XCDs: self-tagged, self-checked 8-byte granules, written and read sc1 (ring_publish above).
// The writer does not wait for its stores and the reader needs one round trip, where a separate flag word needed a
// drain (s_waitcnt vmcnt(0), which also waits for every load the wave has in flight) before the flag store and a
// second, dependent round trip after the poll.

#ifdef RT_TRAV_STATS
#define RT_STAT(I, V) st[I] += (unsigned long long)(V)
#else
#define RT_STAT(I, V) ((void)0)
#endif

// MODE 0: render() (pbrt_engine.cu:569-595).  MODE 1: primary ray of sample 0 per pixel -> hit id, t.
// SPEC 1: speculative traversal (Aila & Laine): a lane that reaches a leaf postpones it and keeps
// traversing nodes; the expensive primitive test runs only when `leaf_batch` lanes have one pending (or no
// lane has node work), so it executes with many lanes instead of the one or two that happen to sit on a leaf.
// OCC: waves per SIMD the register allocation is held to (__launch_bounds__' second argument).
// (7 waves per SIMD, 72 VGPRs, 28 waves per CU on a 12-entry LDS stack: 34 spilled registers, none in the node step,
// and 45.8 ms against 45.7 for 6 -- the seventh wave buys what its spills cost.)
struct KArgs { // k_persistent's arguments as they lie in the kernel-argument segment
    KernelScene S;
    KernelFrame F;
    PersistArgs P;
    float *fb;
    int32_t *ids;
    float *ts;
};
typedef __attribute__((address_space(4))) const KArgs CKArgs;

template <int MODE, int FMT, int SPEC, int OCC, bool TEX, bool DET>
__global__ void __launch_bounds__(64, OCC) k_persistent(KernelScene S, KernelFrame F, PersistArgs P, float *fb, int32_t *ids,
                                                   float *ts) {
    constexpr bool SCALED = false; // camera rays and scattered rays are of ordinary length by construction
    const uint32_t lane = threadIdx.x;
    StackRef K;
    stack_ref_init(K, lane, P.spill, P.lds_entries, P.total_lanes);
    const uint32_t home = xcc_id();
    SlotQueue q;
    q.next = q.end = q.empty = q.round = q.base_slot = 0u;
    q.band_slots = 1u;
    q.cursor = (home + (blockIdx.x >> 3)) & 7u;

    // per-lane path state
    bool has_pixel = false, retired = false, tracing = false, result_ready = false, need_primary = false;
    uint32_t pij = 0; // pixel in hand: j << 16 | i (rt_resize keeps both below 65536)
    Rng g;
    g.v0 = g.v1 = g.v2 = g.v3 = g.v4 = g.d = 0;
    V3 col = mk(0, 0, 0), thr = mk(1, 1, 1);
    int k = 0, bounce = 0;
    Ray cur;
    cur.o = mk(0, 0, 0);
    cur.d = mk(0, 0, 1);
    Trav T;
    trav_begin(T, cur, K.sp0);
    T.cur = RT_REF_NONE; // invariant: a lane that is not tracing has T.cur == T.pend == RT_REF_NONE
    uint32_t nr = 0; // rays started by this wave: wave-uniform, lives in a scalar register
    bool waiting = false;   // slot taken in a round > 0 whose record has not been published yet
    uint32_t slot_round = 0; // slot of the pixel in hand | round << 26 (slots < 2^26, rounds < 64)
    int yield_k = 0;        // sample count at which it is handed on (ns: never)
    const bool ring_on = MODE == 0 && P.rounds > 1;
    const uint32_t rounds = ring_on ? (uint32_t)P.rounds : 1u;
    int since_poll = 0;
#ifdef RT_TRAV_STATS
    unsigned long long st[kTravStats] = {0};
    const unsigned long long wave_t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long wave_t_dry = 0; // when this wave first found every band handed out
#endif

    for (;;) {
        // ---------------- refill phase: lanes that are not traversing
        RT_STAT(kStatRefills, 1);
#ifdef RT_TRAV_STATS
        const unsigned long long clk0 = __builtin_readcyclecounter();
#endif
        since_poll = 0;
        // refill-only launch data comes from the kernel-argument segment afresh every time (scalar loads that are dead
        // again when the phase ends) instead of living in scalar registers across the traversal loop
        CKArgs *A = (CKArgs *)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(A));
        const KernelFrame &RF = *(const KernelFrame *)&A->F;
        const KernelScene &RS = *(const KernelScene *)&A->S;
        const PersistArgs &RP = *(const PersistArgs *)&A->P;
        float *const r_fb = A->fb;
        int32_t *const r_ids = A->ids;
        float *const r_ts = A->ts;
        const uint32_t n_tiles = (uint32_t)RF.n_local_tiles;
        bool want_push = false;
        RT_STAT(kStatShadeLanes, __popcll(ballot(has_pixel && result_ready)));
        if (has_pixel && result_ready) {
            result_ready = false;
            Hit h;
            h.t = T.best_t;
            h.rank = hit_record_index<FMT>(T.best_rank);
            const size_t pixel_index = (size_t)(pij >> 16) * (size_t)RF.nx + (size_t)(pij & 0xFFFFu);
            if (MODE == 1) {
                if (h.rank != RT_NO_HIT) {
                    r_ids[pixel_index] = __float_as_int(RS.prims[3 * (size_t)h.rank + 2].z);
                    r_ts[pixel_index] = h.t;
                } else {
                    r_ids[pixel_index] = -1;
                    r_ts[pixel_index] = 0.0f;
                }
                has_pixel = false;
            } else {
                bool clamp;
                bool go_on = shade_bounce<TEX>(RS, h, cur, thr, g, clamp);
                bounce++;
                if (!go_on || bounce >= RF.max_depth) {
                    if (clamp) clip_color(thr);
                    col = vadd(col, thr);
                    k++;
                    if (k == RF.ns) {
                        if (!RF.raw_sum) {
                            float kk = (float)(1.0 / (double)(float)RF.ns);
                            col = mk(col.x * kk, col.y * kk, col.z * kk);
                            clip_color(col);
                        }
                        r_fb[3 * pixel_index + 0] = col.x;
                        r_fb[3 * pixel_index + 1] = col.y;
                        r_fb[3 * pixel_index + 2] = col.z;
                        has_pixel = false;
                    } else if (ring_on && k == yield_k) {
                        want_push = true;
                    } else {
                        need_primary = true;
                    }
                }
            }
        }
#ifdef RT_TRAV_STATS
        const unsigned long long clk_shade = __builtin_readcyclecounter();
        RT_STAT(kStatShadeClocks, clk_shade - clk0);
#endif
        if (ring_on && want_push) { // end of this round's share: publish the state under the next round's tag
            ring_publish(RP.round_state, RP.round_bytes, slot_round & 0x3FFFFFFu, RP.frame_id * 64u + (slot_round >> 26) + 1u, g.v0, g.v1,
                         g.v2, g.v3, g.v4, g.d, col, k);
            has_pixel = false;
        }
#ifdef RT_TRAV_STATS
        const unsigned long long clk_pub = __builtin_readcyclecounter();
        RT_STAT(kStatPublishClocks, clk_pub - clk_shade);
#endif
        {
            uint32_t slot = 0;
            const bool need = !has_pixel && !retired && !waiting;
            RT_STAT(kStatFetchLanes, __popcll(ballot(need)));
            uint32_t round = 0;
            const bool got = fetch_slot(RP.chunk, RP.band_counters, n_tiles, home, q, need, slot, rounds, round);
#ifdef RT_TRAV_STATS
            if (q.empty == 0xFFu && wave_t_dry == 0) wave_t_dry = __builtin_amdgcn_s_memrealtime();
#endif
            if (need) {
                if (!got) {
                    retired = true;
                } else {
                    const uint32_t local = slot >> 6, within = slot & 63u;
                    const uint32_t gt = local * (uint32_t)RF.shard_world + (uint32_t)RF.shard_rank;
                    int tx, ty;
                    tile_xy(F, gt, tx, ty);
                    const int pi = tx * 8 + (int)(within & 7u), pj = ty * 8 + (int)(within >> 3);
                    if (pi < RF.nx && pj < RF.ny) { // ragged edge tiles: the slot is simply dropped, in every round
                        pij = ((uint32_t)pj << 16) | (uint32_t)pi;
                        slot_round = slot | (round << 26);
                        // this round runs samples up to (round + 1) * ns / rounds; the last round ends at ns
                        yield_k = RP.round_end[round];
                        if (round == 0u) {
                            rng_init(g, (uint64_t)(1995 + pj * RF.nx + pi) + RF.seed_salt);
                            col = mk(0, 0, 0);
                            k = 0;
                            has_pixel = true;
                            need_primary = true;
                        } else {
                            waiting = true;
                        }
                    }
                }
            }
            if (ring_on && waiting) { // has the previous round's lane published it yet?
                uint32_t w[10];
                bool beyond;
                const bool got_record = ring_poll(RP.round_state, RP.round_bytes, slot_round & 0x3FFFFFFu, RP.frame_id * 64u + (slot_round >> 26), w, beyond);
                if (__builtin_expect(beyond, 0)) atomicAdd(&RP.band_counters[kTornCounter], 1u); // a torn granule: counted, polled again
                if (got_record) {
                    g.v0 = w[0], g.v1 = w[1], g.v2 = w[2], g.v3 = w[3], g.v4 = w[4], g.d = w[5];
                    col = mk(__uint_as_float(w[6]), __uint_as_float(w[7]), __uint_as_float(w[8]));
                    k = (int)w[9];
                    has_pixel = true;
                    need_primary = true;
                    waiting = false;
                }
            }
            RT_STAT(kStatPollFails, __popcll(ballot(waiting)));
        }
#ifdef RT_TRAV_STATS
        const unsigned long long clk_fetch = __builtin_readcyclecounter();
        RT_STAT(kStatFetchClocks, clk_fetch - clk_pub);
#endif
        RT_STAT(kStatPrimaryLanes, __popcll(ballot(has_pixel && !tracing && need_primary)));
        if (has_pixel && !tracing) {
            if (need_primary) {
                cur = primary_ray(RF.cam, (int)(pij & 0xFFFFu), (int)(pij >> 16), RF.nx, RF.ny, g);
                thr = mk(1.0f, 1.0f, 1.0f);
                bounce = 0;
                need_primary = false;
            }
#ifdef RT_TRAV_STATS
            if (P.ray_dump) { // analysis only: every ray of the frame, in the order the waves start them
                const unsigned long long k = atomicAdd(P.ray_dump_count, 1ull);
                if (k < P.ray_dump_cap) {
                    float *o = P.ray_dump + 6 * k;
                    o[0] = cur.o.x, o[1] = cur.o.y, o[2] = cur.o.z, o[3] = cur.d.x, o[4] = cur.d.y, o[5] = cur.d.z;
                }
            }
#endif
            trav_begin(T, cur, K.sp0);
            tracing = true;
            nr++;
        }
        const unsigned long long live_m = ballot(!retired);
        if (live_m == 0ull) break;
        // The refill threshold is a share of the lanes that still have work: at the end of a frame retired lanes pile
        // up, and a fixed count would send the wave through the whole refill phase for every single ray that ends.
        const int thr_eff = (RP.threshold * (int)__popcll(live_m) + 63) >> 6;
        const int batch_eff = (RP.leaf_batch * (int)__popcll(live_m) + 63) >> 6; // the leaf batch likewise
        if (ring_on && ballot(tracing) == 0ull) {
            // every live lane waits for a ring item: nothing to do but poll.  Sleep between polls (127 x 64 clocks,
            // ~3.5 us) so that the waiting wave leaves the SIMD's issue slots to the waves that still produce.
            RT_STAT(kStatSpinPolls, 1);
            __builtin_amdgcn_s_sleep(127);
        }

#ifdef RT_TRAV_STATS
        const unsigned long long clk1 = __builtin_readcyclecounter();
        RT_STAT(kStatRefillClocks, clk1 - clk0);
        RT_STAT(kStatPrimaryClocks, clk1 - clk_fetch);
#endif
        // ---------------- traversal phase
        // Ballots are taken from comparisons of T.cur / T.pend made where they are needed: a ballot of a bool that
        // was set in another block costs hipcc a v_cndmask and a v_cmp each time.  The invariant above makes
        // "tracing &&" redundant in them.
        unsigned long long act = ballot(tracing);
        for (;;) {
            if (act == 0ull) break;
            // lanes the refill phase could do something for (a lane that waits for a ring item is not one)
            if ((int)__popcll(act) < thr_eff && ballot(!tracing && !retired && !waiting) != 0ull) break;
            if (ring_on && ++since_poll > 8 && ballot(waiting) != 0ull) break; // poll now and then
            RT_STAT(kStatIters, 1);
            RT_STAT(kStatIdleLanes, 64 - __popcll(act));
            RT_STAT(kStatRetiredLanes, __popcll(ballot(retired && !tracing)));
            RT_STAT(kStatWaitLanes, __popcll(ballot(waiting)));
            if (SPEC == 0) {
                if (tracing) {
                    if (!trav_step<FMT>(T, S, cur, K)) {
                        tracing = false;
                        result_ready = true;
                    }
                }
                act = ballot(tracing);
            } else {
#ifdef RT_TRAV_STATS
                {
                    const int nl = __popcll(ballot(tracing && !ref_is_leaf<FMT>(T.cur)));
                    RT_STAT(kStatNodeLanes, nl);
                    RT_STAT(kStatNodeIters, nl != 0);
                    for (int bb = 0; bb < 8; bb++) { // divergence per bounce: lanes of each bounce depth testing a node
                        const int nb = __popcll(ballot(tracing && !ref_is_leaf<FMT>(T.cur) && (bounce < 7 ? bounce : 7) == bb));
                        RT_STAT(kStatNodeLanesByBounce + bb, nb);
                        RT_STAT(kStatItersWithBounce + bb, nb != 0);
                    }
                    RT_STAT(kStatStalledLanes, __popcll(ballot(tracing && T.cur != RT_REF_NONE && ref_is_leaf<FMT>(T.cur) && T.pend != RT_REF_NONE)));
                }
#endif
#ifdef RT_TRAV_STATS
                const unsigned long long clk_n0 = __builtin_readcyclecounter();
#endif
                // a stack that starts the step below fast_limit ends it at most three entries higher: the pop of the
                // leaf parking below finds its entry in LDS too
                const bool lds_only = T.sp < K.fast_limit;
                if (!ref_is_leaf<FMT>(T.cur)) node_step<FMT, DET>(T, S, K);
#ifdef RT_TRAV_STATS
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the step's stack reads have come back
                RT_STAT(kStatNodeClocks, __builtin_readcyclecounter() - clk_n0);
#endif
                // park a reached leaf in the pending slot and go on with the next stack entry
                if (T.pend == RT_REF_NONE && T.cur != RT_REF_NONE && ref_is_leaf<FMT>(T.cur)) {
                    T.pend = T.cur;
                    RT_CULL_SYNC(T); // every leaf parked before this one has been tested by now
                    if (__builtin_expect(lds_only, 1)) {
                        if (T.sp == K.sp0) {
                            T.cur = RT_REF_NONE;
                        } else {
                            T.sp -= 256u;
                            T.cur = *stk_lds(T.sp);
                        }
                    } else {
                        T.cur = pop_or_none(K, T.sp);
                    }
                }
                const bool want_leaf = T.pend != RT_REF_NONE;
                const unsigned long long leaf_mask = ballot(want_leaf);
                if (leaf_mask != 0ull) {
                    const unsigned long long node_mask = ballot(!ref_is_leaf<FMT>(T.cur));
                    if (node_mask == 0ull || (int)__popcll(leaf_mask) >= batch_eff) {
                        RT_STAT(kStatLeafFires, 1);
                        RT_STAT(kStatLeafLanes, __popcll(leaf_mask));
#ifdef RT_TRAV_STATS
                        {
                            uint32_t np = want_leaf ? leaf_count<FMT>(T.pend) : 0u;
                            for (int o = 32; o > 0; o >>= 1) np += __shfl_xor(np, o);
                            RT_STAT(kStatLeafPrims, np);
                        }
#endif
#ifdef RT_TRAV_STATS
                        const unsigned long long clk_l0 = __builtin_readcyclecounter();
#endif
                        leaf_fire<FMT>(T, S, cur, want_leaf, leaf_mask, lds_addr(s_child));
#ifdef RT_TRAV_STATS
                        RT_STAT(kStatLeafClocks, __builtin_readcyclecounter() - clk_l0);
#endif
                    }
                }
                const bool fin = (T.cur & T.pend) == RT_REF_NONE;
                if (tracing && fin) {
                    tracing = false;
                    result_ready = true;
                }
                act = ballot(!fin);
            }
        }
#ifdef RT_TRAV_STATS
        RT_STAT(kStatTravClocks, __builtin_readcyclecounter() - clk1);
#endif
    }
    add_rays(F.ray_counter, nr);
#ifdef RT_TRAV_STATS
    if (lane == 0 && P.stats)
        for (int i = 0; i < kTravStats; i++)
            if (st[i]) atomicAdd(&P.stats[i], st[i]);
    if (lane == 0 && P.wave_times) {
        P.wave_times[3 * (size_t)blockIdx.x] = wave_t0;
        P.wave_times[3 * (size_t)blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
        P.wave_times[3 * (size_t)blockIdx.x + 2] = wave_t_dry;
    }
#endif
}

// ------------------------------------------------------------------ closest hit for an array of rays
// Persistent waves over a ray array: a wave takes rays 256 at a time from one counter and hands them to its lanes one
// by one; a lane that finishes its ray writes (id, t) and takes the next as soon as fewer than `threshold` lanes are
// still traversing -- taking a ray costs six loads, so the threshold can sit much higher than the render kernel's,
// whose refill is a shading step.  Traversal is k_persistent's (speculative, same node and leaf code).
struct QueueArgs {
    const float *rays; // origin[3], direction[3] per ray
    uint32_t n;
    unsigned int *next; // zeroed before the launch
    uint32_t chunk;     // rays a wave takes per atomic on `next`
    int32_t *ids;
    float *ts;
    unsigned long long *ray_counter;
};
template <int FMT, bool DET>
__global__ void __launch_bounds__(64, 6) k_trace_queue(KernelScene S, PersistArgs P, QueueArgs Q) {
    constexpr bool SCALED = true; // the caller's rays: any direction length (dir_scale)
    const uint32_t lane = threadIdx.x;
    StackRef K;
    stack_ref_init(K, lane, P.spill, P.lds_entries, P.total_lanes);
    uint32_t q_next = 0, q_end = 0; // wave-uniform: this wave's run of ray indices
    bool exhausted = false;          // wave-uniform: the counter has passed n
    bool tracing = false;
    uint32_t my_ray = 0, nr = 0;
    Ray cur;
    cur.o = mk(0, 0, 0);
    cur.d = mk(0, 0, 1);
    Trav T;
    trav_begin<SCALED>(T, cur, K.sp0);
    for (;;) {
        // ---- hand out rays to the idle lanes
        for (int guard = 0; guard < 4; guard++) {
            const unsigned long long idle = ballot(!tracing);
            if (idle == 0ull || exhausted) break;
            if (q_next == q_end) {
                uint32_t base = 0;
                if (lane == 0u) base = atomicAdd(Q.next, Q.chunk);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                if (base >= Q.n) {
                    exhausted = true;
                    break;
                }
                q_next = base;
                q_end = base + Q.chunk < Q.n ? base + Q.chunk : Q.n;
            }
            const uint32_t avail = q_end - q_next;
            const uint32_t before = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            if (!tracing && before < avail) {
                my_ray = q_next + before;
                const float2 *r = reinterpret_cast<const float2 *>(Q.rays + 6 * (size_t)my_ray);
                const float2 a = r[0], b = r[1], c = r[2];
                cur.o = mk(a.x, a.y, b.x);
                cur.d = mk(b.y, c.x, c.y);
                trav_begin<SCALED>(T, cur, K.sp0);
                tracing = true;
                nr++;
            }
            const uint32_t cnt = (uint32_t)__popcll(idle);
            q_next += cnt < avail ? cnt : avail;
        }
        const unsigned long long act0 = ballot(tracing);
        if (act0 == 0ull) break; // nothing in flight and nothing left to take
        // ---- traverse until enough lanes are free to make a refill worth its few instructions
        for (;;) {
            const unsigned long long act = ballot(tracing);
            if (act == 0ull) break;
            if (!exhausted && (int)__popcll(act) < P.threshold) break;
            if (tracing && !ref_is_leaf<FMT>(T.cur)) node_step<FMT, DET, SCALED>(T, S, K);
            if (tracing && T.pend == RT_REF_NONE && T.cur != RT_REF_NONE && ref_is_leaf<FMT>(T.cur)) {
                T.pend = T.cur;
                    RT_CULL_SYNC(T); // every leaf parked before this one has been tested by now
                T.cur = pop_or_none(K, T.sp);
            }
            const bool want_leaf = tracing && T.pend != RT_REF_NONE;
            const unsigned long long leaf_mask = ballot(want_leaf);
            if (leaf_mask != 0ull) {
                const unsigned long long node_mask = ballot(tracing && !ref_is_leaf<FMT>(T.cur));
                // once the rays have run out the batch is a share of the lanes still tracing
                const int batch_eff = exhausted ? (P.leaf_batch * (int)__popcll(act) + 63) >> 6 : P.leaf_batch;
                if (node_mask == 0ull || (int)__popcll(leaf_mask) >= batch_eff)
                    leaf_fire<FMT>(T, S, cur, want_leaf, leaf_mask, lds_addr(s_child));
            }
            if (tracing && T.cur == RT_REF_NONE && T.pend == RT_REF_NONE) {
                tracing = false;
                if (T.best_rank != RT_NO_HIT) {
                    Q.ids[my_ray] = __float_as_int(S.prims[PrimIndex<FMT>::stride * (size_t)T.best_rank + 2].z);
                    Q.ts[my_ray] = T.best_t;
                } else {
                    Q.ids[my_ray] = -1;
                    Q.ts[my_ray] = 0.0f;
                }
            }
        }
    }
    add_rays(Q.ray_counter, nr);
}

// (A kernel with two paths per lane -- the traced one exchanged with a parked one by v_swap_b32 when it finishes,
// shading batched on the parked one -- was built and measured in round 1: bit-exact, 116 VGPRs, 4 waves per SIMD, 73 ms
// against 62 ms.  It lives on branch exp/k-dual.)

// ------------------------------------------------------------------ static scratch (kernels.h, static_scratch)
// In host code the array's name is only a stand-in for the symbol: it is never passed to a kernel or to a HIP call as
// an address.  The address on a device comes from hipGetSymbolAddress, once per device (every GPU holds its own copy of
// the array at its own address), and only that pointer is used afterwards.
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

hipError_t static_scratch(int device, void **ptr) {
    constexpr int kMaxDev = 64;
    static std::mutex mu;
    static void *kept[kMaxDev] = {};
    if (!ptr || device < 0 || device >= kMaxDev) return hipErrorInvalidValue;
    std::lock_guard<std::mutex> lock(mu);
    if (!kept[device]) {
        void *p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        if ((e = hipMemset(p, 0, kStaticScratchBytes)) != hipSuccess) return e; // counters and record tags start at zero
        kept[device] = p;
    }
    *ptr = kept[device];
    return hipSuccess;
}

// ------------------------------------------------------------------ launchers (host)
static inline hipError_t lds_bytes_ok(int stack_entries, size_t &bytes) {
    bytes = (size_t)stack_entries * 64 * sizeof(uint32_t);
    return bytes <= 64 * 1024 ? hipSuccess : hipErrorInvalidValue;
}

hipError_t launch_trace_primary(const KernelScene &S, const KernelFrame &F, int stack_entries, int32_t *ids, float *ts,
                                hipStream_t stream) {
    size_t lds;
    if (lds_bytes_ok(stack_entries, lds) != hipSuccess) return hipErrorInvalidValue;
    if (F.n_local_tiles <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_trace_primary, dim3(F.n_local_tiles), dim3(64), lds, stream, S, F, ids, ts);
    return hipGetLastError();
}

hipError_t launch_trace_rays(const KernelScene &S, int stack_entries, const float *org_dir6, int n, int32_t *ids,
                             float *ts, unsigned long long *counter, hipStream_t stream) {
    size_t lds;
    if (lds_bytes_ok(stack_entries, lds) != hipSuccess) return hipErrorInvalidValue;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_trace_rays, dim3((n + 63) / 64), dim3(64), lds, stream, S, org_dir6, n, ids, ts, counter);
    return hipGetLastError();
}

hipError_t launch_trace_queue(const KernelScene &S, const PersistLaunch &L, const float *rays6, uint32_t n, unsigned int *next,
                              int32_t *ids, float *ts, unsigned long long *ray_counter, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    if (L.lds_entries < 1 || L.grid_waves < 1) return hipErrorInvalidValue;
    const size_t lds = (size_t)L.lds_entries * 64 * sizeof(uint32_t);
    if (lds > 62 * 1024) return hipErrorInvalidValue; // 64 KB per workgroup, less the static child table / leaf exchange
    PersistArgs P = {};
    P.spill = L.spill;
    P.lds_entries = L.lds_entries;
    P.threshold = L.threshold;
    P.leaf_batch = L.leaf_batch;
    P.total_lanes = (uint32_t)L.grid_waves * 64u;
    QueueArgs Q;
    Q.rays = rays6;
    Q.n = n;
    Q.next = next;
    Q.chunk = 256u; // 1024 and 4096 measured no better on 288 M rays (39.1 / 39.4 / 41.4 ms): one counter is enough here
    Q.ids = ids;
    Q.ts = ts;
    Q.ray_counter = ray_counter;
    const uint32_t waves = std::min<uint32_t>((uint32_t)L.grid_waves, (n + 63u) / 64u);
    if (L.node_format == 0)
        hipLaunchKernelGGL((k_trace_queue<0, true>), dim3(waves), dim3(64), lds, stream, S, P, Q);
    else if (L.node_format == 2 && L.deterministic)
        hipLaunchKernelGGL((k_trace_queue<2, true>), dim3(waves), dim3(64), lds, stream, S, P, Q);
    else if (L.node_format == 2)
        hipLaunchKernelGGL((k_trace_queue<2, false>), dim3(waves), dim3(64), lds, stream, S, P, Q);
    else if (L.deterministic)
        hipLaunchKernelGGL((k_trace_queue<1, true>), dim3(waves), dim3(64), lds, stream, S, P, Q);
    else
        hipLaunchKernelGGL((k_trace_queue<1, false>), dim3(waves), dim3(64), lds, stream, S, P, Q);
    return hipGetLastError();
}

hipError_t launch_render(const KernelScene &S, const KernelFrame &F, int stack_entries, float *fb, hipStream_t stream) {
    size_t lds;
    if (lds_bytes_ok(stack_entries, lds) != hipSuccess) return hipErrorInvalidValue;
    if (F.n_local_tiles <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_render, dim3(F.n_local_tiles), dim3(64), lds, stream, S, F, fb);
    return hipGetLastError();
}

hipError_t launch_persistent(int mode, const KernelScene &S, const KernelFrame &F, const PersistLaunch &L, float *fb,
                             int32_t *ids, float *ts, hipStream_t stream) {
    if (F.n_local_tiles <= 0) return hipSuccess;
    if (L.lds_entries < 1 || L.grid_waves < 1 || F.nx > 65535 || F.ny > 65535) return hipErrorInvalidValue; // pij packing
    size_t lds = (size_t)L.lds_entries * 64 * sizeof(uint32_t);
    if (lds > 62 * 1024) return hipErrorInvalidValue; // 64 KB per workgroup, less the static child table / leaf exchange
    PersistArgs P;
    P.band_counters = L.band_counters;
    P.spill = L.spill;
    P.lds_entries = L.lds_entries;
    P.threshold = L.threshold;
    P.leaf_batch = L.leaf_batch;
    P.total_lanes = (uint32_t)L.grid_waves * 64u;
    P.stats = L.stats;
    P.ray_dump = L.ray_dump;
    P.ray_dump_count = L.ray_dump_count;
    P.ray_dump_cap = L.ray_dump_cap;
    P.wave_times = L.wave_times;
    P.rounds = mode == 0 && L.rounds > 1 ? L.rounds : 1;
    for (int r = 0; r < 64; r++) P.round_end[r] = r < P.rounds - 1 ? L.round_end[r] : F.ns;
    for (int r = 0; r < P.rounds; r++) // every round runs at least one sample, the last one ends the pixel
        if (P.round_end[r] <= (r ? P.round_end[r - 1] : 0) || P.round_end[r] > F.ns) return hipErrorInvalidValue;
    // primary rays are short: at 64 slots per atomic the eight band counters (one address each, 32 400 atomics for a
    // 1080p frame) set the kernel's time -- 0.65 ms, 0.53 with 256 slots per atomic
    P.chunk = mode == 1 && L.primary_chunk >= 64 ? (uint32_t)L.primary_chunk : 64u;
    P.round_state = L.round_state;
    P.frame_id = L.frame_id;
    const uint64_t ring_bytes = (uint64_t)F.n_local_tiles * 64u * kRingWords * 4u;
    P.round_bytes = (uint32_t)ring_bytes;
    if (P.rounds > 1 && (!P.round_state || P.frame_id == 0u || P.frame_id >= (1u << 26) || P.rounds > 63 || P.rounds > F.ns ||
                         (uint64_t)F.n_local_tiles * 64u >= (1ull << 26) || ring_bytes > 0xFFFFFFFFull))
        return hipErrorInvalidValue;
    const dim3 grid(L.grid_waves), block(64);
    // DET 0 (cull against the freshest hit, results timing-dependent on ~10^-10 of rays) exists only where it differs:
    // speculative traversal over quantised nodes
    const bool det = L.deterministic || !L.speculative || L.node_format == 0;
    if (L.node_format == 2) { // packed scene: speculative traversal only
        if (!L.speculative) return hipErrorInvalidValue;
        if (mode == 0 && S.uv != nullptr) {
            if (det)
                hipLaunchKernelGGL((k_persistent<0, 2, 1, 4, true, true>), grid, block, lds, stream, S, F, P, fb, ids, ts);
            else
                hipLaunchKernelGGL((k_persistent<0, 2, 1, 4, true, false>), grid, block, lds, stream, S, F, P, fb, ids, ts);
        } else if (mode == 0) {
            if (det)
                hipLaunchKernelGGL((k_persistent<0, 2, 1, 6, false, true>), grid, block, lds, stream, S, F, P, fb, ids, ts);
            else
                hipLaunchKernelGGL((k_persistent<0, 2, 1, 6, false, false>), grid, block, lds, stream, S, F, P, fb, ids, ts);
        } else {
            if (det)
                hipLaunchKernelGGL((k_persistent<1, 2, 1, 6, false, true>), grid, block, lds, stream, S, F, P, fb, ids, ts);
            else
                hipLaunchKernelGGL((k_persistent<1, 2, 1, 6, false, false>), grid, block, lds, stream, S, F, P, fb, ids, ts);
        }
        return hipGetLastError();
    }
    if (mode == 0 && S.uv != nullptr) { // textured scene: the instances with the texture path, 128 VGPRs (4 waves/SIMD)
        if (L.node_format == 0 && !L.speculative)
            hipLaunchKernelGGL((k_persistent<0, 0, 0, 4, true, true>), grid, block, lds, stream, S, F, P, fb, ids, ts);
        else if (L.node_format == 0)
            hipLaunchKernelGGL((k_persistent<0, 0, 1, 4, true, true>), grid, block, lds, stream, S, F, P, fb, ids, ts);
        else if (!L.speculative)
            hipLaunchKernelGGL((k_persistent<0, 1, 0, 4, true, true>), grid, block, lds, stream, S, F, P, fb, ids, ts);
        else if (det)
            hipLaunchKernelGGL((k_persistent<0, 1, 1, 4, true, true>), grid, block, lds, stream, S, F, P, fb, ids, ts);
        else
            hipLaunchKernelGGL((k_persistent<0, 1, 1, 4, true, false>), grid, block, lds, stream, S, F, P, fb, ids, ts);
        return hipGetLastError();
    }
#define RT_LAUNCH(M, Fm, Sp, Oc, Dt)                                                                                   \
    hipLaunchKernelGGL((k_persistent<M, Fm, Sp, Oc, false, Dt>), grid, block, lds, stream, S, F, P, fb, ids, ts)
#define RT_LAUNCH_OCC(M, Sp, Dt)                                                                                       \
    if (L.occupancy >= 8)                                                                                              \
        RT_LAUNCH(M, 1, Sp, 8, Dt);                                                                                    \
    else if (L.occupancy >= 6)                                                                                         \
        RT_LAUNCH(M, 1, Sp, 6, Dt);                                                                                    \
    else                                                                                                               \
        RT_LAUNCH(M, 1, Sp, 4, Dt);
    const int sel = (mode ? 4 : 0) | (L.node_format ? 2 : 0) | (L.speculative ? 1 : 0);
    switch (sel) {
    case 0: RT_LAUNCH(0, 0, 0, 4, true); break;
    case 1: RT_LAUNCH(0, 0, 1, 4, true); break;
    case 2: RT_LAUNCH_OCC(0, 0, true) break;
    case 3:
        if (det) {
            RT_LAUNCH_OCC(0, 1, true)
        } else {
            RT_LAUNCH_OCC(0, 1, false)
        }
        break;
    case 4: RT_LAUNCH(1, 0, 0, 4, true); break;
    case 5: RT_LAUNCH(1, 0, 1, 4, true); break;
    case 6: RT_LAUNCH_OCC(1, 0, true) break;
    default:
        if (det) {
            RT_LAUNCH_OCC(1, 1, true)
        } else {
            RT_LAUNCH_OCC(1, 1, false)
        }
        break;
    }
#undef RT_LAUNCH_OCC
#undef RT_LAUNCH
    return hipGetLastError();
}

} // namespace rtamd
