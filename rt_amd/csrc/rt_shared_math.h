// rt_shared_math.h -- arithmetic that must be bit-identical on the host and on the device:
//   * the cuRAND XORWOW generator restated (reference call sites src/pbrt/pbrt_engine.cu:580,599);
//   * the "shared transcendentals" of DESIGN.md that stand in for CUDA's __cosf/__sinf
//     (src/pbrt/utilities.cu:23-24), tan (include/RT/rt_camera.h:43) and pow (include/RT/material.h:10):
//     binary64, no FMA, k = floor(x*2/pi + 1/2), r = (x - k*PIO2_HI) - k*PIO2_LO, Taylor polynomials of
//     degree 13 / 14 in Horner form on r*r, quadrant fix-up, one rounding to binary32 at the end.
// Compile every user with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#pragma clang fp contract(off)

namespace rtamd {

struct Rng {
    uint32_t v0, v1, v2, v3, v4, d;
};
// curand_init(seed, 0, 0, &s)
__host__ __device__ inline void rng_init(Rng &r, uint64_t seed) {
    uint32_t s0 = (uint32_t)seed ^ 0xaad26b49u;
    uint32_t s1 = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
    uint32_t t0 = 1099087573u * s0;
    uint32_t t1 = 2591861531u * s1;
    r.d = 6615241u + t1 + t0;
    r.v0 = 123456789u + t0;
    r.v1 = 362436069u ^ t0;
    r.v2 = 521288629u + t1;
    r.v3 = 88675123u ^ t1;
    r.v4 = 5783321u + t0;
}
// curand(&s)
__host__ __device__ inline uint32_t rng_next(Rng &r) {
    uint32_t t = r.v0 ^ (r.v0 >> 2);
    r.v0 = r.v1;
    r.v1 = r.v2;
    r.v2 = r.v3;
    r.v3 = r.v4;
    r.v4 = (r.v4 ^ (r.v4 << 4)) ^ (t ^ (t << 1));
    r.d += 362437u;
    return r.v4 + r.d;
}
// curand_uniform(&s): (0, 1]
__host__ __device__ inline float rng_uniform(Rng &r) {
    return (float)rng_next(r) * 2.3283064e-10f + (2.3283064e-10f / 2.0f);
}

__host__ __device__ inline void sincos_sh(double x, double &s, double &c) {
    const double TWO_OVER_PI = 0.63661977236758134308;
    const double PIO2_HI = 1.57079632679489655800e+00;
    const double PIO2_LO = 6.12323399573676603587e-17;
    double kd = __builtin_floor(x * TWO_OVER_PI + 0.5);
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
    int k = (int)(long long)kd & 3;
    s = (k == 0) ? sr : (k == 1) ? cr : (k == 2) ? -sr : -cr;
    c = (k == 0) ? cr : (k == 1) ? -sr : (k == 2) ? -cr : sr;
}
__host__ __device__ inline float tan_sh(float x) {
    double s, c;
    sincos_sh((double)x, s, c);
    return (float)(s / c);
}
__host__ __device__ inline float pow5_sh(float x) {
    double d = (double)x;
    double d2 = d * d;
    double d4 = d2 * d2;
    return (float)(d4 * d);
}

} // namespace rtamd
