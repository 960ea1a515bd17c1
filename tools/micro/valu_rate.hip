// Vector-instruction issue rate of one SIMD with W resident waves: independent v_fma_f32 (and a mix like the node
// test's) in a long unrolled loop.  Prints wave64 instructions per cycle per SIMD at the measured clock.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o tools/micro/valu_rate && tools/micro/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int KIND>
__global__ void __launch_bounds__(64) k_rate(float *out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    unsigned u = threadIdx.x * 2654435761u;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
    const f2 pa = {a, a}, pb = {b, b};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            if (KIND == 0) { // 8 independent FMAs
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
            } else if (KIND == 1) { // the node test's mix: cvt, fma, max3, min3, cndmask, min_u32
                asm volatile("v_cvt_f32_ubyte0 %0, %8\n v_fma_f32 %1, %0, %9, %10\n v_cvt_f32_ubyte1 %2, %8\n v_fma_f32 %3, %2, %9, %10\n"
                             "v_max3_f32 %4, %1, %3, %4\n v_min3_f32 %5, %1, %3, %5\n v_cndmask_b32 %6, %4, %5, vcc\n v_min_u32 %7, %6, %7"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(u), "v"(a), "v"(b) : "vcc");
            } else if (KIND == 3) { // 8 mixed-precision FMAs: src0 an f16 half of a register (low / high alternating), f32 otherwise
                asm volatile("v_fma_mix_f32 %0, %8, %9, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %8, %9, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                             "v_fma_mix_f32 %2, %8, %9, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %8, %9, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                             "v_fma_mix_f32 %4, %8, %9, %4 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %5, %8, %9, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                             "v_fma_mix_f32 %6, %8, %9, %6 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %7, %8, %9, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(u), "v"(a), "v"(b));
            } else { // 4 packed FMAs on register pairs = 8 FMAs
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa), "v"(pb));
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p1.y + p2.x + p3.y;
}

int main() {
    hipDeviceProp_t p;
    CHK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const double ghz = p.clockRate / 1e6;
    float *out;
    CHK(hipMalloc(&out, (size_t)cus * 4 * 8 * 64 * sizeof(float)));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    const int iters = 20000;
    for (int kind = 0; kind < 4; kind++)
        for (int w = 1; w <= 8; w *= 2) {
            const int blocks = cus * 4 * w;
            for (int rep = 0; rep < 2; rep++) {
                CHK(hipEventRecord(e0));
                if (kind == 0) k_rate<0><<<blocks, 64>>>(out, iters, 1.0001f, 0.5f);
                else if (kind == 1) k_rate<1><<<blocks, 64>>>(out, iters, 1.0001f, 0.5f);
                else if (kind == 3) k_rate<3><<<blocks, 64>>>(out, iters, 1.0001f, 0.5f);
                else k_rate<2><<<blocks, 64>>>(out, iters, 1.0001f, 0.5f);
                CHK(hipEventRecord(e1));
                CHK(hipEventSynchronize(e1));
            }
            float ms;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            const double insts = (double)iters * 8 * (kind == 2 ? 4 : 8) * w; // per SIMD
            const double cycles = ms * 1e-3 * ghz * 1e9;
            printf("{\"kind\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"clock_ghz\": %.2f, \"cycles_per_wave64_inst\": %.2f}\n",
                   kind == 0 ? "v_fma_f32" : kind == 1 ? "node-test mix" : kind == 3 ? "v_fma_mix_f32 (f16 src0)" : "v_pk_fma_f32", w, ms, ghz, cycles / insts);
        }
    return 0;
}
