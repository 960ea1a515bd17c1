// How many L1 (TCP) accesses per cycle a CU sustains: every lane of a wave loads 16 bytes from a line of its own, all
// lines resident in the CU's 32 KiB L1 (the pattern of the trace kernel's node step, where lanes on unrelated paths
// share no line), against the same loads with all 64 lanes in ONE line.  Prints lane-loads per cycle and CU at the
// nominal clock, for 4 .. 32 waves per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/l1_rate.hip -o tools/micro/l1_rate && tools/micro/l1_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// table: 16 KiB per workgroup slot (256 lines of 64 B); stride: 64-byte lines between consecutive lanes (2 = a 128-byte
// line each, 1 = a 64-byte line each, 0 = one line for all)
template <int STRIDE>
__global__ void __launch_bounds__(64) k_l1(const uint4 *table, uint4 *out, int iters) {
    const uint4 *base = table + (size_t)(blockIdx.x & 7) * 1024; // a few tables, each L1-resident
    uint32_t line = (threadIdx.x * STRIDE) & 255u;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint4 v = base[line * 4 + (r & 3)]; // 16 bytes of the lane's line
            acc.x ^= v.x, acc.y += v.y, acc.z ^= v.z, acc.w += v.w;
            line = (line + 64u * STRIDE + 1u) & 255u; // next line: data-independent, so loads pipeline
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

int main() {
    hipDeviceProp_t p;
    CHK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const double ghz = p.clockRate / 1e6;
    uint4 *table, *out;
    CHK(hipMalloc(&table, 8 * 16384));
    CHK(hipMemset(table, 1, 8 * 16384));
    CHK(hipMalloc(&out, (size_t)cus * 32 * 64 * sizeof(uint4)));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    const int iters = 4000;
    for (int stride = 2; stride >= 0; stride--)
        for (int w = 4; w <= 32; w *= 2) {
            const int blocks = cus * w;
            for (int rep = 0; rep < 2; rep++) {
                CHK(hipEventRecord(e0));
                if (stride == 2) k_l1<2><<<blocks, 64>>>(table, out, iters);
                else if (stride == 1) k_l1<1><<<blocks, 64>>>(table, out, iters);
                else k_l1<0><<<blocks, 64>>>(table, out, iters);
                CHK(hipEventRecord(e1));
                CHK(hipEventSynchronize(e1));
            }
            float ms;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            const double insts = (double)iters * 8 * w; // wave-level load instructions per CU
            const double cycles = ms * 1e-3 * ghz * 1e9;
            printf("{\"pattern\": \"%s\", \"waves_per_cu\": %d, \"ms\": %.3f, \"clock_ghz\": %.2f, \"cycles_per_wave_load\": %.2f, "
                   "\"lane_loads_per_cycle_per_cu\": %.3f, \"GB_per_s_chip_at_64B_per_access\": %.0f}\n",
                   stride == 2 ? "64 lanes, 64 x 128-byte lines (16 B each)" : stride == 1 ? "64 lanes, 64 x 64-byte lines = 32 x 128-byte lines" : "64 lanes, one line", w, ms, ghz, cycles / insts,
                   (stride ? 64.0 : 1.0) * insts / cycles, (stride ? 64.0 : 1.0) * insts * 64.0 * cus / (ms * 1e-3) / 1e9);
        }
    return 0;
}
