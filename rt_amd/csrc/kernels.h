// kernels.h -- host-visible launchers of the HIP kernels in kernels.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/rt_abi.h"
#include "rt_types.h"

namespace rtamd {

struct KernelScene {
    const float4 *nodes; // WideNode[], 8 float4 each
    const uint4 *qnodes; // QuantNode[], 4 uint4 each (same indices)
    const float4 *prims; // DevPrim[], 3 float4 each, DFS leaf order
    const float4 *shade; // DevShade[], 1 float4 each
    const DevMat *mats;  // RT_MAT_NUM entries
    const float4 *uv;    // DevUV[], 2 float4 each, or nullptr: no triangle of the scene is textured
    const DevTex *tex;   // RT_TEX_NUM entries (only read when uv != nullptr)
};

// n / d for any 32-bit n by a multiply-high and two shifts (Granlund & Montgomery's round-up method): hipcc expands a
// division by a run-time value into ~25 vector instructions, and the kernels divide a tile index by tiles_x for
// every pixel a lane takes.  shift == ~0u stands for d == 1.
struct FastDiv {
    uint32_t mul, shift;
};
inline FastDiv fastdiv_make(uint32_t d) {
    FastDiv f;
    if (d <= 1u) {
        f.mul = 0u;
        f.shift = ~0u;
        return f;
    }
    uint32_t l = 0;
    while ((1ull << l) < d) l++; // ceil(log2 d), >= 1
    f.mul = (uint32_t)((((1ull << 32) * ((1ull << l) - d)) / d) + 1ull);
    f.shift = l - 1u;
    return f;
}
inline uint32_t fastdiv_host(const FastDiv &f, uint32_t n) {
    if (f.shift == ~0u) return n;
    const uint32_t t = (uint32_t)(((uint64_t)f.mul * n) >> 32);
    return (t + ((n - t) >> 1)) >> f.shift;
}

struct KernelFrame {
    DevCamera cam;
    int nx, ny, ns, max_depth;
    int tiles_x, tiles_y;
    FastDiv div_tiles_x; // fastdiv_make(tiles_x)
    int n_local_tiles; // tiles this context renders = grid size
    int shard_rank, shard_world;
    unsigned long long *ray_counter;
    unsigned long long seed_salt; // added to the reference's per-pixel seed 1995 + pixel_index; 0 = the reference's streams
    int raw_sum;                  // 1: write the per-pixel sum of (clamped) samples, not the clamped mean
};

hipError_t launch_trace_primary(const KernelScene &S, const KernelFrame &F, int stack_entries, int32_t *ids, float *ts,
                                hipStream_t stream);
hipError_t launch_trace_rays(const KernelScene &S, int stack_entries, const float *org_dir6, int n, int32_t *ids,
                             float *ts, unsigned long long *counter, hipStream_t stream);
// persistent-wave variant (kernel 1): mode 0 = render, 1 = primary ids
struct PersistLaunch {
    unsigned int *band_counters; // 8 counters, zeroed by the caller on the same stream
    uint32_t *spill;             // (stack_entries - lds_entries) * grid_waves * 64 words
    int lds_entries;
    int threshold;
    int grid_waves;
    int node_format; // 0 = WideNode (128 B float), 1 = QuantNode (64 B), 2 = packed scene (48-byte records; speculative only)
    int speculative; // 1 = postpone leaves (speculative traversal)
    int deterministic = 1; // speculative traversal over quantised nodes: 1 = cull against a value that is refreshed at points of the
                           // lane's own sequence only (reproducible results), 0 = against the freshest (RT_OPT_DETERMINISTIC)
    int leaf_batch;
    int primary_chunk = 256; // primary rays: slots per band-counter atomic (a multiple of 64)
    int occupancy; // 4, 6 or 8 waves per SIMD the kernel's registers are held to (quantized nodes only)
    // rounds (render only): the frame is walked `rounds` times, pixels hand their state on between rounds; see
    // "Rounds" at k_persistent
    int rounds;            // 1 = off; at most 63 and at most ns
    int round_end[64];     // round r runs the samples [round_end[r - 1], round_end[r]) of a pixel; round_end[rounds - 1] = ns
    uint32_t *round_state; // kRingWords words per slot (n_local_tiles * 64 slots), 16-byte aligned, zeroed once at allocation
    uint32_t frame_id;     // never 0, below 2^26, never reused on a record array: tags of earlier launches compare unequal, no clearing per frame
    float *ray_dump = nullptr;                     // analysis builds: rt_test_dump_rays
    unsigned long long *ray_dump_count = nullptr;
    unsigned long long ray_dump_cap = 0;
    unsigned long long *wave_times = nullptr; // analysis builds: rt_test_wave_times
    unsigned long long *stats; // kTravStats counters, zeroed by the caller; only written by a -DRT_TRAV_STATS build
};
enum { kRingWords = 20 }; // ten 8-byte {value, check} granules per pixel slot
enum { kTornCounter = 8 };  // band_counters[8]: polls of a round record in which a granule yielded a tag beyond the expected one (a torn granule)
// Lane-occupancy counters of the persistent kernel (analysis builds only, `make STATS=1`; tools/trav_stats.py)
enum { kStatIters = 0, kStatNodeLanes, kStatStalledLanes, kStatIdleLanes, kStatLeafFires, kStatLeafLanes, kStatLeafPrims,
       kStatRefills, kStatShadeLanes, kStatPrimaryLanes, kStatFetchLanes, kStatNodeIters, kStatRetiredLanes, kStatWaitLanes, kStatSpinPolls, kStatPollFails /* 15: polls of a round flag that found the record missing */,
       kStatNodeLanesByBounce = 16 /* .. 23: bounce 0..6, 7+ */, kStatItersWithBounce = 24 /* .. 31 */,
       kStatRefillClocks = 32, kStatTravClocks, kStatShadeClocks, kStatNodeClocks, kStatLeafClocks, kStatPublishClocks, kStatFetchClocks, kStatPrimaryClocks, kTravStats = 40 };
// Device memory for callers that cannot allocate (rt_render_device_noalloc): ONE file-scope __device__ array in
// kernels.hip, kStaticScratchBytes per GPU, held from the code object's load on that GPU to the end of the process.
// static_scratch() returns its address on `device`, which must be the current device: the first call per device looks
// the symbol up (hipGetSymbolAddress) and zeroes the array; later calls return the kept pointer.  Layout (offsets):
//   kStaticBandsOff    band counters + analysis counters (kStaticBandsBytes)
//   kStaticCounterOff  ray counter (8 bytes)
//   kStaticSpillOff    traversal-stack spill area (kStaticSpillBytes)
//   kStaticRingOff     round records (kStaticRingBytes)
constexpr size_t kStaticScratchBytes = (size_t)64 << 20;
constexpr size_t kStaticBandsOff = 0, kStaticBandsBytes = 512;
constexpr size_t kStaticCounterOff = 512;
// (the bench scene -- 1 M triangles, 41 stack entries of which 16 live in LDS, 24 waves per CU -- spills 37.7 MB; the
// reference's default frame, 640x360, needs 18.4 MB of records)
constexpr size_t kStaticSpillOff = 4096, kStaticSpillBytes = ((size_t)40 << 20) - 4096;
constexpr size_t kStaticRingOff = (size_t)40 << 20, kStaticRingBytes = (size_t)24 << 20;
static_assert(kStaticRingOff + kStaticRingBytes == kStaticScratchBytes && kStaticRingOff % 256 == 0, "static scratch layout");
hipError_t static_scratch(int device, void **ptr);

hipError_t launch_persistent(int mode, const KernelScene &S, const KernelFrame &F, const PersistLaunch &L, float *fb,
                             int32_t *ids, float *ts, hipStream_t stream);
hipError_t launch_render(const KernelScene &S, const KernelFrame &F, int stack_entries, float *fb, hipStream_t stream);
// closest hit for n rays in device memory (k_trace_queue); uses L's grid, stack, threshold, leaf batch, node format
hipError_t launch_trace_queue(const KernelScene &S, const PersistLaunch &L, const float *rays6, uint32_t n, unsigned int *next,
                              int32_t *ids, float *ts, unsigned long long *ray_counter, hipStream_t stream);

} // namespace rtamd
