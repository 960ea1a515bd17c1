// kernels.h -- host-visible launchers of the HIP kernels in kernels.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/rt_abi.h"
#include "rt_types.h"

namespace rtamd {

struct KernelScene {
    const float4 *nodes; // WideNode[], 8 float4 each
    const float4 *prims; // DevPrim[], 3 float4 each, DFS leaf order
    const float4 *shade; // DevShade[], 1 float4 each
    const DevMat *mats;  // RT_MAT_NUM entries
};

struct KernelFrame {
    DevCamera cam;
    int nx, ny, ns, max_depth;
    int tiles_x, tiles_y;
    int n_local_tiles; // tiles this context renders = grid size
    int shard_rank, shard_world;
    unsigned long long *ray_counter;
};

hipError_t launch_trace_primary(const KernelScene &S, const KernelFrame &F, int stack_entries, int32_t *ids, float *ts,
                                hipStream_t stream);
hipError_t launch_trace_rays(const KernelScene &S, int stack_entries, const float *org_dir6, int n, int32_t *ids,
                             float *ts, unsigned long long *counter, hipStream_t stream);
hipError_t launch_render(const KernelScene &S, const KernelFrame &F, int stack_entries, float *fb, hipStream_t stream);

} // namespace rtamd
