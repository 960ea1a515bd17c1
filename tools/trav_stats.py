#!/usr/bin/env python3
"""Lane occupancy of the persistent kernel on the bench workload, from the analysis build:
    make -C rt_amd/csrc STATS=1 && RT_AMD_LIB=rt_amd/librt_amd_stats.so python tools/trav_stats.py [run_frame-like args]
Prints, per launch, how many of the 64 lanes did useful work in each phase of the loop."""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import rt_amd  # noqa: E402
from rt_amd import engine as E  # noqa: E402
from rt_amd import scenes  # noqa: E402

NAMES = ["iters", "node_lanes", "stalled_lanes", "idle_lanes", "leaf_fires", "leaf_lanes", "leaf_prims", "refills",
         "shade_lanes", "primary_lanes", "fetch_lanes", "node_iters", "retired_lanes", "wait_lanes", "spin_polls", "poll_fails"]


def stats(eng):
    out = np.zeros(40, dtype=np.uint64)
    eng._check(eng._lib.rt_test_trav_stats(eng._h, out.ctypes.data_as(C.c_void_p)), "rt_test_trav_stats")
    d = {n: int(out[i]) for i, n in enumerate(NAMES)}
    it = max(d["iters"], 1)
    d["node_lanes_per_iter"] = d["node_lanes"] / it
    d["node_lanes_per_node_iter"] = d["node_lanes"] / max(d["node_iters"], 1)
    d["stalled_per_iter"] = d["stalled_lanes"] / it
    d["idle_per_iter"] = d["idle_lanes"] / it
    d["retired_per_iter"] = d["retired_lanes"] / it
    d["wait_per_iter"] = d["wait_lanes"] / it
    d["leaf_lanes_per_fire"] = d["leaf_lanes"] / max(d["leaf_fires"], 1)
    d["leaf_prims_per_lane"] = d["leaf_prims"] / max(d["leaf_lanes"], 1)
    d["iters_per_leaf_fire"] = it / max(d["leaf_fires"], 1)
    # per bounce depth (0 = primary ray, 7 = seventh bounce and beyond): lanes testing a node, averaged over the
    # iterations in which that depth was present in the wave at all, and that depth's share of all node tests
    d["node_lanes_by_bounce"] = [int(out[16 + b]) / max(int(out[24 + b]), 1) for b in range(8)]
    d["node_share_by_bounce"] = [int(out[16 + b]) / max(d["node_lanes"], 1) for b in range(8)]
    clocks = float(int(out[32]) + int(out[33])) or 1.0  # wave clocks (s_memtime domain), summed over waves
    d["wave_time_share"] = {"refill_phase": int(out[32]) / clocks, "of_which_shading": int(out[34]) / clocks,
                            "traversal_phase": int(out[33]) / clocks,
                            "of_which_node_steps": int(out[35]) / clocks, "of_which_leaf_tests": int(out[36]) / clocks,
                            "refill_publish": int(out[37]) / clocks, "refill_fetch_and_restore": int(out[38]) / clocks,
                            "refill_primary_and_begin": int(out[39]) / clocks}
    d["shade_lanes_per_refill"] = d["shade_lanes"] / max(d["refills"], 1)
    d["iters_per_refill"] = it / max(d["refills"], 1)
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=708)
    ap.add_argument("--nx", type=int, default=1920)
    ap.add_argument("--ny", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--threshold", type=int, default=None)
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--leaf", type=int, default=None)
    ap.add_argument("--kernel", type=int, default=None)
    ap.add_argument("--shade", type=int, default=None)
    ap.add_argument("--swap", type=int, default=None)
    ap.add_argument("--occ", type=int, default=None)
    ap.add_argument("--waves", type=int, default=None)
    ap.add_argument("--tail", type=int, default=None)
    ap.add_argument("--pchunk", type=int, default=None)
    ap.add_argument("--pwaves", type=int, default=None)
    ap.add_argument("--det", type=int, default=None, help="RT_OPT_DETERMINISTIC")
    ap.add_argument("--fmt", type=int, default=None, help="RT_OPT_NODE_FORMAT")
    ap.add_argument("--primary-only", action="store_true", help="with --wave-times: the primary-ray kernel instead of the render")
    ap.add_argument("--wave-times", action="store_true", help="when each wave of the render ends (share of waves still running over time)")
    ap.add_argument("--queue-waves", type=lambda t: [int(x) for x in t.split(",")], default=[16])
    ap.add_argument("--sort", type=lambda t: [int(x) for x in t.split(",")], default=[0],
                    help="with --replay: ray orders to replay: 0 = as the waves started them, 1 = sorted by origin cell then octant, 2 = by octant then cell")
    ap.add_argument("--replay", type=str, default=None, metavar="T1,T2,...",
                    help="record the frame's rays and trace them again with the ray-array kernel at these hand-out thresholds")
    args = ap.parse_args()
    eng = rt_amd.PbrtEngine(args.nx, args.ny, args.spp, device=0)
    if args.threshold is not None:
        eng.setOption(E.OPT_REFILL_THRESHOLD, args.threshold)
    if args.batch is not None:
        eng.setOption(E.OPT_LEAF_BATCH, args.batch)
    if args.leaf is not None:
        eng.setOption(E.OPT_LEAF_SIZE, args.leaf)
    for opt, v in ((E.OPT_KERNEL, args.kernel), (E.OPT_SHADE_BATCH, args.shade), (E.OPT_SWAP_BATCH, args.swap),
                   (E.OPT_OCCUPANCY, args.occ), (E.OPT_WAVES_PER_CU, args.waves), (E.OPT_TAIL_SPLIT, args.tail),
                   (E.OPT_PRIMARY_CHUNK, args.pchunk), (E.OPT_PRIMARY_WAVES, args.pwaves), (E.OPT_DETERMINISTIC, args.det),
                   (E.OPT_NODE_FORMAT, args.fmt)):
        if v is not None:
            eng.setOption(opt, v)
    eng.setTriangles(scenes.heightfield(args.grid))
    eng.buildBvh()
    eng.setMaxDepth(args.depth)
    eng.setCamera(**scenes.BENCH_CAMERA)
    if args.replay:
        eng._check(eng._lib.rt_test_dump_rays(eng._h, 40 * args.nx * args.ny * args.spp // 10), "rt_test_dump_rays")
    if args.wave_times:
        nw = 256 * 32
        wt = np.zeros(3 * nw, dtype=np.uint64)
        eng._check(eng._lib.rt_test_wave_times(eng._h, wt.ctypes.data_as(C.c_void_p), nw), "rt_test_wave_times")
        for _ in range(2):
            if args.primary_only:
                eng.tracePrimary()
            else:
                eng.render()
        eng._check(eng._lib.rt_test_wave_times(eng._h, wt.ctypes.data_as(C.c_void_p), nw), "rt_test_wave_times")
        w = wt.reshape(-1, 3)
        w = w[w[:, 1] > 0].astype(np.float64)
        t0, t1 = w[:, 0].min(), w[:, 1].max()
        span = t1 - t0
        ends = np.sort((w[:, 1] - t0) / span)
        starts = np.sort((w[:, 0] - t0) / span)
        print(json.dumps({"kind": "wave_times", "waves": int(w.shape[0]), "span_ms": span / 1e5, "kernel_ms": eng.last_stats.kernel_ms,
                          "start_quantiles": [float(np.quantile(starts, q)) for q in (0.5, 0.9, 0.99, 1.0)],
                          "end_quantiles": {str(q): float(np.quantile(ends, q)) for q in (0.0, 0.01, 0.1, 0.25, 0.5, 0.75, 0.9, 0.99)},
                          "dry_quantiles": {str(q): float(np.quantile((w[:, 2] - t0) / span, q)) for q in (0.0, 0.1, 0.5, 0.9, 1.0)},
                          "mean_running_share": float(((w[:, 1] - w[:, 0]) / span).mean())}), flush=True)
        return
    eng.render()
    if args.replay:
        rays, render_ms = int(eng.last_stats.rays), eng.last_stats.kernel_ms
        for order in sorted(args.sort):  # 0 first: sorting rewrites the recorded rays in place
            if order:
                eng._check(eng._lib.rt_test_sort_dumped(eng._h, order), "rt_test_sort_dumped")
            for thr, qw in [(int(x), w) for x in args.replay.split(",") for w in args.queue_waves]:
                eng.setOption(E.OPT_QUEUE_THRESHOLD, thr)
                eng.setOption(E.OPT_QUEUE_WAVES, qw)
                for _ in range(2):
                    eng._check(eng._lib.rt_test_trace_dumped(eng._h, C.byref(eng.last_stats)), "rt_test_trace_dumped")
                print(json.dumps({"kind": "replay", "order": ["wave start order", "origin cell, octant", "octant, origin cell"][order],
                                  "threshold": thr, "queue_waves": qw, "rays": int(eng.last_stats.rays), "render_rays": rays,
                                  "spp": args.spp, "render_ms_stats_build": render_ms, "replay_ms": eng.last_stats.kernel_ms,
                                  "mrays_s": eng.last_stats.rays / eng.last_stats.kernel_ms / 1e3}), flush=True)
        eng._check(eng._lib.rt_test_dump_rays(eng._h, 0), "rt_test_dump_rays")
        return
    from tools import profile_hash

    d = stats(eng)
    d.update(kind="render", kernel_ms=eng.last_stats.kernel_ms, rays=int(eng.last_stats.rays))
    d["node_steps_per_ray"] = d["node_lanes"] / d["rays"]
    d["leaf_tests_per_ray"] = d["leaf_lanes"] / d["rays"]
    # what bench.py's roofline object reads: per-ray work of the device tree, keyed by workload and source hash
    d["workload"] = f"g{args.grid}_{args.nx}x{args.ny}x{args.spp}_d{args.depth}"
    d["node_visits_per_ray"] = d["node_steps_per_ray"]
    d["prim_tests_per_ray"] = d["leaf_prims"] / d["rays"]
    d["source_hash"] = profile_hash.kernel_source_hash()
    print(json.dumps(d), flush=True)
    eng.tracePrimary()
    d = stats(eng)
    d.update(kind="primary", kernel_ms=eng.last_stats.kernel_ms, rays=int(eng.last_stats.rays))
    d["node_steps_per_ray"] = d["node_lanes"] / d["rays"]
    d["leaf_tests_per_ray"] = d["leaf_lanes"] / d["rays"]
    print(json.dumps(d), flush=True)


if __name__ == "__main__":
    main()
