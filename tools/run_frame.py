#!/usr/bin/env python3
"""Renders frames of the bench workload through the C ABI (no torch): the program rocprofv3 wraps for the
--kernel-trace and --pmc passes, and the driver of option sweeps.  Prints one JSON line per variant."""
import argparse
import itertools
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rt_amd  # noqa: E402
from rt_amd import engine as E  # noqa: E402
from rt_amd import scenes  # noqa: E402


def ints(s):
    return [int(x) for x in s.split(",")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=708)
    ap.add_argument("--nx", type=int, default=1920)
    ap.add_argument("--ny", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--frames", type=int, default=1)
    ap.add_argument("--primary", type=int, default=1)
    ap.add_argument("--leaf", type=ints, default=[2])
    ap.add_argument("--box-mode", type=ints, default=[0])
    ap.add_argument("--kernel", type=ints, default=[2])
    ap.add_argument("--threshold", type=ints, default=[24])
    ap.add_argument("--waves", type=ints, default=[24])
    ap.add_argument("--lds", type=ints, default=[16])
    ap.add_argument("--fmt", type=ints, default=[1])
    ap.add_argument("--batch", type=ints, default=[16])
    ap.add_argument("--occ", type=ints, default=[6])
    ap.add_argument("--topo", type=ints, default=[1])
    ap.add_argument("--shade", type=ints, default=[32])
    ap.add_argument("--swap", type=ints, default=[8])
    ap.add_argument("--tail", type=ints, default=[0])
    ap.add_argument("--taper", type=ints, default=[150])
    ap.add_argument("--collapse", type=ints, default=[1])
    ap.add_argument("--det", type=int, default=None, help="RT_OPT_DETERMINISTIC (libraries that have the option)")
    ap.add_argument("--pwaves", type=ints, default=[16], help="primary rays: waves per CU")
    ap.add_argument("--pchunk", type=ints, default=[256], help="primary rays: slots per atomic")
    args = ap.parse_args()
    tris = scenes.heightfield(args.grid)
    eng = rt_amd.PbrtEngine(args.nx, args.ny, args.spp, device=0)
    eng.setTriangles(tris)
    eng.setMaxDepth(args.depth)
    eng.setCamera(**scenes.BENCH_CAMERA)
    if args.det is not None:
        eng.setOption(26, args.det)
    nodes = None
    warmed = False
    for leaf, bm, kv, th, wv, ld, fm, lb, oc, tp, sb, wb, ts, tap, col in itertools.product(args.leaf, args.box_mode, args.kernel, args.threshold, args.waves, args.lds, args.fmt, args.batch, args.occ, args.topo, args.shade, args.swap, args.tail, args.taper, args.collapse):
        eng.setOption(E.OPT_LEAF_SIZE, leaf)
        eng.setOption(E.OPT_BOX_MODE, bm)
        eng.setOption(E.OPT_KERNEL, kv)
        eng.setOption(E.OPT_REFILL_THRESHOLD, th)
        eng.setOption(E.OPT_WAVES_PER_CU, wv)
        eng.setOption(E.OPT_LDS_STACK, ld)
        eng.setOption(E.OPT_NODE_FORMAT, fm)
        eng.setOption(E.OPT_LEAF_BATCH, lb)
        eng.setOption(E.OPT_OCCUPANCY, oc)
        eng.setOption(E.OPT_TOPOLOGY, tp)
        eng.setOption(E.OPT_SHADE_BATCH, sb)
        eng.setOption(E.OPT_SWAP_BATCH, wb)
        eng.setOption(E.OPT_TAIL_SPLIT, ts)
        eng.setOption(E.OPT_TAIL_TAPER, tap)
        eng.setOption(E.OPT_COLLAPSE, col)
        if nodes is None:
            eng.buildBvh()
            nodes = eng.getBvhNodes()
        else:
            eng.setBvhNodes(nodes)
        out = {"leaf": leaf, "box_mode": bm, "kernel": kv, "thr": th, "waves": wv, "lds": ld, "fmt": fm, "batch": lb, "occ": oc, "topo": tp, "shade": sb, "swap": wb, "tail": ts, "taper": tap, "collapse": col}
        best = None
        if args.frames and not warmed:  # the first frames of a process run slow (clocks, first touches): not measured
            eng.render()
            eng.render()
            warmed = True
        for _ in range(args.frames):
            eng.render()
            ms = eng.last_stats.kernel_ms
            best = ms if best is None else min(best, ms)
        if args.frames:
            out.update(render_ms=best, rays=eng.last_stats.rays, mrays_s=eng.last_stats.rays / best / 1e3)
        for pw, pc in itertools.product(args.pwaves, args.pchunk):
            eng.setOption(E.OPT_PRIMARY_WAVES, pw)
            eng.setOption(E.OPT_PRIMARY_CHUNK, pc)
            pbest = None
            for _ in range(args.primary):
                eng.tracePrimary()
                ms = eng.last_stats.kernel_ms
                pbest = ms if pbest is None else min(pbest, ms)
            if args.primary:
                out.update(pwaves=pw, pchunk=pc, primary_ms=pbest, primary_mrays_s=args.nx * args.ny / pbest / 1e3)
            out.update(wide_nodes=eng.last_stats.wide_nodes, stack=eng.last_stats.stack_entries)
            print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
