#!/usr/bin/env python3
"""bench.py -- Mrays/s of the trace loop on the 1M-triangle scene (BASELINE.json metric), one process per GPU.

A "step" is one frame: one pass of the hot path (camera rays -> BVH traversal -> hit -> scatter -> accumulate)
over every pixel this rank owns.  N=1 workload = BASELINE.json configs[2]: synthetic 1 002 528(+2)-triangle
heightfield, diffuse path tracing, 1920x1080, 64 spp, max depth 8.  For N>1 the frame grows with N at fixed
aspect and spp (weak scaling: per-GPU pixel count fixed), tiles are interleaved over ranks, and the frame is
assembled on rank 0 with one RCCL reduce (each rank's frame is zero outside its tiles, so the sum is exact), which
runs while the next frame renders into a second buffer.

rays = closest-hit queries (hit() calls, reference pbrt_engine.cu:371), counted on the device.
Scene, BVH and camera are resident in HBM before the timed region; the frame stays on the device.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_and_bytes(tris, nodes, cam_args, nx, ny, ns, depth, stride, threads):
    """Rank 0, N=1 only.  Times the CPU oracle (the reference's visit pattern: full DFS, no early-out, pairwise
    merge -- the reference has no CPU path) on a bounded pixel lattice of the same workload, on all the
    cores this process may use and on one thread (a 16x sparser lattice), and counts the box / triangle tests of an
    ordered, t-shrinking traversal of the same binary tree on the same rays, which define the algorithmic bytes per
    ray B = 32*N_box + 48*N_tri + 36."""
    from tests import oracle_lib  # the oracle is the checker / reported baseline here, never the product path

    o = oracle_lib.load()
    mats = o.material_table()
    cam = o.camera(cam_args["lookfrom"], cam_args["lookat"], 40.0, nx / ny, cam_args["aperture"], cam_args["dist_to_focus"])
    lat = (stride // 2, stride // 2, stride, stride)
    t0 = time.perf_counter()
    _, c_ord = o.render(tris, nodes, mats, cam, nx, ny, ns, depth, mode=1, threads=threads, lattice=lat)
    t1 = time.perf_counter()
    _, c_ref = o.render(tris, nodes, mats, cam, nx, ny, ns, depth, mode=0, threads=threads, lattice=lat)
    t2 = time.perf_counter()
    s1 = 4 * stride
    _, c_one = o.render(tris, nodes, mats, cam, nx, ny, ns, depth, mode=0, threads=1, lattice=(s1 // 2, s1 // 2, s1, s1))
    t3 = time.perf_counter()
    assert c_ord.rays == c_ref.rays
    n_box = c_ord.box_tests / c_ord.rays
    n_tri = c_ord.tri_tests / c_ord.rays
    bytes_per_ray = 32.0 * n_box + 48.0 * n_tri + 36.0
    base = {
        "value": c_ref.rays / (t2 - t1) / 1e6,
        "unit": "Mrays/s",
        "cores": threads,
        "kind": "port",
        "sample": f"every {stride}th pixel of every {stride}th row of the same {nx}x{ny} frame at {ns} spp, depth {depth} "
                  f"({c_ref.rays} rays), reference visit pattern ({c_ref.box_tests / c_ref.rays:.0f} box + "
                  f"{c_ref.tri_tests / c_ref.rays:.1f} tri tests/ray), {t2 - t1:.1f} s",
        "cpu_model": cpu_model(),
        "one_thread_value": c_one.rays / (t3 - t2) / 1e6,
        "one_thread_sample": f"every {s1}th pixel of every {s1}th row ({c_one.rays} rays), {t3 - t2:.1f} s",
        "ordered_value": c_ord.rays / (t1 - t0) / 1e6,
        "ordered_tests_per_ray": {"box": n_box, "tri": n_tri},
    }
    return base, bytes_per_ray


def valu_issue(kernel_s):
    """VALU issue occupancy of the render kernel from the committed counter passes (profiles/r01_pmc_summary.json,
    tools/pmc_passes.sh): wave64 VALU instructions x 4 cycles over the cycles of 1024 SIMDs.  Not measured live."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_summary.json")))
        k = [x for x in d if "k_persistent<0, 1, 1, 6" in x][0]
        v = {c: x["per_dispatch"] for c, x in d[k].items()}
        cycles = v["GRBM_GUI_ACTIVE"] / 8.0  # summed over 8 XCDs
        return {"source": "profiles/r01_pmc_summary.json", "valu_insts_per_launch": v["SQ_INSTS_VALU"],
                # 4 cycles per wave64 instruction; quarter-rate ones (rcp, sqrt, f64 divide steps) take longer, so the
                # estimate can pass 1: read it as "the VALU pipe never idles"
                "busy_frac": v["SQ_INSTS_VALU"] / 1024.0 * 4.0 / cycles,
                # cycles a SIMD spends per wave64 VALU instruction; tools/micro/valu_rate.hip reaches 3.0-3.4 on the
                # node test's instruction mix with no dependencies (profiles/r01_valu_issue_rate.jsonl)
                "cycles_per_inst_per_simd": cycles * 1024.0 / v["SQ_INSTS_VALU"],
                "lane_utilisation": v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU"] * 64.0)}
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--grid", type=int, default=708, help="heightfield cells per side (708 -> 1 002 528 triangles)")
    ap.add_argument("--nx", type=int, default=1920)
    ap.add_argument("--ny", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--leaf", type=int, default=None)
    ap.add_argument("--box-mode", type=int, default=None)
    ap.add_argument("--kernel", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-stride", type=int, default=8, help="CPU baseline lattice: every n-th pixel of every n-th row")
    ap.add_argument("--sample-shard", action="store_true",
                    help="N > 1: split the SAMPLES of one fixed-size frame over ranks (salted streams, summed with one "
                         "all-reduce; SURVEY 8e's non-parity mode, strong scaling) instead of the tiles of a growing frame")
    ap.add_argument("--verify", action="store_true",
                    help="after timing, rank 0 renders the whole frame alone and compares it with the assembled one")
    args = ap.parse_args()

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the trace loop has no CPU fallback")
    # RT_BENCH_REHEARSE=1: run the N > 1 code path on a box with ONE GPU (every rank on device 0, frame assembly over
    # gloo).  For checking the multi-rank plumbing only; the numbers mean nothing.
    rehearse = os.environ.get("RT_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if rehearse:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    import rt_amd
    from rt_amd import engine as E
    from rt_amd import scenes

    # weak scaling: pixels grow with N at fixed aspect
    sample_shard = args.sample_shard and world > 1
    s = 1.0 if sample_shard else math.sqrt(world)
    nx, ny = int(round(args.nx * s)), int(round(args.ny * s))
    tris = scenes.heightfield(args.grid)
    ns_local, salt = E.sample_shard(rank, world, args.spp) if sample_shard else (args.spp, 0)
    eng = rt_amd.PbrtEngine(nx, ny, ns_local, device=local_rank)
    if sample_shard:
        eng.setOption(E.OPT_SEED_SALT, salt)
        eng.setOption(E.OPT_RAW_SUM, 1)
    if args.leaf is not None:
        eng.setOption(E.OPT_LEAF_SIZE, args.leaf)
    if args.box_mode is not None:
        eng.setOption(E.OPT_BOX_MODE, args.box_mode)
    if args.kernel is not None:
        eng.setOption(E.OPT_KERNEL, args.kernel)
    eng.setTriangles(tris)
    t0 = time.perf_counter()
    eng.buildBvh()
    build_s = time.perf_counter() - t0
    eng.setMaxDepth(args.depth)
    eng.setCamera(**scenes.BENCH_CAMERA)
    if not sample_shard:
        eng.setShard(rank, world)
    kk = float(np.float32(1.0 / float(np.float32(args.spp))))
    # Two frame buffers: frame k is assembled on rank 0 (RCCL's stream) while frame k + 1 renders (the library's
    # stream), so the exchange costs the job one reduce at the end, not one per frame.
    fbs = [torch.zeros((ny, nx, 3), dtype=torch.float32, device=f"cuda:{local_rank}") for _ in range(2 if world > 1 else 1)]
    pending = [None, None]
    frame_no = [0]

    def finish(slot):
        """Frame assembly of the buffer in `slot` has completed (host-side): the buffer may be rendered into again."""
        if pending[slot] is not None:
            pending[slot].wait()
            torch.cuda.current_stream().synchronize()
            pending[slot] = None

    def step():
        slot = frame_no[0] % len(fbs)
        frame_no[0] += 1
        fb = fbs[slot]
        finish(slot)  # the reduce of two frames ago; the library memsets and renders on its own stream
        st = eng.renderDevice(fb.data_ptr())  # returns when the frame is in fb
        if world > 1:
            # frame assembly on rank 0 (the host that writes the PPM): x + 0 is exact (tiles) / sums of clamped samples
            # (sample shards).  A reduce moves half the bytes of an all-reduce over the xGMI ring.
            if sample_shard:
                dist.reduce(fb, dst=0)
                if rank == 0:
                    fb.mul_(kk).clamp_(max=1.0)
                torch.cuda.synchronize()
            else:
                pending[slot] = dist.reduce(fb, dst=0, async_op=True)
        return st.rays, st.kernel_ms

    def drain():
        for slot in range(len(fbs)):
            finish(slot)

    for _ in range(args.warmup):
        step()
    drain()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rays_local, kernel_ms = 0, []
    for _ in range(args.steps):
        r, k = step()
        rays_local += r
        kernel_ms.append(k)
    drain()  # every frame of the timed region is assembled on rank 0 before the clock stops
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    fb = fbs[(frame_no[0] - 1) % len(fbs)]  # the last frame

    if world > 1:
        small = "cpu" if rehearse else fb.device
        t = torch.tensor([elapsed], dtype=torch.float64, device=small)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        rr = torch.tensor([rays_local], dtype=torch.int64, device=small)
        dist.all_reduce(rr)
        rays_total = int(rr.item())
    else:
        rays_total = rays_local

    assembled_ok = None
    if args.verify and world > 1 and rank == 0 and not sample_shard:
        whole = torch.zeros_like(fb)
        eng.setShard(0, 1)
        eng.renderDevice(whole.data_ptr())
        eng.setShard(rank, world)
        assembled_ok = bool(torch.equal(whole, fb))

    if rank == 0:
        value = rays_total / elapsed / 1e6
        rays_per_launch = rays_local / args.steps
        kernel_s = float(np.mean(kernel_ms)) / 1e3
        # primary-only rate on the same scene (not the headline; reported beside it)
        eng.setShard(0, 1)
        eng.setNx(args.nx), eng.setNy(args.ny)
        eng.tracePrimary()
        eng.tracePrimary()
        prim = {"mrays_s": eng.last_stats.rays / (eng.last_stats.kernel_ms / 1e3) / 1e6, "kernel_ms": eng.last_stats.kernel_ms,
                "frame": f"{args.nx}x{args.ny}x1"}
        cpu, bytes_per_ray = None, None
        if world == 1 and not args.no_cpu_baseline:
            threads = max(1, min(16, os.cpu_count() or 1))
            cpu, bytes_per_ray = cpu_baseline_and_bytes(tris, eng.getBvhNodes(), scenes.BENCH_CAMERA, args.nx, args.ny,
                                                        args.spp, args.depth, args.cpu_stride, threads)
        roof = None
        bytes_source = "counted live by the oracle's ordered traversal on the CPU sample"
        if bytes_per_ray is None:  # N > 1 or --no-cpu-baseline: the committed N = 1 count of the same scene and camera
            try:
                committed = json.load(open(os.path.join(ROOT, "profiles", "r01_bench_line.json")))
                bytes_per_ray = committed["roofline"]["algorithmic_bytes_per_ray"]
                bytes_source = "profiles/r01_bench_line.json (N = 1 run of the same scene, camera, spp and depth)"
            except Exception:
                bytes_per_ray = None
        if bytes_per_ray is not None:
            achieved = bytes_per_ray * rays_per_launch / kernel_s / 1e9
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                try:
                    tj = json.load(open(tpath))
                    key = f"g{args.grid}_{args.nx}x{args.ny}x{args.spp}_d{args.depth}"
                    traffic = tj.get(key, {}).get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": traffic, "kernel": "k_persistent<0,1,1,6,false> (render, quantized nodes, speculative, 8 rounds)",
                    "kernel_ms": kernel_s * 1e3, "algorithmic_bytes_per_ray": bytes_per_ray, "bytes_per_ray_source": bytes_source,
                    "rays_per_launch": rays_per_launch,
                    # frac > 1 is not an error: SURVEY 8d prices every box test of a binary-tree traversal at 32 HBM
                    # bytes, while the device BVH (30 MB of nodes) is served from L2 / Infinity Cache.
                    # What binds the kernel is VALU issue; its measured share comes from the committed PMC passes.
                    "measured_hbm_gbs": (traffic / kernel_s / 1e9) if traffic else None,
                    "valu_issue": valu_issue(kernel_s)}
        line = {
            "metric": "Mrays/s (node) on 1M-tri BVH; ms/frame at 1080p",
            "value": value,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if sample_shard else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if not rehearse else "synthetic (REHEARSAL: all ranks on one GPU, gloo; not a measurement)",
            "config": {"workload": f"heightfield G={args.grid} ({tris.shape[0]} triangles), diffuse path tracing, "
                                   f"{nx}x{ny}, {args.spp} spp, max depth {args.depth}",
                       "triangles": int(tris.shape[0]), "frame": f"{nx}x{ny}", "spp": args.spp, "max_depth": args.depth,
                       "parallelism": (f"sample-pass x{world} (salted streams; not the parity path)" if sample_shard
                                       else f"tile-interleaved x{world}"), "bvh_build_s": build_s,
                       "wide_nodes": int(eng.last_stats.wide_nodes), "lds_stack_entries": int(eng.last_stats.stack_entries)},
            "roofline": roof,
            "cpu_baseline": cpu,
            "primary": prim,
        }
        if assembled_ok is not None:
            line["assembled_frame_equals_single_gpu_frame"] = assembled_ok
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
