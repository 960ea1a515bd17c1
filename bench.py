#!/usr/bin/env python3
"""bench.py -- Mrays/s of the trace loop on the 1M-triangle scene (BASELINE.json metric), one process per GPU.

A "step" is one frame: one pass of the hot path (camera rays -> BVH traversal -> hit -> scatter -> accumulate)
over every pixel this rank owns.  N=1 workload = BASELINE.json configs[2]: synthetic 1 002 528(+2)-triangle
heightfield, diffuse path tracing, 1920x1080, 64 spp, max depth 8.  For N>1 the frame grows with N at fixed
aspect and spp (weak scaling: per-GPU pixel count fixed), tiles are interleaved over ranks, and the frame is
assembled on rank 0 with one RCCL reduce (each rank's frame is zero outside its tiles, so the sum is exact), which
runs while the next frame renders into a second buffer.

`python bench.py --gpus N` works as typed: a process started without torch.distributed's environment starts the N
ranks itself (python -m torch.distributed.run, 127.0.0.1) BEFORE anything touches a GPU and relays rank 0's line.
--strong is BASELINE.json configs[3]: the same scene at a FIXED 3840x2160, 1024 spp, tiles interleaved over the ranks.

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
    """The host CPU as the box names it.  Where the hypervisor masks the part number ("AMD EPYC" and no more) the
    family / model / stepping numbers and the cache size that /proc/cpuinfo still shows are appended."""
    info = {}
    try:
        for line in open("/proc/cpuinfo"):
            if ":" in line:
                k, v = line.split(":", 1)
                k = k.strip()
                if k in ("model name", "cpu family", "model", "stepping", "cache size") and k not in info:
                    info[k] = v.strip()
            elif info:
                break  # end of the first processor's block
    except OSError:
        return "unknown"
    name = info.get("model name", "unknown")
    if not any(ch.isdigit() for ch in name):
        name += " (part number masked by the host; family {} model {} stepping {}, cache {})".format(
            info.get("cpu family", "?"), info.get("model", "?"), info.get("stepping", "?"), info.get("cache size", "?"))
    return name


def usable_cores():
    try:
        return len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return os.cpu_count() or 1


def cpu_baseline_and_bytes(tris, nodes, cam_args, nx, ny, ns, depth, stride, threads):
    """Rank 0, N=1 only.  Times the CPU oracle (the reference's visit pattern: full DFS, no early-out, pairwise
    merge -- the reference has no CPU path) on a bounded pixel lattice of the same workload, on all the
    cores this process may use and on one thread (a 16x sparser lattice), and counts the box / triangle tests of an
    ordered, t-shrinking traversal of the same binary tree on the same rays, which define the algorithmic bytes per
    ray B = 32*N_box + 48*N_tri + 36.  Also returns the oracle's image on that lattice: the checker
    of the frame the GPU has just been timed on ("lattice_parity" in the line)."""
    from tests import oracle_lib  # the oracle is the checker / reported baseline here, never the product path

    o = oracle_lib.load()
    mats = o.material_table()
    cam = o.camera(cam_args["lookfrom"], cam_args["lookat"], 40.0, nx / ny, cam_args["aperture"], cam_args["dist_to_focus"])
    lat = (stride // 2, stride // 2, stride, stride)
    t0 = time.perf_counter()
    fb_ord, c_ord = o.render(tris, nodes, mats, cam, nx, ny, ns, depth, mode=1, threads=threads, lattice=lat)
    t1 = time.perf_counter()
    fb_ref, c_ref = o.render(tris, nodes, mats, cam, nx, ny, ns, depth, mode=0, threads=threads, lattice=lat)
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
        "cores": threads,  # the threads the timed sample ran on (the GPU box grants one command 16 cores' worth)
        "threads": threads,
        "box_cores": os.cpu_count(),          # logical CPUs of the whole machine
        "usable_cores": usable_cores(),       # CPUs this process may run on
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
    sub = (slice(lat[1], None, lat[3]), slice(lat[0], None, lat[2]))
    assert fb_ord[sub].tobytes() == fb_ref[sub].tobytes()  # the oracle's two traversals give one image
    return base, bytes_per_ray, fb_ref[sub].copy()


PROFILE_ROUND = "r03"  # profiles/<round>_pmc_summary[_<workload>].json, <round>_lane_occupancy.jsonl: tools/profile_round.sh
SIMDS = 1024           # 256 CUs x 4
PEAK_CLOCK_HZ = 2.4e9  # MI355X_MICROARCH.md, chip-level parameters
VALU_ISSUE_CYCLES = 2  # cycles a SIMD needs per wave64 VALU instruction (same guide, wave scheduling)


def committed_profile(workload_key):
    """What the committed rocprofv3 passes say about the render kernel ON THIS WORKLOAD (tools/pmc_passes.sh ->
    tools/pmc_summary.py; tools/trav_stats.py for the analysis build's per-ray counts): profiles/<round>_pmc_summary_
    <workload>.json when there is one, else profiles/<round>_pmc_summary.json (the bench workload).  These are NOT
    measured in this run: the record carries the hash of the kernel sources it was taken from, and `profile_stale`
    says when they have changed since."""
    from tools import profile_hash

    name = f"{PROFILE_ROUND}_pmc_summary_{workload_key}.json"
    if not os.path.exists(os.path.join(ROOT, "profiles", name)):
        name = f"{PROFILE_ROUND}_pmc_summary.json"
    out = {"source": f"profiles/{name}", "profile_stale": True}
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
        out["profile_stale"] = d.get("_meta", {}).get("source_hash") != profile_hash.kernel_source_hash()
        k = [x for x in d if "k_persistent<0," in x][0]  # the render instance that ran
        out["kernel"] = k.replace("void rtamd::", "")
        v = {c: x["per_dispatch"] for c, x in d[k].items()}
        out["valu_insts_per_launch"] = v["SQ_INSTS_VALU"]
        out["cycles_per_launch"] = v["GRBM_GUI_ACTIVE"] / 8.0  # the counter sums the 8 XCDs
        out["lane_utilisation"] = v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU"] * 64.0)
        # SQ_LEVEL_WAVES reads 0 on gfx950 / ROCm 7.2: occupancy from wave-cycles (quad-cycles) over SIMD-cycles
        out["waves_per_simd"] = v["SQ_WAVE_CYCLES"] * 4.0 / (out["cycles_per_launch"] * SIMDS)
        if "TCC_HIT_sum" in v:
            out["l2_hit_rate"] = v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
        # gfx950: FETCH_SIZE (KB) counts half the bytes of 16-B-per-lane loads; WRITE_SIZE is exact.  These are the L2s'
        # memory-side (fabric) requests: Infinity-Cache hits are INCLUDED, so this is an upper bound on HBM traffic.
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            out["fabric_bytes_per_launch"] = (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0
            out["fabric_write_bytes_per_launch"] = v["WRITE_SIZE"] * 1024.0
        # the vector-memory data path of a CU (TA -> TCP = L1 -> TD): busy shares, and L1 tag accesses per cycle and CU
        if "TCP_TOTAL_CACHE_ACCESSES_sum" in v:
            out["l1_accesses_per_launch"] = v["TCP_TOTAL_CACHE_ACCESSES_sum"]
            out["l1_accesses_per_cycle_per_cu"] = v["TCP_TOTAL_CACHE_ACCESSES_sum"] / 256.0 / out["cycles_per_launch"]
            out["l1_hit_rate"] = 1.0 - v["TCP_TCC_READ_REQ_sum"] / v["TCP_TOTAL_CACHE_ACCESSES_sum"]
        if "TD_TD_BUSY_sum" in v:
            out["td_busy_frac"] = v["TD_TD_BUSY_sum"] / 256.0 / out["cycles_per_launch"]
            out["ta_busy_frac"] = v["TA_TA_BUSY_sum"] / 256.0 / out["cycles_per_launch"]
        out["workload"] = d.get("_meta", {}).get("workload")
    except Exception as e:  # noqa: BLE001
        out["error"] = f"{type(e).__name__}: {e}"
    try:
        for line in open(os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_lane_occupancy.jsonl")):
            t = json.loads(line)
            if t.get("workload") == workload_key and "node_visits_per_ray" in t:
                out["node_visits_per_ray"] = t["node_visits_per_ray"]
                out["prim_tests_per_ray"] = t["prim_tests_per_ray"]
    except Exception:  # noqa: BLE001
        pass
    return out


def roofline_of(prof, workload_key, world, kernel_s, rays_per_launch, bytes_per_ray):
    """The `roofline` object of the line.  Three units are priced, each from the committed counter pass of THESE sources
    on THIS workload over this run's own kernel time (no fraction is claimed from counters of other code or another
    scene), and the headline `bound` is the one the evidence says binds:
      valu_issue  wave64 vector instructions per launch against 1024 SIMDs x 2.4 GHz / 2 cycles -- what binds the
                  bench frame: the A/B of node formats (profiles/r03_ab_node_format*) took 19 % of the L1 accesses away
                  for 6 % more vector instructions and the frame got 1.5 % SLOWER;
      hbm         the L2s' fabric traffic against 8 TB/s (an upper bound on HBM bytes: Infinity-Cache hits included)
                  -- what binds scenes far beyond the caches (BASELINE config 5);
      vector_l1   L1 tag accesses x 64 B against one access per cycle and CU -- near its ceiling on the bench frame and,
                  by the same A/B, not what the frame waits for; reported with the share of each touched line a lane
                  uses (16 of 128 bytes), which is where the waste is."""
    same_workload = prof.get("workload") == workload_key and world == 1 and not prof.get("profile_stale", True)
    roof = {"kernel": prof.get("kernel", "k_persistent<0, 1, 1, 6, false, true>") + " (render: quantized 4-wide nodes, speculative traversal, rounds)",
            "kernel_ms": kernel_s * 1e3, "rays_per_launch": rays_per_launch, "committed_profile": prof}
    L1_PEAK_GBS = 256 * 64 * PEAK_CLOCK_HZ / 1e9
    mem = {"hbm_peak": HBM_PEAK_GBS, "unit": "GB/s"}
    if "node_visits_per_ray" in prof:
        dev_bytes = 64.0 * prof["node_visits_per_ray"] + 48.0 * prof["prim_tests_per_ray"] + 36.0
        mem.update({"device_bytes_per_ray": dev_bytes, "device_bytes_rate_gbs": dev_bytes * rays_per_launch / kernel_s / 1e9,
                    "device_bytes_note": "64 B per wide-node visit + 48 B per primitive test + 28 B ray in + 8 B hit out, counts from the analysis build",
                    "node_visits_per_ray": prof["node_visits_per_ray"], "prim_tests_per_ray": prof["prim_tests_per_ray"]})
    if bytes_per_ray is not None:
        mem["algorithmic_bytes_per_ray_binary_tree"] = bytes_per_ray
        mem["binary_tree_rate_gbs"] = bytes_per_ray * rays_per_launch / kernel_s / 1e9
        mem["binary_tree_note"] = "SURVEY 8d's model (32 B per box test of an ordered walk of the reference's BINARY tree); it does not describe this kernel, which visits ~16 wide nodes per ray, not ~570 boxes"
    roof["hbm"] = mem
    roof["traffic"] = None
    if not (same_workload and "valu_insts_per_launch" in prof):
        roof.update({"bound": "valu_issue", "achieved": None, "peak": SIMDS * PEAK_CLOCK_HZ / VALU_ISSUE_CYCLES / 1e9,
                     "unit": "G wave64-inst/s", "frac": None, "profile_stale": bool(prof.get("profile_stale", True)),
                     "note": "no committed counter pass of these sources for this workload / rank count (profile_stale: the "
                             "kernel sources changed since the pass in profiles/ was taken): no fraction is claimed"})
        return roof
    achieved = prof["valu_insts_per_launch"] / kernel_s / 1e9
    peak = SIMDS * PEAK_CLOCK_HZ / VALU_ISSUE_CYCLES / 1e9
    valu = {"achieved": achieved, "peak": peak, "unit": "G wave64-inst/s", "frac": achieved / peak,
            "basis": "SQ_INSTS_VALU per launch / kernel time; 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction",
            "lane_utilisation": prof.get("lane_utilisation"),
            "useful_lane_frac": achieved / peak * prof.get("lane_utilisation", 0.0),
            "waves_per_simd": prof.get("waves_per_simd")}
    # against what a SIMD was MEASURED to issue (tools/micro/valu_rate.hip, 8 waves per SIMD): independent v_fma_f32, and
    # the node test's own instruction mix -- the practical ceilings under the 2-cycle specification
    try:
        for line in open(os.path.join(ROOT, "profiles", "r02_valu_issue_rate.jsonl")):
            t = json.loads(line)
            if t.get("waves_per_simd") == 8 and t.get("kind") in ("v_fma_f32", "node-test mix"):
                key = "frac_of_measured_fma_rate" if t["kind"] == "v_fma_f32" else "frac_of_measured_node_mix_rate"
                valu[key] = achieved / (SIMDS * PEAK_CLOCK_HZ / t["cycles_per_wave64_inst"] / 1e9)
    except Exception:  # noqa: BLE001
        pass
    roof["valu_issue"] = valu
    if "l1_accesses_per_launch" in prof:
        a = prof["l1_accesses_per_launch"] * 64.0 / kernel_s / 1e9
        roof["vector_l1"] = {
            "achieved": a, "peak": L1_PEAK_GBS, "unit": "GB/s", "frac": a / L1_PEAK_GBS,
            "basis": "TCP_TOTAL_CACHE_ACCESSES x 64 B per launch / kernel time; peak = one L1 line access per cycle and CU x 256 CUs "
                     "x 2.4 GHz, the ceiling tools/micro/l1_rate.hip measures (profiles/r02_l1_rate.jsonl).  A tag-rate occupancy, "
                     "not a data rate: each access hands ONE lane 16 bytes",
            "data_delivered_gbs": prof["l1_accesses_per_launch"] * 16.0 / kernel_s / 1e9,
            "useful_l1_bytes_frac": 16.0 / 128.0,  # bytes a lane takes from each 128-byte line it touches
            "binds": False,
            "binds_evidence": "RT_OPT_NODE_FORMAT 2 (three loads per node visit): L1 accesses -19 %, vector instructions +6 %, frame "
                              "+1.5 % (profiles/r03_ab_node_format.log, r03_ab_node_format{1,2}_pmc_summary.json)",
            "accesses_per_cycle_per_cu_at_measured_clock": prof.get("l1_accesses_per_cycle_per_cu"),
            "td_busy_frac": prof.get("td_busy_frac"), "ta_busy_frac": prof.get("ta_busy_frac"),
            "l1_hit_rate": prof.get("l1_hit_rate"), "l2_hit_rate": prof.get("l2_hit_rate")}
    hbm_frac = None
    if "fabric_bytes_per_launch" in prof:
        mem["fabric_traffic_bytes_per_launch"] = prof["fabric_bytes_per_launch"]
        mem["fabric_traffic_gbs"] = prof["fabric_bytes_per_launch"] / kernel_s / 1e9
        hbm_frac = mem["fabric_traffic_gbs"] / HBM_PEAK_GBS
        mem["frac"] = hbm_frac
        mem["frac_note"] = "(2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction) / kernel time / 8 TB/s: the L2s' fabric requests, Infinity-Cache hits included -- an UPPER bound on the HBM fraction"
        if "device_bytes_per_ray" in mem:  # > 1: lines fetched and not used; < 1: the caches absorb the rest
            mem["traffic_over_device_bytes"] = prof["fabric_bytes_per_launch"] / (mem["device_bytes_per_ray"] * rays_per_launch)
        roof["traffic"] = prof["fabric_bytes_per_launch"]
    # the headline: HBM where the scene streams from it, vector-instruction issue otherwise (vector_l1: see above)
    v_frac = valu.get("frac_of_measured_node_mix_rate", valu["frac"])
    if hbm_frac is not None and hbm_frac >= v_frac:
        roof.update({"bound": "hbm", "achieved": mem["fabric_traffic_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_frac})
    else:
        roof.update({"bound": "valu_issue", "achieved": valu["achieved"], "peak": valu["peak"], "unit": valu["unit"], "frac": valu["frac"]})
    return roof


def spawn_ranks(n):
    """`python bench.py --gpus N` typed by hand (or by the driver): start the N ranks as fresh child processes and relay
    their output.  Runs before this process has imported torch or touched a GPU, and never replaces this process."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--grid", type=int, default=708, help="heightfield cells per side (708 -> 1 002 528 triangles)")
    ap.add_argument("--nx", type=int, default=None)
    ap.add_argument("--ny", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--leaf", type=int, default=None)
    ap.add_argument("--box-mode", type=int, default=None)
    ap.add_argument("--kernel", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-stride", type=int, default=8, help="CPU baseline lattice: every n-th pixel of every n-th row")
    ap.add_argument("--strong", action="store_true",
                    help="BASELINE.json configs[3]: a FIXED 3840x2160 frame at 1024 spp, tiles interleaved over the ranks "
                         "(strong scaling; --nx/--ny/--spp override the size)")
    ap.add_argument("--sample-shard", action="store_true",
                    help="N > 1: split the SAMPLES of one fixed-size frame over ranks (salted streams, summed with one "
                         "all-reduce; SURVEY 8e's non-parity mode, strong scaling) instead of the tiles of a growing frame")
    ap.add_argument("--share-scene", choices=["auto", "on", "off"], default="auto",
                    help="N > 1: rank 0 builds the scene once and the other ranks import its device layout from a shared "
                         "mapping in /dev/shm (rt_export_scene / rt_import_scene) instead of every rank generating, building "
                         "and laying out its own copy; auto = for scenes of more than 4 M triangles")
    ap.add_argument("--no-verify", action="store_true",
                    help="N > 1, tiles: skip rank 0's untimed whole-frame render that the assembled frame is compared with")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the trace loop has no CPU fallback")
    # RT_BENCH_REHEARSE=1: run the N > 1 code path on a box with ONE GPU (every rank on device 0, frame assembly over
    # gloo).  For checking the multi-rank plumbing only; the numbers mean nothing.
    rehearse = os.environ.get("RT_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: local rank {local_rank} but this node shows {torch.cuda.device_count()} GPU(s)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if rehearse:
            dist.init_process_group(backend="gloo")
        else:
            n_dev = torch.cuda.device_count()
            if n_dev < world or local_rank >= n_dev:
                raise SystemExit(f"bench.py --gpus {world}: this node shows {n_dev} GPU(s); one process per GPU needs {world} "
                                 f"(local rank {local_rank}).  RT_BENCH_REHEARSE=1 rehearses the plumbing on one GPU over gloo.")
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    import rt_amd
    from rt_amd import engine as E
    from rt_amd import scenes, sharding

    base_nx = args.nx or (3840 if args.strong else 1920)
    base_ny = args.ny or (2160 if args.strong else 1080)
    spp = args.spp or (1024 if args.strong else 64)
    sample_shard = args.sample_shard and world > 1
    fixed_frame = args.strong or sample_shard
    nx, ny = sharding.frame_size(base_nx, base_ny, world, fixed_frame)  # weak scaling: pixels grow with N at fixed aspect
    n_tris = 2 * args.grid * args.grid + 2
    share = world > 1 and (args.share_scene == "on" or (args.share_scene == "auto" and n_tris > 4_000_000))
    ns_local, salt = E.sample_shard(rank, world, spp) if sample_shard else (spp, 0)
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
    tris = None
    t0 = time.perf_counter()
    if not share or rank == 0:
        tris = scenes.heightfield(args.grid)
        assert tris.shape[0] == n_tris
        eng.setTriangles(tris)
        eng.buildBvh()
    if share:
        # One build per node: rank 0 writes the device layout into a file in /dev/shm (host memory, mapped by every
        # rank), the others upload straight from the mapping.  BASELINE config 5 (100 M triangles) would otherwise cost
        # every rank 14.8 GB of records, a 47 s all-core build and ~100 GB of host memory.
        path = f"/dev/shm/rt_amd_scene_{os.environ.get('MASTER_PORT', '0')}_{args.grid}.bin"
        sharding.share_through_file(dist, rank, path, lambda: eng.exportSceneSize(0), lambda mm: eng.exportScene(0, out=mm),
                                    eng.importScene, size_device="cpu" if rehearse else f"cuda:{local_rank}")
    build_s = time.perf_counter() - t0
    eng.setMaxDepth(args.depth)
    eng.setCamera(**scenes.BENCH_CAMERA)
    if not sample_shard:
        eng.setShard(rank, world)
    kk = float(np.float32(1.0 / float(np.float32(spp))))
    # Two frame buffers: frame k is assembled on rank 0 (RCCL's stream) while frame k + 1 renders (the library's
    # stream), so the exchange costs the job one reduce at the end, not one per frame (rt_amd/sharding.py, which
    # tests/test_sharding_gloo.py runs over gloo with 2 and 4 ranks).
    fbs = [torch.zeros((ny, nx, 3), dtype=torch.float32, device=f"cuda:{local_rank}") for _ in range(2 if world > 1 else 1)]

    def render(fb):
        st = eng.renderDevice(fb.data_ptr())  # returns when the frame is in fb
        return st.rays, st.kernel_ms

    loop = sharding.FrameLoop(dist, world, rank, fbs, render, torch.cuda.synchronize, sample_shard=sample_shard,
                              finish_sample_shards=lambda fb: fb.mul_(kk).clamp_(max=1.0))
    elapsed, rays_local, kernel_ms = loop.run(args.warmup, args.steps)
    fb = loop.last_frame()
    elapsed, rays_total = sharding.job_totals(dist, world, elapsed, rays_local, "cpu" if rehearse else fb.device)

    assembled_ok = None
    failed = None
    if world > 1 and rank == 0 and not sample_shard and not args.no_verify:
        whole = torch.zeros_like(fb)  # untimed: rank 0 renders the whole frame alone
        eng.setShard(0, 1)
        eng.renderDevice(whole.data_ptr())
        eng.setShard(rank, world)
        assembled_ok = bool(torch.equal(whole, fb))
        del whole

    if rank == 0:
        value = rays_total / elapsed / 1e6
        rays_per_launch = rays_local / args.steps
        kernel_s = float(np.mean(kernel_ms)) / 1e3
        frame_cpu = fb.cpu().numpy() if (world == 1 and not args.no_cpu_baseline) else None
        # primary-only rate on the same scene (not the headline; reported beside it)
        eng.setShard(0, 1)
        eng.setNx(base_nx), eng.setNy(base_ny)
        eng.tracePrimary()
        eng.tracePrimary()
        prim = {"mrays_s": eng.last_stats.rays / (eng.last_stats.kernel_ms / 1e3) / 1e6, "kernel_ms": eng.last_stats.kernel_ms,
                "frame": f"{base_nx}x{base_ny}x1"}
        cpu, bytes_per_ray, lattice_parity = None, None, None
        if frame_cpu is not None:
            threads = max(1, min(16, usable_cores()))
            # The CPU leg is a bounded sample of the SAME workload -- the frame's own sample count on a pixel lattice --
            # so that the frame the GPU was timed on can be compared with the oracle's image on that lattice.  The
            # lattice is thinned to keep the sample at about 32 400 pixels x 64 spp worth of paths: every 8th pixel at
            # 1080p x 64 spp, every 16th at 2160p x 64 spp, every 64th at 2160p x 1024 spp (config 4: 2 040 pixels).
            st_ = args.cpu_stride * (2 if nx * ny > 1920 * 1080 else 1)
            st_ = int(round(st_ * math.sqrt(max(spp, 64) / 64.0)))
            cpu, bytes_per_ray, lat_img = cpu_baseline_and_bytes(tris, eng.getBvhNodes(), scenes.BENCH_CAMERA, nx, ny, spp,
                                                                 args.depth, st_, threads)
            lattice_parity = bool(frame_cpu[st_ // 2::st_, st_ // 2::st_].tobytes() == lat_img.tobytes())
        workload_key = f"g{args.grid}_{nx}x{ny}x{spp}_d{args.depth}"
        prof = committed_profile(workload_key)
        roof = roofline_of(prof, workload_key, world, kernel_s, rays_per_launch, bytes_per_ray)
        line = {
            "metric": "Mrays/s (node) on 1M-tri BVH; ms/frame at 1080p",
            "value": value,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if fixed_frame else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if not rehearse else "synthetic (REHEARSAL: all ranks on one GPU, gloo; not a measurement)",
            "config": {"workload": f"heightfield G={args.grid} ({n_tris} triangles), diffuse path tracing, "
                                   f"{nx}x{ny}, {spp} spp, max depth {args.depth}",
                       "triangles": n_tris, "frame": f"{nx}x{ny}", "spp": spp, "max_depth": args.depth,
                       "scene_shared_from_rank0": bool(share),
                       "parallelism": (f"sample-pass x{world} (salted streams; not the parity path)" if sample_shard
                                       else f"tile-interleaved x{world}"), "bvh_build_s": build_s,
                       "wide_nodes": int(eng.last_stats.wide_nodes), "lds_stack_entries": int(eng.last_stats.stack_entries)},
            "roofline": roof,
            "cpu_baseline": cpu,
            "primary": prim,
        }
        if lattice_parity is not None:
            line["lattice_parity"] = lattice_parity  # the timed frame's lattice == the oracle's image, byte for byte
        if assembled_ok is not None:
            line["assembled_frame_equals_single_gpu_frame"] = assembled_ok
        print(json.dumps(line), flush=True)
        # a frame that differs from the oracle's, or an assembled frame that differs from the single-GPU frame, is a
        # failed run whatever its rate: the line is printed for the record and the process exits non-zero
        if lattice_parity is False:
            failed = "the timed frame differs from the oracle's image on the lattice (lattice_parity false)"
        elif assembled_ok is False:
            failed = "the frame assembled from the ranks' tiles differs from the frame one GPU renders alone"
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if failed:
        print("bench.py: PARITY FAILURE: " + failed, file=sys.stderr, flush=True)
        raise SystemExit(3)


if __name__ == "__main__":
    main()
