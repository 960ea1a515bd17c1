"""Diagnosis: the same frame rendered with different round counts, several times each; which pixels differ, and who
agrees with the oracle there.  python tools/diag_rounds.py [nx ny ns]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rt_amd  # noqa: E402
from rt_amd import engine, scenes  # noqa: E402
from tests import oracle_lib  # noqa: E402

nx, ny, ns = (int(a) for a in (sys.argv[1:4] or (3840, 2160, 1024)))
configs = [int(a) for a in (sys.argv[4].split(",") if len(sys.argv) > 4 else "0,0,1,1,8".split(","))]
depth = 8
tris = scenes.heightfield(708)
eng = rt_amd.PbrtEngine(nx, ny, ns, device=0)
eng.setTriangles(tris)
eng.buildBvh()
eng.setMaxDepth(depth)
eng.setCamera(**scenes.BENCH_CAMERA)
frames, rays = [], []
for rounds in configs:
    eng.setOption(engine.OPT_TAIL_SPLIT, rounds)
    frames.append(eng.render())
    rays.append(eng.last_stats.rays)
    print(f"rounds {rounds}: {rays[-1]} rays, {eng.last_stats.kernel_ms:.1f} ms", flush=True)
bad = set()
for a in range(len(frames)):
    for b in range(a + 1, len(frames)):
        d = np.argwhere((frames[a] != frames[b]).any(axis=2))
        print(f"run {a} (rounds {configs[a]}) vs run {b} (rounds {configs[b]}): {len(d)} pixels differ, rays {rays[a] - rays[b]:+d}", flush=True)
        for j, i in d[:16]:
            bad.add((int(j), int(i)))
if bad:
    o = oracle_lib.load()
    cam = scenes.BENCH_CAMERA
    ocam = o.camera(cam["lookfrom"], cam["lookat"], 40.0, nx / ny, 0.0, 10.0)
    nodes = eng.getBvhNodes()
    for j, i in sorted(bad):
        fb_o, cnt = o.render(tris, nodes, o.material_table(), ocam, nx, ny, ns, depth, mode=1, threads=1, lattice=(i, j, nx, ny))
        agree = [bool(frames[k][j, i].tobytes() == fb_o[j, i].tobytes()) for k in range(len(frames))]
        print(f"pixel (i={i}, j={j}): oracle {fb_o[j, i]} ({cnt.rays} rays); runs agreeing with the oracle: {agree}; "
              f"values {[frames[k][j, i].tolist() for k in range(len(frames))]}", flush=True)
