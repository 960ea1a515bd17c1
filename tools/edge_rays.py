"""Probe: rays with unusual components (tiny / huge / denormal / zero / -0 direction components, far origins) through rt_trace_rays
against the oracle's reference visit pattern, on several kernel variants.  Prints mismatches per case; what it found in round 3 is in
DESIGN.md 3 (-0.0, direction length) and tests/test_gpu_parity.py::test_unusual_ray_directions."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rt_amd
from rt_amd import engine, scenes
from tests import oracle_lib

o = oracle_lib.load()
rs = np.random.RandomState(5)
import itertools
for (scene_name, tris), (kern, fmt) in itertools.product((("heightfield", scenes.heightfield(48)), ("spheres", scenes.spheres_scene())), ((2, 1), (2, 2), (0, 0))):
    scene_name = f"{scene_name} k{kern} f{fmt}"
    eng = rt_amd.PbrtEngine(32, 32, 1, device=0)
    eng.setOption(engine.OPT_KERNEL, kern); eng.setOption(engine.OPT_NODE_FORMAT, fmt)
    eng.setTriangles(tris); eng.buildBvh(); eng.setMaxDepth(8); eng.setCamera(**scenes.BENCH_CAMERA)
    nodes = eng.getBvhNodes()
    cases = {}
    n = 1500
    base_o = rs.uniform((-9, 0.5, -9), (9, 6, 9), size=(n, 3)).astype(np.float32)
    base_d = rs.normal(size=(n, 3)).astype(np.float32); base_d[:, 1] = -np.abs(base_d[:, 1])
    def mk(name, org, d): cases[name] = (org.astype(np.float32).copy(), d.astype(np.float32).copy())
    mk("plain", base_o, base_d)
    mk("scaled 1e-20", base_o, base_d * np.float32(1e-20))
    mk("scaled 1e-30", base_o, base_d * np.float32(1e-30))
    mk("scaled 1e+20", base_o, base_d * np.float32(1e20))
    d = base_d.copy(); d[:, 0] *= np.float32(1e-25); mk("x tiny", base_o, d)
    d = base_d.copy(); d[:, 0] *= np.float32(1e-38); mk("x denormal", base_o, d)
    d = base_d.copy(); d[:, 2] = np.float32(1e-45); mk("z smallest denormal", base_o, d)
    d = base_d.copy(); d[:, 0] = 0; d[:, 2] = 0; mk("straight down +0", base_o, d)
    og = base_o.copy(); og[:, 0] = np.round(og[:, 0] * 2) / 2; og[:, 2] = np.round(og[:, 2]); d = base_d.copy(); d[:, 0] = 0
    mk("origin on grid lines, dx 0", og, d)
    mk("scaled 1e-36", base_o, base_d * np.float32(1e-36))
    mk("scaled 1e+10", base_o, base_d * np.float32(1e10))
    d = base_d * np.float32(1e-15); d[:, 0] *= np.float32(1e-12); mk("scaled 1e-15, x 1e-27", base_o, d)
    mk("far origin", base_o + np.float32([0, 3e4, 0]), base_d * np.float32([1e-3, 1, 1e-3]))
    mk("very far origin", base_o + np.float32([0, 1e7, 0]), base_d * np.float32([1e-6, 1, 1e-6]))
    up = base_d.copy(); up[:, 1] = np.abs(up[:, 1]); mk("upwards", base_o, up)
    mk("from below", base_o - np.float32([0, 30, 0]), up)
    for name, (org, d) in cases.items():
        ids, ts = eng.traceRays(np.concatenate([org, d], axis=1))
        bad = 0; hits = 0; checked = 0
        first = None
        for k in range(0, n, 6):
            h, i, t, _ = o.hit(tris, nodes, org[k], d[k], "reference")
            checked += 1; hits += int(h)
            if ids[k] != (i if h else -1) or (h and np.float32(t) != ts[k]):
                bad += 1
                if first is None: first = (k, int(ids[k]), float(ts[k]), bool(h), int(i), float(t), org[k].tolist(), d[k].tolist())
        print(f"{scene_name:12s} {name:28s} checked {checked} hits {hits} mismatches {bad}", first or "")
    eng.close()
