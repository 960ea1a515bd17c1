#!/usr/bin/env python3
"""Primary-ray pass of the bench scene from several cameras: how much of its time is the longest rays' serial depth.
Prints kernel ms and, from the analysis build, node steps per ray."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import rt_amd  # noqa: E402
from rt_amd import scenes  # noqa: E402

tris = scenes.heightfield(708)
eng = rt_amd.PbrtEngine(1920, 1080, 1, device=0)
eng.setTriangles(tris)
eng.buildBvh()
eng.setMaxDepth(8)
for name, cam in (("bench", scenes.BENCH_CAMERA),
                  ("top-down", dict(lookfrom=[0, 30, 0.01], lookat=[0, 0, 0], vfov=40.0, dist_to_focus=10.0, aperture=0.0)),
                  ("steep", dict(lookfrom=[0, 24, 10], lookat=[0, 0, 0], vfov=40.0, dist_to_focus=10.0, aperture=0.0)),
                  ("grazing", dict(lookfrom=[0, 3, 18], lookat=[0, 0, 0], vfov=40.0, dist_to_focus=10.0, aperture=0.0))):
    eng.setCamera(**cam)
    best = 1e9
    for _ in range(6):
        ids, ts = eng.tracePrimary()
        best = min(best, eng.last_stats.kernel_ms)
    print(json.dumps({"camera": name, "primary_ms": best, "hit_share": float((ids >= 0).mean())}), flush=True)
