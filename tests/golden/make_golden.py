"""Generates the fixtures under tests/golden/ from the CPU oracle (oracle/rt_oracle.cpp).

The reference holds no golden vectors for this path and cannot be built here (see oracle/rt_oracle.h), so
these fixtures freeze the ORACLE's output (parity unpinned against the reference itself); they guard the
oracle and the HIP path against regressions and travel to the GPU box, where /root/reference is absent.
Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rt_amd import scenes  # noqa: E402
from tests import oracle_lib  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = [
    dict(name="hf24_d8", scene="heightfield", scene_args=[24], nx=64, ny=36, ns=4, depth=8,
         lookfrom=[0.0, 14.0, 18.0], lookat=[0.0, 0.0, 0.0], aperture=0.0, focus=10.0),
    dict(name="hf24_d50", scene="heightfield", scene_args=[24], nx=64, ny=36, ns=4, depth=50,
         lookfrom=[0.0, 14.0, 18.0], lookat=[0.0, 0.0, 0.0], aperture=0.0, focus=10.0),
    dict(name="spheres_d50", scene="spheres_scene", scene_args=[], nx=64, ny=36, ns=4, depth=50,
         lookfrom=[5.5, 2.5, 8.0], lookat=[0.0, 1.3, 0.0], aperture=0.0, focus=10.0),
    dict(name="spheres_lens", scene="spheres_scene", scene_args=[], nx=48, ny=27, ns=3, depth=12,
         lookfrom=[5.5, 2.5, 8.0], lookat=[0.0, 1.3, 0.0], aperture=0.3, focus=9.0),
]


def main():
    o = oracle_lib.load()
    mats = o.material_table()
    np.save(os.path.join(HERE, "material_table.npy"), mats)
    out = []
    for case in CASES:
        tris = getattr(scenes, case["scene"])(*case["scene_args"])
        nodes = o.build_bvh(tris)
        nx, ny = case["nx"], case["ny"]
        cam = o.camera(case["lookfrom"], case["lookat"], 40.0, nx / ny, case["aperture"], case["focus"])
        fb, cnt = o.render(tris, nodes, mats, cam, nx, ny, case["ns"], case["depth"], mode=0)
        ids, ts, _ = o.trace_primary(tris, nodes, cam, nx, ny, mode=0)
        np.save(os.path.join(HERE, case["name"] + "_fb.npy"), fb)
        np.save(os.path.join(HERE, case["name"] + "_ids.npy"), ids)
        np.save(os.path.join(HERE, case["name"] + "_t.npy"), ts)
        c = dict(case)
        c["rays"] = int(cnt.rays)
        c["bvh_sha256"] = hashlib.sha256(nodes.tobytes()).hexdigest()
        c["n_tris"] = int(tris.shape[0])
        out.append(c)
        print(case["name"], "rays", cnt.rays, "mean", float(fb.mean()))
    o.write_ppm(os.path.join(HERE, "hf24_d8.ppm"), np.load(os.path.join(HERE, "hf24_d8_fb.npy")))
    with open(os.path.join(HERE, "cases.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
