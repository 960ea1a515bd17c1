"""Probe: random material tables (kinds, albedos beyond [0, 1], fuzz 0 .. 2, refraction indices 0.3 .. 3 and exactly 1,
negative parameters), deep paths, on the spheres scene and a box of quads, against the oracle."""
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import rt_amd
from rt_amd import engine, scenes

F = np.float32


def tables():
    rs = np.random.RandomState(4)
    out = {}
    base = engine.default_materials() if hasattr(engine, "default_materials") else None
    for name in ("random kinds", "mirror and glass only", "extreme parameters"):
        m = np.zeros(50, dtype=engine.MAT_DTYPE)
        if name == "random kinds":
            m["kind"] = rs.randint(0, 4, size=50)
            m["albedo"] = rs.uniform(0, 1.2, size=(50, 3)).astype(F)
            m["param"] = rs.uniform(0, 2.0, size=50).astype(F)
        elif name == "mirror and glass only":
            m["kind"] = np.where(np.arange(50) % 2 == 0, 1, 2)
            m["albedo"] = rs.uniform(0.5, 1.0, size=(50, 3)).astype(F)
            m["param"] = np.where(np.arange(50) % 2 == 0, 0.0, rs.uniform(0.3, 3.0, size=50)).astype(F)
            m["param"][7] = 1.0  # refraction index exactly 1
        else:
            m["kind"] = rs.randint(0, 3, size=50)
            m["albedo"] = rs.choice(np.array([0.0, 1.0, 5.0, 1e-30, 0.5], dtype=F), size=(50, 3))
            m["param"] = rs.choice(np.array([0.0, -1.0, 1.0, 1e-20, 10.0, 0.999999], dtype=F), size=50)
        m["kind"][3] = 3  # the light stays a light
        m["albedo"][3] = (7, 7, 7)
        out[name] = m
    return out


def main():
    from tests import oracle_lib

    o = oracle_lib.load()
    scn = {"spheres": scenes.spheres_scene(), "heightfield": scenes.heightfield(24)}
    cams = {"default": scenes.DEFAULT_CAMERA, "bench": scenes.BENCH_CAMERA}
    nx, ny, ns, depth = 56, 32, 6, 40
    for (sname, tris), (mname, mats), (kern, fmt) in itertools.product(scn.items(), tables().items(), ((2, 1), (2, 2), (0, 0))):
        cam = cams["default" if sname == "spheres" else "bench"]
        eng = rt_amd.PbrtEngine(nx, ny, ns, device=0)
        eng.setOption(engine.OPT_KERNEL, kern)
        eng.setOption(engine.OPT_NODE_FORMAT, fmt)
        eng.setMaterials(mats)
        eng.setTriangles(tris)
        eng.buildBvh()
        eng.setMaxDepth(depth)
        eng.setCamera(**cam)
        fb = eng.render()
        nodes = eng.getBvhNodes()
        ocam = o.camera(cam["lookfrom"], cam["lookat"], 40.0, nx / ny, cam["aperture"], cam["dist_to_focus"])
        fb_o, cnt = o.render(tris, nodes, mats, ocam, nx, ny, ns, depth, mode=1)
        same = fb.tobytes() == fb_o.tobytes()
        neq = int((fb.view(np.uint32) != fb_o.view(np.uint32)).any(axis=2).sum())
        print(f"{sname:12s} {mname:24s} k{kern} f{fmt}: rays {eng.last_stats.rays} vs {cnt.rays}, differing pixels {neq}, nan pixels {int(np.isnan(fb_o).any(axis=2).sum())}"
              + ("" if same and eng.last_stats.rays == cnt.rays else "   <<<<<<"), flush=True)
        eng.close()


if __name__ == "__main__":
    main()
