"""Probe: unusual scenes (exact duplicates and coplanar overlaps = ties, huge / tiny / mixed coordinate scales, long
thin triangles, everything in one plane, a far-away cluster) rendered and primary-traced on several kernel variants
against the oracle.  Prints mismatching pixels per case."""
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import rt_amd
from rt_amd import engine, scenes
F = np.float32
rs = np.random.RandomState(1)


def light():
    v = np.array([[4, 10, -4], [-4, 10, -4], [-4, 10, 4], [4, 10, 4]], dtype=F)
    return scenes.make_triangles(v[[3, 3]], v[[2, 1]], v[[1, 0]], 3)


def quad(p0, p1, p2, p3, mat):
    p = np.array([p0, p1, p2, p3], dtype=F)
    return scenes.make_triangles(p[[0, 0]], p[[1, 2]], p[[2, 3]], mat)


def rnd_tris(n, lo, hi, size, mat=0):
    c = rs.uniform(lo, hi, size=(n, 3)).astype(F)
    a = c + rs.normal(size=(n, 3)).astype(F) * F(size)
    b = c + rs.normal(size=(n, 3)).astype(F) * F(size)
    return scenes.make_triangles(c, a, b, np.asarray(rs.randint(0, 50, size=n), dtype=np.int32) if mat is None else mat)


def build_cases():
    cases = {}
    ground = quad([-20, 0, -20], [-20, 0, 20], [20, 0, 20], [20, 0, -20], 8)
    base = rnd_tris(300, (-6, 0.2, -6), (6, 4, 6), 0.8, None)
    cases["duplicates (ties by rank)"] = np.concatenate([light(), ground, base, base, base[::-1].copy()])
    cases["coplanar overlapping quads"] = np.concatenate([light(), ground] + [quad([-5 + k, 1, -5], [-5 + k, 1, 5], [3 + k, 1, 5], [3 + k, 1, -5], k) for k in range(6)])
    cases["stack of coincident ground planes"] = np.concatenate([light()] + [quad([-20, 0, -20], [-20, 0, 20], [20, 0, 20], [20, 0, -20], m) for m in (8, 0, 1, 9, 4)])
    cases["long thin triangles"] = np.concatenate([light(), ground, scenes.make_triangles(rs.uniform(-8, 8, (200, 3)).astype(F) * F([1, 0.2, 1]) + F([0, 1, 0]),
                                                                                        rs.uniform(-8, 8, (200, 3)).astype(F) * F([1, 0.2, 1]) + F([0, 1, 0]),
                                                                                        rs.uniform(-8, 8, (200, 3)).astype(F) * F([1, 0.2, 1]) + F([0, 1.001, 0]), 0)])
    big = np.concatenate([light(), ground, base])
    scaled = big.copy()
    cases["mixed scales (tiny cluster + far cluster)"] = np.concatenate([light(), ground, rnd_tris(200, (-0.01, 1, -0.01), (0.01, 1.02, 0.01), 0.002, None),
                                                                        rnd_tris(200, (3000, 0, 3000), (3100, 100, 3100), 20, None), base])
    cases["spheres touching and nested"] = np.concatenate([light(), ground, scenes.make_spheres(np.array([[0, 1, 0], [0, 1, 0], [2, 1, 0], [1, 1, 0], [0, 3, 0]], F), [1.0, 0.5, 1.0, 0.25, 1.0], np.array([7, 4, 7, 0, 6], np.int32))])
    return cases


CAMS = {"default": dict(lookfrom=[3, 4, 11], lookat=[0, 1, 0], vfov=40.0, dist_to_focus=10.0, aperture=0.0),
        "inside the cluster, lens": dict(lookfrom=[0.5, 1.5, 0.5], lookat=[-2, 1, -3], vfov=40.0, dist_to_focus=2.0, aperture=0.3)}

def main():
    from tests import oracle_lib

    o = oracle_lib.load()
    nx, ny, ns, depth = 64, 36, 4, 12
    for (name, tris), (cname, cam), (kern, fmt) in itertools.product(build_cases().items(), CAMS.items(), ((2, 1), (2, 2), (1, 0), (0, 0))):
        eng = rt_amd.PbrtEngine(nx, ny, ns, device=0)
        eng.setOption(engine.OPT_KERNEL, kern)
        eng.setOption(engine.OPT_NODE_FORMAT, fmt)
        eng.setTriangles(tris)
        eng.buildBvh()
        eng.setMaxDepth(depth)
        eng.setCamera(**cam)
        fb = eng.render()
        ids, ts = eng.tracePrimary()
        nodes = eng.getBvhNodes()
        ocam = o.camera(cam["lookfrom"], cam["lookat"], 40.0, nx / ny, cam["aperture"], cam["dist_to_focus"])
        fb_o, cnt = o.render(tris, nodes, o.material_table(), ocam, nx, ny, ns, depth, mode=0)
        ids_o, ts_o, _ = o.trace_primary(tris, nodes, ocam, nx, ny, mode=0)
        bad_px = int((fb != fb_o).any(axis=2).sum())
        bad_id = int((ids != ids_o).sum())
        bad_t = int((ts.view(np.uint32) != ts_o.view(np.uint32)).sum())
        flag = "" if (bad_px, bad_id, bad_t) == (0, 0, 0) and eng.last_stats.rays > 0 else "   <<<<<<"
        print(f"{name:42s} {cname:26s} k{kern} f{fmt}: pixels {bad_px} ids {bad_id} t {bad_t} hits {(ids_o >= 0).sum()}{flag}", flush=True)
        eng.close()


if __name__ == "__main__":
    main()
