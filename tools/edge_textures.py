"""Probe: texture coordinates and texture sizes off the beaten path (negative, huge, NaN / inf coordinates, 1x1 and
non-square and wide textures, a texture index on a sphere and on metal) against the oracle."""
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import rt_amd
from rt_amd import engine, scenes

F = np.float32


def scene(uv_kind):
    v = np.array([[4, 10, -4], [-4, 10, -4], [-4, 10, 4], [4, 10, 4]], dtype=F)
    g = np.array([[-8, 0, -8], [-8, 0, 8], [8, 0, 8], [8, 0, -8]], dtype=F)
    w = np.array([[-6, 0, -5], [6, 0, -5], [6, 6, -5], [-6, 6, -5]], dtype=F)
    lw = np.array([[7, 0, -4], [7, 0, 4], [7, 5, 4], [7, 5, -4]], dtype=F)
    uvs = {"negative and beyond 1": np.array([[-3.25, -0.5], [-3.25, 4.75], [2.5, 4.75], [2.5, -0.5]], F),
           "huge": np.array([[-1e6, 3e7], [1e6, 3e7], [1e6, -2e7], [-1e6, -2e7]], F),
           "1e30 and denormal": np.array([[1e30, 1e-40], [-1e30, 1e-40], [1e30, -1e-40], [0, 0]], F),
           "nan and inf": np.array([[np.nan, 0.5], [0.5, np.inf], [-np.inf, 0.25], [0.25, 0.75]], F),
           "exactly on texel borders": np.array([[0, 0], [0, 1], [1, 1], [1, 0]], F)}
    uv = uvs[uv_kind]
    q = lambda p, mat, tex: scenes.make_triangles(p[[0, 0]], p[[1, 2]], p[[2, 3]], mat, tex_idx=tex, uv=np.stack([uv[[0, 1, 2]], uv[[0, 2, 3]]]))
    sph = scenes.make_spheres(np.array([[0, 1, 0], [2.5, 0.6, 3.5]], F), [1.0, 0.6], np.array([7, 0], np.int32))
    sph["texIdx"] = 1  # a sphere that names a texture: its hit record says -1 (triangle.h:207)
    return np.concatenate([scenes.make_triangles(v[[3, 3]], v[[2, 1]], v[[1, 0]], 3), q(g, 0, 0), q(w, 8, 1), q(lw, 3, 2),
                           q(np.array([[-7, 0, 4], [-7, 0, -4], [-7, 5, -4], [-7, 5, 4]], F), 5, 3), sph]), list(uvs)


def textures():
    rs = np.random.RandomState(2)
    return {0: rs.randint(0, 256, size=(1, 1, 4)).astype(np.uint8), 1: rs.randint(0, 256, size=(3, 257, 4)).astype(np.uint8),
            2: rs.randint(0, 256, size=(64, 2, 4)).astype(np.uint8), 3: rs.randint(0, 256, size=(2, 2, 4)).astype(np.uint8)}


def main():
    from tests import oracle_lib

    o = oracle_lib.load()
    cam = dict(lookfrom=[3, 4, 11], lookat=[0, 1.5, 0], vfov=40.0, dist_to_focus=10.0, aperture=0.0)
    nx, ny, ns, depth = 64, 36, 4, 10
    kinds = scene("huge")[1]
    tex = textures()
    for i, t in tex.items():
        o.set_texture(i, t)
    for kind, (kern, fmt) in itertools.product(kinds, ((2, 1), (2, 2), (0, 0))):
        tris, _ = scene(kind)
        eng = rt_amd.PbrtEngine(nx, ny, ns, device=0)
        eng.setOption(engine.OPT_KERNEL, kern)
        eng.setOption(engine.OPT_NODE_FORMAT, fmt)
        eng.setTriangles(tris)
        eng.buildBvh()
        eng.setMaxDepth(depth)
        eng.setCamera(**cam)
        for i, t in tex.items():
            eng.setTexture(i, t)
        fb = eng.render()
        ocam = o.camera(cam["lookfrom"], cam["lookat"], 40.0, nx / ny, 0.0, 10.0)
        fb_o, cnt = o.render(tris, eng.getBvhNodes(), o.material_table(), ocam, nx, ny, ns, depth, mode=1)
        neq = int((fb.view(np.uint32) != fb_o.view(np.uint32)).any(axis=2).sum())
        print(f"{kind:28s} k{kern} f{fmt}: rays {eng.last_stats.rays} vs {cnt.rays}, differing pixels {neq}, nan pixels {int(np.isnan(fb_o).any(axis=2).sum())}"
              + ("" if neq == 0 and eng.last_stats.rays == cnt.rays else "   <<<<<<"), flush=True)
        eng.close()


if __name__ == "__main__":
    main()
