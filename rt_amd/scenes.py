"""Synthetic scenes for the parity tests and bench.py.

Everything here is integer hashing and IEEE float32 +, -, *, / in numpy: no libm call, so every box
regenerates bit-identical vertices.  Triangle records are filled the way the reference's host-side
constructors do (include/RT/triangle.h:90-111): edge1 = v1 - v0, edge2 = v2 - v0,
normal = unit_vector(cross(edge1, edge2)) with vec3.h's operation order.
"""
from __future__ import annotations

import numpy as np

from .engine import TRI_DTYPE

F = np.float32


def _cross(a, b):
    # vec3.h:113-117
    x = a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1]
    y = -(a[:, 0] * b[:, 2] - a[:, 2] * b[:, 0])
    z = a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]
    return np.stack([x, y, z], axis=1).astype(F)


def make_triangles(v0, v1, v2, mat, tex_idx: int = -1, uv=None) -> np.ndarray:
    """Triangle(v0, v1, v2, matType, texIdx[, u, v]) for arrays of vertices (n, 3) float32.  uv: (n, 3, 2) texture
    coordinates of the three vertices (Mesh::genTriangles passes vec3(uv1[0], uv2[0], uv3[0]) and the same for [1])."""
    v0 = np.ascontiguousarray(v0, dtype=F)
    v1 = np.ascontiguousarray(v1, dtype=F)
    v2 = np.ascontiguousarray(v2, dtype=F)
    n = v0.shape[0]
    t = np.zeros(n, dtype=TRI_DTYPE)
    t["v0"], t["v1"], t["v2"] = v0, v1, v2
    e1 = (v1 - v0).astype(F)
    e2 = (v2 - v0).astype(F)
    t["edge1"], t["edge2"] = e1, e2
    c = _cross(e1, e2)
    # length(): sqrt(x*x + y*y + z*z) summed left to right, float sqrt (vec3.h:44); three divides (vec3.h:101)
    ln = np.sqrt(((c[:, 0] * c[:, 0]).astype(F) + (c[:, 1] * c[:, 1]).astype(F)).astype(F) + (c[:, 2] * c[:, 2]).astype(F))
    ln = ln.astype(F)
    with np.errstate(divide="ignore", invalid="ignore"):
        t["normal"] = (c / ln[:, None]).astype(F)
    t["calNormal"] = -1
    t["matType"] = mat
    t["shapeType"] = 0
    t["texIdx"] = tex_idx
    if uv is not None:
        uv = np.ascontiguousarray(uv, dtype=F).reshape(n, 3, 2)
        t["u"] = uv[:, :, 0]
        t["v"] = uv[:, :, 1]
    return t


def make_spheres(centers, radii, mat) -> np.ndarray:
    """Triangle(center, radius, matType): the sphere variant (triangle.h:95-100)."""
    centers = np.ascontiguousarray(centers, dtype=F).reshape(-1, 3)
    n = centers.shape[0]
    t = np.zeros(n, dtype=TRI_DTYPE)
    t["v0"] = centers
    t["v1"][:, 0] = np.asarray(radii, dtype=F)
    t["calNormal"] = -1
    t["matType"] = mat
    t["shapeType"] = 1
    t["texIdx"] = -1
    return t


def light_quad(material: int = 3) -> np.ndarray:
    """make_a_cornell_light (include/RT/MeshFactory.h:423-429 -> :29-74): corners (+-4, 10, +-4), flipped
    winding {3,2,1, 3,1,0}, material 3, emitted first (pbrt_engine.cu:167-172)."""
    v = np.array([[4, 10, -4], [-4, 10, -4], [-4, 10, 4], [4, 10, 4]], dtype=F)
    idx = [3, 2, 1, 3, 1, 0]
    a = v[[idx[0], idx[3]]]
    b = v[[idx[1], idx[4]]]
    c = v[[idx[2], idx[5]]]
    return make_triangles(a, b, c, material)


def _hash32(x: np.ndarray) -> np.ndarray:
    """Integer avalanche (lowbias32) on uint32 arrays."""
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x = (x * np.uint32(0x7FEB352D)).astype(np.uint32)
    x ^= x >> np.uint32(15)
    x = (x * np.uint32(0x846CA68B)).astype(np.uint32)
    x ^= x >> np.uint32(16)
    return x


def _lattice(ix, iz, seed):
    h = _hash32(ix.astype(np.uint32) * np.uint32(0x9E3779B1) ^ _hash32(iz.astype(np.uint32) + np.uint32(seed)))
    return ((h >> np.uint32(8)).astype(F) / F(16777216.0)).astype(F)  # [0, 1), exact in float32


def _value_noise(i, j, cell, seed):
    """Bilinear value noise on an integer lattice of period `cell`, smoothstep weights; float32 +,-,*,/ only."""
    ci, cj = i // cell, j // cell
    fi = ((i % cell).astype(F) / F(cell)).astype(F)
    fj = ((j % cell).astype(F) / F(cell)).astype(F)
    wi = (fi * fi * (F(3.0) - F(2.0) * fi)).astype(F)
    wj = (fj * fj * (F(3.0) - F(2.0) * fj)).astype(F)
    a = _lattice(ci, cj, seed)
    b = _lattice(ci + 1, cj, seed)
    c = _lattice(ci, cj + 1, seed)
    d = _lattice(ci + 1, cj + 1, seed)
    ab = (a + (b - a) * wi).astype(F)
    cd = (c + (d - c) * wi).astype(F)
    return (ab + (cd - ab) * wj).astype(F)


def heightfield(G: int, with_light: bool = True, amplitude: float = 2.5, seed: int = 7) -> np.ndarray:
    """G x G cells over [-10, 10]^2, two triangles per cell (2*G*G triangles), heights from four octaves of
    integer-lattice value noise, lambertian materials 0/1/8/9 by cell hash, preceded by the light quad.
    G=224 -> 100 352 (+2) triangles, G=708 -> 1 002 528 (+2), G=7072 -> 100 026 368 (+2)."""
    n = G + 1
    ii, jj = np.meshgrid(np.arange(n, dtype=np.int64), np.arange(n, dtype=np.int64), indexing="ij")
    h = np.zeros((n, n), dtype=F)
    amp = F(amplitude)
    cell = max(G // 4, 1)
    for o in range(4):
        h = (h + amp * (_value_noise(ii, jj, cell, seed + 101 * o) - F(0.5))).astype(F)
        amp = (amp * F(0.5)).astype(F)
        cell = max(cell // 3, 1)
    x = (F(-10.0) + (F(20.0) * ii.astype(F)) / F(G)).astype(F)
    z = (F(-10.0) + (F(20.0) * jj.astype(F)) / F(G)).astype(F)
    P = np.stack([x, h, z], axis=2).astype(F)  # (n, n, 3): P[i, j]
    p00 = P[:-1, :-1].reshape(-1, 3)
    p01 = P[:-1, 1:].reshape(-1, 3)  # j+1 (z+)
    p10 = P[1:, :-1].reshape(-1, 3)  # i+1 (x+)
    p11 = P[1:, 1:].reshape(-1, 3)
    cells = G * G
    v0 = np.empty((2 * cells, 3), dtype=F)
    v1 = np.empty((2 * cells, 3), dtype=F)
    v2 = np.empty((2 * cells, 3), dtype=F)
    v0[0::2], v1[0::2], v2[0::2] = p00, p01, p11
    v0[1::2], v1[1::2], v2[1::2] = p00, p11, p10
    ci = np.arange(cells, dtype=np.uint32)
    pal = np.array([0, 1, 8, 9], dtype=np.int32)
    # material varies in patches of 16x16 cells so that neighbouring pixels mostly share one
    patch = ((ci // np.uint32(G)) // np.uint32(16)) * np.uint32(65537) + (ci % np.uint32(G)) // np.uint32(16)
    mat_cell = pal[(_hash32(patch + np.uint32(seed)) & np.uint32(3)).astype(np.int64)]
    mat = np.repeat(mat_cell, 2)
    tris = make_triangles(v0, v1, v2, mat)
    if with_light:
        tris = np.concatenate([light_quad(), tris])
    return tris


# camera of SURVEY.md 8(d): lookfrom (0,14,18), lookat (0,0,0), focus 10, aperture 0 (vfov forced to 40)
BENCH_CAMERA = dict(lookfrom=(0.0, 14.0, 18.0), lookat=(0.0, 0.0, 0.0), vfov=40.0, dist_to_focus=10.0, aperture=0.0)


def spheres_scene(seed: int = 3, grid: int = 11) -> np.ndarray:
    """A restatement in spirit of the reference's default scene (main.cpp:408-500), config 1: the light quad,
    the ground sphere (radius 1000), three big spheres (glass 7, metal 4, metal 6) and a grid of small
    spheres / 12-triangle cubes with lambertian (10..29), metal (30..49) and glass (7) materials.  The
    reference drives this with unseeded rand() and files absent from the checkout, so
    positions come from the integer hash instead; every primitive kind and material kind is exercised."""
    parts = [light_quad()]
    centers = [(0, -1000, 0), (0, 1, 0), (-4, 1, 0), (4, 1, 0)]
    radii = [1000.0, 1.0, 1.0, 1.0]
    mats = [0, 7, 4, 6]
    parts.append(make_spheres(np.array(centers, dtype=F), radii, np.array(mats, dtype=np.int32)))
    k = 0
    sph_c, sph_r, sph_m = [], [], []
    cube_tris = []
    unit = np.array(
        [[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1]], dtype=F
    )
    faces = [(0, 2, 1), (0, 3, 2), (4, 5, 6), (4, 6, 7), (0, 1, 5), (0, 5, 4), (2, 3, 7), (2, 7, 6), (1, 2, 6), (1, 6, 5),
             (0, 4, 7), (0, 7, 3)]
    for a in range(-grid, grid, 2):
        for b in range(-grid, grid, 2):
            hs = _hash32(np.array([k * 4 + 0, k * 4 + 1, k * 4 + 2, k * 4 + 3], dtype=np.uint32) + np.uint32(seed * 7919))
            u = ((hs >> np.uint32(8)).astype(F) / F(16777216.0)).astype(F)
            k += 1
            r = F(0.2) + u[0] / F(10.0)
            c = np.array([F(a) + F(0.4) * (u[1] - F(0.5)), r, F(b) + F(0.4) * (u[2] - F(0.5))], dtype=F)
            if float(np.abs(c[0] - F(4.0))) + float(np.abs(c[2])) < 1.2:
                continue
            choose = float(u[3])
            if choose < 0.3:
                m = 10 + int(hs[0] % 20)
            elif choose < 0.9:
                m = 30 + int(hs[1] % 20)
            else:
                m = 7
            if hs[2] & 1:
                sph_c.append(c)
                sph_r.append(r)
                sph_m.append(m)
            else:
                vv = (unit * r + c).astype(F)
                for f in faces:
                    cube_tris.append((vv[f[0]], vv[f[1]], vv[f[2]], m))
    parts.append(make_spheres(np.array(sph_c, dtype=F), np.array(sph_r, dtype=F), np.array(sph_m, dtype=np.int32)))
    if cube_tris:
        parts.append(
            make_triangles(
                np.array([t[0] for t in cube_tris], dtype=F),
                np.array([t[1] for t in cube_tris], dtype=F),
                np.array([t[2] for t in cube_tris], dtype=F),
                np.array([t[3] for t in cube_tris], dtype=np.int32),
            )
        )
    return np.concatenate(parts)


# the reference's dead 3-arg camera_init values (pbrt_engine.cu:613-622), used for the config-1 restatement
DEFAULT_CAMERA = dict(lookfrom=(5.5, 2.5, 8.0), lookat=(0.0, 1.3, 0.0), vfov=40.0, dist_to_focus=10.0, aperture=0.0)
