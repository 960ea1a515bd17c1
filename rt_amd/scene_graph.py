"""Scene flattening without GL: the step immediately before the trace loop.

Restates, in float32 numpy, the parts of the reference's host scene code that decide the kernel's input:

* ``Mesh`` / ``FlyWeight`` transforms and ``Mesh::genTriangles`` (reference include/RT/mesh.h:253-291, 591-622;
  src/cxx/mesh.cpp:119-156): model = translate . rotation . uniform scale (glm, column-major, float32); a mesh
  flagged ``isSphere`` emits ONE sphere record (centre = model . centroid, radius = |centroid - vertices[0]| .
  scale / 1.7320508), any other mesh one ``Triangle(v1, v2, v3, matType, texIdx)`` per index triple;
* the gather order of ``PbrtEngine::__genTrianglesFromScene`` (src/pbrt/pbrt_engine.cu:167-178): meshes are kept
  in a ``std::map<std::string, MESH_PTR>`` (main.cpp:711-713), i.e. name-sorted; the mesh named "light" goes first;
* the default scene of ``main()`` (main.cpp:408-500).

The reference's default scene cannot be reproduced as its author ran it: it needs
``models/teapot.obj`` and ``models/cube.obj`` through assimp, per-mesh pose files under ``.\\cache\\position`` and a
binary camera snapshot, all git-ignored and absent.  ``default_scene()`` is therefore the documented restatement:
everything the source specifies (light quad; four big spheres; the 11 x 11 field driven by the unseeded ``rand()``
with the reference's draw order; half of the small bodies instanced as cubes; materials 10+U*20 / 30+U*20 / 7), a
canonical +-1 twelve-triangle cube for ``cube.obj``, every cache-dependent scale = 1, the teapot omitted.  It also
returns how many ``rand()`` calls the scene consumed, because the BVH builder continues the same sequence
(``rt_build_bvh(ctx, rand_calls_before)``).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import scenes
from .engine import TRI_DTYPE, load_library

F = np.float32


class GlibcRand:
    """glibc rand() in its default state, through the library's restatement (rt_test_glibc_rand)."""

    def __init__(self, seed: int = 1, capacity: int = 4096):
        lib = load_library()
        self._buf = np.zeros(capacity, dtype=np.int32)
        rc = lib.rt_test_glibc_rand(seed, capacity, self._buf.ctypes.data_as(C.c_void_p))
        assert rc == 0
        self.calls = 0

    def rand(self) -> int:
        v = int(self._buf[self.calls])
        self.calls += 1
        return v

    def drand48(self) -> np.float32:
        """The reference's wrapper (include/RT/triangle.h:12-15): rand() / (float)RAND_MAX."""
        return F(F(self.rand()) / F(2147483647))


# glm is not in the reference's checkout (a system dependency, version unpinned): the operations below restate its
# published float arithmetic (glm/detail/type_mat4x4.inl, glm/ext/matrix_transform.inl), column-major, in float32.
def _translate(m: np.ndarray, v) -> np.ndarray:
    """glm::translate(m, v): m[3] = m[0]*v.x + m[1]*v.y + m[2]*v.z + m[3]  (columns, summed left to right)."""
    r = m.copy()
    acc = (m[:, 0] * F(v[0])).astype(F)
    acc = (acc + (m[:, 1] * F(v[1])).astype(F)).astype(F)
    acc = (acc + (m[:, 2] * F(v[2])).astype(F)).astype(F)
    r[:, 3] = (acc + m[:, 3]).astype(F)
    return r


def _scale(m: np.ndarray, s) -> np.ndarray:
    r = m.copy()
    for k in range(3):
        r[:, k] = (m[:, k] * F(s)).astype(F)
    return r


def _matmul(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """mat4 * mat4: column j = a[0]*b[j][0] + a[1]*b[j][1] + a[2]*b[j][2] + a[3]*b[j][3], summed left to right."""
    r = np.zeros((4, 4), dtype=F)
    for j in range(4):
        acc = (a[:, 0] * b[0, j]).astype(F)
        for k in range(1, 4):
            acc = (acc + (a[:, k] * b[k, j]).astype(F)).astype(F)
        r[:, j] = acc
    return r


def _sincos(x) -> tuple:
    """The library's sine / cosine (rt_test_sincosf): the same values in this file and in rt_scene.hpp."""
    s, c = C.c_float(), C.c_float()
    rc = load_library().rt_test_sincosf(C.c_float(float(x)), C.byref(s), C.byref(c))
    assert rc == 0
    return F(s.value), F(c.value)


def _rotate_identity(angle, axis) -> np.ndarray:
    """glm::rotate(mat4(1), angle, axis): Rodrigues' matrix as glm writes it (axis normalised by the caller)."""
    s, c = _sincos(angle)
    ax = np.asarray(axis, dtype=F)
    temp = ((F(1) - c) * ax).astype(F)
    rot = np.zeros((3, 3), dtype=F)  # rot[col][row]
    rot[0][0] = c + temp[0] * ax[0]
    rot[0][1] = temp[0] * ax[1] + s * ax[2]
    rot[0][2] = temp[0] * ax[2] - s * ax[1]
    rot[1][0] = temp[1] * ax[0] - s * ax[2]
    rot[1][1] = c + temp[1] * ax[1]
    rot[1][2] = temp[1] * ax[2] + s * ax[0]
    rot[2][0] = temp[2] * ax[0] + s * ax[1]
    rot[2][1] = temp[2] * ax[1] - s * ax[0]
    rot[2][2] = c + temp[2] * ax[2]
    r = np.eye(4, dtype=F)
    for col in range(3):  # Result[col] = I[0]*rot[col][0] + I[1]*rot[col][1] + I[2]*rot[col][2]
        r[0:3, col] = rot[col]
    return r


def _apply(m: np.ndarray, p: np.ndarray) -> np.ndarray:
    """vec3(m * vec4(p, 1)) for an (n, 3) array, as glm's mat4 * vec4 associates: (m0*x + m1*y) + (m2*z + m3*1)."""
    p = np.asarray(p, dtype=F).reshape(-1, 3)
    out = np.zeros_like(p)
    for r in range(3):
        add0 = ((m[r, 0] * p[:, 0]).astype(F) + (m[r, 1] * p[:, 1]).astype(F)).astype(F)
        add1 = ((m[r, 2] * p[:, 2]).astype(F) + (m[r, 3] * F(1)).astype(F)).astype(F)
        out[:, r] = (add0 + add1).astype(F)
    return out


class Mesh:
    """GL-free restatement of the fields of ``Mesh`` that reach the kernel."""

    def __init__(self, name: str, vertices, indices, mat_type: int = 0, tex_idx: int = -1, texcoords=None):
        self.name = name
        self.vertices = np.asarray(vertices, dtype=F).reshape(-1, 3)
        # Vertex::TexCoords; zero where the reference's loader would have supplied them (assimp, absent)
        self.texcoords = (np.zeros((self.vertices.shape[0], 2), dtype=F) if texcoords is None
                          else np.asarray(texcoords, dtype=F).reshape(-1, 2))
        self.indices = np.asarray(indices, dtype=np.int64).reshape(-1)
        self.mat_type = mat_type
        self.tex_idx = tex_idx
        self.translation = np.zeros(3, dtype=F)
        self.rotation = np.eye(4, dtype=F)
        self.scale = F(1.0)  # the reference's default is -1 until a pose file is loaded (mesh.h:85); see module doc
        self.is_sphere = False
        self.top = None  # FlyWeight support: setTopMesh (mesh.h:112-114)

    # -- console commands (mesh.h:293-364)
    def scale_command(self, amp: float):
        self.scale = F(self.scale * F(amp))

    def translate_command(self, x: float = 0.0, y: float = 0.0, z: float = 0.0):
        self.translation = (self.translation + np.array([x, y, z], dtype=F)).astype(F)

    def rotation_command(self, axis: int, degrees: float):
        """rotation{X,Y,Z}Command: R = glm::rotate(I, d / 180 * PI, axis) * R.  cos/sin are the library's (the
        reference's are its C runtime's cosf/sinf): last-bit agreement of a rotated pose with the reference is not
        claimed, agreement between this file and rt_scene.hpp is."""
        a = F(float(F(degrees) / F(180.0)) * 3.1415926)  # getRadians: _d / 180.f * PI, PI the double 3.1415926 (mesh.h:38,297-299)
        ax = [F(0), F(0), F(0)]
        ax[axis] = F(1)
        self.rotation = _matmul(_rotate_identity(a, ax), self.rotation)

    def reset_all(self):
        self.rotation = np.eye(4, dtype=F)
        self.translation = np.zeros(3, dtype=F)
        self.scale = F(1.0)

    def centroid(self) -> np.ndarray:
        """resolvePosition (mesh.h:243-251): the vertex mean.  A mesh without vertices (a FlyWeight asked directly)
        gives NaN, as the reference's 0/0 does."""
        if self.vertices.shape[0] == 0:
            return np.full(3, np.nan, dtype=F)
        acc = np.zeros(3, dtype=F)
        for v in self.vertices:  # m_vec3Position += v.Position, in order
            acc = (acc + v).astype(F)
        return (acc / F(self.vertices.shape[0])).astype(F)

    def get_position(self) -> np.ndarray:
        """getPosition (mesh.h:267-272): model . centroid."""
        return _apply(self.get_model(), self.centroid()[None, :])[0]

    def average_radius(self) -> np.float32:
        """getAverageRadius (mesh.h:253-265): |centroid - vertices[0]| * scale / sqrt(3), scale from the top mesh."""
        if self.vertices.shape[0] == 0:
            return F(np.nan)
        d = (self.centroid() - self.vertices[0]).astype(F)
        dist = np.sqrt((d[0] * d[0] + d[1] * d[1] + d[2] * d[2]).astype(F)).astype(F)
        return F(dist * self._who().scale / F(1.7320508075688772))

    # -- what genTriangles reads, honouring the top mesh of a FlyWeight (mesh.h:104-144, 279-281)
    def _who(self):
        return self.top if self.top is not None else self

    def get_mat_type(self) -> int:
        return self._who().mat_type

    def get_is_sphere(self) -> bool:
        return self._who().is_sphere

    def get_model(self) -> np.ndarray:
        w = self._who()
        m = np.eye(4, dtype=F)
        m = _translate(m, w.translation)
        m = _matmul(m, w.rotation)
        return _scale(m, w.scale)

    def gen_triangles(self) -> np.ndarray:
        """Mesh::genTriangles, src/cxx/mesh.cpp:119-156."""
        model = self.get_model()
        if self.get_is_sphere():
            centre = _apply(model, self.centroid()[None, :])
            return scenes.make_spheres(centre, [self.average_radius()], self.get_mat_type())
        tri = self.indices.reshape(-1, 3)
        w = _apply(model, self.vertices)
        uv = np.stack([self.texcoords[tri[:, 0]], self.texcoords[tri[:, 1]], self.texcoords[tri[:, 2]]], axis=1)
        return scenes.make_triangles(w[tri[:, 0]], w[tri[:, 1]], w[tri[:, 2]], self.get_mat_type(), self.tex_idx, uv=uv)


class FlyWeight(Mesh):
    """Re-emits a shared mesh under its own transform / material / sphere flag (mesh.h:591-622)."""

    def __init__(self, name: str, orig: Mesh):
        super().__init__(name, np.zeros((0, 3), dtype=F), [], 0)
        self.orig = orig

    def gen_triangles(self) -> np.ndarray:
        self.orig.top = self
        try:
            return self.orig.gen_triangles()
        finally:
            self.orig.top = None


# ---------------------------------------------------------------------------------------------------------------
# The reference's hand-rolled OBJ reader (include/RT/triangle.h:604-676): real meshes without
# assimp.  Only "v x y z" and "f a/b/c a/b/c a/b/c [a/b/c]" lines count (1-based vertex indices; a quad becomes
# (1,2,3) + (1,3,4)); any other face syntax is an error, as there.  Lines are read with getline(buffer, 255): a last
# line without a newline is dropped (the loop tests eof() first), a line of 255+ characters is refused here (the
# reference would spin on the stream's fail bit).
_FLOAT = __import__("re").compile(r"[+-]?(?:\d+\.?\d*(?:[eE][+-]?\d+)?|\.\d+(?:[eE][+-]?\d+)?|inf(?:inity)?|nan)", __import__("re").I)
_INT = __import__("re").compile(r"[+-]?\d+")


def _scan(buf: str, fmt) -> list:
    """sscanf for the two formats the reader uses: fmt is a list of ' ' (any white space), '%d', '%f' or a literal
    character; returns the conversions made before the first mismatch."""
    pos, out = 0, []
    for f in fmt:
        if f == " ":
            while pos < len(buf) and buf[pos].isspace():
                pos += 1
        elif f in ("%d", "%f"):
            while pos < len(buf) and buf[pos].isspace():
                pos += 1
            m = (_INT if f == "%d" else _FLOAT).match(buf, pos)
            if not m:
                break
            out.append(int(m.group()) if f == "%d" else F(float(m.group())))
            pos = m.end()
        else:
            if pos >= len(buf) or buf[pos] != f:
                break
            pos += 1
    return out


_V_FMT = ["v", " ", "%f", " ", "%f", " ", "%f"]
_F_FMT = ["f"] + [x for _ in range(4) for x in (" ", "%d", "/", "%d", "/", "%d")]


def load_obj(text: str):
    """loadObj on the text of a file.  Returns (vertices (n, 3) float32, faces (m, 3) int, 1-based as stored)."""
    vertices, faces = [], []
    lines = text.split("\n")
    for line in lines[:-1]:  # what follows the last newline is never processed (eof() ends the loop first)
        if len(line) > 254:
            raise ValueError("OBJ line longer than 254 characters")
        if len(line) >= 2 and line[0] == "v" and line[1] == " ":
            v = _scan(line, _V_FMT)
            if len(v) != 3:
                raise ValueError("ERROR: vertex not in wanted format in OBJLoader")
            vertices.append(v)
        elif len(line) >= 2 and line[0] == "f" and line[1] == " ":
            d = _scan(line, _F_FMT)
            if len(d) == 12:
                faces.append([d[0], d[3], d[6]])
                faces.append([d[0], d[6], d[9]])
            elif len(d) == 9:
                faces.append([d[0], d[3], d[6]])
            else:
                raise ValueError("ERROR: I don't know the format of that FaceMtl")
    return np.asarray(vertices, dtype=F).reshape(-1, 3), np.asarray(faces, dtype=np.int64).reshape(-1, 3)


def mesh_from_obj(name: str, text: str, mat_type: int = 0) -> Mesh:
    """A Mesh over an OBJ file's triangles (indices made 0-based, as loadTriangleFromMesh's `f.x() - 1`,
    pbrt_engine.cu:254-258)."""
    v, f = load_obj(text)
    if f.size and (f.min() < 1 or f.max() > v.shape[0]):
        raise ValueError("OBJ face index out of range")
    return Mesh(name, v, (f - 1).reshape(-1), mat_type=mat_type)


def gather(meshes) -> np.ndarray:
    """__genTrianglesFromScene (pbrt_engine.cu:167-178) over a std::map-ordered (name-sorted) collection."""
    by_name = {}
    for m in meshes:
        by_name.setdefault(m.name, m)  # registerThisModel ignores a repeated name (mesh.h:232-239)
    ordered = [by_name[k] for k in sorted(by_name, key=lambda s: s.encode())]
    parts = [m.gen_triangles() for m in ordered if m.name == "light"]
    parts += [m.gen_triangles() for m in ordered if m.name != "light"]
    return np.concatenate(parts) if parts else np.zeros(0, dtype=TRI_DTYPE)


def light_mesh() -> Mesh:
    """make_a_cornell_light (include/RT/MeshFactory.h:423-429 -> 29-74): flipped quad at y = 10, material 3."""
    v = [[4, 10, -4], [-4, 10, -4], [-4, 10, 4], [4, 10, 4]]
    return Mesh("light", v, [3, 2, 1, 3, 1, 0], mat_type=3, texcoords=_FLIP_UV)


_UV = [[0, 1], [0, 0], [1, 0], [1, 1]]       # TexCoords of the four corners of every factory quad
_FLIP_UV = [[1, 1], [1, 0], [0, 0], [0, 1]]  # ... of its flipped variant


def plane_mesh(kind: str, w: float = 1.0, h: float = 1.0, tex_idx: int = 0, flip: bool = False) -> Mesh:
    """make_a_{xy,yz,zx}_plane(w, h, flip) as addXyPlane / addYzPlane / addZxPlane register it (MeshFactory.h:76-126,
    181-230, 285-335; main.cpp:43-66): a quad of two triangles named "xy" / "yz" / "zx" (+ "_flip"), material
    LAMBERTIAN = 10 (main.cpp:20), texture index 0."""
    hw, hh = F(F(w) / F(2)), F(F(h) / F(2))
    if kind == "xy":
        v = [[-hw, hh, 0], [-hw, -hh, 0], [hw, -hh, 0], [hw, hh, 0]]
    elif kind == "yz":
        v = [[0, hh, -hw], [0, -hh, -hw], [0, -hh, hw], [0, hh, hw]]
    elif kind == "zx":
        v = [[hh, 0, -hw], [-hh, 0, -hw], [-hh, 0, hw], [hh, 0, hw]]
    else:
        raise ValueError("plane kind must be xy, yz or zx")
    idx = [3, 2, 1, 3, 1, 0] if flip else [0, 1, 2, 0, 2, 3]
    return Mesh(kind + ("_flip" if flip else ""), v, idx, mat_type=10, tex_idx=tex_idx, texcoords=_FLIP_UV if flip else _UV)


def default_texture(n: int = 64) -> np.ndarray:
    """Stand-in for test01.png (absent; decoding images is the caller's business): an n x n RGBA8 checkerboard of
    8 x 8 squares, computed with integers only."""
    i = np.arange(n)
    on = (((i[:, None] * 8) // n + (i[None, :] * 8) // n) % 2) == 0
    t = np.zeros((n, n, 4), dtype=np.uint8)
    t[..., 0] = np.where(on, 230, 40)
    t[..., 1] = np.where(on, 230, 60)
    t[..., 2] = np.where(on, 220, 160)
    t[..., 3] = 255
    return t


def unit_cube_mesh(name: str = "cube") -> Mesh:
    """Stand-in for models/cube.obj (absent): the canonical +-1 cube, 12 triangles."""
    v = [[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1]]
    idx = [0, 2, 1, 0, 3, 2, 4, 5, 6, 4, 6, 7, 0, 1, 5, 0, 5, 4, 2, 3, 7, 2, 7, 6, 1, 2, 6, 1, 6, 5, 0, 4, 7, 0, 7, 3]
    return Mesh(name, v, idx, mat_type=8)


def positioned_plane(kind: str, lb_a: float, lb_b: float, rt_a: float, rt_b: float, at: float, flip: bool = False) -> Mesh:
    """The positioned overloads make_a_xy_plane(lbx, lby, rtx, rty, z, flip) / make_a_yz_plane(lbz, lby, rtz, rty, x,
    flip) / make_a_zx_plane(lbz, lbx, rtz, rtx, y, flip) (MeshFactory.h:128-179, 232-283, 338-390): a quad between two
    corners in the plane `at`; named "xy" / "yz" / "zx" (+ "_flip") until the caller renames it; material 0 and no
    texture index until the caller sets them (the Mesh constructor's defaults, mesh.h:70-71)."""
    a0, b0, a1, b1, c = F(lb_a), F(lb_b), F(rt_a), F(rt_b), F(at)
    if kind == "xy":    # (lbx, lby, rtx, rty, z)
        v = [[a0, b1, c], [a0, b0, c], [a1, b0, c], [a1, b1, c]]
    elif kind == "yz":  # (lbz, lby, rtz, rty, x)
        v = [[c, b1, a0], [c, b0, a0], [c, b0, a1], [c, b1, a1]]
    elif kind == "zx":  # (lbz, lbx, rtz, rtx, y)
        v = [[b1, c, a0], [b0, c, a0], [b0, c, a1], [b1, c, a1]]
    else:
        raise ValueError("plane kind must be xy, yz or zx")
    idx = [3, 2, 1, 3, 1, 0] if flip else [0, 1, 2, 0, 2, 3]
    return Mesh(kind + ("_flip" if flip else ""), v, idx, mat_type=0, tex_idx=-1, texcoords=_FLIP_UV if flip else _UV)


class CubeMesh(Mesh):
    """Six faces under one pose (mesh.h:457-535).  Every face keeps the cube as its top mesh for good, so a face's
    triangles take the CUBE's model matrix and the CUBE's material (getMatType / getModel / isSphere look at the top
    mesh, mesh.h:104-144, 278-281) -- setMatTypeSubMesh writes the faces' own field, which nothing reads while the
    top mesh is set; the texture index is the face's own."""

    def __init__(self, front: Mesh, back: Mesh, left: Mesh, right: Mesh, up: Mesh, bottom: Mesh, name: str = "None"):
        super().__init__(name, np.zeros((0, 3), dtype=F), [], 0)
        self.faces = [front, back, left, right, up, bottom]  # APPLY_TO_ALL's order (mesh.h:492-497)
        for f in self.faces:
            f.top = self

    def set_mat_type_sub_mesh(self, m: int):
        for f in self.faces:
            f.mat_type = m

    def gen_triangles(self) -> np.ndarray:
        return np.concatenate([f.gen_triangles() for f in self.faces])

    def centroid(self) -> np.ndarray:
        """resolvePosition (mesh.h:511-522): the mean of the six faces' vertex means."""
        acc = np.zeros(3, dtype=F)
        for f in self.faces:
            acc = (acc + f.centroid()).astype(F)
        return (acc / F(6.0)).astype(F)

    def share_mesh_data(self, target: Mesh):
        """shareMeshData (mesh.h:165-179, 524-526): every face appends its vertices and (re-based) indices to
        `target`; the cube's pose does not travel."""
        for f in self.faces:
            base = target.vertices.shape[0]
            target.vertices = np.concatenate([target.vertices, f.vertices]).astype(F)
            target.texcoords = np.concatenate([target.texcoords, f.texcoords]).astype(F)
            target.indices = np.concatenate([target.indices, f.indices + base])


def make_a_cube_mesh(lb, rt, ry: float, translation, name: str = "None") -> CubeMesh:
    """make_a_cube_mesh(lb, rt, ry, translation, _setup) (MeshFactory.h:392-421): six positioned planes between the
    corners lb and rt, the back / right / bottom ones flipped, rotated by ry degrees about y and translated as ONE
    body; every face names texture 3."""
    front = positioned_plane("xy", lb[0], lb[1], rt[0], rt[1], rt[2], False)
    back = positioned_plane("xy", lb[0], lb[1], rt[0], rt[1], lb[2], True)
    left = positioned_plane("yz", lb[2], lb[1], rt[2], rt[1], lb[0], False)
    right = positioned_plane("yz", lb[2], lb[1], rt[2], rt[1], rt[0], True)
    top = positioned_plane("zx", lb[2], lb[0], rt[2], rt[0], rt[1], False)
    bottom = positioned_plane("zx", lb[2], lb[0], rt[2], rt[0], lb[1], True)
    cube = CubeMesh(front, back, left, right, top, bottom, name)
    cube.rotation_command(1, ry)
    cube.translate_command(*[float(t) for t in translation])
    for f in cube.faces:
        f.tex_idx = 3
    return cube


# Stand-in for .\\cache\\cube_locations.txt (absent, like every file under the reference's cache directory): the first
# line is the cubes' footprint, every other line "z x" of one cube -- here the letters R and T on a 5 x 7 dot matrix.
SIGNATURE_LOCATIONS = "60\n" + "".join(
    f"{-150 + 75 * row} {-460 + 75 * col}\n"
    for row, bits in enumerate(("1110.11111", "1001.00100", "1110.00100", "1010.00100", "1001.00100"))
    for col, ch in enumerate(bits) if ch == "1")


def signature_mesh(text: str = SIGNATURE_LOCATIONS, r: float = 0.5) -> Mesh:
    """signature_reader (main.cpp:334-369): one box per "z x" line, footprint width / 2.5 either side, all of one
    height width * (2 r + 1), their faces' vertices shared into ONE mesh "meshList" (so the boxes lose their own
    pose objects; they were made with none).  `r` stands for the reference's `rand() / RAND_MAX` right after
    `srand(time(nullptr))`, which no restatement can reproduce: it is a parameter here, and the session's rand()
    sequence is NOT reseeded (the builder continues the unseeded sequence, as for the default scene)."""
    lines = text.split("\n")
    width = F(float(lines[0]))
    locations = []
    for line in lines[1:]:
        v = [F(float(t)) for t in line.split()]
        if not line.strip():
            continue  # getline's last, empty line
        if len(v) != 2:
            raise ValueError("failed in signature reader.")
        locations.append((v[0], v[1]))
    mesh_list = Mesh("meshList", np.zeros((0, 3), dtype=F), [], 0)
    hw = F(float(width) / 2.5)                       # float hw = width / 2.5;
    h = F(float(width) * (float(r) * 2.0 + 1.0))     # float h = width * (r * 2.0f + 1.0);   r is a double
    for z, x in locations:
        box = make_a_cube_mesh((F(x - hw), F(0), F(z - hw)), (F(x + hw), h, F(z + hw)), 0.0, (0.0, 0.0, 0.0))
        box.share_mesh_data(mesh_list)
    return mesh_list


def stand_in_texture(index: int, n: int = 64) -> np.ndarray:
    """Stand-ins for the four images the reference's textured scene names (texture units 0..3; the files are absent):
    n x n RGBA8, integers only, one recognisable pattern each -- 0 the checkerboard of default_texture, 1 vertical
    stripes, 2 a two-way gradient, 3 a grid of dots."""
    if index == 0:
        return default_texture(n)
    i = np.arange(n)
    jj, ii = np.meshgrid(i, i, indexing="ij")  # jj = row, ii = column
    t = np.zeros((n, n, 4), dtype=np.uint8)
    if index == 1:
        on = ((ii * 8) // n) % 2 == 0
        t[..., 0], t[..., 1], t[..., 2] = np.where(on, 200, 60), np.where(on, 180, 90), np.where(on, 60, 200)
    elif index == 2:
        t[..., 0], t[..., 1], t[..., 2] = (ii * 255) // (n - 1), (jj * 255) // (n - 1), 128
    elif index == 3:
        on = (((ii * 8) % n) * 2 < n) & (((jj * 8) % n) * 2 < n)
        t[..., 0], t[..., 1], t[..., 2] = np.where(on, 250, 90), np.where(on, 240, 140), np.where(on, 200, 90)
    else:
        raise ValueError("stand-in textures 0..3")
    t[..., 3] = 255
    return t


def cat_meshes(r: float = 0.5, locations: str = SIGNATURE_LOCATIONS):
    """The reference's second scene, the `#ifdef CAT` block of main() (main.cpp:596-662): a textured box whose faces
    name texture 3, one sphere, a textured metal wall (material 4, texture 1), two textured LIGHT walls (textures 2
    and 0), the ceiling light, the signature boxes in metal and a floor.  Stand-ins where the reference's files are
    absent: models/sphere.obj -> the sphere-flagged unit cube (as for the default scene), cache\\cube_locations.txt ->
    SIGNATURE_LOCATIONS, the images -> stand_in_texture(0..3), every pose file -> scale 1, the camera snapshot -> the
    console's --camera.  Returns (meshes in registration order, rand() calls consumed = 0)."""
    meshes = []
    box02 = make_a_cube_mesh((0, 0, 0), (800, 450, 450), 15, (0, 0, 0), "box02")
    box02.set_mat_type_sub_mesh(3)  # LIGHT -- on the faces' own field, which the cube's material (0) overrides
    meshes.append(box02)
    s1 = unit_cube_mesh("s1")  # Model("models/sphere.obj", "s1"): material 5, flagged as a sphere
    s1.mat_type = 5
    s1.is_sphere = True
    meshes.append(s1)
    height = 800
    left_wall = positioned_plane("yz", -1000, 0, 1000, height, -1000, True)
    left_wall.name, left_wall.mat_type, left_wall.tex_idx = "left_wall", 4, 1
    right_wall = positioned_plane("yz", -1000, 0, 1000, height, 1000)
    right_wall.name, right_wall.mat_type, right_wall.tex_idx = "right_wall", 3, 2
    back_wall = positioned_plane("xy", -1000, 0, 1000, height, -1000)
    back_wall.name, back_wall.mat_type, back_wall.tex_idx = "back_wall", 3, 0
    meshes += [left_wall, right_wall, back_wall, light_mesh()]
    m = signature_mesh(locations, r)
    m.mat_type = 6  # METAL
    meshes.append(m)
    floor = positioned_plane("zx", -1000, -1000, 1000, 1000, 0, False)
    floor.name, floor.mat_type = "floor", 0  # setMatType(10), then setMatType(0)
    meshes.append(floor)
    return meshes, 0


def default_meshes():
    """main.cpp:408-500 restated (see the module docstring for what is assumed).  Returns (meshes, rand_calls):
    the registered meshes in registration order and the number of rand() calls the scene setup consumed."""
    rng = GlibcRand()
    meshes = [light_mesh()]
    cube = unit_cube_mesh("cube")  # mesh3: matType 8, flagged as a sphere (main.cpp:435-445)
    cube.is_sphere = True
    meshes.append(cube)
    centers = [np.array(c, dtype=F) for c in ((0, -1000, 0), (0, 1, 0), (-4, 1, 0), (4, 1, 0))]
    radii = [F(1000.0), F(1.0), F(1.0), F(1.0)]
    mats = [0, 7, 4, 6]
    flags = [True] * 4

    def local_random():  # 2 * (drand48() - 0.5), double arithmetic on a float (main.cpp:449-451)
        return F(2 * (float(rng.drand48()) - 0.5))

    for a in range(-11, 11, 2):
        for b in range(-11, 11, 2):
            choose_mat = rng.drand48()
            r = 0.2 + float(F(rng.drand48()) / F(10))  # float / int is a float division; `auto r` is a double (main.cpp:460)
            cx = F(a + 0.2 * float(local_random()))  # x argument drawn first
            cz = F(b + 0.2 * float(local_random()))
            center = np.array([cx, F(r), cz], dtype=F)
            d = (center - np.array([4, 0.2, 0], dtype=F)).astype(F)
            length = np.sqrt((d[0] * d[0] + d[1] * d[1] + d[2] * d[2]).astype(F))
            if float(length) > 0.9:
                centers.append(center)
                radii.append(F(r))
                flags.append(not (float(rng.drand48()) < 0.5))
                if float(choose_mat) < 0.3:
                    mats.append(int(F(10) + F(rng.drand48()) * F(20)))  # int + float * int: float arithmetic, then truncation
                elif float(choose_mat) < 0.9:
                    mats.append(int(F(30) + F(rng.drand48()) * F(20)))
                else:
                    mats.append(7)
    for i, (c, r, m, flag) in enumerate(zip(centers, radii, mats, flags)):
        fw = FlyWeight("c" + str(i), cube)
        fw.mat_type = m
        fw.translation = c.astype(F)  # setPositionFromVec3
        fw.scale = F(r)
        fw.is_sphere = bool(flag)
        meshes.append(fw)
    return meshes, rng.calls


def default_scene():
    """The default scene flattened as ``__genTrianglesFromScene`` does.  Returns (triangles, rand_calls): the
    records in kernel order and the rand() calls consumed before the BVH build."""
    meshes, calls = default_meshes()
    return gather(meshes), calls


def build_rand_calls(n: int) -> int:
    """rand() calls one ``make_bvh_nodes`` build over n primitives makes (pbrt_engine.cu:335-368: one call per
    invocation; an invocation over cnt > 2 primitives recurses into cnt/2 and cnt - cnt/2).  The reference never
    reseeds, so the second ``render`` of a session continues where the first build stopped."""
    memo = {}

    def calls(cnt):
        if cnt <= 2:
            return 1
        if cnt not in memo:
            memo[cnt] = 1 + calls(cnt // 2) + calls(cnt - cnt // 2)
        return memo[cnt]

    return calls(n) if n > 0 else 0
