"""ctypes binding of include/rt_abi.h and a PbrtEngine mirror.

Reference surface mirrored (include/RT/pbrt_engine.h:51-203, src/pbrt/pbrt_engine.cu:65-236):
``setNx/setNy/setNs``, ``uponPixelNumChange``, ``renderScene``, plus the pieces the reference keeps in
other classes and that feed the kernel: the triangle list (``m_vecTriangles``), the camera values read
at pbrt_engine.cu:104-106, and the material table of material_init (:25-47).
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_AMD_LIB: load another build of the same library (the analysis build librt_amd_stats.so, tools/trav_stats.py)
LIB_PATH = os.environ.get("RT_AMD_LIB") or os.path.join(_HERE, "librt_amd.so")

# reference Triangle, include/RT/triangle.h:345-354 (148 bytes)
TRI_DTYPE = np.dtype(
    [
        ("v0", "<f4", 3), ("v1", "<f4", 3), ("v2", "<f4", 3),
        ("normal", "<f4", 3),
        ("edge1", "<f4", 3), ("edge2", "<f4", 3),
        ("u", "<f4", 3), ("v", "<f4", 3),
        ("n0", "<f4", 3), ("n1", "<f4", 3), ("n2", "<f4", 3),
        ("calNormal", "<i4"), ("matType", "<i4"), ("shapeType", "<i4"), ("texIdx", "<i4"),
    ]
)
assert TRI_DTYPE.itemsize == 148
# reference bvh_node, include/RT/triangle.h:364-391 (32 bytes)
NODE_DTYPE = np.dtype([("box_min", "<f4", 3), ("box_max", "<f4", 3), ("left", "<i4"), ("right", "<i4")])
assert NODE_DTYPE.itemsize == 32
MAT_DTYPE = np.dtype([("kind", "<i4"), ("albedo", "<f4", 3), ("param", "<f4")])
assert MAT_DTYPE.itemsize == 20

MAT_LAMBERTIAN, MAT_METAL, MAT_DIELECTRIC, MAT_LIGHT = 0, 1, 2, 3
OPT_MAX_DEPTH, OPT_HONOR_VFOV, OPT_LEAF_SIZE, OPT_BOX_MODE, OPT_SHARD_RANK, OPT_SHARD_WORLD, OPT_KERNEL = 1, 2, 3, 4, 5, 6, 7
OPT_REFILL_THRESHOLD, OPT_WAVES_PER_CU, OPT_LDS_STACK, OPT_NODE_FORMAT, OPT_LEAF_BATCH, OPT_OCCUPANCY, OPT_TOPOLOGY = 8, 9, 10, 11, 12, 13, 14
OPT_SHADE_BATCH, OPT_SWAP_BATCH, OPT_TAIL_SPLIT, OPT_SEED_SALT, OPT_RAW_SUM = 15, 16, 17, 18, 19
OPT_QUEUE_THRESHOLD = 20
OPT_QUEUE_WAVES = 21
OPT_PRIMARY_WAVES = 22
OPT_PRIMARY_CHUNK = 23
OPT_TAIL_TAPER = 24
OPT_COLLAPSE = 25
OPT_DETERMINISTIC = 26


class RtStats(C.Structure):
    _fields_ = [
        ("rays", C.c_uint64),
        ("kernel_ms", C.c_double),
        ("total_ms", C.c_double),
        ("wide_nodes", C.c_uint64),
        ("prims", C.c_uint64),
        ("stack_entries", C.c_int32),
        ("handoff_torn", C.c_int32),
    ]


class RtError(RuntimeError):
    pass


_lib = None

# every symbol include/rt_abi.h declares
ABI_SYMBOLS = [
    "rt_abi_version", "rt_create", "rt_destroy", "rt_last_error", "rt_resize", "rt_set_scene", "rt_build_bvh",
    "rt_set_bvh_nodes", "rt_get_bvh_nodes", "rt_default_materials", "rt_set_materials", "rt_set_camera",
    "rt_set_option", "rt_get_option", "rt_render", "rt_render_device", "rt_trace_primary", "rt_trace_rays",
    "rt_write_ppm", "rt_test_glibc_rand", "rt_test_introsort", "rt_test_build_bvh", "rt_test_wide_layout", "rt_test_tile_owner", "rt_test_fastdiv",
    "rt_test_trav_stats", "rt_test_sincosf", "rt_set_texture", "rt_trace_rays_device", "rt_test_dump_rays", "rt_test_wave_times",
    "rt_test_trace_dumped", "rt_render_device_noalloc", "rt_test_sort_dumped", "rt_export_scene", "rt_import_scene",
    "rt_test_scene_blob", "rt_test_scene_blob_info", "rt_test_scene_blob_check", "rt_test_pack_layout",
]


def load_library(path: str = LIB_PATH):
    """Load librt_amd.so.  Raises if it has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise RtError(f"{path} is missing: build it with `make -C rt_amd/csrc` (there is no CPU fallback)")
    lib = C.CDLL(path)
    vp, ip, fp = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_float)
    lib.rt_abi_version.restype = C.c_int
    lib.rt_create.argtypes = [C.POINTER(vp), C.c_int]
    lib.rt_destroy.argtypes = [vp]
    lib.rt_last_error.argtypes = [vp]
    lib.rt_last_error.restype = C.c_char_p
    lib.rt_resize.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    lib.rt_set_scene.argtypes = [vp, vp, C.c_int]
    lib.rt_build_bvh.argtypes = [vp, C.c_int64]
    lib.rt_set_bvh_nodes.argtypes = [vp, vp, C.c_int]
    lib.rt_get_bvh_nodes.argtypes = [vp, vp, C.c_int, C.POINTER(C.c_int)]
    lib.rt_default_materials.argtypes = [vp]
    lib.rt_set_materials.argtypes = [vp, vp, C.c_int]
    lib.rt_set_camera.argtypes = [vp, fp, fp, C.c_float, C.c_float, C.c_float]
    lib.rt_set_texture.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int]
    lib.rt_trace_rays_device.argtypes = [vp, vp, C.c_int, vp, vp, C.POINTER(RtStats)]
    lib.rt_test_dump_rays.argtypes = [vp, C.c_uint64]
    lib.rt_test_wave_times.argtypes = [vp, vp, C.c_uint64]
    lib.rt_test_trace_dumped.argtypes = [vp, C.POINTER(RtStats)]
    lib.rt_test_sort_dumped.argtypes = [vp, C.c_int]
    lib.rt_export_scene.argtypes = [vp, C.c_int, vp, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.rt_import_scene.argtypes = [vp, vp, C.c_uint64]
    lib.rt_test_scene_blob.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.rt_test_scene_blob_info.argtypes = [vp, C.c_uint64, vp]
    lib.rt_test_scene_blob_check.argtypes = [vp, C.c_uint64, C.c_int, C.c_char_p, C.c_int]
    lib.rt_set_option.argtypes = [vp, C.c_int, C.c_int64]
    lib.rt_get_option.argtypes = [vp, C.c_int, C.POINTER(C.c_int64)]
    lib.rt_render.argtypes = [vp, vp, C.POINTER(RtStats)]
    lib.rt_render_device.argtypes = [vp, vp, C.POINTER(RtStats)]
    lib.rt_render_device_noalloc.argtypes = [vp, vp, C.POINTER(RtStats)]
    lib.rt_trace_primary.argtypes = [vp, vp, vp, C.POINTER(RtStats)]
    lib.rt_trace_rays.argtypes = [vp, vp, C.c_int, vp, vp, C.POINTER(RtStats)]
    lib.rt_write_ppm.argtypes = [C.c_char_p, vp, C.c_int, C.c_int]
    lib.rt_test_glibc_rand.argtypes = [C.c_uint32, C.c_int, vp]
    lib.rt_test_introsort.argtypes = [vp, vp, C.c_int]
    lib.rt_test_build_bvh.argtypes = [vp, C.c_int, C.c_int64, vp]
    lib.rt_test_tile_owner.argtypes = [C.c_int, C.c_int, C.c_int, vp]
    lib.rt_test_fastdiv.argtypes = [C.c_uint32, vp, C.c_int, vp]
    lib.rt_test_wide_layout.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]
    lib.rt_test_pack_layout.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp]
    for name in ABI_SYMBOLS:
        if name != "rt_last_error":
            getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def default_materials() -> np.ndarray:
    """material_init + rand_init (pbrt_engine.cu:25-47,597-601) as a 50-entry table."""
    lib = load_library()
    out = np.zeros(50, dtype=MAT_DTYPE)
    rc = lib.rt_default_materials(_ptr(out))
    if rc:
        raise RtError(f"rt_default_materials failed ({rc})")
    return out


def host_build_bvh(tris: np.ndarray, rand_calls_before: int = 0) -> np.ndarray:
    """The bit-exact host builder alone (make_bvh_nodes, pbrt_engine.cu:335-368); needs no GPU."""
    lib = load_library()
    tris = np.ascontiguousarray(tris, dtype=TRI_DTYPE)
    out = np.zeros(2 * tris.shape[0] - 1, dtype=NODE_DTYPE)
    rc = lib.rt_test_build_bvh(_ptr(tris), tris.shape[0], rand_calls_before, _ptr(out))
    if rc:
        raise RtError(f"rt_test_build_bvh failed ({rc})")
    return out


def tile_owner(nx: int, ny: int, world: int) -> np.ndarray:
    """Which rank renders each pixel under tile sharding (RT_OPT_SHARD_*): (ny, nx) int32.  Needs no GPU."""
    lib = load_library()
    out = np.zeros((ny, nx), dtype=np.int32)
    rc = lib.rt_test_tile_owner(nx, ny, world, _ptr(out))
    if rc:
        raise RtError(f"rt_test_tile_owner failed ({rc})")
    return out


WIDE_DTYPE = np.dtype([("plane", "<f4", (6, 4)), ("child", "<u4", 4), ("aux", "<u4", 4)])
QUANT_DTYPE = np.dtype([("origin", "<f4", 3), ("step_x", "<f4"), ("q", "<u4", 6), ("step_yz", "<f4", 2), ("child", "<u4", 4)])
PRIM_DTYPE = np.dtype([("a", "<f4", 4), ("b", "<f4", 4), ("e2z", "<f4"), ("shape", "<i4"), ("orig", "<i4"), ("ref_rank", "<i4")])


def host_wide_layout(tris: np.ndarray, nodes: np.ndarray, leaf_size: int = 1, box_mode: int = 0, quant: bool = False,
                     topology: int = 0):
    """The device layout alone (wide nodes, re-ordered primitives, depth, stack entries); needs no GPU."""
    lib = load_library()
    tris = np.ascontiguousarray(tris, dtype=TRI_DTYPE)
    nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
    info = np.zeros(4, dtype=np.int32)
    rc = lib.rt_test_wide_layout(_ptr(tris), tris.shape[0], _ptr(nodes), nodes.shape[0], leaf_size, box_mode, topology, None,
                                 None, None, _ptr(info))
    if rc:
        raise RtError(f"rt_test_wide_layout failed ({rc})")
    wide = np.zeros(int(info[0]), dtype=WIDE_DTYPE)
    prims = np.zeros(int(info[1]), dtype=PRIM_DTYPE)
    qn = np.zeros(int(info[0]), dtype=QUANT_DTYPE)
    rc = lib.rt_test_wide_layout(_ptr(tris), tris.shape[0], _ptr(nodes), nodes.shape[0], leaf_size, box_mode, topology,
                                 _ptr(wide), _ptr(qn), _ptr(prims), _ptr(info))
    if rc:
        raise RtError(f"rt_test_wide_layout failed ({rc})")
    if quant:
        return wide, qn, prims, int(info[2]), int(info[3])
    return wide, prims, int(info[2]), int(info[3])


PACK_NODE_DTYPE = np.dtype([("origin", "<f4", 3), ("child_bytes", "<u4"), ("q", "<u4", 6), ("base4", "<u4"), ("exps", "<u4")])


def host_pack_layout(tris: np.ndarray, nodes: np.ndarray, leaf_size: int = 2, topology: int = 3):
    """The packed scene of RT_OPT_NODE_FORMAT 2 alone (no GPU): (records as uint32[n, 12], rec_of_prim, wide nodes among
    the records), or None when the tree does not fit the format."""
    lib = load_library()
    tris = np.ascontiguousarray(tris, dtype=TRI_DTYPE)
    nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
    info = np.zeros(2, dtype=np.int64)
    args = (_ptr(tris), tris.shape[0], _ptr(nodes), nodes.shape[0], leaf_size, topology)
    if lib.rt_test_pack_layout(*args, None, None, _ptr(info)):
        raise RtError("rt_test_pack_layout failed")
    if info[0] == 0:
        return None
    recs = np.zeros((int(info[0]), 12), dtype=np.uint32)
    rec_of_prim = np.zeros(tris.shape[0], dtype=np.uint32)
    if lib.rt_test_pack_layout(*args, _ptr(recs), _ptr(rec_of_prim), _ptr(info)):
        raise RtError("rt_test_pack_layout failed")
    return recs, rec_of_prim, int(info[1])


def host_scene_blob(tris: np.ndarray, nodes: np.ndarray, leaf_size: int = 2, box_mode: int = 0, topology: int = 3, flags: int = 0) -> np.ndarray:
    """rt_export_scene's block from records and reference nodes alone (no GPU): uint8 array."""
    lib = load_library()
    tris = np.ascontiguousarray(tris, dtype=TRI_DTYPE)
    nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
    n = C.c_uint64(0)
    args = (_ptr(tris), tris.shape[0], _ptr(nodes), nodes.shape[0], leaf_size, box_mode, topology, flags)
    if lib.rt_test_scene_blob(*args, None, 0, C.byref(n)):
        raise RtError("rt_test_scene_blob failed")
    out = np.zeros(n.value, dtype=np.uint8)
    if lib.rt_test_scene_blob(*args, _ptr(out), out.nbytes, C.byref(n)):
        raise RtError("rt_test_scene_blob failed")
    return out


def scene_blob_info(blob: np.ndarray):
    """Header of a scene block, or None if rt_import_scene would refuse it."""
    lib = load_library()
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    info = np.zeros(16, dtype=np.uint64)
    if lib.rt_test_scene_blob_info(_ptr(blob), blob.nbytes, _ptr(info)):
        return None
    keys = ["wide_nodes", "prims", "uv", "float_nodes", "binary_nodes", "total_bytes", "depth", "stack_entries",
            "off_qnodes", "off_prims", "off_shade", "off_uv", "off_float_nodes", "off_binary_nodes"]
    return {k: int(v) for k, v in zip(keys, info)}


def scene_blob_check(blob: np.ndarray, rehash: bool = False) -> str:
    """"" if rt_import_scene would accept the block, else the reason it gives.  rehash: first rewrite the header's
    checksum from the payload as it is (in place), so that an edited block is judged by the structural checks."""
    lib = load_library()
    assert blob.dtype == np.uint8 and blob.flags["C_CONTIGUOUS"]
    buf = C.create_string_buffer(512)
    rc = lib.rt_test_scene_blob_check(_ptr(blob), blob.nbytes, 1 if rehash else 0, buf, 512)
    return "" if rc == 0 else (buf.value.decode() or f"error {rc}")


def write_ppm(path: str, fb: np.ndarray) -> None:
    """PPM writer of __renderScene (pbrt_engine.cu:123-133).  fb: (ny, nx, 3) float32, row 0 = bottom."""
    lib = load_library()
    fb = np.ascontiguousarray(fb, dtype=np.float32)
    ny, nx = fb.shape[0], fb.shape[1]
    rc = lib.rt_write_ppm(path.encode(), _ptr(fb), nx, ny)
    if rc:
        raise RtError(f"rt_write_ppm failed ({rc})")


def sample_shard(rank: int, world: int, ns_total: int):
    """Sample-pass sharding, the non-parity multi-GPU mode: rank r renders the WHOLE frame with
    ns_total / world samples (the first ranks take the remainder) from its own streams (seed salt r << 32) and writes
    per-pixel sums (OPT_RAW_SUM); the sums are added over ranks (all-reduce) and finished by assemble_sample_shards.
    Valid because the reference clamps per sample, so the clamped samples simply add up.  Returns (ns_local, salt)."""
    ns_local = ns_total // world + (1 if rank < ns_total % world else 0)
    return ns_local, rank << 32


def assemble_sample_shards(sum_fb: np.ndarray, ns_total: int) -> np.ndarray:
    """col /= float(ns) as the reference does it (multiply by float(1.0 / ns), vec3.h:158-165), then the final clamp."""
    kk = np.float32(1.0 / float(np.float32(ns_total)))
    out = (np.asarray(sum_fb, dtype=np.float32) * kk).astype(np.float32)
    return np.where(out.astype(np.float64) > 1.0, np.float32(1.0), out).astype(np.float32)


class PbrtEngine:
    """Mirror of the reference's PbrtEngine over the C ABI.  One engine drives one GPU."""

    def __init__(self, nx: int = 1280 // 2, ny: int = 720 // 2, ns: int = 100 // 10, device: int = 0):
        # NX/NY/NS, include/RT/pbrt_engine.h:24-26
        self._lib = load_library()
        self._h = C.c_void_p()
        rc = self._lib.rt_create(C.byref(self._h), device)
        if rc:
            self._h = C.c_void_p()
            raise RtError(f"rt_create failed ({rc}): no usable HIP device {device}; there is no CPU fallback")
        self.nx, self.ny, self.ns = nx, ny, ns
        self.tx = self.ty = 16  # printed only; the launch geometry is the library's (8x8 tiles, 64-lane waves)
        self.m_vecScene = []    # registered meshes (addMesh, pbrt_engine.h:99-100): flattened at render time
        self._textures = []     # (index, rgba8) registered for genTextures (pbrt_engine.h:133)
        self.num_triangles = 0
        self.nx_ny_changed = False
        self.rand_calls_before = 0  # rand() calls the host made before the next build (main.cpp:449-474 shares rand())
        self.last_stats = RtStats()
        self.uponPixelNumChange()

    # -- lifetime ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.rt_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc:
            msg = self._lib.rt_last_error(self._h)
            raise RtError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    # -- reference setters (pbrt_engine.h:65-91) ------------------------------------------
    def setNx(self, nx: int):
        if nx != self.nx:
            self.nx_ny_changed = True
        self.nx = nx

    def setNy(self, ny: int):
        if ny != self.ny:
            self.nx_ny_changed = True
        self.ny = ny

    def setNs(self, ns: int):
        self.ns = ns

    def set(self, nx: int, ny: int, ns: int):
        self.nx, self.ny, self.ns = nx, ny, ns

    def uponPixelNumChange(self):
        self._check(self._lib.rt_resize(self._h, self.nx, self.ny, self.ns), "rt_resize")

    # -- scene registration (pbrt_engine.h:99-100,133; pbrt_engine.cu:159-178) -------------
    def addMesh(self, m):
        """A list replaces the scene, a single mesh is appended (the reference's two overloads).  A mesh is anything
        with a ``name`` and ``gen_triangles()`` returning TRI_DTYPE records (rt_amd.scene_graph.Mesh)."""
        if isinstance(m, (list, tuple)):
            self.m_vecScene = list(m)
        else:
            self.m_vecScene.append(m)

    def registerTexture(self, index: int, rgba8: np.ndarray):
        self._textures.append((index, np.ascontiguousarray(rgba8, dtype=np.uint8)))

    def genTextures(self):
        """The reference builds its texture objects from image files here (cuManyTextures); the host registers decoded
        RGBA8 images instead and this hands them to the library."""
        for index, t in self._textures:
            self.setTexture(index, t)

    def _gather(self) -> np.ndarray:
        """__genTrianglesFromScene, pbrt_engine.cu:167-178: the mesh named "light" first, then the others in order."""
        parts = [m.gen_triangles() for m in self.m_vecScene if m.name == "light"]
        parts += [m.gen_triangles() for m in self.m_vecScene if m.name != "light"]
        return np.concatenate(parts) if parts else np.zeros(0, dtype=TRI_DTYPE)

    # -- scene ------------------------------------------------------------------------------
    def setTriangles(self, tris: np.ndarray):
        """m_vecTriangles (light first), as produced by Mesh::genTriangles."""
        tris = np.ascontiguousarray(tris, dtype=TRI_DTYPE)
        self._check(self._lib.rt_set_scene(self._h, _ptr(tris), int(tris.shape[0])), "rt_set_scene")
        self.num_triangles = int(tris.shape[0])

    def buildBvh(self, rand_calls_before: int = 0):
        self._check(self._lib.rt_build_bvh(self._h, rand_calls_before), "rt_build_bvh")

    def setBvhNodes(self, nodes: np.ndarray):
        nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
        self._check(self._lib.rt_set_bvh_nodes(self._h, _ptr(nodes), int(nodes.shape[0])), "rt_set_bvh_nodes")

    def exportSceneSize(self, flags: int = 0) -> int:
        n = C.c_uint64(0)
        self._check(self._lib.rt_export_scene(self._h, flags, None, 0, C.byref(n)), "rt_export_scene")
        return int(n.value)

    def exportScene(self, flags: int = 0, out: np.ndarray | None = None) -> np.ndarray:
        """The built device scene as one block of bytes (rt_export_scene); `out`: a uint8 buffer to fill, e.g. a
        np.memmap of a file in /dev/shm that the other ranks of the node map."""
        size = self.exportSceneSize(flags)
        if out is None:
            out = np.zeros(size, dtype=np.uint8)
        assert out.dtype == np.uint8 and out.nbytes >= size
        n = C.c_uint64(0)
        self._check(self._lib.rt_export_scene(self._h, flags, _ptr(out), out.nbytes, C.byref(n)), "rt_export_scene")
        return out

    def importScene(self, blob: np.ndarray):
        blob = np.asarray(blob)
        assert blob.dtype == np.uint8 and blob.flags["C_CONTIGUOUS"]
        self._check(self._lib.rt_import_scene(self._h, C.c_void_p(blob.ctypes.data), blob.nbytes), "rt_import_scene")

    def getBvhNodes(self) -> np.ndarray:
        m = C.c_int(0)
        self._check(self._lib.rt_get_bvh_nodes(self._h, None, 0, C.byref(m)), "rt_get_bvh_nodes")
        out = np.zeros(m.value, dtype=NODE_DTYPE)
        self._check(self._lib.rt_get_bvh_nodes(self._h, _ptr(out), m.value, C.byref(m)), "rt_get_bvh_nodes")
        return out

    def setMaterials(self, mats: np.ndarray):
        mats = np.ascontiguousarray(mats, dtype=MAT_DTYPE)
        self._check(self._lib.rt_set_materials(self._h, _ptr(mats), int(mats.shape[0])), "rt_set_materials")

    def setTexture(self, index: int, rgba8: np.ndarray):
        """One RGBA8 texture, shape (h, w, 4) uint8, row 0 first (Texture::getTexture, src/cxx/texture.cpp:45-86)."""
        t = np.ascontiguousarray(rgba8, dtype=np.uint8)
        assert t.ndim == 3 and t.shape[2] == 4
        self._check(self._lib.rt_set_texture(self._h, index, _ptr(t), t.shape[1], t.shape[0]), "rt_set_texture")

    def setCamera(self, lookfrom, lookat, vfov: float = 40.0, dist_to_focus: float = 10.0, aperture: float = 0.0):
        a = (C.c_float * 3)(*[float(x) for x in lookfrom])
        b = (C.c_float * 3)(*[float(x) for x in lookat])
        self._check(self._lib.rt_set_camera(self._h, a, b, vfov, dist_to_focus, aperture), "rt_set_camera")

    def setOption(self, opt: int, value: int):
        self._check(self._lib.rt_set_option(self._h, opt, int(value)), "rt_set_option")

    def getOption(self, opt: int) -> int:
        v = C.c_int64(0)
        self._check(self._lib.rt_get_option(self._h, opt, C.byref(v)), "rt_get_option")
        return int(v.value)

    def setMaxDepth(self, d: int):
        self.setOption(OPT_MAX_DEPTH, d)

    def setShard(self, rank: int, world: int):
        self.setOption(OPT_SHARD_WORLD, world)
        self.setOption(OPT_SHARD_RANK, rank)

    # -- render -----------------------------------------------------------------------------
    def render(self) -> np.ndarray:
        """render<<<>>> + sync + D2H.  Returns (ny, nx, 3) float32, row 0 = bottom row."""
        self.uponPixelNumChange()
        fb = np.zeros((self.ny, self.nx, 3), dtype=np.float32)
        self._check(self._lib.rt_render(self._h, _ptr(fb), C.byref(self.last_stats)), "rt_render")
        return fb

    def renderDevice(self, device_ptr: int) -> RtStats:
        """Same, into caller-owned device memory (e.g. a torch tensor's data_ptr()) of nx*ny*3 floats for the CURRENT
        nx, ny: a setNx/setNy/setNs since the last render takes effect here, as in render()."""
        self.uponPixelNumChange()
        self._check(self._lib.rt_render_device(self._h, C.c_void_p(device_ptr), C.byref(self.last_stats)),
                    "rt_render_device")
        return self.last_stats

    def renderDeviceNoAlloc(self, device_ptr: int) -> RtStats:
        """renderDevice through rt_render_device_noalloc: scratch from the library's static device array, no hipMalloc,
        nothing uploaded.  The caller's buffer must hold nx*ny*3 floats for the size of the last uponPixelNumChange()."""
        self._check(self._lib.rt_render_device_noalloc(self._h, C.c_void_p(device_ptr), C.byref(self.last_stats)),
                    "rt_render_device_noalloc")
        return self.last_stats

    def tracePrimary(self):
        self.uponPixelNumChange()
        ids = np.zeros((self.ny, self.nx), dtype=np.int32)
        ts = np.zeros((self.ny, self.nx), dtype=np.float32)
        self._check(self._lib.rt_trace_primary(self._h, _ptr(ids), _ptr(ts), C.byref(self.last_stats)),
                    "rt_trace_primary")
        return ids, ts

    def traceRays(self, org_dir6: np.ndarray):
        rays = np.ascontiguousarray(org_dir6, dtype=np.float32).reshape(-1, 6)
        n = rays.shape[0]
        ids = np.zeros(n, dtype=np.int32)
        ts = np.zeros(n, dtype=np.float32)
        self._check(self._lib.rt_trace_rays(self._h, _ptr(rays), n, _ptr(ids), _ptr(ts), C.byref(self.last_stats)),
                    "rt_trace_rays")
        return ids, ts

    def traceRaysDevice(self, rays_ptr: int, n: int, ids_ptr: int, ts_ptr: int) -> RtStats:
        """Closest hits for n rays already in this GPU's memory (6 floats each); ids (int32) and t (float32) stay there."""
        self._check(self._lib.rt_trace_rays_device(self._h, C.c_void_p(rays_ptr), n, C.c_void_p(ids_ptr), C.c_void_p(ts_ptr),
                                                   C.byref(self.last_stats)), "rt_trace_rays_device")
        return self.last_stats

    def renderScene(self, _buildScene: bool = True, out_path: str = "./output.ppm", file=sys.stdout) -> np.ndarray:
        """renderScene + __renderScene + join (pbrt_engine.cu:89-157): same stdout lines, same PPM bytes."""
        print(f"rendering a {self.nx}[width]x{self.ny}[height] image with {self.ns} samples per pixel in "
              f"{self.tx}x{self.ty} blocks.", file=file)
        if _buildScene:
            print("INFO: generate scene.", file=file)
            if self.m_vecScene:  # __renderScene, pbrt_engine.cu:96-98: clear, gather, build
                self.setTriangles(self._gather())
            print(f"Total num of Triangles: {self.num_triangles}", file=file)
            self.buildBvh(self.rand_calls_before)
        else:
            print("INFO: using scene generated.", file=file)
        fb = self.render()
        print(f"took {self.last_stats.kernel_ms / 1000.0:g} seconds.", file=file)
        write_ppm(out_path, fb)
        print("PbrtEngine: rendering finished.", file=file)
        return fb
