"""rt_render_device_noalloc (include/rt_abi.h): the entry point for callers that cannot allocate.  Its frame must be
rt_render_device's, bit for bit (integer-exact bar of the other parity tests: same kernels, same arithmetic), on two
calls in a row, the second with a smaller input -- the static scratch array keeps the first call's counters, spill
words and round records, and the second call must not be confused by them.  Collected last on purpose."""
import ctypes as C

import numpy as np
import pytest

import rt_amd
from rt_amd import engine, scenes

pytestmark = pytest.mark.gpu


class _Hip:
    """Device buffers straight from the HIP runtime the library already loaded (no torch in this process)."""

    def __init__(self):
        self.lib = C.CDLL("libamdhip64.so")
        self.lib.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.lib.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.lib.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.lib.hipFree.argtypes = [C.c_void_p]

    def malloc(self, nbytes):
        p = C.c_void_p()
        assert self.lib.hipMalloc(C.byref(p), nbytes) == 0
        assert self.lib.hipMemset(p, 0xFF, nbytes) == 0  # poison: every pixel must be written
        return p

    def to_host(self, p, shape):
        out = np.zeros(shape, dtype=np.float32)
        assert self.lib.hipMemcpy(out.ctypes.data_as(C.c_void_p), p, out.nbytes, 2) == 0
        return out


def _frames(eng, hip, nx, ny):
    """The same frame through rt_render_device and through rt_render_device_noalloc."""
    eng.uponPixelNumChange()
    buf = hip.malloc(nx * ny * 12)
    eng.renderDevice(buf.value)
    want, rays = hip.to_host(buf, (ny, nx, 3)), eng.last_stats.rays
    assert hip.lib.hipMemset(buf, 0xFF, nx * ny * 12) == 0
    eng.renderDeviceNoAlloc(buf.value)
    got = hip.to_host(buf, (ny, nx, 3))
    hip.lib.hipFree(buf)
    assert eng.last_stats.rays == rays
    return want, got


def test_noalloc_equals_render_device_twice_second_smaller(oracle):
    hip = _Hip()
    # first call: a frame rendered in rounds (round records in the static array), few waves so that records are
    # waited for, an LDS stack of 2 entries so that the spill area is used; 8 spp over 8 rounds
    tris = scenes.heightfield(64)
    eng = rt_amd.PbrtEngine(200, 112, 8, device=0)
    eng.setOption(engine.OPT_TAIL_SPLIT, 8)
    eng.setOption(engine.OPT_LDS_STACK, 2)
    eng.setOption(engine.OPT_WAVES_PER_CU, 2)
    eng.setTriangles(tris)
    eng.buildBvh()
    eng.setMaxDepth(8)
    eng.setCamera(**scenes.BENCH_CAMERA)
    want, got = _frames(eng, hip, 200, 112)
    assert got.tobytes() == want.tobytes()
    # the frame is also the oracle's (the bar of every other parity test)
    cam = scenes.BENCH_CAMERA
    ocam = oracle.camera(cam["lookfrom"], cam["lookat"], 40.0, 200 / 112, 0.0, 10.0)
    fb_o, _ = oracle.render(tris, eng.getBvhNodes(), oracle.material_table(), ocam, 200, 112, 8, 8, mode=1)
    assert got.tobytes() == fb_o.tobytes()
    # second call, smaller input: fewer pixels, fewer samples, ragged tiles -- on the same context ...
    eng.setNx(93), eng.setNy(61), eng.setNs(5)
    want2, got2 = _frames(eng, hip, 93, 61)
    assert got2.tobytes() == want2.tobytes()
    # ... and on another context of the same GPU with another scene: the array is shared per GPU, not per context
    eng2 = rt_amd.PbrtEngine(90, 50, 3, device=0)
    eng2.setOption(engine.OPT_TAIL_SPLIT, 3)
    eng2.setTriangles(scenes.spheres_scene())
    eng2.buildBvh()
    eng2.setMaxDepth(30)
    eng2.setCamera(**scenes.DEFAULT_CAMERA)
    want3, got3 = _frames(eng2, hip, 90, 50)
    assert got3.tobytes() == want3.tobytes()
    # back on the first context: still right after another context has used the array
    want4, got4 = _frames(eng, hip, 93, 61)
    assert got4.tobytes() == want4.tobytes() and got4.tobytes() == got2.tobytes()
    eng.close()
    eng2.close()


def test_noalloc_limits_and_preconditions():
    hip = _Hip()
    eng = rt_amd.PbrtEngine(64, 64, 2, device=0)
    eng.setTriangles(scenes.heightfield(8))
    eng.setCamera(**scenes.BENCH_CAMERA)
    buf = hip.malloc(1280 * 720 * 12)
    with pytest.raises(rt_amd.RtError, match="no scene / no BVH"):
        eng.renderDeviceNoAlloc(buf.value)
    eng.buildBvh()
    eng.renderDeviceNoAlloc(buf.value)
    eng.setOption(engine.OPT_LEAF_SIZE, 4)  # the device layout is stale now: this entry point must not rebuild it
    with pytest.raises(rt_amd.RtError, match="not resident"):
        eng.renderDeviceNoAlloc(buf.value)
    eng.buildBvh()
    eng.setNx(1280), eng.setNy(720)
    eng.uponPixelNumChange()
    with pytest.raises(rt_amd.RtError, match="frame too large for the static scratch"):
        eng.renderDeviceNoAlloc(buf.value)
    eng.setOption(engine.OPT_TAIL_SPLIT, 1)  # one round needs no records: any frame size
    eng.renderDeviceNoAlloc(buf.value)
    assert eng.last_stats.rays >= 1280 * 720 * 2
    hip.lib.hipFree(buf)
    eng.close()
