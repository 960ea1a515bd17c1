"""CPU tests of the product's host side (no GPU needed): the C-ABI library loads and exports every symbol
include/rt_abi.h declares; the restated glibc rand() and libstdc++ std::sort match the real ones; the
bit-exact host BVH builder matches the oracle's (which uses rand()/std::sort directly, as the reference
does, pbrt_engine.cu:335-368); the wide device layout covers every primitive and returns the oracle's hits
when traversed by a plain Python loop."""
import os
import re

import numpy as np
import pytest

from rt_amd import engine, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol():
    lib = engine.load_library()
    hdr = open(os.path.join(ROOT, "include", "rt_abi.h")).read()
    declared = set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in rt_abi.h but not exported"
    assert declared == set(engine.ABI_SYMBOLS)
    assert lib.rt_abi_version() == 1


def test_struct_sizes():
    assert engine.TRI_DTYPE.itemsize == 148
    assert engine.NODE_DTYPE.itemsize == 32
    assert engine.WIDE_DTYPE.itemsize == 128
    assert engine.PRIM_DTYPE.itemsize == 48


def test_glibc_rand_restatement(oracle):
    lib = engine.load_library()
    for seed in (1, 2, 12345, 0):
        want = oracle.libc_rand(seed, 5000)
        got = np.zeros(5000, dtype=np.int32)
        assert lib.rt_test_glibc_rand(seed, 5000, engine._ptr(got)) == 0
        assert (want == got).all()
    assert oracle.libc_rand(1, 1)[0] == 1804289383


def _keys(kind, n, rs):
    if kind == "random":
        return rs.rand(n).astype(np.float32)
    if kind == "ties":
        return rs.randint(0, max(2, n // 50), n).astype(np.float32)
    if kind == "sorted":
        return np.arange(n, dtype=np.float32)
    if kind == "reverse":
        return np.arange(n, 0, -1).astype(np.float32)
    if kind == "equal":
        return np.zeros(n, dtype=np.float32)
    if kind == "organ":
        h = n // 2
        return np.concatenate([np.arange(h), np.arange(n - h, 0, -1)]).astype(np.float32)
    if kind == "killer":  # median-of-3 killer: drives introsort into its heapsort fallback
        k = n // 2
        a = np.zeros(2 * k, dtype=np.float32)
        for i in range(k):
            if i % 2 == 0:
                a[i] = i + 1
            else:
                a[i] = k + i + (k % 2 == 0)
            a[k + i] = 2 * (i + 1)
        return a
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["random", "ties", "sorted", "reverse", "equal", "organ", "killer"])
def test_introsort_restatement(oracle, kind):
    lib = engine.load_library()
    rs = np.random.RandomState(11)
    for n in (0, 1, 2, 3, 15, 16, 17, 33, 100, 1000, 4096, 50001):
        key = _keys(kind, n, rs)
        n = key.shape[0]
        idx = np.arange(n, dtype=np.int32)
        assert lib.rt_test_introsort(engine._ptr(key), engine._ptr(idx), n) == 0
        assert (idx == oracle.std_sort(key)).all(), (kind, n)


@pytest.mark.parametrize("scene", ["hf5", "hf40", "hf224", "spheres"])
def test_host_builder_bit_exact(oracle, scene):
    tris = {"hf5": lambda: scenes.heightfield(5), "hf40": lambda: scenes.heightfield(40),
            "hf224": lambda: scenes.heightfield(224), "spheres": scenes.spheres_scene}[scene]()
    want = oracle.build_bvh(tris)
    got = engine.host_build_bvh(tris)
    assert want.tobytes() == got.tobytes()


def test_host_builder_tiny_and_offset(oracle):
    tris = scenes.light_quad()
    assert engine.host_build_bvh(tris).tobytes() == oracle.build_bvh(tris).tobytes()
    one = tris[:1]
    assert engine.host_build_bvh(one).shape[0] == 1
    # a different number of preceding rand() calls changes the axis sequence (main.cpp:449-474 shares rand())
    t = scenes.heightfield(9)
    assert engine.host_build_bvh(t, 0).tobytes() != engine.host_build_bvh(t, 5).tobytes()


def _py_closest(wide, prims, org, d):
    """Plain-Python traversal of the wide layout (float64 boxes, float32-faithful primitive test is the
    oracle's job; here ids only)."""
    best_t, best = np.inf, -1
    stack = [0]
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / d.astype(np.float64)
    while stack:
        cur = stack.pop()
        if cur & 0x80000000:
            start, cnt = cur & 0x0FFFFFFF, ((cur >> 28) & 7) + 1
            for r in range(start, start + cnt):
                p = prims[r]
                if p["shape"] == 0:
                    v0 = p["a"][:3].astype(np.float64)
                    e1 = np.array([p["a"][3], p["b"][0], p["b"][1]], dtype=np.float64)
                    e2 = np.array([p["b"][2], p["b"][3], p["e2z"]], dtype=np.float64)
                    pv = np.cross(d, e2)
                    det = e1 @ pv
                    if det == 0:
                        continue
                    tv = org - v0
                    u = (tv @ pv) / det
                    qv = np.cross(tv, e1)
                    v = (d @ qv) / det
                    t = (e2 @ qv) / det
                    ok = 0 <= u <= 1 and v >= 0 and u + v <= 1 and t > 0.001
                else:
                    c, rad = p["a"][:3].astype(np.float64), float(p["a"][3])
                    oc = org - c
                    a_, b_, c_ = d @ d, 2 * (oc @ d), oc @ oc - rad * rad
                    disc = b_ * b_ - 4 * a_ * c_
                    ok = disc > 0
                    if ok:
                        t = (-b_ - np.sqrt(disc)) / (2 * a_)
                        ok = t > 0.001
                if ok and (t < best_t):
                    best_t, best = t, int(p["orig"])
            continue
        nd = wide[cur]
        for s in range(4):
            ch = int(nd["child"][s])
            if ch == 0xFFFFFFFF:
                continue
            lo, hi = nd["plane"][:3, s].astype(np.float64), nd["plane"][3:, s].astype(np.float64)
            t0, t1 = (lo - org) * inv, (hi - org) * inv
            tn, tf = np.minimum(t0, t1).max(), np.maximum(t0, t1).min()
            if max(tn, 0.001) <= min(tf, best_t):
                stack.append(ch)
    return best, best_t


@pytest.mark.parametrize("leaf,box_mode,topology", [(1, 0, 0), (4, 0, 0), (8, 0, 0), (4, 1, 0), (2, 1, 0), (1, 0, 1), (4, 0, 1),
                                                   (2, 0, 3), (1, 0, 2), (4, 1, 2), (8, 0, 3)])  # topology bit 1: collapse by cost
def test_wide_layout_covers_and_hits(oracle, leaf, box_mode, topology):
    for tris, lo, hi in ((scenes.heightfield(14), (-9, 3, -9), (9, 8, 9)), (scenes.spheres_scene(), (-8, 0.5, -8), (8, 6, 8))):
        nodes = engine.host_build_bvh(tris)
        wide, prims, depth, stack_entries = engine.host_wide_layout(tris, nodes, leaf, box_mode, topology=topology)
        n = tris.shape[0]
        assert prims.shape[0] == n and sorted(prims["orig"].tolist()) == list(range(n))
        # tie order = DFS leaf rank in the REFERENCE tree; with the reference topology it is also the device order
        rank = oracle.dfs_rank(nodes, n)
        assert (rank[prims["orig"]] == prims["ref_rank"]).all()
        if topology & 1 == 0:
            assert (prims["ref_rank"] == np.arange(n)).all()
        # every leaf range is referenced exactly once; every node but the root exactly once
        seen = np.zeros(n, dtype=np.int32)
        node_refs = np.zeros(wide.shape[0], dtype=np.int32)
        for nd in wide:
            for ch in nd["child"]:
                ch = int(ch)
                if ch == 0xFFFFFFFF:
                    continue
                if ch & 0x80000000:
                    s, c = ch & 0x0FFFFFFF, ((ch >> 28) & 7) + 1
                    assert c <= leaf
                    seen[s:s + c] += 1
                else:
                    node_refs[ch] += 1
        assert (seen == 1).all() and (node_refs[1:] == 1).all() and node_refs[0] == 0
        assert stack_entries >= 3 * depth
        rs = np.random.RandomState(2)
        hits = 0
        for _ in range(60):
            org = rs.uniform(lo, hi).astype(np.float32)
            d = rs.normal(size=3).astype(np.float32)
            d[1] = -abs(d[1])
            h, i, t, _ = oracle.hit(tris, nodes, org, d, "reference")
            got, gt = _py_closest(wide, prims, org.astype(np.float64), d.astype(np.float64))
            assert got == (i if h else -1), (org, d, i, got, t, gt)
            hits += h
        assert hits > 10


def test_division_by_pi_shortcut_is_exact():
    """kernels.hip div_ref_pi: (float)((double)c / 3.1415926) computed as (float)((double)c * (1 / 3.1415926)) unless the
    product's 29 bits below float precision sit within 8 of a rounding midpoint.  The claim behind it, checked on 40 M
    floats (dense near 0, 1 and the powers of two, random elsewhere): outside that guard the two agree bit for bit."""
    P = np.float64(3.1415926)
    r = np.float64(1.0) / P
    rs = np.random.RandomState(11)
    chunks = [rs.uniform(0, 1, size=10_000_000).astype(np.float32) for _ in range(3)]
    chunks.append(np.exp2(rs.uniform(-90, 0, size=10_000_000)).astype(np.float32))
    edge = np.concatenate([np.float32(2.0) ** np.arange(-90, 1, dtype=np.float32) * f for f in (np.float32(1), np.nextafter(np.float32(1), np.float32(2)), np.nextafter(np.float32(1), np.float32(0)))])
    chunks.append(edge.astype(np.float32))
    guarded = 0
    for c in chunks:
        c = c[c >= np.float32(1e-30)]
        q = c.astype(np.float64) * r
        low = q.view(np.uint64) & np.uint64(0x1FFFFFFF)
        risky = np.abs(low.astype(np.int64) - 0x10000000) <= 8
        guarded += int(risky.sum())
        fast = q.astype(np.float32)
        exact = (c.astype(np.float64) / P).astype(np.float32)
        assert (fast[~risky].view(np.uint32) == exact[~risky].view(np.uint32)).all()
    assert guarded < 100  # the guard is rare: ~17 in 2^29


def test_fastdiv_is_exact():
    """The kernels divide tile indices by tiles_x with a multiply-high and two shifts: exact for every 32-bit
    dividend, for the divisors a frame can have (1..8192) and awkward ones beyond."""
    import ctypes as C

    lib = engine.load_library()
    rs = np.random.RandomState(5)
    n = np.concatenate([np.arange(0, 70000, dtype=np.uint32), rs.randint(0, 2**32, size=200000, dtype=np.uint64).astype(np.uint32),
                        np.array([2**32 - 1, 2**31, 2**31 - 1, 2**26, 2**26 - 1], dtype=np.uint32)])
    q = np.zeros_like(n)
    for d in list(range(1, 300)) + [1023, 1024, 1025, 4095, 4096, 8191, 8192, 65535, 65536, 2**20 + 7, 2**31 - 1, 2**31, 2**32 - 1]:
        assert lib.rt_test_fastdiv(C.c_uint32(d), n.ctypes.data_as(C.c_void_p), n.shape[0], q.ctypes.data_as(C.c_void_p)) == 0
        assert (q == n // np.uint32(d)).all(), d


def test_collapse_by_cost_is_no_worse_than_greedy():
    """RT_OPT_COLLAPSE 1 minimises the summed surface area of the wide tree's inner nodes (the expected number of node
    steps of a random ray) for the given binary tree: it can never come out above the greedy collapse, and on these
    scenes it needs fewer nodes."""
    def inner_area(wide):
        area = 0.0
        for nd in wide:
            for s in range(4):
                ch = int(nd["child"][s])
                if ch != 0xFFFFFFFF and not ch & 0x80000000:
                    d = nd["plane"][3:6, s].astype(np.float64) - nd["plane"][0:3, s].astype(np.float64)
                    area += d[0] * d[1] + d[1] * d[2] + d[2] * d[0]
        return area

    for tris in (scenes.heightfield(30), scenes.spheres_scene()):
        nodes = engine.host_build_bvh(tris)
        for topo in (0, 1):
            for leaf in (1, 2, 4):
                greedy = engine.host_wide_layout(tris, nodes, leaf, 0, topology=topo)[0]
                cost = engine.host_wide_layout(tris, nodes, leaf, 0, topology=topo | 2)[0]
                assert inner_area(cost) <= inner_area(greedy) * (1 + 1e-6)
                assert cost.shape[0] <= greedy.shape[0]


def test_quantized_nodes_are_conservative():
    """64-byte quantized nodes: every decoded child box contains the float child box it came from, and is at
    most one grid step larger per side; child references are unchanged."""
    for tris in (scenes.heightfield(30), scenes.spheres_scene()):
        nodes = engine.host_build_bvh(tris)
        wide, qn, prims, _, _ = engine.host_wide_layout(tris, nodes, 1, 0, quant=True)
        assert engine.QUANT_DTYPE.itemsize == 64 and qn.shape[0] == wide.shape[0]
        assert (qn["child"] == wide["child"]).all()
        valid = wide["child"] != 0xFFFFFFFF
        for a in range(3):
            step = (qn["step_x"] if a == 0 else qn["step_yz"][:, a - 1]).astype(np.float64)
            assert (np.frexp(step)[0] == 0.5).all()  # powers of two
            for s in range(4):
                lo = qn["origin"][:, a].astype(np.float64) + ((qn["q"][:, 2 * a] >> (8 * s)) & 0xFF) * step
                hi = qn["origin"][:, a].astype(np.float64) + ((qn["q"][:, 2 * a + 1] >> (8 * s)) & 0xFF) * step
                v = valid[:, s]
                flo, fhi = wide["plane"][:, a, s].astype(np.float64), wide["plane"][:, 3 + a, s].astype(np.float64)
                assert (lo[v] <= flo[v]).all() and (hi[v] >= fhi[v]).all()
                assert (flo[v] - lo[v] <= step[v]).all() and (hi[v] - fhi[v] <= step[v]).all()
                assert (lo[~v] > hi[~v]).all()  # empty slots are inverted


@pytest.mark.parametrize("leaf,topology", [(2, 3), (1, 3), (2, 0), (2, 1), (1, 2)])
def test_packed_scene_is_the_quantized_tree(leaf, topology):
    """RT_OPT_NODE_FORMAT 2: nodes and primitives in one array of 48-byte records.  Walking it from record 0 by the
    rebuilt references (base4 + child byte) visits every record exactly once and meets, child for child, the
    quantised tree it was made from: the same boxes (origin, byte offsets, grid step as an exponent byte), inner
    children where the tree has inner children, and under every leaf the same primitive records in the same order.
    A tree with larger leaves does not fit the format and is refused (format 1 then runs)."""
    for tris in (scenes.heightfield(23), scenes.spheres_scene()):
        nodes = engine.host_build_bvh(tris)
        wide, qn, prims, _, _ = engine.host_wide_layout(tris, nodes, leaf, 0, quant=True, topology=topology)
        recs, rec_of_prim, n_wide = engine.host_pack_layout(tris, nodes, leaf, topology)
        assert n_wide == qn.shape[0] and recs.shape == (qn.shape[0] + prims.shape[0], 12)
        assert 3 * recs.shape[0] < 2 ** 28
        prim_words = prims.view(np.uint32).reshape(-1, 12)
        assert (recs[rec_of_prim] == prim_words).all() and np.unique(rec_of_prim).size == prims.shape[0]
        seen = np.zeros(recs.shape[0], dtype=np.int32)
        todo = [(0, 0)]  # (unit address, wide node index)
        while todo:
            unit, i = todo.pop()
            assert unit % 3 == 0
            seen[unit // 3] += 1
            pk = recs[unit // 3].view(engine.PACK_NODE_DTYPE)[0]
            q = qn[i]
            assert pk["origin"].tobytes() == q["origin"].tobytes() and pk["q"].tobytes() == q["q"].tobytes()
            steps = [q["step_x"], q["step_yz"][0], q["step_yz"][1]]
            for a in range(3):
                e = (int(pk["exps"]) >> (8 * a)) & 0xFF
                e = e - 256 if e >= 128 else e
                assert np.float32(2.0) ** e == steps[a]
            assert int(pk["base4"]) % 12 == 0 and int(pk["base4"]) // 4 > unit  # a child block lies behind its parent
            for s in range(4):
                ch, b = int(q["child"][s]), (int(pk["child_bytes"]) >> (8 * s)) & 0xFF
                if ch == 0xFFFFFFFF:
                    assert b == 0xFF
                    continue
                ref = int(pk["base4"]) + b
                if ch & 0x80000000:
                    first, cnt = ch & 0x0FFFFFFF, ((ch >> 28) & 7) + 1
                    assert ref & 2 and (ref & 1) + 1 == cnt
                    for k in range(cnt):
                        r = (ref >> 2) // 3 + k
                        assert (ref >> 2) % 3 == 0 and rec_of_prim[first + k] == r
                        seen[r] += 1
                else:
                    assert ref & 3 == 0
                    todo.append((ref >> 2, ch))
        assert (seen == 1).all()
    tris = scenes.heightfield(23)
    assert engine.host_pack_layout(tris, engine.host_build_bvh(tris), 4, 3) is None  # leaves of up to four primitives


def test_host_builder_bit_exact_on_ties(oracle):
    """make_bvh_nodes sorts by padded box corners with std::sort, whose order among EQUAL keys is whatever introsort
    leaves (median-of-three, unguarded partition, insertion sort below 16, heapsort past the depth limit): the restated
    sort must leave the same one.  Scenes made of ties: tools/edge_scenes.py (exact duplicates, coincident planes), a
    heightfield five times over, 3 000 identical triangles (every key equal at every level), and 135 000 records with
    three copies of each triangle (the parallel builder's subtrees)."""
    from tools import edge_scenes

    cases = dict(edge_scenes.build_cases())
    cases["heightfield x 5"] = np.concatenate([scenes.heightfield(30)] * 5)
    one = scenes.make_triangles(np.float32([[1, 2, 3]]), np.float32([[2, 2, 3]]), np.float32([[1, 3, 3]]), 0)
    cases["3000 identical"] = np.concatenate([scenes.light_quad(), np.repeat(one, 3000)])
    cases["135 k, every triangle three times"] = np.concatenate([scenes.heightfield(150)] * 3)
    for name, tris in cases.items():
        assert engine.host_build_bvh(tris).tobytes() == oracle.build_bvh(tris).tobytes(), name


def test_wide_layout_rejects_malformed_tree():
    tris = scenes.heightfield(4)
    nodes = engine.host_build_bvh(tris)
    bad = nodes.copy()
    bad["left"][-1] = bad.shape[0] - 1  # root points at itself
    with pytest.raises(engine.RtError):
        engine.host_wide_layout(tris, bad)
    bad = nodes.copy()
    leaf = np.nonzero(bad["left"] == bad["right"])[0]
    bad["left"][leaf[0]] = bad["right"][leaf[0]] = bad["left"][leaf[1]]  # a primitive referenced twice
    with pytest.raises(engine.RtError):
        engine.host_wide_layout(tris, bad)


def test_scene_generators_are_deterministic():
    a = scenes.heightfield(10)
    b = scenes.heightfield(10)
    assert a.tobytes() == b.tobytes()
    assert a.shape[0] == 2 * 10 * 10 + 2
    assert a["shapeType"][0] == 0 and a["matType"][0] == 3  # the light comes first
    assert (a["texIdx"] == -1).all()
    assert np.isfinite(a["normal"]).all()
    assert scenes.heightfield(224).shape[0] == 100352 + 2


def test_null_context_is_an_error_not_a_crash():
    """No GPU here, so no context can exist: every entry point that takes one must refuse a null handle with a
    status code (nothing exits or throws across the boundary, rt_abi.h), and rt_create must fail loudly."""
    import ctypes as C

    lib = engine.load_library()
    h = C.c_void_p()
    rc = lib.rt_create(C.byref(h), 0)
    if rc == 0:  # a GPU is present after all: nothing to check here
        lib.rt_destroy(h)
        pytest.skip("a HIP device is available")
    assert rc != 0 and not h.value
    null, buf = C.c_void_p(), (C.c_float * 16)()
    i64 = C.c_int64()
    assert lib.rt_resize(null, 8, 8, 1) != 0
    assert lib.rt_set_scene(null, buf, 1) != 0
    assert lib.rt_build_bvh(null, 0) != 0
    assert lib.rt_set_bvh_nodes(null, buf, 1) != 0
    assert lib.rt_get_bvh_nodes(null, None, 0, C.byref(C.c_int())) != 0
    assert lib.rt_set_materials(null, buf, 1) != 0
    assert lib.rt_set_texture(null, 0, buf, 1, 1) != 0
    assert lib.rt_set_camera(null, (C.c_float * 3)(), (C.c_float * 3)(), 40.0, 10.0, 0.0) != 0
    assert lib.rt_set_option(null, 1, 5) != 0 and lib.rt_get_option(null, 1, C.byref(i64)) != 0
    assert lib.rt_render(null, buf, None) != 0 and lib.rt_render_device(null, buf, None) != 0
    assert lib.rt_trace_primary(null, buf, buf, None) != 0
    assert lib.rt_trace_rays(null, buf, 1, buf, buf, None) != 0
    assert lib.rt_trace_rays_device(null, buf, 1, buf, buf, None) != 0
    lib.rt_destroy(null)  # destroying nothing must not crash; its status is not specified
    assert lib.rt_write_ppm(None, buf, 1, 1) != 0
    with pytest.raises(engine.RtError, match="no usable HIP device"):
        engine.PbrtEngine(8, 8, 1)


@pytest.mark.parametrize("flags", [0, 1, 2, 3])
def test_scene_blob_round_trip(flags):
    """rt_export_scene's block (here from the no-GPU hook): its sections are, byte for byte, the arrays of the device
    layout built from the same records and nodes; the header says what is in it; truncated or altered blocks are what
    rt_import_scene refuses."""
    tris = np.concatenate([scenes.heightfield(20), scenes.make_spheres(np.array([[0, 2, 0]], np.float32), [1.0], 7)])
    nodes = engine.host_build_bvh(tris)
    blob = engine.host_scene_blob(tris, nodes, leaf_size=2, box_mode=0, topology=3, flags=flags)
    info = engine.scene_blob_info(blob)
    assert info is not None and info["total_bytes"] == blob.nbytes and blob.nbytes % 256 == 0
    wide, qn, prims, depth, stack = engine.host_wide_layout(tris, nodes, leaf_size=2, box_mode=0, quant=True, topology=3)
    assert info["wide_nodes"] == qn.shape[0] and info["prims"] == prims.shape[0] == tris.shape[0]
    assert info["depth"] == depth and info["stack_entries"] == stack and info["uv"] == 0
    assert info["float_nodes"] == (wide.shape[0] if flags & 1 else 0)
    assert info["binary_nodes"] == (nodes.shape[0] if flags & 2 else 0)
    sec = lambda off, arr: blob[off:off + arr.nbytes].tobytes() == arr.tobytes()  # noqa: E731
    assert sec(info["off_qnodes"], qn) and sec(info["off_prims"], prims)
    if flags & 1:
        assert sec(info["off_float_nodes"], wide)
    if flags & 2:
        assert sec(info["off_binary_nodes"], nodes)
    for off in ("off_qnodes", "off_prims", "off_shade", "off_uv", "off_float_nodes", "off_binary_nodes"):
        assert info[off] % 256 == 0
    # what an importer refuses: a short block, a wrong magic, a header that promises more than the block holds
    assert engine.scene_blob_info(blob[:-256]) is None
    assert engine.scene_blob_info(blob[:100]) is None
    bad = blob.copy()
    bad[0] ^= 0xFF
    assert engine.scene_blob_info(bad) is None
    bad = blob.copy()
    bad[8:16] = np.frombuffer(np.uint64(info["wide_nodes"] + 4).tobytes(), dtype=np.uint8)  # +256 bytes of nodes
    assert engine.scene_blob_info(bad) is None


def test_scene_blob_with_valid_header_and_corrupt_sections_is_refused():
    """a block whose header is fine but whose sections are not -- stale, torn, padded after a truncation, or
    written by another build -- must be an RT_E_ARG on the host, not out-of-bounds reads or a stack overflow in the
    traversal kernel.  First line of defence: the payload checksum.  Second (shown by re-hashing the edited block): the
    structural pass over child references, leaf ranges, depth, material and texture indices."""
    tris = scenes.heightfield(12).copy()
    tris["texIdx"][7] = 2
    nodes = engine.host_build_bvh(tris)
    blob = engine.host_scene_blob(tris, nodes, flags=3)
    info = engine.scene_blob_info(blob)
    assert engine.scene_blob_check(blob.copy()) == ""
    q0 = info["off_qnodes"]

    def child_words(b, node):  # the four child references of a 64-byte quantised node
        return b[q0 + 64 * node + 48:q0 + 64 * node + 64].view(np.uint32)

    # any flipped payload bit: the checksum
    bad = blob.copy()
    bad[info["off_prims"] + 5] ^= 0x10
    assert "checksum" in engine.scene_blob_check(bad) and engine.scene_blob_info(bad) is None
    # a truncated-then-padded block: the tail zeroed
    bad = blob.copy()
    bad[-4096:] = 0
    assert "checksum" in engine.scene_blob_check(bad)
    # child reference beyond the node array (valid header, valid checksum after the re-hash)
    bad = blob.copy()
    refs = child_words(bad, 0)
    inner = [k for k in range(4) if refs[k] != 0xFFFFFFFF and not refs[k] & 0x80000000]
    assert inner
    refs[inner[0]] = info["wide_nodes"] + 3
    assert "out of range" in engine.scene_blob_check(bad, rehash=True)
    assert engine.scene_blob_check(bad) != ""  # and without the re-hash the checksum already refuses it
    # a reference back to the root: a cycle
    bad = blob.copy()
    child_words(bad, 1)[0] = 0
    assert engine.scene_blob_check(bad, rehash=True) != ""
    # a leaf range that runs off the primitive array
    bad = blob.copy()
    last = info["wide_nodes"] - 1
    refs = child_words(bad, last)
    leaf = [k for k in range(4) if refs[k] != 0xFFFFFFFF and refs[k] & 0x80000000]
    refs[leaf[0]] = 0x80000000 | (7 << 28) | (info["prims"] - 2)
    assert "primitive array" in engine.scene_blob_check(bad, rehash=True)
    # a stack announcement smaller than the tree needs
    bad = blob.copy()
    bad[8 + 6 * 8 + 4:8 + 6 * 8 + 8].view(np.int32)[0] = 4  # stack_entries
    assert "deeper" in engine.scene_blob_check(bad, rehash=True)
    # material and texture indices
    bad = blob.copy()
    bad[info["off_shade"] + 16 * 9 + 12:info["off_shade"] + 16 * 9 + 16].view(np.int32)[0] = 50
    assert "material" in engine.scene_blob_check(bad, rehash=True)
    bad = blob.copy()
    bad[info["off_uv"] + 32 * 3 + 24:info["off_uv"] + 32 * 3 + 28].view(np.int32)[0] = 16
    assert "texture" in engine.scene_blob_check(bad, rehash=True)
    # a block from another build: other record sizes / layout version
    bad = blob.copy()
    bad[7] = ord("1")
    assert "layout version" in engine.scene_blob_check(bad, rehash=True)
    bad = blob.copy()
    hdr = bad[:256].view(np.uint32)
    assert hdr[18] == 2 and hdr[19] == 64 and hdr[20] == 48  # layout_version, sizeof(QuantNode), sizeof(DevPrim)
    hdr[19] = 48
    assert "another build" in engine.scene_blob_check(bad, rehash=True)


def test_scene_blob_carries_texture_coordinates():
    tris = scenes.heightfield(6).copy()
    tris["texIdx"][5] = 3
    tris["u"][5] = [0.0, 1.0, 0.0]
    tris["v"][5] = [0.0, 0.0, 1.0]
    nodes = engine.host_build_bvh(tris)
    info = engine.scene_blob_info(engine.host_scene_blob(tris, nodes))
    assert info["uv"] == info["prims"] == tris.shape[0]


def test_shim_accepts_the_reference_mesh_signature(tmp_path):
    """PbrtEngine::addMesh of the C++ shim must take a class with the REFERENCE's signature,
    `virtual void genTriangles(std::vector<Triangle>&, bool _setNormal = false)` (include/RT/mesh.h:241), as well as
    rtscene::Mesh's one-argument form, const or not.  Compile-only (no GPU, nothing linked)."""
    import shutil
    import subprocess

    cxx = shutil.which("g++") or shutil.which("c++")
    if cxx is None:
        pytest.fail("no host C++ compiler")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "mock.cpp"
    src.write_text('''
#include "rt_amd/csrc/rt_engine.hpp"
struct Triangle { float f[33]; int calNormal, matType, shapeType, texIdx; };   // 148 bytes, as triangle.h:345-354
static_assert(sizeof(Triangle) == 148, "mock record");
struct RefMesh {                                    // the reference's Mesh, as far as the engine sees it
    virtual ~RefMesh() {}
    std::string getName() { return "light"; }
    virtual void genTriangles(std::vector<Triangle>& v, bool _setNormal = false) { (void)_setNormal; v.emplace_back(); }
};
struct ConstMesh {
    const char* getName() const { return "c"; }
    void genTriangles(std::vector<Triangle>& v, bool) const { v.emplace_back(); }
};
struct OneArg {
    std::string getName() const { return "o"; }
    void genTriangles(std::vector<rt_triangle148>& v) const { v.emplace_back(); }
};
void use(PbrtEngine& e, RefMesh* a, ConstMesh* b, OneArg* c, std::vector<RefMesh*>& all) {
    e.addMesh(a); e.addMesh(b); e.addMesh(c); e.addMesh(all);
}
''')
    out = subprocess.run([cxx, "-std=c++17", "-fsyntax-only", "-I", root, str(src)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-3000:]
