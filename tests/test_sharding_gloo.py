"""The N>1 path on CPU: two ranks over gloo.  Each rank holds the pixels the library's tile-sharding rule gives
it (zeros elsewhere, exactly what rt_render_device writes with RT_OPT_SHARD_*), and one all_reduce(sum) assembles
the frame: it must equal the single-rank frame bit for bit (x + 0 is exact).  The per-rank pixel values come from
the CPU oracle here (no GPU in this container); bench.py runs the same exchange with RCCL on the GPU frames."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, nx, ny, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from rt_amd import engine, scenes
    from tests import oracle_lib

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = oracle_lib.load()
    tris = scenes.heightfield(10)
    nodes = engine.host_build_bvh(tris)
    cam = o.camera((0, 14, 18), (0, 0, 0), 40.0, nx / ny, 0.0, 10.0)
    full, cnt = o.render(tris, nodes, o.material_table(), cam, nx, ny, 2, 8, mode=1, threads=2)
    owner = engine.tile_owner(nx, ny, world)
    mine = np.where((owner == rank)[..., None], full, np.float32(0.0)).astype(np.float32)
    t = torch.from_numpy(mine.copy())
    dist.all_reduce(t)  # frame assembly
    rays = torch.tensor([int((owner == rank).sum())], dtype=torch.int64)
    dist.all_reduce(rays)
    elapsed = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)  # bench.py takes the max over ranks
    ok = t.numpy().tobytes() == full.tobytes() and int(rays.item()) == nx * ny and abs(float(elapsed.item()) - 0.1 * world) < 1e-12
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok)))


@pytest.mark.parametrize("nx,ny", [(40, 24), (37, 21)])
def test_two_rank_frame_assembly(nx, ny):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + nx
    procs = [ctx.Process(target=_worker, args=(r, 2, port, nx, ny, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]


def _loop_worker(rank, world, port, base_nx, base_ny, strong, q):
    """bench.py's N > 1 half (rt_amd/sharding.py: FrameLoop, job_totals, share_through_file) on CPU tensors over gloo.
    The `render` stand-in writes what rt_render_device writes with RT_OPT_SHARD_*: this rank's tiles of the oracle's
    frame, zeros elsewhere (into a buffer poisoned first, as a real frame buffer holds the frame of two steps ago)."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from rt_amd import engine, scenes, sharding
    from tests import oracle_lib

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = True
    try:
        nx, ny = sharding.frame_size(base_nx, base_ny, world, strong)
        o = oracle_lib.load()
        tris = scenes.heightfield(8)
        nodes = engine.host_build_bvh(tris)
        cam = o.camera((0, 14, 18), (0, 0, 0), 40.0, nx / ny, 0.0, 10.0)
        frames = [o.render(tris, nodes, o.material_table(), cam, nx, ny, 1, 4 + k, mode=1, threads=1)[0] for k in range(2)]
        owner = engine.tile_owner(nx, ny, world)
        mine = owner == rank
        calls = [0]

        def render(fb):
            full = frames[calls[0] % 2]  # successive frames differ: a stale buffer would show
            calls[0] += 1
            fb.fill_(float("nan"))
            fb.copy_(torch.from_numpy(np.where(mine[..., None], full, np.float32(0.0)).astype(np.float32)))
            return int(mine.sum()), 1.0 + rank

        bufs = [torch.zeros((ny, nx, 3), dtype=torch.float32) for _ in range(2)]
        loop = sharding.FrameLoop(dist, world, rank, bufs, render, lambda: None)
        warmup, steps = 1, 3
        elapsed, rays_local, kernel_ms = loop.run(warmup, steps)
        elapsed, rays_total = sharding.job_totals(dist, world, elapsed, rays_local, "cpu")
        ok = ok and rays_total == nx * ny * steps and kernel_ms == [1.0 + rank] * steps and elapsed > 0.0
        assert ok, (rays_total, nx * ny * steps, kernel_ms, elapsed)
        if rank == 0:  # the last frame, assembled: the whole frame, bit for bit
            last = frames[(warmup + steps - 1) % 2]
            ok = ok and loop.last_frame().numpy().tobytes() == last.tobytes()
            assert ok, "assembled frame differs"
        # the scene block of `--share-scene on`: rank 0 writes it, the others map it; the file is gone afterwards
        blob = engine.host_scene_blob(tris, nodes)
        seen = {}

        def export_into(mm):
            mm[:] = blob

        def import_from(mm):
            seen["n_prims"] = int(engine.scene_blob_info(np.asarray(mm))["prims"])

        path = f"/dev/shm/rt_amd_gloo_test_{port}.bin"
        size = sharding.share_through_file(dist, rank, path, lambda: blob.nbytes, export_into, import_from)
        assert size == blob.nbytes and not os.path.exists(path), ("first hand-over", size, blob.nbytes)
        dist.barrier()  # (rank 0 is about to create the file again)
        if rank != 0:
            ok = ok and seen.get("n_prims") == tris.shape[0]
        # a rank whose import fails leaves no file behind either (the name is gone before any import starts)
        def bad_import(mm):
            raise RuntimeError("import failed")

        raised = False
        try:
            sharding.share_through_file(dist, rank, path, lambda: blob.nbytes, export_into, bad_import if rank == world - 1 else import_from)
        except RuntimeError:
            raised = True
        assert raised == (rank == world - 1) and not os.path.exists(path), ("second hand-over", raised, os.path.exists(path))
        dist.barrier()
    except Exception as e:  # noqa: BLE001
        ok = False
        print(f"rank {rank}: {type(e).__name__}: {e}", flush=True)
    q.put((rank, bool(ok)))
    try:
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        pass


@pytest.mark.parametrize("world,nx,ny,strong", [(2, 40, 24, False), (2, 37, 21, True), (4, 36, 20, False), (4, 45, 27, True)])
def test_frame_loop_over_gloo(world, nx, ny, strong):
    """The never-executed RCCL branch of bench.py, as far as it can be checked without several GPUs: the same FrameLoop
    (double-buffered asynchronous reduce to rank 0, barrier-bracketed timing, max / sum over ranks) with 2 and 4 ranks,
    weak (the frame grows with the ranks) and strong (fixed, ragged frame)."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + 7 * nx + world
    procs = [ctx.Process(target=_loop_worker, args=(r, world, port, nx, ny, strong, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(r, True) for r in range(world)]


def test_tile_owner_partitions_the_frame():
    from rt_amd import engine

    for nx, ny, world in ((64, 64, 2), (100, 60, 3), (37, 21, 8), (1920, 1080, 8)):
        owner = engine.tile_owner(nx, ny, world)
        assert owner.min() == 0 and owner.max() == min(world, ((nx + 7) // 8) * ((ny + 7) // 8)) - 1
        tiles_x = (nx + 7) // 8
        jj, ii = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
        assert (owner == ((jj // 8) * tiles_x + ii // 8) % world).all()  # 8x8 tiles, row-major tile index % world
        counts = np.bincount(owner.ravel(), minlength=world)
        if nx % 8 == 0 and ny % 8 == 0:
            assert counts.max() - counts.min() <= 64  # balanced to within one tile
