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
