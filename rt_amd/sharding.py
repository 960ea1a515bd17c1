"""Frame assembly over ranks: the N > 1 half of bench.py, kept free of anything GPU-specific so that the same code runs
under RCCL on the GPUs (backend "nccl") and under gloo on CPU tensors in tests/test_sharding_gloo.py.

The reference is a single-device program (src/pbrt/pbrt_engine.cu:110-116); sharding is this project's:
every rank renders the pixels of its own 8x8 tiles (tile_index % world == rank) into a full-size frame that is zero
elsewhere, and ONE reduce(sum) to rank 0 assembles the frame exactly (x + 0).  Frames are rendered into two buffers in
turn and the reduce is issued asynchronously, so that frame k is assembled while frame k + 1 renders; a buffer is
rendered into again only after its reduce has completed.
"""
from __future__ import annotations

import time


class FrameLoop:
    """render(buffer) -> (rays, kernel_ms) fills `buffer` with this rank's share of a frame and returns when it is there.
    `dist` is torch.distributed (or None for one rank); `device_sync` blocks until the device has finished everything
    issued so far (torch.cuda.synchronize on a GPU, a no-op on CPU tensors); `finish_sample_shards(buffer)` turns summed
    raw per-pixel sums into the frame on rank 0 (sample-pass sharding only)."""

    def __init__(self, dist, world, rank, buffers, render, device_sync, sample_shard=False, finish_sample_shards=None):
        if world > 1 and dist is None:
            raise ValueError("world > 1 needs a torch.distributed backend")
        if world > 1 and not sample_shard and len(buffers) < 2:
            raise ValueError("tile sharding over several ranks renders into two buffers in turn")
        self.dist, self.world, self.rank = dist, world, rank
        self.buffers = list(buffers)
        self.render = render
        self.device_sync = device_sync
        self.sample_shard = sample_shard
        self.finish_sample_shards = finish_sample_shards
        self.pending = [None] * len(self.buffers)
        self.frame_no = 0

    def _finish(self, slot):
        """The reduce issued on the buffer in `slot` has completed: it may be rendered into again."""
        if self.pending[slot] is not None:
            self.pending[slot].wait()
            self.device_sync()
            self.pending[slot] = None

    def step(self):
        slot = self.frame_no % len(self.buffers)
        self.frame_no += 1
        fb = self.buffers[slot]
        self._finish(slot)  # the reduce of two frames ago
        rays, kernel_ms = self.render(fb)
        if self.world > 1:
            if self.sample_shard:
                self.dist.reduce(fb, dst=0)
                if self.rank == 0 and self.finish_sample_shards is not None:
                    self.finish_sample_shards(fb)
                self.device_sync()
            else:
                self.pending[slot] = self.dist.reduce(fb, dst=0, async_op=True)
        return rays, kernel_ms

    def drain(self):
        for slot in range(len(self.buffers)):
            self._finish(slot)

    def last_frame(self):
        return self.buffers[(self.frame_no - 1) % len(self.buffers)]

    def run(self, warmup, steps):
        """`warmup` untimed frames, then exactly `steps` timed ones between barrier + device sync on both sides; every
        frame of the timed region is assembled on rank 0 before the clock stops.  Returns this rank's (elapsed seconds,
        rays, [kernel ms per step])."""
        for _ in range(warmup):
            self.step()
        self.drain()
        if self.world > 1:
            self.dist.barrier()
        self.device_sync()
        t0 = time.perf_counter()
        rays_local, kernel_ms = 0, []
        for _ in range(steps):
            r, k = self.step()
            rays_local += r
            kernel_ms.append(k)
        self.drain()
        if self.world > 1:
            self.dist.barrier()
        self.device_sync()
        return time.perf_counter() - t0, rays_local, kernel_ms


def job_totals(dist, world, elapsed, rays_local, device):
    """max over ranks of the elapsed time, sum over ranks of the rays (tensors on `device`: the ranks' GPUs under RCCL,
    "cpu" under gloo)."""
    if world == 1:
        return elapsed, rays_local
    import torch

    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    rr = torch.tensor([rays_local], dtype=torch.int64, device=device)
    dist.all_reduce(rr)
    return float(t.item()), int(rr.item())


def frame_size(base_nx, base_ny, world, fixed_frame):
    """Weak scaling (the default): the pixel count grows with the rank count at fixed aspect; strong scaling and
    sample-pass sharding keep the frame."""
    import math

    s = 1.0 if fixed_frame else math.sqrt(world)
    return int(round(base_nx * s)), int(round(base_ny * s))


def share_through_file(dist, rank, path, export_size, export_into, import_from, size_device="cpu"):
    """One build per node: rank 0 writes a block of `export_size()` bytes into the file `path` (in /dev/shm: host memory,
    mapped by every rank) through `export_into(memmap)`, the other ranks read it through `import_from(memmap)`.  The
    broadcast of the size doubles as the barrier that says the file is complete.  The file pins host memory for as long
    as it has a name, so rank 0 unlinks it as soon as every rank holds its mapping (the pages live on until the last
    mapping closes) -- BEFORE the imports, where uploads can fail -- and also on the way out of any exception, a SIGTERM
    from the launcher (another rank died) included."""
    import os
    import signal
    import threading

    import numpy as np
    import torch

    def on_term(signum, frame):
        raise SystemExit(128 + signum)

    old_handler = None
    if rank == 0 and threading.current_thread() is threading.main_thread():
        old_handler = signal.signal(signal.SIGTERM, on_term)
    size = torch.zeros(1, dtype=torch.int64)
    try:
        if rank == 0:
            size[0] = int(export_size())
            mm = np.memmap(path, dtype=np.uint8, mode="w+", shape=(int(size[0]),))
            export_into(mm)
            mm.flush()
            del mm
        size = size.to(size_device)
        dist.broadcast(size, src=0)
        size = size.cpu()
        mapping = None
        if rank != 0:
            mapping = np.memmap(path, dtype=np.uint8, mode="r", shape=(int(size[0]),))
        dist.barrier()  # every rank has the file mapped
        if rank == 0:
            os.unlink(path)
        failure = None
        if rank != 0:
            try:
                import_from(mapping)
            except Exception as e:  # noqa: BLE001 -- the other ranks wait at the barrier below: reach it, then raise
                failure = e
            del mapping
        dist.barrier()
        if failure is not None:
            raise failure
    finally:
        if rank == 0 and os.path.exists(path):
            os.unlink(path)
        if old_handler is not None:
            signal.signal(signal.SIGTERM, old_handler)
    return int(size[0])
