"""CPU, world_size 2 over gloo: the patch-sharded mode's exchange step (all-gather or gather-to-rank-0 of the
per-rank prediction stacks + reorder to tile order) reproduces the single-process ordering, and the tile plan
is rank 0's on every rank.  (The sharded FORWARD itself is emulated on one GPU in tests/test_hip_models.py.)"""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from patchrefinerv2_amd import models as M


class _Host(M._PatchModel):
    def __init__(self):
        super().__init__()
        self.patch_process_shape = (4, 6)

    def _pack(self):
        pass


def _worker(rank, world, port, n_tiles, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.arange(n_tiles * 24, dtype=torch.float32).view(n_tiles, 1, 4, 6)
        mine = full[rank::world].contiguous()  # tile i -> rank i mod world
        h = _Host()
        got = h._gather_predictions(mine, n_tiles, (rank, world))              # all-gather: every rank gets the frame
        ok = torch.equal(got, full)
        got0 = h._gather_predictions(mine, n_tiles, (rank, world), dst=0)      # gather-to-rank-0: only rank 0 does
        ok = ok and ((torch.equal(got0, full)) if rank == 0 else got0 is None)
        # the tile plan that counts is rank 0's: a rank whose ``random`` state differs is overruled
        import random
        random.seed(621 + rank)
        h.patch_process_shape = (448, 448)
        plan = h._sync_plan(h.plan_tiles(h.prepare_tile_cfg([2160, 3840], [4, 4]), "r8", 4))
        import zlib
        sig = torch.tensor([zlib.crc32(str(plan).encode())])
        both = [torch.zeros_like(sig) for _ in range(world)]
        dist.all_gather(both, sig)
        ok = ok and all(int(b) == int(both[0]) for b in both)
        flag = torch.tensor([1 if ok else 0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if rank == 0:
            out.put(int(flag.item()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_tiles", [81, 16, 5])
def test_patch_shard_gather_world2(n_tiles):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500) + n_tiles
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_tiles, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) == 1


def test_shard_assignment_covers_every_tile_once():
    import random
    h = _Host()
    h.patch_process_shape = (448, 448)
    tc = h.prepare_tile_cfg([2160, 3840], [4, 4])
    random.seed(621)
    flat = [t for p in h.plan_tiles(tc, "r32", 4) for t in p["raw"]]
    for world in (2, 4, 8):
        parts = [list(range(len(flat)))[r::world] for r in range(world)]
        assert sorted(i for p in parts for i in p) == list(range(81))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
