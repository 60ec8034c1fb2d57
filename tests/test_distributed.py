"""CPU, world_size 2 over gloo: the patch-sharded mode's host / exchange logic -- the tile-to-rank layout with its gather
groups, the asynchronous exchange of the per-rank prediction stacks (all-gather or gather-to-rank-0) and the permutation back
to tile order reproduce the single-process ordering; the tile plan tensor is rank 0's on every rank; Tester's cross-rank
result collection keeps the dataset order.  (The sharded FORWARD itself is emulated on one GPU in tests/test_hip_models.py
and tests/test_headline_parity.py.)"""
import os
import random

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


def _plan(h, mode, pn=4, raw=(2160, 3840), split=(4, 4)):
    passes = h.plan_tiles(h.prepare_tile_cfg(list(raw), list(split)), mode, pn)
    return [p["kind"] for p in passes], [len(p["raw"]) for p in passes], passes


def _worker(rank, world, port, mode, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        h = _Host()
        h.patch_process_shape = (448, 448)
        random.seed(621 + rank)          # every rank draws ANOTHER plan ...
        kinds, counts, passes = _plan(h, mode)
        n_all = sum(counts)
        plan_t = torch.tensor([t for p in passes for t in p["raw"]] + [t for p in passes for t in p["proc"]], dtype=torch.int32).view(-1, 2)
        mine_before = plan_t.clone()
        plan_t = h._sync_plan_tensor(plan_t)   # ... and rank 0's overrules
        both = [torch.zeros_like(plan_t) for _ in range(world)]
        dist.all_gather(both, plan_t)
        ok = all(torch.equal(b, both[0]) for b in both)
        if rank == 0:
            ok = ok and torch.equal(plan_t, mine_before)
        h.patch_process_shape = (4, 6)
        full = torch.arange(n_all * 24, dtype=torch.float32).view(n_all, 1, 4, 6)   # "prediction" of tile i
        for dst in (None, 0):
            groups = h.shard_layout(kinds, counts, world, dst)
            assert sum(g["n"] for g in groups) == n_all
            got = []
            for gi, g in enumerate(groups):
                stack = torch.zeros((g["per"], 1, 4, 6))
                for sl, i in enumerate(g["mine"][rank]):
                    stack[sl] = full[g["base"] + i]
                allp = h._exchange_begin(stack, (rank, world), dst, gi)()      # async collective + wait
                if allp is not None:
                    got.append(allp.view(world * g["per"], 1, 4, 6).index_select(0, torch.tensor(g["perm"])))
            if dst is None or rank == dst:
                ok = ok and torch.equal(torch.cat(got), full)
            else:
                ok = ok and not got
        flag = torch.tensor([1 if ok else 0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        # Tester's cross-rank collection (tester.py:124-127, collect_results_gpu): frame f ran on rank f % world
        from patchrefinerv2_amd.tester import collect_results
        n = 5
        part = [dict(name=f"f{i}", rank=rank) for i in range(rank, n, world)]
        allr = collect_results(part, n)
        if rank == 0:
            ok2 = [r["name"] for r in allr] == [f"f{i}" for i in range(n)] and [r["rank"] for r in allr] == [i % world for i in range(n)]
            out.put(int(flag.item()) * int(ok2))
        else:
            assert allr is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["r32", "m1", "r8"])
def test_patch_shard_exchange_world2(mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500) + len(mode) + ord(mode[-1])
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=5) == 1


@pytest.mark.parametrize("mode,n", [("r32", 81), ("r64", 113), ("r128", 177), ("m2", 49), ("m1", 16), ("r3", 49)])
def test_shard_layout_covers_every_tile_once(mode, n):
    h = _Host()
    h.patch_process_shape = (448, 448)
    random.seed(621)
    kinds, counts, _ = _plan(h, mode)
    assert sum(counts) == n
    for world in (1, 2, 4, 8):
        for dst in (None, 0, world - 1):
            groups = h.shard_layout(kinds, counts, world, dst)
            two = mode[0] == "r" and int(mode[1:]) >= 4
            if two and world >= 8:   # from 8 ranks on the two groups merge while the first has < 8 tiles per rank (49 fixed tiles: 6 each)
                two = sum(c for k, c in zip(kinds, counts) if k != "random") >= h.SHARD_MERGE_BELOW * world
            assert len(groups) == (2 if two else 1)
            seen = []
            for g in groups:
                assert sorted(i for m in g["mine"] for i in m) == list(range(g["n"]))          # every tile exactly once
                assert sorted(g["perm"]) == sorted(q * g["per"] + s for q in range(world) for s in range(len(g["mine"][q])))
                assert all(len(g["mine"][q]) == g["share"][q] <= g["per"] for q in range(world))
                others = [g["share"][q] for q in range(world) if q != dst]
                if others:
                    assert max(others) - min(others) <= 1
                    if dst is not None and world > 1:                # the blending rank never computes more; fewer in the last group
                        assert g["share"][dst] <= min(others) and g["share"][dst] >= min(others) - 3
                        if g is groups[-1] and g["n"] >= 2 * world:
                            assert g["share"][dst] < max(others)
                seen += [g["base"] + i for m in g["mine"] for i in m]
            assert sorted(seen) == list(range(n))


@pytest.mark.parametrize("mode,n", [("r32", 81), ("r64", 113), ("m1", 16)])
def test_shard_layout_with_a_rotating_coarse_owner(mode, n):
    """round 6: the rank that computes the NEXT frame's coarse pyramid beside this frame's tiles (models._prefetch_coarse_sharded) hands tiles of the last
    gather group to the others (shard_owner_cost tile-times); every tile is still computed exactly once, whoever the owner is, also when the owner is
    the blending rank; without an owner the layout is the round-5 one"""
    h = _Host()
    h.patch_process_shape = (448, 448)
    random.seed(621)
    kinds, counts, _ = _plan(h, mode)
    for world in (2, 4, 8):
        base = h.shard_layout(kinds, counts, world, 0)
        assert h.shard_layout(kinds, counts, world, 0, None) is base                  # (cached; owner None == the old call)
        for owner in range(world):
            groups = h.shard_layout(kinds, counts, world, 0, owner)
            seen = []
            for g in groups:
                assert sorted(i for m in g["mine"] for i in m) == list(range(g["n"]))
                assert all(len(g["mine"][q]) == g["share"][q] <= g["per"] for q in range(world))
                seen += [g["base"] + i for m in g["mine"] for i in m]
            assert sorted(seen) == list(range(n))
            tiles = [sum(g["share"][q] for g in groups) for q in range(world)]
            others = [tiles[q] for q in range(world) if q not in (0, owner)]
            if n >= 4 * world and others:
                assert tiles[owner] <= min(others), (mode, world, owner, tiles)       # the owner never computes more tiles than a rank without side work
                if owner != 0:
                    assert tiles[owner] < max(others) or groups[-1]["share"][owner] == 0, (mode, world, owner, tiles)


def test_boxes_on_device_equal_host_boxes():
    """_boxes_dev (patch-sharded mode: boxes from the broadcast plan tensor) == _boxes (host numpy) bit for bit"""
    h = _Host()
    for pps, raw, split in (((384, 512), (2160, 3840), (4, 4)), ((448, 448), (2160, 3840), (4, 4)), ((392, 518), (1080, 1920), (2, 2))):
        h.patch_process_shape = pps
        tc = h.prepare_tile_cfg(list(raw), list(split))
        random.seed(5)
        tiles = [(random.randint(0, raw[0] - tc["patch_raw_shape"][0] - 1), random.randint(0, raw[1] - tc["patch_raw_shape"][1] - 1)) for _ in range(300)]
        want = torch.from_numpy(h._boxes(tiles, tc))
        got = h._boxes_dev(torch.tensor(tiles, dtype=torch.int32), tc)
        assert torch.equal(got, want)


def _run_bench(args, env_extra=None, drop=()):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT") + tuple(drop)}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bench_gpus_n_launches_itself():
    """``python bench.py --gpus 2`` started the way the driver starts ``--gpus 1`` (no launcher, no WORLD_SIZE) must run TWO ranks
    (docs/user_infer.md:124-129: the reference's launcher contract) -- the parent spawns torch.distributed.run as a child before any GPU
    call and relays rank 0's line; gloo + a stub model here (no GPU in this container)"""
    r, line = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--backend", "gloo", "--stub-model", "--no-roofline", "--no-cpu-baseline"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert line is not None and line["n_gpus"] == 2 and line["config"]["ranks_seen"] == 2 and line["steps"] == 3, r.stdout


def test_bench_refuses_a_world_size_mismatch():
    """a launcher that started another number of ranks than ``--gpus`` says: refuse loudly in every case (also WORLD_SIZE=1 vs --gpus 2)"""
    for ws, gpus in (("1", "2"), ("2", "1")):
        r, line = _run_bench(["--gpus", gpus, "--backend", "gloo", "--stub-model"], env_extra=dict(WORLD_SIZE=ws, RANK="0", LOCAL_RANK="0"))
        assert r.returncode != 0 and line is None and "WORLD_SIZE" in r.stderr, (r.stdout, r.stderr[-500:])


def test_bench_child_failure_is_reported():
    """a rank that fails makes the self-launching parent exit non-zero"""
    r, line = _run_bench(["--gpus", "2", "--backend", "gloo", "--stub-model", "--workload", "no_such_workload", "--steps", "-1"],
                         env_extra=dict(PRV2_BENCH_STUB_FAIL="1"))
    assert r.returncode != 0, (r.stdout, r.stderr[-500:])
