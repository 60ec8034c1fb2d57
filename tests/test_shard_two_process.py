"""The REAL patch-sharded forward in two OS processes on the one GPU: ``_infer_sharded`` un-patched -- the plan tensor's
broadcast, the asynchronous per-group exchange (``_exchange_begin``: all_gather_into_tensor / gather with async_op, every Work handle
waited for), stream ordering and buffer lifetimes -- over the gloo backend, which carries CUDA tensors (RCCL refuses two ranks on one
device).  Each rank is a fresh child (torch.multiprocessing spawn: no process that touched the GPU is ever re-executed); the sharded
frame must be bit-identical to the unsharded frame computed by the same child.  Rank 1 seeds ``random`` differently: only rank 0's
broadcast plan can make the ranks agree.  SURVEY.md 8e; reference launcher contract docs/user_infer.md:124-129."""
import os
import random
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    torch.set_grad_enabled(False)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.cases import E2E_V1, E2E_V2, e2e_v1_sd, e2e_v2_sd, rand_image
        from test_hip_models import _build
        ok, notes = True, []
        # (the third run: the f16f6 arithmetic's range guard all-reduces its table over the ranks after every sharded frame)
        for kind, c, sd, mode, prec in (("PatchRefinerPlus", E2E_V2, e2e_v2_sd(), "r4", "bf16x3"), ("PatchRefiner", E2E_V1, e2e_v1_sd(), "r8", "bf16x3"),
                                        ("PatchRefinerPlus", E2E_V2, e2e_v2_sd(), "r4", "f16f6")):
            m = _build(kind, c, sd, prec=prec, n_streams=2, max_batch=3)
            hr = rand_image(c["seed"], 1, *c["raw"]).to("cuda")
            lr = m.resizer(hr)
            tc = dict(image_raw_shape=c["raw"], patch_split_num=c["split"])

            def run(seed, **kw):
                random.seed(seed)
                return m(mode="infer", cai_mode=mode, process_num=4, tile_cfg=tc, image_lr=lr, image_hr=hr, **kw)[0]

            full = run(621)                                             # the unsharded frame on rank 0's random draws
            for dst in (None, 0):                                       # --gather all / --gather rank0
                for rep in range(2):                                    # (twice: buffers of the previous frame's exchange are reused / freed)
                    got = run(621 + 1000 * rank, shard=(rank, world), gather_dst=dst, next_image_lr=lr if rep == 0 else None)
                    if dst is None or rank == dst:
                        same = got is not None and torch.equal(got, full)
                    else:
                        same = got is None
                    ok = ok and same
                    notes.append((kind, mode, prec, dst, rep, bool(same)))
            assert len(m.last_shard_layout) == 2                        # [init + grids | random tiles]: two exchanges per frame
            if prec == "f16f6":  # the guard (and its all-reduce over the ranks) really ran: the model was built with device='cuda', the frames ask 'cuda:0'
                from patchrefinerv2_amd import ops
                has_f6 = bool(getattr(m.refiner_fusion_model, "f16f6", False)) and ops.F6Range.active("cuda")
                assert not has_f6 or getattr(m, "f6_guarded_frames", 0) >= 5, (has_f6, getattr(m, "f6_guarded_frames", 0))
                notes.append(("f6 guard", has_f6, getattr(m, "f6_guarded_frames", 0)))
        # ---- a frame LOOP with the next frame announced: from the second announcement on ONE rank computes the next frame's coarse pyramid + tap
        # tables (the owner rotates with the frame index) and broadcasts them; the others receive.  Five different frames, each bit-equal to the
        # unsharded frame; the coarse forwards this rank ran are counted.
        for kind, c, sd, mode, prec in (("PatchRefinerPlus", E2E_V2, e2e_v2_sd(), "r4", "bf16x3"), ("PatchRefinerPlus", E2E_V2, e2e_v2_sd(), "r4", "f16f6")):
            m = _build(kind, c, sd, prec=prec, n_streams=2, max_batch=3)
            tc = dict(image_raw_shape=c["raw"], patch_split_num=c["split"])
            hrs = [rand_image(c["seed"] + 10 * i, 1, *c["raw"]).to("cuda") for i in range(5)]
            lrs = [m.resizer(h) for h in hrs]
            fulls = []
            for i in range(5):
                random.seed(700 + i)
                fulls.append(m(mode="infer", cai_mode=mode, process_num=4, tile_cfg=tc, image_lr=lrs[i], image_hr=hrs[i])[0])
            calls = [0]
            orig = m.coarse_forward

            def counted(x, _o=orig, _c=calls):
                _c[0] += 1
                return _o(x)
            m.coarse_forward = counted
            owners = []
            for i in range(5):
                random.seed(700 + i + 1000 * rank)
                got = m(mode="infer", cai_mode=mode, process_num=4, tile_cfg=tc, image_lr=lrs[i], image_hr=hrs[i], shard=(rank, world),
                        next_image_lr=lrs[i + 1] if i + 1 < 5 else None)[0]
                same = torch.equal(got, fulls[i])
                ok = ok and same
                owners.append(m.last_coarse_owner)
                notes.append((kind, mode, prec, "loop", i, m.last_coarse_owner, bool(same)))
            # frame 0: its own coarse + the local prefetch of frame 1 (recipe recorded); frames 1..3 announce frames 2..4: owners (i + 1) % 2
            assert owners == [None, 0, 1, 0, None], owners
            assert calls[0] == 2 + sum(1 for o in owners if o == rank), (calls[0], owners)
            assert getattr(m, "last_coarse_bcast_bytes", 0) > 0
        flag = torch.tensor([1 if ok else 0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        q.put((rank, int(flag.item()), notes))
    finally:
        dist.destroy_process_group()


def test_sharded_forward_in_two_processes_on_one_gpu():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 200)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0, "a rank failed (see its traceback above)"
    res = sorted(q.get(timeout=10) for _ in range(2))
    assert [r[1] for r in res] == [1, 1], res
