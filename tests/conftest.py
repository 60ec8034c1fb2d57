import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load


class ShardEmulation:
    """The patch-sharded forward on ONE GPU: ``record()`` -- run ``shard=(r, world)`` for every r, keeping the stack each rank
    would send per gather group; ``deliver()`` -- hand a rank the rank-major concatenation RCCL's gather / all-gather would
    deliver (models._PatchModel._exchange contract)."""

    def __init__(self, model, world):
        self.model, self.world, self.stacks = model, world, {}

    def record(self):
        def rec(mine, shard, dst, group=0):
            self.stacks[(shard[0], group)] = mine.clone()
            return None                                   # "this rank does not receive": forward returns depth None
        self.model._exchange = rec

    def deliver(self):
        import torch

        def dlv(mine, shard, dst, group=0):
            assert torch.equal(mine, self.stacks[(shard[0], group)])      # deterministic per-rank work
            return torch.cat([self.stacks[(r, group)] for r in range(self.world)], dim=0)
        self.model._exchange = dlv

    def restore(self):
        self.model.__dict__.pop("_exchange", None)
