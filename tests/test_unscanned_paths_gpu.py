"""-m gpu: a pre kernel gets its group offsets from the byte table of the tile counts (the mirrors always pass it), from the int32
counts (callers without the byte table, up to 512 groups) or from the scan launch (beyond): the three give the same step.
Callers of the C ABI may leave the byte tables NULL, so those two paths stay covered here."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (the adapters build a task with a loaded synthetic state)
from helpers import assert_equal  # noqa: E402
from test_graph_step_gpu import _tensors, _views  # noqa: E402

pytestmark = pytest.mark.gpu
U8_FIELDS = ("tile_counts_u8", "reset_counts_u8", "target_counts_u8")
TASKS = ["Humanoid", "Anymal", "Quadcopter", "BallBalance", "Ingenuity", "FrankaCabinet", "TomoSPCCabinet", "TomoDJnewhandTendonCabinet"]


def _build(name, n, with_u8):
    torch.manual_seed(7)
    b = bench.ADAPTERS[name](name, n, "cuda:0", 11)
    for v in _views(b.task).values():
        v.apply_resets = True
    if not with_u8:
        inner = b.task._buffers

        def no_u8(*a, **k):
            s = inner(*a, **k)
            s = type(s).from_buffer_copy(s)  # a copy: the mirror keeps (and may re-check) its own block
            cleared = 0
            for f in U8_FIELDS:
                if hasattr(s, f):
                    setattr(s, f, None)
                    cleared += 1
            assert cleared > 0
            return s

        b.task._buffers = no_u8
    return b


@pytest.mark.parametrize("name", TASKS)
@pytest.mark.parametrize("n", [300, 140_000])  # without the byte table: 2 groups -> int32 counts; 547 groups -> the scan launch
def test_step_is_the_same_without_the_byte_table(name, n):
    a, b = _build(name, n, True), _build(name, n, False)
    g = torch.Generator().manual_seed(5)
    resets = 0
    for step in range(3):
        act = (torch.rand(n, a.task.num_actions, generator=g) * 2 - 1).to("cuda:0")
        for t in (a.task, b.task):
            torch.manual_seed(100 + step)  # (tasks that draw from torch's generator see the same stream)
            t.pre_physics_step(act)
            t.post_physics_step()
        torch.cuda.synchronize()
        resets += int(a.task._reset_out["count"].item())
        A_, B_ = _tensors(a.task), _tensors(b.task)
        for key in A_:
            if key in ("_tile_counts_u8", "_reset_counts_u8", "_target_counts_u8", "_reset_workspace", "_target_workspace"):
                continue  # (the group-offset slots of the workspace are only filled by the scan)
            assert_equal(B_[key], A_[key], f"{name} step {step}: {key}")
    assert resets > 0  # the compaction was exercised
