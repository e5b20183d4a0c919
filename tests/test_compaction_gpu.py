"""-m gpu: ordered stream compaction == reset_buf.nonzero().squeeze(-1), bit-exact, incl. the
empty / all-set / ragged-tail cases the reference can hit."""
import pytest
import torch

from omniisaacgymenvs_amd import _lib

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 63, 64, 65, 255, 256, 257, 1000, 70001, 1 << 20])
@pytest.mark.parametrize("frac", [0.0, 0.001, 0.5, 1.0])
def test_compact_matches_nonzero(n, frac):
    dev = "cuda:0"
    g = torch.Generator().manual_seed(n)
    flags = (torch.rand(n, generator=g) < frac).long()
    if frac == 0.5:
        flags = flags * torch.randint(-3, 4, (n,), generator=g)  # any non-zero value counts
    flags = flags.to(dev)
    ws = _lib.new_workspace(n, dev)
    ids = torch.full((n,), -1, dtype=torch.int64, device=dev)
    count = torch.full((1,), -1, dtype=torch.int64, device=dev)
    _lib.compact(flags, ws, ids, count)
    torch.cuda.synchronize()
    want = flags.nonzero(as_tuple=False).squeeze(-1)
    k = int(count.item())
    assert k == want.numel()
    assert torch.equal(ids[:k], want)
    assert bool((ids[k:] == -1).all())


def test_group_scan_ignores_the_padding_slots_of_the_last_group():
    """the post kernels write the tile counts of real tiles only; the scan must not depend on the padding slots
    of the last 256-env group having been zeroed (257 envs: 5 tiles, 2 groups, 3 padding slots)."""
    from helpers import make_task
    from omniisaacgymenvs_amd.sim import workloads

    n = 257
    env, task = make_task("Ant", n)
    st = workloads.locomotion_state(n, task.dof_limits_lower.cpu(), task.dof_limits_upper.cpu(), 4, device="cpu")
    workloads.load_locomotion_state(task, st)
    task._reset_workspace[5:8] = 99  # poison the padding
    task.post_physics_step()
    want = task.reset_buf.nonzero().squeeze(-1)
    assert want.numel() > 0
    task.pre_physics_step(st["actions"].to(task.device))
    torch.cuda.synchronize()
    k = int(task._reset_out["count"])
    assert k == want.numel() and torch.equal(task._reset_out["env_ids"][:k], want)
