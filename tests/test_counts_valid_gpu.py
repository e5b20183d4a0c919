"""-m gpu: the tile counts a post kernel's DONE stage leaves (OIGE_PRE_COUNTS_VALID) give the next pre_physics_step exactly
what its own counting launch would have: Crazyflie and Ingenuity, whose second list (reset | progress % 500 == 0) depends on
progress_buf as well; and the shortcut is dropped whenever one of the flag buffers may have moved."""
import pytest
import torch

from helpers import make_task
from omniisaacgymenvs_amd import _lib
from omniisaacgymenvs_amd.sim import workloads

pytestmark = pytest.mark.gpu


def _task(name, n):
    env, task = make_task(name, n, seed=5)
    if name == "Crazyflie":
        st = workloads.crazyflie_state(n, task._env_pos.cpu(), device="cpu", seed=9)
        st["progress"][: n // 8] = 499  # -> 500 after the post step: on the target list without being reset
        workloads.load_crazyflie_state(task, st)
        acts = torch.rand(n, 4, generator=torch.Generator().manual_seed(3)) * 2 - 1
    else:
        st = workloads.ingenuity_state(n, task._env_pos.cpu(), device="cpu", seed=9)
        st["progress"][: n // 8] = 499
        workloads.load_ingenuity_state(task, st)
        acts = torch.rand(n, 6, generator=torch.Generator().manual_seed(3)) * 2 - 1
    for v in (task._copters, task._balls):
        v.apply_resets = False
    return task, acts.to(task.device)


@pytest.mark.parametrize("name", ["Crazyflie", "Ingenuity"])
@pytest.mark.parametrize("n", [300, 5000])
def test_post_kernel_counts_equal_a_recount(name, n):
    outs = []
    for recount in (False, True):
        torch.manual_seed(17)  # Crazyflie draws its thrust noise from torch's generator
        task, acts = _task(name, n)
        task.post_physics_step()
        assert task._counts_valid(task.reset_buf, task.progress_buf)
        if recount:
            task._counts_key = None
        task.pre_physics_step(acts)
        assert not task._counts_valid(task.reset_buf, task.progress_buf)  # the pre kernel cleared flags: counts are stale
        torch.cuda.synchronize()
        o = task._reset_out
        k, kt = int(o["count"].item()), int(o["target_count"].item())
        assert kt >= k > 0 and kt > n // 16
        outs.append([o["env_ids"][:k].clone(), o["target_ids"][:kt].clone(), task.reset_buf.clone(), task.progress_buf.clone(),
                     task.thrusts.clone(), task.dof_vel.clone(), task.target_positions.clone(), o["ball_pos"][:kt].clone()])
    for x, y in zip(*outs):
        assert torch.equal(x, y)


@pytest.mark.parametrize("name", ["Crazyflie", "Ingenuity"])
def test_shortcut_is_dropped_when_a_flag_buffer_moves(name):
    task, acts = _task(name, 512)
    task.post_physics_step()
    task.progress_buf += 1  # a torch write: the version counter moves
    assert not task._counts_valid(task.reset_buf, task.progress_buf)
    task.post_physics_step()
    task._post(_lib.STAGE_PROGRESS)  # a raw-pointer write to progress_buf
    assert not task._counts_valid(task.reset_buf, task.progress_buf)
