"""-m gpu: the task mirrors keep their ctypes argument blocks between calls (RLTask._block), keyed on the identity of the
tensors the blocks point into and on the value of their scalars.  A rebound buffer or an edited scalar must therefore take
effect on the NEXT call by itself, on every mirror: task B rebinds obs_buf / rew_buf after two steps and must end up where the
untouched task A does, with the old tensors left alone."""
import pytest
import torch

from helpers import assert_equal, make_task

pytestmark = pytest.mark.gpu
TASKS = ["Cartpole", "Ant", "Humanoid", "ShadowHand", "AllegroHand", "Anymal", "AnymalTerrain", "Quadcopter", "BallBalance",
         "Ingenuity", "Crazyflie", "FrankaCabinet", "TomoSPCCabinet", "TomoDJnewhandCabinet", "TomoDJnewhandTendonCabinet"]


def _run(name, n, rebind):
    torch.manual_seed(5)
    env, task = make_task(name, n)
    env.reset()
    g = torch.Generator().manual_seed(6)
    old = None
    for step in range(4):
        torch.manual_seed(100 + step)   # Crazyflie draws its thrust noise with torch
        actions = (torch.rand(n, task.num_actions, generator=g) * 2 - 1).to(task.rl_device)
        if rebind and step == 2:
            old = (task.obs_buf, task.rew_buf, task.obs_buf.clone(), task.rew_buf.clone())
            task.obs_buf = torch.full_like(task.obs_buf, -7.0)
            task.rew_buf = torch.full_like(task.rew_buf, -7.0)
        task.pre_physics_step(actions)
        task.post_physics_step()
    torch.cuda.synchronize()
    return task, old


@pytest.mark.parametrize("name", TASKS)
def test_rebound_buffers_take_effect_on_the_next_call(name):
    n = 300
    a, _ = _run(name, n, rebind=False)
    b, old = _run(name, n, rebind=True)
    assert_equal(b.obs_buf, a.obs_buf, "obs_buf (rebound)")
    assert_equal(b.rew_buf, a.rew_buf, "rew_buf (rebound)")
    assert_equal(b.progress_buf, a.progress_buf, "progress_buf")
    obs_old, rew_old, obs_snap, rew_snap = old
    assert_equal(obs_old, obs_snap, "the old obs_buf is left alone")
    assert_equal(rew_old, rew_snap, "the old rew_buf is left alone")


@pytest.mark.parametrize("name", ["Cartpole", "Quadcopter", "FrankaCabinet", "Anymal"])
def test_edited_scalar_takes_effect_on_the_next_call(name):
    """max_episode_length is a scalar of every params block: shortened to 2, every env must be flagged at the next step."""
    n = 200
    torch.manual_seed(7)
    env, task = make_task(name, n)
    env.reset()
    actions = torch.zeros(n, task.num_actions, device=task.rl_device)
    for _ in range(2):
        task.pre_physics_step(actions)
        task.post_physics_step()
    task._max_episode_length = 2
    if hasattr(task, "max_episode_length"):
        task.max_episode_length = 2
    task.pre_physics_step(actions)
    task.post_physics_step()
    torch.cuda.synchronize()
    assert int((task.reset_buf != 0).sum()) == n
