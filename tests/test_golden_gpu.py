"""-m gpu: the HIP path reproduces the committed golden vectors (tests/golden/*.npz, produced in
the build container by executing the reference's own function source).  Nothing here reads
/root/reference.  Bar: reset / progress / ids bit-exact, fp32 obs and rewards within
1e-5 rel / 1e-6 abs (north_star)."""
import os

import pytest
import torch

import golden_io
from helpers import assert_close, assert_equal, make_task
from omniisaacgymenvs_amd.sim import workloads

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,name", [("ant", "Ant"), ("humanoid", "Humanoid")])
@pytest.mark.parametrize("fused", [True, False])
def test_locomotion_hip_reproduces_golden(kind, name, fused):
    for path in golden_io.files(kind):
        g = golden_io.load(path)
        n = g["in_pos"].shape[0]
        env, task = make_task(name, n)
        task.fused = fused
        dev = task.device
        # the golden vectors carry their own DOF limit / gear tables
        task.dof_limits_lower, task.dof_limits_upper = g["lo"].to(dev), g["hi"].to(dev)
        task.joint_gears, task.motor_effort_ratio = g["gears"].to(dev), g["motor_effort_ratio"].to(dev)
        task._refresh_host_consts()
        st = {k[3:]: v for k, v in g.items() if k.startswith("in_")}
        workloads.load_locomotion_state(task, st)
        obs, rew, reset, _ = task.post_physics_step()
        torch.cuda.synchronize()
        tag = f"{os.path.basename(path)} fused={fused}"
        assert_equal(reset, g["out_reset"], f"{tag} reset")
        assert_equal(task.progress_buf, g["out_progress"], f"{tag} progress")
        assert_equal(task.prev_potentials, g["out_prev_potentials"], f"{tag} prev_potentials")
        assert_equal(task.potentials, g["out_potentials"], f"{tag} potentials")
        assert_close(obs, g["out_obs"], f"{tag} obs")
        assert_close(task.up_vec, g["out_up_vec"], f"{tag} up_vec")
        assert_close(task.heading_vec, g["out_heading_vec"], f"{tag} heading_vec")
        assert_close(rew, g["out_rew"], f"{tag} rew")  # every row: the golden files hold up_proj = 0.93 +- 1 ulp rows on purpose


@pytest.mark.parametrize("fused", [True, False])
def test_cartpole_hip_reproduces_golden(fused):
    for path in golden_io.files("cartpole"):
        g = golden_io.load(path)
        n = g["in_dof_pos"].shape[0]
        env, task = make_task("Cartpole", n)
        task.fused = fused
        dev = task.device
        task._cartpoles.joint_positions.copy_(g["in_dof_pos"].to(dev))
        task._cartpoles.joint_velocities.copy_(g["in_dof_vel"].to(dev))
        task.progress_buf.copy_(g["in_progress"].to(dev))
        task.reset_buf[:] = 1
        obs, rew, reset, _ = task.post_physics_step()
        torch.cuda.synchronize()
        assert_equal(obs, g["out_obs"], "obs")
        assert_equal(rew, g["out_rew"], "rew")
        assert_equal(reset, g["out_reset"], "reset")
        assert_equal(task.progress_buf, g["out_progress"], "progress")
        # reset_idx on the golden ids with the golden draws (kernel takes pre-drawn uniforms)
        import ctypes as C
        from omniisaacgymenvs_amd import _lib
        ids, u = g["reset_ids"].to(dev), g["reset_u"].to(dev).contiguous()
        k = ids.numel()
        o = task._reset_out
        o["env_ids"][:k] = ids
        o["count"].fill_(k)
        p, b, r = task._params(), task._buffers(), task._reset_io(u)
        _lib.check(_lib.load().oige_cartpole_reset_idx(C.byref(p), C.byref(b), C.byref(r), n, k,
                                                       _lib.current_stream(torch.device(dev))), "reset_idx")
        torch.cuda.synchronize()
        assert_equal(o["dof_pos"][:k], g["reset_dof_pos"], "reset dof_pos")
        assert_equal(o["dof_vel"][:k], g["reset_dof_vel"], "reset dof_vel")
        assert_equal(task.reset_buf, g["reset_reset_buf"], "reset_buf after reset")
        assert_equal(task.progress_buf, g["reset_progress_buf"], "progress_buf after reset")
