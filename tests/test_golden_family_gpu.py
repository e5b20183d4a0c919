"""-m gpu: the HIP path (task classes -> C ABI) against the golden vectors that oracle/gen_golden.py produced by running the
reference's own method source: in-hand family (ShadowHand, AllegroHand), Anymal (flat), AnymalTerrain post_physics_step.
Bar: flags / counters bit-exact, fp32 within 1e-5 rel / 1e-6 abs; envs whose decision compare sits within 1e-6 of its
threshold are excluded (as in test_hand_gpu)."""
import os

import pytest
import torch

import golden_io
from helpers import assert_close, assert_equal, make_task
from omniisaacgymenvs_amd.sim import workloads
from oracle import anymal_terrain_oracle as ao
from test_oracle_golden_family_cpu import HAND_KINDS, anymal_flat_case, anymal_terrain_case, hand_case

pytestmark = pytest.mark.gpu

HAND_CFG = {
    "shadow_openai_ff": ("ShadowHandOpenAI_FF", {}),
    "shadow_full": ("ShadowHand", {}),
    "allegro_full": ("AllegroHand", {}),
    "allegro_mcs_ema": ("AllegroHand", dict(maxConsecutiveSuccesses=50, fallPenalty=-50.0, actionsMovingAverage=0.3,
                                            resetDofVelRandomInterval=0.1)),
}


@pytest.mark.parametrize("kind", HAND_KINDS)
def test_hand_hip_reproduces_golden(kind):
    for path in golden_io.files(f"hand_{kind}"):
        tag = os.path.basename(path)
        g, spec, s, b, init = hand_case(kind, path)
        n = s.object_pos.shape[0]
        cfg, over = HAND_CFG[kind]
        env, task = make_task(cfg, n, **over)
        dev = task.device
        assert task.max_episode_length == spec.max_episode_length and task.obs_type == spec.obs_type
        assert torch.equal(task.hand_dof_lower_limits.cpu(), spec.lower) and torch.equal(task.hand_dof_upper_limits.cpu(), spec.upper)
        task._env_pos.copy_(s.env_pos.to(dev))
        task.goal_pos.copy_(b.goal_pos.to(dev))
        nf = spec.num_fingertips
        st = dict(object_pos=s.object_pos, object_rot=s.object_rot, object_vel=s.object_velocities, goal_rot=b.goal_rot,
                  dof_pos=s.hand_dof_pos, dof_vel=s.hand_dof_vel, actions=b.actions, successes=b.successes, progress=b.progress_buf,
                  reset=b.reset_buf, reset_goal=b.reset_goal_buf)
        if nf:
            st.update(ft_pos=s.fingertip_pos, ft_rot=s.fingertip_rot, ft_vel=s.fingertip_velocities, sensors=s.vec_sensor)
        workloads.load_hand_state(task, st)
        task.consecutive_successes.copy_(b.consecutive_successes.to(dev))
        ok = torch.ones(n, dtype=torch.bool)  # every row is compared, the golden files' on-threshold rows included
        obs, rew, reset, extras = task.post_physics_step()
        torch.cuda.synchronize()
        assert_equal(reset[ok], g["post_reset_buf"][ok], f"{tag} reset_buf")
        assert_equal(task.reset_goal_buf[ok], g["post_reset_goal_buf"][ok], f"{tag} reset_goal_buf")
        assert_equal(task.progress_buf[ok], g["post_progress_buf"][ok], f"{tag} progress_buf")
        assert_close(obs, g["post_obs_buf"], f"{tag} obs_buf")
        if spec.asymmetric_obs:
            assert_close(task.states_buf, g["post_states_buf"], f"{tag} states_buf")
        assert_close(rew[ok], g["post_rew_buf"][ok], f"{tag} rew_buf")
        assert_close(task.successes[ok], g["post_successes"][ok], f"{tag} successes")
        if bool(ok.all()):
            assert_close(task.consecutive_successes, g["post_consecutive_successes"], f"{tag} consecutive_successes")


def test_anymal_flat_hip_reproduces_golden():
    for path in golden_io.files("anymalflat"):
        tag = os.path.basename(path)
        g, spec, st, b = anymal_flat_case(path)
        n = st["pos"].shape[0]
        env, task = make_task("Anymal", n)
        assert torch.equal(task.default_dof_pos.cpu(), b.default_dof_pos)
        workloads.load_anymal_flat_state(task, dict(st, commands=b.commands, actions=b.actions, last_actions=b.last_actions,
                                                    last_dof_vel=b.last_dof_vel, current_targets=b.current_targets,
                                                    progress=b.progress_buf, reset=b.reset_buf))
        obs, rew, reset, _ = task.post_physics_step()
        torch.cuda.synchronize()
        near = (st["pos"][:, 2] - spec.base_threshold).abs() < 1e-6 if hasattr(spec, "base_threshold") else torch.zeros(n, dtype=torch.bool)
        ok = ~near
        assert_equal(reset[ok], g["post_reset_buf"][ok], f"{tag} reset_buf")
        assert_equal(task.progress_buf, g["post_progress_buf"], f"{tag} progress_buf")
        assert_close(obs, g["post_obs_buf"], f"{tag} obs_buf")
        assert_close(rew[ok], g["post_rew_buf"][ok], f"{tag} rew_buf")
        assert_equal(task.last_actions, g["post_last_actions"], f"{tag} last_actions")
        assert_equal(task.last_dof_vel, g["post_last_dof_vel"], f"{tag} last_dof_vel")


def test_anymal_terrain_hip_reproduces_golden():
    """The in-step reset of post_physics_step draws from the in-kernel Philox here and from torch in the golden run, so the
    rows that depend on those draws (reset envs: obs DOF / height columns, commands) are compared on the other envs only;
    every flag, counter, reward and episode sum is compared on all of them."""
    for path in golden_io.files("anymalterrain"):
        tag = os.path.basename(path)
        g, spec, s, b, hs, origins = anymal_terrain_case(path)
        n = s.base_pos.shape[0]
        env, task = make_task("AnymalTerrain", n)
        dev = task.device
        assert tuple(task.height_samples.shape) == tuple(hs.shape) and tuple(task.terrain_origins.shape) == tuple(origins.shape)
        task.height_samples.copy_(hs.to(dev))
        task.terrain_origins.copy_(origins.to(dev))
        st = dict(base_pos=s.base_pos, base_quat=s.base_quat, base_vel=s.base_velocities, dof_pos=s.dof_pos, dof_vel=s.dof_vel,
                  knee_contact=s.knee_contact, base_contact=s.base_contact, commands=b.commands, actions=b.actions,
                  last_actions=b.last_actions, last_dof_vel=b.last_dof_vel, torques=b.torques, feet_air_time=b.feet_air_time,
                  episode_sums=torch.stack([b.episode_sums[k] for k in ao.EPISODE_KEYS]), terrain_levels=b.terrain_levels,
                  terrain_types=b.terrain_types, env_origins=b.env_origins, progress=b.progress_buf)
        workloads.load_anymal_state(task, st)
        task.post_physics_step(noise_u=g["noise_u"].to(dev).contiguous())
        torch.cuda.synchronize()
        reset = g["post_reset_buf"]
        keep = ~reset
        assert int(reset.sum()) > 0 and int(keep.sum()) > 0
        assert_equal(task.reset_buf, reset, f"{tag} reset_buf")
        assert_equal(task.timeout_buf, g["post_timeout_buf"], f"{tag} timeout_buf")
        assert_equal(task.progress_buf, g["post_progress_buf"], f"{tag} progress_buf")
        assert_equal(task.terrain_levels, g["post_terrain_levels"], f"{tag} terrain_levels")
        assert_equal(task.env_origins, g["post_env_origins"], f"{tag} env_origins")
        assert_close(task.rew_buf, g["post_rew_buf"], f"{tag} rew_buf")
        assert_close(task.obs_buf[keep], g["post_obs_buf"][keep], f"{tag} obs_buf (envs that did not reset)")
        assert_close(task.obs_buf[:, 0:9], g["post_obs_buf"][:, 0:9], f"{tag} obs_buf velocity / gravity columns")
        assert_close(task.commands[keep], g["post_commands"][keep], f"{tag} commands")
        assert_equal(task.last_actions, g["post_last_actions"], f"{tag} last_actions")
        for i, k in enumerate(ao.EPISODE_KEYS):
            assert_close(task._episode_sums[i], g[f"post_episode_sums__{k}"], f"{tag} episode_sums {k}")
            assert_close(task.extras["episode"]["rew_" + k].reshape(1), g[f"post_extras__{k}"].reshape(1), f"{tag} extras {k}")
        assert_close(task.extras["episode"]["terrain_level"].reshape(1), g["post_extras__terrain_level"].reshape(1), f"{tag} terrain_level")
