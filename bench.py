#!/usr/bin/env python3
"""bench.py -- fused obs+reward+reset env-steps/s (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--envs E] [--task Ant|Humanoid|ShadowHand|AnymalTerrain|Anymal|AllegroHand|Quadcopter|BallBalance|Ingenuity|Crazyflie|FrankaCabinet|TomoDJCabinet|TomoSPCCabinet|TomoDJnewhandCabinet|TomoDJnewhandTendonCabinet|TomoSPCnewhandCabinet]

One "step" = one pre_physics_step (reset-id compaction, reset of the flagged envs, action scaling /
PD torques) + one post_physics_step (progress, obs, reward, done[, in-step reset]) over a batch of
synthetic state resident in HBM, through the RLTask API -> ctypes -> liboige_hip.so.  With
``--task ShadowHand`` the wrapper-side observation and action noise (DR) runs too, as BASELINE.json
configs[3] asks.  Physics is delegated (north_star) and not part of the step.
N=1 default workload: BASELINE.json configs[1], "Ant obs(60)+reward+reset fused kernel, 1M envs
synthetic root/DOF state on one MI355X".  For N>1 every rank owns its own --envs envs (weak scaling;
envs are independent, so there is no data-path collective); ranks all-reduce a 32-byte
episode-statistics vector over RCCL once per timed region.

Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import contextlib
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 achievable)


# ---------------------------------------------------------------------------- algorithmic bytes
def locomotion_bytes(d: int, s: int, with_vecs: bool) -> dict:
    obs = 12 + 3 * d + 6 * s
    post_r = 12 + 16 + 24 + 4 * d + 4 * d + 24 * s + 4 * d + 4 + 8 + 8
    post_w = 4 * obs + 4 + 4 + 4 + 8 + 8 + (24 if with_vecs else 0)
    return {"post": post_r + post_w, "obs": obs}


HAND_POST_BYTES = 744 + 28 + 948        # openai obs 42 + asymmetric states 187 + reward/flags
ANYMAL_POST_BYTES = 512 + 560 + 913     # state + 280 int16 gathers + obs/flags/sums


# ---------------------------------------------------------------------------- per-task adapters
def _pick(n, frac, dev, seed):
    """Bernoulli(frac) mask over the envs (the envs that terminate on every step of a --reset-frac workload)."""
    g = torch.Generator(device=dev).manual_seed(seed + 77)
    return torch.rand(n, device=dev, generator=g) < frac


def _steady_hand_state(st, t, n, frac, dev, seed):
    """--reset-frac for the in-hand tasks: `frac` of the objects beyond fall_dist, as many at the goal rotation; no stale flags."""
    far, hit = _pick(n, frac, dev, seed), _pick(n, frac, dev, seed + 1)
    off = st["object_pos"] - (t.goal_pos + t._env_pos)
    off = off * (0.15 / off.norm(dim=-1, keepdim=True).clamp(min=0.15))          # everyone within 0.15 m ...
    off = torch.where(far.unsqueeze(-1), off * (0.3 / 0.15), off)                  # ... but these, at 0.3 m (fall_dist 0.24)
    st["object_pos"] = (t.goal_pos + t._env_pos + off).contiguous()
    q = torch.randn(n, 4, device=dev)
    q = q / q.norm(dim=-1, keepdim=True)
    st["object_rot"] = torch.where(hit.unsqueeze(-1), st["goal_rot"], q).contiguous()
    st["reset"].zero_()
    st["reset_goal"].zero_()
    st["successes"].zero_()


def _steady_cabinet_state(st, n, frac, dev, seed, open_at):
    """--reset-frac for the cabinet tasks: the drawer joint below its done threshold `open_at` except for `frac` of the envs;
    no stale flags.  The 500-step timeouts stay (progress is uniform over the episode: one env in 500 times out per step)."""
    bad = _pick(n, frac, dev, seed)
    d = st["cabinet_dof_pos"]
    d[:, 3] = torch.where(bad, torch.full_like(d[:, 3], open_at + 0.03), d[:, 3].clamp(max=open_at - 0.02))
    st["reset"].zero_()
    if "terminate" in st:
        st["terminate"].zero_()


class LocomotionBench:
    def __init__(self, name, n, dev, seed, reset_frac=None):
        from omniisaacgymenvs_amd.utils.task_util import make_task
        from omniisaacgymenvs_amd.sim import workloads

        self.name = name
        self.env, self.task = make_task(name, n, device=dev, seed=seed)
        t = self.task
        self.d, self.s = t._robots.num_dof, t._num_sensors
        z = (0.2, 0.9) if name == "Ant" else (0.5, 1.5)
        st = workloads.locomotion_state(n, t.dof_limits_lower, t.dof_limits_upper, self.s, device=dev, seed=seed, z_range=z)
        if reset_frac is not None:  # only `reset_frac` of the torsos below the termination height (+ the 1/1000 timeouts)
            low = _pick(n, reset_frac, dev, seed)
            zz = st["pos"][:, 2]
            st["pos"][:, 2] = torch.where(low, torch.full_like(zz, t.termination_height - 0.05),
                                          zz.clamp(min=t.termination_height + 0.02))
        workloads.load_locomotion_state(t, st)
        t._robots.apply_resets = False  # PhysX-side scatter is delegated; the kernels still compute every reset row
        self.actions = st["actions"]
        ab = locomotion_bytes(self.d, self.s, t.materialize_vecs)
        self.post_bytes = ab["post"]
        # pre_physics_step: R actions + W actions clone + W forces + R reset flag per env; per reset env R initial dof / root
        # (4D + 28), W dof_pos / dof_vel rows, root pose + velocity rows, id, potentials x2, reset, progress (8D + 92)
        self.pre_bytes = lambda resets, d=self.d: (12 * d + 8) * n + (12 * d + 120) * resets
        self.kernel = "locomotion_post_kernel"
        self.workload = f"{name} obs({ab['obs']})+reward+reset fused, {n} envs per GPU, synthetic root/DOF state"
        self.extra_cfg = {"num_dof": self.d, "num_sensors": self.s}

    def pre(self):
        self.task.pre_physics_step(self.actions)

    def post(self):
        self.task.post_physics_step()

    def resets(self):
        return int(self.task._reset_out["count"].item())


    @staticmethod
    def cpu_baseline(name, n, budget_s):
        from omniisaacgymenvs_amd.sim import workloads
        from oracle import locomotion_oracle as lo

        spec = lo.ant_spec() if name == "Ant" else lo.humanoid_spec()
        z = (0.2, 0.9) if name == "Ant" else (0.5, 1.5)
        st = workloads.locomotion_state(n, spec.dof_limits_lower, spec.dof_limits_upper, spec.num_sensors, "cpu", 1234, z)
        b = lo.LocomotionBuffers.fresh(spec, n)
        b.reset_buf[:] = 0
        b.progress_buf[:] = st["progress"]
        b.potentials[:] = st["potentials"]
        init_pos, init_rot = st["pos"].clone(), st["rot"].clone()
        init_dof = torch.zeros(n, spec.num_dof)
        g = torch.Generator().manual_seed(0)
        draw = lambda ids: (torch.rand(len(ids), spec.num_dof, generator=g), torch.rand(len(ids), spec.num_dof, generator=g))

        def step():
            lo.pre_physics(spec, b, st["actions"], init_pos, init_rot, init_dof, draw=draw)
            lo.post_physics(spec, b, st["pos"], st["rot"], st["vel"], st["dof_pos"], st["dof_vel"], st["sensors"])

        return step


class HandBench:
    """C4: ShadowHandOpenAI_FF (openai obs + asymmetric full_state states + goal-rotation reward) with DR noise."""

    def __init__(self, name, n, dev, seed, reset_frac=None):
        from omniisaacgymenvs_amd.utils.task_util import make_task
        from omniisaacgymenvs_amd.sim import workloads
        from omniisaacgymenvs_amd.utils.config import TASK_DEFAULTS

        dr = copy.deepcopy(TASK_DEFAULTS["ShadowHandOpenAI_FF"]["domain_randomization"])
        dr["randomize"] = True
        self.env, self.task = make_task("ShadowHandOpenAI_FF", n, device=dev, seed=seed, domain_randomization=dr)
        t = self.task
        st = workloads.hand_state(n, t._env_pos, t.goal_pos, t.hand_dof_lower_limits, t.hand_dof_upper_limits, device=dev, seed=seed)
        if reset_frac is not None:
            _steady_hand_state(st, t, n, reset_frac, dev, seed)
        workloads.load_hand_state(t, st)
        for v in (t._hands, t._objects, t._goals):
            v.apply_resets = False
        self.actions = st["actions"]
        self._noisy = self.actions.clone()  # the policy's actions; noised in place from then on (see pre())
        self._obs_clamped = torch.empty_like(t.obs_buf)
        # OIGE_BENCH_SEPARATE_DR=1: the noise as two separate launches (what VecEnvRLGames does with fused_dr = False)
        self.fused_dr = os.environ.get("OIGE_BENCH_SEPARATE_DR", "0") != "1"
        # in-kernel DR: the post kernel also reads the correlated noise (168), updates the counter (8) and writes the
        # clamped copy the policy receives (168) next to the noised obs_buf
        self.post_bytes = HAND_POST_BYTES + (168 + 8 + 168 if self.fused_dr else 0)
        # pre_physics_step with the action noise inside: R actions 80, W noised actions 80 (the caller's tensor, in place), R
        # correlated noise 80, counter RW 8, W actions clone 80, R prev_targets 96, W prev / cur targets 192, W targets_out
        # 96, R reset + reset_goal flags 16; per reset env ~ 53 draws -> object pose / velocity rows 52, goal rows 28 x2,
        # dof_pos / dof_vel rows 192, id 8, flags 24, W correlated noise 80; per goal reset: goal rows 28 + 28, id 8, flag 8
        self.pre_bytes = lambda resets: 728 * n + 412 * resets
        self.kernel = "hand_post_kernel"
        self.workload = (f"ShadowHand openai obs(42) + full_state states(187) + goal-rotation reward + reset_target_pose/reset_idx "
                         f"+ obs/action DR noise ({'applied inside the pre/post kernels' if self.fused_dr else 'separate launches'}), "
                         f"{n} envs per GPU, synthetic state")
        self.extra_cfg = {"observationType": t.obs_type, "asymmetric_observations": True, "dr_noise": True,
                          "dr_in_kernel": self.fused_dr}

    def pre(self):
        t = self.task
        # The action noise is applied IN PLACE to the tensor the policy handed over (randomize.py:314-342), so `_noisy` carries the
        # previous steps' noise into this one (a bounded random walk; the kernels are branch-free in the values).  Refreshing it
        # from `actions` every step would put an 84 MB device copy -- the stand-in for the policy writing its output, not part of
        # pre_physics_step -- inside the timed step.
        if self.fused_dr:
            t.clamp_actions_in_kernel = True
            t.pre_physics_step(self._noisy, action_noise=t._dr_randomizer.fused_actions_args(self._noisy))
        else:
            a = t._dr_randomizer.apply_actions_randomization(actions=self._noisy, reset_buf=t.reset_buf)
            t.pre_physics_step(a)

    def post(self):
        t = self.task
        if self.fused_dr:
            t.post_physics_step(obs_noise=t._dr_randomizer.fused_observations_args(t.obs_buf, clamped_out=self._obs_clamped,
                                                                                   clip=t.clip_obs))
        else:
            t.post_physics_step()

    def after_post(self):  # wrapper-side observation noise (vec_env_rlgames.py:94-97); outside the post-kernel event bracket
        t = self.task
        if not self.fused_dr:
            t._dr_randomizer.apply_observations_randomization(observations=t.obs_buf, reset_buf=t.reset_buf,
                                                              clamped_out=self._obs_clamped, clip=t.clip_obs)

    def resets(self):
        return int(self.task._reset_out["count"].item())


    @staticmethod
    def cpu_baseline(name, n, budget_s):
        from omniisaacgymenvs_amd.sim import workloads
        from omniisaacgymenvs_amd.tasks.shadow_hand import SHADOW_HAND_ACTUATED, shadow_hand_stand_in_limits
        from omniisaacgymenvs_amd.utils.config import TASK_DEFAULTS
        from oracle import dr_noise_oracle as dro
        from oracle import hand_oracle as ho

        lim = shadow_hand_stand_in_limits()
        e = TASK_DEFAULTS["ShadowHandOpenAI_FF"]["env"]
        spec = ho.HandSpec(obs_type="openai", asymmetric_obs=True, fall_penalty=e["fallPenalty"], success_tolerance=e["successTolerance"],
                           act_moving_average=e["actionsMovingAverage"], max_episode_length=160,
                           max_consecutive_successes=e["maxConsecutiveSuccesses"], av_factor=e["averFactor"],
                           actuated_dof_indices=list(SHADOW_HAND_ACTUATED), lower=lim[:, 0].contiguous(), upper=lim[:, 1].contiguous())
        env_pos = torch.zeros(n, 3)
        goal_pos = torch.tensor([0.0, -0.39, 0.56]).repeat(n, 1)
        st = workloads.hand_state(n, env_pos, goal_pos, spec.lower, spec.upper, device="cpu", seed=1234)
        s = ho.HandState(st["object_pos"], st["object_rot"], st["object_vel"], st["ft_pos"], st["ft_rot"], st["ft_vel"],
                         st["dof_pos"], st["dof_vel"], st["sensors"], env_pos)
        b = ho.HandBuffers(torch.zeros(n, 42), torch.zeros(n, 187), torch.zeros(n), st["reset"].clone(), st["reset_goal"].clone(),
                           st["progress"].clone(), st["successes"].clone(), torch.zeros(1), goal_pos.clone(), st["goal_rot"].clone(),
                           st["actions"].clone(), torch.zeros(n, 24), torch.zeros(n, 24))
        init = ho.HandInit(goal_pos.clone(), goal_pos.clone(), env_pos, torch.zeros(24), torch.zeros(24))
        dr = TASK_DEFAULTS["ShadowHand"]["domain_randomization"]["randomization_params"]
        oc, occ = torch.zeros(n, dtype=torch.int), torch.zeros(n, 42)
        ac, acc = torch.zeros(n, dtype=torch.int), torch.zeros(n, 20)
        u = lambda ids, k: 2.0 * torch.rand(len(ids), k) - 1.0

        def step():
            a = dro.apply_randomization(st["actions"].clone(), b.reset_buf, ac, acc, dr["actions"])
            ho.pre_physics(spec, b, init, a, draw_goal=lambda ids: u(ids, 4), draw_reset=lambda ids: (u(ids, 53), u(ids, 4)))
            ho.post_physics(spec, s, b)
            dro.apply_randomization(b.obs_buf, b.reset_buf, oc, occ, dr["observations"])

        return step


class AnymalBench:
    """C5 per-GPU shard: AnymalTerrain height-scan obs + 11-term reward + in-step reset."""

    def __init__(self, name, n, dev, seed, reset_frac=None):
        from omniisaacgymenvs_amd.utils.task_util import make_task
        from omniisaacgymenvs_amd.sim import workloads

        self.env, self.task = make_task("AnymalTerrain", n, device=dev, seed=seed)
        t = self.task
        st = workloads.anymal_state(n, t.terrain_origins, t.default_dof_pos, device=dev, seed=seed)
        if reset_frac is not None:  # contacts below the 1 N thresholds except for `reset_frac` of the bases
            fall = _pick(n, reset_frac, dev, seed)
            st["knee_contact"] = st["knee_contact"].clamp(max=0.5)
            st["base_contact"] = torch.where(fall.unsqueeze(-1), torch.full_like(st["base_contact"], 2.0), st["base_contact"].clamp(max=0.5))
        workloads.load_anymal_state(t, st)
        t._anymals.apply_resets = False
        t.decimation = 1  # one PD-torque sub-step per env-step (the other three belong to the delegated physics loop)
        self.actions = st["actions"]
        self.post_bytes = ANYMAL_POST_BYTES
        self.pre_bytes = lambda resets: 240 * n  # PD torques: R actions, dof_pos, dof_vel + W torques + W actions clone (the resets are in-step, post side)
        self.kernel = "anymal_head_kernel + anymal_scan_kernel (+ flags / extras: the whole post_physics_step)"
        self.workload = (f"AnymalTerrain height-scan obs(188) + 11-term reward + in-step reset + obs noise, {n} envs per GPU, "
                         "synthetic state, 1200x2000 int16 heightfield")
        self.extra_cfg = {"scan_points": 140, "decimation_substeps_timed": 1}

    def pre(self):
        self.task.pre_physics_step(self.actions)

    def post(self):
        self.task.post_physics_step()

    def resets(self):
        return int(self.task._reset_out["count"].item())


    @staticmethod
    def cpu_baseline(name, n, budget_s):
        from omniisaacgymenvs_amd.sim import workloads
        from omniisaacgymenvs_amd.sim.terrain import SyntheticTerrain
        from omniisaacgymenvs_amd.utils.config import TASK_DEFAULTS
        from oracle import anymal_terrain_oracle as ao

        spec = ao.AnymalSpec()
        terrain = SyntheticTerrain(TASK_DEFAULTS["AnymalTerrain"]["env"]["terrain"], n)
        hs = torch.tensor(terrain.heightsamples)
        origins = torch.from_numpy(terrain.env_origins).float()
        default = torch.tensor(ao.DEFAULT_JOINT_ANGLES).repeat(n, 1)
        st = workloads.anymal_state(n, origins, default, device="cpu", seed=1234)
        s = ao.AnymalState(st["base_pos"], st["base_quat"], st["base_vel"], st["dof_pos"], st["dof_vel"], st["knee_contact"],
                           st["base_contact"])
        b = ao.AnymalBuffers(
            obs_buf=torch.zeros(n, 188), rew_buf=torch.zeros(n), reset_buf=torch.ones(n, dtype=torch.long),
            timeout_buf=torch.zeros(n, dtype=torch.long), progress_buf=st["progress"].clone(), commands=st["commands"].clone(),
            actions=st["actions"].clone(), last_actions=st["last_actions"].clone(), last_dof_vel=st["last_dof_vel"].clone(),
            torques=st["torques"].clone(), feet_air_time=st["feet_air_time"].clone(),
            episode_sums={k: st["episode_sums"][i].clone() for i, k in enumerate(ao.EPISODE_KEYS)},
            terrain_levels=st["terrain_levels"].clone(), terrain_types=st["terrain_types"].clone(),
            env_origins=st["env_origins"].clone(), default_dof_pos=default)

        def draw(ids):
            k = len(ids)
            return ao.AnymalResetDraws(torch.rand(k, 12), torch.rand(k, 12), torch.rand(k, 2), torch.rand(k, 1), torch.rand(k, 1),
                                       torch.rand(k, 1))

        def step():
            b.torques = ao.pd_torques(spec, b.actions, default, s.dof_pos, s.dof_vel)
            ao.post_physics(spec, b, s, hs, origins, draw=draw)

        return step


class AllegroHandBench:
    """AllegroHand, `full` obs(72) + goal-rotation reward + resets (the YAML has no DR section)."""

    def __init__(self, name, n, dev, seed, reset_frac=None):
        from omniisaacgymenvs_amd.utils.task_util import make_task
        from omniisaacgymenvs_amd.sim import workloads

        self.env, self.task = make_task("AllegroHand", n, device=dev, seed=seed)
        t = self.task
        st = workloads.hand_state(n, t._env_pos, t.goal_pos, t.hand_dof_lower_limits, t.hand_dof_upper_limits, device=dev, seed=seed,
                                  num_actions=16, fingertips=0)
        if reset_frac is not None:
            _steady_hand_state(st, t, n, reset_frac, dev, seed)
        workloads.load_hand_state(t, st)
        for v in (t._hands, t._objects, t._goals):
            v.apply_resets = False
        self.actions = st["actions"]
        self.post_bytes = 312 + 320  # csrc/hand.hip header
        self.kernel = "allegro_post_kernel"
        self.workload = f"AllegroHand full obs(72) + goal-rotation reward + reset_target_pose/reset_idx, {n} envs per GPU, synthetic state"
        self.extra_cfg = {"observationType": t.obs_type, "num_dof": 16}

    def pre(self):
        self.task.pre_physics_step(self.actions)

    def post(self):
        self.task.post_physics_step()

    def resets(self):
        return int(self.task._reset_out["count"].item())

    @staticmethod
    def cpu_baseline(name, n, budget_s):
        from omniisaacgymenvs_amd.sim import workloads
        from oracle import hand_oracle as ho

        spec = ho.allegro_spec("full")
        env_pos = torch.zeros(n, 3)
        goal_pos = torch.tensor([0.0, -0.2, 0.52]).repeat(n, 1)
        st = workloads.hand_state(n, env_pos, goal_pos, spec.lower, spec.upper, device="cpu", seed=1234, num_actions=16, fingertips=0)
        s = ho.HandState(st["object_pos"], st["object_rot"], st["object_vel"], None, None, None, st["dof_pos"], st["dof_vel"], None, env_pos)
        b = ho.HandBuffers(torch.zeros(n, 72), torch.zeros(n, 0), torch.zeros(n), st["reset"].clone(), st["reset_goal"].clone(),
                           st["progress"].clone(), st["successes"].clone(), torch.zeros(1), goal_pos.clone(), st["goal_rot"].clone(),
                           st["actions"].clone(), torch.zeros(n, 16), torch.zeros(n, 16))
        init = ho.HandInit(goal_pos.clone(), goal_pos.clone(), env_pos, torch.zeros(16), torch.zeros(16))
        u = lambda ids, k: 2.0 * torch.rand(len(ids), k) - 1.0

        def step():
            ho.pre_physics(spec, b, init, st["actions"], draw_goal=lambda ids: u(ids, 4), draw_reset=lambda ids: (u(ids, 37), u(ids, 4)))
            ho.post_physics(spec, s, b)

        return step


class QuadcopterBench:
    """Quadcopter obs(21) + inverse-quadratic reward + non-sticky done, targets / thrust integration."""

    def __init__(self, name, n, dev, seed, reset_frac=None):
        from omniisaacgymenvs_amd.utils.task_util import make_task
        from omniisaacgymenvs_amd.sim import workloads

        self.env, self.task = make_task("Quadcopter", n, device=dev, seed=seed)
        t = self.task
        st = workloads.quadcopter_state(n, t._env_pos, device=dev, seed=seed)
        if reset_frac is not None:  # everyone hovers near the target (0, 0, 1) except `reset_frac` of the copters below 0.3 m; no stale flags
            bad = _pick(n, reset_frac, dev, seed)
            off = torch.cat([torch.randn(n, 2, device=dev) * 0.3, torch.rand(n, 1, device=dev) * 0.4 + 0.8], -1)
            off[bad, 2] = 0.1
            st["root_pos"] = (t._env_pos + off).contiguous()
            st["reset"].zero_()
        workloads.load_quadcopter_state(t, st)
        t._copters.apply_resets = False
        self.actions = st["actions"]
        self.post_bytes = 116 + 104  # csrc/quadcopter.hip header
        self.kernel = "quadcopter_post_kernel"
        self.workload = f"Quadcopter obs(21)+reward+reset fused, {n} envs per GPU, synthetic root/DOF state"
        self.extra_cfg = {"num_dof": 8}

    def pre(self):
        self.task.pre_physics_step(self.actions)

    def post(self):
        self.task.post_physics_step()

    def resets(self):
        return int(self.task._reset_out["count"].item())

    @staticmethod
    def cpu_baseline(name, n, budget_s):
        from omniisaacgymenvs_amd.sim import workloads
        from oracle import quadcopter_oracle as qo

        spec = qo.QuadcopterSpec()
        env_pos = torch.zeros(n, 3)
        st = workloads.quadcopter_state(n, env_pos, device="cpu", seed=1234)
        tp = torch.zeros(n, 3)
        tp[:, 2] = 1.0
        b = qo.QuadcopterBuffers(torch.zeros(n, 21), torch.zeros(n), st["reset"].clone(), st["progress"].clone(), tp,
                                 st["targets"].clone(), st["thrusts"].clone(), torch.zeros(n, 4, 3))
        init_pos, init_rot = st["root_pos"].clone(), st["root_rot"].clone()
        draw = lambda ids: (torch.rand(len(ids), 8), torch.rand(len(ids), 1), torch.rand(len(ids), 1), torch.rand(len(ids), 1))

        def step():
            qo.pre_physics(spec, b, st["actions"], init_pos, init_rot, draw)
            qo.post_physics(spec, b, st["root_pos"], st["root_rot"], st["root_vel"], st["dof_pos"], env_pos)

        return step


class CartpoleBench:
    """C1 (BASELINE.json configs[0], the plumbing config): Cartpole obs(4) + reward + done + reset, 4096 envs, cartpole.py:91-164.
    At this size the step is two launches of a few microseconds each: the line measures launch + marshalling cost, not HBM."""

    def __init__(self, name, n, dev, seed, reset_frac=None):
        from omniisaacgymenvs_amd.utils.task_util import make_task
        from omniisaacgymenvs_amd.sim import workloads

        self.env, self.task = make_task("Cartpole", n, device=dev, seed=seed)
        t = self.task
        st = workloads.cartpole_state(n, device=dev, seed=seed)
        if reset_frac is not None:  # everyone upright and centred except `reset_frac` of the carts beyond resetDist
            bad = _pick(n, reset_frac, dev, seed)
            st["dof_pos"] = (torch.rand(n, 2, device=dev) * 0.4 - 0.2).contiguous()
            st["dof_pos"][bad, 0] = 3.2
            st["progress"] = torch.randint(0, 400, (n,), device=dev)
        t._cartpoles.joint_positions.copy_(st["dof_pos"])
        t._cartpoles.joint_velocities.copy_(st["dof_vel"])
        t.progress_buf.copy_(st["progress"])
        t.reset_buf.zero_()
        t._cartpoles.apply_resets = False
        self.actions = st["actions"]
        self.post_bytes = 60  # R dof_pos 8 + dof_vel 8 + progress 8; W obs 16 + rew 4 + reset 8 + progress 8
        # pre: R actions 4 + reset flag 8, W forces 8; per reset env W dof_pos / dof_vel rows 16, id 8, reset + progress 16
        self.pre_bytes = lambda resets: 20 * n + 40 * resets
        self.kernel = "cartpole_post_kernel"
        self.workload = f"Cartpole obs(4)+reward+reset fused, {n} envs per GPU, synthetic DOF state"
        self.extra_cfg = {"num_dof": 2}

    def pre(self):
        self.task.pre_physics_step(self.actions)

    def post(self):
        self.task.post_physics_step()

    def resets(self):
        return int(self.task._reset_out["count"].item())

    @staticmethod
    def cpu_baseline(name, n, budget_s):
        from omniisaacgymenvs_amd.sim import workloads
        from oracle import cartpole_oracle as co

        spec = co.CartpoleSpec()
        st = workloads.cartpole_state(n, device="cpu", seed=1234)
        obs, rew = torch.zeros(n, 4), torch.zeros(n)
        reset, progress = torch.zeros(n, dtype=torch.long), st["progress"].clone()
        draw = lambda ids: torch.rand(4, len(ids))

        def step():
            co.pre_physics(spec, st["actions"], reset, progress, draw=draw)
            co.post_physics(spec, st["dof_pos"], st["dof_vel"], obs, rew, reset, progress)

        return step


class BallBalanceBench:
    """BallBalance obs(24) + reward + sticky done, position-target integration."""

    def __init__(self, name, n, dev, seed, reset_frac=None):
        from omniisaacgymenvs_amd.utils.task_util import make_task
        from omniisaacgymenvs_amd.sim import workloads

        self.env, self.task = make_task("BallBalance", n, device=dev, seed=seed)
        t = self.task
        st = workloads.ball_balance_state(n, t._env_pos, device=dev, seed=seed)
        if reset_frac is not None:  # balls 0.5-0.9 m up except `reset_frac` of them below the 0.15 m reset height; no stale flags
            bad = _pick(n, reset_frac, dev, seed)
            z = torch.rand(n, device=dev) * 0.4 + 0.5
            z[bad] = 0.05
            st["ball_pos"][:, 2] = t._env_pos[:, 2] + z
            st["reset"].zero_()
        workloads.load_ball_balance_state(t, st)
        for v in (t._balls, t._balance_bots):
            v.apply_resets = False
        self.actions = st["actions"]
        self.post_bytes = 184 + 116  # csrc/ball_balance.hip header
        self.kernel = "ball_balance_post_kernel"
        self.workload = f"BallBalance obs(24)+reward+reset fused, {n} envs per GPU, synthetic ball/DOF/sensor state"
        self.extra_cfg = {"num_dof": 6}

    def pre(self):
        self.task.pre_physics_step(self.actions)

    def post(self):
        self.task.post_physics_step()

    def resets(self):
        return int(self.task._reset_out["count"].item())

    @staticmethod
    def cpu_baseline(name, n, budget_s):
        from omniisaacgymenvs_amd.sim import workloads
        from oracle import ball_balance_oracle as bo

        spec = bo.BallBalanceSpec()
        env_pos = torch.zeros(n, 3)
        st = workloads.ball_balance_state(n, env_pos, device="cpu", seed=1234)
        b = bo.BallBalanceBuffers(torch.zeros(n, 24), torch.zeros(n), st["reset"].clone(), st["progress"].clone(), st["targets"].clone())
        q = torch.zeros(n, 4)
        q[:, 0] = 1
        init = bo.BallBalanceInit(st["ball_pos"].clone(), q, torch.zeros(n, 6), env_pos.clone(), q.clone(), torch.zeros(n, 6),
                                  torch.zeros(n, 6))
        draw = lambda ids: torch.rand(len(ids), 5)

        def step():
            bo.pre_physics(spec, b, init, st["actions"], draw)
            bo.post_physics(spec, b, st["ball_pos"], st["ball_vel"], st["dof_pos"], st["dof_vel"], st["sensors"], env_pos)

        return step


class IngenuityBench:
    """Ingenuity obs(13) + reward + done, reset + set_targets compactions, thrust vectors."""

    def __init__(self, name, n, dev, seed, reset_frac=None):
        from omniisaacgymenvs_amd.utils.task_util import make_task
        from omniisaacgymenvs_amd.sim import workloads

        self.env, self.task = make_task("Ingenuity", n, device=dev, seed=seed)
        t = self.task
        st = workloads.ingenuity_state(n, t._env_pos, device=dev, seed=seed)
        if reset_frac is not None:  # within ~0.2 m of the target (z >= 0.8) except `reset_frac` of the roots below 0.5 m; no stale flags
            bad = _pick(n, reset_frac, dev, seed)
            off = st["targets"] + torch.randn(n, 3, device=dev) * 0.07
            off[bad, 2] = 0.2
            st["root_pos"] = (t._env_pos + off).contiguous()
            st["reset"].zero_()
        workloads.load_ingenuity_state(t, st)
        for v in (t._copters, t._balls):
            v.apply_resets = False
        self.actions = st["actions"]
        self.post_bytes = 84 + 72  # R root pos 12 + rot 16 + vel 24 + target 12 + env_pos 12 + progress 8; W obs 52 + rew 4 + reset 8 + progress 8
        self.kernel = "ingenuity_post_kernel"
        self.workload = f"Ingenuity obs(13)+reward+reset fused, {n} envs per GPU, synthetic root state"
        self.extra_cfg = {"num_dof": 4}

    def pre(self):
        self.task.pre_physics_step(self.actions)

    def post(self):
        self.task.post_physics_step()

    def resets(self):
        return int(self.task._reset_out["count"].item())

    @staticmethod
    def cpu_baseline(name, n, budget_s):
        from omniisaacgymenvs_amd.sim import workloads
        from oracle import ingenuity_oracle as io

        spec = io.IngenuitySpec()
        env_pos = torch.zeros(n, 3)
        st = workloads.ingenuity_state(n, env_pos, device="cpu", seed=1234)
        b = io.IngenuityBuffers(torch.zeros(n, 13), torch.zeros(n), st["reset"].clone(), st["progress"].clone(), st["targets"].clone(),
                                torch.zeros(n, 2, 3), st["dof_pos"].clone(), st["dof_vel"].clone())
        init_pos, init_rot = st["root_pos"].clone(), st["root_rot"].clone()

        def step():
            io.pre_physics(spec, b, st["actions"], env_pos, init_pos, init_rot, init_rot,
                           lambda ids: torch.rand(len(ids), 5), lambda ids: (torch.rand(len(ids), 2), torch.rand(len(ids))))
            io.post_physics(spec, b, st["root_pos"], st["root_rot"], st["root_vel"], env_pos)

        return step


class CrazyflieBench:
    """Crazyflie obs(18) + 4-term reward + 8 episode sums + done; motor filter, two compactions, extras."""

    def __init__(self, name, n, dev, seed, reset_frac=None):
        from omniisaacgymenvs_amd.utils.task_util import make_task
        from omniisaacgymenvs_amd.sim import workloads

        self.env, self.task = make_task("Crazyflie", n, device=dev, seed=seed)
        t = self.task
        st = workloads.crazyflie_state(n, t._env_pos, device=dev, seed=seed)
        if reset_frac is not None:  # upright, within ~0.5 m of the target (0, 0, 2) except `reset_frac` of the roots below 0.5 m; no stale flags
            bad = _pick(n, reset_frac, dev, seed)
            q = torch.randn(n, 4, device=dev) * 0.1
            q[:, 0] += 1.0
            st["root_rot"] = (q / q.norm(dim=-1, keepdim=True)).contiguous()
            off = st["targets"] + torch.randn(n, 3, device=dev) * 0.2
            off[bad, 2] = 0.3
            st["root_pos"] = (t._env_pos + off).contiguous()
            st["reset"].zero_()
        workloads.load_crazyflie_state(t, st)
        for v in (t._copters, t._balls):
            v.apply_resets = False
        self.actions = st["actions"]
        self.post_bytes = 132 + 124  # csrc/crazyflie.hip header
        self.kernel = "crazyflie_post_kernel"
        self.workload = f"Crazyflie obs(18)+reward+episode sums+reset fused, {n} envs per GPU, synthetic root state"
        self.extra_cfg = {"num_dof": 4}

    def pre(self):
        self.task.pre_physics_step(self.actions)

    def post(self):
        self.task.post_physics_step()

    def resets(self):
        return int(self.task._reset_out["count"].item())

    @staticmethod
    def cpu_baseline(name, n, budget_s):
        from omniisaacgymenvs_amd.sim import workloads
        from oracle import crazyflie_oracle as co

        spec = co.CrazyflieSpec()
        env_pos = torch.zeros(n, 3)
        st = workloads.crazyflie_state(n, env_pos, device="cpu", seed=1234)
        b = co.CrazyflieBuffers(torch.zeros(n, 18), torch.zeros(n), st["reset"].clone(), st["progress"].clone(), st["targets"].clone(),
                                st["prev_actions"].clone(), torch.zeros(n, 4, 3), st["cmds_damp"].clone(), st["rot_damp"].clone(),
                                torch.zeros(n, 4), torch.zeros(n, 4), {k: st["episode_sums"][i].clone() for i, k in enumerate(co.EPISODE_KEYS)})
        init_pos, init_rot = st["root_pos"].clone(), st["root_rot"].clone()

        def step():
            co.pre_physics(spec, b, st["actions"], st["root_rot"], env_pos, init_pos, init_rot, init_rot, 0.01 * torch.randn(4))
            co.post_physics(spec, b, st["root_pos"], st["root_rot"], st["root_vel"], env_pos)

        return step


class FrankaCabinetBench:
    """FrankaCabinet grasp transforms + obs(23) + 8-term reward + sticky done; target integration, resets."""

    def __init__(self, name, n, dev, seed, reset_frac=None):
        from omniisaacgymenvs_amd.utils.task_util import make_task
        from omniisaacgymenvs_amd.sim import workloads

        self.env, self.task = make_task(name, n, device=dev, seed=seed)
        t = self.task
        st = workloads.franka_cabinet_state(n, t._env_pos, t.franka_local_grasp_pos[0], device=dev, seed=seed, num_dofs=t.NUM_DOFS,
                                            drawer_scale=0.45 if name == "FrankaCabinet" else 0.22)
        if reset_frac is not None:  # the drawer joint's done threshold: 0.39 (franka_cabinet.py:601), 0.19 in the Tomo clones
            _steady_cabinet_state(st, n, reset_frac, dev, seed, float(t.OPEN_DONE_THRESHOLD))
        workloads.load_franka_cabinet_state(t, st)
        for v in (t._frankas, t._cabinets, t._props):
            v.apply_resets = False
        self.actions = st["actions"]
        D = t.NUM_DOFS
        self.post_bytes = (112 + 12 * D) + (76 + 4 * (2 * D + 5))  # csrc/franka_cabinet.hip header: 220 + 168 at 9 DOFs
        self.kernel = "franka_cabinet_post_kernel"
        self.workload = f"{name} grasp transforms+obs({2 * D + 5})+reward+reset fused, {n} envs per GPU, synthetic hand/drawer/DOF state"
        self.extra_cfg = {"num_dof": D, "num_props": t.num_props}

    def pre(self):
        self.task.pre_physics_step(self.actions)

    def post(self):
        self.task.post_physics_step()

    def resets(self):
        return int(self.task._reset_out["count"].item())

    @staticmethod
    def cpu_baseline(name, n, budget_s):
        from omniisaacgymenvs_amd.sim import workloads
        from oracle import franka_cabinet_oracle as fo

        spec = {"FrankaCabinet": fo.FrankaCabinetSpec, "TomoDJCabinet": fo.tomodj_spec, "TomoSPCCabinet": fo.tomospc_spec}[name]()
        D = spec.num_dofs
        env_pos = torch.zeros(n, 3)
        st = workloads.franka_cabinet_state(n, env_pos, torch.tensor(spec.franka_local_grasp_pos), device="cpu", seed=1234, num_dofs=D,
                                            drawer_scale=0.45 if name == "FrankaCabinet" else 0.22)
        if name == "FrankaCabinet":
            st["rfinger_pos"] = st["lfinger_pos"]
        b = fo.FrankaCabinetBuffers(torch.zeros(n, 2 * D + 5), torch.zeros(n), st["reset"].clone(), st["progress"].clone(), st["prev_actions"].clone(),
                                    st["targets"].clone(), torch.zeros(n, 4), torch.zeros(n, 3), torch.zeros(n, 4), torch.zeros(n, 3))
        lo = torch.tensor([-2.8973, -1.7628, -2.8973, -3.0718, -2.8973, -0.0175, -2.8973, 0.0, 0.0])[9 - D:]
        hi = torch.tensor([2.8973, 1.7628, 2.8973, -0.0698, 2.8973, 3.7525, 2.8973, 0.04, 0.04])[9 - D:]
        speed = torch.ones(D)
        speed[D - 2:] = 0.1
        pp, pr = torch.zeros(n * 4, 3), torch.zeros(n * 4, 4)

        def step():
            fo.pre_physics(spec, b, st["actions"], lo, hi, speed, lambda ids: torch.rand(len(ids), D), num_props=4, default_prop_pos=pp,
                           default_prop_rot=pr)
            fo.post_physics(spec, b, st, lo, hi)

        return step


class NewhandCabinetBench:
    """the fork's five-finger cabinet tasks (10 keypoints, obs(2D+5), reward, sticky done [+ terminate])."""

    def __init__(self, name, n, dev, seed, reset_frac=None):
        from omniisaacgymenvs_amd.utils.task_util import make_task
        from omniisaacgymenvs_amd.sim import workloads

        self.env, self.task = make_task(name, n, device=dev, seed=seed)
        t = self.task
        st = workloads.newhand_cabinet_state(n, t._env_pos, t.robot_dof_lower_limits, t.robot_dof_upper_limits, t.num_actions, device=dev,
                                             seed=seed)
        if reset_frac is not None:  # drawer below 0.19 and (Tendon's terminate_buf) every joint inside its limits
            _steady_cabinet_state(st, n, reset_frac, dev, seed, 0.19)
            lo_, hi_ = t.robot_dof_lower_limits.to(dev), t.robot_dof_upper_limits.to(dev)
            st["dof_pos"] = torch.maximum(torch.minimum(st["dof_pos"], hi_), lo_).contiguous()
        workloads.load_newhand_cabinet_state(t, st)
        for v in (t._robots, t._cabinets, t._props):
            v.apply_resets = False
        self.actions = st["actions"]
        D, A, term = t.NUM_DOFS, t.num_actions, 8 if t.TERMINATE_ON_LIMITS else 0
        self.post_bytes = (16 + 5 * 28 + 28 + 8 * D + 32 + 4 * A + 16 + term) + (4 * (2 * D + 5) + 20 + 120 + term)  # csrc/newhand_cabinet.hip
        self.kernel = "newhand_cabinet_post_kernel"
        self.workload = f"{name} keypoints+obs({2 * D + 5})+reward+reset fused, {n} envs per GPU, synthetic finger/drawer/DOF state"
        self.extra_cfg = {"num_dof": D, "num_actions": A, "num_props": t.num_props}

    def pre(self):
        self.task.pre_physics_step(self.actions)

    def post(self):
        self.task.post_physics_step()

    def resets(self):
        return int(self.task._reset_out["count"].item())

    @staticmethod
    def cpu_baseline(name, n, budget_s):
        from omniisaacgymenvs_amd.sim import workloads
        from omniisaacgymenvs_amd.tasks import newhand_cabinet as nh
        from oracle import newhand_cabinet_oracle as no

        cls = getattr(nh, name + "Task")
        spec = {"TomoDJnewhandCabinet": no.tomodj_newhand_spec, "TomoSPCnewhandCabinet": no.tomospc_newhand_spec,
                "TomoDJnewhandTendonCabinet": lambda: no.tomodj_newhand_tendon_spec(cls.ACTUATED)}[name]()
        D, A = spec.num_dofs, spec.num_actions
        g = torch.Generator().manual_seed(D)
        lo, hi = -(torch.rand(D, generator=g) + 0.3), torch.rand(D, generator=g) + 0.3
        st = workloads.newhand_cabinet_state(n, torch.zeros(n, 3), lo, hi, A, device="cpu", seed=1234)
        b = no.NewhandBuffers(torch.zeros(n, 2 * D + 5), torch.zeros(n), st["reset"].clone(), st["progress"].clone(), st["prev_actions"].clone(),
                              st["targets"].clone(), torch.zeros(n, 5, 3), torch.zeros(n, 5, 3), st["terminate"].clone())
        speed = torch.ones(D)
        speed[D - 19:] = 5
        pp, pr = torch.zeros(n * 4, 3), torch.zeros(n * 4, 4)

        def step():
            no.pre_physics(spec, b, st["actions"], lo, hi, speed, lambda ids: torch.rand(len(ids), D), default_prop_pos=pp, default_prop_rot=pr)
            no.post_physics(spec, b, st, lo, hi)

        return step


class AnymalFlatBench:
    """Anymal on flat ground (obs 48, 6-term reward, non-sticky done, position targets)."""

    def __init__(self, name, n, dev, seed, reset_frac=None):
        from omniisaacgymenvs_amd.utils.task_util import make_task
        from omniisaacgymenvs_amd.sim import workloads

        self.env, self.task = make_task("Anymal", n, device=dev, seed=seed)
        t = self.task
        st = workloads.anymal_flat_state(n, t.default_dof_pos, device=dev, seed=seed)
        if reset_frac is not None:  # bases at 0.55-0.75 m except `reset_frac` of them below the 0.51 m fall threshold; no stale flags
            bad = _pick(n, reset_frac, dev, seed)
            z = torch.rand(n, device=dev) * 0.2 + 0.55
            z[bad] = 0.4
            st["pos"][:, 2] = z
            st["reset"].zero_()
        workloads.load_anymal_flat_state(t, st)
        t._anymals.apply_resets = False
        self.actions = st["actions"]
        self.post_bytes = 312 + 308  # csrc/anymal.hip header
        self.kernel = "anymal_flat_post_kernel"
        self.workload = f"Anymal (flat) obs(48)+reward+reset fused, {n} envs per GPU, synthetic root/DOF state"
        self.extra_cfg = {"num_dof": 12}

    def pre(self):
        self.task.pre_physics_step(self.actions)

    def post(self):
        self.task.post_physics_step()

    def resets(self):
        return int(self.task._reset_out["count"].item())

    @staticmethod
    def cpu_baseline(name, n, budget_s):
        from omniisaacgymenvs_amd.sim import workloads
        from oracle import anymal_oracle as an

        spec = an.AnymalFlatSpec()
        default = torch.tensor(an.DEFAULT_JOINT_ANGLES).repeat(n, 1)
        st = workloads.anymal_flat_state(n, default, device="cpu", seed=1234)
        b = an.AnymalFlatBuffers(torch.zeros(n, 48), torch.zeros(n), st["reset"].clone(), st["progress"].clone(), st["commands"].clone(),
                                 st["actions"].clone(), st["last_actions"].clone(), st["last_dof_vel"].clone(),
                                 st["current_targets"].clone(), default)
        lower, upper = default[0] - 1.0, default[0] + 1.0
        draw = lambda ids: (torch.rand(len(ids), 12), torch.rand(len(ids), 1), torch.rand(len(ids), 1), torch.rand(len(ids), 1))

        def step():
            an.pre_physics(spec, b, st["actions"], lower, upper, st["pos"], st["rot"], draw=draw)
            an.post_physics(spec, b, st["pos"], st["rot"], st["vel"], st["dof_pos"], st["dof_vel"])

        return step


ADAPTERS = {"Cartpole": CartpoleBench, "TomoDJnewhandCabinet": NewhandCabinetBench, "TomoDJnewhandTendonCabinet": NewhandCabinetBench, "TomoSPCnewhandCabinet": NewhandCabinetBench, "TomoDJCabinet": FrankaCabinetBench, "TomoSPCCabinet": FrankaCabinetBench, "FrankaCabinet": FrankaCabinetBench, "Crazyflie": CrazyflieBench, "Ingenuity": IngenuityBench, "BallBalance": BallBalanceBench, "Quadcopter": QuadcopterBench, "AllegroHand": AllegroHandBench, "Anymal": AnymalFlatBench, "Ant": LocomotionBench, "Humanoid": LocomotionBench, "ShadowHand": HandBench, "AnymalTerrain": AnymalBench}
DEFAULT_ENVS = {"Cartpole": 4096, "TomoDJnewhandCabinet": 1 << 19, "TomoDJnewhandTendonCabinet": 1 << 19, "TomoSPCnewhandCabinet": 1 << 19, "TomoDJCabinet": 1 << 20, "TomoSPCCabinet": 1 << 20, "FrankaCabinet": 1 << 20, "Crazyflie": 1 << 20, "Ingenuity": 1 << 20, "BallBalance": 1 << 20, "Quadcopter": 1 << 20, "AllegroHand": 1 << 20, "Anymal": 1 << 20, "Ant": 1 << 20, "Humanoid": 1 << 19, "ShadowHand": 1 << 20, "AnymalTerrain": 1 << 18}


def measured_copy_gbs(dev: str, footprint_bytes: int) -> dict:
    """Calibration: liboige's plain streaming copy (read + write) at this kernel's footprint, over 1 and 12 lock-step
    sub-arrays (a task kernel walks about a dozen separate (N,k) tensors at once) -- the practical ceilings to read the
    roofline fraction against.  GB/s of read+write traffic, HIP events on the launch stream."""
    from omniisaacgymenvs_amd import _lib

    lib = _lib.load()
    half = max((footprint_bytes // 2) // (16 * 12) * (16 * 12), 16 * 12)
    src = torch.empty(half, dtype=torch.uint8, device=dev).random_(0, 255)
    dst = torch.empty_like(src)
    st = torch.cuda.current_stream().cuda_stream
    out = {}
    for streams in (1, 12):
        for _ in range(5):
            _lib.check(lib.oige_stream_copy(src.data_ptr(), dst.data_ptr(), half, streams, st), "oige_stream_copy")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            lib.oige_stream_copy(src.data_ptr(), dst.data_ptr(), half, streams, st)
        e1.record()
        torch.cuda.synchronize()
        out[f"streams_{streams}"] = 2 * half / (e0.elapsed_time(e1) / 50 * 1e-3) / 1e9
    return out


def cpu_baseline(task_name: str, n: int, budget_s: float = 10.0):
    """The oracle (CPU torch restatement of the reference) timed on the host cores: a bounded sample of about ``budget_s``
    seconds of CPU work (at least 20 steps)."""
    step = ADAPTERS[task_name].cpu_baseline(task_name, n, budget_s)
    step()  # warm-up
    t0 = time.perf_counter()
    k = 0
    while True:
        step()
        k += 1
        el = time.perf_counter() - t0
        if (el > budget_s and k >= 20) or el > 3 * budget_s:
            break
    dt = (time.perf_counter() - t0) / k
    cores = torch.get_num_threads()
    # the same step on ONE thread (SURVEY.md 8d asks for both figures): a short sample, at most ~3 s
    torch.set_num_threads(1)
    try:
        step()
        t1 = time.perf_counter()
        k1 = 0
        while True:
            step()
            k1 += 1
            if time.perf_counter() - t1 > 2.0 or k1 >= 20:
                break
        dt1 = (time.perf_counter() - t1) / k1
    finally:
        torch.set_num_threads(cores)
    return {
        "value": n / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
        "sample": f"{task_name} oracle (CPU torch restatement of the reference) pre+post, {n} envs, {k} steps, {dt*1e3:.1f} ms/step",
        "value_1_thread": n / dt1,
    }


class DryBench:
    """``--task _dry``: no HIP, no GPU -- CPU tensors and a trivial step, so the launcher, the rendezvous, the statistics
    all-reduce and the max-over-ranks timing of the N>1 path can run where there is no GPU (tests/test_bench_launcher_cpu.py).
    Its line is marked ``"data": "dry-run"``: it measures nothing."""

    def __init__(self, name, n, dev, seed, reset_frac=None):
        import types

        g = torch.Generator().manual_seed(seed)
        self.task = types.SimpleNamespace(rew_buf=torch.randn(n, generator=g), reset_buf=(torch.rand(n, generator=g) < 0.1).long(),
                                          progress_buf=torch.randint(0, 1000, (n,), generator=g))
        self.post_bytes, self.kernel, self.extra_cfg = 0, "none", {}
        self.workload = f"dry run of the launcher / collective path, {n} CPU rows per rank, no HIP kernels"

    def pre(self):
        self.task.progress_buf += 1

    def post(self):
        self.task.rew_buf *= 1.0

    def resets(self):
        return int(self.task.reset_buf.sum())


ADAPTERS["_dry"] = DryBench
DEFAULT_ENVS["_dry"] = 1 << 12


def _free_port() -> int:
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n_ranks: int, argv) -> int:
    """``python bench.py --gpus N`` without a launcher around it: start N ranks of this script under
    ``python -m torch.distributed.run`` (the reference's launch, README.md:292-314), one process per GPU, and return the
    launcher's exit code.  The parent has not touched the GPU (``device_count`` does not initialise it) and execs nothing: the
    ranks are children.  With fewer visible GPUs than ranks (a rehearsal on a one-GPU box) the ranks share devices, which
    RCCL cannot do, so the statistics collective falls back to gloo unless OIGE_DIST_BACKEND says otherwise."""
    import subprocess

    env = os.environ.copy()
    if "OIGE_DIST_BACKEND" not in env and torch.cuda.device_count() < n_ranks:
        env["OIGE_DIST_BACKEND"] = "gloo"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    print("[bench] launching", " ".join(cmd), f"(backend {env.get('OIGE_DIST_BACKEND', 'nccl')})", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def init_rank(dry: bool):
    """(dist | None, world, rank, device string, backend name) of this rank; per-rank device as rlgames_train.py:114-120."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    gpu = 0
    if not dry:
        gpu = local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(gpu)
    dev = "cpu" if dry else f"cuda:{gpu}"
    # OIGE_BENCH_FORCE_GROUP=1: rehearsal of the N>1 code path on ONE GPU with the real backend -- a one-rank RCCL
    # communicator, so init_process_group(device_id=...), the per-step async all-reduce, Work.wait() on the streams, the
    # barriers and the MAX reduction of the timing all execute on RCCL (profiles/r03/bench_rccl_one_rank.json)
    force = os.environ.get("OIGE_BENCH_FORCE_GROUP") == "1"
    if world == 1 and not force:
        return None, 1, 0, dev, "none"
    import torch.distributed as dist

    if force and world == 1:
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # OIGE_DIST_BACKEND=gloo: rehearsal of the N>1 path on a box with fewer GPUs than ranks (ranks share devices, the
    # statistics all-reduce runs over gloo on the host); the real runs use RCCL (backend "nccl" on ROCm)
    backend = "gloo" if dry else os.environ.get("OIGE_DIST_BACKEND", "nccl")
    if backend == "nccl":
        dist.init_process_group(backend="nccl", device_id=torch.device(dev))
    else:
        dist.init_process_group(backend=backend)
    return dist, world, rank, dev, backend


def reduce_timing(dist, elapsed: float, dev: str, backend: str) -> dict:
    """MAX over ranks of the timed region (the bench contract) + the spread and a head count of the ranks that took part."""
    if dist is None:
        return {"elapsed": elapsed, "min": elapsed, "max": elapsed, "ranks_seen": 1}
    on = dev if backend == "nccl" else "cpu"
    t = torch.tensor([elapsed, -elapsed], dtype=torch.float64, device=on)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ones = torch.ones(1, dtype=torch.int64, device=on)
    dist.all_reduce(ones, op=dist.ReduceOp.SUM)
    hi, neg_lo = t.tolist()
    return {"elapsed": hi, "min": -neg_lo, "max": hi, "ranks_seen": int(ones.item())}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs", type=int, default=0, help="envs per GPU (default: the BASELINE.json config's per-GPU size)")
    ap.add_argument("--task", default="Ant", choices=sorted(ADAPTERS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="replay the whole step as one HIP graph (RLTask.capture_step): the launch-bound regime of the "
                         "reference's own sizes, e.g. --envs 4096 (not for ShadowHand's DR adapter)")
    ap.add_argument("--no-graph-leg", action="store_true", help="skip the extra graph-replay timing (graph_ms_per_step)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: --envs per GPU (default); strong: the one-GPU env count split over the ranks")
    ap.add_argument("--cpu-envs", type=int, default=0)
    ap.add_argument("--reset-frac", type=float, default=None,
                    help="second workload: only this fraction of the envs terminates per step (0.001 = the reference's steady "
                         "state, about N/1000 resets per step, SURVEY.md a8); default: the r01 workload, 13-25 %% per step")
    ap.add_argument("--event-every", type=int, default=int(os.environ.get("OIGE_BENCH_EVENT_EVERY", "5")),
                    help="HIP-event brackets around pre / post on every k-th timed step (1: every step, as r01 / r02 did)")
    ap.add_argument("--stats-every", type=int, default=0,
                    help="N > 1: the episode-statistics all-reduce on every k-th step (this rank's sums are kept every step).  "
                         "0 (default): the trainer's own cadence, one per epoch = horizon_length steps of the task's "
                         "cfg/train/*PPO.yaml (Ant 16, Humanoid 32, ShadowHand 16, AnymalTerrain 48), at most --steps; 1: every step")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "traffic.json"),
                    help="PMC-derived HBM bytes per launch of the dominant kernel (profiles/run_profiles.sh) -> roofline.traffic")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:  # bare `python bench.py --gpus N`: be the launcher
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    dry = args.task == "_dry"
    dist, world, rank, dev, backend = init_rank(dry)
    if args.gpus != world and rank == 0:
        print(f"[bench] --gpus {args.gpus} but the launcher started {world} rank(s): reporting n_gpus = {world}", file=sys.stderr)
    sync = (lambda: None) if dry else torch.cuda.synchronize

    from omniisaacgymenvs_amd.utils import distributed as od

    n = args.envs or DEFAULT_ENVS[args.task]
    if args.scaling == "strong":
        if n % (64 * world):
            raise SystemExit(f"--scaling strong: {n} envs do not split into {world} ranks of whole 64-env tiles")
        n //= world
    with contextlib.redirect_stdout(sys.stderr):  # the task constructors print like the reference's; stdout carries ONE JSON line
        bench = ADAPTERS[args.task](args.task, n, dev, od.rank_seed(1234, rank), reset_frac=args.reset_frac)

    # HIP events around the post kernel of every timed step.  They are CREATED here: hipEventCreate inside the loop costs
    # ~100 us per step (measured), recording a ready-made one does not.
    class _NoEvent:
        def record(self):
            pass

    if dry:
        ev_pairs = [(_NoEvent(), _NoEvent()) for _ in range(args.steps)]
        ev_start = [_NoEvent() for _ in range(args.steps)]
    else:
        ev_pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        ev_start = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
        # Three event records per step cost 16 us of GPU idle time per step (a 10 us hole between post and the next pre, 6 us
        # between pre and post: profiles/r03/step_gaps_Ant.txt; without them consecutive kernels start 0.0-0.1 us apart), 12 %
        # of Ant's step.  So the per-kernel brackets are taken on every `--event-every`-th step of the timed region and the
        # span from the first step's first launch to the last step's last kernel.
        every = 1 if os.environ.get("OIGE_BENCH_DUMP_STEPS") else max(1, args.event_every)
        if every != 1:
            for i in range(args.steps):
                if i % every:
                    ev_pairs[i] = (_NoEvent() if i else ev_pairs[i][0], _NoEvent() if i + 1 < args.steps else ev_pairs[i][1])
                    ev_start[i] = _NoEvent() if i else ev_start[i]
    after = getattr(bench, "after_post", lambda: None)
    if args.graph:
        if hasattr(bench, "after_post"):  # HandBench: noise arguments per call, wrapper-side leg
            raise SystemExit(f"--graph: the {args.task} adapter does more than the two task calls per step; "
                             "see tests/test_graph_step_gpu.py for its captured step")
        with contextlib.redirect_stdout(sys.stderr):
            bench.task.capture_step(bench.actions)
        bench.pre, bench.post = bench.task.replay_step, (lambda: None)

    # N > 1: the episode-statistics all-reduce runs INSIDE every step (utils/distributed.py: StatsReducer -- device-side fold of
    # the buffers the post kernel just wrote, one 40-byte all_reduce(async_op=True) on the communicator's stream, consumed one
    # step late), as a trainer has it (rlgames_utils.py:72-96).  N = 1 has nothing to exchange and runs no fold.
    # rl_games reduces the episode statistics once per epoch (rlgames_utils.py:72-96: after_print_stats); an epoch is horizon_length
    # steps (cfg/train/AntPPO.yaml:67 16, HumanoidPPO.yaml:66 32, ShadowHandPPO.yaml:67 16, AnymalTerrainPPO.yaml:72 48)
    horizon = {"Ant": 16, "Humanoid": 32, "ShadowHand": 16, "AllegroHand": 16, "AnymalTerrain": 48, "Anymal": 24}.get(args.task, 16)
    stats_every = args.stats_every if args.stats_every > 0 else max(1, min(horizon, args.steps))
    reducer = (od.StatsReducer("cpu" if dry else dev, every=stats_every, force=os.environ.get("OIGE_BENCH_FORCE_GROUP") == "1",
                               consume=os.environ.get("OIGE_STATS_CONSUME", "poll"))
               if dist is not None else None)
    if reducer is not None:
        reducer.attach(bench.task)  # tasks whose post kernel keeps the statistics slots itself: no per-step launch
    coll_host = [0.0]

    def exchange():
        if reducer is not None:
            h = time.perf_counter()
            tk = bench.task
            reducer.step(tk.rew_buf, tk.reset_buf, tk.progress_buf, getattr(tk, "successes", None))
            coll_host[0] += time.perf_counter() - h

    def timed_step(es, e0, e1):
        es.record()
        bench.pre()
        e0.record()  # torch's current stream == the stream the kernels are launched on (_lib.current_stream)
        bench.post()
        e1.record()
        after()
        exchange()

    # the copy calibration (a few dozen streaming copies of the post kernel's footprint) runs first: by the time the W
    # warmup steps start the GPU is at its running clocks, which a 5-step warmup of 0.15 ms steps does not reach by itself
    copy_gbs = measured_copy_gbs(dev, bench.post_bytes * n) if not dry else None
    # first use of torch's reduction kernels (lazy code-object load) and of the communicator (RCCL sets its rings up
    # lazily): keep both out of the timed region -- and IN FRONT of the warmup steps: the reduction's blocking read-backs
    # leave the host core in a low-power state and the step's Python working set cold, and the first step after them costs
    # 150-200 us more on the host (profiles/tools/dbg_first_step.py): 7 % of a 20-step run when it was the first timed step
    t = bench.task
    od.all_reduce_stats(od.local_stats(t.rew_buf, t.reset_buf, t.progress_buf, getattr(t, "successes", None)))
    # What a bracket adds to the kernel it brackets: an ADJACENT pair of records (nothing between them) behind a step's kernels,
    # during the warm-up.  Reported next to the raw brackets (roofline.event_overhead_ms, *_net); rocprofv3's kernel
    # durations (profiles/r03/stats) are the check.
    cal = [] if dry else [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.warmup)]
    for i in range(args.warmup):
        bench.pre()
        bench.post()
        if cal:
            cal[i][0].record()
            cal[i][1].record()
        after()
        exchange()
    if reducer is not None:  # one whole collective as warm-up, then the timed region's statistics start from zero
        reducer.warm()
        reducer.reset()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    coll_host[0] = 0.0
    t0 = time.perf_counter()
    if args.graph:  # one graph launch per step, nothing else on the stream; events only around the whole loop
        ev_start[0].record()
        for _ in range(args.steps):
            bench.pre()
            exchange()
        ev_pairs[-1][1].record()
    else:
        if os.environ.get("OIGE_BENCH_DUMP_STEPS"):  # host-side time of each call of the first steps, to stderr
            host = []
            for es, (e0, e1) in zip(ev_start, ev_pairs):
                h0 = time.perf_counter(); es.record(); h1 = time.perf_counter(); bench.pre(); h2 = time.perf_counter()
                e0.record(); bench.post(); h3 = time.perf_counter(); e1.record(); after(); exchange()
                host.append(tuple(round((b - a) * 1e6, 1) for a, b in ((h0, h1), (h1, h2), (h2, h3))))
            print("host us (record, pre, post):", host[:8], file=sys.stderr)
        else:
            for es, (e0, e1) in zip(ev_start, ev_pairs):
                timed_step(es, e0, e1)
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    timing = reduce_timing(dist, time.perf_counter() - t0, dev, backend)
    elapsed = timing["elapsed"]
    # episode statistics [sum reward, n done, sum episode length of done envs, successes, n envs]: the path's only
    # cross-rank exchange, one 40-byte all-reduce over RCCL (utils/distributed.py).  Once per run and not a step of the
    # path, so outside the bracket that times EXACTLY K steps (r01 had it inside: 5 % of a 20-step run).
    t = bench.task
    if reducer is not None:  # whole-job means over the K timed steps, from the per-step collectives (drained here, outside the bracket)
        job_stats = reducer.finish()
    else:
        job_stats = od.all_reduce_stats(od.local_stats(t.rew_buf, t.reset_buf, t.progress_buf, getattr(t, "successes", None)))

    timed = []
    if dry:
        post_ms = pre_ms = gpu_span_ms = None
    elif args.graph:  # the kernels are nodes of one graph: the whole replayed step stands in for the post kernel (a lower bound)
        post_ms, pre_ms = ev_start[0].elapsed_time(ev_pairs[-1][1]) / len(ev_pairs), None
    else:
        timed = [(es, e0, e1) for es, (e0, e1) in zip(ev_start, ev_pairs)
                 if not any(isinstance(e, _NoEvent) for e in (es, e0, e1))]
        post_ms = sum(a.elapsed_time(b) for _, a, b in timed) / max(1, len(timed))
        pre_ms = sum(es.elapsed_time(e0) for es, e0, _ in timed) / max(1, len(timed))
    event_overhead_ms = None
    if not dry and cal:
        gaps = sorted(a.elapsed_time(b) for a, b in cal)
        event_overhead_ms = gaps[len(gaps) // 2]
    if not dry:
        gpu_span_ms = ev_start[0].elapsed_time(ev_pairs[-1][1]) / len(ev_pairs)  # first pre launch -> last post end, per step
        if os.environ.get("OIGE_BENCH_DUMP_STEPS") and not args.graph:  # per-step (pre, post, gap to the next step) in us, to stderr
            rows = []
            for i, (es, (e0, e1)) in enumerate(zip(ev_start, ev_pairs)):
                gap = e1.elapsed_time(ev_start[i + 1]) if i + 1 < len(ev_start) else 0.0
                rows.append((round(es.elapsed_time(e0) * 1e3, 1), round(e0.elapsed_time(e1) * 1e3, 1), round(gap * 1e3, 1)))
            print("steps (pre us, post us, gap us):", rows[:40], file=sys.stderr)

    # second leg (default line): the same K steps replayed as ONE HIP graph per step (RLTask.capture_step), so the line
    # shows what launch gaps and Python marshalling cost next to the eager figure.  Outside the timed region above.
    graph_ms = graph_note = None
    if not dry and not args.graph and not args.no_graph_leg:
        graph_ms, graph_note = graph_leg(bench, args.steps, dist)

    traffic = traffic_source = pre_traffic = None
    if args.traffic_json and os.path.exists(args.traffic_json) and not dry:
        with open(args.traffic_json) as f:
            recs = json.load(f)
        rec = recs.get(args.task)
        pre_rec = recs.get(args.task + ":pre")  # the pre-physics kernels, captured on the default workload
        if pre_rec and args.reset_frac is None:
            pre_traffic = pre_rec["hbm_bytes_per_launch"] * n / pre_rec["envs"]
        if rec:  # measured at rec["envs"] envs per launch; the kernels stream, so bytes scale with the env count
            traffic = rec["hbm_bytes_per_launch"] * n / rec["envs"]
            traffic_source = (f"{os.path.relpath(args.traffic_json, ROOT)}: rocprofv3 --pmc 2 x FETCH_SIZE + WRITE_SIZE of {rec.get('kernel', 'the post kernel')}"
                              f" at {rec['envs']} envs ({rec.get('captured', 'profiles/run_profiles.sh')}), scaled to this run's env count; not measured in this run")

    if rank == 0:
        cfg = {"workload": bench.workload, "envs_per_gpu": n, "resets_per_step": bench.resets(),
               "step": "pre_physics_step + post_physics_step via RLTask API (physics delegated)"}
        cfg.update(bench.extra_cfg)
        if args.graph:
            cfg["step"] += ", replayed as one HIP graph (RLTask.capture_step)"
        cfg["episode_stats"] = {k: round(v, 4) for k, v in job_stats.items()}
        out = {
            "metric": "fused obs+reward+reset env-steps/s",
            "value": world * n * args.steps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "dry-run" if dry else "synthetic",
            "config": cfg,
            "ranks_seen": timing["ranks_seen"],
            "rank_ms_per_step": {"min": timing["min"] / args.steps * 1e3, "max": timing["max"] / args.steps * 1e3},
            "collective_backend": {"nccl": "rccl"}.get(backend, backend),
        }
        if reducer is not None:
            out["collective"] = {"what": "episode-statistics all_reduce(SUM, async_op=True) of 5 f64 inside the timed steps, consumed by a later step",
                                 "bytes": 40, "every": reducer.every, "calls_in_timed_region": reducer.collectives,
                                 "rank_sums": "kept by the post kernel (per-tile slots)" if reducer._fused else "one oige_episode_stats_accumulate launch per step",
                                 "host_ms_per_step": coll_host[0] / args.steps * 1e3}
            out["collective_ms_per_step"] = coll_host[0] / args.steps * 1e3
        if graph_ms is not None or graph_note is not None:
            out["graph_ms_per_step"] = graph_ms
            out["graph_value"] = None if graph_ms is None else world * n / (graph_ms * 1e-3)
            if graph_note:
                out["graph_note"] = graph_note
        if not dry:
            achieved = bench.post_bytes * n / (post_ms * 1e-3) / 1e9
            out["roofline"] = {
                "bound": "hbm", "kernel": bench.kernel + (" (+ pre_physics: whole replayed graph timed)" if args.graph else ""), "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                "algorithmic_bytes_per_env": bench.post_bytes, "avg_kernel_ms": post_ms,
                "pre_physics_ms": pre_ms, "gpu_ms_per_step": gpu_span_ms,
                "event_brackets": f"every {max(1, args.event_every)}. timed step ({len(timed)} of {args.steps})",
                "event_overhead_ms": event_overhead_ms,
                "measured_copy_gbs": copy_gbs,
                "frac_of_measured_copy_12_streams": achieved / copy_gbs["streams_12"],
            }
            if hasattr(bench, "pre_bytes") and pre_ms:  # the other half of the step, and the step as a whole
                resets = cfg["resets_per_step"]
                pb = float(bench.pre_bytes(resets))
                pre_gbs = pb / (pre_ms * 1e-3) / 1e9
                step_gbs = (pb + bench.post_bytes * n) / (gpu_span_ms * 1e-3) / 1e9
                out["roofline"]["pre"] = {"what": "everything the step launches before the post kernel (compaction scan, resets, action scaling)",
                                          "algorithmic_bytes": pb, "ms": pre_ms, "achieved": pre_gbs, "frac": pre_gbs / HBM_PEAK_GBS,
                                          "traffic": pre_traffic}
                if event_overhead_ms is not None and not args.graph:
                    net_post, net_pre = max(post_ms - event_overhead_ms, 1e-9), max(pre_ms - event_overhead_ms, 1e-9)
                    out["roofline"]["net_of_event_overhead"] = {
                        "what": "the brackets minus one adjacent-record gap (event_overhead_ms); compare with rocprofv3 kernel durations",
                        "avg_kernel_ms": net_post, "frac": bench.post_bytes * n / (net_post * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "pre_physics_ms": net_pre, "pre_frac": pb / (net_pre * 1e-3) / 1e9 / HBM_PEAK_GBS}
                out["roofline"]["step"] = {"what": "pre + post algorithmic bytes over gpu_ms_per_step (first launch to last kernel end)",
                                           "algorithmic_bytes": pb + bench.post_bytes * n, "ms": gpu_span_ms, "achieved": step_gbs,
                                           "frac": step_gbs / HBM_PEAK_GBS}
            if args.reset_frac is not None:
                cfg["reset_frac"] = args.reset_frac
        if not args.no_cpu_baseline and world == 1 and not dry:
            out["cpu_baseline"] = cpu_baseline(args.task, args.cpu_envs or n)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def graph_leg(bench, steps: int, dist):
    """(ms per step, note): ``steps`` replays of the captured whole step, max over ranks; (None, reason) where the
    adapter's step is more than the two task calls or the task refuses capture."""
    if hasattr(bench, "graph_capture"):
        capture, replay = bench.graph_capture, bench.graph_replay
    elif hasattr(bench, "after_post"):
        return None, "adapter runs wrapper-side legs outside the two task calls"
    else:
        capture, replay = (lambda: bench.task.capture_step(bench.actions)), (lambda: bench.task.replay_step())
    try:
        with contextlib.redirect_stdout(sys.stderr):
            capture()
    except NotImplementedError as e:
        return None, str(e)
    for _ in range(5):
        replay()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        replay()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    return el / steps * 1e3, None


if __name__ == "__main__":
    main()
