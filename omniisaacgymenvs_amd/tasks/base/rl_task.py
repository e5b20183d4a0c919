"""RLTask: the task base class contract of the reference, kept so task code and the
VecEnvRLGames wrapper work unchanged (/root/reference/omniisaacgymenvs/tasks/base/rl_task.py).

Kept: constructor signature and attribute names (:55-156), the buffer dtype contract of
``cleanup`` (:158-167: obs/states/rew fp32, reset_buf int64 = 1, progress_buf int64 = 0),
``reset`` (:274-276), the ``post_physics_step`` stage order (:278-298) and the ``world`` /
``cfg`` properties (:300-317).  Out of scope and absent: GridCloner/USD scene cloning,
cameras, replicator (they belong to the delegated simulator).  ``set_up_scene`` only lays the
env origins out on the same square grid the cloner would use.
"""
from __future__ import annotations

import gc
import math
import os
from typing import Optional

import numpy as np
import torch

from ...utils.domain_randomization.randomize import Randomizer
from ...utils.spaces import Box


def quiesce_gc() -> bool:
    """Collect now and switch the cyclic collector off; returns whether it was on (the caller switches it back on).

    torch.cuda.graph() no longer collects on entry (torch >= 2.6: ``force_cudagraph_gc`` is off), and a collection that
    starts inside a stream capture may drop an older ``CUDAGraph`` that sat in a reference cycle: its destructor calls
    ``hipGraphExecDestroy`` while the capture is open, which HIP rejects, and an error in a destructor ends the process
    (SIGABRT, seen once per some hundred captures when several tasks were captured in one process)."""
    on = gc.isenabled()
    gc.collect()
    gc.disable()
    return on


# OIGE_CHECK_BLOCKS=1 (tests): every cache hit also rebuilds the block and compares it with the cached one, field for field --
# a dependency missing from a block's key then fails loudly instead of handing the kernels a stale pointer.
_CHECK_BLOCKS = os.environ.get("OIGE_CHECK_BLOCKS") == "1"
_VOLATILE_FIELDS = ("step",)  # written by the caller on every call, behind the cache


def _assert_same_block(name, cached, fresh, volatile=()) -> None:
    import ctypes

    if isinstance(cached, ctypes.Structure):
        for field in cached._fields_:
            f = field[0]
            if f in _VOLATILE_FIELDS or f in volatile:
                continue
            a, b = getattr(cached, f), getattr(fresh, f)
            if isinstance(a, ctypes.Array):
                a, b = bytes(a), bytes(b)
            elif isinstance(a, ctypes.Structure):
                _assert_same_block(f"{name}.{f}", a, b)
                continue
            if a != b and not (a != a and b != b):
                raise AssertionError(f"cached argument block '{name}' is stale: field {f} is {a!r}, a fresh build gives {b!r}")
    elif isinstance(cached, dict):
        for k in cached:
            _assert_same_block(f"{name}[{k!r}]", cached[k], fresh[k])
    elif isinstance(cached, (tuple, list)):
        for i, (a, b) in enumerate(zip(cached, fresh)):
            _assert_same_block(f"{name}[{i}]", a, b)
    elif isinstance(cached, torch.Tensor):
        if cached is not fresh and cached.data_ptr() != fresh.data_ptr():
            raise AssertionError(f"cached '{name}' is stale: another tensor")
    elif cached != fresh:
        raise AssertionError(f"cached '{name}' is stale: {cached!r} != {fresh!r}")


class RLTask:
    def __init__(self, name, env, offset=None) -> None:
        self._name = name
        self._offset = offset

        if not hasattr(self, "_cfg"):
            self._cfg = {}
        if not hasattr(self, "_task_cfg"):
            self._task_cfg = {}

        self._rand_seed = self._cfg["seed"]
        self.test = self._cfg["test"]
        self._device = self._cfg["sim_device"]

        self._dr_randomizer = Randomizer(self._cfg, self._task_cfg)

        self.enable_cameras = self._task_cfg.get("sim", {}).get("enable_cameras", False)
        self.randomize_actions = False
        self.randomize_observations = False

        env_cfg = self._task_cfg.get("env", {})
        self.clip_obs = env_cfg.get("clipObservations", np.inf)
        self.clip_actions = env_cfg.get("clipActions", np.inf)
        self.rl_device = self._cfg.get("rl_device", "cuda:0")

        self.control_frequency_inv = env_cfg.get("controlFrequencyInv", 1)
        self.rendering_interval = self._task_cfg.get("renderingInterval", 1)

        self._env = env
        self.is_extension = False
        # fast path of VecEnvRLGames (envs/vec_env_rlgames.py): tasks whose pre_physics kernel can fold the wrapper's
        # torch.clamp(actions, -clip, clip) (vec_env_rlgames.py:77) set supports_action_clamp; the wrapper then turns
        # clamp_actions_in_kernel on instead of clamping with torch
        if not hasattr(self, "supports_action_clamp"):
            self.supports_action_clamp = False
        self.clamp_actions_in_kernel = False

        for attr, default in (
            ("_num_agents", 1),
            ("_num_states", 0),
            ("_num_envs", 0),
            ("_num_actions", 0),
            ("_num_observations", 0),
        ):
            if not hasattr(self, attr):
                setattr(self, attr, default)

        if not hasattr(self, "action_space"):
            self.action_space = Box(-np.ones(self._num_actions, dtype=np.float32), np.ones(self._num_actions, dtype=np.float32))
        if not hasattr(self, "observation_space"):
            hi = np.full(self._num_observations, np.inf, dtype=np.float32)
            self.observation_space = Box(-hi, hi)
        if not hasattr(self, "state_space"):
            hi = np.full(self._num_states, np.inf, dtype=np.float32)
            self.state_space = Box(-hi, hi)

        self._env_pos = torch.tensor([], dtype=torch.float32, device=self._device)
        self.cleanup()

    # ------------------------------------------------------------------ buffers
    def cleanup(self) -> None:
        """Allocate the RL buffers with the reference's dtypes and initial values (:158-167)."""
        dev = self._device
        self.obs_buf = torch.zeros((self._num_envs, self._num_observations), device=dev, dtype=torch.float)
        self.states_buf = torch.zeros((self._num_envs, self._num_states), device=dev, dtype=torch.float)
        self.rew_buf = torch.zeros(self._num_envs, device=dev, dtype=torch.float)
        self.reset_buf = torch.ones(self._num_envs, device=dev, dtype=torch.long)
        self.progress_buf = torch.zeros(self._num_envs, device=dev, dtype=torch.long)
        self.extras = {}

    # ------------------------------------------------------------------ scene
    def set_up_scene(self, scene=None, **_ignored) -> None:
        """Env origins on the square grid GridCloner(spacing) produces (:189-207); no USD."""
        spacing = getattr(self, "_env_spacing", 1.0)
        n = self._num_envs
        per_row = max(int(math.ceil(math.sqrt(max(n, 1)))), 1)
        idx = torch.arange(n, dtype=torch.float32)
        rows = torch.floor(idx / per_row)
        cols = idx - rows * per_row
        off = (per_row - 1) * spacing / 2.0
        pos = torch.stack([rows * spacing - off, cols * spacing - off, torch.zeros(n)], dim=-1)
        self._env_pos = pos.to(device=self._device, dtype=torch.float)

    def initialize_views(self, scene=None) -> None:
        self.set_up_scene(scene)

    @property
    def default_base_env_path(self) -> str:
        return "/World/envs"

    @property
    def default_zero_env_path(self) -> str:
        return f"{self.default_base_env_path}/env_0"

    # ------------------------------------------------------------------ HIP-graph step (launch-bound sizes)
    graph_capturable = True   # tasks whose step has host-side branches that change from step to step set this to a reason string
    _capturing = False
    _step_dev = None

    def _graph_step_ptr(self):
        """``reset_io.step_ptr``: the device-side Philox step counter while a step is being captured, else NULL."""
        return self._step_dev.data_ptr() if self._capturing else None

    def _advance_step_counter(self) -> None:
        self._step_count = getattr(self, "_step_count", 0) + 1
        if self._step_dev is not None and not self._capturing:
            self._step_dev += 1  # an eager step between replays: keep the captured graph's counter in step

    def set_reset_scatter(self, mode) -> int:
        """Tell every synthetic view of the task (and the views nested in them: fingers, props) how to consume the
        ``(values, indices, count)`` reset hand-over: ``True`` host read of the count, ``False`` drop, ``"device"``
        sync-free masked scatter (what a captured step needs).  Real PhysX views have no such switch; returns how many
        views were set."""
        done = 0
        for v in list(vars(self).values()):
            if hasattr(v, "apply_resets"):
                v.apply_resets = mode
                done += 1
                for vv in vars(v).values():
                    if hasattr(vv, "apply_resets"):
                        vv.apply_resets = mode
                        done += 1
        return done

    def capture_step(self, actions: torch.Tensor, split: bool = False) -> None:
        """Capture ``pre_physics_step(actions); post_physics_step()`` into one HIP graph (torch.cuda.CUDAGraph on ROCm), or
        with ``split=True`` into two (``replay_pre()`` / ``replay_post()``), for a physics step that runs between them.

        At the reference's own sizes (4096 envs) the step is bound by launches and Python marshalling, not by the GPU; a
        replay is one graph launch.  ``actions`` is the static input buffer: refill it in place (``actions.copy_(...)``)
        before each ``replay_step()``.  The sync-free, fused path is required; the view setters must consume
        ``(values, indices, count)`` on the device (synthetic views: ``apply_resets="device"`` or ``False``).  By-value kernel
        arguments are baked into the graph, so the Philox step counter moves to a device scalar that the graph itself
        advances (``reset_io.step_ptr``), ``OIGE_PRE_COUNTS_VALID`` is never baked in, and scalars edited on the task afterwards
        (``task.heading_weight = ...``) or rebound buffers need a new capture.  One eager step runs first as warm-up (lazy
        code-object loads are illegal during capture): this call advances the task by one step and leaves the second one
        captured, not executed."""
        if self.graph_capturable is not True:
            raise NotImplementedError(f"{type(self).__name__}.capture_step: {self.graph_capturable}")
        if not (getattr(self, "sync_free", False) and getattr(self, "fused", False)):
            raise RuntimeError("capture_step needs the sync-free fused path (task.sync_free = task.fused = True)")
        if not self.world.is_playing():
            raise RuntimeError("capture_step: the world is not playing")
        from ... import _lib

        dev = self.obs_buf.device
        _lib.require_device_tensor(actions, torch.float32, (self._num_envs, self.num_actions), "actions")
        # graphs of an earlier capture of this task are released NOW, before the window opens (their destructors call
        # hipGraphExecDestroy, which HIP rejects while a capture is open)
        self._graph = self._graph_post = None
        self._step_dev = torch.full((1,), getattr(self, "_step_count", 0), dtype=torch.int64, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        gc_on = None
        try:
            self._capturing = True
            gc_on = quiesce_gc()
            with torch.cuda.stream(side):  # warm-up: one real step on the side stream
                self._step_dev += 1
                self.pre_physics_step(actions)
                self.post_physics_step()
            torch.cuda.current_stream(dev).wait_stream(side)
            graph, graph_post = torch.cuda.CUDAGraph(), None
            count0 = getattr(self, "_step_count", 0)
            with torch.cuda.graph(graph, stream=side):
                self._step_dev += 1
                self.pre_physics_step(actions)
                if not split:
                    self.post_physics_step()
            if split:
                graph_post = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph_post, stream=side):
                    self.post_physics_step()
            self._step_count = count0  # the captured step has not run yet
        finally:
            self._capturing = False
            if gc_on:
                gc.enable()
        self._graph, self._graph_post, self._graph_actions = graph, graph_post, actions

    def replay_step(self):
        """One captured step; returns what ``post_physics_step`` returns.  ``task._graph_actions`` holds the actions."""
        self.replay_pre()
        if self._graph_post is not None:
            return self.replay_post()
        return self.obs_buf, self.rew_buf, self.reset_buf, self.extras

    def replay_pre(self) -> None:
        """The captured ``pre_physics_step`` (with ``split=False``: the whole step)."""
        self._graph.replay()
        self._step_count = getattr(self, "_step_count", 0) + 1
        self._counts_key = None

    def replay_post(self):
        """The captured ``post_physics_step`` of a ``split=True`` capture."""
        self._graph_post.replay()
        return self.obs_buf, self.rew_buf, self.reset_buf, self.extras

    # ------------------------------------------------------------------ cached C-ABI argument blocks
    # A step's ctypes blocks (params / buffers / state / reset_io) hold the same pointers and scalars step after step;
    # rebuilding and re-validating them was 30-45 us of host time per step, the whole step at the reference's own sizes
    # (VERDICT r2 #8).  A block is rebuilt only when its key changes: the IDENTITY of every tensor it points into (the cache
    # keeps the tensors alive, so an id cannot be handed out again while it is a key) and the value of every scalar.  A
    # rebound buffer (``task.obs_buf = ...``) or an edited scalar (``task.heading_weight = 0.3``) therefore takes effect on
    # the next call, as before.  Not seen: a tensor whose STORAGE is swapped in place (``.set_()``, ``.resize_()``,
    # ``.data = ...``) -- call ``task.invalidate_blocks()`` after such a change.
    def _block(self, name, tensors, scalars, build, volatile=()):
        """``volatile``: fields the caller rewrites on every call behind the cache (not compared in check mode)."""
        key = (*map(id, tensors), *scalars)
        blocks = self.__dict__.get("_blocks")
        if blocks is None:
            blocks = self._blocks = {}
        hit = blocks.get(name)
        if hit is not None and hit[0] == key:
            if _CHECK_BLOCKS:
                _assert_same_block(name, hit[2], build(), volatile)
            return hit[2]
        blk = build()
        blocks[name] = (key, tensors, blk)
        return blk

    def invalidate_blocks(self) -> None:
        self.__dict__.pop("_blocks", None)

    # ------------------------------------------------------------------ OIGE_PRE_COUNTS_VALID bookkeeping
    # The post kernels leave the per-tile counts of the reset flags they wrote; the next pre_physics_step may skip its
    # recount only while those flags are untouched.  The key holds the flag TENSORS (not their addresses: a rebound
    # reset_buf is a different object even when the allocator hands the old address out again, and holding the old one
    # keeps its memory from being recycled) and their torch version counters.  Writers torch cannot see (another library
    # writing through the raw pointer) must set ``task._counts_key = None`` themselves.
    _counts_key = None

    def _mark_counts(self, *flag_bufs) -> None:
        self._counts_key = tuple((b, b._version) for b in flag_bufs)

    def _counts_valid(self, *flag_bufs) -> bool:
        k = self._counts_key
        return (k is not None and not self._capturing and len(k) == len(flag_bufs)
                and all(t is b and v == b._version for (t, v), b in zip(k, flag_bufs)))

    # ------------------------------------------------------------------ RL interface
    def reset(self) -> None:
        """Flag every env for reset (:274-276; rebinds, as the reference does)."""
        self.reset_buf = torch.ones_like(self.reset_buf)
        self._counts_key = None

    def post_physics_step(self):
        """progress += 1, then obs -> states -> metrics -> done -> extras (:289-298).

        Task classes with a fused HIP kernel override this with one launch that keeps the
        same order; this generic version drives the individually callable methods.
        """
        self.progress_buf[:] += 1
        if self._env.world.is_playing():
            self.get_observations()
            self.get_states()
            self.calculate_metrics()
            self.is_done()
            self.get_extras()
        return self.obs_buf, self.rew_buf, self.reset_buf, self.extras

    def get_states(self):
        return self.states_buf

    def get_extras(self):
        return self.extras

    def pre_physics_step(self, actions) -> None:
        raise NotImplementedError

    def get_observations(self):
        raise NotImplementedError

    def calculate_metrics(self) -> None:
        raise NotImplementedError

    def is_done(self) -> None:
        raise NotImplementedError

    def post_reset(self) -> None:
        pass

    # ------------------------------------------------------------------ properties
    @property
    def name(self) -> str:
        return self._name

    @property
    def world(self):
        return self._env.world

    @property
    def cfg(self):
        return self._cfg

    @property
    def device(self):
        return self._device

    @property
    def num_envs(self) -> int:
        return self._num_envs

    @property
    def num_actions(self) -> int:
        return self._num_actions

    @property
    def num_observations(self) -> int:
        return self._num_observations

    @property
    def num_states(self) -> int:
        return self._num_states

    @property
    def num_agents(self) -> int:
        return self._num_agents

    def set_is_extension(self, is_extension) -> None:
        self.is_extension = is_extension

    def ensure_tensor(self, var) -> torch.Tensor:
        if isinstance(var, torch.Tensor):
            return var
        if isinstance(var, np.ndarray):
            return torch.from_numpy(var).to(device=self._device)
        if isinstance(var, list):
            return torch.tensor(var, device=self._device)
        raise TypeError("Unsupported type for conversion to Tensor")
