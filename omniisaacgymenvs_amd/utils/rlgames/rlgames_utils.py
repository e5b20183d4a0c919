"""RLGPUAlgoObserver -- host-side mirror of /root/reference/omniisaacgymenvs/utils/rlgames/rlgames_utils.py:39-96 with the
episode statistics kept on the device (liboige_hip.so, csrc/observer.hip).  Same methods, same writer calls:
``after_init / process_infos / after_clear_stats / after_print_stats``.

The reference appends ``infos["episode"]`` to a list every step and, per epoch and key, grows a tensor with ``torch.cat`` once
per appended step before ``torch.mean`` (:73-85).  Here ``process_infos`` is one launch that adds the step's values to running
f64 sums (device tensors: by pointer, no sync; Python numbers: on the host), and ``after_print_stats`` one launch plus the
single read-back the writer needs.  No torch/CPU fallback for device values.

rl_games is not importable here: ``AverageMeter`` restates the slice of ``rl_games.algos_torch.torch_ext.AverageMeter`` the
observer touches (``clear``, ``current_size``, ``get_mean``, ``update``); the ``AlgoObserver`` base class is not needed.
"""
from __future__ import annotations

import ctypes as C

import torch

from ... import _lib


class AverageMeter(torch.nn.Module):
    """Windowed running mean as rl_games' torch_ext.AverageMeter (in_shape, max_size): restated, parity unpinned."""

    def __init__(self, in_shape, max_size):
        super().__init__()
        self.max_size = max_size
        self.current_size = 0
        self.register_buffer("mean", torch.zeros(in_shape, dtype=torch.float32))

    def update(self, values):
        size = values.size()[0]
        if size == 0:
            return
        new_mean = torch.mean(values.float(), dim=0)
        size = min(size, self.max_size)
        old_size = min(self.max_size - size, self.current_size)
        size_sum = old_size + size
        self.current_size = size_sum
        self.mean = (self.mean * old_size + new_mean * size) / size_sum

    def clear(self):
        self.current_size = 0
        self.mean.fill_(0)

    def __len__(self):
        return self.current_size

    def get_mean(self):
        return self.mean.squeeze(0).cpu().numpy()


class RLGPUAlgoObserver:
    """Allows us to log stats from the env along with the algorithm running stats."""

    def __init__(self):
        pass

    def after_init(self, algo):
        self.algo = algo
        self.mean_scores = AverageMeter(1, self.algo.games_to_track).to(self.algo.ppo_device)
        self.direct_info = {}
        self.writer = self.algo.writer
        self._keys = None          # key order of the first appended episode dict (the reference iterates ep_infos[0], :74)
        self._appended = 0         # len(self.ep_infos) of the reference
        self._sums = self._counts = self._means = None
        self._host_sums, self._host_counts = {}, {}

    @property
    def ep_infos(self):
        """Length-compatible stand-in for the reference's list: truthy while something was appended since the last print."""
        return [None] * self._appended

    def _ensure(self, episode: dict) -> None:
        if self._appended > 0 or self._keys == list(episode.keys()):
            return  # the key order is that of the first dict appended since the last print (the reference iterates ep_infos[0])
        self._keys = list(episode.keys())
        if len(self._keys) > _lib.OBSERVER_MAX_KEYS:
            raise ValueError(f"RLGPUAlgoObserver: {len(self._keys)} episode keys, at most {_lib.OBSERVER_MAX_KEYS} supported")
        dev = torch.device(self.algo.device)
        k = max(len(self._keys), 1)
        self._sums = torch.zeros(k, dtype=torch.float64, device=dev)
        self._counts = torch.zeros(k, dtype=torch.int64, device=dev)
        self._means = torch.zeros(k, dtype=torch.float32, device=dev)
        self._host_sums = {key: 0.0 for key in self._keys}
        self._host_counts = {key: 0 for key in self._keys}

    def process_infos(self, infos, done_indices):
        assert isinstance(infos, dict), "RLGPUAlgoObserver expects dict info"
        if "episode" in infos:
            episode = infos["episode"]
            self._ensure(episode)
            batch, keep = _lib.ObserverBatch(), []
            batch.num_keys = len(self._keys)
            dev = self._sums.device
            for i, key in enumerate(self._keys):
                v = episode[key]  # KeyError if a later dict lacks a key of the first one, as in the reference (:78)
                if not isinstance(v, torch.Tensor):
                    self._host_sums[key] += float(torch.tensor([v], dtype=torch.float32)[0])  # torch.Tensor([x]) rounds to fp32 (:79)
                    self._host_counts[key] += 1
                    continue
                if v.dim() > 1:
                    raise RuntimeError(f"RLGPUAlgoObserver: episode info {key!r} has {v.dim()} dimensions; torch.cat in the reference "
                                       "needs 0-d or 1-d values")
                if v.device != dev or v.dtype != torch.float32 or not v.is_contiguous():
                    v = v.to(device=dev, dtype=torch.float32).contiguous()  # .to(self.algo.device) (:82)
                    keep.append(v)
                batch.values[i] = v.data_ptr()
                batch.lengths[i] = v.numel()
            rc = _lib.load().oige_observer_accumulate(C.byref(batch), self._sums.data_ptr(), self._counts.data_ptr(),
                                                      _lib.current_stream(dev))
            _lib.check(rc, "oige_observer_accumulate")
            self._appended += 1

        if len(infos) > 0 and isinstance(infos, dict):  # allow direct logging from env
            self.direct_info = {}
            for k, v in infos.items():
                # only log scalars
                if isinstance(v, float) or isinstance(v, int) or (isinstance(v, torch.Tensor) and len(v.shape) == 0):
                    self.direct_info[k] = v

    def after_clear_stats(self):
        self.mean_scores.clear()

    def after_print_stats(self, frame, epoch_num, total_time):
        if self._appended:
            dev = self._sums.device
            if any(self._host_counts.values()):  # some values were Python numbers: merge the host and device halves
                sums, counts = self._sums.cpu(), self._counts.cpu()
                self._sums.zero_()
                self._counts.zero_()
                values = []
                for i, key in enumerate(self._keys):
                    total, count = float(sums[i]) + self._host_sums[key], int(counts[i]) + self._host_counts[key]
                    values.append(torch.tensor(total / count if count else float("nan"), dtype=torch.float32))
                    self._host_sums[key], self._host_counts[key] = 0.0, 0
            else:  # one launch (means + clear) and the epoch's one read-back: the writer needs host numbers
                rc = _lib.load().oige_observer_means(self._sums.data_ptr(), self._counts.data_ptr(), self._means.data_ptr(),
                                                     len(self._keys), 1, _lib.current_stream(dev))
                _lib.check(rc, "oige_observer_means")
                values = list(self._means.cpu()[: len(self._keys)])
            for key, value in zip(self._keys, values):
                self.writer.add_scalar("Episode/" + key, value, epoch_num)
            self._appended = 0

        for k, v in self.direct_info.items():
            self.writer.add_scalar(f"{k}/frame", v, frame)
            self.writer.add_scalar(f"{k}/iter", v, epoch_num)
            self.writer.add_scalar(f"{k}/time", v, total_time)

        if self.mean_scores.current_size > 0:
            mean_scores = self.mean_scores.get_mean()
            self.writer.add_scalar("scores/mean", mean_scores, frame)
            self.writer.add_scalar("scores/iter", mean_scores, epoch_num)
            self.writer.add_scalar("scores/time", mean_scores, total_time)


# rl_games' two registries, as far as rlgames_train.py:68-70 uses them: the names are kept so that
# ``env_configurations.register("rlgpu", {"vecenv_type": "RLGPU", "env_creator": lambda **kwargs: env})`` reads the same.
class _EnvConfigurations:
    def __init__(self):
        self.configurations = {}

    def register(self, name, config):
        self.configurations[name] = config


env_configurations = _EnvConfigurations()


class RLGPUEnv:
    """The rl_games vec-env adapter over ``VecEnvRLGames`` (rlgames_utils.py:99-124): same five methods, same
    ``get_env_info`` keys.  ``vecenv.IVecEnv`` is an interface-only base class in rl_games; without it this is a plain class."""

    def __init__(self, config_name, num_actors, **kwargs):
        self.env = env_configurations.configurations[config_name]["env_creator"](**kwargs)

    def step(self, action):
        return self.env.step(action)

    def reset(self):
        return self.env.reset()

    def get_number_of_agents(self):
        # the reference forwards to the omni.isaac.gym base class (not in its tree); one agent per env for every task here
        get = getattr(self.env, "get_number_of_agents", None)
        return get() if get is not None else 1

    def get_env_info(self):
        info = {"action_space": self.env.action_space, "observation_space": self.env.observation_space}
        if self.env.num_states > 0:
            info["state_space"] = self.env.state_space
            print(info["action_space"], info["observation_space"], info["state_space"])
        else:
            print(info["action_space"], info["observation_space"])
        return info
