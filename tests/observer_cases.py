"""Shared by the observer tests: a recording writer and a seeded stream of ``infos`` dicts shaped like the tasks' extras
(0-d views of one buffer, a 1-d tensor, a Python float, direct scalars)."""
import torch


class Writer:
    def __init__(self):
        self.calls = []

    def add_scalar(self, tag, value, step):
        self.calls.append((tag, float(value), float(step)))


def infos_stream(steps, seed, device="cpu", with_host_number=True, with_vector=True):
    g = torch.Generator().manual_seed(seed)
    keys = ["rew_lin_vel_xy", "rew_ang_vel_z", "rew_torque", "terrain_level"]
    out = []
    for t in range(steps):
        buf = (torch.randn(len(keys), generator=g) * 3 + 1).to(device)
        ep = {k: buf[i] for i, k in enumerate(keys)}                     # 0-d views of one buffer, as the task mirrors bind them
        if with_vector:
            ep["per_term"] = torch.rand(1 + t % 5, generator=g).to(device)   # 1-d, varying length
        if with_host_number:
            ep["python_number"] = 0.1 * t
        infos = {"episode": ep, "consecutive_successes": torch.tensor(float(t), device=device), "time_outs": torch.zeros(8, device=device),
                 "successes_py": 2.5 + t}
        out.append(infos if t % 7 != 3 else {"time_outs": torch.zeros(8, device=device)})   # some steps carry no episode dict
    return out
