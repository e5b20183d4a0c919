"""ORACLE (test infrastructure only) -- RLGPUAlgoObserver episode statistics on CPU torch.

Restates /root/reference/omniisaacgymenvs/utils/rlgames/rlgames_utils.py:52-96: ``process_infos`` appends
``infos["episode"]`` and keeps the scalar entries of ``infos`` as ``direct_info``; ``after_print_stats`` concatenates, per key
of the first appended dict, every appended value (numbers become ``torch.Tensor([x])``, 0-d tensors are unsqueezed) and logs
``torch.mean`` under "Episode/<key>", then the direct scalars under "<k>/frame|iter|time", then the mean score.
Pinning: tests/test_oracle_pinned_observer.py (container only) runs the reference's own method source on a stub algo / writer.
"""
from __future__ import annotations

import torch


class ObserverOracle:
    def __init__(self, writer, device="cpu"):
        self.writer, self.device = writer, device
        self.ep_infos, self.direct_info = [], {}

    def process_infos(self, infos, done_indices=None):
        if "episode" in infos:
            self.ep_infos.append(infos["episode"])
        if len(infos) > 0:
            self.direct_info = {k: v for k, v in infos.items()
                                if isinstance(v, (float, int)) or (isinstance(v, torch.Tensor) and len(v.shape) == 0)}

    def after_print_stats(self, frame, epoch_num, total_time):
        if self.ep_infos:
            for key in self.ep_infos[0]:
                parts = []
                for ep_info in self.ep_infos:
                    v = ep_info[key]
                    if not isinstance(v, torch.Tensor):
                        v = torch.Tensor([v])
                    if len(v.shape) == 0:
                        v = v.unsqueeze(0)
                    parts.append(v.to(self.device))
                self.writer.add_scalar("Episode/" + key, torch.mean(torch.cat([torch.tensor([], device=self.device)] + parts)), epoch_num)
            self.ep_infos.clear()
        for k, v in self.direct_info.items():
            self.writer.add_scalar(f"{k}/frame", v, frame)
            self.writer.add_scalar(f"{k}/iter", v, epoch_num)
            self.writer.add_scalar(f"{k}/time", v, total_time)
