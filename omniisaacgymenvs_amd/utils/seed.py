"""``set_seed`` as the launch scripts use it (scripts/random_policy.py:96, rlgames_train.py:168-171; semantics in
docs/framework/reproducibility.md:7-31): -1 picks a random seed, the chosen seed goes to ``random``, ``numpy`` and
``torch`` and comes back to be stored in the config.  The omni helper itself is closed source: parity unpinned, this follows
the documented behaviour."""
from __future__ import annotations

import os
import random

import numpy as np
import torch


def set_seed(seed: int, torch_deterministic: bool = False) -> int:
    if seed == -1:
        seed = 42 if torch_deterministic else int(np.random.randint(0, 10000))
    print(f"Setting seed: {seed}")
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    if torch_deterministic:
        torch.use_deterministic_algorithms(True, warn_only=True)
    return seed
