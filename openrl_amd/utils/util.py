"""Seeding and host<->device shims (``openrl/utils/util.py:13-38``)."""
import random

import numpy as np
import torch


def set_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def _t2n(x):
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return x


def check(x):
    return torch.from_numpy(x) if isinstance(x, np.ndarray) else x
