"""Seeding helpers (rlpyt/utils/seed.py)."""
import time

import numpy as np
import torch


def make_seed():
    """A seed from the clock's microseconds, as the reference does."""
    d = 10000
    t = time.time()
    sub1 = int(t * d) % d
    sub2 = int(t * d ** 2) % d
    s = 1e-3
    s_inv = 1. / s
    time.sleep(s * sub2 / d)
    t2 = time.time()
    t2 = t2 - int(t2)
    t2 = int(t2 * d * s_inv) % d
    time.sleep(s * sub1 / d)
    t3 = time.time()
    t3 = t3 - int(t3)
    t3 = int(t3 * d * s_inv * 10) % 10
    return (t3 - 1) * d + t2


def set_seed(seed):
    seed %= 4294967294
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)


def set_envs_seeds(envs, seed):
    if seed is not None:
        for i, env in enumerate(envs):
            env.seed(seed + i)
