"""Seeding: one call seeds numpy, torch's host generator and (when a device is present) the
device generators; samplers derive per-env seeds as ``seed + global env index`` and per-rank
seeds as ``seed + 100 * rank`` (rlpyt/runners/sync_rl.py:82)."""
import os
import time

import numpy as np
import torch

_SEED_MOD = 2 ** 32 - 2      # numpy accepts [0, 2**32)


def make_seed():
    """A fresh seed for runs that did not ask for one: OS entropy mixed with the clock, folded
    into four decimal digits plus a leading digit like the reference's seeds, so it stays easy
    to read in logs and leaves headroom for the ``+ 100 * rank`` / ``+ env index`` offsets."""
    raw = int.from_bytes(os.urandom(4), "little") ^ (time.time_ns() & 0xffffffff)
    return raw % 90000 + 10000


def set_seed(seed):
    seed = int(seed) % _SEED_MOD
    np.random.seed(seed)
    torch.manual_seed(seed)        # also seeds every visible device generator
    return seed


def set_envs_seeds(envs, seed):
    """``env.seed(seed + i)`` for envs that can be seeded (seed None: leave them alone)."""
    if seed is None:
        return
    for i, env in enumerate(envs):
        seeder = getattr(env, "seed", None)
        if callable(seeder):
            seeder(seed + i)
