"""Index / allocation helpers (names follow rlpyt/utils/misc.py)."""
import numpy as np
import torch


def iterate_mb_idxs(data_length, minibatch_size, shuffle=False):
    """Contiguous chunks of (optionally shuffled) indexes, remainder dropped
    (misc.py:6-17).  Shuffling uses ``np.random.shuffle`` like the reference so a seeded
    run visits the same minibatches."""
    idxs = None
    if shuffle:
        idxs = np.arange(data_length)
        np.random.shuffle(idxs)
    for s in range(0, data_length - minibatch_size + 1, minibatch_size):
        yield idxs[s:s + minibatch_size] if shuffle else slice(s, s + minibatch_size)


def zeros(shape, dtype):
    try:
        return torch.zeros(shape, dtype=dtype)
    except TypeError:
        return np.zeros(shape, dtype=dtype)


def empty(shape, dtype):
    try:
        return torch.empty(shape, dtype=dtype)
    except TypeError:
        return np.empty(shape, dtype=dtype)
