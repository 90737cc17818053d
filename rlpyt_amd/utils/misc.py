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


def usable_cpus():
    """CPUs this process can actually run on at once: the smallest of the hardware thread
    count, the scheduler affinity mask and the cgroup CPU quota (``cpu.max`` of cgroup v2 or
    ``cpu.cfs_quota_us`` of v1).  The bench boxes show 256 threads under a 16-CPU quota; sizing
    the env-worker pool from ``os.cpu_count()`` alone oversubscribes the quota and gets the
    whole process tree throttled."""
    import os
    n = float(os.cpu_count() or 1)
    try:
        n = min(n, float(len(os.sched_getaffinity(0))))
    except (AttributeError, OSError):
        pass
    for path, parse in (
            ("/sys/fs/cgroup/cpu.max",
             lambda s: None if s.split()[0] == "max" else float(s.split()[0]) / float(s.split()[1])),
            ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda s: float(s)),):
        try:
            with open(path) as f:
                q = parse(f.read().strip())
            if path.endswith("cfs_quota_us"):
                if q is None or q <= 0:
                    continue
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    q = q / float(f.read().strip())
            if q:
                n = min(n, q)
        except (OSError, ValueError, IndexError, ZeroDivisionError):
            continue
    return max(n, 1.)
