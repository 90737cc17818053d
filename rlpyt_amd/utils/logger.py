"""A small tabular logger with the calls the runners use (subset of the rllab-style
rlpyt/utils/logging/logger.py API: log, record_tabular[_misc_stat], dump_tabular,
save_itr_params, snapshot dir/mode, tabular prefixes)."""
import csv
import os
import sys
import time
from contextlib import contextmanager

import numpy as np

_prefix = []
_tabular_prefix = []
_tabular = []
_snapshot_dir = None
_snapshot_mode = "none"
_snapshot_gap = 1
_csv_path = None
_csv_header = None
_quiet = False
_iteration = 0


def set_quiet(q=True):
    global _quiet
    _quiet = q


def log(s, with_timestamp=True):
    if _quiet:
        return
    out = "".join(_prefix) + str(s)
    if with_timestamp:
        out = time.strftime("%Y-%m-%d %H:%M:%S") + " | " + out
    print(out, file=sys.stdout, flush=True)


def set_iteration(itr):
    global _iteration
    _iteration = itr


def push_prefix(p):
    _prefix.append(p)


def pop_prefix():
    _prefix.pop()


@contextmanager
def prefix(p):
    push_prefix(p)
    try:
        yield
    finally:
        pop_prefix()


@contextmanager
def tabular_prefix(p):
    _tabular_prefix.append(p)
    try:
        yield
    finally:
        _tabular_prefix.pop()


def record_tabular(key, val, *args, **kwargs):
    _tabular.append(("".join(_tabular_prefix) + str(key), val))


def record_tabular_misc_stat(key, values, placement="back"):
    def name(s):
        return f"{key}{s}" if placement == "back" else f"{s}{key}"
    if values is not None and len(values) > 0:
        v = np.asarray(values, dtype=np.float64)
        stats = (np.average(v), np.std(v), np.median(v), np.min(v), np.max(v))
    else:
        stats = (np.nan,) * 5
    for s, x in zip(("Average", "Std", "Median", "Min", "Max"), stats):
        record_tabular(name(s), x)


def get_tabular():
    return dict(_tabular)


def dump_tabular(*args, **kwargs):
    global _csv_header
    row = dict(_tabular)
    if not _quiet:
        w = max((len(k) for k in row), default=0)
        for k, v in row.items():
            log(f"{k:<{w}}  {v}", with_timestamp=False)
    if _csv_path is not None:
        keys = list(row)
        if _csv_header != keys:  # new keys: rewrite the file with a merged header
            old = []
            if os.path.exists(_csv_path):
                with open(_csv_path) as f:
                    old = list(csv.DictReader(f))
            _csv_header = keys
            with open(_csv_path, "w", newline="") as f:
                wr = csv.DictWriter(f, fieldnames=keys, extrasaction="ignore")
                wr.writeheader()
                for r in old:
                    wr.writerow(r)
        with open(_csv_path, "a", newline="") as f:
            csv.DictWriter(f, fieldnames=_csv_header, extrasaction="ignore").writerow(row)
    del _tabular[:]
    return row


def set_snapshot_dir(d):
    global _snapshot_dir, _csv_path
    _snapshot_dir = d
    if d is not None:
        os.makedirs(d, exist_ok=True)
        _csv_path = os.path.join(d, "progress.csv")


def get_snapshot_dir():
    return _snapshot_dir


def set_snapshot_mode(mode):
    global _snapshot_mode
    _snapshot_mode = mode


def set_snapshot_gap(gap):
    global _snapshot_gap
    _snapshot_gap = gap


def save_itr_params(itr, params):
    """torch.save the snapshot per snapshot_mode in {none,last,all,gap,last+gap}
    (rlpyt/utils/logging/logger.py:332-353)."""
    if _snapshot_dir is None or _snapshot_mode == "none":
        return
    import torch
    if _snapshot_mode == "all":
        torch.save(params, os.path.join(_snapshot_dir, f"itr_{itr}.pkl"))
    elif _snapshot_mode == "last":
        torch.save(params, os.path.join(_snapshot_dir, "params.pkl"))
    elif _snapshot_mode in ("gap", "last+gap"):
        if itr == 0 or (itr + 1) % _snapshot_gap == 0:
            torch.save(params, os.path.join(_snapshot_dir, f"itr_{itr}.pkl"))
        if _snapshot_mode == "last+gap":
            torch.save(params, os.path.join(_snapshot_dir, "params.pkl"))
    else:
        raise NotImplementedError(_snapshot_mode)


@contextmanager
def logger_context(log_dir, run_ID, name, log_params=None, snapshot_mode="none"):
    """Minimal counterpart of rlpyt/utils/logging/context.py:24-83."""
    import json
    exp_dir = os.path.join(log_dir, f"run_{run_ID}")
    set_snapshot_mode(snapshot_mode)
    set_snapshot_dir(exp_dir)
    push_prefix(f"{name}_{run_ID} ")
    if log_params is not None:
        with open(os.path.join(exp_dir, "params.json"), "w") as f:
            json.dump(dict(name=name, run_ID=run_ID, **log_params), f, default=str)
    try:
        yield
    finally:
        pop_prefix()
        set_snapshot_dir(None)
