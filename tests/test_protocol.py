"""The drop-in boundary as data: ``tests/golden/protocol.json`` holds
``inspect.signature`` of every protocol member the reference's runner / sampler / algorithm /
agent / replay classes call on each other, recorded from the reference classes themselves
(``make_golden.py protocol``).  This repo's classes must accept a superset:

* every reference parameter exists here under the same name, positional parameters in the same
  order, same simple defaults -- so call sites written against the reference bind identically;
* parameters this repo adds all carry defaults (or are ``*args`` / ``**kwargs``);
* class attributes the runner reads (``bootstrap_value``, ``opt_info_fields``) are equal.

A second test (build container only: needs /root/reference) runs the REFERENCE's own
``MinibatchRl`` over this repo's sampler + agent + algorithm, unmodified.
"""
import importlib
import inspect
import json
import os
import sys

import pytest

from conftest import GOLDEN

sys.path.insert(0, GOLDEN)
from protocol_cases import PROTOCOL  # noqa: E402

with open(os.path.join(GOLDEN, "protocol.json")) as _f:
    RECORDED = json.load(_f)

POSITIONAL = ("POSITIONAL_ONLY", "POSITIONAL_OR_KEYWORD")


def _resolve(path):
    mod, _, attr = path.rpartition(".")
    return getattr(importlib.import_module(mod), attr)


def _check_signature(where, ref_params, fn):
    ours = inspect.signature(fn).parameters
    var_kw = any(p.kind.name == "VAR_KEYWORD" for p in ours.values())
    var_pos = any(p.kind.name == "VAR_POSITIONAL" for p in ours.values())
    ours_pos = [n for n, p in ours.items() if p.kind.name in POSITIONAL]
    ref_pos = [p["name"] for p in ref_params if p["kind"] in POSITIONAL]
    ref_names = {p["name"] for p in ref_params}
    # positional parameters the reference has: same order at the front (unless *args absorbs them)
    if not var_pos:
        shared = [n for n in ref_pos if n in ours]
        assert [n for n in ours_pos if n in ref_names] == shared, \
            f"{where}: positional order differs: ours {ours_pos} vs reference {ref_pos}"
    for p in ref_params:
        if p["kind"] in ("VAR_POSITIONAL", "VAR_KEYWORD"):
            continue
        name = p["name"]
        if name not in ours:
            assert var_kw or (var_pos and p["kind"] in POSITIONAL), \
                f"{where}: reference parameter {name!r} is not accepted"
            continue
        mine = ours[name]
        if p["has_default"]:
            assert mine.default is not inspect.Parameter.empty, \
                f"{where}: {name!r} is optional in the reference, required here"
            d = p["default"]
            if isinstance(d, (int, float, bool)) or d is None:
                if isinstance(mine.default, (int, float, bool)) or mine.default is None:
                    assert mine.default == d, \
                        f"{where}: default of {name!r} is {mine.default!r}, reference {d!r}"
    for name, mine in ours.items():            # what we add must be optional
        if name in ref_names or mine.kind.name in ("VAR_POSITIONAL", "VAR_KEYWORD"):
            continue
        assert mine.default is not inspect.Parameter.empty, \
            f"{where}: extra parameter {name!r} has no default (reference call sites would break)"


@pytest.mark.parametrize("ref_path", sorted(PROTOCOL))
def test_accepts_reference_call_signatures(ref_path):
    ours_path, members = PROTOCOL[ref_path]
    rec = RECORDED[ref_path]
    obj = _resolve(ours_path)
    if not inspect.isclass(obj):
        _check_signature(ours_path, rec["__call__"], obj)
        return
    assert set(rec) == set(members), "protocol.json is stale: rerun make_golden.py protocol"
    for m in members:
        assert hasattr(obj, m), f"{ours_path} lacks {m}"
        r = rec[m]
        if r == "property":
            continue
        if isinstance(r, dict):                 # class attribute read by the runner
            v = getattr(obj, m)
            assert (list(v) if isinstance(v, (tuple, list)) else v) == r["attr"], (ours_path, m)
            continue
        _check_signature(f"{ours_path}.{m}", r, getattr(obj, m))


REF = os.environ.get("RLPYT_REFERENCE", "/root/reference")


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "rlpyt")),
                    reason="needs the reference checkout (build container only)")
def test_reference_runner_drives_these_classes(tmp_path):
    """The reference's OWN ``MinibatchRl`` (rlpyt/runners/minibatch_rl.py:74-96,253-262), imported
    from /root/reference and unmodified, trains with this repo's sampler, agent and algorithm: the
    host-logic plumbing case of BASELINE config #1 (tiny discrete env, CPU tensors; the HIP touch
    points of the algorithm are stubbed by the test-only oracle subclass, there is no GPU here).
    The sampler takes its worker count from ``affinity["workers_cpus"]`` as the reference's
    samplers do.  Runs in a subprocess so the reference's modules never mix with this suite's."""
    import subprocess
    import textwrap
    code = textwrap.dedent(f"""
        import sys, types
        sys.path.insert(0, {REF!r}); sys.path.insert(0, {os.path.dirname(GOLDEN)!r})
        sys.path.insert(0, {os.path.dirname(os.path.dirname(GOLDEN))!r})
        pp = types.ModuleType("pyprind")
        class ProgBar:
            def __init__(self, n, **k): self.active = True
            def update(self, *a, **k): pass
            def stop(self): self.active = False
        pp.ProgBar = ProgBar; sys.modules["pyprind"] = pp
        from rlpyt.runners.minibatch_rl import MinibatchRl          # the REFERENCE runner
        from rlpyt.utils.logging import logger
        from rlpyt_amd.samplers.gpu import GpuSampler
        from rlpyt_amd.agents.pg.atari import MlpCategoricalPgAgent
        from rlpyt_amd.envs.synthetic import TinyDiscreteEnv
        from test_host_logic import OraclePPO
        sampler = GpuSampler(EnvCls=TinyDiscreteEnv, env_kwargs=dict(), batch_T=8, batch_B=4,
                             max_decorrelation_steps=0)
        algo = OraclePPO(minibatches=2, epochs=1, linear_lr_schedule=False)
        agent = MlpCategoricalPgAgent()
        runner = MinibatchRl(algo=algo, agent=agent, sampler=sampler, n_steps=8 * 4 * 6,
                             log_interval_steps=8 * 4 * 3, seed=0,
                             affinity=dict(cuda_idx=None, workers_cpus=[0, 1], set_affinity=False))
        rows = []
        orig = logger.dump_tabular
        def capture(*a, **k):
            rows.append(dict(logger._tabular)); return orig(*a, **k)
        logger.dump_tabular = capture
        runner.train()
        assert sampler.n_workers == 2, sampler.n_workers
        assert algo.update_counter == 6 * 2, algo.update_counter
        last = {{k.split("/")[-1]: v for k, v in rows[-1].items()}}
        assert len(rows) == 2 and float(last["StepsPerSecond"]) > 0, rows
        assert int(last["CumSteps"]) == 8 * 4 * 6 and int(last["CumUpdates"]) == 12, last
        print("REFERENCE_RUNNER_OK", last["CumSteps"], last["CumUpdates"])
    """)
    env = dict(os.environ, PYTHONPATH="")
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                         cwd=str(tmp_path), env=env)
    assert res.returncode == 0 and "REFERENCE_RUNNER_OK" in res.stdout, res.stdout[-3000:] + res.stderr[-3000:]
