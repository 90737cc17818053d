"""The drop-in boundary shown on the MI355X: the REFERENCE's own
runners, imported unmodified from ``oracle/_ref`` (the copy ``oracle/make_ref.py`` ships to the GPU
box; tests may import it), train over the PRODUCT classes -- ``GpuSampler`` + ``PPO`` +
``AtariFfAgent`` on ``cuda:0``:

* ``rlpyt.runners.minibatch_rl.MinibatchRl`` (rlpyt/runners/minibatch_rl.py:74-96,253-262):
  3 iterations at [32, 64] (two minibatches of M = 1024, the size from which the update runs the
  bf16x6 trunk GEMMs), the tabular keys of ``tests/golden/runner_keys.json`` (recorded from the
  reference's runner over the reference's classes), finite diagnostics, and -- from the launch
  counters of the C-ABI library -- that the product's rollout and update kernels did the work;
* ``rlpyt.runners.sync_rl.SyncRl`` (rlpyt/runners/sync_rl.py:60-101): the reference's FORK launch
  -- the master forks ``world_size - 1`` runner replicas holding the same sampler / algo / agent
  objects, every replica joins the process group and wraps the agent in DistributedDataParallel.
  With >= 2 devices: rank r on cuda:r over RCCL, as the reference places them.  On a 1-GPU box the
  same unmodified runner code runs with both ranks on cuda:0 and the process-group BACKEND string
  swapped to gloo inside ``torch.distributed.init_process_group`` (RCCL refuses two ranks on one
  device); parameters must be identical across the ranks at the end.

Each case runs in its own python process so that the reference's modules never mix with this
suite's and no HIP context exists in the process that forks.
"""
import json
import os
import subprocess
import sys
import textwrap

import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HAVE_REF = os.path.isfile(os.path.join(ROOT, "oracle", "_ref", "rlpyt", "__init__.py"))
needs_ref = pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref absent (python oracle/make_ref.py "
                                                    "where /root/reference exists)")

PRELUDE = f"""
import json, os, sys
sys.path.insert(0, {ROOT!r})
from oracle import ref_runner
assert ref_runner.load(), "oracle/_ref did not load"
import rlpyt, torch
assert os.path.realpath(rlpyt.__file__).startswith(os.path.realpath({os.path.join(ROOT, "oracle", "_ref")!r}))
from rlpyt.utils.logging import logger
from rlpyt_amd import _lib
from rlpyt_amd.agents.pg.atari import AtariFfAgent
from rlpyt_amd.algos.pg.ppo import PPO
from rlpyt_amd.envs.synthetic import SyntheticPong
from rlpyt_amd.samplers.gpu import GpuSampler
T, B = 32, 64
rows = []
_orig_dump = logger.dump_tabular
def _capture(*a, **k):
    rows.append(dict(logger._tabular)); return _orig_dump(*a, **k)
logger.dump_tabular = _capture
def make():
    sampler = GpuSampler(EnvCls=SyntheticPong, env_kwargs=dict(points_to_end=1, max_steps=24),
                         batch_T=T, batch_B=B, max_decorrelation_steps=10)
    algo = PPO(discount=0.99, learning_rate=1e-3, gae_lambda=0.98, minibatches=2, epochs=2,
               ratio_clip=0.1, linear_lr_schedule=True, normalize_advantage=False)
    return sampler, algo, AtariFfAgent()
"""


def _run(code, tmp_path, timeout=900):
    env = dict(os.environ, PYTHONPATH="", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    res = subprocess.run([sys.executable, "-c", textwrap.dedent(PRELUDE) + textwrap.dedent(code)],
                         capture_output=True, text=True, timeout=timeout, cwd=str(tmp_path), env=env)
    assert res.returncode == 0, res.stdout[-4000:] + "\n" + res.stderr[-4000:]
    return res.stdout


@needs_ref
def test_reference_minibatch_rl_trains_product_classes_on_the_gpu(tmp_path):
    out = _run("""
        from rlpyt.runners.minibatch_rl import MinibatchRl          # the REFERENCE runner
        sampler, algo, agent = make()
        runner = MinibatchRl(algo=algo, agent=agent, sampler=sampler, n_steps=T * B * 3,
                             log_interval_steps=T * B, seed=0,
                             affinity=dict(cuda_idx=0, workers_cpus=[0, 1, 2, 3], set_affinity=False))
        _lib.variant_reset()
        runner.train()
        torch.cuda.synchronize()
        assert next(agent.parameters()).is_cuda and agent.device.type == "cuda"
        assert sampler.n_workers == 4, sampler.n_workers
        assert algo.update_counter == 3 * 4, algo.update_counter
        assert all(bool(torch.isfinite(p).all()) for p in agent.parameters())
        ran = sorted(k for k, v in _lib.variant_counts().items() if v > 0)
        print("DROPIN_RESULT " + json.dumps(dict(rows=rows, ran=ran)))
    """, tmp_path)
    line = [ln for ln in out.splitlines() if ln.startswith("DROPIN_RESULT ")][-1]
    res = json.loads(line[len("DROPIN_RESULT "):])
    rows, ran = res["rows"], res["ran"]
    assert len(rows) == 3
    with open(os.path.join(GOLDEN, "runner_keys.json")) as f:
        keys = json.load(f)["train"]
    last = rows[-1]
    assert list(last) == keys, (sorted(set(keys) - set(last)), sorted(set(last) - set(keys)))
    flat = {k.split("/")[-1]: v for k, v in last.items()}
    assert int(flat["CumSteps"]) == 32 * 64 * 3 and int(flat["CumUpdates"]) == 12
    assert float(flat["StepsPerSecond"]) > 0 and int(flat["CumCompletedTrajs"]) > 0
    for k in ("lossAverage", "gradNormAverage", "entropyAverage", "perplexityAverage",
              "ReturnAverage", "LengthAverage"):
        v = float(flat[k])
        assert v == v and abs(v) != float("inf"), (k, flat[k])
    assert 0. < float(flat["entropyAverage"]) <= 1.7918 and 1. <= float(flat["perplexityAverage"]) <= 6.0001
    # the product kernels did the work: rollout step graph, GAE scan, the whole update
    for name in ("sample_convs_kernel", "rollout_fc_kernel", "rollout_head_kernel", "scan_exact_kernel",
                 "convs_fwd_fused_kernel", "gemm_nt_x6_kernel", "gemm_tn_x6_kernel",
                 "ppo_head_loss_kernel", "conv2_bwd_x6_kernel", "conv1_wgrad_kernel",
                 "clip_adam_norm_kernel", "clip_adam_apply_kernel"):
        assert any(name in k for k in ran), (name, ran)


SYNC_BODY = """
    from rlpyt.runners.sync_rl import SyncRl                        # the REFERENCE runner
    import torch.distributed as dist
    SAME_GPU = {same_gpu}
    if SAME_GPU:
        # RCCL refuses two ranks on one device: keep the reference's launch code as it is and swap
        # only the backend string where it reaches torch.distributed (forked replicas inherit this)
        _init = dist.init_process_group
        def _init_gloo(backend=None, **kw):
            return _init(backend="gloo", **kw)
        dist.init_process_group = _init_gloo
    sampler, algo, agent = make()
    affinities = [dict(cuda_idx=0 if SAME_GPU else r, workers_cpus=[2 * r, 2 * r + 1],
                       set_affinity=False) for r in range(2)]
    runner = SyncRl(algo=algo, agent=agent, sampler=sampler, n_steps=T * B * 2 * 2,
                    log_interval_steps=T * B * 2, seed=3, affinity=affinities)
    # every rank leaves a checksum of its parameters behind when its sampler shuts down
    import hashlib
    _shutdown = GpuSampler.shutdown
    def _shutdown_and_report(self):
        flat = torch.cat([p.detach().reshape(-1) for p in agent.parameters()]).cpu().numpy()
        r = dist.get_rank()
        with open(f"rank{{r}}.json", "w") as f:
            json.dump(dict(rank=r, world=dist.get_world_size(), device=str(agent.device),
                           ddp=type(agent.model).__name__, sha=hashlib.sha1(flat.tobytes()).hexdigest(),
                           finite=bool((flat == flat).all()), updates=algo.update_counter,
                           seed_workers=sampler.n_workers,
                           kernels=sorted(k for k, v in _lib.variant_counts().items() if v > 0)), f)
        return _shutdown(self)
    GpuSampler.shutdown = _shutdown_and_report
    runner.train()
    for w in runner.workers:
        w.join(120)
        assert w.exitcode == 0, w.exitcode
    print("SYNC_ROWS " + json.dumps(rows))
"""


def _check_sync(out, tmp_path, same_gpu):
    line = [ln for ln in out.splitlines() if ln.startswith("SYNC_ROWS ")][-1]
    rows = json.loads(line[len("SYNC_ROWS "):])
    assert len(rows) == 2
    flat = {k.split("/")[-1]: v for k, v in rows[-1].items()}
    # weak scaling: the runner counts world_size x the sampler batch per iteration (sync_rl.py:39-45)
    assert int(flat["CumSteps"]) == 32 * 64 * 2 * 2 and int(flat["CumUpdates"]) == 2 * 4
    ranks = []
    for r in range(2):
        with open(os.path.join(str(tmp_path), f"rank{r}.json")) as f:
            ranks.append(json.load(f))
    assert [x["world"] for x in ranks] == [2, 2]
    assert all(x["ddp"] == "DistributedDataParallel" and x["finite"] for x in ranks), ranks
    assert ranks[0]["sha"] == ranks[1]["sha"], "ranks diverged under the reference's SyncRl launch"
    assert [x["updates"] for x in ranks] == [8, 8]
    assert [x["device"] for x in ranks] == (["cuda:0", "cuda:0"] if same_gpu else ["cuda:0", "cuda:1"])
    for x in ranks:
        for name in ("sample_convs_kernel", "gemm_nt_x6_kernel", "gemm_tn_x6_kernel",
                     "ppo_head_loss_kernel", "conv2_bwd_x6_kernel", "clip_adam_apply_kernel"):
            assert any(name in k for k in x["kernels"]), (x["rank"], name)


@needs_ref
def test_reference_sync_rl_fork_launch_two_ranks_on_one_gpu(tmp_path):
    out = _run(SYNC_BODY.format(same_gpu=True), tmp_path)
    _check_sync(out, tmp_path, same_gpu=True)


@needs_ref
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one device per rank")
def test_reference_sync_rl_fork_launch_over_rccl(tmp_path):
    out = _run(SYNC_BODY.format(same_gpu=False), tmp_path)
    _check_sync(out, tmp_path, same_gpu=False)
