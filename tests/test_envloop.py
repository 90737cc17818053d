"""rlpyt_amd._envloop (csrc/envloop.c): the sampler workers' per-env loop body in C against the Python
loop body of ``EnvRunner.step_all`` -- the worker side of rlpyt/samplers/parallel/gpu/collectors.py:18-50
plus the stock trajectory statistics (rlpyt/samplers/collections.py:30-56).  Same step-buffer bytes,
same env_info rows, same completed-trajectory records, field for field and TYPE for type (the types are
what numpy's promotion rules make of the reference's per-step updates)."""
import numpy as np
import pytest

from rlpyt_amd.envs import EnvStep
from rlpyt_amd.envs.synthetic import SyntheticPong, TinyDiscreteEnv
from rlpyt_amd.samplers.collections import AtariTrajInfo, StepBuffer, StepBufferFs, TrajInfo
from rlpyt_amd.samplers.workers import EnvRunner
from rlpyt_amd.utils.collections import namedarraytuple

EnvInfo = namedarraytuple("EnvInfo", ["game_score", "traj_done"])


class Float64RewardPong(SyntheticPong):
    """Rewards as np.float64 (what rlpyt's own AtariEnv hands out: np.sign(game_score))."""

    def step(self, action):
        o, r, d, info = super().step(action)
        return EnvStep(o, np.float64(r), d, info)


def _runner(EnvCls, n, T, native, TrajInfoCls, lazy, frames=True, mbr=True, **env_kw):
    envs = [EnvCls(seed=100 + i, **env_kw) for i in range(n)]
    o = envs[0].reset()
    fields = dict(observation=np.zeros((n,) + o.shape, o.dtype), action=np.zeros(n, np.int64),
                  reward=np.zeros(n, np.float32), done=np.zeros(n, bool))
    if frames:
        step = StepBufferFs(frame=np.zeros((n,) + o.shape[1:], o.dtype), reset=np.zeros(n, bool), **fields)
    else:
        step = StepBuffer(**fields)
    # env_info as the sampler hands it to a worker: COLUMN slices of the [T, B] arrays
    big = EnvInfo(game_score=np.zeros((T, n + 3), np.float32), traj_done=np.zeros((T, n + 3), bool))
    info = EnvInfo(game_score=big.game_score[:, 2:2 + n], traj_done=big.traj_done[:, 2:2 + n]) \
        if EnvCls is not TinyDiscreteEnv else None
    r = EnvRunner(envs, step, info, TrajInfoCls, mbr)
    r.use_native = native
    if lazy:
        class L:
            value = True
        r.lazy_obs, r.batch_T = L(), T
    np.random.seed(5)
    r.start(7)
    return r, step, big


@pytest.mark.parametrize("mbr", [True, False], ids=["reset", "wait_reset"])
@pytest.mark.parametrize("EnvCls,TI,env_kw,frames,lazy", [
    (SyntheticPong, AtariTrajInfo, dict(points_to_end=1, max_steps=40), True, True),
    (SyntheticPong, TrajInfo, dict(points_to_end=2, max_steps=25), True, False),
    (Float64RewardPong, AtariTrajInfo, dict(points_to_end=1, max_steps=30), True, True),
    (TinyDiscreteEnv, TrajInfo, dict(horizon=9), False, False),
])
def test_native_loop_body_equals_python_loop_body(EnvCls, TI, env_kw, frames, lazy, mbr):
    """Reset collector and wait-reset collector (a finished env idles with a blank observation row
    until ``begin_batch`` resets it, collectors.py:73-126)."""
    n, T, n_batches = 5, 16, 6 if mbr else 14
    TI._discount = 0.97
    try:
        runs = []
        for native in (True, False):
            r, step, big = _runner(EnvCls, n, T, native, TI, lazy, frames, mbr, **env_kw)
            assert (r._native is not None) == native
            rng = np.random.RandomState(3)
            rows, completed = [], []
            for _ in range(n_batches):
                r.begin_batch()
                rows.append([np.array(x).copy() for x in step])
                for t in range(T):
                    step.action[:] = rng.randint(0, 2 if EnvCls is TinyDiscreteEnv else 6, n)
                    r.step_all(t, completed)
                    rows.append([np.array(x).copy() for x in step])
                rows.append([big.game_score.copy(), big.traj_done.copy()])
            runs.append((rows, completed))
        (rows_n, comp_n), (rows_p, comp_p) = runs
        for a, b in zip(rows_n, rows_p):
            for x, y in zip(a, b):
                assert x.dtype == y.dtype and np.array_equal(x, y)
        assert len(comp_n) == len(comp_p) > 5
        for a, b in zip(comp_n, comp_p):
            assert list(a.keys()) == list(b.keys())
            for k in a:
                assert type(a[k]) is type(b[k]), (k, type(a[k]), type(b[k]))
                assert a[k] == b[k], (k, a[k], b[k])
    finally:
        TI._discount = 1


def test_native_loop_declines_what_it_does_not_cover():
    class MyTrajInfo(TrajInfo):        # a user's own statistics: their step() must be called
        pass
    r, _, _ = _runner(SyntheticPong, 3, 4, True, MyTrajInfo, False)
    assert r._native is None
    envs = [SyntheticPong(seed=i) for i in range(2)]
    o = envs[0].reset()
    step = StepBuffer(observation=np.zeros((2,) + o.shape, o.dtype), action=np.zeros(2, np.int64),
                      reward=np.zeros(2, np.float32), done=np.zeros(2, bool))
    r = EnvRunner(envs, step, None, TrajInfo, False)     # wait-reset collector: covered since round 5
    r.start(0)
    assert r._native is not None


def test_native_loop_declines_env_info_dtypes_the_c_body_cannot_store():
    """an env_info field of a dtype csrc/envloop.c does not store (int16 here) must
    leave the worker on the Python loop body instead of raising in ``start()``."""
    n, T = 3, 4
    envs = [SyntheticPong(seed=i) for i in range(n)]
    o = envs[0].reset()
    step = StepBufferFs(observation=np.zeros((n,) + o.shape, o.dtype), action=np.zeros(n, np.int64),
                        reward=np.zeros(n, np.float32), done=np.zeros(n, bool),
                        frame=np.zeros((n,) + o.shape[1:], o.dtype), reset=np.zeros(n, bool))
    info = EnvInfo(game_score=np.zeros((T, n), np.int16), traj_done=np.zeros((T, n), bool))
    r = EnvRunner(envs, step, info, AtariTrajInfo, True)
    r.start(0)
    assert r._native is None
    completed = []
    step.action[:] = 2
    r.step_all(0, completed)            # the Python body runs
    assert step.frame.any()


class _InfoA(tuple):
    traj_done = False


def test_native_loop_probes_traj_done_per_info_type():
    """``traj_done`` / ``game_score`` are looked up per info TYPE, not once for the
    whole loop -- an env whose first infos lack ``traj_done`` must still end its trajectory when a
    later info (another type) carries it."""
    from collections import namedtuple
    WithTd = namedtuple("WithTd", ["game_score", "traj_done"])
    Plain = namedtuple("Plain", ["game_score"])

    class TwoInfoEnv(SyntheticPong):
        k = 0

        def step(self, action):
            o, r, d, info = super().step(action)
            self.k += 1
            if self.k < 3:
                return EnvStep(o, r, False, Plain(0.0))
            return EnvStep(o, r, False, WithTd(0.0, self.k == 5))

    n, T = 2, 8
    outs = []
    for native in (True, False):
        envs = [TwoInfoEnv(seed=i) for i in range(n)]
        o = envs[0].reset()
        step = StepBufferFs(observation=np.zeros((n,) + o.shape, o.dtype),
                            action=np.zeros(n, np.int64), reward=np.zeros(n, np.float32),
                            done=np.zeros(n, bool), frame=np.zeros((n,) + o.shape[1:], o.dtype),
                            reset=np.zeros(n, bool))
        r = EnvRunner(envs, step, None, TrajInfo, True)
        r.use_native = native
        r.start(0)
        assert (r._native is not None) == native
        completed = []
        for t in range(T):
            r.step_all(t, completed)
        outs.append([dict(c) for c in completed])
    assert len(outs[0]) == len(outs[1]) == n          # one finished trajectory per env (k == 5)
    assert outs[0] == outs[1]
