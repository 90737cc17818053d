"""The UNMODIFIED reference timed on host cores -- TEST / BASELINE INFRASTRUCTURE ONLY.

``bench.py``'s ``cpu_baseline`` leg (SURVEY 8(d) "CPU baseline, same run"): the reference's own
``SerialSampler`` (``rlpyt/samplers/serial/sampler.py:10``, default ``CpuResetCollector``) + its own
algorithm + agent, driven by the statement sequence of ``MinibatchRl.train``
(``rlpyt/runners/minibatch_rl.py:253-262``: ``sample_mode`` -> ``obtain_samples`` -> ``train_mode``
-> ``optimize_agent``; logging left out), on this repo's synthetic Atari-shaped env (the reference's
``AtariEnv`` needs ``atari_py`` + ``cv2``, absent from the image).  The reference package is the copy
``oracle/make_ref.py`` puts under ``oracle/_ref`` (git-ignored, ships to the GPU box).

Nothing under ``rlpyt_amd/`` imports this module; it is never the thing measured as ``value``.
"""
import os
import sys
import time
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "_ref")


def _install_pyprind_shim():
    """``rlpyt.utils.prog_bar`` imports ``pyprind`` (absent here); only ``ProgBar`` is used."""
    if "pyprind" in sys.modules:
        return
    pp = types.ModuleType("pyprind")

    class ProgBar:
        def __init__(self, n, **k):
            self.active = True

        def update(self, *a, **k):
            pass

        def stop(self):
            self.active = False
    pp.ProgBar = ProgBar
    sys.modules["pyprind"] = pp


def available():
    return os.path.isfile(os.path.join(REF_DIR, "rlpyt", "__init__.py"))


def load():
    """Make ``import rlpyt`` resolve to ``oracle/_ref/rlpyt``; False when the copy is absent."""
    if not available():
        return False
    if REF_DIR not in sys.path:
        sys.path.insert(0, REF_DIR)
    _install_pyprind_shim()
    import rlpyt  # noqa: F401
    return os.path.realpath(os.path.dirname(rlpyt.__file__)).startswith(os.path.realpath(REF_DIR))


def _quiet():
    from rlpyt.utils.logging import logger
    try:
        logger.set_log_tabular_only(True)      # silences logger.log() prints
    except Exception:  # noqa: BLE001
        pass


def _loop(sampler, agent, algo, first_itr, n_itr):
    """minibatch_rl.py:253-262 without the logging."""
    infos = None
    for itr in range(first_itr, first_itr + n_itr):
        agent.sample_mode(itr)
        samples, _traj = sampler.obtain_samples(itr)
        agent.train_mode(itr)
        infos = algo.optimize_agent(itr, samples)
    return infos


def time_ppo(EnvCls, env_kwargs, T=128, B=256, iters=1, threads=None, seed=0):
    """env-steps/s of the reference PPO iteration at [T, B] (BASELINE config #2's hyper-parameters,
    the ones bench.py gives the GPU run): ``iters`` timed iterations, none untimed (one iteration
    at [128, 256] is ~20-80 s of CPU; allocation and the first-touch of the samples buffer are
    part of what the reference pays too, and are < 1 % of that)."""
    import torch
    assert load(), "oracle/_ref is absent (python oracle/make_ref.py in the build container)"
    from rlpyt.agents.pg.atari import AtariFfAgent
    from rlpyt.algos.pg.ppo import PPO
    from rlpyt.samplers.serial.sampler import SerialSampler
    _quiet()
    if threads:
        torch.set_num_threads(int(threads))       # affinity["master_torch_threads"] in the runner
    sampler = SerialSampler(EnvCls=EnvCls, env_kwargs=dict(env_kwargs), batch_T=T, batch_B=B,
                            max_decorrelation_steps=0)
    agent = AtariFfAgent()
    algo = PPO(discount=0.99, learning_rate=1e-3, value_loss_coeff=1., entropy_loss_coeff=0.01,
               clip_grad_norm=1., gae_lambda=0.98, minibatches=4, epochs=4, ratio_clip=0.1,
               linear_lr_schedule=True, normalize_advantage=False)
    examples = sampler.initialize(agent, seed=seed, bootstrap_value=True)
    algo.initialize(agent=agent, n_itr=max(iters, 1), batch_spec=sampler.batch_spec,
                    mid_batch_reset=True, examples=examples)
    t0 = time.perf_counter()
    info = _loop(sampler, agent, algo, 0, iters)
    dt = time.perf_counter() - t0
    return dict(value=T * B * iters / dt, seconds=dt, T=T, B=B, iters=iters,
                cores=torch.get_num_threads(), last_loss=float(info.loss[-1]),
                classes="rlpyt.samplers.serial.sampler.SerialSampler + rlpyt.algos.pg.ppo.PPO + "
                        "rlpyt.agents.pg.atari.AtariFfAgent")


def time_dqn(EnvCls, env_kwargs, iters=20, fill_iters=8, threads=None, seed=0,
             replay_size=int(1e5)):
    """The reference DQN iteration of BASELINE config #3 (sampler [2, 16], batch 128, prioritized
    frame replay, replay_ratio 8 -> 2 updates per iteration) on host cores.  ``replay_size`` is
    REDUCED from the config's 1e6 frames (an 8.3 GB host ring for a 20-iteration sample; the ring
    size only sets the sum tree's depth, 18 instead of 21 levels -- the time is in the model)."""
    import torch
    assert load(), "oracle/_ref is absent"
    from rlpyt.agents.dqn.atari.atari_dqn_agent import AtariDqnAgent
    from rlpyt.algos.dqn.dqn import DQN
    from rlpyt.samplers.serial.sampler import SerialSampler
    _quiet()
    if threads:
        torch.set_num_threads(int(threads))
    T, B = 2, 16
    sampler = SerialSampler(EnvCls=EnvCls, env_kwargs=dict(env_kwargs), batch_T=T, batch_B=B,
                            max_decorrelation_steps=0)
    agent = AtariDqnAgent()
    algo = DQN(discount=0.99, batch_size=128, learning_rate=1e-4, clip_grad_norm=10.,
               min_steps_learn=fill_iters * T * B, double_dqn=False, prioritized_replay=True,
               n_step_return=1, replay_size=int(replay_size))
    examples = sampler.initialize(agent, seed=seed, bootstrap_value=False)
    n_itr = fill_iters + 2 + iters
    algo.initialize(agent=agent, n_itr=n_itr, batch_spec=sampler.batch_spec,
                    mid_batch_reset=True, examples=examples)
    _loop(sampler, agent, algo, 0, fill_iters + 2)       # fill + 2 learning iterations untimed
    u0 = algo.update_counter
    t0 = time.perf_counter()
    info = _loop(sampler, agent, algo, fill_iters + 2, iters)
    dt = time.perf_counter() - t0
    return dict(value=T * B * iters / dt, updates_per_s=(algo.update_counter - u0) / dt,
                seconds=dt, T=T, B=B, iters=iters, cores=torch.get_num_threads(),
                replay_frames=int(replay_size), last_loss=float(info.loss[-1]),
                classes="SerialSampler + rlpyt.algos.dqn.dqn.DQN + AtariDqnAgent + "
                        "PrioritizedReplayFrameBuffer")


def time_r2d1(EnvCls, env_kwargs, iters=1, threads=None, seed=0, replay_size=int(2e5)):
    """The reference R2D1 iteration of BASELINE config #5 (sampler [40, 192], sequences
    [40 + 80 + 5, 64], prioritized sequence frame replay) on host cores: sampling-only iterations
    until sequences can be drawn, then ``iters`` timed iterations (each: 7 680 env steps + 1 update
    over 64 sequences of 125 steps).  ``replay_size`` reduced from 4e6 frames (33 GB host ring)."""
    import torch
    assert load(), "oracle/_ref is absent"
    from rlpyt.agents.dqn.atari.atari_r2d1_agent import AtariR2d1Agent
    from rlpyt.algos.dqn.r2d1 import R2D1
    from rlpyt.samplers.parallel.cpu.collectors import CpuWaitResetCollector
    from rlpyt.samplers.serial.sampler import SerialSampler
    _quiet()
    if threads:
        torch.set_num_threads(int(threads))
    T, B = 40, 192
    sampler = SerialSampler(EnvCls=EnvCls, env_kwargs=dict(env_kwargs), batch_T=T, batch_B=B,
                            max_decorrelation_steps=0, CollectorCls=CpuWaitResetCollector)
    agent = AtariR2d1Agent(eps_final=0.1, eps_final_min=0.0005)
    fill_iters = 5          # 200 ring rows: 125-step sequences + n-step + the forbidden zones fit
    algo = R2D1(discount=0.997, batch_T=80, batch_B=64, warmup_T=40, store_rnn_state_interval=40,
                replay_ratio=1, learning_rate=1e-4, clip_grad_norm=80.,
                min_steps_learn=fill_iters * T * B, double_dqn=True, prioritized_replay=True,
                n_step_return=5, pri_alpha=0.9, pri_beta_init=0.6, pri_beta_final=0.6,
                input_priority_shift=2, replay_size=int(replay_size))
    examples = sampler.initialize(agent, seed=seed, bootstrap_value=False)
    algo.initialize(agent=agent, n_itr=fill_iters + iters, batch_spec=sampler.batch_spec,
                    mid_batch_reset=False, examples=examples)
    _loop(sampler, agent, algo, 0, fill_iters)           # sampling only (untimed)
    u0 = algo.update_counter
    t0 = time.perf_counter()
    info = _loop(sampler, agent, algo, fill_iters, iters)
    dt = time.perf_counter() - t0
    return dict(value=T * B * iters / dt, updates_per_s=(algo.update_counter - u0) / dt,
                updates=int(algo.update_counter - u0), seconds=dt, T=T, B=B, iters=iters,
                cores=torch.get_num_threads(), replay_frames=int(replay_size),
                last_loss=float(info.loss[-1]) if len(info.loss) else None,
                classes="SerialSampler(CpuWaitResetCollector) + rlpyt.algos.dqn.r2d1.R2D1 + "
                        "AtariR2d1Agent + PrioritizedSequenceReplayFrameBuffer")
