"""Iteration-level parity cases shared by ``make_golden.py`` (reference PPO / A2C +
AtariFfAgent on CPU) and ``tests/test_algo_parity.py`` (this repo's classes on the GPU): the same
seeded sample batch, the same seeds for parameter init and minibatch shuffling."""
import numpy as np
import torch

T, B, A = 8, 6, 6
INIT_SEED, SHUFFLE_SEED, N_ITR, N_RUN = 4321, 3, 4, 2

# name, algo, algo kwargs, mid_batch_reset
CASES = [
    ("ppo", "PPO", dict(discount=0.99, learning_rate=1e-3, value_loss_coeff=1.,
                        entropy_loss_coeff=0.01, clip_grad_norm=1., gae_lambda=0.98,
                        minibatches=4, epochs=2, ratio_clip=0.1, linear_lr_schedule=True,
                        normalize_advantage=False), True),
    ("ppo_norm_valid", "PPO", dict(discount=0.95, learning_rate=5e-4, value_loss_coeff=0.5,
                                   entropy_loss_coeff=0.02, clip_grad_norm=0.5, gae_lambda=1,
                                   minibatches=2, epochs=1, ratio_clip=0.2,
                                   linear_lr_schedule=False, normalize_advantage=True), False),
    ("a2c", "A2C", dict(discount=0.99, learning_rate=1e-3, value_loss_coeff=0.5,
                        entropy_loss_coeff=0.01, clip_grad_norm=1., gae_lambda=0.97,
                        normalize_advantage=False), True),
    # the same PPO iteration under plain SGD: the update is LINEAR in the gradient, so a tight
    # tolerance on every later update separates a second-order bug from Adam's sign-like
    # amplification of round-off (which the 1e-2 band of the cases above has to allow for)
    ("ppo_sgd", "PPO", dict(discount=0.99, learning_rate=2e-2, value_loss_coeff=1.,
                            entropy_loss_coeff=0.01, clip_grad_norm=1., gae_lambda=0.98,
                            minibatches=4, epochs=2, ratio_clip=0.1, linear_lr_schedule=True,
                            normalize_advantage=False, OptimCls=torch.optim.SGD), True),
    # recurrent policy-gradient path (rlpyt/algos/pg/ppo.py:84-99, a2c.py:80-85): AtariLstmAgent,
    # whole columns per minibatch, LSTM restarted from the state recorded at row 0, valid mask
    ("ppo_lstm", "PPO", dict(discount=0.99, learning_rate=2e-2, value_loss_coeff=1.,
                             entropy_loss_coeff=0.01, clip_grad_norm=1., gae_lambda=0.95,
                             minibatches=2, epochs=2, ratio_clip=0.1, linear_lr_schedule=False,
                             normalize_advantage=True, OptimCls=torch.optim.SGD), False),
    ("a2c_lstm", "A2C", dict(discount=0.99, learning_rate=2e-2, value_loss_coeff=0.5,
                             entropy_loss_coeff=0.01, clip_grad_norm=1., gae_lambda=0.97,
                             normalize_advantage=False, OptimCls=torch.optim.SGD), False),
]
# cases whose optimizer is linear in the gradient: every update is held to fp32 tolerance
TIGHT_CASES = ("ppo_sgd", "ppo_lstm", "a2c_lstm")
LSTM_CASES = ("ppo_lstm", "a2c_lstm")


def lstm_init_state(lstm=512):
    """Seeded LSTM state the columns of the recurrent cases start from: (h, c), each [B, 1, H]."""
    g = torch.Generator().manual_seed(55)
    return (0.1 * torch.randn(B, 1, lstm, generator=g), 0.1 * torch.randn(B, 1, lstm, generator=g))


# The same iteration at a shape that SELECTS the bench-path kernels: M =
# T*B / minibatches = 1024 rows per minibatch -> bf16x6 trunk GEMMs (forward, input and weight
# gradient), conv2_fwd_x6, several images per persistent conv workgroup.  SGD, so every update of
# both reference iterations is held to fp32 tolerance.  (The observations are regenerated from
# the seed on both sides; the fixture holds only the reference's outputs.)
BIG_T, BIG_B = 64, 64
BIG_CASE = ("ppo_sgd_big", "PPO", dict(discount=0.99, learning_rate=2e-2, value_loss_coeff=1.,
                                       entropy_loss_coeff=0.01, clip_grad_norm=1., gae_lambda=0.98,
                                       minibatches=4, epochs=2, ratio_clip=0.1,
                                       linear_lr_schedule=True, normalize_advantage=False,
                                       OptimCls=torch.optim.SGD), True)


def batch_inputs(T=T, B=B, seed=77):
    """Env-side fields of the sample batch (the agent-side fields -- old probabilities, values,
    bootstrap value -- come from the reference agent's own forward and live in the fixture)."""
    g = torch.Generator().manual_seed(seed)
    obs = torch.randint(0, 256, (T, B, 4, 104, 80), dtype=torch.uint8, generator=g)
    keep = torch.rand((T, B, 4, 104, 80), generator=g) < 0.15
    obs = obs * keep.to(torch.uint8)
    all_action = torch.randint(0, A, (T + 1, B), generator=g)
    all_reward = torch.randint(-1, 2, (T + 1, B), generator=g).float()
    done = torch.rand(T, B, generator=g) < 0.08
    return dict(observation=obs, all_action=all_action, all_reward=all_reward, done=done)


def param_stats(params):
    return (np.array([p.detach().double().sum().item() for p in params]),
            np.array([p.detach().double().abs().sum().item() for p in params]))


# ---- DQN family: replay-driven iterations -------------------------------------------------
DQN_T, DQN_B, DQN_ITRS = 4, 8, 6
# name, algo kwargs (both sides), notes: uniform replay keeps the sampled indices independent of
# fp32 round-off over many updates; the prioritized case runs fewer updates (priorities feed back
# into the sampling, so late-update drift could pick different rows)
DQN_CASES = [
    ("catdqn_pri", dict(V_min=-3, V_max=3, discount=0.99, batch_size=16, min_steps_learn=64,
                        replay_size=512, replay_ratio=2, target_update_interval=3,
                        n_step_return=2, learning_rate=1e-4, clip_grad_norm=10., double_dqn=True,
                        prioritized_replay=True), 4),
    ("dqn_uniform", dict(discount=0.99, batch_size=16, min_steps_learn=64, replay_size=512,
                         replay_ratio=2, target_update_interval=3, n_step_return=2,
                         learning_rate=1e-4, clip_grad_norm=10., double_dqn=True,
                         prioritized_replay=False, delta_clip=1.), 6),
    ("dqn_pri", dict(discount=0.99, batch_size=16, min_steps_learn=64, replay_size=512,
                     replay_ratio=2, target_update_interval=2, n_step_return=1,
                     learning_rate=1e-4, clip_grad_norm=10., double_dqn=False,
                     prioritized_replay=True, delta_clip=1.), 4),
    # the "tight" case (VERDICT r3 weak #1a): plain SGD instead of Adam, so that round-off is not
    # amplified by Adam's sign-like first steps and every update can be held to a tight tolerance
    ("dqn_uniform_sgd", dict(discount=0.99, batch_size=16, min_steps_learn=64, replay_size=512,
                             replay_ratio=2, target_update_interval=3, n_step_return=2,
                             learning_rate=1e-3, clip_grad_norm=10., double_dqn=True,
                             prioritized_replay=False, delta_clip=1., OptimCls=torch.optim.SGD,
                             optim_kwargs=dict()), 6),
]


def dqn_batches(n_itr=DQN_ITRS):
    """Sampler batches fed to optimize_agent, one per iteration (fields of SamplesToBuffer)."""
    g = torch.Generator().manual_seed(177)
    out = []
    for _ in range(n_itr):
        obs = torch.randint(0, 256, (DQN_T, DQN_B, 4, 104, 80), dtype=torch.uint8, generator=g)
        keep = torch.rand((DQN_T, DQN_B, 4, 104, 80), generator=g) < 0.15
        out.append(dict(observation=obs * keep.to(torch.uint8),
                        action=torch.randint(0, A, (DQN_T, DQN_B), generator=g),
                        reward=torch.randint(-1, 2, (DQN_T, DQN_B), generator=g).float(),
                        done=torch.rand(DQN_T, DQN_B, generator=g) < 0.1))
    return out


# ---- R2D1: sequence replay + LSTM ---------------------------------------------------------------
R2D1_T, R2D1_B, R2D1_ITRS, R2D1_H = 8, 4, 9, 32
R2D1_MODEL = dict(fc_size=64, lstm_size=R2D1_H, head_size=32)
R2D1_KWARGS = dict(batch_T=8, batch_B=6, warmup_T=8, store_rnn_state_interval=8,
                   min_steps_learn=5 * R2D1_T * R2D1_B, replay_size=R2D1_T * R2D1_B * 16,
                   n_step_return=2, target_update_interval=2, prioritized_replay=True,
                   input_priorities=True, learning_rate=1e-4, double_dqn=True)


def r2d1_batches():
    """Sampler batches (env side + the agent_info a recurrent DQN agent records)."""
    g = torch.Generator().manual_seed(277)
    T, B, H = R2D1_T, R2D1_B, R2D1_H
    out = []
    for _ in range(R2D1_ITRS):
        obs = torch.randint(0, 256, (T, B, 4, 104, 80), dtype=torch.uint8, generator=g)
        keep = torch.rand((T, B, 4, 104, 80), generator=g) < 0.15
        all_action = torch.randint(0, A, (T + 1, B), generator=g)
        all_reward = torch.randint(-1, 2, (T + 1, B), generator=g).float()
        out.append(dict(observation=obs * keep.to(torch.uint8), all_action=all_action,
                        all_reward=all_reward, done=torch.rand(T, B, generator=g) < 0.04,
                        q=torch.randn(T, B, A, generator=g),
                        h=0.3 * torch.randn(T, B, 1, H, generator=g),
                        c=0.3 * torch.randn(T, B, 1, H, generator=g)))
    return out
