"""CPU port of the reference's PPO iteration -- TEST / BASELINE INFRASTRUCTURE ONLY.

Restates, for timing on the bench box's host cores (``bench.py`` ``cpu_baseline`` leg,
``kind: "port"``) and for end-to-end parity tests, what the reference executes for
BASELINE.json's metric when run as SerialSampler + PPO + AtariFfAgent on CPU:

* rollout: ``CpuResetCollector.collect_batch`` -- per time-step one batched
  ``agent.step`` (model forward + ``torch.multinomial``) then a Python loop stepping each
  env (rlpyt/samplers/parallel/cpu/collectors.py:25-65);
* returns: ``PolicyGradientAlgo.process_returns`` with the Python time loop of
  ``generalized_advantage_estimation`` on torch CPU tensors (rlpyt/algos/pg/base.py:41-75,
  rlpyt/algos/utils.py:24-40);
* update: ``PPO.optimize_agent`` -- epochs x minibatches of {index, forward, loss of ~15
  torch ops, backward, clip_grad_norm_, Adam, 4 ``.item()`` syncs}
  (rlpyt/algos/pg/ppo.py:59-154).

The model is the reference's AtariFfModel architecture (rlpyt/models/pg/atari_ff_model.py)
written out flat.  Nothing under ``rlpyt_amd/`` imports this module.
"""
import time

import numpy as np
import torch
import torch.nn.functional as F

from . import np_oracle as O


class AtariFfModelCpu(torch.nn.Module):
    """conv(4->16,k8,s4) ReLU conv(16->32,k4,s2,p1) ReLU FC512 ReLU -> softmax pi, value."""

    def __init__(self, image_shape=(4, 104, 80), n_actions=6):
        super().__init__()
        c, h, w = image_shape
        self.c1 = torch.nn.Conv2d(c, 16, 8, 4, 0)
        self.c2 = torch.nn.Conv2d(16, 32, 4, 2, 1)
        h1, w1 = (h - 8) // 4 + 1, (w - 8) // 4 + 1
        h2, w2 = (h1 + 2 - 4) // 2 + 1, (w1 + 2 - 4) // 2 + 1
        self.fc = torch.nn.Linear(32 * h2 * w2, 512)
        self.pi = torch.nn.Linear(512, n_actions)
        self.value = torch.nn.Linear(512, 1)

    def forward(self, image):
        lead = image.shape[:-3]
        x = image.reshape((-1,) + tuple(image.shape[-3:])).type(torch.float).mul_(1. / 255)
        x = F.relu(self.c2(F.relu(self.c1(x))))
        x = F.relu(self.fc(x.reshape(x.shape[0], -1)))
        pi = F.softmax(self.pi(x), dim=-1)
        v = self.value(x).squeeze(-1)
        return pi.reshape(lead + pi.shape[-1:]), v.reshape(lead)


def gae_torch_loop(reward, value, done, bootstrap_value, discount, gae_lambda):
    """The reference's torch time loop, verbatim in structure (utils.py:24-40)."""
    advantage = torch.zeros(reward.shape, dtype=reward.dtype)
    nd = 1 - done
    nd = nd.type(reward.dtype)
    advantage[-1] = reward[-1] + discount * bootstrap_value * nd[-1] - value[-1]
    for t in reversed(range(len(reward) - 1)):
        delta = reward[t] + discount * value[t + 1] * nd[t] - value[t]
        advantage[t] = delta + discount * gae_lambda * nd[t] * advantage[t + 1]
    return advantage, advantage + value


class PpoCpuPort:
    def __init__(self, EnvCls, env_kwargs, T, B, seed=0, discount=0.99, learning_rate=1e-3,
                 value_loss_coeff=1., entropy_loss_coeff=0.01, clip_grad_norm=1.,
                 gae_lambda=0.98, minibatches=4, epochs=4, ratio_clip=0.1, threads=None):
        if threads:
            torch.set_num_threads(threads)
        self.T, self.B = T, B
        torch.manual_seed(seed)
        np.random.seed(seed)
        self.envs = [EnvCls(**env_kwargs) for _ in range(B)]
        for i, e in enumerate(self.envs):
            e.seed(seed + i)
        sp = self.envs[0].spaces
        self.A = sp.action.n
        self.model = AtariFfModelCpu(sp.observation.shape, self.A)
        self.opt = torch.optim.Adam(self.model.parameters(), lr=learning_rate)
        self.hp = dict(discount=discount, c_v=value_loss_coeff, c_e=entropy_loss_coeff,
                       clip_grad_norm=clip_grad_norm, lam=gae_lambda, minibatches=minibatches,
                       epochs=epochs, ratio_clip=ratio_clip)
        self.obs = np.stack([e.reset() for e in self.envs])
        self.buf = dict(
            observation=np.zeros((T, B) + sp.observation.shape, dtype=np.uint8),
            action=np.zeros((T, B), dtype=np.int64), reward=np.zeros((T, B), dtype=np.float32),
            done=np.zeros((T, B), dtype=bool), prob=np.zeros((T, B, self.A), dtype=np.float32),
            value=np.zeros((T, B), dtype=np.float32))

    def collect(self):
        buf = self.buf
        self.model.eval()
        for t in range(self.T):
            buf["observation"][t] = self.obs
            with torch.no_grad():
                pi, v = self.model(torch.from_numpy(self.obs))
                action = torch.multinomial(pi, num_samples=1).squeeze(-1)
            a_np = action.numpy()
            for b, env in enumerate(self.envs):
                o, r, d, info = env.step(a_np[b])
                if getattr(info, "traj_done", d):
                    o = env.reset()
                self.obs[b] = o
                buf["reward"][t, b] = r
                buf["done"][t, b] = d
            buf["action"][t] = a_np
            buf["prob"][t] = pi.numpy()
            buf["value"][t] = v.numpy()
        with torch.no_grad():
            _, bv = self.model(torch.from_numpy(self.obs))
        return bv[None]

    def optimize(self, bootstrap_value):
        hp, T, B = self.hp, self.T, self.B
        tb = {k: torch.from_numpy(v) for k, v in self.buf.items()}
        done_f = tb["done"].type(tb["reward"].dtype)
        advantage, return_ = gae_torch_loop(tb["reward"], tb["value"], done_f, bootstrap_value,
                                            hp["discount"], hp["lam"])
        self.model.train()
        mb = T * B // hp["minibatches"]
        infos = []
        for _ in range(hp["epochs"]):
            for idxs in O.iterate_mb_idxs(T * B, mb):
                Ti, Bi = idxs % T, idxs // T
                self.opt.zero_grad()
                pi, v = self.model(tb["observation"][Ti, Bi])
                loss, _pl, _vl, ent, ppl = O.ppo_loss_torch(
                    pi, v, tb["prob"][Ti, Bi], tb["action"][Ti, Bi], advantage[Ti, Bi],
                    return_[Ti, Bi], None, hp["ratio_clip"], hp["c_v"], hp["c_e"])
                loss.backward()
                gn = torch.nn.utils.clip_grad_norm_(self.model.parameters(),
                                                    hp["clip_grad_norm"])
                self.opt.step()
                infos.append((loss.item(), float(gn), ent.item(), ppl.item()))
        return infos

    def iteration(self):
        bv = self.collect()
        return self.optimize(bv)


def calibrate_threads(candidates=(4, 8, 16, 32, 64, 128), n_obs=256, max_threads=None):
    """Pick the torch intra-op thread count with the best fwd+bwd rate of the model on this
    box (more threads is NOT faster for these small convolutions) -> (threads, obs/s).
    ``max_threads``: the CPUs the process may actually use (cgroup quota); thread counts above
    it only measure the throttling."""
    import os
    ncpu = os.cpu_count() or 8
    if max_threads:
        ncpu = max(1, min(ncpu, int(max_threads)))
    model = AtariFfModelCpu()
    x = torch.randint(0, 256, (n_obs, 4, 104, 80), dtype=torch.uint8)
    best = (None, 0.)
    for th in candidates:
        if th > ncpu:
            continue
        torch.set_num_threads(th)
        rate = 0.
        for rep in range(2):  # first pass warms the thread pool
            t0 = time.perf_counter()
            pi, v = model(x)
            (pi.sum() + v.sum()).backward()
            rate = n_obs / (time.perf_counter() - t0)
        if rate > best[1]:
            best = (th, rate)
    return best


def time_cpu_baseline(EnvCls, env_kwargs, T=128, B=None, iters=1, threads=None, seed=0,
                      target_seconds=15., max_threads=None):
    """env-steps/sec of the CPU port on a bounded sample: ``iters`` PPO iterations at
    [T, B] (same hyper-parameters as the GPU run).  ``threads=None`` calibrates the thread
    count; ``B=None`` sizes the sample for about ``target_seconds`` of CPU work."""
    rate = None
    if threads is None:
        threads, rate = calibrate_threads(max_threads=max_threads)
    if B is None:
        if rate is None:
            _, rate = calibrate_threads((threads,))
        B = int(target_seconds * rate / (T * 4.4))  # 4 epochs fwd+bwd + sampling forward
        B = max(8, min(64, 1 << max(B, 1).bit_length() - 1))
    port = PpoCpuPort(EnvCls, env_kwargs, T, B, seed=seed, threads=threads)
    t0 = time.perf_counter()
    for _ in range(iters):
        port.iteration()
    dt = time.perf_counter() - t0
    return dict(value=T * B * iters / dt, seconds=dt, T=T, B=B, iters=iters,
                cores=torch.get_num_threads())
