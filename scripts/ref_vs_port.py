"""Build-container check (needs /root/reference): the oracle's CPU port of the PPO iteration timed beside
the REAL reference iteration (rlpyt SerialSampler + PPO + AtariFfAgent) on the same cores, same env, same
hyper-parameters -- the port is what bench.py reports as ``cpu_baseline`` (kind "port") on the GPU box,
where the reference itself is not available.  usage: ref_vs_port.py [T=128] [B=8] [timed iterations=2].  Runs: [128, 8] reference 1416 SPS, port 1516 SPS
(8 threads, round 2); [128, 256]: see DESIGN.md section 5 (round 4)."""
import sys, types, time, numpy as np, torch
sys.path.insert(0, "/root/reference"); sys.path.insert(0, "/root/repo")
from rlpyt.samplers.serial.sampler import SerialSampler
from rlpyt.algos.pg.ppo import PPO
from rlpyt.agents.pg.atari import AtariFfAgent
from rlpyt.utils.logging import logger
from rlpyt_amd.envs.synthetic import SyntheticPong
from oracle.ppo_cpu_port import PpoCpuPort
torch.set_num_threads(8)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
N_TIMED = int(sys.argv[3]) if len(sys.argv) > 3 else 2
s = SerialSampler(EnvCls=SyntheticPong, env_kwargs={}, batch_T=T, batch_B=B, max_decorrelation_steps=0)
agent = AtariFfAgent()
algo = PPO(discount=0.99, learning_rate=1e-3, value_loss_coeff=1., entropy_loss_coeff=0.01, clip_grad_norm=1., gae_lambda=0.98, minibatches=4, epochs=4, ratio_clip=0.1)
ex = s.initialize(agent, seed=1, bootstrap_value=True)
algo.initialize(agent=agent, n_itr=10, batch_spec=s.batch_spec, mid_batch_reset=True, examples=ex)
def ref_iter(itr):
    agent.sample_mode(itr); smp, _ = s.obtain_samples(itr); agent.train_mode(itr); algo.optimize_agent(itr, smp)
ref_iter(0)
t0 = time.perf_counter()
for i in range(1, 1 + N_TIMED): ref_iter(i)
ref = N_TIMED * T * B / (time.perf_counter() - t0)
port = PpoCpuPort(SyntheticPong, {}, T, B, seed=1, threads=8)
port.iteration()
t0 = time.perf_counter()
for i in range(N_TIMED): port.iteration()
pt = N_TIMED * T * B / (time.perf_counter() - t0)
print(f"[T={T}, B={B}], {N_TIMED} timed iteration(s), 8 threads: reference SerialSampler+PPO+AtariFfAgent: {ref:.0f} SPS ; oracle CPU port: {pt:.0f} SPS ; ratio port/ref {pt/ref:.2f}")
