import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rlpyt_amd.agents.pg.atari import AtariFfAgent
from rlpyt_amd.envs.synthetic import SyntheticPong
from rlpyt_amd.samplers.gpu import GpuSampler
from rlpyt_amd.utils import logger
logger.set_quiet(True)

def run(dedup, nw, ng, graph=True):
    s = GpuSampler(SyntheticPong, dict(points_to_end=1, max_steps=7), batch_T=6, batch_B=8,
                   n_workers=nw, n_groups=ng, frame_dedup=dedup, use_graph=graph, max_decorrelation_steps=0)
    a = AtariFfAgent()
    torch.manual_seed(11); np.random.seed(11)
    s.initialize(a, seed=4, bootstrap_value=True)
    torch.cuda.set_device(0); a.to_device(0)
    torch.manual_seed(12)
    out = []
    for itr in range(5):
        smp, _ = s.obtain_samples(itr); torch.cuda.synchronize()
        out.append([x.clone() for x in (smp.env.observation, smp.agent.action, smp.env.reward, smp.env.done)])
    s.shutdown()
    return out

def diff(a, b, tag):
    for i, (x, y) in enumerate(zip(a, b)):
        for name, u, v in zip(["obs", "act", "rew", "done"], x, y):
            if not torch.equal(u, v):
                ne = (u != v)
                while ne.dim() > 2: ne = ne.flatten(2).any(2)
                print(tag, "batch", i, name, "mismatch at [t,b]:", ne.nonzero().tolist()[:12]); return
    print(tag, "identical")


for nw, ng in [(2, 2), (3, 2)]:
    f1, f2 = run(False, nw, ng), run(False, nw, ng)
    diff(f1, f2, f"nw={nw} ng={ng} full-vs-full")
    d1 = run(True, nw, ng)
    diff(d1, f1, f"nw={nw} ng={ng} dedup-vs-full")
