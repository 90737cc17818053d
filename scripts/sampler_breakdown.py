"""Host-side timing of the phases of one sampler time-step on the MI355X box."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlpyt_amd.agents.pg.atari import AtariFfAgent  # noqa: E402
from rlpyt_amd.envs.synthetic import SyntheticPong  # noqa: E402
from rlpyt_amd.samplers.gpu import GpuSampler  # noqa: E402
from rlpyt_amd.utils import logger  # noqa: E402

logger.set_quiet(True)


def main():
    B, T = 256, 128
    nw = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    s = GpuSampler(SyntheticPong, {}, batch_T=T, batch_B=B, n_workers=nw,
                   max_decorrelation_steps=10)
    a = AtariFfAgent()
    s.initialize(a, seed=0, bootstrap_value=True)
    torch.cuda.set_device(0)
    a.to_device(0)
    for itr in range(2):
        s.obtain_samples(itr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for itr in range(3):
        s.obtain_samples(itr)
    torch.cuda.synchronize()
    per_step = (time.perf_counter() - t0) / 3 / T
    # isolated pieces
    step = s.step_pyt
    obs_dev = s.samples.env.observation[0]
    sync = torch.cuda.synchronize

    def tm(fn, n=200):
        fn(); sync()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        sync()
        return (time.perf_counter() - t) / n * 1e3
    res = dict(workers=nw, per_step_ms=per_step * 1e3)
    res["h2d_obs_ms"] = tm(lambda: obs_dev.copy_(step.observation, non_blocking=True))
    pa, pr = s._all_action[0], s._all_reward[0]
    a.sample_mode(0)
    res["agent_step_ms"] = tm(lambda: a.step(obs_dev, pa, pr))
    act = a.step(obs_dev, pa, pr).action

    def d2h():
        step.action.copy_(act, non_blocking=True)
        torch.cuda.current_stream().synchronize()
    res["d2h_action_sync_ms"] = tm(d2h)
    t = time.perf_counter()
    for _ in range(50):
        for sem in s.ctrl.act_ready:
            sem.release()
        s._wait_obs()
    res["env_step_roundtrip_ms"] = (time.perf_counter() - t) / 50 * 1e3 if nw else None
    print(json.dumps(res), flush=True)
    os._exit(0)


if __name__ == "__main__":
    main()
