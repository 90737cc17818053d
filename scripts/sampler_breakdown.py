"""Host-side timing of the rollout phase on the MI355X box: per-step wall time and where the
master spends it (waiting for env workers / issuing device work / waiting for the device).

usage: sampler_breakdown.py [workers=64] [groups=2] [graph=1] [env_cost_us=0]"""
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
    arg = lambda i, d: type(d)(sys.argv[i]) if len(sys.argv) > i else d  # noqa: E731
    nw, ng, graph, cost = arg(1, 64), arg(2, 2), arg(3, 1), arg(4, 0.)
    s = GpuSampler(SyntheticPong, dict(step_cost_us=cost), batch_T=T, batch_B=B, n_workers=nw,
                   n_groups=ng, use_graph=bool(graph), max_decorrelation_steps=10)
    a = AtariFfAgent()
    s.initialize(a, seed=0, bootstrap_value=True)
    torch.cuda.set_device(0)
    a.to_device(0)
    for itr in range(2):
        s.obtain_samples(itr)
    torch.cuda.synchronize()
    for k in s.timing:
        s.timing[k] = 0.
    n = 4
    t0 = time.perf_counter()
    for itr in range(n):
        s.obtain_samples(2 + itr)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    res = dict(workers=nw, groups=s.n_groups, graph=bool(graph), env_cost_us=cost,
               per_step_ms=wall / n / T * 1e3, sps=n * T * B / wall)
    for k in ("wait_env_s", "device_issue_s", "device_wait_s"):
        res[k.replace("_s", "_ms_per_step")] = s.timing[k] / n / T * 1e3
    # isolated device pieces of one group-step (HIP events, 100 reps each)
    G = s.groups[0]

    def ev(fn, n=100):
        st = G.stream or torch.cuda.current_stream()
        with torch.cuda.stream(st):
            fn()
            a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a_.record()
            for _ in range(n):
                fn()
            b_.record()
        st.synchronize()
        return a_.elapsed_time(b_) / n * 1e3
    res["Bg"] = G.Bg
    res["h2d_us"] = ev(lambda: s._upload_steady(G, True))
    res["dedup"] = bool(G.dedup)
    if G.graph is not None:
        def rep():
            G.graph.replay()
        res["graph_us"] = ev(rep)
    else:
        def body():
            s._step_body(G)
        res["eager_body_us"] = ev(body)
    print(json.dumps(res), flush=True)
    s.shutdown()
    os._exit(0)


if __name__ == "__main__":
    main()
