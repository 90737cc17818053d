"""bench.py -- BASELINE.json's metric on the MI355X: env-steps/sec (SPS), PPO Atari
[T=128, B=256] per GPU, 1/2/4/8 GPUs (weak scaling, one process per GPU over RCCL).

A "step" = one full PPO iteration of the hot path on one synthetic batch:
  rollout of T x B env steps (host envs, batched action selection on the device, sample
  batch resident in HBM) -> fused GAE scan -> epochs x minibatches of {device gather,
  AtariFfModel forward, fused PPO loss fwd+bwd kernel, backward, grad-clip, Adam}
  (under N>1: DistributedDataParallel all-reduces the 7.14 MB of gradients per minibatch).

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      -- the dominant own kernel of the timed region (the HBM->HBM minibatch
                   observation gather), HIP-event timed live inside the timed steps;
  kernels       -- the same live measurement for the other path kernels (GAE scan, PPO loss);
  roofline_gae_scaled -- the GAE scan at T=128, N=2^20 columns (the shape at which the
                   HBM criterion of BASELINE.md section 3 is meaningful), timed in this run;
  cpu_baseline  -- the oracle's CPU port of the reference iteration (oracle/ppo_cpu_port.py)
                   timed on this box's host cores on a bounded sample (rank 0, N=1 only).
Data: synthetic Atari-shaped env (no ALE in the image), random-init weights.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense f32 peak
KERNEL_NAMES = {
    "conv1_fwd": "conv1_fwd_kernel (gather + u8->f32 + conv 4->16 k8 s4 + bias + ReLU, fp32 MFMA)",
    "conv2_fwd": "conv2_fwd_kernel (conv 16->32 k4 s2 p1 + bias + ReLU, fp32 MFMA)",
    "conv2_bwd": "conv2_bwd_kernel (dgrad + ReLU masks + weight/bias grad in one pass, fp32 MFMA)",
    "conv2_dgrad": "conv2_dgrad_kernel (transposed conv + ReLU masks, fp32 MFMA)",
    "conv2_wgrad": "conv2_wgrad_kernel (+ bias grad, fp32 MFMA)",
    "conv1_wgrad": "conv1_wgrad_kernel (gather + u8->f32 + weight/bias grad, fp32 MFMA)",
    "obs_to_nhwc": "obs_to_nhwc_f32_kernel (minibatch gather + u8->f32 + CHW->HWC)",
    "gather_tb": "gather_wide_kernel (minibatch observation gather)",
    "gae": "scan_exact_kernel<GAE>", "ppo_loss": "pg_loss_kernel<PPO>",
    "gather_tb_small": "gather_flat_kernel (minibatch scalar fields)"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch-T", type=int, default=128)
    ap.add_argument("--batch-B", type=int, default=256)
    ap.add_argument("--workers", type=int, default=-1, help="env worker processes per rank "
                    "(-1: host cores / ranks, capped at B/10)")
    ap.add_argument("--env-cost-us", type=float, default=0., help="declared extra host cost "
                    "per env step (busy wait) to emulate an ALE-like emulator")
    ap.add_argument("--groups", type=int, default=-1, help="sampler pipeline groups (-1: auto)")
    ap.add_argument("--no-graph", action="store_true", help="no hipGraph for the sampling step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-B", type=int, default=0,
                    help="B of the bounded CPU sample (0: sized for ~15 s of CPU work)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--split-workers", action="store_true",
                    help="each env worker serves one pipeline group only (A/B lead, see DESIGN.md)")
    ap.add_argument("--no-fused-push", action="store_true",
                    help="debug: separate frame_push / conv1 / conv2 launches in the sampling step")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL)")
    ap.add_argument("--same-gpu", action="store_true",
                    help="debug: every rank uses cuda:0 (with --backend gloo) to exercise the "
                         "N>1 code path on a single-GPU box")
    ap.add_argument("--check-params", action="store_true",
                    help="debug: assert that all ranks hold identical parameters at the end")
    return ap.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    assert world == args.gpus or world == 1, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    import torch.distributed as dist

    from rlpyt_amd.agents.pg.atari import AtariFfAgent
    from rlpyt_amd.algos.pg.ppo import PPO
    from rlpyt_amd.envs.synthetic import SyntheticPong
    from rlpyt_amd.samplers.collections import AtariTrajInfo
    from rlpyt_amd.samplers.gpu import GpuSampler
    from rlpyt_amd.utils import ktimer, logger
    from rlpyt_amd.utils.seed import set_seed
    logger.set_quiet(True)

    T, B = args.batch_T, args.batch_B
    from rlpyt_amd.utils.misc import usable_cpus
    ncpu = os.cpu_count() or 8
    cpus = usable_cpus()          # honours the cgroup quota (16 CPUs on the 256-thread bench box)
    workers = args.workers
    if workers < 0:
        # ~10 envs per worker: waking more workers per step costs the master more than their
        # extra parallelism returns (measured 12..64 workers at B=256: 20-32 best, flat).
        # Workers sleep on a futex most of a step (25 of them keep ~10 CPUs busy), so the pool
        # may exceed this rank's CPU share by ~1.6x but not more, or the quota throttles it.
        workers = max(min(int(round(1.6 * cpus / world)) - 1, B // 10), 1)
    env_kwargs = dict(step_cost_us=args.env_cost_us)
    n_itr_total = args.warmup + args.steps

    # --- build the stack in the reference's order: sampler (forks workers) BEFORE any HIP
    #     call, then device placement, then DDP, then the algorithm ------------------------
    seed = 0 + 100 * rank
    set_seed(seed)
    sampler = GpuSampler(SyntheticPong, env_kwargs, batch_T=T, batch_B=B, n_workers=workers,
                         TrajInfoCls=AtariTrajInfo, max_decorrelation_steps=100,
                         n_groups=None if args.groups < 0 else args.groups,
                         use_graph=not args.no_graph, fused_push=not args.no_fused_push,
                         split_workers=args.split_workers)
    agent = AtariFfAgent()
    algo = PPO(discount=0.99, learning_rate=1e-3, value_loss_coeff=1., entropy_loss_coeff=0.01,
               clip_grad_norm=1., gae_lambda=0.98, minibatches=4, epochs=4, ratio_clip=0.1,
               linear_lr_schedule=True, normalize_advantage=False)
    examples = sampler.initialize(agent, seed=seed + 1, bootstrap_value=True, rank=rank,
                                  world_size=world)
    if args.same_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group(backend=args.backend, rank=rank, world_size=world)
    agent.to_device(local_rank)
    if world > 1:
        agent.data_parallel()
    algo.initialize(agent=agent, n_itr=max(n_itr_total, 1), batch_spec=sampler.batch_spec,
                    mid_batch_reset=sampler.mid_batch_reset, examples=examples,
                    world_size=world, rank=rank)

    def one_step(itr):
        agent.sample_mode(itr)
        samples, _infos = sampler.obtain_samples(itr)
        agent.train_mode(itr)
        return algo.optimize_agent(itr, samples)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for itr in range(args.warmup):
        one_step(itr)
    if not args.no_kernel_timing:
        ktimer.reset()
        ktimer.enable(True)
    for k in sampler.timing:
        sampler.timing[k] = 0.
    sync()
    t0 = time.perf_counter()
    t_sample = 0.
    for k in range(args.steps):
        itr = args.warmup + k
        ts = time.perf_counter()
        agent.sample_mode(itr)
        samples, _infos = sampler.obtain_samples(itr)
        t_sample += time.perf_counter() - ts
        agent.train_mode(itr)
        opt_info = algo.optimize_agent(itr, samples)
    sync()
    elapsed = time.perf_counter() - t0
    ktimer.enable(False)
    el = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = el.item()
    ksum = ktimer.summary() if not args.no_kernel_timing else {}
    sampler.shutdown()
    if args.check_params and world > 1:
        flat = torch.cat([p.detach().reshape(-1) for p in agent.parameters()])
        lo, hi = flat.clone(), flat.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert torch.equal(lo, hi), "ranks diverged: DDP gradient averaging is broken"
        if rank == 0:
            print("check-params: all ranks hold bit-identical parameters", file=sys.stderr)

    if rank == 0:
        steps_total = T * B * world * args.steps
        out = {
            "metric": "env-steps/sec (SPS) whole node, PPO Atari [T=128,B=256]",
            "value": steps_total / elapsed, "unit": "env-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (Atari-shaped SyntheticPong env on host cores, random-init "
                    "AtariFfModel)",
            "config": {"workload": f"PPO AtariFfAgent, GpuSampler T={T} B={B} per GPU, "
                                   "4 epochs x 4 minibatches, gae_lambda=0.98, Adam lr=1e-3",
                       "T": T, "B": B, "env_workers_per_gpu": workers,
                       "sampler_pipeline_groups": sampler.n_groups,
                       "env_step_cost_us": args.env_cost_us, "host_cores": ncpu,
                       "host_cpu_quota": cpus,
                       "parallelism": f"dp{world}"},
            "sampling_frac_of_step": t_sample / (elapsed if elapsed > 0 else 1.),
            "sampler": {"pipeline_groups": sampler.n_groups, "hip_graph": not args.no_graph,
                        "ms_per_time_step": t_sample / args.steps / T * 1e3,
                        "master_wait_env_ms": sampler.timing["wait_env_s"] / args.steps / T * 1e3,
                        "master_issue_ms": sampler.timing["device_issue_s"] / args.steps / T * 1e3,
                        "master_wait_device_ms":
                            sampler.timing["device_wait_s"] / args.steps / T * 1e3,
                        "per_batch_ms": {k[:-2]: sampler.timing[k] / args.steps * 1e3
                                         for k in ("pre_s", "loop_s", "tail_s", "post_s")}},
            "last_loss": opt_info.loss[-1] if opt_info.loss else None,
        }
        if ksum:
            # dominant own kernel of the timed region = largest total HIP-event time
            name, g = max(ksum.items(), key=lambda kv: kv[1]["avg_us"] * kv[1]["launches"])
            if "TFLOPs" in g:   # dense contraction: priced against the fp32 MFMA peak
                out["roofline"] = {"kernel": KERNEL_NAMES.get(name, name), "bound": "mfma",
                                   "achieved": g["TFLOPs"], "peak": F32_MFMA_PEAK_TFLOPS,
                                   "unit": "TFLOP/s", "frac": g["TFLOPs"] / F32_MFMA_PEAK_TFLOPS,
                                   "traffic": None, "avg_us": g["avg_us"],
                                   "launches": g["launches"],
                                   "alg_flops_per_launch": g["alg_flops_per_launch"],
                                   "alg_bytes_per_launch": g["alg_bytes_per_launch"]}
            else:
                out["roofline"] = {"kernel": KERNEL_NAMES.get(name, name),
                                   "bound": "hbm", "achieved": g["GBps"],
                                   "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                   "frac": g["GBps"] / HBM_PEAK_GBPS, "traffic": None,
                                   "avg_us": g["avg_us"], "launches": g["launches"],
                                   "alg_bytes_per_launch": g["alg_bytes_per_launch"]}
            # HBM traffic of that kernel from the separate rocprofv3 --pmc passes (a PMC pass
            # cannot run inside this timed process); committed under profiles/
            traffic = pmc_traffic(name, g)
            if traffic is not None:
                out["roofline"]["traffic"] = traffic["bytes_per_launch"]
                out["roofline"]["traffic_source"] = traffic["source"]
            out["kernels"] = {k: {kk: (round(vv, 3) if isinstance(vv, float) else vv)
                                  for kk, vv in v.items()} for k, v in ksum.items()}
        out["roofline_gae_scaled"] = gae_scaled_roofline()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(T, args.cpu_baseline_B, env_kwargs)
            # the isolated hot-path functions, HIP kernel beside the CPU restatement
            # on the same synthetic inputs (also yields the replay-kernel HBM rooflines)
            fn = isolated_functions(T, B)
            out["roofline_replay"] = fn.pop("roofline_replay")
            out["cpu_baseline"]["functions"] = fn
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic(name, g):
    """HBM bytes per launch of kernel ``name`` from profiles/r1_conv_pmc_counters.json
    ((2 * FETCH_SIZE + WRITE_SIZE) * 1024 per MI355X_MICROARCH.md), rescaled by algorithmic
    bytes when the bench launch is not the profiled M = 8192 one."""
    path = os.path.join(ROOT, "profiles", "r1_conv_pmc_counters.json")
    try:
        with open(path) as f:
            k = json.load(f)["kernels"][name]
    except (OSError, KeyError, ValueError):
        return None
    scale = g["alg_bytes_per_launch"] / k["alg_bytes"] if k.get("alg_bytes") else 1.
    return {"bytes_per_launch": k["hbm_bytes_corrected"] * scale,
            "source": "profiles/r1_conv_pmc_counters.json (rocprofv3 --pmc FETCH_SIZE / "
                      "WRITE_SIZE passes over scripts/conv_bench.py, M=8192)"}


def gae_scaled_roofline(T=128, log2n=20, iters=20):
    """GAE scan at a shape where HBM is the bound: [128, 2^20] (17 B/element)."""
    from rlpyt_amd import ops
    N = 1 << log2n
    r = torch.randn(T, N, device="cuda") * 0.5
    v = torch.randn(T, N, device="cuda")
    d = torch.rand(T, N, device="cuda") < 0.01
    bv = torch.randn(1, N, device="cuda")
    adv, ret = torch.empty_like(r), torch.empty_like(r)
    for _ in range(3):
        ops.gae(r, v, d, bv, 0.99, 0.98, advantage_dest=adv, return_dest=ret)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        ops.gae(r, v, d, bv, 0.99, 0.98, advantage_dest=adv, return_dest=ret)
    e.record()
    torch.cuda.synchronize()
    secs = s.elapsed_time(e) * 1e-3 / iters
    nbytes = T * N * 17 + 4 * N
    return {"kernel": "scan_exact_kernel<GAE> [128, 2^20]", "bound": "hbm",
            "achieved": nbytes / secs / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": nbytes / secs / 1e9 / HBM_PEAK_GBPS, "avg_us": secs * 1e6,
            "alg_bytes_per_launch": nbytes}


def _hip_us(fn, iters=30, warmup=3):
    for _ in range(warmup):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


def _cpu_us(fn, budget_s=0.5, max_reps=20):
    fn()
    t0 = time.perf_counter()
    n = 0
    while n < max_reps and (time.perf_counter() - t0 < budget_s or n == 0):
        fn()
        n += 1
    return (time.perf_counter() - t0) / n * 1e6


def isolated_functions(T, B):
    """Per-function latency at the BASELINE shapes (SURVEY 8(d) synthetic inputs): the HIP kernel
    through the C ABI (HIP events, data resident in HBM) and the oracle's CPU restatement of the
    reference function (host arrays, this box's cores), both in microseconds per call."""
    import numpy as np

    from oracle import np_oracle as O
    from rlpyt_amd import ops
    rng = np.random.RandomState(0)
    g = torch.Generator().manual_seed(0)
    res = {}

    def both(name, hip, cpu, **extra):
        res[name] = dict(hip_us=round(_hip_us(hip), 2), cpu_us=round(_cpu_us(cpu), 1), **extra)
        res[name]["ratio"] = round(res[name]["cpu_us"] / res[name]["hip_us"], 1)

    # ---- scans at [T, B] ------------------------------------------------------------------
    reward = (0.5 * torch.randn(T, B, generator=g))
    value = torch.randn(T, B, generator=g)
    done = torch.rand(T, B, generator=g) < 0.01
    bv = torch.randn(1, B, generator=g)
    r_d, v_d, d_d, bv_d = reward.cuda(), value.cuda(), done.cuda(), bv.cuda()
    r_n, v_n, d_n, bv_n = reward.numpy(), value.numpy(), done.numpy(), bv.numpy()
    both("gae", lambda: ops.gae(r_d, v_d, d_d, bv_d, 0.99, 0.98),
         lambda: O.generalized_advantage_estimation(r_n, v_n, d_n, bv_n, 0.99, 0.98),
         shape=[T, B])
    both("discount_return", lambda: ops.discount_return(r_d, d_d, bv_d, 0.99),
         lambda: O.discount_return(r_n, d_n, bv_n, 0.99), shape=[T, B])
    both("valid_from_done", lambda: ops.valid_from_done(d_d), lambda: O.valid_from_done(d_n),
         shape=[T, B])
    # ---- PPO loss forward + backward at M = T*B/4, A = 6 ------------------------------------
    M, A = T * B // 4, 6
    pn = torch.softmax(torch.randn(M, A, generator=g), -1)
    po = torch.softmax(torch.randn(M, A, generator=g), -1)
    val, adv, ret = (torch.randn(M, generator=g) for _ in range(3))
    act = torch.randint(0, A, (M,), generator=g)
    pn_d, po_d, val_d, adv_d, ret_d, act_d = (x.cuda() for x in (pn, po, val, adv, ret, act))

    def ppo_hip():
        p, v = pn_d.clone().requires_grad_(True), val_d.clone().requires_grad_(True)
        ops.ppo_loss(p, v, po_d, act_d, adv_d, ret_d, None, 0.1, 1., 0.01)[0].backward()

    def ppo_cpu():
        p, v = pn.clone().requires_grad_(True), val.clone().requires_grad_(True)
        O.ppo_loss_torch(p, v, po, act, adv, ret, None, 0.1, 1., 0.01)[0].backward()
    both("ppo_loss_fwd_bwd", ppo_hip, ppo_cpu, shape=[M, A],
         note="whole autograd op (clone + forward + backward), not the bare kernel")
    # ---- prioritized replay at the DQN config: 1M-leaf f64 tree, batch 128 ------------------
    Tr, Br, n = 62500, 16, 128
    tree_d = ops.DeviceSumTree(Tr, Br, 1, 3, default_value=1.0)
    tree_c = O.SumTree(Tr, Br, 1, 3, default_value=1.0)
    pri = np.abs(rng.randn(2000, Br)) ** 0.6
    tree_d.advance(2000)
    tree_c.advance(2000)
    u = rng.rand(n)
    u_d = torch.from_numpy(u).cuda()
    newp = np.abs(rng.randn(n)) ** 0.6
    newp_d = torch.from_numpy(newp).cuda()
    del pri

    def tree_hip():
        tree_d.sample(u_d)
        tree_d.update_batch_priorities(newp_d)

    def tree_cpu():
        tree_c.sample_with(u)
        tree_c.update_batch_priorities(newp)
    both("sumtree_sample_update", tree_hip, tree_cpu, leaves=Tr * Br, n=n)
    us = _hip_us(lambda: tree_d.sample(u_d))
    replay = {"sumtree_sample": {"bound": "latency", "avg_us": round(us, 2),
                                 "ns_per_sample": round(us * 1e3 / n, 1),
                                 "levels": int(tree_d.tree_levels)}}
    # ---- frame gathers (uint8, bit-exact): DQN batch and the R2D1 sequence batch ------------
    C, H, W, Tf, Bf = 4, 104, 80, 4096, 16
    frames = torch.randint(0, 256, (Tf + C - 1, Bf, H, W), dtype=torch.uint8, generator=g)
    fdone = torch.rand(Tf, Bf, generator=g) < 0.005
    f_d, fd_d = frames.cuda(), fdone.cuda()
    f_n, fd_n = frames.numpy(), fdone.numpy()
    ti = rng.randint(C, Tf - 200, size=n)
    bi = rng.randint(0, Bf, size=n)
    ti_d, bi_d = torch.from_numpy(ti).cuda(), torch.from_numpy(bi).cuda()
    out_d = torch.empty((n, C, H, W), dtype=torch.uint8, device="cuda")
    both("frames_gather", lambda: ops.frames_gather(f_d, fd_d, ti_d, bi_d, C, out=out_d),
         lambda: O.frames_gather(f_n, fd_n, ti, bi, C), n=n)
    nb = n * C * H * W * 2
    replay["frames_gather"] = {"bound": "hbm", "avg_us": res["frames_gather"]["hip_us"],
                               "alg_bytes_per_launch": nb,
                               "achieved": nb / res["frames_gather"]["hip_us"] / 1e3,
                               "peak": HBM_PEAK_GBPS, "unit": "GB/s"}
    ns, seq_T = 64, 125
    out_s = torch.empty((seq_T, ns, C, H, W), dtype=torch.uint8, device="cuda")
    both("frames_gather_seq",
         lambda: ops.frames_gather_seq(f_d, fd_d, ti_d[:ns], bi_d[:ns], C, seq_T, out=out_s),
         lambda: O.frames_gather_seq(f_n, fd_n, ti[:ns], bi[:ns], C, seq_T), n=ns, seq_T=seq_T)
    nb = ns * (seq_T + C - 1) * H * W + seq_T * ns * C * H * W      # 334 MB
    replay["frames_gather_seq"] = {"bound": "hbm",
                                   "avg_us": res["frames_gather_seq"]["hip_us"],
                                   "alg_bytes_per_launch": nb,
                                   "achieved": nb / res["frames_gather_seq"]["hip_us"] / 1e3,
                                   "peak": HBM_PEAK_GBPS, "unit": "GB/s"}
    for k in ("frames_gather", "frames_gather_seq"):
        replay[k]["frac"] = replay[k]["achieved"] / HBM_PEAK_GBPS
    res["roofline_replay"] = replay
    return res


def cpu_baseline(T, B_cpu, env_kwargs):
    """Oracle CPU port of the reference iteration on this box's host cores (bounded sample)."""
    from oracle.ppo_cpu_port import time_cpu_baseline
    from rlpyt_amd.envs.synthetic import SyntheticPong
    from rlpyt_amd.utils.misc import usable_cpus
    res = time_cpu_baseline(SyntheticPong, env_kwargs, T=T, B=B_cpu if B_cpu > 0 else None,
                            iters=1, threads=None, max_threads=usable_cpus())
    B_cpu = res["B"]
    return {"value": res["value"], "unit": "env-steps/s", "cores": res["cores"], "kind": "port",
            "sample": f"1 PPO iteration at [T={T}, B={B_cpu}] ({T * B_cpu} env steps, 16 "
                      f"minibatch updates), torch CPU with {res['cores']} threads (best of a thread-count "
                      f"calibration within the {usable_cpus():.0f} CPUs this process may use, "
                      f"{os.cpu_count()} hardware threads on the box), "
                      f"{res['seconds']:.1f} s",
            "seconds": res["seconds"]}


if __name__ == "__main__":
    main()
