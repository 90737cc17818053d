"""bench.py -- BASELINE.json's metric on the MI355X: env-steps/sec (SPS), PPO Atari
[T=128, B=256] per GPU, 1/2/4/8 GPUs (weak scaling, one process per GPU over RCCL).

A "step" = one full PPO iteration of the hot path on one synthetic batch:
  rollout of T x B env steps (host envs, batched action selection on the device, sample
  batch resident in HBM) -> fused GAE scan -> epochs x minibatches of {index-mode MFMA conv
  stack, trunk GEMM x W^T (bf16x6), fused trunk bias/ReLU + heads + PPO loss kernel, backward
  (trunk g W and g^T x on the same bf16x6 kernel body, conv2_bwd, conv1_wgrad), clip + Adam in two
  launches} -- no vendor GEMM / convolution in the update
  (under N>1: DistributedDataParallel all-reduces the 7.14 MB of gradients per minibatch).
`python bench.py --gpus N` launches its own N ranks (one process per GPU, RCCL).

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      -- the kernel with the largest TOTAL time over a whole timed iteration, rollout
                   kernels included (512 + launches each per iteration against 16 of every update
                   kernel); a rollout kernel (a captured graph node: no events of its own) is priced
                   on its IN-PIPELINE average from the committed rocprofv3 summary of this command
                   (profiles/r*_bench_kernel_stats.csv), its isolated live timing is the side field
                   `avg_us_isolated`; an f32-MFMA kernel is priced against the 157.3 TFLOP/s f32 MFMA peak; a
                   bf16-split kernel (f32 contraction issued as 3 / 6 bf16 MFMAs per MAC)
                   against both the HBM peak and the 2.5 PFLOP/s dense bf16 peak with its ISSUED
                   flops; roofline_update = the largest kernel of the update, roofline_gemm_tn the
                   weight gradient; durations measured live in this run;
  kernels       -- the same live measurement for every own kernel of the update;
  roofline_gae_scaled -- the GAE scan at T=128, N=2^20 columns (the shape at which the
                   HBM criterion of BASELINE.md section 3 is meaningful), timed in this run;
  value_env200us -- the same metric with a declared ALE-like host cost of 200 us per env step
                   (busy wait in the env workers; `value` itself is measured at 0 us: the
                   framework's own ceiling);
  cpu_baseline_env200us, ratio_env200us -- the reference iteration with the SAME declared 200 us per
                   env step, and value_env200us over it (like for like);
  configs       -- (1 GPU) compact objects of BASELINE configs #3 (dqn) and #5 (r2d1): child runs of
                   `bench.py --config ...` after the headline's timed region (--no-extra-configs skips);
  cpu_baseline  -- the UNMODIFIED reference (oracle/_ref, copied from /root/reference at build time by
                   oracle/make_ref.py: SerialSampler + PPO + AtariFfAgent) timed on this box's host
                   cores at the full [128, 256] batch (rank 0, N=1 only; kind "reference"); the oracle's
                   CPU port (kind "port") only when that copy is absent;
  multi_gpu     -- (N>1) world size as torch.distributed sees it, per-rank usable CPUs, the
                   all-reduce time of one gradient set measured live, per-GPU SPS.
`--config dqn` / `--config r2d1` run BASELINE configs #3 / #5 end to end instead (full-size HBM
replay, prioritized tree, frame / sequence gathers; every no-grad network pass on own kernels, the
DQN update as one captured hipGraph) and report SPS, updates/s, the replay kernels' rooflines
taken from those buffers, the rollout's hand-off breakdown (`sampler`) and the reference's own
iteration on the host cores (`cpu_baseline`).
`--trace-markers` brackets the timed region with a marker kernel so that
`scripts/trace_region.py` can cut a rocprofv3 kernel trace to exactly that region.
Data: synthetic Atari-shaped env (no ALE in the image), random-init weights.
"""
import argparse
import json
import os
import sys
import time

import torch


def resolve_opt_info(info):
    """Wait for an optimize_agent call's diagnostics (rlpyt_amd/utils/deferred.py)."""
    from rlpyt_amd.utils.deferred import resolve
    return resolve(info)


ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense f32 peak
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (no sparsity)
# bf16-split kernels (DESIGN.md "fp32 contractions on the bf16 pipe"): issued bf16 MFMA flops per
# algorithmic fp32 flop (3 exact pieces of one operand; 6 products of two 3-piece operands)
# (convs_fwd: conv1 as bf16x3 + conv2 as bf16x6 in one kernel: (3 x 3.891 + 6 x 1.769) / 5.660 MFLOP / image)
BF16_SPLIT = {"convs_fwd": 3.94, "conv1_wgrad": 3, "conv2_bwd": 6, "gemm_nt": 6,
              "gemm_nt_dgrad": 6, "gemm_tn": 6}
KERNEL_NAMES = {
    "convs_fwd": "convs_fwd_fused_kernel (gather + u8->bf16 + conv 4->16 k8 s4 + bias + ReLU as an exact "
                 "bf16x3 split of w1, y1 to HBM once and through LDS into conv 16->32 k4 s2 p1 + bias + ReLU "
                 "+ sign mask of y2 as a bf16x6 split; f32 accumulate, dropped terms <= 2^-24, 2^-27 rms)",
    "conv2_bwd": "conv2_bwd_x6_kernel (dgrad + ReLU masks + weight/bias grad in one pass over the "
                 "images, conv2's ReLU mask from the forward pass's sign bits; bf16x6 split of both "
                 "operands of both contractions, f32 accumulate)",
    "conv1_wgrad": "conv1_wgrad_kernel (gather + u8->bf16 + weight/bias grad; exact bf16x3 split "
                   "of dy1, f32 accumulate)",
    "gemm_nt": "gemm_nt_x6_kernel<128> (update trunk forward x W^T [8192,3456]x[512,3456]^T: f32 GEMM "
               "from three-piece bf16 splits of both operands, six products, f32 accumulate, dropped "
               "terms <= 2^-24, 2^-27 rms)",
    "gemm_nt_dgrad": "gemm_nt_x6_kernel<256> (update trunk input gradient g W [8192,512]x[512,3456] as "
                     "g (W^T)^T on a transposed copy of W; same bf16x6 arithmetic)",
    "gemm_tn": "gemm_tn_x6_kernel + gemm_reduce_slots_kernel (update trunk weight gradient g^T x "
               "[8192,512]^T x [8192,3456]: 8 K chunks <-> XCDs, partial tiles, fixed-order sum; "
               "bf16x6 in lock step, K-major LDS tiles read with ds_read_b64_tr_b16)",
    "obs_to_nhwc": "obs_to_nhwc_f32_kernel (minibatch gather + u8->f32 + CHW->HWC)",
    "gather_tb": "gather_wide_kernel (minibatch observation gather)",
    "gae": "scan_exact_kernel<GAE>", "ppo_loss": "pg_loss_kernel<PPO>",
    "gather_tb_small": "gather_flat_kernel (minibatch scalar fields)"}


def sampler_stats(sampler, timing, wt, wt0, steps, T, t_sample, use_graph):
    """The `sampler` object of a bench line: master-side times per time step, per-batch phases and
    what the env workers did (waiting for the master's actions / stepping envs)."""
    worker_ms = None
    if wt0 is not None:
        d = (wt - wt0) / (steps * T) * 1e-6
        dd = wt - wt0
        worker_ms = {"wait_mean": float(d[:, 0].mean()), "wait_max": float(d[:, 0].max()),
                     "step_mean": float(d[:, 1].mean()), "step_max": float(d[:, 1].max()),
                     "waited_frac": float(dd[:, 4].sum() / max(dd[:, 2].sum(), 1.)),
                     "wake_us_mean": float(dd[:, 3].sum() / max(dd[:, 4].sum(), 1.) * 1e-3)}
        n_gs = max(timing.get("chain_steps", 0.), 1.)
        worker_ms["chain_us"] = {k: timing.get(f"chain_{k}_s", 0.) / n_gs * 1e6
                                 for k in ("issue", "device", "post")}
    return {"pipeline_groups": sampler.n_groups, "hip_graph": use_graph,
            "native_serve_loop": getattr(sampler, "_native", None) is not None,
            "ms_per_time_step": t_sample / steps / T * 1e3,
            "master_wait_env_ms": timing["wait_env_s"] / steps / T * 1e3,
            "master_issue_ms": timing["device_issue_s"] / steps / T * 1e3,
            "master_wait_device_ms": timing["device_wait_s"] / steps / T * 1e3,
            "per_batch_ms": {k[:-2]: timing[k] / steps * 1e3
                             for k in ("pre_s", "loop_s", "tail_s", "post_s")},
            "worker_ms_per_time_step": worker_ms}


def trace_marker(args):
    if args.trace_markers:
        torch.full((64,), 0.5, device="cuda").erfinv_()
        torch.cuda.synchronize()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed iterations (default: 20 for ppo / r2d1, 300 for dqn -- a few seconds)")
    ap.add_argument("--warmup", type=int, default=None,
                    help="untimed iterations before them (default: 5 / 3 / 20): step and tail graphs are "
                         "captured and the C serve loop takes over during the first two")
    ap.add_argument("--batch-T", type=int, default=128)
    ap.add_argument("--batch-B", type=int, default=256)
    ap.add_argument("--workers", "--workers-per-rank", dest="workers", type=int, default=-1,
                    help="env worker processes per rank (-1: 1.25 x this rank's share of the CPU "
                         "quota, capped at B/10; set it on boxes without the 16-CPU quota)")
    ap.add_argument("--env-cost-us", type=float, default=0., help="declared extra host cost "
                    "per env step (busy wait) to emulate an ALE-like emulator")
    ap.add_argument("--config", default="ppo", choices=["ppo", "dqn", "r2d1"],
                    help="ppo = BASELINE config #2/#4 (the metric); dqn = #3; r2d1 = #5")
    ap.add_argument("--env-cost-leg-us", type=float, default=200.,
                    help="second, shorter leg with this declared env cost (0: skip)")
    ap.add_argument("--env-cost-leg-steps", type=int, default=2)
    ap.add_argument("--pin-workers", action="store_true",
                    help="pin env worker w of rank r to CPU affinity['workers_cpus'][w] "
                         "(the reference's set_affinity)")
    ap.add_argument("--replay-fill-itrs", type=int, default=-1,
                    help="dqn / r2d1: sampling-only iterations before the timed ones (-1: auto)")
    ap.add_argument("--groups", type=int, default=-1, help="sampler pipeline groups (-1: auto)")
    ap.add_argument("--no-graph", action="store_true", help="no hipGraph for the sampling step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-baseline-leg", action="store_true",
                    help="skip the second cpu_baseline iteration at the declared env cost (~25 s)")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="ppo, 1 GPU: do not append the compact BASELINE config #3 / #5 objects "
                         "(`configs.dqn`, `configs.r2d1`: each a `bench.py --config ...` child run "
                         "after the headline measurement, ~1.5 min each)")
    ap.add_argument("--cpu-baseline-B", type=int, default=256,
                    help="B of the CPU sample (default: the full batch, one iteration ~20 s; "
                         "0: sized for ~15 s of CPU work)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--split-workers", action="store_true",
                    help="each env worker serves one pipeline group only (A/B lead, see DESIGN.md)")
    ap.add_argument("--no-zero-copy", action="store_true",
                    help="A/B: hand the sampled actions to the env workers through a D2H copy instead "
                         "of letting the step's head kernel write them in the page-locked step buffer")
    ap.add_argument("--frozen-env", action="store_true",
                    help="diagnostics: env.step() returns the standing observation (no dynamics, no "
                         "drawing) -- the rollout's floor without the synthetic env's own cost")
    ap.add_argument("--no-fused-push", action="store_true",
                    help="debug: separate frame_push / conv1 / conv2 launches in the sampling step")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL)")
    ap.add_argument("--same-gpu", action="store_true",
                    help="debug: every rank uses cuda:0 (with --backend gloo) to exercise the "
                         "N>1 code path on a single-GPU box")
    ap.add_argument("--check-params", action="store_true",
                    help="debug: assert that all ranks hold identical parameters at the end")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch contract only: every rank joins the process group, reports what "
                         "it would use (device, CPUs, env workers) and exits without touching a GPU")
    ap.add_argument("--trace-markers", action="store_true",
                    help="profiling: one erfinv kernel (used nowhere else) right before and right after "
                         "the timed region, so scripts/trace_region.py can cut a rocprofv3 kernel trace "
                         "to exactly that region")
    ap.add_argument("--master-port", type=int, default=0,
                    help="rendezvous port of the self-launched ranks (0: a free one)")
    return ap.parse_args()


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args):
    """``python bench.py --gpus N`` without a launcher: re-exec this command line under
    ``python -m torch.distributed.run`` with N ranks on 127.0.0.1 (rank r <-> GPU r, one process
    per GPU -- rlpyt/runners/sync_rl.py:60-101), stream the ranks' output through, and return
    the launcher's exit code.  Refuses to run when the box has fewer than N devices (unless
    --same-gpu / --dry-run), instead of silently measuring fewer ranks."""
    import subprocess
    if not (args.same_gpu or args.dry_run):
        n_dev = torch.cuda.device_count()
        if n_dev < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but this box shows {n_dev} device(s); "
                  "refusing to run fewer ranks than asked (use --same-gpu --backend gloo to "
                  "exercise the N>1 code path on one GPU)", file=sys.stderr)
            return 2
    port = args.master_port or _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.config != "ppo":
        return replay_config_main(args)
    if args.steps is None:
        args.steps = 20
    if args.warmup is None:
        args.warmup = 5
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}: the launcher and "
                         "the flag disagree (python bench.py --gpus N launches its own N ranks)")
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
        # ~1.25 workers per CPU of this rank's quota share, at most one per 10 envs: waking more
        # workers per step costs the master more than their extra parallelism returns.  With the
        # worker loop body in C (round 3) fewer workers do: interleaved runs at B=256 on the 16-CPU
        # box, 16 / 20 / 25 workers = 793 / 799 / 781 K SPS (means of 3; round 2, Python loop body:
        # 20-32 flat) -- profiles/r3_rollout_sweep.jsonl.
        workers = max(min(int(round(1.25 * cpus / world)), B // 10), 1)
    import multiprocessing as mp
    # declared host cost per env step, in fork-shared memory so that the second leg can change it
    # under the already forked env workers
    cost_ref = mp.get_context("fork").RawValue("d", float(args.env_cost_us))
    env_kwargs = dict(step_cost_ref=cost_ref)
    if args.frozen_env:
        env_kwargs["frozen"] = True
    leg_steps = args.env_cost_leg_steps if (args.env_cost_leg_us > 0 and args.env_cost_us == 0) else 0
    n_itr_total = (args.warmup + args.steps + max(1, min(args.steps, 5))
                   + (0 if args.no_kernel_timing else 2)
                   + (1 + leg_steps if leg_steps else 0) + (2 if world > 1 else 0))
    # worker processes and their CPUs the reference's way: one worker per entry of
    # affinity["workers_cpus"] (rlpyt/samplers/parallel/base.py:157-172); rank r takes the r-th
    # block of the hardware threads
    # block of the CPUs this process tree may be scheduled on (the affinity mask, not the raw
    # hardware thread count); the POOL SIZE above comes from the cgroup quota share cpus / world
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        allowed = list(range(ncpu))
    per_rank = max(len(allowed) // max(world, 1), 1)
    block = allowed[rank * per_rank:(rank + 1) * per_rank] or allowed
    workers_cpus = [block[w % len(block)] for w in range(workers)]
    affinity = dict(cuda_idx=local_rank, workers_cpus=workers_cpus,
                    set_affinity=bool(args.pin_workers))
    if args.dry_run:
        return dry_run(args, rank, world, local_rank, cpus, workers, block)

    # --- build the stack in the reference's order: sampler (forks workers) BEFORE any HIP
    #     call, then device placement, then DDP, then the algorithm ------------------------
    seed = 0 + 100 * rank
    set_seed(seed)
    sampler = GpuSampler(SyntheticPong, env_kwargs, batch_T=T, batch_B=B,
                         TrajInfoCls=AtariTrajInfo, max_decorrelation_steps=100,
                         n_groups=None if args.groups < 0 else args.groups,
                         use_graph=not args.no_graph, fused_push=not args.no_fused_push,
                         split_workers=args.split_workers, zero_copy=not args.no_zero_copy)
    agent = AtariFfAgent()
    algo = PPO(discount=0.99, learning_rate=1e-3, value_loss_coeff=1., entropy_loss_coeff=0.01,
               clip_grad_norm=1., gae_lambda=0.98, minibatches=4, epochs=4, ratio_clip=0.1,
               linear_lr_schedule=True, normalize_advantage=False)
    examples = sampler.initialize(agent, affinity=affinity, seed=seed + 1, bootstrap_value=True,
                                  rank=rank, world_size=world)
    assert sampler.n_workers == min(workers, B)
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
    for k in sampler.timing:
        sampler.timing[k] = 0.
    wt = getattr(getattr(sampler, "ctrl", None), "worker_timing", None)
    wt0 = None if wt is None else wt.copy()
    sync()
    trace_marker(args)
    cg0 = cgroup_cpu_stat()
    t0 = time.perf_counter()
    t_sample = 0.
    itr_starts, episodes_done, sampling_s = [], [], []
    for k in range(args.steps):
        itr = args.warmup + k
        ts = time.perf_counter()
        itr_starts.append(ts)
        agent.sample_mode(itr)
        samples, _infos = sampler.obtain_samples(itr)
        t_sample += time.perf_counter() - ts
        sampling_s.append(time.perf_counter() - ts)
        episodes_done.append(len(_infos))
        agent.train_mode(itr)
        opt_info = algo.optimize_agent(itr, samples)
    sync()
    elapsed = time.perf_counter() - t0
    host_quota = cgroup_cpu_delta(cg0, cgroup_cpu_stat(), elapsed, itr_starts + [t0 + elapsed])
    host_quota["episodes_completed"] = episodes_done[:64]
    # (host clock around obtain_samples: includes the wait for the previous iteration's updates the host ran ahead of)
    host_quota["obtain_samples_ms"] = [round(x * 1e3, 1) for x in sampling_s[:64]]
    trace_marker(args)
    # ---- phase-timing leg: where an iteration's time goes.  In the timed region optimize_agent hands
    # back diagnostics whose copy to the host is still in flight (utils/deferred.py) and nothing reads
    # them, so the host is already inside the next sampling phase while the last minibatches run -- a host
    # clock around the sampler then includes that wait.  Here every iteration reads its diagnostics at
    # once (as a runner that stores them per iteration does) and the clocks measure the sampler alone.
    ph_steps = max(1, min(args.steps, 5))
    for k in sampler.timing:
        sampler.timing[k] = 0.
    wt0 = None if wt is None else wt.copy()
    sync()
    tp = time.perf_counter()
    t_sample = 0.
    for k in range(ph_steps):
        itr = args.warmup + args.steps + k
        ts = time.perf_counter()
        agent.sample_mode(itr)
        samples, _infos = sampler.obtain_samples(itr)
        t_sample += time.perf_counter() - ts
        agent.train_mode(itr)
        resolve_opt_info(algo.optimize_agent(itr, samples))
    sync()
    ph_elapsed = time.perf_counter() - tp
    timing = dict(sampler.timing)         # (the env-cost leg below keeps adding to sampler.timing)
    worker_ms = None
    if wt0 is not None:
        # per env worker and time step: ms waiting for the master's actions / ms stepping envs
        d = (wt - wt0) / (ph_steps * T) * 1e-6
        dd = wt - wt0
        worker_ms = {"wait_mean": float(d[:, 0].mean()), "wait_max": float(d[:, 0].max()),
                     "step_mean": float(d[:, 1].mean()), "step_max": float(d[:, 1].max()),
                     # of the group-steps a worker really had to wait for: share, and us from the
                     # master's post to the worker running again
                     "waited_frac": float(dd[:, 4].sum() / max(dd[:, 2].sum(), 1.)),
                     "wake_us_mean": float(dd[:, 3].sum() / max(dd[:, 4].sum(), 1.) * 1e-3)}
        n_gs = max(timing.get("chain_steps", 0.), 1.)
        worker_ms["chain_us"] = {k: timing.get(f"chain_{k}_s", 0.) / n_gs * 1e6
                                 for k in ("issue", "device", "post")}
    el = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = el.item()
    # ---- kernel-timing leg: HIP events around every launch of the path's kernels.  Outside the
    # timed region (events between the kernels of the timed iterations can only
    # slow the headline and fold launch gaps into the averages): two more iterations of the same
    # workload.
    ksum = {}
    if not args.no_kernel_timing:
        ktimer.reset()
        ktimer.enable(True)
        for k in range(2):
            one_step(args.warmup + args.steps + ph_steps + k)
        sync()
        ktimer.enable(False)
        ksum = ktimer.summary()
    # ---- second leg: the same iteration with a declared ALE-like emulator cost per env step ----
    leg = None
    if leg_steps:
        cost_ref.value = float(args.env_cost_leg_us)
        one_step(args.warmup + args.steps + ph_steps + 2)           # untimed: workers pick up the new cost
        sync()
        tl = time.perf_counter()
        ts_leg = 0.
        for k in range(leg_steps):
            itr = args.warmup + args.steps + ph_steps + 3 + k
            t1 = time.perf_counter()
            agent.sample_mode(itr)
            samples, _infos = sampler.obtain_samples(itr)
            ts_leg += time.perf_counter() - t1
            agent.train_mode(itr)
            algo.optimize_agent(itr, samples)
        sync()
        el2 = torch.tensor([time.perf_counter() - tl], dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(el2, op=dist.ReduceOp.MAX)
        leg = dict(elapsed=el2.item(), steps=leg_steps, sampling=ts_leg)
        cost_ref.value = float(args.env_cost_us)
    # ---- N > 1: what the process group looks like, and one gradient all-reduce timed live --------
    multi = None
    if world > 1:
        nparam = sum(p.numel() for p in agent.parameters())
        buf = torch.zeros(nparam, dtype=torch.float32, device="cuda")
        for _ in range(3):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        dist.barrier()
        ta = time.perf_counter()
        n_ar = 32
        for _ in range(n_ar):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        ar_us = (time.perf_counter() - ta) / n_ar * 1e6
        # exposed all-reduce time of an iteration's 16 updates: the same update with DDP's gradient
        # synchronisation on and off (model.no_sync()), both bracketed by device synchronisation, on
        # the batch of one more rollout.  Run LAST: without the all-reduce the ranks' parameters
        # drift apart (nothing is measured after this).
        upd_ms = {}
        if not args.check_params:
            import contextlib
            itr_x = n_itr_total - 2
            agent.sample_mode(itr_x)
            samples, _ = sampler.obtain_samples(itr_x)
            agent.train_mode(itr_x)
            for key, ctx in (("sync", contextlib.nullcontext), ("no_sync", agent.model.no_sync)):
                sync()
                tu = time.perf_counter()
                with ctx():
                    algo.optimize_agent(itr_x, samples)
                torch.cuda.synchronize()
                upd_ms[key] = (time.perf_counter() - tu) * 1e3
        cpu_list = [None] * world
        dist.all_gather_object(cpu_list, dict(
            rank=rank, device=torch.cuda.current_device(),
            device_name=torch.cuda.get_device_name(torch.cuda.current_device()),
            cpu_quota_share=round(cpus / world, 2), env_workers=sampler.n_workers,
            cpu_block=[block[0], block[-1]],
            # where this rank's iteration goes: a curve that bends with rollout_ms is host-bound (CPUs
            # per rank), one that bends with allreduce_exposed_ms is xGMI-bound
            rollout_ms=t_sample / ph_steps * 1e3,
            update_ms=(ph_elapsed - t_sample) / ph_steps * 1e3,
            ms_per_time_step=t_sample / ph_steps / T * 1e3,
            update_ms_ddp_sync=upd_ms.get("sync"), update_ms_no_sync=upd_ms.get("no_sync"),
            allreduce_exposed_ms=(upd_ms["sync"] - upd_ms["no_sync"]) if upd_ms else None))
        multi = dict(dist_world_size=dist.get_world_size(), backend=dist.get_backend(),
                     rccl_version=_rccl_version(), host_cpu_quota=cpus,
                     ranks=cpu_list, grad_bytes=nparam * 4,
                     allreduce_us_per_minibatch=ar_us,
                     allreduce_ms_per_iteration=ar_us * algo.epochs * algo.minibatches / 1e3,
                     note="allreduce_us_per_minibatch: stand-alone all-reduce of one gradient-sized "
                          "buffer after the timed region; ranks[].allreduce_exposed_ms: one iteration's "
                          "16 updates with DDP's gradient synchronisation on minus off (no_sync), i.e. "
                          "what of the all-reduce is NOT hidden under backward")
    sampler.shutdown()
    # a throughput number over non-finite training is not a measurement
    assert all(torch.isfinite(p).all().item() for p in agent.parameters()), \
        "non-finite parameters after the timed iterations"
    if args.check_params and world > 1:
        flat = torch.cat([p.detach().reshape(-1) for p in agent.parameters()])
        lo, hi = flat.clone(), flat.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if not torch.equal(lo, hi):
            d = (hi - lo)
            names = []
            off = 0
            for n_, p_ in agent.model.named_parameters():
                k_ = p_.numel()
                m_ = d[off:off + k_].abs().max().item()
                if m_ > 0:
                    names.append(f"{n_}: max diff {m_:.3e} (max |p| {p_.abs().max().item():.3e})")
                off += k_
            raise AssertionError("ranks diverged: DDP gradient averaging is broken -- " + "; ".join(names))
        if rank == 0:
            print("check-params: all ranks hold bit-identical parameters", file=sys.stderr)

    if rank == 0:
        steps_total = T * B * world * args.steps
        out = {
            "metric": "env-steps/sec (SPS) whole node, PPO Atari [T=128,B=256]",
            "value": steps_total / elapsed, "unit": "env-steps/s",
            "n_gpus": dist.get_world_size() if world > 1 else 1,
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
            "host_quota_in_timed_region": host_quota,
            "sampling_frac_of_step": t_sample / (ph_elapsed if ph_elapsed > 0 else 1.),
            "phase_timing": f"sampling_frac_of_step and the sampler object: {ph_steps} iterations after "
                            "the timed region with the diagnostics read back inside optimize_agent "
                            f"({ph_elapsed / ph_steps * 1e3:.2f} ms per iteration there); in the timed "
                            "region their copy to the host is left in flight (utils/deferred.py)",
            # how the SAME launch rule lays out 2 / 4 / 8 ranks under this box's CPU quota: the
            # rollout needs ~B env steps of host CPU per time step and rank, so the first scaling
            # curve on a quota-limited box is a curve of host CPUs per rank
            "layout_at_n_ranks": {str(n): {"cpu_quota_share": round(cpus / n, 2),
                                           "env_workers": max(min(int(round(1.25 * cpus / n)), B // 10), 1),
                                           "serve_threads_spin": bool(cpus / n >= 6)}
                                  for n in (1, 2, 4, 8)},
            "sampler": {"pipeline_groups": sampler.n_groups, "hip_graph": not args.no_graph,
                        "ms_per_time_step": t_sample / ph_steps / T * 1e3,
                        "master_wait_env_ms": timing["wait_env_s"] / ph_steps / T * 1e3,
                        "master_issue_ms": timing["device_issue_s"] / ph_steps / T * 1e3,
                        "master_wait_device_ms":
                            timing["device_wait_s"] / ph_steps / T * 1e3,
                        "per_batch_ms": {k[:-2]: timing[k] / ph_steps * 1e3
                                         for k in ("pre_s", "loop_s", "tail_s", "post_s")},
                        "worker_ms_per_time_step": worker_ms},
            "last_loss": opt_info.loss[-1] if opt_info.loss else None,
            "losses_finite": all(x == x and abs(x) != float("inf") for x in opt_info.loss),
        }
        out["config"]["env_worker_cpus"] = ("pinned (affinity['workers_cpus'])" if args.pin_workers
                                            else "not pinned (set_affinity=False)")
        if leg is not None:
            key = f"value_env{int(args.env_cost_leg_us)}us"
            out[key] = T * B * world * leg["steps"] / leg["elapsed"]
            out[key + "_detail"] = {
                "env_step_cost_us": args.env_cost_leg_us, "steps": leg["steps"],
                "ms_per_step": leg["elapsed"] / leg["steps"] * 1e3,
                "sampling_frac_of_step": leg["sampling"] / leg["elapsed"],
                "note": "declared ALE-like emulator cost: busy wait in the env workers; the "
                        f"{T * B} env steps of a batch then cost {T * B * args.env_cost_leg_us / 1e6:.2f} "
                        f"CPU-seconds against the {cpus:.0f} CPUs this process may use"}
        if multi is not None:
            multi["per_gpu_value"] = out["value"] / world
            out["multi_gpu"] = multi
        rollout = None
        if world == 1 and not args.no_kernel_timing:
            rollout = rollout_step_roofline(B // max(sampler.n_groups, 1), T, B)
            out["roofline_rollout"] = rollout
        if ksum:
            out.update(roofline_objects(ksum, rollout, T, sampler.n_groups))
            out["kernels"] = {k: {kk: (round(vv, 3) if isinstance(vv, float) else vv)
                                  for kk, vv in v.items()} for k, v in ksum.items()}
        out["roofline_gae_scaled"] = gae_scaled_roofline()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(T, args.cpu_baseline_B, dict(step_cost_us=args.env_cost_us))
            if leg is not None and not args.no_cpu_baseline_leg:
                # like-for-like at the declared emulator cost: the same reference
                # iteration with the same busy wait per env step, beside value_env<cost>us
                key = f"value_env{int(args.env_cost_leg_us)}us"
                cb = cpu_baseline(T, args.cpu_baseline_B, dict(step_cost_us=args.env_cost_leg_us))
                out["cpu_baseline_env%dus" % int(args.env_cost_leg_us)] = {
                    k: cb[k] for k in ("value", "unit", "cores", "kind", "sample", "seconds")}
                out["ratio_env%dus" % int(args.env_cost_leg_us)] = {
                    "value": out[key] / cb["value"],
                    "note": f"{key} / cpu_baseline_env{int(args.env_cost_leg_us)}us.value: both sides pay "
                            f"{args.env_cost_leg_us:.0f} us of host CPU per env step (here spread over "
                            f"{workers} env worker processes, there inside the reference's SerialSampler "
                            "process, which is what north_star names as the baseline)"}
                out["ratio_env0us"] = {"value": out["value"] / out["cpu_baseline"]["value"],
                                       "note": "value / cpu_baseline.value (zero-cost synthetic env)"}
            # the isolated hot-path functions, HIP kernel beside the CPU restatement
            # on the same synthetic inputs (also yields the replay-kernel HBM rooflines)
            fn = isolated_functions(T, B)
            out["roofline_replay"] = fn.pop("roofline_replay")
            out["cpu_baseline"]["functions"] = fn
        if world == 1 and not args.no_extra_configs and (T, B) == (128, 256):
            out["configs"] = extra_configs(args)
        _canary_report(out)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def extra_configs(args, timeout_s=300):
    """BASELINE configs #3 (DQN) and #5 (R2D1) under the same clock as the headline (VERDICT r5 item
    6): each is a child ``python bench.py --config <c>`` run AFTER the PPO measurement is over (its
    sampler shut down, nothing of it timed any more), with that config's own default steps / warmup
    and a full-size replay ring wrapped before its timed region; the child's JSON line is cut down
    to the fields a reader needs.  A child that fails is reported as such -- it never takes the
    headline line with it."""
    import subprocess
    keep = ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "updates_per_s", "updates",
            "sampling_frac_of_step", "dtype", "roofline", "cpu_baseline", "last_loss")
    res = {}
    for cfg in ("dqn", "r2d1"):
        cmd = [sys.executable, os.path.abspath(__file__), "--config", cfg]
        if args.no_cpu_baseline:
            cmd.append("--no-cpu-baseline")
        t0 = time.perf_counter()
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s,
                               env=dict(os.environ))
            lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            if p.returncode != 0 or not lines:
                res[cfg] = {"error": f"exit code {p.returncode}", "stderr_tail": p.stderr[-600:]}
                continue
            doc = json.loads(lines[-1])
            obj = {k: doc[k] for k in keep if k in doc}
            obj["config"] = {k: doc["config"][k] for k in ("workload", "T", "B", "replay_frames",
                                                          "ring_wrapped", "updates_per_iteration",
                                                          "batch_size") if k in doc["config"]}
            obj["ms_per_time_step"] = doc.get("sampler", {}).get("ms_per_time_step")
            if isinstance(obj.get("cpu_baseline"), dict):
                obj["cpu_baseline"] = {k: v for k, v in obj["cpu_baseline"].items() if k != "kind_note"}
            obj["wall_s"] = round(time.perf_counter() - t0, 1)
            res[cfg] = obj
        except subprocess.TimeoutExpired:
            res[cfg] = {"error": f"no line within {timeout_s} s"}
        except (ValueError, KeyError, OSError) as e:
            res[cfg] = {"error": f"{type(e).__name__}: {e}"}
    res["note"] = ("child runs of `python bench.py --config dqn|r2d1` after the headline's timed region; "
                   "full lines: the same commands on their own (profiles/r*_bench_{dqn,r2d1}.json)")
    return res


# ---------------------------------------------------------------------------------------------
# roofline objects of the line: which kernel, priced how
# ---------------------------------------------------------------------------------------------
KTIMER_NOTE = ("HIP events around each launch, on the launching stream, in 2 iterations of the same "
               "workload run right after the timed region (the timed region itself carries no "
               "per-launch events: they would sit between its kernels)")
# bench region -> kernel names as rocprofv3 prints them (profiles/r*_bench_kernel_stats.csv)
ROCPROF_NAMES = {"convs_fwd": ["convs_fwd_fused_kernel"],
                 "conv2_bwd": ["conv2_bwd_x6_kernel"], "conv1_wgrad": ["conv1_wgrad_kernel"],
                 # (gemm_nt / gemm_nt_dgrad: since round 6 the input gradient is a launch of 256-row tiles
                 #  + a tail of 128-row tiles under the forward GEMM's kernel name -- the summary cannot
                 #  tell the two apart, so both regions are ranked by their live timings)
                 "gemm_tn": ["gemm_tn_x6_kernel", "gemm_reduce_slots_kernel"],
                 "ppo_head_loss": ["ppo_head_loss_kernel"],
                 "sample_convs_kernel": ["sample_convs_kernel"],
                 "rollout_fc_kernel": ["rollout_fc_kernel"],
                 "rollout_head_kernel<2>": ["rollout_head_kernel<2>"]}


def rocprof_kernel_stats():
    """{kernel name substring: average us} from the newest committed rocprofv3 --kernel-trace
    --stats summary of THIS command (profiles/r*_bench_kernel_stats.csv): the event-free,
    in-pipeline durations.  ({} when none is committed.)"""
    import csv
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_kernel_stats.csv")), reverse=True)
    if not paths:
        return {}, None
    rows = {}
    with open(paths[0]) as f:
        for r in csv.DictReader(f):
            rows[r["Name"]] = (float(r["AverageNs"]) * 1e-3, int(r["Calls"]))
    return rows, os.path.relpath(paths[0], ROOT)


def _rocprof_avg(rows, region):
    """Average us per launch of a bench region in the rocprof summary (sum over its kernels)."""
    tot = 0.
    for pat in ROCPROF_NAMES.get(region, []):
        hit = [v for k, v in rows.items() if pat in k]
        if not hit:
            return None
        tot += max(hit, key=lambda v: v[1])[0]
    return tot or None


def price_update_kernel(name, g, names, split):
    """Roofline object of one update-kernel region from its live HIP-event timing ``g``: an f32
    contraction issued as split[name] bf16 MFMAs per algorithmic MAC is priced against BOTH
    ceilings (HBM with its algorithmic bytes, the 2.5 PFLOP/s dense bf16 peak with its ISSUED flops),
    ``bound`` = the one it sits closer to; other flop-carrying regions against the f32 MFMA peak;
    the rest against HBM."""
    base = {"kernel": names.get(name, name), "avg_us": g["avg_us"], "avg_us_source": KTIMER_NOTE,
            "launches": g["launches"], "alg_bytes_per_launch": g["alg_bytes_per_launch"],
            "traffic": None}
    if name in split:
        nx = split[name]
        issued = g["TFLOPs"] * nx
        f_hbm, f_mfma = g["GBps"] / HBM_PEAK_GBPS, issued / BF16_MFMA_PEAK_TFLOPS
        hbm = f_hbm >= f_mfma
        base.update({"bound": "hbm" if hbm else "mfma", "achieved": g["GBps"] if hbm else issued,
                     "peak": HBM_PEAK_GBPS if hbm else BF16_MFMA_PEAK_TFLOPS,
                     "unit": "GB/s" if hbm else "TFLOP/s", "frac": max(f_hbm, f_mfma),
                     "frac_note": f"mfma side: ISSUED bf16 flops ({nx} MFMAs per algorithmic MAC) "
                                  f"over the dense bf16 peak = the fraction of the bf16x{nx} "
                                  f"emulation ceiling ({BF16_MFMA_PEAK_TFLOPS / nx:.0f} TFLOP/s "
                                  "algorithmic)",
                     "frac_hbm": f_hbm, f"frac_of_bf16x{nx}_ceiling": f_mfma,
                     "alg_fp32_TFLOPs": g["TFLOPs"],
                     "alg_over_f32_mfma_peak": g["TFLOPs"] / F32_MFMA_PEAK_TFLOPS,
                     "alg_flops_per_launch": g["alg_flops_per_launch"]})
    elif "TFLOPs" in g:
        base.update({"bound": "mfma", "achieved": g["TFLOPs"], "peak": F32_MFMA_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": g["TFLOPs"] / F32_MFMA_PEAK_TFLOPS,
                     "alg_flops_per_launch": g["alg_flops_per_launch"]})
    else:
        base.update({"bound": "hbm", "achieved": g["GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": g["GBps"] / HBM_PEAK_GBPS})
    tr = pmc_traffic(name, g)
    if tr is not None:
        base.update(traffic=tr["bytes_per_launch"], traffic_over_alg=tr["traffic_over_alg"],
                    traffic_source=tr["source"])
    return base


def roofline_objects(ksum, rollout, T, n_groups):
    """``roofline`` = the kernel with the largest TOTAL time over a whole timed iteration (rollout
    kernels included: T x n_groups + n_groups launches each per iteration, against 16 of every
    update kernel), ``roofline_update`` = the largest update kernel, ``roofline_gemm_tn`` = the
    weight gradient (largest matrix-pipe-side kernel).  Totals use the in-pipeline averages of the
    committed rocprofv3 summary of this command where there is one (event-free), else the live
    measurements; every object's own ``avg_us`` / ``achieved`` is the LIVE measurement of this run."""
    names, split = KERNEL_NAMES, BF16_SPLIT
    rows, csv_path = rocprof_kernel_stats()
    totals = {}
    for k, g in ksum.items():
        per_iter = g["launches"] / 2.                      # the timing leg runs 2 iterations
        us = _rocprof_avg(rows, k) or g["avg_us"]
        totals[k] = per_iter * us
    n_roll = T * n_groups + n_groups
    if rollout is not None:
        for k in ("sample_convs_kernel", "rollout_fc_kernel", "rollout_head_kernel<2>"):
            us = _rocprof_avg(rows, k) or rollout[k]["us_per_launch"]
            totals[k] = n_roll * us
    out = {}
    upd = max((k for k in ksum), key=lambda k: totals[k])
    out["roofline_update"] = price_update_kernel(upd, ksum[upd], names, split)
    top = max(totals, key=totals.get)
    if top in ksum:
        out["roofline"] = dict(out["roofline_update"])
    else:
        r = rollout[top]
        us_pipe = _rocprof_avg(rows, top)
        # priced on the IN-PIPELINE average (what the timed region runs: the kernel queued behind the
        # other pipeline groups' kernels) whenever a rocprofv3 summary of this command is committed;
        # the isolated graph-replay figure of this run is the side field
        us = us_pipe or r["us_per_launch"]
        tf = r["alg_flops"] / us * 1e-6
        out["roofline"] = {
            "kernel": top + " (rollout group-step: frame-stack rebuild + conv1 + conv2 of "
                            f"{rollout['Bg']} environments per launch, f32 MFMA)"
            if top == "sample_convs_kernel" else top,
            "bound": "mfma", "achieved": tf, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": tf / F32_MFMA_PEAK_TFLOPS,
            "frac_hbm": r["alg_bytes"] / us * 1e-3 / HBM_PEAK_GBPS,
            "avg_us": us,
            "avg_us_source": (f"in-pipeline average of the committed rocprofv3 --kernel-trace --stats "
                              f"summary of this command ({csv_path}): a captured graph node cannot "
                              "carry its own HIP events" if us_pipe else
                              "live, isolated (no rocprofv3 summary of this command is committed): "
                              + rollout["how"]),
            "avg_us_isolated": r["us_per_launch"], "frac_isolated": r["frac_f32_mfma_peak"],
            "avg_us_isolated_source": "live in this run: " + rollout["how"],
            "launches_per_iteration": n_roll, "alg_flops_per_launch": r["alg_flops"],
            "alg_bytes_per_launch": r["alg_bytes"], "traffic": r.get("traffic"),
            "traffic_over_alg": (round(r["traffic"] / r["alg_bytes"], 3) if r.get("traffic") else None),
            "traffic_source": r.get("traffic_source"),
            "note": "latency-class launch (64 environments): the empty-launch floor is "
                    f"{rollout['empty_launch_us']} us; alone it runs {r['us_per_launch']} us"}
        ro = out["roofline"]
        if ro["frac_hbm"] > ro["frac"]:
            # (the trunk kernel of the rollout step streams the 7 MB weight per launch: its byte side
            #  sits as near its roofline as its flop side; report the nearer one)
            ro.update(bound="hbm", achieved=r["alg_bytes"] / us * 1e-3, peak=HBM_PEAK_GBPS, unit="GB/s",
                      frac_mfma=ro["frac"], frac=ro["frac_hbm"])
    out["roofline"]["share_of_iteration_kernel_time"] = totals[top] / max(sum(totals.values()), 1e-9)
    out["roofline"]["selection"] = ("largest total kernel time per iteration; per-launch averages for the "
                                    "ranking from " + (csv_path or "this run's live timings"))
    if upd != "gemm_tn" and "gemm_tn" in ksum:
        out["roofline_gemm_tn"] = price_update_kernel("gemm_tn", ksum["gemm_tn"], names, split)
    if upd != "conv2_bwd" and "conv2_bwd" in ksum:
        out["roofline_conv2_bwd"] = price_update_kernel("conv2_bwd", ksum["conv2_bwd"], names, split)
    return out



def _rccl_version():
    try:
        v = torch.cuda.nccl.version()
        return ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
    except Exception as e:  # noqa: BLE001  (CPU-only build container)
        return f"unavailable ({type(e).__name__})"


def dry_run(args, rank, world, local_rank, cpus, workers, block):
    """Launch contract without a GPU: every rank joins the process group and reports the device,
    CPU block and env-worker count it WOULD use; rank 0 prints one JSON line whose ``n_gpus`` is
    the world size torch.distributed reports (tests/test_host_logic.py)."""
    import torch.distributed as dist
    backend = args.backend if torch.cuda.is_available() and not args.same_gpu else "gloo"
    info = dict(rank=rank, device=0 if args.same_gpu else local_rank,
                cpu_quota_share=round(cpus / world, 2), env_workers=workers,
                cpu_block=[block[0], block[-1]], pid=os.getpid(),
                # filled by a real run (per rank): see multi_gpu.ranks[] of the bench line
                rollout_ms=None, update_ms=None, ms_per_time_step=None,
                update_ms_ddp_sync=None, update_ms_no_sync=None, allreduce_exposed_ms=None)
    ranks = [info]
    if world > 1:
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
        ranks = [None] * world
        dist.all_gather_object(ranks, info)
        n = dist.get_world_size()
        dist.barrier()
        dist.destroy_process_group()
    else:
        n = 1
    if rank == 0:
        print(json.dumps({"metric": "env-steps/sec (SPS) whole node, PPO Atari [T=128,B=256]",
                          "dry_run": True, "value": None, "n_gpus": n, "backend": backend,
                          "rccl_version": _rccl_version(), "host_cpu_quota": cpus,
                          "ranks": ranks}), flush=True)


def cgroup_cpu_stat():
    """cpu.stat of this process's cgroup (v2), or None."""
    try:
        with open("/sys/fs/cgroup/cpu.stat") as f:
            t = f.read().split()
        return {t[i]: int(t[i + 1]) for i in range(0, len(t) - 1, 2)}
    except (OSError, ValueError):
        return None


def cgroup_cpu_delta(a, b, elapsed_s, marks):
    """What the box's CPU quota did to the timed region: CPU-seconds used, enforcement periods, periods in
    which the cgroup was THROTTLED (all of its threads -- env workers, the serve loop, the launching thread --
    frozen until the period ends) and the longest / median host-side iteration, so that a line whose value is
    low because the host was frozen for part of it can be told from one that is slow."""
    raw = [y - x for x, y in zip(marks[:-1], marks[1:])]
    gaps = sorted(raw)
    out = {"iteration_ms_median": round(gaps[len(gaps) // 2] * 1e3, 3) if gaps else None,
           "iteration_ms_max": round(gaps[-1] * 1e3, 3) if gaps else None,
           "iteration_ms": [round(g * 1e3, 1) for g in raw[:64]]}
    if a is None or b is None:
        out["cpu_stat"] = None
        return out
    d = {k: b[k] - a.get(k, 0) for k in b}
    out.update(cpu_seconds_used=round(d.get("usage_usec", 0) / 1e6, 3),
               cpus_busy_mean=round(d.get("usage_usec", 0) / 1e6 / elapsed_s, 2) if elapsed_s > 0 else None,
               periods=d.get("nr_periods"), throttled_periods=d.get("nr_throttled"),
               throttled_usec_sum_over_cpus=d.get("throttled_usec"))
    return out


def pmc_traffic(name, g):
    """HBM bytes per launch of the kernel(s) of bench region ``name`` from the newest
    profiles/r*pmc_counters.json that holds it ((2 * FETCH_SIZE + WRITE_SIZE) * 1024 per
    MI355X_MICROARCH.md; scripts/pmc_update.sh), rescaled by algorithmic bytes when the bench
    launch is not the profiled M = 8192 one."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*pmc_counters.json")), reverse=True):
        try:
            with open(path) as f:
                doc = json.load(f)
            k = doc["kernels"][name]
        except (OSError, KeyError, ValueError):
            continue
        scale = g["alg_bytes_per_launch"] / k["alg_bytes"] if k.get("alg_bytes") else 1.
        rel = os.path.relpath(path, ROOT)
        return {"bytes_per_launch": k["hbm_bytes_corrected"] * scale,
                "traffic_over_alg": k.get("traffic_over_alg"),
                "source": rel + (f" @ {doc['commit']}" if doc.get("commit") else "") + " (" +
                          (doc.get("note") or "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one pass each; "
                                              "scripts/pmc_update.sh").split(";")[0] + ")"}
    return None


def rollout_step_roofline(Bg=64, T=128, B=256):
    """The three kernels of a rollout group-step, each timed in isolation:
    20 back-to-back launches of ONE kernel captured in a hipGraph, replayed 30x, HIP events around
    the replays (steady-state time per launch including its dispatch, which is most of it at this
    size).  Algorithmic flops / bytes per launch for ``Bg`` environments:
      sample_convs: conv1 475 x 256 x 16 + conv2 108 x 256 x 32 MACs per env (5.66 MFLOP);
                    bytes = frame read 8.3 KB + previous stack 25 KB read + stack 33 KB write +
                    y2 13.8 KB write per env, + 50 KB of weights per workgroup-independent set;
      trunk:        3456 x 512 MACs per env; bytes = 7.08 MB of weights + x 13.8 KB per env +
                    27 partial slices x 2 KB per env written;
      head:         reads the partials back + 7 x 512 head weights, writes 8 floats per env.
    Priced against the fp32 MFMA peak and against HBM; both fractions are tiny BY DESIGN of the
    workload (64 rows per launch): these launches are latency-class, the number that matters is
    us per launch."""
    from rlpyt_amd import ops
    from rlpyt_amd.models.pg.atari_ff_model import AtariFfModel
    m = AtariFfModel((4, 104, 80), 6).cuda().eval()
    c1, c2 = m.conv.conv.conv[0], m.conv.conv.conv[2]
    lin = m._single_fc()
    obs = torch.randint(0, 256, (T, B, 4, 104, 80), dtype=torch.uint8, device="cuda")
    new_frame = torch.randint(0, 256, (Bg, 104, 80), dtype=torch.uint8, device="cuda")
    full_rows = torch.zeros((Bg, 4, 104, 80), dtype=torch.uint8, device="cuda")
    slot = torch.full((Bg,), -1, dtype=torch.int32, device="cuda")
    t_dev = torch.tensor([5], dtype=torch.int64, device="cuda")
    rew, dn = torch.zeros(T + 1, B, device="cuda"), torch.zeros(T + 1, B, dtype=torch.bool, device="cuda")
    rs, ds = torch.zeros(Bg, device="cuda"), torch.zeros(Bg, dtype=torch.bool, device="cuda")
    y2 = torch.empty((Bg, 3456), device="cuda")
    prob, value = torch.zeros((T, B, 6), device="cuda"), torch.zeros((T, B), device="cuda")
    action = torch.zeros((T + 1, B), dtype=torch.int64, device="cuda")
    action_out = torch.zeros(Bg, dtype=torch.int64, device="cuda")
    u = torch.rand(T, Bg, device="cuda")

    def timed(fn, n_inner=20, reps=30):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(n_inner):
                fn()
        g.replay()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            g.replay()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / (reps * n_inner) * 1e3

    def convs():
        ops.atari_sample_convs(obs, t_dev, 0, new_frame, full_rows, slot, c1.weight, c1.bias,
                               c2.weight, c2.bias, scalar_rows=(rew, rs, dn, ds), out=y2)
    part, ks = ops.rollout_fc_partials(y2, lin.weight)

    def trunk():
        ops.rollout_fc_partials(y2, lin.weight)

    def head():
        ops.rollout_head(part, ks, lin.bias, m.pi.weight, m.pi.bias, m.value.weight, m.value.bias,
                         u, t_dev, Bg, prob, value, action, 0, action_out)
    x = torch.zeros(64, device="cuda")
    floor_us = timed(lambda: x.add_(1))
    K, N = lin.weight.shape[1], lin.weight.shape[0]
    rows = {
        "sample_convs_kernel": (convs, Bg * 2 * (475 * 256 * 16 + 108 * 256 * 32),
                                Bg * (8320 + 24960 + 33280 + 3456 * 4) + 4 * (16 * 256 + 32 * 256)),
        "rollout_fc_kernel": (trunk, Bg * 2 * K * N, 4 * K * N + Bg * K * 4 + ks * Bg * N * 4),
        "rollout_head_kernel<2>": (head, Bg * 2 * N * 7, ks * Bg * N * 4 + 7 * N * 4 + Bg * 64),
    }
    out = {"Bg": Bg, "empty_launch_us": round(floor_us, 2),
           "how": "20 back-to-back launches of one kernel in a hipGraph, 30 replays, HIP events"}
    total = 0.
    for name, (fn, flops, nbytes) in rows.items():
        us = timed(fn)
        total += us
        out[name] = {"us_per_launch": round(us, 2), "alg_flops": flops, "alg_bytes": nbytes,
                     "TFLOPs": round(flops / us * 1e-6, 2), "GBps": round(nbytes / us * 1e-3, 1),
                     "frac_f32_mfma_peak": round(flops / us * 1e-6 / F32_MFMA_PEAK_TFLOPS, 4),
                     "frac_hbm_peak": round(nbytes / us * 1e-3 / HBM_PEAK_GBPS, 4),
                     "traffic": None}
    out["chain_us_per_group_step"] = round(total, 2)
    # HBM traffic from the PMC passes over scripts/step_microbench.py (scripts/rollout_pmc.sh), newest
    # committed file, (2 * FETCH_SIZE + WRITE_SIZE) * 1024 for all three kernels: the calibration probe of
    # round 6 (scripts/debug/fetch_probe.hip, profiles/r6_fetch_calibration.json) reads FETCH_SIZE at
    # exactly half the known bytes for EVERY read pattern of the path -- 16 B / lane streams, 4 B / lane
    # streams and the 64-byte row runs of rollout_fc's weight read alike -- and WRITE_SIZE at the bytes.
    try:
        import glob
        path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_rollout_pmc.json")))[-1]
        with open(path) as f:
            pmc = json.load(f)
        for name in rows:
            k = pmc.get("kernels", {}).get(name)
            if k:
                out[name]["traffic"] = k["hbm_bytes_corrected"]
                out[name]["traffic_over_alg"] = round(k["hbm_bytes_corrected"] / out[name]["alg_bytes"], 3)
                out[name]["traffic_source"] = (os.path.relpath(path, ROOT) + " (counter factors: "
                                               "profiles/r6_fetch_calibration.json)")
    except (OSError, ValueError, IndexError):
        pass
    return out


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
    # the segmented variant (time axis split over lanes, affine-map composition: re-associated,
    # within f32 rounding of the exact one) beside the exact scan the algorithms use
    res["gae"]["hip_us_segmented"] = round(_hip_us(lambda: ops.gae(
        r_d, v_d, d_d, bv_d, 0.99, 0.98, variant=ops.SCAN_SEGMENTED)), 2)
    res["gae"]["note"] = ("hip_us = RLPYT_SCAN_EXACT (bit-exact with the reference's association; the "
                          "product path), hip_us_segmented = RLPYT_SCAN_SEGMENTED; both latency-class "
                          "at [128, 256] (0.56 MB)")
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
    # ... and what a DQN batch actually issues: the agent and the target stack in one launch
    out_p = torch.empty((2, n, C, H, W), dtype=torch.uint8, device="cuda")
    both("frames_gather_pair", lambda: ops.frames_gather_pair(f_d, fd_d, ti_d, bi_d, C, 1, out=out_p),
         lambda: (O.frames_gather(f_n, fd_n, ti, bi, C), O.frames_gather(f_n, fd_n, ti + 1, bi, C)),
         n=n)
    replay["frames_gather_pair"] = {"bound": "hbm", "avg_us": res["frames_gather_pair"]["hip_us"],
                                    "alg_bytes_per_launch": 2 * nb,
                                    "achieved": 2 * nb / res["frames_gather_pair"]["hip_us"] / 1e3,
                                    "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                    "note": "agent + n-step target observation of one batch"}
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
    for k in ("frames_gather", "frames_gather_pair", "frames_gather_seq"):
        replay[k]["frac"] = replay[k]["achieved"] / HBM_PEAK_GBPS
    res["roofline_replay"] = replay
    return res


def cpu_baseline(T, B_cpu, env_kwargs):
    """The reference iteration on this box's host cores, ONE whole iteration at the bench batch
    [T, 256] (SerialSampler rollout + the reference's GAE loop + 16 updates).  ``kind: "reference"``
    = the unmodified reference from ``oracle/_ref`` (oracle/make_ref.py; present whenever the repo
    was built where /root/reference exists, and shipped with the snapshot); ``kind: "port"`` = the
    oracle's CPU port (oracle/ppo_cpu_port.py) when that copy is absent."""
    from oracle import ref_runner
    from oracle.ppo_cpu_port import calibrate_threads, time_cpu_baseline
    from rlpyt_amd.envs.synthetic import SyntheticPong
    from rlpyt_amd.utils.misc import usable_cpus
    quota = usable_cpus()
    if ref_runner.available():
        B = B_cpu if B_cpu > 0 else 64
        threads, _rate = calibrate_threads(max_threads=quota)
        res = ref_runner.time_ppo(SyntheticPong, env_kwargs, T=T, B=B, iters=1, threads=threads)
        return {"value": res["value"], "unit": "env-steps/s", "cores": res["cores"],
                "kind": "reference",
                "kind_note": "the UNMODIFIED reference (copy of /root/reference/rlpyt made by "
                             "oracle/make_ref.py at build time, git-ignored, shipped with the "
                             "snapshot): " + res["classes"] + ", driven by the statement sequence of "
                             "MinibatchRl.train (rlpyt/runners/minibatch_rl.py:253-262) on this "
                             "repo's SyntheticPong env, CPU tensors",
                "functions_kind": "numpy / torch-CPU oracle restatements (oracle/np_oracle.py) -- FASTER "
                                  "than the reference's own torch loops (GAE: 0.3 ms here vs 7.2 ms for "
                                  "rlpyt/algos/utils.py:24-40 on CPU tensors, SURVEY 8a), so the per-"
                                  "function ratios understate the gain over the reference",
                "sample": f"1 PPO iteration at [T={T}, B={B}] ({T * B} env steps, 16 minibatch "
                          f"updates), torch CPU with {res['cores']} threads (best fwd+bwd rate of a "
                          f"thread-count calibration within the {quota:.0f} CPUs this process may use, "
                          f"{os.cpu_count()} hardware threads on the box), {res['seconds']:.1f} s",
                "seconds": res["seconds"], "last_loss": res["last_loss"]}
    res = time_cpu_baseline(SyntheticPong, env_kwargs, T=T, B=B_cpu if B_cpu > 0 else None,
                            iters=1, threads=None, max_threads=quota)
    B_cpu = res["B"]
    return {"value": res["value"], "unit": "env-steps/s", "cores": res["cores"], "kind": "port",
            "kind_note": "oracle/_ref is absent on this box (the repo was not built where "
                         "/root/reference exists): CPU port of the reference iteration, pinned to the "
                         "reference's own PPO.optimize_agent run at 1e-5 (tests/test_oracle_golden.py); "
                         "timed beside the real reference at [128, 256]: port 432 vs reference 398 SPS",
            "sample": f"1 PPO iteration at [T={T}, B={B_cpu}] ({T * B_cpu} env steps, 16 "
                      f"minibatch updates), torch CPU with {res['cores']} threads, {res['seconds']:.1f} s",
            "seconds": res["seconds"]}


def cpu_baseline_replay(config, env_kwargs):
    """``cpu_baseline`` of the ``--config dqn|r2d1`` lines: the reference's own DQN / R2D1 iteration
    (SerialSampler + algorithm + agent + its host replay buffer) on this box's cores, bounded
    sample, reduced replay ring (stated).  None when ``oracle/_ref`` is absent."""
    from oracle import ref_runner
    from oracle.ppo_cpu_port import calibrate_threads
    from rlpyt_amd.envs.synthetic import SyntheticPong
    from rlpyt_amd.utils.misc import usable_cpus
    if not ref_runner.available():
        return {"value": None, "kind": "absent",
                "kind_note": "oracle/_ref (the reference copy made by oracle/make_ref.py) is not on "
                             "this box"}
    quota = usable_cpus()
    threads, _ = calibrate_threads(max_threads=quota)
    if config == "dqn":
        res = ref_runner.time_dqn(SyntheticPong, env_kwargs, iters=40, threads=threads)
        sample = (f"{res['iters']} DQN iterations of [2, 16] env steps + 2 updates of batch 128 after "
                  f"10 untimed ones; host replay ring reduced to {res['replay_frames']} frames "
                  "(the config's 1e6 = 8.3 GB; the ring size only sets the tree depth)")
    else:
        res = ref_runner.time_r2d1(SyntheticPong, env_kwargs, iters=1, threads=threads)
        sample = (f"{res['iters']} R2D1 iteration of [40, 192] env steps + {res['updates']} update(s) "
                  f"over 64 sequences of 125 steps, after 5 untimed sampling-only iterations; host "
                  f"replay ring reduced to {res['replay_frames']} frames (the config's 4e6 = 33 GB)")
    return {"value": res["value"], "unit": "env-steps/s", "updates_per_s": res["updates_per_s"],
            "cores": res["cores"], "kind": "reference", "kind_note": res["classes"] +
            " -- the unmodified reference from oracle/_ref on this repo's SyntheticPong env",
            "sample": sample + f", {res['seconds']:.1f} s", "seconds": res["seconds"]}


# ============================================================================================
# BASELINE configs #3 (DQN, prioritized 1 M-frame replay) and #5 (R2D2 / R2D1 sequence replay)
# ============================================================================================
def replay_config_main(args):
    """End-to-end iterations of the replay-based configs on one GPU: sampler -> append into the
    HBM ring -> {tree sample, frame / sequence gather, model, fused loss, priority update} x
    updates.  Hyper-parameters of rlpyt/experiments/configs/atari/dqn/atari_dqn.py
    ("prioritized": sampler [2, 16], batch 128, replay 1e6, replay_ratio 8) and atari_r2d1.py
    ("r2d1_long": sampler [40, 192], batch_T 80 + warmup 40, batch_B 64, n_step 5, replay 4e6)."""
    import numpy as np

    from rlpyt_amd import ops
    from rlpyt_amd.envs.synthetic import SyntheticPong
    from rlpyt_amd.samplers.collections import AtariTrajInfo
    from rlpyt_amd.samplers.gpu import GpuSampler
    from rlpyt_amd.utils import ktimer, logger
    from rlpyt_amd.utils.misc import usable_cpus
    from rlpyt_amd.utils.seed import set_seed
    logger.set_quiet(True)
    assert args.gpus == 1 and int(os.environ.get("WORLD_SIZE", 1)) == 1, \
        "--config dqn / r2d1 are single-GPU configs (BASELINE #3 / #5)"
    cpus = usable_cpus()
    set_seed(0)
    if args.config == "dqn":
        from rlpyt_amd.agents.dqn.dqn_agent import AtariDqnAgent
        from rlpyt_amd.algos.dqn.dqn import DQN
        T, B = 2, 16
        agent = AtariDqnAgent()
        algo = DQN(discount=0.99, batch_size=128, learning_rate=1e-4, clip_grad_norm=10.,
                   min_steps_learn=0, double_dqn=False, prioritized_replay=True, n_step_return=1,
                   replay_size=int(1e6))
        # sampling-only iterations before the timed region: by default until the 62 500-row ring
        # has WRAPPED (the timed sampling / updates then run on a full ring: every leaf of the
        # 1M-leaf tree live, frame windows crossing the wrap point)
        # (16 envs on 4 worker processes: 1486 / 1560 updates/s against 1404 / 1474 on 2, 1421 / 1515 on 8,
        #  1458 / 1516 on 16 -- interleaved on one box, profiles/r6_dqn_workers_sweep.jsonl)
        workers = args.workers if args.workers > 0 else 4
        fill = int(1e6) // B // T + 400 if args.replay_fill_itrs < 0 else args.replay_fill_itrs
        steps = args.steps if args.steps is not None else 300
        warmup = args.warmup if args.warmup is not None else 20
        name = "DQN AtariDqnAgent, PrioritizedReplayFrameBuffer 1e6 frames, sampler [2,16], batch 128"
    else:
        from rlpyt_amd.agents.dqn.r2d1_agent import AtariR2d1Agent
        from rlpyt_amd.algos.dqn.r2d1 import R2D1
        T, B = 40, 192
        agent = AtariR2d1Agent(eps_final=0.1, eps_final_min=0.0005)
        algo = R2D1(discount=0.997, batch_T=80, batch_B=64, warmup_T=40, store_rnn_state_interval=40,
                    replay_ratio=1, learning_rate=1e-4, clip_grad_norm=80., min_steps_learn=0,
                    double_dqn=True, prioritized_replay=True, n_step_return=5, pri_alpha=0.9,
                    pri_beta_init=0.6, pri_beta_final=0.6, input_priority_shift=2,
                    replay_size=int(4e6))
        workers = args.workers if args.workers > 0 else max(min(int(round(1.6 * cpus)) - 1, B // 10), 1)
        fill = int(4e6) // B // T + 8 if args.replay_fill_itrs < 0 else args.replay_fill_itrs   # wraps
        steps = args.steps if args.steps is not None else 20
        warmup = args.warmup if args.warmup is not None else 3
        name = ("R2D1 AtariR2d1Agent (conv + LSTM 512), PrioritizedSequenceReplayFrameBuffer 4e6 "
                "frames, sampler [40,192], sequences [40+80+5, 64]")
    sampler = GpuSampler(SyntheticPong, dict(step_cost_us=args.env_cost_us), batch_T=T, batch_B=B,
                         n_workers=workers, TrajInfoCls=AtariTrajInfo, max_decorrelation_steps=100,
                         mid_batch_reset=(args.config == "dqn"),
                         n_groups=((1 if args.config == "dqn" else None) if args.groups < 0
                                   else args.groups),
                         use_graph=not args.no_graph)
    # (dqn: ONE pipeline group for the 16 envs -- 958 vs 932 and 846 vs 828 updates/s against the
    # sampler's default of two, profiles/r5_ab_groups_replay.jsonl, r5_dqn_knobs.txt)
    examples = sampler.initialize(agent, seed=1, bootstrap_value=False)
    torch.cuda.set_device(0)
    agent.to_device(0)
    n_itr = fill + warmup + steps + 200
    algo.initialize(agent=agent, n_itr=n_itr, batch_spec=sampler.batch_spec,
                    mid_batch_reset=sampler.mid_batch_reset, examples=examples)
    rb = algo.replay_buffer
    learn_from = fill
    algo.min_itr_learn = learn_from            # sampling-only iterations fill the ring first

    def one(itr):
        agent.sample_mode(itr)
        samples, _ = sampler.obtain_samples(itr)
        agent.train_mode(itr)
        return algo.optimize_agent(itr, samples)

    for itr in range(fill + warmup):
        one(itr)
    torch.cuda.synchronize()
    for k in sampler.timing:
        sampler.timing[k] = 0.
    wt = getattr(getattr(sampler, "ctrl", None), "worker_timing", None)
    wt0 = None if wt is None else wt.copy()
    trace_marker(args)
    u0 = algo.update_counter
    t0 = time.perf_counter()
    t_sample = 0.
    for k in range(steps):
        itr = fill + warmup + k
        ts = time.perf_counter()
        agent.sample_mode(itr)
        samples, _ = sampler.obtain_samples(itr)
        t_sample += time.perf_counter() - ts
        agent.train_mode(itr)
        info = algo.optimize_agent(itr, samples)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    trace_marker(args)
    updates = algo.update_counter - u0
    # ---- phase-timing leg (see main()): iterations whose diagnostics are read back at once, so that the
    # host clock around the sampler does not include the wait for the previous iteration's updates
    ph_steps = max(1, min(steps, 200 if args.config == "dqn" else 5))
    for k in sampler.timing:
        sampler.timing[k] = 0.
    wt0 = None if wt is None else wt.copy()
    tp = time.perf_counter()
    t_sample = 0.
    for k in range(ph_steps):
        itr = fill + warmup + steps + k
        ts = time.perf_counter()
        agent.sample_mode(itr)
        samples, _ = sampler.obtain_samples(itr)
        t_sample += time.perf_counter() - ts
        agent.train_mode(itr)
        resolve_opt_info(algo.optimize_agent(itr, samples))
    torch.cuda.synchronize()
    ph_elapsed = time.perf_counter() - tp
    sampler_obj = sampler_stats(sampler, dict(sampler.timing), wt, wt0, ph_steps, T, t_sample,
                                not args.no_graph)
    sampler.shutdown()

    # ---- replay kernels on the REAL buffers (HIP events), at the config's batch shape ------------
    replay = {}
    frames = rb.samples_frames
    C = 4
    Tr = frames.shape[0] - (C - 1)
    Br = frames.shape[1]
    hw = frames.shape[2] * frames.shape[3]
    tree = rb.priority_tree
    filled_T = min(int(rb.t) if not getattr(rb, "_buffer_full", False) else Tr, Tr)
    g = torch.Generator().manual_seed(0)
    if args.config == "dqn":
        n = 128
        u = torch.rand(n, generator=g, dtype=torch.float64).cuda()
        us = _hip_us(lambda: tree.sample(u))
        replay["sumtree_sample"] = {"bound": "latency", "avg_us": round(us, 2),
                                    "ns_per_sample": round(us * 1e3 / n, 1),
                                    "levels": int(tree.tree_levels), "leaves": Tr * Br}
        ti = torch.randint(4, max(filled_T - 4, 8), (n,), generator=g).cuda()
        bi = torch.randint(0, Br, (n,), generator=g).cuda()
        out_d = torch.empty((n, C) + tuple(frames.shape[2:]), dtype=torch.uint8, device="cuda")
        us = _hip_us(lambda: ops.frames_gather(frames, rb.samples.done, ti, bi, C, out=out_d))
        nb = n * C * hw * 2
        replay["frames_gather"] = {"bound": "hbm", "avg_us": round(us, 2), "alg_bytes_per_launch": nb,
                                   "achieved": nb / us / 1e3, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                   "frac": nb / us / 1e3 / HBM_PEAK_GBPS,
                                   "note": "8.5 MB per launch: latency-class at this batch size"}
        # what sample_batch issues: agent + target stack in one launch
        out_p = torch.empty((2, n, C) + tuple(frames.shape[2:]), dtype=torch.uint8, device="cuda")
        us = _hip_us(lambda: ops.frames_gather_pair(frames, rb.samples.done, ti, bi, C,
                                                    rb.n_step_return, out=out_p))
        replay["frames_gather_pair"] = {"bound": "hbm", "avg_us": round(us, 2),
                                        "alg_bytes_per_launch": 2 * nb,
                                        "achieved": 2 * nb / us / 1e3, "peak": HBM_PEAK_GBPS,
                                        "unit": "GB/s", "frac": 2 * nb / us / 1e3 / HBM_PEAK_GBPS,
                                        "kernel": "frames_gather_kernel<B16> [2, 128, 4, 104, 80] "
                                                  "(agent + n-step target stacks of one batch)"}
        us = _hip_us(lambda: rb.sample_batch(n), iters=20)
        replay["sample_batch_total_us"] = round(us, 1)
    else:
        n, seq_T = 64, 125
        rsi = 40
        hi = max((filled_T - seq_T - 8) // rsi, 1)
        ti = (torch.randint(0, hi, (n,), generator=g) * rsi).cuda()
        bi = torch.randint(0, Br, (n,), generator=g).cuda()
        out_s = torch.empty((seq_T, n, C) + tuple(frames.shape[2:]), dtype=torch.uint8, device="cuda")
        us = _hip_us(lambda: ops.frames_gather_seq(frames, rb.samples.done, ti, bi, C, seq_T, out=out_s),
                     iters=20)
        nb = n * (seq_T + C - 1) * hw + seq_T * n * C * hw      # 334 MB
        replay["frames_gather_seq"] = {"bound": "hbm", "avg_us": round(us, 2),
                                       "alg_bytes_per_launch": nb, "achieved": nb / us / 1e3,
                                       "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                       "frac": nb / us / 1e3 / HBM_PEAK_GBPS,
                                       "kernel": "frames_gather_wide_kernel [125, 64, 4, 104, 80]"}
        u = torch.rand(n, generator=g, dtype=torch.float64).cuda()
        us = _hip_us(lambda: tree.sample(u))
        replay["sumtree_sample"] = {"bound": "latency", "avg_us": round(us, 2),
                                    "ns_per_sample": round(us * 1e3 / n, 1),
                                    "levels": int(tree.tree_levels)}
        us = _hip_us(lambda: rb.sample_batch(n), iters=10)
        replay["sample_batch_total_us"] = round(us, 1)
    key = "frames_gather_pair" if args.config == "dqn" else "frames_gather_seq"
    traffic = pmc_traffic(key, replay[key])
    out = {
        "metric": f"env-steps/sec (SPS), {args.config.upper()} Atari end to end, 1 GPU "
                  "(BASELINE config #%d)" % (3 if args.config == "dqn" else 5),
        "value": T * B * steps / elapsed, "unit": "env-steps/s", "n_gpus": 1, "steps": steps,
        "warmup": warmup, "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic (Atari-shaped SyntheticPong env on host cores, random-init model)",
        "config": {"workload": name, "T": T, "B": B, "env_workers": workers,
                   "env_step_cost_us": args.env_cost_us, "host_cpu_quota": cpus,
                   "replay_frames": int(Tr * Br), "replay_ring_T": int(Tr), "replay_B": int(Br),
                   "frame_store_GB": round(frames.numel() / 1e9, 2),
                   "tree_leaves": int(tree.T * tree.B), "tree_levels": int(tree.tree_levels),
                   "ring_rows_filled": int(filled_T), "fill_iterations": fill,
                   "ring_wrapped": bool(getattr(rb, "_buffer_full", False)),
                   "updates_per_iteration": algo.updates_per_optimize,
                   "batch_size": int(algo.batch_size)},
        "updates_per_s": updates / elapsed, "updates": updates,
        "sampling_frac_of_step": t_sample / ph_elapsed,
        "phase_timing": f"sampling_frac_of_step and the sampler object: {ph_steps} iterations after the "
                        "timed region with the diagnostics read back inside optimize_agent "
                        f"({ph_elapsed / ph_steps * 1e3:.3f} ms per iteration there); in the timed region "
                        "their copy to the host is left in flight (utils/deferred.py)",
        "sampler": sampler_obj,
        "roofline": dict(kernel=replay[key].get("kernel", key),
                         **{k: v for k, v in replay[key].items() if k != "kernel"},
                         traffic=None if traffic is None else traffic["bytes_per_launch"],
                         traffic_over_alg=None if traffic is None else traffic["traffic_over_alg"],
                         traffic_source=None if traffic is None else traffic["source"],
                         note="the replay kernel of this config (SURVEY 8(d)); where the iteration's "
                              "device time goes, kernel by kernel: profiles/r6_" + args.config +
                              "_region_final.txt (timed-region statistics of this command)"),
        "roofline_replay": replay,
        "last_loss": (info.loss[-1] if info.loss else None),
    }
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_replay(args.config, dict(step_cost_us=args.env_cost_us))
    _canary_report(out)
    print(json.dumps(out), flush=True)


def _canary_report(out):
    """RLPYT_CANARY=1 (debug): every device buffer of the package sits between 0xFF guard bands
    (rlpyt_amd/utils/canary.py); verify them once the run is over and say so in the line."""
    if os.environ.get("RLPYT_CANARY", "0") == "1":
        from rlpyt_amd.utils import canary
        n = canary.check("at the end of bench.py")
        out["canary"] = dict(canary.stats(), buffers_checked=n, result="clean")


if __name__ == "__main__":
    if os.environ.get("RLPYT_CANARY", "0") == "1":
        from rlpyt_amd.utils import canary as _c
        _c.enable()
    main()
