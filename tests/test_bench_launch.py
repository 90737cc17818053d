"""`python bench.py --gpus N` launches its own N ranks (one process per GPU as
rlpyt/runners/sync_rl.py:60-101): the launch contract on CPU (``--dry-run``: ranks join the
process group and report, no GPU work) and, on the GPU box, one real step of two ranks sharing
cuda:0 over gloo."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)       # the point: no launcher around bench.py
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bench_gpus2_launches_two_ranks_without_torchrun():
    r, line = _run(["--gpus", "2", "--dry-run"], 300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert line["n_gpus"] == 2 and line["dry_run"] is True
    ranks = sorted(line["ranks"], key=lambda x: x["rank"])
    assert [x["rank"] for x in ranks] == [0, 1]
    assert [x["device"] for x in ranks] == [0, 1]            # rank r <-> GPU r
    assert ranks[0]["pid"] != ranks[1]["pid"]                  # one process per rank
    # env workers are sized from the QUOTA share of each rank, and the ranks' CPU blocks are disjoint
    for x in ranks:
        assert x["env_workers"] <= max(int(round(1.6 * x["cpu_quota_share"])) - 1, 1)
    if ranks[0]["cpu_block"] != ranks[1]["cpu_block"]:
        assert ranks[0]["cpu_block"][1] < ranks[1]["cpu_block"][0]


def test_bench_rejects_mismatched_world_size():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1 but --gpus 2" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_bench_refuses_more_gpus_than_devices():
    import torch
    n = torch.cuda.device_count()
    r, line = _run(["--gpus", str(n + 1), "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], 300)
    assert r.returncode == 2 and line is None
    assert "refusing to run fewer ranks" in r.stderr


@pytest.mark.gpu
def test_bench_gpus2_same_gpu_runs_two_ranks():
    """The real bench step with two self-launched ranks on cuda:0 (gloo): the line reports the
    world size torch.distributed saw, and DDP kept the ranks' parameters identical."""
    r, line = _run(["--gpus", "2", "--same-gpu", "--backend", "gloo", "--steps", "1", "--warmup", "1",
                    "--no-cpu-baseline", "--env-cost-leg-us", "0", "--check-params",
                    "--batch-T", "32", "--batch-B", "128"], 900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert line["n_gpus"] == 2 and line["multi_gpu"]["dist_world_size"] == 2
    assert line["config"]["parallelism"] == "dp2"
    assert "bit-identical parameters" in r.stderr
    assert len({x["rank"] for x in line["multi_gpu"]["ranks"]}) == 2


def test_pmc_traffic_reads_the_committed_counters():
    """``bench.pmc_traffic`` (the ``roofline.traffic`` of the driver line) finds every region of the
    update and both replay gathers in the newest committed ``profiles/r*pmc_counters.json``, rescales
    by algorithmic bytes and names the file (+ its commit stamp when it has one) in its source string."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    gemm_alg = 4 * (8192 * 3456 + 512 * 3456 + 8192 * 512)
    for name, alg in (("conv1_fwd", 8192 * (33280 + 4 * 7600)), ("conv2_fwd", 8192 * 4 * (7600 + 3456)),
                      ("conv2_bwd", 8192 * 4 * (2 * 3456 + 2 * 7600)), ("conv1_wgrad", 8192 * (33280 + 4 * 7600)),
                      ("gemm_nt", gemm_alg), ("gemm_nt_dgrad", gemm_alg), ("gemm_tn", gemm_alg),
                      ("frames_gather_pair", 2 * 128 * 4 * 8320 * 2),
                      ("frames_gather_seq", 64 * 128 * 8320 + 125 * 64 * 4 * 8320)):
        t = bench.pmc_traffic(name, {"alg_bytes_per_launch": alg})
        assert t is not None, name
        assert 0.9 * alg < t["bytes_per_launch"] < 3.0 * alg, (name, t)
        assert "profiles/r" in t["source"] and "rocprofv3 --pmc" in t["source"], t["source"]
        half = bench.pmc_traffic(name, {"alg_bytes_per_launch": alg // 2})
        assert abs(half["bytes_per_launch"] * 2 - t["bytes_per_launch"]) <= 2 + 1e-6 * t["bytes_per_launch"]
    assert bench.pmc_traffic("no_such_region", {"alg_bytes_per_launch": 1}) is None


def test_cgroup_cpu_delta_reports_throttling_and_iteration_spread():
    """``bench.cgroup_cpu_delta`` (the line's ``host_quota_in_timed_region``): CPU-seconds, enforcement and
    throttled periods between two ``cpu.stat`` readings, median / longest host-side iteration; without a
    cgroup-v2 ``cpu.stat`` the iteration times are still there."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    a = dict(usage_usec=1_000_000, nr_periods=10, nr_throttled=1, throttled_usec=5_000)
    b = dict(usage_usec=9_000_000, nr_periods=17, nr_throttled=3, throttled_usec=45_000)
    marks = [0.0, 0.030, 0.070, 0.100, 0.160]
    d = bench.cgroup_cpu_delta(a, b, 0.8, marks)
    assert d["cpu_seconds_used"] == 8.0 and d["cpus_busy_mean"] == 10.0
    assert d["periods"] == 7 and d["throttled_periods"] == 2 and d["throttled_usec_sum_over_cpus"] == 40_000
    assert d["iteration_ms"] == [30.0, 40.0, 30.0, 60.0]
    assert d["iteration_ms_max"] == 60.0 and d["iteration_ms_median"] == 40.0
    d0 = bench.cgroup_cpu_delta(None, b, 0.8, marks)
    assert d0["cpu_stat"] is None and d0["iteration_ms_max"] == 60.0
    st = bench.cgroup_cpu_stat()
    assert st is None or "usage_usec" in st


def test_trace_region_cuts_a_kernel_trace_to_the_marked_region(tmp_path):
    """scripts/trace_region.py: dispatches between the two ``erfinv`` markers of ``bench.py
    --trace-markers``, per-kernel statistics and the device-busy share (union of the intervals)."""
    import csv
    import json
    import subprocess
    import sys
    trace = tmp_path / "kernel_trace.csv"
    rows = [("warmup_kernel", 0, 10_000), ("void at::native::erfinv_kernel", 20_000, 21_000),
            ("b_kernel", 40_000, 50_000), ("c_kernel", 45_000, 70_000), ("b_kernel", 80_000, 90_000),
            ("void at::native::erfinv_kernel", 121_000, 122_000), ("after_kernel", 130_000, 140_000)]
    with open(trace, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_ALL)
        w.writerow(["Kind", "Kernel_Name", "Start_Timestamp", "End_Timestamp"])
        for n, s, e in rows:
            w.writerow(["KERNEL_DISPATCH", n, s, e])
    out = tmp_path / "region.json"
    script = os.path.join(ROOT, "scripts", "trace_region.py")
    r = subprocess.run([sys.executable, script, str(trace), "--steps", "2", "--json", str(out)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    d = json.loads(out.read_text())
    assert d["dispatches"] == 3 and abs(d["region_wall_ms"] - 0.1) < 1e-9
    assert abs(d["device_busy_ms"] - 0.04) < 1e-9 and abs(d["kernel_sum_ms"] - 0.045) < 1e-9
    names = {k["name"]: k for k in d["kernels"]}
    assert set(names) == {"b_kernel", "c_kernel"} and names["b_kernel"]["calls"] == 2
    assert names["b_kernel"]["calls_per_step"] == 1.0
    # no markers: a clear message instead of statistics of the whole run
    rows2 = tmp_path / "nomark.csv"
    rows2.write_text('"Kind","Kernel_Name","Start_Timestamp","End_Timestamp"\\n"K","a",0,10\\n')
    r = subprocess.run([sys.executable, script, str(rows2)], capture_output=True, text=True)
    assert r.returncode != 0 and "trace-markers" in (r.stderr + r.stdout)
