#!/bin/bash
# Rollout experiments on the GPU box (DESIGN.md section 3a "leads"): SPS of the default PPO bench
# line under sampler variants and declared env costs.  One JSON line per run.
# usage: scripts/rollout_sweep.sh <out.jsonl>
OUT=${1:-gpurun_out/rollout_sweep.jsonl}
: > $OUT
run() {
  local tag="$1"; shift
  local line
  line=$(python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing --env-cost-leg-us 0 "$@" 2>/dev/null | tail -1)
  python - "$tag" "$line" >> $OUT <<'PY'
import json, sys
tag, line = sys.argv[1], sys.argv[2]
try:
    d = json.loads(line)
    s = d["sampler"]
    print(json.dumps(dict(tag=tag, sps=round(d["value"]), ms_per_step=round(d["ms_per_step"], 2),
                          sampling_frac=round(d["sampling_frac_of_step"], 3),
                          ms_per_time_step=round(s["ms_per_time_step"], 4),
                          wait_env_ms=round(s["master_wait_env_ms"], 4),
                          issue_ms=round(s["master_issue_ms"], 4),
                          wait_device_ms=round(s["master_wait_device_ms"], 4),
                          workers=d["config"]["env_workers_per_gpu"], groups=s["pipeline_groups"],
                          env_cost_us=d["config"]["env_step_cost_us"])))
except Exception as e:
    print(json.dumps(dict(tag=tag, error=str(e), raw=line[:200])))
PY
}
run default
RLPYT_ENVLOOP=0 run python_worker_loop           # the worker's per-step loop body in Python (round 2)
run default_again
RLPYT_SERVE_SPIN=0 run serve_spin0               # serve threads sleep instead of polling
RLPYT_SERVE_SPIN=2000 run serve_spin2000
run zero_copy_frames --zero-copy-frames                        # kernels read frames / write actions in the pinned step buffer
run groups3 --groups 3
run groups6 --groups 6
run workers16 --workers 16
run workers32 --workers 32
run env50us --env-cost-us 50
run env100us --env-cost-us 100
run env200us --env-cost-us 200
cat $OUT
