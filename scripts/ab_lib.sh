#!/bin/bash
# Interleaved A/B runs of the default bench on ONE box: `source scripts/ab_lib.sh; OUT=...jsonl;
# [ENV=..] run <tag> [bench args]` appends one summary line per run (SPS, ms per time step, the
# sampler's per-batch phases, worker wait / step split).
run() {
  local tag="$1"; shift
  local line
  line=$(python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-timing --env-cost-leg-us 0 "$@" 2>gpurun_out/ab_${tag}.err | tail -1)
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
                          per_batch_ms={k: round(v, 3) for k, v in s.get("per_batch_ms", {}).items()},
                          worker_ms={k: (round(v, 4) if not isinstance(v, dict) else {a: round(b, 1) for a, b in v.items()}) for k, v in (s.get("worker_ms_per_time_step") or {}).items()},
                          workers=d["config"]["env_workers_per_gpu"], groups=s["pipeline_groups"],
                          env_cost_us=d["config"]["env_step_cost_us"])))
except Exception as e:
    print(json.dumps(dict(tag=tag, error=str(e), raw=line[:200])))
PY
}
