#!/bin/bash
# (ON THE GPU BOX) the default PPO line a few times per env-worker count, with what the CPU quota did to the
# TIMED REGION (bench.py: host_quota_in_timed_region): throttled periods, CPUs busy, longest iteration.
#   usage: scripts/debug/quota_runs.sh "<worker counts>" [reps=3]   -> gpurun_out/quota_runs.jsonl
WS=${1:-"20 16"}; REPS=${2:-3}
OUT=$PWD/gpurun_out/quota_runs.jsonl; rm -f $OUT
for rep in $(seq $REPS); do
  for w in $WS; do
    python bench.py --workers $w --no-cpu-baseline --no-extra-configs --env-cost-leg-us 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); q=d['host_quota_in_timed_region']; s=d['sampler']
print(json.dumps(dict(workers=$w, rep=$rep, sps=round(d['value']), ms_per_step=round(d['ms_per_step'],3), ms_per_time_step=round(s['ms_per_time_step'],4), wait_env=round(s['master_wait_env_ms'],4), wait_dev=round(s['master_wait_device_ms'],4), **q)))" | tee -a $OUT
  done
done
