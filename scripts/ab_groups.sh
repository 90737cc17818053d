#!/bin/bash
# PPO bench: pipeline groups 2 / 3 / 4 / 5 interleaved (ON THE GPU BOX) -> gpurun_out/r5_ab_groups.jsonl
OUT=$PWD/gpurun_out/r5_ab_groups.jsonl; rm -f $OUT
for rep in 1 2; do
  for g in 4 3 2 5; do
    timeout 200 python bench.py --steps 12 --warmup 4 --groups $g --no-cpu-baseline --env-cost-leg-us 0 --no-kernel-timing 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['sampler']
print(json.dumps(dict(groups=$g, rep=$rep, sps=round(d['value']), ms_per_step=round(d['ms_per_step'],2), ms_per_time_step=round(s['ms_per_time_step'],4), wait_env=round(s['master_wait_env_ms'],4), wait_dev=round(s['master_wait_device_ms'],4), worker=s['worker_ms_per_time_step'])))" | tee -a $OUT
  done
done
