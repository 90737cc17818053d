#!/bin/bash
# R2D1 / DQN lines over the number of pipeline groups (ON THE GPU BOX)
OUT=$PWD/gpurun_out/r5_ab_groups_replay.jsonl; rm -f $OUT
for rep in 1 2; do
  for g in 4 2 1 3; do
    timeout 200 python bench.py --config r2d1 --replay-fill-itrs 60 --steps 15 --no-cpu-baseline --groups $g 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['sampler']
print(json.dumps(dict(cfg='r2d1', groups=$g, rep=$rep, sps=round(d['value']), ms_per_step=round(d['ms_per_step'],2), ms_per_time_step=round(s['ms_per_time_step'],4), chain=s['worker_ms_per_time_step']['chain_us'], wwait=round(s['worker_ms_per_time_step']['wait_mean'],4), wstep=round(s['worker_ms_per_time_step']['step_mean'],4))))" | tee -a $OUT
  done
done
for g in 2 1; do
  timeout 200 python bench.py --config dqn --replay-fill-itrs 3000 --no-cpu-baseline --groups $g 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['sampler']
print(json.dumps(dict(cfg='dqn', groups=$g, sps=round(d['value']), ms_per_step=round(d['ms_per_step'],3), updates_per_s=round(d['updates_per_s'],1), ms_per_time_step=round(s['ms_per_time_step'],4))))" | tee -a $OUT
done
