#!/bin/bash
# Bench lines over the number of rollout pipeline groups (ON THE GPU BOX), interleaved, two repetitions.
#   usage: scripts/ab_groups.sh <ppo|dqn|r2d1> "<group counts>"      -> gpurun_out/ab_groups_<cfg>.jsonl
#   (records of round 5: profiles/r5_ab_groups.jsonl, r5_ab_groups_replay.jsonl)
CFG=${1:-ppo}; GROUPS_=${2:-"4 3 2 5"}
OUT=$PWD/gpurun_out/ab_groups_$CFG.jsonl; rm -f $OUT
case $CFG in
  dqn)  A="--config dqn --replay-fill-itrs 3000";;
  r2d1) A="--config r2d1 --replay-fill-itrs 60 --steps 15";;
  ppo)  A="--steps 12 --warmup 4 --env-cost-leg-us 0 --no-kernel-timing";;
esac
for rep in 1 2; do
  for g in $GROUPS_; do
    timeout 200 python bench.py $A --groups $g --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['sampler']; w=s['worker_ms_per_time_step'] or {}
print(json.dumps(dict(cfg='$CFG', groups=$g, rep=$rep, sps=round(d['value']), ms_per_step=round(d['ms_per_step'],3), ms_per_time_step=round(s['ms_per_time_step'],4), chain_us=w.get('chain_us'), worker_wait=round(w.get('wait_mean',0),4), worker_step=round(w.get('step_mean',0),4))))" | tee -a $OUT
  done
done
