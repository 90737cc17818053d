#!/bin/bash
# A/B of one environment variable over arbitrary VALUES on bench lines (ON THE GPU BOX), values
# interleaved, two repetitions ("-" = variable unset).  Extra fixed settings: EXTRA="A=1 B=2".
#   usage: scripts/ab_env.sh <ENV_VAR> "<values>" "<configs>" [tag]   e.g.  GPU_MAX_HW_QUEUES "- 8" "ppo r2d1"
#   -> gpurun_out/<tag>/ab.jsonl
VAR=$1; VALS=$2; CFGS=${3:-"ppo"}; TAG=${4:-ab_$1}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT; rm -f $OUT/ab.jsonl
for rep in 1 2; do
  for v in $VALS; do
    for cfg in $CFGS; do
      case $cfg in
        dqn)  A="--config dqn --replay-fill-itrs 3000";;
        r2d1) A="--config r2d1 --replay-fill-itrs 60 --steps 15";;
        ppo)  A="--steps 12 --warmup 4 --env-cost-leg-us 0 --no-kernel-timing --no-extra-configs";;
      esac
      if [ "$v" = "-" ]; then SET="-u $VAR"; else SET="$VAR=$v"; fi
      env $SET $EXTRA timeout 300 python bench.py $A --no-cpu-baseline 2> $OUT/${cfg}_${v}_${rep}.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d.get('sampler') or {}
print(json.dumps(dict(cfg='$cfg', var='$VAR', value='$v', extra='$EXTRA', rep=$rep, sps=round(d['value']), ms_per_step=round(d['ms_per_step'],3), updates_per_s=round(d.get('updates_per_s') or 0,1), sampling_frac=round(d.get('sampling_frac_of_step',0),3), ms_per_time_step=round(s.get('ms_per_time_step',0),4))))" | tee -a $OUT/ab.jsonl
    done
  done
done
