#!/bin/bash
# A/B of one environment switch on bench lines (ON THE GPU BOX): the switch 0 / 1 interleaved, two
# repetitions, short ring fill (the per-iteration work does not depend on how full the ring is).
#   usage: scripts/ab_switch.sh <ENV_VAR> "<configs>" [tag]     e.g.  RLPYT_DQN_CONVS "dqn r2d1"
#   switches of round 5: RLPYT_DQN_CONVS (own conv kernels in no-grad passes), RLPYT_LSTM_SEQ (one launch
#   per LSTM time step), RLPYT_R2D1_FUSED_STEP (reset handling folded into the sampling step),
#   RLPYT_Q_HEAD (split-K Q head), RLPYT_ENVLOOP (env workers' loop body in C), RLPYT_DQN_GRAPH
#   -> gpurun_out/<tag>/ab.jsonl   (records of this round: profiles/r5_ab_*.jsonl)
VAR=$1; CFGS=${2:-"dqn r2d1"}; TAG=${3:-ab_$1}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT; rm -f $OUT/ab.jsonl
for rep in 1 2; do
  for v in 0 1; do
    for cfg in $CFGS; do
      case $cfg in
        dqn)  A="--config dqn --replay-fill-itrs 3000";;
        r2d1) A="--config r2d1 --replay-fill-itrs 60 --steps 15";;
        ppo)  A="--steps 12 --warmup 4 --env-cost-leg-us 0 --no-kernel-timing --no-extra-configs";;
      esac
      env $VAR=$v timeout 300 python bench.py $A --no-cpu-baseline 2> $OUT/${cfg}_${v}_${rep}.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d.get('sampler') or {}
print(json.dumps(dict(cfg='$cfg', switch='$VAR', on=$v, rep=$rep, sps=round(d['value']), ms_per_step=round(d['ms_per_step'],3), updates_per_s=round(d.get('updates_per_s') or 0,1), sampling_frac=round(d.get('sampling_frac_of_step',0),3), ms_per_time_step=round(s.get('ms_per_time_step',0),4))))" | tee -a $OUT/ab.jsonl
    done
  done
done
