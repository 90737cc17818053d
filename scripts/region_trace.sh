#!/bin/bash
# rocprofv3 kernel trace of the TIMED REGION of the DQN / R2D1 / PPO bench lines (ON THE GPU BOX)
#   -> gpurun_out/<tag>_region_trace/{dqn,r2d1,ppo}_region.{txt,json}
# (a short fill keeps the trace small; the region's kernels do not depend on how full the ring is)
TAG=${1:-r5}; shift
CFGS=${@:-dqn r2d1 ppo}
OUT=$PWD/gpurun_out/${TAG}_region_trace
mkdir -p $OUT
export TMPDIR=/tmp
for cfg in $CFGS; do
  case $cfg in
    dqn)  ARGS="--config dqn --replay-fill-itrs 600 --steps 300 --warmup 20"; STEPS=300;;
    r2d1) ARGS="--config r2d1 --replay-fill-itrs 40 --steps 10 --warmup 3"; STEPS=10;;
    ppo)  ARGS="--steps 10 --warmup 3 --env-cost-leg-us 0 --no-kernel-timing --no-extra-configs"; STEPS=10;;
  esac
  rm -rf $OUT/raw_$cfg
  rocprofv3 --kernel-trace --output-format csv -d $OUT/raw_$cfg -- python bench.py $ARGS --no-cpu-baseline --trace-markers $EXTRA > $OUT/bench_$cfg.json 2> $OUT/prof_$cfg.log
  python scripts/trace_region.py "$(find $OUT/raw_$cfg -name '*kernel_trace.csv' | head -1)" --steps $STEPS --top 45 --json $OUT/${cfg}_region.json > $OUT/${cfg}_region.txt 2>&1
  rm -rf $OUT/raw_$cfg
  python -c "
import json,sys
d=json.loads(open('$OUT/bench_$cfg.json').read().strip().splitlines()[-1])
print('$cfg', 'SPS', round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'updates/s', d.get('updates_per_s'))"
  head -48 $OUT/${cfg}_region.txt | cut -c1-190
done
