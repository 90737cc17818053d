#!/bin/bash
# (ON THE GPU BOX) DQN config #3 against the number of env worker processes (16 envs, one pipeline group):
# interleaved, two rounds  -> gpurun_out/<tag>/sweep.jsonl
TAG=${1:-r6_dqn_workers}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; rm -f $OUT/sweep.jsonl
for rep in 1 2; do
  for w in ${WORKERS:-2 4 8 16}; do
    timeout 300 python bench.py --config dqn --replay-fill-itrs 3000 --no-cpu-baseline --workers $w 2> $OUT/w${w}_${rep}.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d.get('sampler') or {}; w=s.get('worker_ms_per_time_step') or {}
print(json.dumps(dict(workers=$w, rep=$rep, updates_per_s=round(d['updates_per_s'],1), ms_per_step=round(d['ms_per_step'],3), sampling_frac=round(d['sampling_frac_of_step'],3), ms_per_time_step=round(s.get('ms_per_time_step',0),4), env_step_ms=round(w.get('step_mean',0),4))))" | tee -a $OUT/sweep.jsonl
  done
done
