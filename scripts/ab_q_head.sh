#!/bin/bash
# tests + DQN / R2D1 lines with the fused Q head on / off (ON THE GPU BOX)
OUT=$PWD/gpurun_out/r5_q_head
rm -rf $OUT; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_dqn_gpu.py tests/test_dqn_convs_gpu.py tests/test_lstm_seq_gpu.py tests/test_variants.py tests/test_algo_parity.py tests/test_sampler_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider > $OUT/tests.log 2>&1
echo "pytest rc=$?" >> $OUT/tests.log
tail -12 $OUT/tests.log
for rep in 1 2; do
  for v in 0 1; do
    for cfg in dqn r2d1; do
      if [ $cfg = dqn ]; then A="--replay-fill-itrs 3000"; else A="--replay-fill-itrs 60 --steps 15"; fi
      RLPYT_Q_HEAD=$v timeout 300 python bench.py --config $cfg $A --no-cpu-baseline 2> $OUT/${cfg}_${v}_${rep}.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps(dict(cfg='$cfg', q_head=$v, rep=$rep, sps=round(d['value']), ms_per_step=round(d['ms_per_step'],3), updates_per_s=round(d.get('updates_per_s') or 0,1), sampling_frac=round(d.get('sampling_frac_of_step',0),3))))" | tee -a $OUT/ab.jsonl
    done
  done
done
