#!/bin/bash
# A/B (ON THE GPU BOX): R2D1's training pass of the LSTM on the own kernels (1) / the library RNN (0)
OUT=$PWD/gpurun_out/${1:-r6q_ab_lstm_train}; mkdir -p $OUT; rm -f $OUT/ab.jsonl
for rep in 1 2; do for v in 0 1; do
RLPYT_LSTM_SEQ_TRAIN=$v timeout 300 python bench.py --config r2d1 --replay-fill-itrs 60 --steps 15 --no-cpu-baseline 2> $OUT/r2d1_${v}_${rep}.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps(dict(cfg='r2d1', own_lstm_train=$v, rep=$rep, sps=round(d['value']), ms_per_step=round(d['ms_per_step'],3), updates_per_s=round(d.get('updates_per_s') or 0,2), sampling_frac=round(d.get('sampling_frac_of_step',0),3), loss=d.get('last_loss'))))" | tee -a $OUT/ab.jsonl
done; done
