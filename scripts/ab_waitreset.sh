#!/bin/bash
# wait-reset collector's loop body in C: GPU tests + R2D1 line with RLPYT_ENVLOOP=0|1 (ON THE GPU BOX)
OUT=$PWD/gpurun_out/r5_waitreset
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_sampler_gpu_parity.py tests/test_dqn_gpu.py tests/test_variants.py -m gpu -q --timeout 300 -p no:cacheprovider > $OUT/tests.log 2>&1
echo "pytest rc=$?" >> $OUT/tests.log
tail -8 $OUT/tests.log
for rep in 1 2; do
  for v in 0 1; do
    RLPYT_ENVLOOP=$v timeout 300 python bench.py --config r2d1 --replay-fill-itrs 60 --steps 15 --no-cpu-baseline 2> $OUT/r2d1_${v}_${rep}.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps(dict(cfg='r2d1', envloop=$v, rep=$rep, sps=round(d['value']), ms_per_step=round(d['ms_per_step'],3), updates_per_s=round(d.get('updates_per_s') or 0,1), sampling_frac=round(d.get('sampling_frac_of_step',0),3))))" | tee -a $OUT/ab.jsonl
  done
done
