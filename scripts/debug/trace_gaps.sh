#!/bin/bash
# On the GPU box: kernel trace of a short bench; per-minibatch busy / idle time of the update phase.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-gaps}
rm -rf $OUT; mkdir -p $OUT
timeout -k 5 240 rocprofv3 --kernel-trace --output-format csv -d $OUT/raw -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --env-cost-leg-us 0 > $OUT/bench.log 2>&1
f=$(find $OUT/raw -name '*kernel_trace.csv' | head -1)
python scripts/debug/trace_gaps.py "$f" | tee $OUT/gaps.txt
rm -rf $OUT/raw
