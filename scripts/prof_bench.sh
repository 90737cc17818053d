#!/bin/bash
# rocprofv3 kernel trace of the default bench (on the GPU box) -> per-kernel CSV summary.
# usage: scripts/prof_bench.sh <tag> [bench args...]
TAG=${1:-prof}; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${TAG}
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -- python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-extra-configs --env-cost-leg-us 0 "$@" > $OUT/bench.log 2>&1
f=$(find $OUT/raw -name '*kernel_stats.csv' | head -1)
cp "$f" $OUT/kernel_stats.csv 2>/dev/null
rm -rf $OUT/raw
head -40 $OUT/kernel_stats.csv | cut -c1-200
