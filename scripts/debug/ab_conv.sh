#!/bin/bash
# On the GPU box: scripts/conv_bench.py entries for each conv.hip flag set (A/B of one kernel change).
# usage: ab_conv.sh <tag> <only-list> "<flags>" ...
TAG=$1; ONLY=$2; shift 2
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result"
OUT=gpurun_out/${TAG}_ab.log
: > $OUT
for rep in 1 2; do
for V in "$@"; do
  make -C rlpyt_amd/csrc conv.o CXXFLAGS="$FL $V" -B > /dev/null 2>&1 && make -C rlpyt_amd/csrc > /dev/null 2>&1
  echo "== [$V]" >> $OUT
  python scripts/conv_bench.py 8192 --no-model --only=$ONLY 2>/dev/null >> $OUT
done
done
make -C rlpyt_amd/csrc conv.o CXXFLAGS="$FL" -B > /dev/null 2>&1 && make -C rlpyt_amd/csrc > /dev/null 2>&1
cat $OUT
