#!/bin/bash
# On the GPU box: rebuild conv.o with each flag set and time the named kernels.
# usage: scripts/debug/variant_sweep.sh <tag> <kernels,comma> "<flags A>" "<flags B>" ...
TAG=$1; ONLY=$2; shift 2
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result"
OUT=gpurun_out/${TAG}_sweep.log
: > $OUT
for V in "$@"; do
  make -C rlpyt_amd/csrc conv.o CXXFLAGS="$FL $V" -B > /dev/null 2>&1 && make -C rlpyt_amd/csrc > /dev/null 2>&1
  echo "== $V" >> $OUT
  python scripts/conv_bench.py 8192 --no-model --only=$ONLY 2> /dev/null | tail -1 >> $OUT
  python scripts/conv_bench.py 8192 --no-model --only=$ONLY 2> /dev/null | tail -1 >> $OUT
done
make -C rlpyt_amd/csrc conv.o CXXFLAGS="$FL" -B > /dev/null 2>&1 && make -C rlpyt_amd/csrc > /dev/null 2>&1
cat $OUT
