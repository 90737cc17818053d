#!/bin/bash
# On the GPU box: phase timing of the bf16x3 kernel for each flag set.  usage: x3_sweep.sh <tag> "<flags>" ...
TAG=$1; shift
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result"
OUT=gpurun_out/${TAG}_sweep.log
: > $OUT
for V in "$@"; do
  make -C rlpyt_amd/csrc conv.o CXXFLAGS="$FL $V -DRLPYT_X3_TIMING" -B > /dev/null 2>&1 && make -C rlpyt_amd/csrc > /dev/null 2>&1
  echo "== $V" >> $OUT
  python scripts/debug/x3_timing.py 2>&1 | tail -8 >> $OUT
done
make -C rlpyt_amd/csrc conv.o CXXFLAGS="$FL" -B > /dev/null 2>&1 && make -C rlpyt_amd/csrc > /dev/null 2>&1
cat $OUT
