#!/bin/bash
# On the GPU box: phase timing of one conv kernel for each flag set.
# usage: phase_sweep.sh <tag> <kernel> <n_waves> "<flags>" ...
TAG=$1; KERN=$2; NW=$3; shift 3
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result"
OUT=gpurun_out/${TAG}_sweep.log
: > $OUT
for V in "$@"; do
  make -C rlpyt_amd/csrc conv.o CXXFLAGS="$FL $V -DRLPYT_TIMING" -B > /dev/null 2>&1 && make -C rlpyt_amd/csrc > /dev/null 2>&1
  echo "== $V" >> $OUT
  python scripts/debug/phase_timing.py $KERN $NW 2>&1 | tail -$((NW + 1)) >> $OUT
done
make -C rlpyt_amd/csrc conv.o CXXFLAGS="$FL" -B > /dev/null 2>&1 && make -C rlpyt_amd/csrc > /dev/null 2>&1
cat $OUT
