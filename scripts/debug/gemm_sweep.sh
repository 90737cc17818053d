#!/bin/bash
# On the GPU box: rebuild gemm.o with each flag set and time the two trunk shapes.
# usage: scripts/debug/gemm_sweep.sh <tag> "<flags A>" "<flags B>" ...
TAG=$1; shift
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result"
OUT=gpurun_out/${TAG}_gemm_sweep.log
: > $OUT
for V in "$@"; do
  make -C rlpyt_amd/csrc gemm.o CXXFLAGS="$FL $V" -B > /dev/null 2>&1 && make -C rlpyt_amd/csrc > /dev/null 2>&1
  echo "== $V" >> $OUT
  python scripts/gemm_bench.py 2> /dev/null | tail -1 >> $OUT
done
make -C rlpyt_amd/csrc gemm.o CXXFLAGS="$FL" -B > /dev/null 2>&1 && make -C rlpyt_amd/csrc > /dev/null 2>&1
cat $OUT
