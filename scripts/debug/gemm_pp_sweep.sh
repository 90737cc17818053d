#!/bin/bash
# On the GPU box: the ping-pong GEMM with one phase left out at a time (timing builds), then the
# product build restored.  usage: gemm_pp_sweep.sh "<skip masks>" [extra -D flags]
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result"
OUT=gpurun_out/${3:-gemm_pp_sweep}.log
: > $OUT
for sk in ${1:-0 1 2 4 8}; do
  echo "=== PP_SKIP=$sk $2" | tee -a $OUT
  make -C rlpyt_amd/csrc gemm_pp.o CXXFLAGS="$FL -DRLPYT_TIMING -DPP_SKIP=$sk $2" -B > /dev/null 2>&1 && make -C rlpyt_amd/csrc > /dev/null 2>&1
  timeout 120 python scripts/debug/gemm_pp_timing.py 2>&1 | grep -v amdgpu | tee -a $OUT
done
make -C rlpyt_amd/csrc gemm_pp.o CXXFLAGS="$FL" -B > /dev/null 2>&1 && make -C rlpyt_amd/csrc > /dev/null 2>&1
