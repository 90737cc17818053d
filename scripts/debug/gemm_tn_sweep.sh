#!/bin/bash
# On the GPU box: gemm_tn_x6_kernel timing (scripts/gemm_bench.py, wgrad row) for each flag set.
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result"
OUT=gpurun_out/${1:-tn}_sweep.log; shift
: > $OUT
for V in "$@"; do
  make -C rlpyt_amd/csrc gemm_pp.o CXXFLAGS="$FL $V" -B > /dev/null 2>&1 && make -C rlpyt_amd/csrc > /dev/null 2>&1
  echo "== $V" >> $OUT
  python scripts/gemm_bench.py 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print({k:v['us'] for k,v in d.items() if k.startswith('wgrad_tn')})" >> $OUT
done
make -C rlpyt_amd/csrc gemm_pp.o CXXFLAGS="$FL" -B > /dev/null 2>&1 && make -C rlpyt_amd/csrc > /dev/null 2>&1
cat $OUT
