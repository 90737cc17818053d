#!/bin/bash
# On the GPU box: timing build of gemm_pp.hip, phase report, product build restored.
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result"
make -C rlpyt_amd/csrc gemm_pp.o CXXFLAGS="$FL -DRLPYT_TIMING" -B > /dev/null 2>&1 && make -C rlpyt_amd/csrc > /dev/null 2>&1
python scripts/debug/gemm_pp_timing.py 2>&1 | grep -v amdgpu | tee gpurun_out/${1:-gemm_pp_timing}.log
make -C rlpyt_amd/csrc gemm_pp.o CXXFLAGS="$FL" -B > /dev/null 2>&1 && make -C rlpyt_amd/csrc > /dev/null 2>&1
