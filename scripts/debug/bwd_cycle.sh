#!/bin/bash
# On the GPU box: phase timing of conv2_bwd (debug build), then the product build: tests + microbench.
# usage: scripts/debug/bwd_cycle.sh <tag> ["-DVARIANT_FLAG ..."]   (extra flags apply to every build)
TAG=${1:-x}
EXTRA=${2:-}
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result $EXTRA"
make -C rlpyt_amd/csrc conv.o CXXFLAGS="$FL -DRLPYT_B2_TIMING" -B > /dev/null 2>&1 && make -C rlpyt_amd/csrc > /dev/null 2>&1
python scripts/debug/bwd_timing.py > gpurun_out/${TAG}_timing.log 2>&1
make -C rlpyt_amd/csrc conv.o CXXFLAGS="$FL" -B > /dev/null 2>&1 && make -C rlpyt_amd/csrc > /dev/null 2>&1
python -m pytest tests/test_conv_gpu.py -q -m gpu -x 2>&1 | tail -3 > gpurun_out/${TAG}_test.log
python scripts/conv_bench.py 8192 --no-model > gpurun_out/${TAG}_bench.json 2> /dev/null
python scripts/conv_bench.py 8192 --no-model >> gpurun_out/${TAG}_bench.json 2> /dev/null
