#!/bin/bash
# On the GPU box: phase timing of the bf16x3 conv1 kernel (debug build), then the product build: tests + microbench.
TAG=${1:-x}
EXTRA=${2:-}
ONLY=${3:-conv1_wgrad}
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result $EXTRA"
make -C rlpyt_amd/csrc conv.o CXXFLAGS="$FL -DRLPYT_X3_TIMING" -B > /dev/null 2>&1 && make -C rlpyt_amd/csrc > /dev/null 2>&1
python scripts/debug/x3_timing.py 2>&1 | tail -10 > gpurun_out/${TAG}_timing.log
make -C rlpyt_amd/csrc conv.o CXXFLAGS="$FL" -B > /dev/null 2>&1 && make -C rlpyt_amd/csrc > /dev/null 2>&1
python -m pytest tests/test_conv_gpu.py -q -m gpu -x 2>&1 | tail -3 > gpurun_out/${TAG}_test.log
python scripts/conv_bench.py 8192 --no-model --only=$ONLY 2> /dev/null | tail -1 > gpurun_out/${TAG}_bench.json
python scripts/conv_bench.py 8192 --no-model --only=$ONLY 2> /dev/null | tail -1 >> gpurun_out/${TAG}_bench.json
cat gpurun_out/${TAG}_timing.log gpurun_out/${TAG}_test.log gpurun_out/${TAG}_bench.json
