echo "--- product build, phase_timing script"; timeout 120 python scripts/debug/phase_timing.py conv2_bwd_x6 8 2>&1 | tail -3
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result"
make -C rlpyt_amd/csrc conv.o CXXFLAGS="$FL -DRLPYT_TIMING" -B > /dev/null 2>&1 && make -C rlpyt_amd/csrc > /dev/null 2>&1
echo "--- timing build, pytest"; timeout 200 python -m pytest tests/test_conv_gpu.py -q -x -k "backward" 2>&1 | tail -3
echo "--- timing build, conv_bench"; timeout 100 python scripts/conv_bench.py 8192 --no-model --only=conv2_bwd_x6 2>&1 | tail -2
echo "--- timing build, phase_timing"; timeout 120 python scripts/debug/phase_timing.py conv2_bwd_x6 8 2>&1 | tail -10
