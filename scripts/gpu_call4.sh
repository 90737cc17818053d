#!/bin/bash
# round-5 GPU call 4: pre-split trunk weight (W pieces made once per optimizer step) -- parity, isolated
# GEMM timings, interleaved A/B of the whole bench.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest "tests/test_conv_gpu.py" tests/test_variants.py tests/test_bench_path_gpu.py "tests/test_algo_parity.py::test_iterations_match_reference" "tests/test_algo_parity.py::test_iterations_match_reference_at_split_kernel_size" tests/test_sync_gpu.py -m gpu -q --maxfail=25 --timeout 420 -p no:cacheprovider -k "gemm or variant or bench or iterations or linear or ranks or run_to_run" > gpurun_out/r5_gpu_tests_4.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r5_gpu_tests_4.log
tail -6 gpurun_out/r5_gpu_tests_4.log
python scripts/gemm_bench.py > gpurun_out/r5_gemm_bench.json 2> gpurun_out/r5_gemm_bench.err; cat gpurun_out/r5_gemm_bench.json | cut -c1-1200
source scripts/ab_lib.sh
OUT=gpurun_out/r5_ab_presplit.jsonl; : > $OUT
run presplit_1
RLPYT_W_PRESPLIT=0 run insplit_1
run presplit_2
RLPYT_W_PRESPLIT=0 run insplit_2
cut -c1-200 $OUT
# per-kernel timing of both arms (kernel-timing leg on)
for arm in 1 0; do
RLPYT_W_PRESPLIT=$arm timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --env-cost-leg-us 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('W_PRESPLIT=$arm', 'SPS', round(d['value']), {k: round(v['avg_us'],1) for k,v in d['kernels'].items()})
"
done
