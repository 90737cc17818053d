#!/bin/bash
# round-5 GPU call 2: the new parity / captured-update / tail tests, native-tail A/B, DQN + R2D1 lines.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_sampler_gpu_parity.py tests/test_sampler_gpu.py tests/test_dqn_gpu.py "tests/test_algo_parity.py" "tests/test_conv_gpu.py::test_bf16_split_kernels_are_f32_accurate" -m gpu -q --maxfail=25 --timeout 420 -p no:cacheprovider > gpurun_out/r5_gpu_tests_2.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r5_gpu_tests_2.log
tail -8 gpurun_out/r5_gpu_tests_2.log
source scripts/ab_lib.sh
OUT=gpurun_out/r5_ab_tail.jsonl; : > $OUT
run tail_native_1
RLPYT_NATIVE_TAIL=0 run tail_python_1
run tail_native_2
RLPYT_NATIVE_TAIL=0 run tail_python_2
cut -c1-330 $OUT
timeout 400 python bench.py --config dqn > gpurun_out/r5_bench_dqn_1.json 2> gpurun_out/r5_bench_dqn_1.err
RLPYT_DQN_GRAPH=0 timeout 400 python bench.py --config dqn --no-cpu-baseline > gpurun_out/r5_bench_dqn_eager.json 2> gpurun_out/r5_bench_dqn_eager.err
timeout 500 python bench.py --config r2d1 > gpurun_out/r5_bench_r2d1_1.json 2> gpurun_out/r5_bench_r2d1_1.err
python - <<'PY'
import json
for f in ("r5_bench_dqn_1", "r5_bench_dqn_eager", "r5_bench_r2d1_1"):
    try:
        d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, "SPS", round(d["value"]), "ms/step", round(d["ms_per_step"], 3), "updates/s", round(d["updates_per_s"], 1),
              "sampling frac", round(d["sampling_frac_of_step"], 3), "cpu", (d.get("cpu_baseline") or {}).get("value"),
              (d.get("cpu_baseline") or {}).get("updates_per_s"))
    except Exception as e:
        print(f, "FAILED", e)
        print(open(f"gpurun_out/{f}.err").read()[-1500:])
PY
