#!/bin/bash
# round-5 GPU call 3: fixed parity tests, the fused replay field gather, DQN knobs A/B.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_sampler_gpu_parity.py "tests/test_algo_parity.py" tests/test_variants.py tests/test_hip_parity.py tests/test_dqn_gpu.py -m gpu -q --maxfail=25 --timeout 420 -p no:cacheprovider > gpurun_out/r5_gpu_tests_3.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r5_gpu_tests_3.log
tail -6 gpurun_out/r5_gpu_tests_3.log
dq() {  # tag, env..., args...
  local tag=$1; shift
  timeout 300 python bench.py --config dqn --no-cpu-baseline "$@" > gpurun_out/r5_dqn_$tag.json 2> gpurun_out/r5_dqn_$tag.err
  python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/r5_dqn_{tag}.json").read().strip().splitlines()[-1])
    print(tag, "SPS", round(d["value"]), "ms/iter", round(d["ms_per_step"], 3), "updates/s", round(d["updates_per_s"], 1),
          "sampling frac", round(d["sampling_frac_of_step"], 3), "sample_batch_us", d["roofline_replay"].get("sample_batch_total_us"))
except Exception as e:
    print(tag, "FAILED", e); print(open(f"gpurun_out/r5_dqn_{tag}.err").read()[-800:])
PY
}
dq fused
RLPYT_FUSED_FIELDS=0 dq rowwise
dq groups1 --groups 1
dq groups1_w1 --groups 1 --workers 1
