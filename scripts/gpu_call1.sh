#!/bin/bash
# round-5 GPU call 1: whole GPU suite after the prune / sampler split / sign mask / stream-write
# completion, then interleaved A/B of the completion marker, then one full bench line.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --maxfail=25 --timeout 420 -p no:cacheprovider > gpurun_out/r5_gpu_tests_1.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r5_gpu_tests_1.log
tail -5 gpurun_out/r5_gpu_tests_1.log
source scripts/ab_lib.sh
OUT=gpurun_out/r5_ab_complete.jsonl; : > $OUT
run writevalue_1
RLPYT_SERVE_EVENT=1 run event_1
run writevalue_2
RLPYT_SERVE_EVENT=1 run event_2
cat $OUT
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r5_bench_ppo_1.json 2> gpurun_out/r5_bench_ppo_1.err
tail -c 1500 gpurun_out/r5_bench_ppo_1.json | head -c 600; echo
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5_bench_ppo_1.json").read().strip().splitlines()[-1])
print("SPS", round(d["value"]), "ms/step", round(d["ms_per_step"], 2), "time step ms", round(d["sampler"]["ms_per_time_step"], 4))
print("roofline", d["roofline"]["kernel"][:40], round(d["roofline"]["frac"], 3))
for k, v in d["kernels"].items():
    print(k, v["launches"], round(v["avg_us"], 1))
print("cpu_baseline", d["cpu_baseline"]["kind"], round(d["cpu_baseline"]["value"]), d["cpu_baseline"]["cores"])
PY
