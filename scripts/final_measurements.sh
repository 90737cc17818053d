#!/bin/bash
# The measurements committed under profiles/r6_* (ON THE GPU BOX; ~20 min):
#   whole GPU suite, the three bench lines (ppo / dqn / r2d1), rocprofv3 kernel trace of the ppo bench,
#   PMC passes (FETCH_SIZE / WRITE_SIZE / SQ) over the update's kernels and the rollout-step kernels,
#   timed-region kernel statistics of the three lines.
# usage: [SKIP_TESTS=1] scripts/final_measurements.sh [tag=r6]      -> gpurun_out/<tag>_final/
TAG=${1:-r6}
OUT=$PWD/gpurun_out/${TAG}_final
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then      # (SKIP_TESTS=1: the suite was run on this tree in its own call)
timeout 1500 python -m pytest tests -m gpu -q --timeout 420 -p no:cacheprovider > $OUT/gpu_tests.log 2>&1
echo "pytest rc=$?" >> $OUT/gpu_tests.log
tail -4 $OUT/gpu_tests.log
fi
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_ppo.json 2> $OUT/bench_ppo.err
timeout 500 python bench.py --config dqn > $OUT/bench_dqn.json 2> $OUT/bench_dqn.err
timeout 600 python bench.py --config r2d1 > $OUT/bench_r2d1.json 2> $OUT/bench_r2d1.err
# event-free kernel durations of the same command (kernel trace only)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -- python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-extra-configs --env-cost-leg-us 0 > $OUT/prof_bench.log 2>&1
cp "$(find $OUT/raw -name '*kernel_stats.csv' | head -1)" $OUT/bench_kernel_stats.csv 2>/dev/null
rm -rf $OUT/raw
head -14 $OUT/bench_kernel_stats.csv | cut -c1-110
# PMC: counters in their own passes, no other trace domain
bash scripts/pmc_update.sh ${TAG}_final_upd > $OUT/pmc_update.log 2>&1
cp gpurun_out/${TAG}_final_upd_pmc/counters.json $OUT/pmc_counters.json 2>/dev/null
bash scripts/rollout_pmc.sh ${TAG}_final_roll > $OUT/pmc_rollout.log 2>&1
cp gpurun_out/${TAG}_final_roll/rollout_pmc.json $OUT/rollout_pmc.json 2>/dev/null
# kernel statistics of the TIMED REGION of each line (bench.py --trace-markers + scripts/trace_region.py)
bash scripts/region_trace.sh ${TAG}_final > $OUT/region_traces.log 2>&1
cp gpurun_out/${TAG}_final_region_trace/*_region.txt gpurun_out/${TAG}_final_region_trace/*_region.json $OUT/ 2>/dev/null
python - $OUT <<'PY'
import json, sys
out = sys.argv[1]
for f in ("bench_ppo", "bench_dqn", "bench_r2d1"):
    try:
        d = json.loads(open(f"{out}/{f}.json").read().strip().splitlines()[-1])
        cb = d.get("cpu_baseline") or {}
        print(f, "SPS", round(d["value"]), "ms/step", round(d["ms_per_step"], 3), "updates/s", d.get("updates_per_s"),
              "cpu", cb.get("kind"), cb.get("value"), "roofline", d["roofline"].get("kernel", "")[:30], d["roofline"].get("frac"))
    except Exception as e:
        print(f, "FAILED", e)
PY
