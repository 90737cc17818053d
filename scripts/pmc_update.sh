#!/bin/bash
# rocprofv3 PMC passes over the update's kernels at M = 8192 (ON THE GPU BOX): scripts/conv_bench.py
# and scripts/gemm_bench.py, one counter set per pass, no other trace domain.
# Output: gpurun_out/<tag>_pmc/{conv,gemm}_{fetch,write,sq}.csv + counters.json (the file
# bench.pmc_traffic() reads once copied to profiles/<tag>_pmc_counters.json).
# usage: scripts/pmc_update.sh <tag>
set -u
TAG=${1:-pmc}
OUT=$PWD/gpurun_out/${TAG}_pmc
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
run() {  # name program... -- counters...
  local name=$1; shift
  local prog=()
  while [ "$1" != "--" ]; do prog+=("$1"); shift; done
  shift
  timeout -k 5 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python "${prog[@]}" > $OUT/$name.log 2>&1
  find $OUT/$name -name '*counter_collection.csv' -exec cp {} $OUT/$name.csv \;
  rm -rf $OUT/$name
}
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
for p in conv gemm; do
  if [ $p = conv ]; then PROG="scripts/conv_bench.py 8192 --no-model"; else PROG="scripts/gemm_bench.py"; fi
  run ${p}_fetch $PROG -- FETCH_SIZE
  run ${p}_write $PROG -- WRITE_SIZE
  run ${p}_sq $PROG -- $SQ
done
python scripts/pmc_update_json.py $OUT > $OUT/counters.json
python -c "
import json; d = json.load(open('$OUT/counters.json'))
for k, v in d['kernels'].items(): print(k, v.get('dispatches'), 'traffic/alg', v.get('traffic_over_alg'), 'mfma_busy/gui', round(v.get('mfma_busy_cycles', 0) / max(v.get('gui_active', 1), 1) / 1024, 3))
"
