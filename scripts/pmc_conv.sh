#!/bin/bash
# rocprofv3 PMC passes over scripts/conv_bench.py (run ON THE GPU BOX via gpurun); counters in
# separate passes, never combined with other trace domains.  Output: gpurun_out/<tag>_pmc/*.csv
# usage: scripts/pmc_conv.sh <tag>
set -u
TAG=${1:-pmc}
OUT=$PWD/gpurun_out/${TAG}_pmc
mkdir -p $OUT
export TMPDIR=/tmp
run() {  # name counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python scripts/conv_bench.py 8192 --no-model > $OUT/$name.log 2>&1
  find $OUT/$name -name '*counter_collection.csv' -exec cp {} $OUT/$name.csv \;
  rm -rf $OUT/$name
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES
run sq3 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run fetch FETCH_SIZE
run write WRITE_SIZE
python scripts/pmc_summary.py $OUT > $OUT/summary.json
cat $OUT/summary.json
