#!/bin/bash
# rocprofv3 PMC passes over scripts/gemm_bench.py (on the GPU box); separate passes, no other trace domain.
set -u
TAG=${1:-g}
OUT=$PWD/gpurun_out/${TAG}_pmc
mkdir -p $OUT
export TMPDIR=/tmp
run() {
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python scripts/gemm_bench.py > $OUT/$name.log 2>&1
  find $OUT/$name -name '*counter_collection.csv' -exec cp {} $OUT/$name.csv \;
  rm -rf $OUT/$name
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU
run sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM
run fetch FETCH_SIZE
run write WRITE_SIZE
python - "$OUT" <<'PY'
import csv, glob, os, sys, json
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for path in sorted(glob.glob(os.path.join(sys.argv[1], "*.csv"))):
    for row in csv.DictReader(open(path)):
        name = (row.get("Kernel_Name") or "")
        if "gemm_nt_x6" in name: key = "gemm_nt_x6"
        elif name.startswith("Cijk"): key = name[:40]
        else: continue
        acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {k: {c: round(sum(v) / len(v), 1) for c, v in d.items()} for k, d in acc.items()}
print(json.dumps(out, indent=1, sort_keys=True))
PY
