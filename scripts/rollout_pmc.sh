#!/bin/bash
# rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; one counter per pass, no other trace domain) over
# the isolated rollout-step kernels (scripts/step_microbench.py, Bg = 64) -> gpurun_out/<tag>/
# rollout_pmc.json (copied to profiles/rN_rollout_pmc.json; bench.py reads the newest for
# roofline_rollout.*.traffic).   usage: scripts/rollout_pmc.sh <tag>
set -u
TAG=${1:-rollout_pmc}
OUT=$PWD/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 200 rocprofv3 --pmc $c --output-format csv -d $OUT/$c -- python scripts/step_microbench.py 64 > $OUT/$c.log 2>&1
  find $OUT/$c -name '*counter_collection.csv' -exec cp {} $OUT/$c.csv \;
  rm -rf $OUT/$c
done
python - $OUT <<'PY' > $OUT/rollout_pmc.json
import csv, json, sys, os
from collections import defaultdict
d = sys.argv[1]
names = ["sample_convs_kernel", "rollout_fc_kernel", "rollout_head_kernel<2>"]
acc = defaultdict(lambda: defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    p = os.path.join(d, c + ".csv")
    if not os.path.exists(p):
        continue
    for row in csv.DictReader(open(p)):
        k = next((n for n in names if n.split("<")[0] in (row.get("Kernel_Name") or "")
                  and (("<" not in n) or n in row["Kernel_Name"].replace(" ", ""))), None)
        if k:
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {"note": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (one pass each) over scripts/step_microbench.py 64; "
               "hbm_bytes_corrected = (2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md; the x2 holds for every "
               "read pattern of these kernels, 64-byte row runs included (scripts/debug/fetch_probe.hip, "
               "profiles/r6_fetch_calibration.json: factor 1.99-2.00 on known byte counts, WRITE_SIZE factor 1.00); "
               "hbm_bytes_raw = (FETCH_SIZE + WRITE_SIZE)*1024 beside it; means over launches",
       "kernels": {}}
for k, v in acc.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        f, w = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"]), sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"])
        out["kernels"][k] = {"dispatches": len(v["FETCH_SIZE"]), "FETCH_SIZE_KB": round(f, 1),
                             "WRITE_SIZE_KB": round(w, 1), "hbm_bytes_corrected": int((2 * f + w) * 1024),
                             "hbm_bytes_raw": int((f + w) * 1024)}
print(json.dumps(out, indent=1))
PY
cat $OUT/rollout_pmc.json
