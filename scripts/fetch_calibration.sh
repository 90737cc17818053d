#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration (ON THE GPU BOX): builds scripts/debug/fetch_probe.hip, runs it under
# rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (one counter per pass, no trace domain) and writes
# gpurun_out/<tag>/fetch_calibration.json = counter KB per launch against the bytes each probe kernel is
# KNOWN to touch, per access pattern (copy to profiles/rN_fetch_calibration.json; bench.py quotes the
# rollout_fc traffic with the factor this file supports).     usage: scripts/fetch_calibration.sh <tag>
set -u
TAG=${1:-fetch_calibration}
OUT=$PWD/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/debug/fetch_probe.hip -o $OUT/fetch_probe || exit 1
$OUT/fetch_probe 5 > $OUT/probe.json || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 200 rocprofv3 --pmc $c --output-format csv -d $OUT/$c -- $OUT/fetch_probe 5 > $OUT/$c.log 2>&1
  find $OUT/$c -name '*counter_collection.csv' -exec cp {} $OUT/$c.csv \;
  rm -rf $OUT/$c
done
rm -f $OUT/fetch_probe
python - $OUT <<'PY' > $OUT/fetch_calibration.json
import csv, json, os, sys
from collections import defaultdict
d = sys.argv[1]
probe = json.load(open(os.path.join(d, "probe.json")))
acc = defaultdict(lambda: defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    p = os.path.join(d, c + ".csv")
    if not os.path.exists(p):
        continue
    for row in csv.DictReader(open(p)):
        name = (row.get("Kernel_Name") or "").replace("void ", "").split("(")[0].strip()
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {"note": "scripts/debug/fetch_probe.hip under rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (one pass each); every "
               "kernel touches each byte of its buffer exactly once; factor = known bytes / (counter KB x 1024) = "
               "what the raw counter must be multiplied with for that access pattern; means over launches",
       "patterns": []}
for k in probe["launch_order"]:
    v = acc.get(k["kernel"], {})
    cname = "FETCH_SIZE" if k["kind"] == "read" else "WRITE_SIZE"
    vals = v.get(cname, [])
    other = v.get("WRITE_SIZE" if k["kind"] == "read" else "FETCH_SIZE", [])
    kb = sum(vals) / len(vals) if vals else None
    out["patterns"].append({**k, "counter": cname, "launches": len(vals),
                            "counter_KB_mean": None if kb is None else round(kb, 1),
                            "counter_KB_min_max": [round(min(vals), 1), round(max(vals), 1)] if vals else None,
                            "factor_bytes_over_counter": None if not kb else round(k["bytes"] / (kb * 1024), 3),
                            "other_counter_KB_mean": round(sum(other) / len(other), 1) if other else None})
print(json.dumps(out, indent=1))
PY
cat $OUT/fetch_calibration.json
