#!/bin/bash
# rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; one counter per pass, no other trace domain) over the
# replay-sample kernels at the shapes of BASELINE configs #3 / #5 (scripts/replay_microbench.py) ->
# gpurun_out/<tag>/replay_pmc_counters.json (copy to profiles/rN_replay_pmc_counters.json: bench.py's
# pmc_traffic() reads the newest for the dqn / r2d1 lines).   usage: scripts/replay_pmc.sh <tag>
set -u
TAG=${1:-replay_pmc}
OUT=$PWD/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
python scripts/replay_microbench.py 30 > $OUT/timing.json 2> $OUT/timing.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 300 rocprofv3 --pmc $c --output-format csv -d $OUT/$c -- python scripts/replay_microbench.py 12 > $OUT/$c.log 2>&1
  f=$(find $OUT/$c -name '*counter_collection.csv' | head -1)
  head -1 "$f" > $OUT/$c.csv; grep -E "frames_gather|replay_step_fields|find_kernel" "$f" >> $OUT/$c.csv
  rm -rf $OUT/$c
done
python - $OUT <<'PY' > $OUT/replay_pmc_counters.json
import collections, csv, json, os, subprocess, sys
d = sys.argv[1]
timing = json.load(open(os.path.join(d, "timing.json")))
regions = {"frames_gather_pair": "frames_gather_pair", "frames_gather_seq": "frames_gather_wide_kernel",
           "replay_step_fields": "replay_step_fields_kernel", "sumtree_sample": "find_kernel"}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
names = collections.defaultdict(set)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    p = os.path.join(d, c + ".csv")
    if not os.path.exists(p):
        continue
    for row in csv.DictReader(open(p)):
        kn = row.get("Kernel_Name") or ""
        for region, pat in regions.items():
            if pat in kn or (region == "frames_gather_pair" and "frames_gather_kernel" in kn) \
                    or (region == "frames_gather_seq" and "frames_gather_seq" in kn):
                acc[region][c].append(float(row["Counter_Value"]))
                names[region].add(kn.split("(")[0])
                break
try:
    commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    commit = None
out = {"note": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (one pass each, no other trace domain) over "
               "scripts/replay_microbench.py: the replay kernels at the batch shapes of BASELINE configs #3 / #5 on "
               "8 GB frame rings, a fresh index set per launch; means over launches; hbm_bytes_corrected = "
               "(2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md (see profiles/r6_fetch_calibration.json for "
               "what the raw counters read on known byte counts in these access patterns)",
       "commit": commit, "kernels": {}}
for region, v in acc.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        f, w = (sum(v[c]) / len(v[c]) for c in ("FETCH_SIZE", "WRITE_SIZE"))
        alg = timing.get(region, {}).get("alg_bytes")
        out["kernels"][region] = {"kernels": sorted(names[region]), "dispatches": len(v["FETCH_SIZE"]),
                                  "FETCH_SIZE_KB": round(f, 1), "WRITE_SIZE_KB": round(w, 1),
                                  "hbm_bytes_corrected": int((2 * f + w) * 1024),
                                  "hbm_bytes_raw": int((f + w) * 1024), "alg_bytes": alg,
                                  "traffic_over_alg": round((2 * f + w) * 1024 / alg, 3) if alg else None,
                                  "us_per_launch": timing.get(region, {}).get("us")}
print(json.dumps(out, indent=1))
PY
cat $OUT/replay_pmc_counters.json
