#!/bin/bash
# ON THE GPU BOX: rocprofv3 kernel stats of bench.py --config dqn|r2d1 on a wrapped ring, then
# separate FETCH_SIZE / WRITE_SIZE passes (short fill: the counters are per launch) for its replay
# frame-gather kernel -> gpurun_out/<tag>/{bench.json,kernel_stats.csv,replay_pmc.json}
# usage: scripts/prof_config.sh dqn|r2d1 <tag>
CFG=$1; TAG=${2:-$1}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
timeout -k 5 400 python bench.py --config $CFG > $OUT/bench.json 2> $OUT/bench.err; tail -c 2000 $OUT/bench.err > $OUT/bench.err.tail; rm -f $OUT/bench.err
if [ $CFG = dqn ]; then SHORT="--replay-fill-itrs 1500 --steps 150 --warmup 20"; PM="--replay-fill-itrs 400 --steps 40 --warmup 5"; KERN="frames_gather_kernel";
else SHORT="--replay-fill-itrs 10 --steps 6 --warmup 2"; PM="--replay-fill-itrs 6 --steps 3 --warmup 1"; KERN="frames_gather_wide_kernel"; fi
timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -- python bench.py --config $CFG $SHORT > /dev/null 2>&1
f=$(find $OUT/raw -name '*kernel_stats.csv' | head -1); cp "$f" $OUT/kernel_stats.csv 2>/dev/null; rm -rf $OUT/raw
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 300 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- python bench.py --config $CFG $PM > /dev/null 2>&1
  # (one row per dispatch of EVERY kernel: keep the header and the gather kernel's rows only)
  f=$(find $OUT/pmc_$c -name '*counter_collection.csv' | head -1)
  head -1 "$f" > $OUT/pmc_$c.csv; grep "$KERN" "$f" >> $OUT/pmc_$c.csv
  rm -rf $OUT/pmc_$c
done
python - $OUT $KERN $CFG > $OUT/replay_pmc.json <<'PY'
import csv, json, sys, collections
out, kern, cfg = sys.argv[1:4]
acc = collections.defaultdict(list)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    try:
        for row in csv.DictReader(open(f"{out}/pmc_{c}.csv")):
            if kern in row["Kernel_Name"]:
                acc[c].append((float(row["Counter_Value"]), row.get("Grid_Size") or row.get("Workgroup_Size") or ""))
    except OSError:
        pass
res = {}
for c, v in acc.items():
    # the bench's own roofline probe launches dominate neither count nor size: take the launches
    # of the most frequent grid size (the batch shape the algorithm issues)
    grids = collections.Counter(g for _, g in v)
    top = grids.most_common(1)[0][0]
    vals = [x for x, g in v if g == top]
    res[c] = {"mean_KB": sum(vals) / len(vals), "launches": len(vals), "grid": top}
print(json.dumps({"config": cfg, "kernel": kern, "counters": res}, indent=1))
PY
cat $OUT/replay_pmc.json
head -12 $OUT/kernel_stats.csv | cut -c1-160
python -c "
import json; d = json.load(open('$OUT/bench.json')); print(d['value'], d['ms_per_step'], d['config']['ring_wrapped'], d['config']['ring_rows_filled'], d['updates_per_s']); print(d['roofline'])"
