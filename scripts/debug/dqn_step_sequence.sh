#!/bin/bash
# (ON THE GPU BOX) the kernel sequence of ONE DQN iteration (2 sampling steps + 2 captured updates) from a
# rocprofv3 kernel trace: names in start order with durations and gaps.
OUT=$PWD/gpurun_out/${1:-r6w_seq}; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/raw -- python bench.py --config dqn --replay-fill-itrs 300 --steps 60 --warmup 20 --no-cpu-baseline > $OUT/bench.json 2> $OUT/prof.log
python - "$(find $OUT/raw -name '*kernel_trace.csv' | head -1)" <<'PY' > $OUT/sequence.txt
import csv, sys, re
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ticks = [i for i, r in enumerate(rows) if "update_tick_kernel" in r["Kernel_Name"]]
i0 = ticks[-40]            # an update well inside the timed region
# back up to the previous sampling step's first kernel (frame_push)
j = i0
while j > 0 and "frame_push_kernel" not in rows[j]["Kernel_Name"]:
    j -= 1
while j > 0 and int(rows[j]["Start_Timestamp"]) - int(rows[j - 1]["End_Timestamp"]) < 30000 and j > i0 - 40:
    j -= 1
end = ticks[-38]
prev_end = None
for r in rows[j:end]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("rlpyt::(anonymous namespace)::", "").replace("void ", "")
    name = re.sub(r"at::native::(\(anonymous namespace\)::)?", "", name)
    name = re.sub(r"\(.*", "", name)[:110]
    gap = 0 if prev_end is None else (s - prev_end) / 1e3
    print(f"{gap:8.1f} gap {(e - s) / 1e3:7.1f} us  {name}")
    prev_end = e
PY
rm -rf $OUT/raw
cat $OUT/sequence.txt
