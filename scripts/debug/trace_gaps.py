"""Busy / idle GPU time between consecutive conv1_fwd launches of the PPO update (one minibatch each)
from a rocprofv3 --kernel-trace CSV."""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# minibatch boundaries: conv1_fwd_kernel launches followed (soon) by conv2_fwd_x6
starts = [i for i, r in enumerate(rows) if "conv2_fwd_x6_kernel" in r[2]]
spans = []
for a, b in zip(starts[:-1], starts[1:]):
    seg = rows[a:b]
    wall = seg[-1][1] - seg[0][0]
    nxt = rows[b][0] - seg[0][0]
    busy = sum(e - s for s, e, _ in seg)
    if nxt < 3e6:      # consecutive minibatches (< 3 ms apart)
        spans.append((nxt, busy, len(seg)))
if not spans:
    sys.exit("no minibatch spans found")
n = len(spans)
print(f"{n} consecutive minibatches: period {sum(s[0] for s in spans)/n/1e3:.1f} us, "
      f"kernel busy {sum(s[1] for s in spans)/n/1e3:.1f} us, kernels per minibatch {sum(s[2] for s in spans)/n:.1f}")
# largest gaps inside one typical minibatch
a, b = starts[len(starts) // 2], starts[len(starts) // 2 + 1]
seg = rows[a:b + 1]
gaps = sorted(((seg[i + 1][0] - seg[i][1], seg[i][2][:50], seg[i + 1][2][:50]) for i in range(len(seg) - 1)), reverse=True)
for g, k0, k1 in gaps[:8]:
    print(f"  gap {g/1e3:7.1f} us after {k0} -> {k1}")
