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

# whole-iteration view: update phase = first conv2_fwd_x6 .. last clip_adam_apply of a run of 16
# minibatches; what sits between two update phases is the rollout (+ GAE + batch assembly)
ada = [i for i, r in enumerate(rows) if "clip_adam_apply" in r[2]]
if len(starts) >= 32 and ada:
    # group minibatch starts into iterations (gap > 3 ms between groups)
    groups, cur = [], [starts[0]]
    for a, b in zip(starts[:-1], starts[1:]):
        if rows[b][0] - rows[a][0] > 3e6:
            groups.append(cur)
            cur = []
        cur.append(b)
    groups.append(cur)
    for gi, g in enumerate(groups):
        t0 = rows[g[0] - 1][0] if g[0] > 0 else rows[g[0]][0]      # conv1_fwd before the first conv2
        last = max(i for i in ada if i >= g[-1] and (gi + 1 == len(groups) or i < groups[gi + 1][0]))
        t1 = rows[last][1]
        seg = rows[g[0] - 1:last + 1]
        busy = sum(e - s for s, e, _ in seg)
        nxt = rows[groups[gi + 1][0] - 1][0] - t1 if gi + 1 < len(groups) else float("nan")
        print(f"iteration {gi}: update {len(g)} minibatches, {(t1 - t0)/1e6:.2f} ms wall, "
              f"{busy/1e6:.2f} ms kernel busy, {len(seg)} kernels; then {nxt/1e6:.2f} ms to the next update")
    # between updates: kernel busy time and the largest kernels' share
    if len(groups) >= 2:
        g0, g1 = groups[-2], groups[-1]
        last0 = max(i for i in ada if g0[-1] <= i < g1[0])
        seg = rows[last0 + 1:g1[0] - 1]
        busy = sum(e - s for s, e, _ in seg)
        print(f"between the last two updates: {len(seg)} kernels, {busy/1e6:.2f} ms busy of "
              f"{(seg[-1][1] - seg[0][0])/1e6:.2f} ms")
        # the head (post-update, pre-rollout) and the tail (post-rollout, pre-update): first / last 300 kernels
        def span(s):
            return (s[-1][1] - s[0][0]) / 1e3, sum(e - b for b, e, _ in s) / 1e3
        # the rollout's per-time-step pattern: sample_convs launches
        sc = [i for i, r in enumerate(seg) if "sample_convs" in r[2]]
        if sc:
            head, tail = seg[:sc[0]], seg[sc[-1]:]
            print(f"  before the first sampling kernel: {len(head)} kernels, wall/busy us {span(head) if head else None}")
            print(f"  after the last sampling kernel : {len(tail)} kernels, wall/busy us {span(tail)}")
            for s_, e_, k_ in tail[:40]:
                print(f"     +{(s_ - tail[0][0])/1e3:8.1f} us  {(e_ - s_)/1e3:7.1f} us  {k_[:90]}")
