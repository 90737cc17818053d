"""Device occupancy of the ROLLOUT phases of a DQN-family bench line from a rocprofv3 kernel trace
(``--kernel-trace --output-format csv``): the update phases are cut out (first replay gather .. last Adam
launch of each update), what is left is the sampling; prints wall, union-busy time, summed kernel time
(= mean concurrency when divided by busy) and the kernels that fill the rest.
usage: python scripts/debug/rollout_occupancy.py <kernel_trace.csv> [first_kernel_substr] [last_kernel_substr]"""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else "frames_gather_wide"
last = sys.argv[3] if len(sys.argv) > 3 else "clip_adam_apply"
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
rows.sort()
# update windows
wins, cur = [], None
for s, e, n, q in rows:
    if first in n:
        cur = [s, e]
    elif cur is not None:
        cur[1] = max(cur[1], e)
        if last in n:
            wins.append(tuple(cur))
            cur = None
wins = wins[2:-1]                     # steady state
if len(wins) < 3:
    sys.exit(f"only {len(wins)} update windows found")


def union(iv):
    iv.sort()
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            cs, ce = s, e
    return tot + (ce - cs if cs is not None else 0)


upd_wall = sum(e - s for s, e in wins)
roll = [(wins[i][1], wins[i + 1][0]) for i in range(len(wins) - 1)]
roll_wall = sum(e - s for s, e in roll)
stats = {"update": [upd_wall, [], 0], "rollout": [roll_wall, [], 0]}
per_kernel = defaultdict(lambda: [0, 0])
queues = defaultdict(int)
for s, e, n, q in rows:
    for (ws, we) in wins[:-1]:
        if s >= ws and e <= we:
            stats["update"][1].append((s, e)); stats["update"][2] += e - s
    for (ws, we) in roll:
        if s >= ws and e <= we:
            stats["rollout"][1].append((s, e)); stats["rollout"][2] += e - s
            k = n.split("(")[0][-60:]
            per_kernel[k][0] += 1; per_kernel[k][1] += e - s
            queues[q] += e - s
n_roll = len(roll)
for name, (wall, iv, ksum) in stats.items():
    n = len(wins) - 1
    busy = union(iv)
    print(f"{name}: {n} phases, wall {wall / n / 1e6:.3f} ms each, device busy {busy / n / 1e6:.3f} ms "
          f"({busy / wall:.3f} of wall), kernel time {ksum / n / 1e6:.3f} ms (concurrency {ksum / max(busy, 1):.2f})")
print("rollout kernels (per phase):")
for k, (c, t) in sorted(per_kernel.items(), key=lambda x: -x[1][1])[:16]:
    print(f"  {k:60s} {c / n_roll:7.1f} x {t / c / 1e3:6.1f} us = {t / n_roll / 1e6:.3f} ms")
print("rollout kernel time by queue:", {q: round(t / n_roll / 1e6, 3) for q, t in queues.items()})

# idle gaps inside the update phases: where the device waits for the host (kernel before -> kernel after)
gaps = defaultdict(lambda: [0, 0])
tot_gap = 0
for (ws, we) in wins[:-1]:
    ks = [(s, e, n) for s, e, n, q in rows if s >= ws and e <= we]
    end, prev = ks[0][1], ks[0][2]
    for s, e, n in ks[1:]:
        if s > end:
            key = (prev.split("(")[0][-38:], n.split("(")[0][-38:])
            gaps[key][0] += 1; gaps[key][1] += s - end
            tot_gap += s - end
        if e > end:
            end, prev = e, n
n_upd = len(wins) - 1
print(f"update: idle {tot_gap / n_upd / 1e6:.3f} ms per phase; largest gap classes (per phase):")
for (a, b), (c, t) in sorted(gaps.items(), key=lambda x: -x[1][1])[:14]:
    print(f"  {a:38s} -> {b:38s} {c / n_upd:6.1f} x {t / c / 1e3:7.1f} us = {t / n_upd / 1e6:.3f} ms")
