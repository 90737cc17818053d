"""Per-kernel statistics of the TIMED REGION of a bench.py run out of a rocprofv3 kernel trace.

``bench.py --trace-markers`` launches one ``erfinv`` kernel (used nowhere else) right before and right
after the timed region; this script keeps the dispatches between the two and prints, per kernel name:
calls, average and total duration, share -- plus how much of the region's wall time the device was
busy (union of the kernel intervals), i.e. whether the region is kernel-bound or launch / host-bound.

usage: python scripts/trace_region.py <kernel_trace.csv> [--steps K] [--top N] [--json out.json]
"""
import argparse
import csv
import json
import sys


def load(path):
    rows = []
    with open(path, newline="") as f:
        r = csv.DictReader(f)
        for x in r:
            rows.append((x["Kernel_Name"], int(x["Start_Timestamp"]), int(x["End_Timestamp"])))
    rows.sort(key=lambda t: t[1])
    return rows


def region(rows):
    marks = [i for i, (n, _, _) in enumerate(rows) if "erfinv" in n]
    if len(marks) < 2:
        raise SystemExit(f"need two erfinv markers, found {len(marks)} (bench.py --trace-markers)")
    a, b = marks[0], marks[1]
    return rows[a + 1:b], rows[a][2], rows[b][1]


def busy_ns(rows):
    busy, cur_s, cur_e = 0, None, None
    for _, s, e in rows:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        busy += cur_e - cur_s
    return busy


def short(name):
    name = name.replace("rlpyt::(anonymous namespace)::", "").replace("void ", "")
    name = name.replace("at::native::", "").replace("(anonymous namespace)::", "")
    return name[:100]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=int, default=0, help="timed steps (adds per-step columns)")
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    rows, t0, t1 = region(load(a.trace))
    wall = t1 - t0
    per = {}
    for n, s, e in rows:
        c = per.setdefault(n, [0, 0])
        c[0] += 1
        c[1] += e - s
    total = sum(c[1] for c in per.values())
    busy = busy_ns(rows)
    out = dict(region_wall_ms=wall / 1e6, kernel_sum_ms=total / 1e6, device_busy_ms=busy / 1e6,
               device_busy_frac=busy / wall, dispatches=len(rows), steps=a.steps, kernels=[])
    print(f"timed region: wall {wall / 1e6:.2f} ms, {len(rows)} dispatches, kernel time {total / 1e6:.2f} ms, "
          f"device busy {busy / 1e6:.2f} ms = {busy / wall:.3f} of the wall"
          + (f"; per step: wall {wall / 1e6 / a.steps:.3f} ms, busy {busy / 1e6 / a.steps:.3f} ms, "
             f"{len(rows) / a.steps:.1f} dispatches" if a.steps else ""))
    for n, (calls, ns) in sorted(per.items(), key=lambda kv: -kv[1][1])[:a.top]:
        k = dict(name=short(n), calls=calls, avg_us=ns / calls / 1e3, total_ms=ns / 1e6, share=ns / total)
        if a.steps:
            k["calls_per_step"] = calls / a.steps
            k["us_per_step"] = ns / 1e3 / a.steps
        out["kernels"].append(k)
        print(f"  {k['name']:100s} {calls:7d} {k['avg_us']:8.1f} us {k['total_ms']:9.2f} ms {k['share']:6.3f}"
              + (f" {k['calls_per_step']:7.1f}/step {k['us_per_step']:8.1f} us/step" if a.steps else ""))
    if a.json:
        json.dump(out, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    sys.exit(main())
