"""Compact per-kernel summary (CSV) out of a rocprofv3 rocpd sqlite database."""
import csv
import re
import sqlite3
import sys


def short(name, n=110):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\s+", " ", name)
    return name if len(name) <= n else name[:n - 3] + "..."


def main(db_path, out_csv):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name,total_calls,total_duration,average,percentage "
                      "from top_kernels").fetchall()
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "pct"])
        for name, calls, tot, avg, pct in rows:
            w.writerow([short(name), calls, round(tot / 1e3, 1) if tot > 1e6 else round(tot, 1),
                        round(avg, 2), round(pct, 2)])
    print(f"wrote {len(rows)} kernels to {out_csv}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
