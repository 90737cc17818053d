"""Per-kernel averages of the rocprofv3 --pmc CSVs written by scripts/pmc_conv.sh."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def main(d):
    acc = defaultdict(lambda: defaultdict(list))
    for path in sorted(glob.glob(os.path.join(d, "*.csv"))):
        with open(path) as f:
            for row in csv.DictReader(f):
                name = row.get("Kernel_Name") or row.get("kernel_name") or ""
                name = re.sub(r"\(anonymous namespace\)::|rlpyt::|void ", "", name).split("(")[0]
                ctr = row.get("Counter_Name") or row.get("counter_name")
                val = row.get("Counter_Value") or row.get("counter_value")
                if ctr and val and "conv" in name:
                    acc[name][ctr].append(float(val))
    out = {}
    for name, ctrs in acc.items():
        out[name] = {c: round(sum(v) / len(v), 1) for c, v in ctrs.items()}
        out[name]["dispatches"] = max(len(v) for v in ctrs.values())
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main(sys.argv[1])
