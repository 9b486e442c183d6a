"""Summarise a rocprofv3 --pmc counter_collection CSV: per-kernel per-launch averages of every counter.

    python scripts/pmc_summary.py gpurun_out/pmc/**/*counter_collection.csv > profiles/rNN_pmc.json
"""
import csv
import json
import re
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
launches = defaultdict(set)
for path in sys.argv[1:]:
    with open(path) as f:
        for row in csv.DictReader(f):
            kn = row["Kernel_Name"].replace("(anonymous namespace)::", "")
            m = re.match(r"(?:void )?([\w:]+)(<[^(]*>)?\(", kn)
            name = (m.group(1).split("::")[-1] + (m.group(2) or "")) if m else kn
            acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
            launches[(name, row["Counter_Name"])].add(row["Dispatch_Id"])
out = {}
for name, counters in acc.items():
    out[name] = {c: v / max(1, len(launches[(name, c)])) for c, v in sorted(counters.items())}
print(json.dumps(out, indent=1, sort_keys=True))
