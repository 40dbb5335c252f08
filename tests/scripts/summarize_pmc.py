#!/usr/bin/env python3
"""Per-kernel averages of the rocprofv3 counter CSVs under <dir>/pmc*/ (one JSON object on stdout)."""
import collections, csv, glob, json, re, sys

def short(name: str) -> str:
    m = re.search(r"(k_[a-z0-9_]+)", name)
    if m:
        return m.group(1)
    if "rocprim" in name:
        return "rocprim_sort"
    return name.split("(")[0][-40:]

acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in sorted(glob.glob(sys.argv[1] + "/pmc*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        a = acc[short(r["Kernel_Name"])][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
out = {k: {c: {"avg_per_launch": v[0] / v[1], "launches": v[1]} for c, v in d.items()} for k, d in acc.items()}
json.dump(out, sys.stdout, indent=1, sort_keys=True)
