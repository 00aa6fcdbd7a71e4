#!/usr/bin/env python
"""Per-kernel averages of a rocprofv3 `--pmc X --kernel-trace --output-format csv` run (…_counter_collection.csv).
Usage: summarize_pmc.py counter_collection.csv [more.csv ...] > out.txt    (FETCH_SIZE / WRITE_SIZE are in KiB)"""
import collections
import csv
import sys

csv.field_size_limit(1 << 30)
agg = collections.defaultdict(lambda: [0, 0.0])
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        name = r['Kernel_Name']
        name = name[:name.index('(')] if '(' in name and not name.startswith('void at::') else name[:60]
        k = (name, r['Counter_Name'])
        agg[k][0] += 1
        agg[k][1] += float(r['Counter_Value'])
print(f"{'kernel':64s} {'counter':14s} {'launches':>9s} {'avg/launch':>14s} {'total':>16s}")
for (name, ctr), (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{name[:64]:64s} {ctr:14s} {n:9d} {tot / n:14.1f} {tot:16.1f}")
