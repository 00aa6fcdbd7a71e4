#!/usr/bin/env python
"""Turn a rocprofv3 `--kernel-trace --stats` result database (rocpd sqlite, the default output of ROCm 7.2's
rocprofv3) into the per-kernel summary committed under profiles/.  Usage: summarize_rocpd.py results.db passes > out.txt
(`passes` = number of identical net passes in the profiled command, used for the per-pass columns)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))  # durations in us
tot = sum(r[2] for r in rows)
print(f"# kernels: {len(rows)}   total kernel time: {tot / 1e3:.2f} ms   per pass ({passes} passes): {tot / 1e3 / passes:.2f} ms")
print(f"{'kernel':70s} {'calls':>8s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s} {'ms/pass':>9s}")
for name, calls, dur, avg, pct in rows:
    print(f"{name[:70]:70s} {calls:8d} {dur / 1e3:10.3f} {avg:10.2f} {pct:6.2f} {dur / 1e3 / passes:9.3f}")
