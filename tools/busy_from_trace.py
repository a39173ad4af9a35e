#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace csv: over the last `window_ms` of the trace, the sum of the kernel durations, the time at least one kernel
was running (union of the intervals) and the wall time -- is a loop device-bound (busy ~ wall) or launch-bound (busy << wall)?
    python tools/busy_from_trace.py trace.csv [t_from_end_ms t_to_end_ms]"""
import csv
import sys

rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
t_end = max(e for _, e, _ in rows)
a = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
b = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
lo, hi = t_end - int(a * 1e6), t_end - int(b * 1e6)
sel = [(s, e, n) for s, e, n in rows if s >= lo and e <= hi]
tot = sum(e - s for s, e, _ in sel)
busy, cur_s, cur_e = 0, None, None
for s, e, _ in sel:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
if cur_e is not None:
    busy += cur_e - cur_s
wall = sel[-1][1] - sel[0][0]
print(f"window {a}..{b} ms before the end: {len(sel)} kernels, wall {wall / 1e6:.2f} ms, some kernel running {busy / 1e6:.2f} ms ({100 * busy / wall:.1f} %), "
      f"sum of kernel durations {tot / 1e6:.2f} ms (average concurrency {tot / busy:.2f})")
