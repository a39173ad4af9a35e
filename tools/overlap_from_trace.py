#!/usr/bin/env python3
"""How much of kernel A's device time runs WHILE kernel B runs, from a rocprofv3 --kernel-trace csv:
    python tools/overlap_from_trace.py trace.csv adam_half_kernel sum_tiles_dir_kernel [last N launches of A, default 256]"""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    a_name, b_name = sys.argv[2], sys.argv[3]
    last = int(sys.argv[4]) if len(sys.argv) > 4 else 256
    A = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if a_name in r["Kernel_Name"])[-last:]
    B = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if b_name in r["Kernel_Name"])
    if not A or not B:
        print("kernels not found")
        return
    t_lo = A[0][0]
    B = [b for b in B if b[1] >= t_lo]
    tot = ov = 0
    for s, e in A:
        tot += e - s
        for bs, be in B:
            if be <= s:
                continue
            if bs >= e:
                break
            ov += min(e, be) - max(s, bs)
    print(f"{a_name}: {len(A)} launches, {tot / len(A) / 1e3:.1f} us each; {100.0 * ov / tot:.1f} % of that time beside {b_name} ({sum(e - s for s, e in B) / max(len(B), 1) / 1e3:.1f} us each, {len(B)} launches)")


if __name__ == "__main__":
    main()
