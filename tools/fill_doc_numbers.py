#!/usr/bin/env python3
"""Fill the @@KEY@@ placeholders of DESIGN.md / README.md from profiles/r06_bench.json (the final build's bench line), so that the prose quotes the
numbers of record and nothing else.  python tools/fill_doc_numbers.py [--check]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
j = json.loads([ln for ln in open(os.path.join(ROOT, "profiles", "r06_bench.json")) if ln.startswith("{")][-1])
r = j["roofline"]
k = r["all_kernels_avg_us"]
g2 = r["kernels_avg_us"]


def other(sub):
    return next((o for o in j["other_config"] if sub in o["workload"]), None)


curved = other("configs[3]")
bf16 = other("in bf16 through ngp_harness.accelerate")
vals = {
    "MS": f"{j['ms_per_step']:.4f}", "VAL": f"{j['value'] / 1e6:.0f}",
    "BIN": f"{g2['bin_fill_dir_kernel']:.0f}", "SUM": f"{g2['sum_tiles_dir_kernel']:.0f}", "COMB": f"{g2['combine_tiles_kernel']:.0f}",
    "G2": f"{sum(g2.values()):.0f}",
    "BWD": f"{2 * k['ffmlp_backward_fused_kernel']:.0f}", "RED": f"{k['ffmlp_wgrad_reduce2_kernel']:.1f}" if "ffmlp_wgrad_reduce2_kernel" in k else "0 (its reduction rides on the fill launch)", "G1": f"{k['grid_forward_level_kernel']:.0f}",
    "FWD": f"{k['field_forward_kernel']:.0f}",
    "COMPOSITE": f"{sum(v for n, v in k.items() if n.startswith('composite_') or n == 'render_tail_forward_kernel'):.0f}",
    "ADAM": f"{k['adam_half_kernel']:.0f}", "MARCH": f"{k['march_count_parallel_kernel']:.0f} + {k['march_expand_kernel']:.0f}",
    "ACH": f"{r['achieved']:.0f}", "FRAC": f"{r['frac']:.3f}", "FRAC_BWD": f"{r['frac_backward_bytes_only']:.3f}",
    "TRAFFIC": f"{r['traffic'] / 1e6:.0f}", "TOA": f"{r['traffic_over_algorithmic']:.2f}",
    "MPIX": f"{j['rendered']['mpix_per_s']:.1f}",
    "CURVED_EAGER": f"{curved['field_forward_us']:.0f}" if curved else "?", "CURVED_GRAPH": f"{curved['field_forward_graphed_us']:.0f}" if curved and curved.get("field_forward_graphed_us") else "?",
    "BF16": f"{bf16['value'] / 1e6:.0f}" if bf16 else "?", "OCC": f"{j['value_including_occupancy_update'] / 1e6:.0f}",
}
check = "--check" in sys.argv
for name in ("DESIGN.md", "README.md"):
    path = os.path.join(ROOT, name)
    s = open(path).read()
    keys = set(re.findall(r"@@([A-Z0-9_]+)@@", s))
    missing = keys - set(vals)
    assert not missing, (name, missing)
    if check:
        print(name, sorted(keys))
        continue
    for key in keys:
        s = s.replace(f"@@{key}@@", vals[key])
    open(path, "w").write(s)
    print(name, "filled", len(keys), "keys")
print({a: b for a, b in vals.items()})
