#!/bin/bash
# steps per replayed graph: sweep
for G in ${GROUPS_:-1 2 4 8 16}; do
  timeout 300 python bench.py --no-cpu-baseline --no-other --no-infer --no-kernel-timing --steps-per-graph $G > /tmp/b_$G.json 2>/tmp/b_$G.err
  python - <<PY
import json
d = json.loads(open("/tmp/b_$G.json").read().strip().splitlines()[-1])
print("G=$G", round(d["value"] / 1e6, 1), "M/s", round(d["ms_per_step"] * 1e3, 1), "us", d["config"]["samples_per_step_per_gpu"])
PY
done
