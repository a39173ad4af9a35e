#!/bin/bash
# A/B of two builds of the library on the bench's headline (NERFTEX_HIP_LIB): in-tree vs the one named by $1
mkdir -p gpurun_out/ab
for lib in "" "$1" "" "$1"; do
  tag=$( [ -z "$lib" ] && echo intree || basename $lib .so )
  NERFTEX_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-replay-profile 2>/dev/null | grep '^{' | tail -1 >> gpurun_out/ab/$tag.jsonl
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/ab/*.jsonl")):
    for l in open(f):
        r=json.loads(l)
        k=r["config"].get("kernels_us") or r["config"].get("kernel_us") or {}
        print(f, "ms/step", r["ms_per_step"], "value", r["value"], "infer", (r["config"].get("inference") or {}).get("mpix_per_s"), {n:v for n,v in list(k.items())[:12]} if isinstance(k,dict) else "")
PY
