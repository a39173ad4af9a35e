#!/bin/bash
# A/B of builds of the library on the bench's headline (NERFTEX_HIP_LIB): in-tree vs the ones named on the command line, two rounds
mkdir -p gpurun_out/ab
for round in 1 2; do
for lib in "" "$@"; do
  tag=$( [ -z "$lib" ] && echo intree || basename $lib .so )
  NERFTEX_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-replay-profile 2>/dev/null | grep '^{' | tail -1 >> gpurun_out/ab/$tag.jsonl
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/ab/*.jsonl")):
    print(f, [round(json.loads(l)["ms_per_step"]*1e3,1) for l in open(f)])
PY
