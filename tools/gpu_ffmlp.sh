#!/bin/bash
# FFMLP-side check: ffmlp / field-glue / training tests, default bench without baselines
out=$PWD/gpurun_out/${1:-ffmlp}
mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ffmlp.py tests/test_gpu_field_glue.py tests/test_gpu_training.py tests/test_independent_anchors.py -m gpu -q > $out/pytest.log 2>&1; echo "rc=$?" >> $out/pytest.log
timeout 400 python bench.py --no-cpu-baseline --no-other --no-infer > $out/bench.json 2>> $out/err.log
tail -3 $out/pytest.log
python - <<PY
import json
d = json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
k = d["roofline"].get("all_kernels_avg_us") or {}
for n, v in sorted(k.items(), key=lambda kv: -kv[1])[:18]: print(f"{n:45s} {v:8.2f}")
PY
