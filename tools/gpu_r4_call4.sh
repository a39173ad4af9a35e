#!/bin/bash
# round 4, call 4: pipelined transposing-read MLP backward A/B + N4 drop-in tests
out=$PWD/gpurun_out/${1:-r4c4}
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ffmlp.py tests/test_gpu_field_glue.py tests/test_gpu_round4.py tests/test_gpu_raytracer.py -m gpu -q -x -s > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
timeout 600 python -m pytest tests/test_gpu_round3.py -m gpu -q -x -k "projector or curved" > $out/pytest3.log 2>&1; echo "pytest rc=$?" >> $out/pytest3.log
for sel in 0 1 0 1; do
  NERFTEX_TUNE="ffmlp_bwd_sel=$sel" timeout 400 python bench.py --no-cpu-baseline --no-other --no-infer --no-replay-profile --no-occupancy-timing > $out/bench_sel$sel.json 2>> $out/err.log
  python - <<PY
import json
d = json.loads(open("$out/bench_sel$sel.json").read().strip().splitlines()[-1])
a = d["roofline"]["all_kernels_avg_us"]
print("sel", $sel, round(d["value"] / 1e6, 1), round(d["ms_per_step"], 4), d["ms_per_step_spread"]["median"], {n: a[n] for n in a if "field" in n or "reduce" in n})
PY
done
tail -3 $out/pytest.log $out/pytest3.log
grep -h "cosine\|fp32 chain" $out/pytest.log
tail -3 $out/err.log
