#!/bin/bash
# round 4, call 3: the transposing-read MLP backward -- tests, A/B timing against the selection-matrix form, occupancy update breakdown
out=$PWD/gpurun_out/${1:-r4c3}
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ffmlp.py tests/test_gpu_field_glue.py tests/test_gpu_round4.py tests/test_gpu_training.py -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
for sel in 0 1; do
  NERFTEX_TUNE="ffmlp_bwd_sel=$sel" timeout 300 python tools/bench_kernels.py --rays 8192 --ops ffmlp --kernels > $out/ffmlp_sel$sel.json 2>> $out/err.log
  NERFTEX_TUNE="ffmlp_bwd_sel=$sel" timeout 400 python bench.py --no-cpu-baseline --no-other --no-infer --no-replay-profile --no-occupancy-timing > $out/bench_sel$sel.json 2>> $out/err.log
done
timeout 300 python tools/occupancy_breakdown.py > $out/occupancy.json 2>> $out/err.log
tail -4 $out/pytest.log
python - <<PY
import json
for sel in (0, 1):
    k = json.load(open("$out/ffmlp_sel%d.json" % sel))
    print("sel", sel, {n: round(v["ms"] * 1e3, 1) for n, v in k.items() if isinstance(v, dict) and "ms" in v and "bwd" in n})
    d = json.loads(open("$out/bench_sel%d.json" % sel).read().strip().splitlines()[-1])
    a = d["roofline"]["all_kernels_avg_us"]
    print("   bench", round(d["value"] / 1e6, 1), round(d["ms_per_step"], 4), d["ms_per_step_spread"]["median"], {n: a[n] for n in a if "field" in n or "reduce" in n})
print(open("$out/occupancy.json").read())
PY
tail -3 $out/err.log
