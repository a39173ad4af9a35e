#!/bin/bash
out=$PWD/gpurun_out/r05_call12
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_field_glue.py tests/test_gpu_ffmlp.py tests/test_gpu_training.py -q -x -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -4 $out/pytest.log
timeout 600 python tools/ffmlp_skip_zero_ab.py > $out/skip_zero.json 2> $out/skip_zero.err; tail -2 $out/skip_zero.err
python - <<'PY'
import json
j=json.load(open("gpurun_out/r05_call12/skip_zero.json"))
for k,v in j.items():
    print(k, v["zero_gradient_samples"], v["all_zero_steps_of_32"], "mlp bwd", v["skip_zero=0"]["mlp_backward_us"], "->", v["skip_zero=1"]["mlp_backward_us"], "G2", v["skip_zero=0"]["hash_grid_backward_us"])
PY
timeout 300 python bench.py --no-cpu-baseline --no-other --no-infer --no-occupancy-timing --no-traffic-profile 2>/dev/null | grep '^{' | tail -1 > $out/bench.json
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r05_call12/bench.json").read())
print(j["ms_per_step"], {k:v for k,v in j["roofline"]["all_kernels_avg_us"].items() if "ffmlp" in k or "field" in k})
PY
