#!/bin/bash
out=$PWD/gpurun_out/r05_call11
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_field_glue.py tests/test_gpu_ffmlp.py tests/test_gpu_training.py -q -x -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -6 $out/pytest.log
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-other --no-infer --no-occupancy-timing --no-traffic-profile 2>/dev/null | grep '^{' | tail -1 > $out/bench_$i.json; done
python - <<'PY'
import json
for i in (1,2):
    j=json.loads(open("gpurun_out/r05_call11/bench_%d.json"%i).read())
    print(j["ms_per_step"], {k:v for k,v in j["roofline"]["all_kernels_avg_us"].items() if "ffmlp" in k or "field" in k})
PY
