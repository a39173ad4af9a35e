#!/bin/bash
out=$PWD/gpurun_out/r05_call5
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_training.py -q -x -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -25 $out/pytest.log
timeout 600 python bench.py --no-cpu-baseline --no-other --no-replay-profile --no-occupancy-timing > $out/bench.json 2> $out/bench.err
tail -3 $out/bench.err
python - <<'PY'
import json
j=json.loads([l for l in open("gpurun_out/r05_call5/bench.json") if l.startswith("{")][-1])
print("ms_per_step", j["ms_per_step"])
r=j["rendered"]
print(json.dumps({k:v for k,v in r.items() if k!="loop"})[:1800])
PY
