#!/bin/bash
out=$PWD/gpurun_out/${1:-r4c16}
mkdir -p $out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_round4.py tests/test_gpu_reference_python.py tests/test_gpu_dp_shared_gpu.py -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
timeout 600 python bench.py --no-other --no-cpu-baseline --no-infer --no-replay-profile > $out/bench.json 2> $out/bench.err
tail -4 $out/pytest.log
tail -3 $out/bench.err
python - <<PY
import json
d=json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_spread','value_including_occupancy_update')}, d['config']['headline_loop'])
print([(o['workload'][:50], round(o['value']/1e6,1), round(o.get('ms_per_step',0),4)) for o in (d['other_config'] or [])])
PY
