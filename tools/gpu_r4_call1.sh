#!/bin/bash
# round 4, call 1: full GPU suite, default bench, the precision table
out=$PWD/gpurun_out/${1:-r4c1}
mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --durations=15 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
timeout 300 python -m pytest tests/test_gpu_round4.py -m gpu -q -s > $out/pytest_r4.log 2>&1; echo "pytest rc=$?" >> $out/pytest_r4.log
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
timeout 300 python tools/precision_table.py $out/precision.json > /dev/null 2> $out/precision.err
grep -E "^(FAILED|ERROR)|passed|failed" $out/pytest.log | tail -20
tail -5 $out/pytest_r4.log
tail -5 $out/bench.err $out/precision.err
python - <<PY
import json
d=json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_spread','occupancy_update','value_including_occupancy_update')})
r=d['roofline']; print(r['frac'], r['avg_launch_ms'], r['eager_avg_launch_ms'], r['durations_from'][:40], d.get('rendered',{}).get('mpix_per_s'))
print(r['all_kernels_avg_us'])
PY
