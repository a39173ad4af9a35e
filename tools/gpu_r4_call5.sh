#!/bin/bash
# round 4, call 5: step_group (fresh-ray headline), N4 gradient test, TR-vs-SEL equivalence test, full bench with the rocprofv3 child
out=$PWD/gpurun_out/${1:-r4c5}
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_ffmlp.py -m gpu -q -x -s > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
timeout 600 python -m pytest tests/test_gpu_round3.py -m gpu -q -x -k "accelerate" > $out/pytest3.log 2>&1; echo "pytest rc=$?" >> $out/pytest3.log
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
tail -4 $out/pytest.log; tail -3 $out/pytest3.log
grep -h "cosine" $out/pytest.log
tail -5 $out/bench.err
python - <<PY
import json
d=json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_spread','value_including_occupancy_update')}, d['config']['headline_loop'])
r=d['roofline']; print(r['frac'], r['avg_launch_ms'], r['eager_avg_launch_ms'], r['durations_from'][:60], d.get('rendered',{}).get('mpix_per_s'))
print(r['all_kernels_avg_us'])
print([(o['workload'][:50], round(o['value']/1e6,1), round(o.get('ms_per_step',0),4)) for o in d['other_config']])
PY
