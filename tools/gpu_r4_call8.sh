#!/bin/bash
out=$PWD/gpurun_out/${1:-r4c8}
mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_reference_python.py -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
timeout 300 python bench.py --steps 20 --no-other --no-cpu-baseline --no-infer --no-replay-profile > $out/bench20.json 2>> $out/bench.err
tail -3 $out/pytest.log
tail -5 $out/bench.err
python - <<PY
import json
for f in ("bench", "bench20"):
    d=json.loads(open("$out/%s.json" % f).read().strip().splitlines()[-1])
    print(f, {k:d[k] for k in ('value','ms_per_step','ms_per_step_spread','value_including_occupancy_update')}, d['config']['headline_loop'])
    r=d['roofline']; print(r['frac'], r['avg_launch_ms'], r['eager_avg_launch_ms'], r['durations_from'][:60], (d.get('rendered') or {}).get('mpix_per_s'))
    print([(o['workload'][:50], round(o['value']/1e6,1), round(o.get('ms_per_step',0),4)) for o in (d['other_config'] or [])])
PY
