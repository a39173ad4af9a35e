#!/bin/bash
# full GPU suite + default bench (no CPU baseline) + inference microbench
out=$PWD/gpurun_out/${1:-check}
mkdir -p $out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
timeout 400 python bench.py --no-cpu-baseline --no-other > $out/bench.json 2>> $out/err.log
timeout 300 python tools/bench_infer.py 3 4 > $out/infer.json 2>> $out/err.log
grep -E "^(FAILED|ERROR)|passed|failed" $out/pytest.log | tail -20
python - <<PY
import json
d=json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d.get('rendered',{}).get('mpix_per_s'))
print(d['roofline']['all_kernels_avg_us'])
PY
grep -h "parts\": 3" $out/infer.json; tail -1 $out/infer.json | python -c "
import json,sys
k=json.loads(sys.stdin.read())
print({n: round(v['total_us']) for n,v in k.items()})"
