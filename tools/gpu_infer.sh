#!/bin/bash
# inference march A/B: parity tests of the inference loop, then the rendered-frame microbench with the serial and the data-parallel march
out=$PWD/gpurun_out/${1:-infer}
mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -k "inference or infer or march" > $out/pytest.log 2>&1; echo "rc=$?" >> $out/pytest.log
NERFTEX_TUNE="march_infer_serial=1" timeout 300 python tools/bench_infer.py 3 4 > $out/serial.json 2>> $out/err.log
timeout 300 python tools/bench_infer.py 3 4 > $out/parallel.json 2>> $out/err.log
tail -4 $out/pytest.log
for f in serial parallel; do echo $f; head -2 $out/$f.json; grep -h "slots_per_ray\": 4, \"parts\": 3" $out/$f.json; tail -1 $out/$f.json | python -c "
import json,sys
k=json.loads(sys.stdin.read())
print({n: (v['calls'], round(v['total_us'])) for n,v in k.items() if 'march' in n or 'grid' in n or 'field' in n or 'compos' in n or 'compact' in n})"; done
