#!/bin/bash
out=$PWD/gpurun_out/${1:-r4c18}
mkdir -p $out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_round4.py -m gpu -q -x -k "chunk or phased or group" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -3 $out/pytest.log
export NERFTEX_DP_SHARE_GPU=1
for cfg in "2 --no-graph" "2 --no-graph" "2 --no-graph --no-fused-opt" "2 --no-graph --no-fused-opt" "2 --no-graph --wire fp32" "2 --no-graph --wire fp32"; do
  set -- $cfg
  python bench.py --gpus $1 --steps 16 --warmup 0 --rays 8192 --no-cpu-baseline --no-other --no-infer --no-kernel-timing --warm-seconds 0 $2 $3 $4 $5 2>> $out/err.log | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['config']
print('$cfg', repr(c['param_l1_after_run']), c['replicas_identical_after_run'])"
done
