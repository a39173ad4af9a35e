#!/bin/bash
out=$PWD/gpurun_out/${1:-r4c17}
mkdir -p $out
export TMPDIR=/tmp NERFTEX_DP_SHARE_GPU=1
for cfg in "1" "1" "3" "1 --no-graph" "3 --no-graph" "3 --no-graph"; do
  set -- $cfg
  python bench.py --gpus 2 --steps 16 --warmup 4 --rays 8192 --no-cpu-baseline --no-other --no-infer --no-kernel-timing --warm-seconds 0 --allreduce-chunks $1 $2 2>> $out/err.log | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['config']
print('chunks $1 $2', repr(c['param_l1_after_run']), c['replicas_identical_after_run'], c['collective']['table_gradient_chunks'], round(d['ms_per_step'],3))"
done
tail -3 $out/err.log
