#!/bin/bash
out=$PWD/gpurun_out/${1:-r4c19}
mkdir -p $out
export TMPDIR=/tmp
run() {
  python bench.py --gpus $1 --steps $2 --warmup 0 --rays 8192 --no-cpu-baseline --no-other --no-infer --no-kernel-timing --warm-seconds 0 --no-replay-profile $3 $4 2>> $out/err.log | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['config']
print('gpus $1 steps $2 $3 $4', repr(c['param_l1_after_run']), c['replicas_identical_after_run'])"
}
run 1 16 --no-graph --baked-pool
run 1 16 --no-graph --baked-pool
export NERFTEX_DP_SHARE_GPU=1
run 2 1 --no-graph
run 2 1 --no-graph
run 2 1 --no-graph
