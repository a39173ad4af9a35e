#!/bin/bash
run1() { python bench.py --gpus 1 --steps $2 --warmup 0 --rays 8192 --no-cpu-baseline --no-other --no-infer --no-kernel-timing --warm-seconds 0 --no-replay-profile --no-graph --baked-pool $3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('steps $2 $3 run $1', repr(d['config']['param_l1_after_run']), d['config']['mean_count'])"; }
for st in 1 4 12; do
  run1 solo $st
  run1 a $st & run1 b $st & run1 c $st & wait
done
run1 solo 12 --no-fused-tail
run1 a 12 --no-fused-tail & run1 b 12 --no-fused-tail & run1 c 12 --no-fused-tail & wait
