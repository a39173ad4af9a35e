#!/bin/bash
out=$PWD/gpurun_out/r4c21
mkdir -p $out
python tools/gloo_cuda_probe.py 2>&1 | grep -v "amdgpu.ids\|Gloo\]" | tail -3
run1() { python bench.py --gpus 1 --steps 16 --warmup 0 --rays 8192 --no-cpu-baseline --no-other --no-infer --no-kernel-timing --warm-seconds 0 --no-replay-profile --no-graph --baked-pool 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', repr(d['config']['param_l1_after_run']))"; }
run1 a & run1 b & wait
run1 c & run1 d & wait
