#!/bin/bash
python -m pytest tests/test_gpu_parity.py tests/test_gpu_round4.py tests/test_gpu_field_glue.py -m gpu -x -q -p no:cacheprovider 2>&1 | grep -v amdgpu | tail -3
for i in 1 2; do python bench.py --no-cpu-baseline --no-replay-profile 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('ms/step', r['ms_per_step'], 'G1 eager us', r['roofline']['all_kernels_avg_us_eager'].get('grid_forward_level_kernel'))"; done
bash tools/gpu_soak_beside_neighbour.sh 2>&1 | tail -8
