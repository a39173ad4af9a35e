#!/bin/bash
# quick GPU check: grid parity tests, grid microbench (8192 rays) with per-kernel times, default bench without baselines
out=$PWD/gpurun_out/${1:-quick}
mkdir -p $out
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "grid" > $out/pytest_grid.log 2>&1
timeout 200 python tools/bench_kernels.py --ops grid_fwd,grid_bwd --rays 8192 --kernels > $out/kern.json 2>> $out/err.log
timeout 400 python bench.py --no-cpu-baseline --no-other --no-infer > $out/bench.json 2>> $out/err.log
