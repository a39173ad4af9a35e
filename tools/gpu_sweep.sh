#!/bin/bash
# sweep K4 slice size (grid backward) on the 8192-ray batch; then the default bench
out=$PWD/gpurun_out/${1:-sweep}
mkdir -p $out
for sl in 16384 32768 65536 131072; do
  NERFTEX_GRID_BWD_SLICE=$sl timeout 200 python tools/bench_kernels.py --ops grid_bwd --rays 8192 --kernels > $out/bwd_slice_$sl.json 2>> $out/err.log
done
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "grid" > $out/pytest_grid.log 2>&1
timeout 400 python bench.py --no-cpu-baseline --no-other --no-infer > $out/bench.json 2>> $out/err.log
timeout 400 python bench.py --no-cpu-baseline --no-other --no-infer --no-kernel-timing > $out/bench_nt.json 2>> $out/err.log
