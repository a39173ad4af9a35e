#!/bin/bash
out=$PWD/gpurun_out/${1:-filldbg}
mkdir -p $out
run() { tag=$1; shift; env "$@" timeout 120 python tools/bench_kernels.py --ops grid_bwd --dtypes f16 --rays 8192 --kernels > $out/$tag.json 2>> $out/err.log; }
run base A=1
run grid1 NERFTEX_FILL_GRID=1
run grid2 NERFTEX_FILL_GRID=2
run grid4 NERFTEX_FILL_GRID=4
run nocopy NERFTEX_FILL_DBG=1
run noemit NERFTEX_FILL_DBG=3
run nosample NERFTEX_FILL_DBG=4
run none NERFTEX_FILL_DBG=7
run nomerge NERFTEX_GRID_BWD_NOMERGE=1
