#!/bin/bash
# MFMA-utilisation counters for the FFMLP kernels (microbench at 8192 rays = 456,064 rows per launch)
out=$PWD/gpurun_out/${1:-pmcffmlp}
mkdir -p $out
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --kernel-trace --output-format csv -d $out/p1 -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py --ops ffmlp --rays 8192 > $out/p1.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/p2 -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py --ops ffmlp --rays 8192 > $out/p2.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/p3 -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py --ops ffmlp --rays 8192 > $out/p3.log 2>&1 )
find $out -name "*kernel_trace.csv" -size +5M -delete
ls $out/p1/* | head; tail -3 $out/p1.log
