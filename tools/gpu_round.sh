#!/bin/bash
# One gpurun call: full GPU test suite, default bench (+ no-timing control), rocprofv3 kernel stats of the bench command,
# PMC passes (FETCH_SIZE / WRITE_SIZE separately) over the hash-grid microbench.  Outputs under gpurun_out/<tag>/.
tag=${1:-round}
out=$PWD/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err
timeout 300 python bench.py --no-kernel-timing --no-cpu-baseline --no-other --no-infer > $out/bench_notiming.json 2>> $out/bench.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-other --no-infer > $out/prof_bench.json 2> $out/prof.err )
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py --ops grid_fwd,grid_bwd --rays 8192 > $out/pmc_$c.log 2>&1 )
done
find $out -name "*.csv" -size +20M -delete
ls -la $out $out/prof/* 2>/dev/null | head -40
