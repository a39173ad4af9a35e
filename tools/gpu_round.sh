#!/bin/bash
# One gpurun call: full GPU test suite, default bench, rocprofv3 kernel stats of the bench command, PMC passes (FETCH_SIZE / WRITE_SIZE in
# separate runs) over the hash-grid microbench and the FFMLP microbench.  Outputs under gpurun_out/<tag>/.
tag=${1:-round}
out=$PWD/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-other --no-infer > $out/prof_bench.json 2> $out/prof.err )
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py --ops grid_fwd,grid_bwd --rays 8192 --dtypes f16 > $out/pmc_$c.log 2>&1 )
done
timeout 200 python tools/g2_experiments.py > $out/g2.json 2> $out/g2.err
timeout 200 python tools/bench_kernels.py --rays 8192 --kernels > $out/kernels.json 2>> $out/g2.err
find $out -name "*.csv" -size +20M -delete
find $out -name "*_agent_info.csv" -delete
ls $out $out/prof/* 2>/dev/null | head -40
tail -3 $out/pytest.log
