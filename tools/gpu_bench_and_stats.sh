#!/bin/bash
# the two judged measurements only: the default bench line and the rocprofv3 kernel statistics of the same command (tools/gpu_profiles_r06.sh has the rest)
tag=${1:-r04final}
out=$PWD/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 100 python bench.py > $out/bench.json 2> $out/bench.err
( cd /tmp && timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python $R/bench.py --no-cpu-baseline --no-other --no-infer --no-replay-profile --baked-pool > $out/bench_under_rocprof.json 2> $out/prof.err )
find $out -name "*_agent_info.csv" -delete; find $out -name "*kernel_trace.csv" -delete
ls -R $out | head -20
