#!/bin/bash
# Everything profiles/r04_* is made from, in one gpurun call (outputs under gpurun_out/<tag>/; tools/collect_profiles_r04.py copies the summaries)
tag=${1:-r04prof}
out=$PWD/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
# 1. the default bench line (roofline + cpu_baseline + rendered + other_config)
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
# 2. rocprofv3 kernel stats of the same command (without the baselines)
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python $R/bench.py --no-cpu-baseline --no-other --no-infer --no-replay-profile --baked-pool > $out/bench_under_rocprof.json 2> $out/prof.err )
# 3. memory-side bytes of the hash-grid kernels (separate passes)
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -- python $R/tools/bench_kernels.py --ops grid_fwd,grid_bwd --rays 8192 --dtypes f16 > $out/pmc_$c.log 2>&1 )
done
python tools/pmc_summary.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE > $out/pmc_grid.txt 2>&1
# 4. L1 / L2 counters of the gather kernels
bash tools/gpu_pmc_l2.sh $tag/l2 > /dev/null 2>&1
# 5. SQ counters: FFMLP kernels (eager bench steps), hash-grid kernels (microbench)
LINES_OUT=400 bash tools/gpu_pmc_any.sh $tag/sq_ffmlp "field_forward|ffmlp_backward|wgrad_reduce" -- $R/bench.py --no-graph --steps 16 --warmup 4 --warm-seconds 0 --no-kernel-timing --no-cpu-baseline --no-other --no-infer --baked-pool > /dev/null 2>&1
LINES_OUT=400 bash tools/gpu_pmc_any.sh $tag/sq_grid "bin_fill_dir|sum_tiles_dir|combine_tiles|grid_forward_level|march_count_parallel|march_rays_kernel" -- $R/tools/bench_kernels.py --ops grid_fwd,grid_bwd,march --rays 8192 --dtypes f16 > /dev/null 2>&1
# 6. per-level cost of G1, G2 experiments, kernel microbench, timeline of the replayed step, inference frame
timeout 200 python tools/g2_experiments.py > $out/g2.json 2> $out/g2.err
timeout 200 python tools/bench_kernels.py --rays 8192 --kernels > $out/kernels.json 2>> $out/g2.err
bash tools/gpu_timeline.sh $tag/timeline > /dev/null 2>&1
timeout 300 python tools/bench_infer.py > $out/infer.json 2> $out/infer.err
timeout 300 python tools/precision_table.py $out/precision.json > /dev/null 2> $out/precision.err
timeout 200 python tools/occupancy_breakdown.py > $out/occupancy.json 2> $out/occupancy.err
find $out -name "*.csv" -size +20M -delete
find $out -name "*_agent_info.csv" -delete
find $out -name "*kernel_trace.csv" -delete
find $out -name "*counter_collection.csv" -delete
ls -R $out | head -60
