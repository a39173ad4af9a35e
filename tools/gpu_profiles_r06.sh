#!/bin/bash
# Round 6, the FINAL build: the GPU suite, the suite beside a training neighbour (soak), the default bench line (with its own rocprofv3 children:
# kernel stats of the replayed step + FETCH_SIZE / WRITE_SIZE / TCP_TCC_READ_REQ passes), the replayed step's timeline, SQ counters of the MLP
# and hash-grid kernels, L2 counters, the occupancy breakdown.  Outputs under gpurun_out/<tag>/; tools/collect_profiles_r06.py copies the summaries.
tag=${1:-r06final}
out=$PWD/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -4 $out/pytest.log
NERFTEX_KEEP_PMC=$out/pmc_bench.txt NERFTEX_KEEP_STATS=$out/kernel_stats.csv timeout 1200 python bench.py > $out/bench.json 2> $out/bench.err
tail -2 $out/bench.err
bash tools/gpu_timeline.sh $tag/timeline > /dev/null 2>&1
LINES_OUT=10 bash tools/gpu_pmc_any.sh $tag/sq_ffmlp "field_forward|ffmlp_backward|wgrad_reduce" -- $R/bench.py --no-graph --steps 16 --warmup 4 --warm-seconds 0 --no-kernel-timing --no-cpu-baseline --no-other --no-infer --baked-pool > /dev/null 2>&1
LINES_OUT=10 bash tools/gpu_pmc_any.sh $tag/sq_grid "bin_fill_dir|sum_tiles_dir|combine_tiles|grid_forward_level|march_count_parallel|march_rays_kernel" -- $R/tools/bench_kernels.py --ops grid_fwd,grid_bwd,march --rays 8192 --dtypes f16 > /dev/null 2>&1
LINES_OUT=10 bash tools/gpu_pmc_any.sh $tag/sq_adam "sum_tiles_dir|combine_tiles|adam_half|bin_fill_dir" -- $R/tools/tile_adam_probe.py --reps 12 > /dev/null 2>&1
bash tools/gpu_pmc_l2.sh $tag/l2 > /dev/null 2>&1
timeout 200 python tools/occupancy_breakdown.py > $out/occupancy.json 2> $out/occupancy.err
timeout 200 python tools/bench_kernels.py --rays 8192 --kernels > $out/kernels.json 2> $out/kernels.err
timeout 300 python tools/tile_adam_probe.py > $out/tile_adam.json 2> $out/tile_adam.err
timeout 300 python tools/table_update_ab.py > $out/table_update_ab.json 2> $out/table_update_ab.err
timeout 300 python tools/dead_skip_probe.py > $out/dead_skip.json 2> $out/dead_skip.err
timeout 300 python tools/composite_step_ab.py --keeps 1,2,4 > $out/composite_step_ab.json 2> $out/composite_step_ab.err
timeout 200 python tools/composite_step_probe.py 1,2,4 > $out/composite_step_probe.json 2> $out/composite_step_probe.err
# the parity suite beside a training process (any kernel whose result depends on co-scheduling fails an oracle comparison)
bash tools/gpu_soak_beside_neighbour.sh > $out/soak.log 2>&1
cp gpurun_out/soak/pytest_tail.txt $out/soak_pytest_tail.txt 2>/dev/null
tail -4 $out/soak_pytest_tail.txt
find $out -name "*.csv" -size +20M -delete
find $out -name "*_agent_info.csv" -delete
find $out -name "*kernel_trace.csv" -delete
find $out -name "*counter_collection.csv" -delete
TAG=$tag python - <<'PY'
import json
j=json.loads([l for l in open("gpurun_out/" + __import__("os").environ.get("TAG", "r06final") + "/bench.json") if l.startswith("{")][-1])
print("ms_per_step", j["ms_per_step"], "value", j["value"], "incl occ", j.get("value_including_occupancy_update"))
r=j["roofline"]; print("roofline frac", r["frac"], "traffic", r["traffic"], r.get("traffic_over_algorithmic"), r["kernels_avg_us"])
print("G1", {k:r["other"]["grid_encode_forward"].get(k) for k in ("ms","frac_of_hbm_peak","frac_of_l2_peak")})
print("cpu", j["cpu_baseline"])
for o in j.get("other_config") or []:
    print(round(o.get("ms_per_step",0),4), round(o.get("value",0)/1e6,1), o.get("dtype"), o["workload"][:100])
print("rendered", j["rendered"]["mpix_per_s"], j["rendered"].get("graphed",{}).get("ms_per_frame_min_median_max"), j["rendered"]["host_launched"])
PY
