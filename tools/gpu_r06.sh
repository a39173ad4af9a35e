#!/bin/bash
# Round 6's gpurun calls, by mode (one file instead of one script per call):  bash tools/gpu_r06.sh <mode> [tag]
#   tests-new   the round-6 GPU tests + the tests of the files the round touched
#   tests       the whole GPU suite
#   ab          tools/table_update_ab.py
#   ab-prof     the same under rocprofv3 --kernel-trace --stats
#   trained     bench.py's 'trained state' entry alone
#   comp        tools/composite_step_ab.py (the step's compositing as one launch against the three launches, per composite_keep)
#   comp-prof   the same under rocprofv3 --kernel-trace --stats (KEEPS=1,2,4)
#   probe / dead   tools/tile_adam_probe.py / tools/dead_skip_probe.py
#   bench       python bench.py (default run)
mode=${1:-tests-new}
tag=${2:-r06_$mode}
out=$PWD/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
case $mode in
  tests-new)
    timeout 1200 python -m pytest tests/test_gpu_round6.py tests/test_gpu_trainstep.py -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
    tail -25 $out/pytest.log ;;
  tests)
    timeout 1700 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
    tail -15 $out/pytest.log ;;
  ab)
    timeout 600 python tools/table_update_ab.py > $out/ab.json 2> $out/ab.err; tail -3 $out/ab.err; cat $out/ab.json ;;
  ab-prof)   # per-kernel durations of both forms of the step (one rep each), replayed graphs included
    ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python $GRAFT_REPO_ROOT/tools/table_update_ab.py --reps 1 --steps 416 > $out/ab.json 2> $out/ab.err )
    find $out -name "*_kernel_trace.csv" -size +40M -delete; find $out -name "*_agent_info.csv" -delete
    f=$(find $out/prof -name "*kernel_stats.csv" | head -1); head -30 $f | cut -c1-200 ;;
  comp)     # the one-launch compositing of a training step against the three launches, per composite_keep
    timeout 900 python tools/composite_step_ab.py > $out/comp.json 2> $out/comp.err; tail -3 $out/comp.err; cat $out/comp.json ;;
  comp-prof)
    ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python $GRAFT_REPO_ROOT/tools/composite_step_ab.py --reps 1 --keeps ${KEEPS:-2} > $out/comp.json 2> $out/comp.err )
    find $out -name "*_kernel_trace.csv" -size +40M -delete; find $out -name "*_agent_info.csv" -delete
    f=$(find $out/prof -name "*kernel_stats.csv" | head -1); grep -i "composite\|render_tail\|reduce2\|Name" $f | cut -c1-260 ;;
  probe)    # the tile-owner Adam alone, per mode of the grid_adam_mode knob
    timeout 600 python tools/tile_adam_probe.py > $out/probe.json 2> $out/probe.err; tail -3 $out/probe.err; cat $out/probe.json ;;
  trained)  # only the 'trained state' entry of the bench (train against rendered targets, then time with / without the dead-sample skip)
    timeout 900 python - > $out/trained.json 2> $out/trained.err <<'PY'
import json, sys, torch
sys.argv = [sys.argv[0]]
import bench
from ngp_harness import scene
args = bench.parse()
sc = scene.Scene(bound=args.bound, seed=0)
grid, _, _ = sc.bitfield()
print(json.dumps(bench.measure_trained_state(args, torch.device("cuda:0"), sc, grid), indent=1))
PY
    tail -5 $out/trained.err; cat $out/trained.json ;;
  dead)     # the dead-sample skip against the dead fraction
    timeout 600 python tools/dead_skip_probe.py > $out/dead.json 2> $out/dead.err; tail -3 $out/dead.err; python - <<PY
import json
d = json.load(open("$out/dead.json"))
for r in d["rows"]:
    print(r["density_scale"], r.get("dead_step_fraction"), "mlp", r["plain"]["mlp_backward_us"], "->", r["skip"]["mlp_backward_us"], " bin_fill", r["plain"]["record_builder_us"], "->", r["skip"]["record_builder_us"], r["skip"]["kernels_avg_us"].get("composite_tail_bwd_kernel"), r["plain"]["kernels_avg_us"].get("composite_tail_bwd_kernel"))
PY
    ;;
  bench)
    timeout 1200 python bench.py > $out/bench.json 2> $out/bench.err; tail -5 $out/bench.err; cat $out/bench.json | head -c 3000 ;;
esac
