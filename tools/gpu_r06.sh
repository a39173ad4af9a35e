#!/bin/bash
# Round 6's gpurun calls, by mode (one file instead of one script per call):  bash tools/gpu_r06.sh <mode> [tag]
#   tests-new   the round-6 GPU tests + the tests of the files the round touched
#   tests       the whole GPU suite
#   ab          tools/table_update_ab.py
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
  bench)
    timeout 1200 python bench.py > $out/bench.json 2> $out/bench.err; tail -5 $out/bench.err; cat $out/bench.json | head -c 3000 ;;
esac
