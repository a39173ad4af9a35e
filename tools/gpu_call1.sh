#!/bin/bash
out=$PWD/gpurun_out/${1:-r3c1}
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_dp_shared_gpu.py > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
timeout 300 python -m pytest tests/test_gpu_round3.py tests/test_march_property.py tests/test_fox_table_anchors.py -m gpu -q > $out/pytest_new.log 2>&1; echo "rc=$?" >> $out/pytest_new.log
timeout 400 python bench.py --no-cpu-baseline --no-other > $out/bench.json 2>> $out/err.log
timeout 200 python tools/bench_kernels.py --rays 8192 --kernels > $out/kernels.json 2>> $out/err.log
tail -5 $out/pytest.log; tail -30 $out/pytest_new.log
