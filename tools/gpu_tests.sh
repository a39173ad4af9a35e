#!/bin/bash
# full GPU suite (no -x)
out=$PWD/gpurun_out/${1:-tests}
mkdir -p $out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" $out/pytest.log | tail -30
