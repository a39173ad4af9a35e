#!/bin/bash
out=$PWD/gpurun_out/r05_infer
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 python tools/infer_sweep.py > $out/sweep.json 2> $out/sweep.err
cat $out/sweep.json
( cd /tmp && TRACE=1 timeout 200 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python $R/tools/infer_sweep.py > $out/trace.log 2>&1 )
tail -2 $out/trace.log
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
echo "host-launched frames:" > $out/busy.txt; python tools/busy_from_trace.py $f 45 2 >> $out/busy.txt
echo "graph-replayed frames:" >> $out/busy.txt; python tools/busy_from_trace.py $f 118 78 >> $out/busy.txt
cat $out/busy.txt
find $out -name "*.csv" -size +3M -delete
