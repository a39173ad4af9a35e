#!/bin/bash
# bash tools/probes/bench_queue_trace.sh : both modes, timing without the log first, then the queue trace
out=$PWD/gpurun_out/r06_queues; mkdir -p $out
for m in late pool; do
  MODE=$m timeout 300 python tools/probes/bench_queue_trace.py 2>&1 >/dev/null | grep "###" > $out/timing_$m.txt
  MODE=$m AMD_LOG_LEVEL=3 timeout 600 python tools/probes/bench_queue_trace.py 2>&1 >/dev/null | grep -E "###|acquireQueue|Selected queue|releaseQueue" | sed -E 's/^.*(acquireQueue|Selected queue|releaseQueue)/\1/' > $out/trace_$m.txt
done
tail -n 100 $out/timing_*.txt
