#!/bin/bash
# hash-grid backward reproducibility beside a training process, for each library build named on the command line
mkdir -p gpurun_out/g2probe
for lib in "$@"; do
  NERFTEX_HIP_LIB=$lib python tools/g2_concurrency_probe.py --neighbour process --launches 2000 2>&1 | grep '^{' | tail -1 | tee -a gpurun_out/g2probe/variants.jsonl
done
