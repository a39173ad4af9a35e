#!/bin/bash
# G2 session: parity of the hash-grid kernels, A/B + ablation of the binning kernels, atomic probe
out=$PWD/gpurun_out/${1:-g2}
mkdir -p $out
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "grid" > $out/pytest_grid.log 2>&1; tail -3 $out/pytest_grid.log
timeout 400 python tools/g2_experiments.py > $out/g2.json 2> $out/g2.err; tail -5 $out/g2.err
timeout 120 tools/probes/_bin/atomic_probe > $out/atomic_probe.txt 2>&1
cat $out/g2.json | head -120
