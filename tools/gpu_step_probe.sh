#!/bin/bash
mkdir -p gpurun_out/g2probe
for lib in "" nerf-texture_amd/lib/ab/libnerftex_hip_slp.so; do
  for nb in none process; do
    NERFTEX_HIP_LIB=$lib python tools/step_concurrency_probe.py --neighbour $nb --iters 200 2>&1 | grep -v amdgpu | tail -3 | tee -a gpurun_out/g2probe/step_results2.jsonl
  done
done
