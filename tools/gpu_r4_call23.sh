#!/bin/bash
cd gpurun_out && rm -rf detprobe && cd ..
export RELEASE=1
python tools/determinism_probe.py solo
for r in 1 2; do python tools/determinism_probe.py a$r & python tools/determinism_probe.py b$r & python tools/determinism_probe.py c$r & wait; done
NERFTEX_POISON_WORKSPACE=255 python tools/determinism_probe.py poison255
NERFTEX_POISON_WORKSPACE=0 python tools/determinism_probe.py poison0
cd gpurun_out/detprobe; for f in run_a* run_b* run_c* run_poison*; do cmp -s run_solo.txt $f && echo "$f same" || (echo "$f DIFFERS at:"; diff run_solo.txt $f | head -4 | cut -c1-200); done
