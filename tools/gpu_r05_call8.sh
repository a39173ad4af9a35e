#!/bin/bash
out=$PWD/gpurun_out/r05_call8
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round5.py -q -x -p no:cacheprovider -k "accelerate or round5 or record_builder or checkpoint" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -5 $out/pytest.log
python - > $out/c1.json 2> $out/c1.err <<'PY'
import json, sys, torch
sys.argv=[sys.argv[0]]
import bench
from ngp_harness import scene
args=bench.parse()
dev=torch.device("cuda:0"); torch.cuda.set_device(dev)
grid,_,_=scene.Scene(bound=args.bound, seed=0).bitfield()
out={}
for k in range(2):
    r=bench.measure_accelerated(args,"torch",4096,32,dev,grid)
    out.setdefault("configs[1] through accelerate (HalfLeafAdam + FusedAmp), ms_per_step",[]).append(round(r["ms_per_step"],4))
    out.setdefault("value M/s",[]).append(round(r["value"]/1e6,1))
print(json.dumps(out))
PY
cat $out/c1.json; tail -3 $out/c1.err
