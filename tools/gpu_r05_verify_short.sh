#!/bin/bash
# the driver's three steps on the final commit (GPU suite, smoke, bench as the driver calls it), without the soak of gpu_r05_verify.sh
out=$PWD/gpurun_out/r05verify
mkdir -p $out
export TMPDIR=/tmp
timeout 520 python -m pytest tests -m gpu -q -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -3 $out/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_style.json 2> $out/bench.err; tail -2 $out/bench.err
python - <<'PY'
import json
j=json.loads([l for l in open("gpurun_out/r05verify/bench_driver_style.json") if l.startswith("{")][-1])
print({k:j[k] for k in ("metric","value","unit","n_gpus","steps","warmup","ms_per_step","dtype","scaling","vs_baseline")})
print("roofline", {k:j["roofline"][k] for k in ("bound","achieved","peak","frac","traffic")}, "cpu", j["cpu_baseline"]["value"], j["cpu_baseline"]["cores"])
print("rendered", j["rendered"]["mpix_per_s"], j["rendered"]["mpix_per_s_is"])
PY
