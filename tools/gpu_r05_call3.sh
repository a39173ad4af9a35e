#!/bin/bash
out=$PWD/gpurun_out/r05_call3
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_occupancy.py tests/test_gpu_trainstep.py tests/test_gpu_field_glue.py tests/test_gpu_ffmlp.py -q -x -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -15 $out/pytest.log
timeout 200 tools/probes/_bin/pk_mfma_probe 1.5 all 26 > $out/pk_chain_probe.jsonl 2>&1
cat $out/pk_chain_probe.jsonl | cut -c1-330
bash tools/gpu_r05_curved.sh > $out/curved.log 2>&1
tail -12 $out/curved.log | cut -c1-400
NERFTEX_KEEP_PMC=$out/pmc_bench.txt NERFTEX_KEEP_STATS=$out/kernel_stats.csv timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
tail -3 $out/bench.err
python - <<'PY'
import json
j=json.loads([l for l in open("gpurun_out/r05_call3/bench.json") if l.startswith("{")][-1])
print("ms_per_step", j["ms_per_step"], "value", j["value"])
print("occ", {k:v for k,v in (j.get("occupancy_update") or {}).items() if k!="note"})
for o in j.get("other_config") or []:
    print(round(o.get("ms_per_step",0),4), round(o.get("value",0)/1e6,1), o.get("dtype"), o["workload"][:110])
print("rendered", json.dumps(j.get("rendered"))[:600])
PY
