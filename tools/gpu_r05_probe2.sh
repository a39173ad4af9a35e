#!/bin/bash
# Round 5, call 2: per-instruction probe beside MFMA neighbours; side-stream priority A/B of the reduction probe; the bench with its measured-traffic passes
out=$PWD/gpurun_out/r05_probe2
mkdir -p $out
export TMPDIR=/tmp
timeout 300 tools/probes/_bin/pk_mfma_probe 1.0 all > $out/pk_mfma_probe.jsonl 2>&1
for prio in high normal; do
  for v in 0 2; do
    timeout 60 tools/probes/_bin/k3d_reduce victim $v 3000 --side mfma --prio $prio >> $out/k3d_prio.jsonl 2>&1
  done
done
NERFTEX_KEEP_PMC=$out/pmc_bench.txt NERFTEX_KEEP_STATS=$out/kernel_stats.csv timeout 600 python bench.py --no-cpu-baseline --no-other --no-infer > $out/bench.json 2> $out/bench.err
tail -5 $out/bench.err
python - <<'PY'
import json
for l in open("gpurun_out/r05_probe2/pk_mfma_probe.jsonl"):
    try:
        j=json.loads(l); print("%-45s %-26s exec %.2e differ %d threads %d" % (j["instruction"], j["neighbour"], j["lane_executions"], j["differing_pairs"], j["threads_with_a_difference"]))
    except Exception: print(l[:200])
for l in open("gpurun_out/r05_probe2/k3d_prio.jsonl"):
    j=json.loads(l); print(j["victim_variant"], j["side"], j["side_priority"], j["mismatching_words"], j["seconds"])
j=json.loads([l for l in open("gpurun_out/r05_probe2/bench.json") if l.startswith("{")][-1])
print(j["ms_per_step"], json.dumps(j["roofline"])[:1500])
PY
