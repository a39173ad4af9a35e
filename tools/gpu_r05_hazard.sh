#!/bin/bash
# Round 5, call 1: (a) the cost of building the whole library without packed fp32 (bench A/B: in-tree vs lib/ab/libnerftex_hip_packed.so),
# (b) the reduction probe of the co-scheduling fault (tools/probes/k3d_reduce.hip) beside neighbours of different kinds, (c) the known-failing
# control (the packed library's hash-grid backward beside a training process), (d) every stage of the step beside a training process on the
# shipped build, (e) the GPU test suite on the new build.
out=$PWD/gpurun_out/r05_hazard
mkdir -p $out
export TMPDIR=/tmp
P=tools/probes/_bin/k3d_reduce
PACKED=nerf-texture_amd/lib/ab/libnerftex_hip_packed.so

# ---- (b) reduction probe.  quiet, then in-process side streams of every kind (high priority), 3000 launches each
: > $out/k3d_reduce.jsonl
for v in 0 1 2 3; do
  timeout 60 $P victim $v 3000 >> $out/k3d_reduce.jsonl 2>&1
  for k in fp32 fp64 mfma pk imul trans lds mem; do
    timeout 60 $P victim $v 3000 --side $k --prio high >> $out/k3d_reduce.jsonl 2>&1
  done
done
# a PROCESS neighbour of each kind
: > $out/k3d_reduce_process.jsonl
for k in fp32 fp64 mfma pk imul trans lds mem; do
  $P neighbour $k 14 > /dev/null 2>&1 &
  nb=$!
  sleep 1.5
  for v in 0 1 2 3; do
    echo -n "{\"process_neighbour\": \"$k\", \"result\": " >> $out/k3d_reduce_process.jsonl
    timeout 30 $P victim $v 3000 >> $out/k3d_reduce_process.jsonl 2>&1
    echo "}" >> $out/k3d_reduce_process.jsonl
  done
  kill $nb 2>/dev/null; wait $nb 2>/dev/null
done
# the neighbour that is KNOWN to trigger the fault in K3d: a training process
rm -f /tmp/ready
READY_FILE=/tmp/ready STEPS=400000 python tools/determinism_probe.py neighbour > /dev/null 2>&1 &
tr=$!
for i in $(seq 1 600); do [ -f /tmp/ready ] && break; sleep 0.2; done
: > $out/k3d_reduce_trainer.jsonl
for v in 0 1 2 3; do
  timeout 60 $P victim $v 6000 >> $out/k3d_reduce_trainer.jsonl 2>&1
  timeout 60 ${P}_nopk victim $v 6000 >> $out/k3d_reduce_trainer_nopk.jsonl 2>&1
done
# (c) the control: full K3d, packed build vs shipped build, beside the same trainer (tools/g2_concurrency_probe.py uses its own neighbour: use none here)
kill $tr 2>/dev/null; wait $tr 2>/dev/null
NERFTEX_HIP_LIB=$PACKED timeout 200 python tools/g2_concurrency_probe.py --neighbour process --launches 2000 > $out/g2_packed_process.json 2> $out/g2_packed_process.err
NERFTEX_HIP_LIB=$PACKED timeout 200 python tools/g2_concurrency_probe.py --neighbour stream --launches 2000 > $out/g2_packed_stream.json 2>> $out/g2_packed_process.err
timeout 200 python tools/g2_concurrency_probe.py --neighbour process --launches 2000 > $out/g2_shipped_process.json 2>> $out/g2_packed_process.err
# (d) every stage of the step, shipped build, beside a trainer
timeout 400 python tools/step_concurrency_probe.py --neighbour process --iters 1500 > $out/step_shipped_process.json 2> $out/step.err

# ---- (a) bench A/B, two rounds each, interleaved
for round in 1 2; do
  for lib in "" $PACKED; do
    tag=$( [ -z "$lib" ] && echo intree || echo packed )
    NERFTEX_HIP_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-replay-profile --no-other --no-infer --no-occupancy-timing 2>/dev/null | grep '^{' | tail -1 >> $out/bench_$tag.jsonl
  done
done
# ---- (e) the GPU suite on the new build
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -3 $out/pytest.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_hazard/bench_*.jsonl")):
    print(f, [round(json.loads(l)["ms_per_step"]*1e3,1) for l in open(f)])
for f in ["k3d_reduce.jsonl","k3d_reduce_trainer.jsonl","k3d_reduce_trainer_nopk.jsonl"]:
    for l in open("gpurun_out/r05_hazard/"+f):
        try:
            j=json.loads(l)
            if j["mismatching_words"] or j["quiet_mismatching_words"]: print(f, j["victim_variant"], j["side"], j["mismatching_words"], j["of_which_zero_pair_masks"])
        except Exception as e: print(f, "unparsed", l[:100])
PY
cat $out/g2_*.json $out/step_shipped_process.json | cut -c1-400
