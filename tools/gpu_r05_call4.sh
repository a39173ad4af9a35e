#!/bin/bash
out=$PWD/gpurun_out/r05_call4
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ffmlp.py tests/test_gpu_round5.py -q -x -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -8 $out/pytest.log
timeout 200 tools/probes/_bin/pk_mfma_probe 1.0 all 32 > $out/pk_pair_probe.jsonl 2>&1
cut -c1-300 $out/pk_pair_probe.jsonl
timeout 600 python tools/pipeline_adam_ab.py --groups 0,2,4,8 --steps 208 --rounds 2 > $out/pipeline_ab.json 2> $out/pipeline_ab.err
cat $out/pipeline_ab.json
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python $R/tools/pipeline_adam_ab.py --groups 4 --steps 64 --rounds 1 > $out/trace.log 2>&1 )
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python tools/overlap_from_trace.py $f adam_half_kernel sum_tiles_dir_kernel 256 > $out/pipeline_overlap.txt 2>&1
python tools/overlap_from_trace.py $f adam_half_kernel combine_tiles_kernel 256 >> $out/pipeline_overlap.txt 2>&1
python - "$f" >> $out/pipeline_overlap.txt <<'PY'
import csv, sys, re
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "bin_fill_dir_kernel" in r["Kernel_Name"]]
a = idx[-6]
t0 = int(rows[a]["Start_Timestamp"])
print("one step from its bin_fill on (start us, duration us, queue, kernel):")
for r in rows[a:a + 16]:
    n = re.sub(r"<.*", "", re.sub(r"void |\(anonymous namespace\)::|nerftex::|gridenc::|ffmlp_f16::", "", r["Kernel_Name"]))[:48]
    print("%8.1f %7.1f  q%s  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", "?"), n))
PY
cat $out/pipeline_overlap.txt
find $out -name "*.csv" -size +3M -delete
