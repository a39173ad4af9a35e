#!/bin/bash
# kernel-by-kernel order of one eager training step (rocprofv3 --kernel-trace), for finding the framework launches between the library's
tag=${1:-trace}
out=$PWD/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/prof -- python $GRAFT_REPO_ROOT/bench.py --no-graph --steps 4 --warmup 20 --no-kernel-timing --no-cpu-baseline --no-other --no-infer > $out/bench.json 2> $out/err.log )
f=$(find $out/prof -name "*kernel_trace.csv" | head -1)
python - "$f" > $out/step.txt <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "near_far_kernel" in n]
a, b = idx[-3], idx[-2]   # one full step between two near_far launches
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    n = re.sub(r"void |at::native::|\(anonymous namespace\)::|nerftex::", "", r["Kernel_Name"])[:110]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  {n}")
PY
find $out -name "*.csv" -size +5M -delete
