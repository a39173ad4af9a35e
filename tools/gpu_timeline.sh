#!/bin/bash
# timeline of the REPLAYED step (rocprofv3 --kernel-trace of the default bench): per step, kernel by kernel with start offset, duration, queue,
# and the idle time between consecutive kernels on the main queue
tag=${1:-timeline}
out=$PWD/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d $out/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 32 --warmup 16 --warm-seconds 0 --baked-pool --no-kernel-timing --no-cpu-baseline --no-other --no-infer ${EXTRA} > $out/bench.json 2> $out/err.log )
f=$(find $out/prof -name "*kernel_trace.csv" | head -1)
python - "$f" > $out/step.txt <<'PY'
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# the timed region = the last 32 adam launches; take steps 8..24 of it
adam = [i for i, n in enumerate(names) if "adam_half_kernel" in n]
a, b = adam[-20], adam[-4]
sel = rows[a + 1:b + 1]
t0 = int(sel[0]["Start_Timestamp"])
qkey = "Queue_Id" if "Queue_Id" in sel[0] else "Stream_Id"
queues = collections.Counter(r[qkey] for r in sel)
main = queues.most_common(1)[0][0]
steps = 16
wall = (int(sel[-1]["End_Timestamp"]) - t0) / 1e3 / steps
busy = collections.defaultdict(float); cnt = collections.Counter(); gap = 0.0; last_end = None; gaps = collections.defaultdict(float)
for r in sel:
    n = re.sub(r"void |at::native::|\(anonymous namespace\)::|nerftex::|ffmlp_f16::", "", r["Kernel_Name"])
    n = re.sub(r"<.*", "", n)[:60]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy[(r[qkey] == main, n)] += (e - s) / 1e3; cnt[(r[qkey] == main, n)] += 1
    if r[qkey] == main:
        if last_end is not None and s > last_end:
            gap += (s - last_end) / 1e3; gaps[n] += (s - last_end) / 1e3
        last_end = max(last_end or 0, e)
print(f"wall per step {wall:.1f} us; queues {dict(queues)}; main-queue idle per step {gap / steps:.1f} us")
print("main queue: kernel, launches/step, us/step, idle before it us/step")
for (m, n), v in sorted(busy.items(), key=lambda kv: -kv[1]):
    print(f"{'main' if m else 'side'} {n:60s} {cnt[(m, n)] / steps:5.2f} {v / steps:8.2f} {gaps.get(n, 0) / steps if m else 0:7.2f}")
print(f"sum main {sum(v for (m, n), v in busy.items() if m) / steps:.1f}  sum side {sum(v for (m, n), v in busy.items() if not m) / steps:.1f}")
# one step in order
idx = [i for i, r in enumerate(sel) if "adam_half_kernel" in r["Kernel_Name"]]
for r in sel[idx[4] + 1: idx[5] + 1]:
    n = re.sub(r"void |at::native::|\(anonymous namespace\)::|nerftex::|ffmlp_f16::", "", r["Kernel_Name"]); n = re.sub(r"<.*", "", n)[:70]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - int(sel[idx[4]]['End_Timestamp'])) / 1e3:9.1f} {(e - s) / 1e3:7.1f} {'M' if r[qkey] == main else 's'} {n}")
PY
find $out -name "*.csv" -size +5M -delete
head -60 $out/step.txt
