"""Per-kernel averages of the FFMLP PMC passes of tools/gpu_pmc_ffmlp.sh (what profiles/*pmc_ffmlp.txt holds).

    python tools/pmc_ffmlp_summary.py gpurun_out/<tag>
"""
import collections
import csv
import glob
import re
import sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "ffmlp" not in n:
            continue
        m = re.search(r"(ffmlp_\w+?_kernel)I(.*?)E+v", n)
        key = n[:60] if not m else m.group(1) + "<" + m.group(2).replace("Li", "").replace("Lb", "b").replace("E", ",").rstrip(",") + ">"
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key in sorted(acc):
    c = {k: sum(v) / len(v) for k, v in acc[key].items()}
    waves = c.get("SQ_WAVES", 0)
    line = f"{key}\n   launches {len(next(iter(acc[key].values())))}  waves {waves:.0f}"
    if waves:
        line += (f"  MFMA instr/wave {c.get('SQ_INSTS_MFMA', 0) / waves:.1f}  VALU instr/wave {c.get('SQ_INSTS_VALU', 0) / waves:.1f}"
                 f"  MFMA busy / SQ busy {c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(c.get('SQ_BUSY_CYCLES', 1), 1):.4f}"
                 f"  wave-cycles parked (WAIT_ANY/WAVE_CYCLES) {c.get('SQ_WAIT_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):.2f}"
                 f"  wave-cycles/wave {c.get('SQ_WAVE_CYCLES', 0) / waves:.0f}")
    if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
        line += f"\n   FETCH_SIZE {c.get('FETCH_SIZE', 0) * 1024 / 1e6:.1f} MB as reported (x2 for coalesced streams on gfx950)  WRITE_SIZE {c.get('WRITE_SIZE', 0) * 1024 / 1e6:.1f} MB"
    print(line)
