#!/usr/bin/env python3
"""Static instruction budget of bin_fill_dir_kernel (the record builder of the hash-grid backward, K3d) by phase.

A throw-away copy of csrc/gridencoder_binned.hip + grid_record.hpp gets a comment marker (`asm volatile("; PHASE n")`) at every phase boundary, is
compiled to gfx950 assembly with the Makefile's flags, and the instructions between consecutive markers of the fp16 D = 3 instantiation are
counted by kind.  Markers are scheduling fences for volatile operations only; the compiler may still move arithmetic across them, so the
split is approximate (the TOTAL is compared with the unmarked build).  Straight-line counts: the merge steps and the copy-out loop execute
their bodies a data-dependent number of times -- the dynamic figures are the SQ counters of profiles/r04_pmc_sq_grid.txt and the phase
ablation times of profiles/r04_g2_phase_ablation.json, printed beside them.

    python tools/g2_phase_budget.py > profiles/r05_g2_phase_budget.txt
"""
import collections
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nerf-texture_amd", "csrc")
FLAGS = "-O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -Xclang -target-feature -Xclang -packed-fp32-ops -fno-slp-vectorize -mllvm -disable-vector-combine".split()

MARKS = [  # (file, text to find, marker inserted BEFORE it, phase that starts there)
    ("gridencoder_binned.hip", "    if (threadIdx.x < kMaxTilesPerLevel) hist[threadIdx.x] = 0;", "1 loads (coordinates, gradient) + level constants"),
    ("gridencoder_binned.hip", "    const IndexFn<D> index_of(gridtype, align_corners, hashmap_size, lc.resolution[level]);", "2 make_sample: positions, hash terms, weights, pairability"),
    ("grid_record.hpp", "    // head of a run: the previous lane (same 16-lane row) is not a valid sample of the same cell", "3 make_sample: run heads (DPP), w*g products, merge / split"),
    ("gridencoder_binned.hip", "    // ---- count per tile.", "4 count per tile (LDS atomics, rank kept)"),
    ("gridencoder_binned.hip", "    if (threadIdx.x < kWave) {  // wave 0: exclusive prefix over the level's tiles", "5 scan + directory (wave 0)"),
    ("gridencoder_binned.hip", "    // ---- place: LDS for the first kStageRecords slots of the block", "6 record build + placement in LDS"),
    ("gridencoder_binned.hip", "    const uint32_t total = min(lbase[kMaxTilesPerLevel], stage_cap);", "7 copy-out of the block (LDS -> region)"),
]


def main():
    tmp = tempfile.mkdtemp(prefix="g2budget_")
    for f in os.listdir(CSRC):
        if f.endswith((".hpp", ".hip", ".inc")):
            shutil.copy(os.path.join(CSRC, f), tmp)
    os.makedirs(os.path.join(tmp, "..", "..", "include"), exist_ok=True)
    inc = os.path.join(os.path.dirname(os.path.dirname(tmp)), "include")
    # common.hpp includes "../../include/nerftex_hip.h": give the copy the same relative layout
    base = tempfile.mkdtemp(prefix="g2budget_root_")
    work = os.path.join(base, "a", "b")
    shutil.copytree(tmp, work)
    os.makedirs(os.path.join(base, "include"))
    shutil.copy(os.path.join(ROOT, "include", "nerftex_hip.h"), os.path.join(base, "include"))

    def compile_asm(marked):
        for fname in ("gridencoder_binned.hip", "grid_record.hpp"):
            src = open(os.path.join(CSRC, fname)).read()
            if marked:
                for f, needle, phase in MARKS:
                    if f == fname:
                        assert src.count(needle) == 1, (fname, needle)
                        src = src.replace(needle, f'    asm volatile("; PHASE {phase}");\n' + needle)
            open(os.path.join(work, fname), "w").write(src)
        out = os.path.join(work, "k.s")
        subprocess.run(["hipcc", *FLAGS, "--cuda-device-only", "-S", "gridencoder_binned.hip", "-o", out], cwd=work, check=True, capture_output=True)
        lines = open(out).read().split("\n")
        start = next(i for i, l in enumerate(lines) if re.match(r"_ZN7nerftex7gridenc\S*bin_fill_dir_kernelIDF16_Li3ELb1E\S*:", l))
        end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
        return lines[start + 1:end]

    def kind(op):
        return ("valu" if op.startswith("v_") else "salu" if op.startswith("s_") and not op.startswith(("s_waitcnt", "s_nop", "s_barrier")) else
                "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "wait/barrier/nop")

    plain = [l.split()[0] for l in compile_asm(False) if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    body = compile_asm(True)
    phases = collections.OrderedDict()
    cur = "0 prologue: block -> (chunk, level), table check, region pointers"
    seen = []
    for l in body:
        m = re.search(r"; PHASE (.*)$", l)
        if m:
            cur = m.group(1).strip()
            continue
        if l.startswith("\t") and not l.strip().startswith((".", ";")):
            if cur not in phases:
                phases[cur] = collections.Counter()
                seen.append(cur)
            phases[cur][kind(l.split()[0])] += 1
    total_marked = sum(sum(c.values()) for c in phases.values())
    print("# bin_fill_dir_kernel<half, D = 3, [B, L*C] gradients>: static instructions by phase (gfx950, the Makefile's flags; tools/g2_phase_budget.py)")
    print(f"# unmarked build: {len(plain)} instructions ({sum(1 for o in plain if o.startswith('v_'))} VALU); with the phase markers: {total_marked}")
    print(f"# {'phase':78s} {'VALU':>6s} {'SALU':>6s} {'LDS':>5s} {'VMEM':>5s} {'wait':>5s}  share of VALU")
    valu_total = sum(c["valu"] for c in phases.values())
    for name, c in phases.items():
        print(f"  {name:78s} {c['valu']:6d} {c['salu']:6d} {c['lds']:5d} {c['vmem']:5d} {c['wait/barrier/nop']:5d}  {100.0 * c['valu'] / valu_total:5.1f} %")
    print("# (make_sample is compiled three times -- IndexFn modes 0 / 1 / 2, one body per level kind -- so phases 2 and 3 hold ~3x what one workgroup executes;")
    print("#  the merge steps of phase 3 run only in waves that have a follower lane; the copy-out loop of phase 7 runs total / 2048 times)")
    abl = os.path.join(ROOT, "profiles", "r04_g2_phase_ablation.json")
    if os.path.exists(abl):
        a = json.load(open(abl))
        t5, t4, t1, t2, t3, full = a["probe_k3phase5"], a["probe_k3phase4"], a["probe_k3phase1"], a["probe_k3phase2"], a["probe_k3phase3"], a["default"]["bin_fill"]
        print("#\n# dynamic: device time of the kernel cut short after each phase (profiles/r04_g2_phase_ablation.json, alone on the GPU, 456 k samples x 16 levels):")
        for name, us in (("launch only (7168 workgroups of 1024 threads)", t5), ("+ loads", t4 - t5), ("+ make_sample (phases 2-3)", t1 - t4), ("+ count + scan + directory (4-5)", t2 - t1),
                         ("+ record build + placement (6)", t3 - t2), ("+ copy-out (7)", full - t3)):
            print(f"#   {name:52s} {us:6.1f} us  ({100.0 * us / full:4.1f} %)")
        print(f"#   total {full:.1f} us alone; 107.9 us in step (profiles/r05 kernel stats); SQ: 393 VALU instructions per thread-wave, ~60 % VALU issue (profiles/r04_pmc_sq_grid.txt)")
    shutil.rmtree(tmp, ignore_errors=True)
    shutil.rmtree(base, ignore_errors=True)


if __name__ == "__main__":
    main()
