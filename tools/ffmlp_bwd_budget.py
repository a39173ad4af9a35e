#!/usr/bin/env python3
"""Per-phase budget of the fused FFMLP backward: what one 32-row step of ffmlp_backward_fused_kernel<FIELD = 1, 2> is made of.

CPU only (hipcc cross-compiles).  csrc/ffmlp.hip is compiled for gfx950 with the library's own flags + -gline-tables-only (line tables do not change
the code: the kernels' sizes are those of the shipped build), every instruction of the two field kernels is attributed -- through its inline stack
(llvm-symbolizer --inlines) -- to the source line of the KERNEL BODY it was inlined into, and the lines are grouped into the phases of a step:

  load        step bookkeeping, the one-step-ahead prefetch (global_load_lds), ring reads, the glue arithmetic folded into the load stage
  recompute   the forward chain recomputed from the inputs (layer products + activations + operand packing)
  dgrad_out   dL/d(last hidden) = W_out^T . grad^T
  transpose   0/1 selection-matrix MFMAs + packing that turn (lane = row) tensors into (lane = feature) operands of the weight-gradient MFMAs
  wgrad       the weight-gradient MFMAs (accumulators live in registers for the whole kernel)
  act_bwd     activation derivative + packing of the next operand
  dgrad       W_l^T . dPre products of the hidden layers
  dX          the input gradient (products + stores)
and, outside the step loop, prologue (weight fragments into LDS, accumulators) and epilogue (4-wave combine + partial store).

Per phase: static instruction counts by unit and the cycles they need to ISSUE on one SIMD with one wave (MFMA 16x16x32 f16: 16 cycles -- 4 passes,
as SQ_VALU_MFMA_BUSY_CYCLES / SQ_INSTS_MFMA gives in profiles/r05_pmc_ffmlp.txt; other VALU: 4; transcendental: 16; LDS / VMEM: 4 issue cycles each, their
latency not included).  Against the MEASURED cycles per step (--us kernel time, rows, 2.4 GHz) the difference is what the single wave per SIMD spends
waiting on dependencies: MFMA -> VALU -> MFMA chains, LDS fragment reads, the prefetch.

    python tools/ffmlp_bwd_budget.py [--us-colour 60.3 --us-sigma 45.5 --rows 459264] > profiles/r06_ffmlp_bwd_budget.txt
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nerf-texture_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
KERNELS = {"colour (FIELD = 1: 3 hidden layers, 32 -> 64 -> 64 -> 64 -> 16)": "ffmlp_backward_fused_kernelILi64ELi3ELi2ELb1ELi0ELi1ELb0ELb0EE",
           "sigma  (FIELD = 2: 2 hidden layers, 32 -> 64 -> 64 -> 16)": "ffmlp_backward_fused_kernelILi64ELi2ELi2ELb1ELi0ELi2ELb0ELb0EE"}
PHASES = ["prologue", "load", "recompute", "dgrad_out", "transpose", "wgrad", "act_bwd", "dgrad", "dX", "epilogue"]


def phase_of_lines():
    """source line of ffmlp_body.inc (inside ffmlp_backward_fused_kernel) -> phase."""
    src = open(os.path.join(CSRC, "ffmlp_body.inc")).read().split("\n")

    def line_of(text, start=0):
        for i in range(start, len(src)):
            if text in src[i]:
                return i + 1
        raise KeyError(text)

    k0 = line_of("void ffmlp_backward_fused_kernel(")
    loop = line_of("for (; row0 < B; row0 = next, step++) {", k0)
    rec = line_of("// yop[j][t][s] = post-activations", loop)
    dout = line_of("// ---- dL/d(last hidden activation) = W_out^T . grad^T", rec)
    chain = line_of("// A operands (lane = output neuron", dout)
    dw0 = line_of("// ---- first matrix: dW_0 += dPre_0^T . X", chain)
    dx = line_of("if (grad_inputs) {  // dL/dX = W_0^T . dPre_0", dw0)
    end = line_of('if constexpr (RECOMPUTE) asm volatile("s_waitcnt vmcnt(0)"', dx)
    k1 = line_of("// sum the per-workgroup partials", end)
    act0 = line_of("// through the activation of layer NL-1-j", chain)
    out = {}
    for ln in range(k0, k1):
        text = src[ln - 1]
        if ln < loop:
            ph = "prologue"
        elif ln < rec:
            ph = "load"
        elif ln < dout:
            ph = "recompute"
        elif ln < chain:
            ph = "dgrad_out"
        elif ln < dx:
            if re.search(r"mfma16\([^;]*sel(P?[01])\b", text):
                ph = "transpose"
            elif re.search(r"mfma16_acc\([^;]*gw_(out|hid|in)\[", text) or re.search(r"gw_(out|hid|in)\[[^;]*= mfma16", text):
                ph = "wgrad"
            elif "layer_products<OT, KSH, NT>" in text:
                ph = "dgrad"
            elif act0 <= ln < dw0:
                ph = "act_bwd" if ln < line_of("if constexpr (TR) {", act0) else "transpose"
            else:
                ph = "transpose" if ln < act0 or ln >= dw0 else "act_bwd"
        elif ln < end:
            ph = "dX"
        else:
            ph = "epilogue"
        out[ln] = ph
    return out, (k0, k1)


def build(workdir):
    bundle, co = os.path.join(workdir, "ffmlp_dev.o"), os.path.join(workdir, "ffmlp_gfx950.co")
    dry = subprocess.run(["make", "-C", CSRC, "-n", "-W", "ffmlp.hip", "../lib/obj/ffmlp.o"], capture_output=True, text=True, check=True).stdout
    ln = [x for x in dry.splitlines() if " ffmlp.hip" in x and " -c " in x][0]
    cmd = re.split(r'"| 2>|;', ln[ln.index("hipcc"):])[0].split()
    cmd[cmd.index("-c"):cmd.index("-c") + 1] = ["-gline-tables-only", "--cuda-device-only", "-c"]
    cmd[cmd.index("-o") + 1] = bundle
    subprocess.run(cmd, cwd=CSRC, check=True, capture_output=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={bundle}", f"--output={co}"], check=True)
    return co


def unit_of(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("v_accvgpr", "v_mov_b32")):
        return "move"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
        return "trans"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier")):
        return "wait"
    return "salu"


ISSUE = {"mfma": 16, "valu": 4, "move": 4, "trans": 16, "lds": 4, "vmem": 4, "wait": 0, "salu": 0}


def analyse(co, mangled, phase_map, krange):
    syms = subprocess.run([f"{LLVM}/llvm-readelf", "--syms", co], capture_output=True, text=True, check=True).stdout
    name = next(x.split()[-1] for x in syms.splitlines() if mangled in x and not x.split()[-1].endswith((".kd", ".num_vgpr", ".num_agpr")) and " FUNC " in x)
    dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", f"--disassemble-symbols={name}", co], capture_output=True, text=True, check=True).stdout
    ins = []
    for ln in dis.splitlines():
        m = re.match(r"^\s+([a-z_0-9]+)\b.*//\s*([0-9A-F]+):", ln)
        if m:
            ins.append((int(m.group(2), 16), m.group(1)))
    sym = subprocess.run([f"{LLVM}/llvm-symbolizer", "-e", co, "--inlines"], input="\n".join(hex(a) for a, _ in ins) + "\n", capture_output=True, text=True, check=True).stdout
    blocks = [b for b in sym.split("\n\n") if b.strip()]
    assert len(blocks) == len(ins), (len(blocks), len(ins))
    # the step loop by ADDRESS: the backward branch with the longest span (code in front of it that the line table books on the `for` line --
    # the accumulators' initialisation -- is prologue, not part of a step)
    lo = hi = None
    for ln in dis.splitlines():
        m = re.match(r"^\s+s_cbranch_\w+\s+\S+\s+//\s*([0-9A-F]+):.*<[^>]*\+0x([0-9A-Fa-f]+)>", ln) or re.match(r"^\s+s_branch\s+\S+\s+//\s*([0-9A-F]+):.*<[^>]*\+0x([0-9A-Fa-f]+)>", ln)
        if m:
            at, off = int(m.group(1), 16), int(m.group(2), 16)
            target = ins[0][0] + off
            if target < at and (lo is None or at - target > hi - lo):
                lo, hi = target, at
    table = {p: collections.Counter() for p in PHASES}
    for (addr, op), blk in zip(ins, blocks):
        lines = [int(m.group(1)) for m in re.finditer(r"ffmlp_body\.inc:(\d+)", blk)]
        body = [ln for ln in lines if krange[0] <= ln < krange[1]]
        ph = phase_map.get(body[-1], "prologue") if body else "prologue"  # the outermost frame inside the kernel body
        if lo is not None and addr < lo:
            ph = "prologue"
        elif hi is not None and addr > hi:
            ph = "epilogue"
        elif ph in ("prologue", "epilogue"):
            ph = "load"  # (inside the loop by address: loop bookkeeping)
        table[ph][unit_of(op)] += 1
    return table, len(ins)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--us-colour", type=float, default=60.3, help="device time of the colour kernel in the replayed step (profiles/r06_kernel_stats.csv)")
    ap.add_argument("--us-sigma", type=float, default=45.5)
    ap.add_argument("--rows", type=int, default=459264)
    ap.add_argument("--ghz", type=float, default=2.4)
    ap.add_argument("--csrc", default=None, help="another copy of nerf-texture_amd/csrc to analyse (an older revision)")
    ap.add_argument("--co", default=None, help="a prebuilt gfx950 code object of ffmlp.hip with line tables (else it is built: ~2 min)")
    a = ap.parse_args()
    if a.csrc:
        global CSRC
        CSRC = os.path.abspath(a.csrc)
    phase_map, krange = phase_of_lines()
    with tempfile.TemporaryDirectory() as tmp:
        co = a.co or build(tmp)
        steps = a.rows / 32 / 1024  # 256 workgroups x 4 waves: steps per wave
        print(__doc__.split("\n\n")[0])
        print(f"\nworkload: {a.rows} rows = {steps:.2f} 32-row steps per wave (256 workgroups x 4 waves, one wave per SIMD); clock {a.ghz} GHz\n")
        for (label, mangled), us in zip(KERNELS.items(), (a.us_colour, a.us_sigma)):
            table, n = analyse(co, mangled, phase_map, krange)
            print(f"== {label}: {n} static instructions, {us} us in the replayed step")
            hdr = f"{'phase':10s} {'mfma':>5s} {'valu':>5s} {'move':>5s} {'trans':>5s} {'lds':>4s} {'vmem':>4s} {'salu':>5s} {'wait':>4s} | {'issue cycles':>12s} {'of which mfma':>13s}"
            print(hdr)
            tot = collections.Counter()
            step_issue = step_mfma = 0
            for p in PHASES:
                c = table[p]
                issue = sum(ISSUE[u] * c[u] for u in c)
                print(f"{p:10s} {c['mfma']:5d} {c['valu']:5d} {c['move']:5d} {c['trans']:5d} {c['lds']:4d} {c['vmem']:4d} {c['salu']:5d} {c['wait']:4d} | {issue:12d} {16 * c['mfma']:13d}")
                if p not in ("prologue", "epilogue"):
                    step_issue += issue
                    step_mfma += 16 * c["mfma"]
                    tot.update(c)
            measured = us * 1e-6 * a.ghz * 1e9 / steps
            print(f"one step : {tot['mfma']} MFMA, {tot['valu'] + tot['move'] + tot['trans']} other VALU, {tot['lds']} LDS, {tot['vmem']} VMEM instructions -> {step_issue} issue cycles "
                  f"({step_mfma} on the MFMA pipe)")
            print(f"measured : {measured:.0f} cycles per step ({us} us x {a.ghz} GHz / {steps:.2f} steps; prologue + epilogue included: an upper bound)")
            print(f"           issue {100 * step_issue / measured:.0f} % of the step's cycles, MFMA pipe {100 * step_mfma / measured:.0f} %, "
                  f"waiting on dependencies / latency {100 * (1 - step_issue / measured):.0f} %\n")


if __name__ == "__main__":
    sys.exit(main())
