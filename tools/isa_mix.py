#!/usr/bin/env python3
"""Instruction mix of one kernel in a `hipcc -S --cuda-device-only` dump:  tools/isa_mix.py file.s <substring of the mangled name> [...]"""
import collections
import sys


def analyze(lines, tag):
    start = next((i for i, l in enumerate(lines) if l.startswith("_Z") and tag in l.split(":")[0]), None)
    if start is None:
        print("not found:", tag)
        return
    end = start + 1
    while not lines[end].startswith(".Lfunc_end"):
        end += 1
    ins = [l.strip().split()[0] for l in lines[start + 1:end] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    kinds = collections.Counter("valu" if i.startswith("v_") else "salu" if i.startswith("s_") else "lds" if i.startswith("ds_") else
                                "vmem" if i.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other" for i in ins)
    print(lines[start].split(":")[0][:110])
    print("  static instructions:", len(ins), dict(kinds))
    print("  top:", collections.Counter(ins).most_common(int(sys.argv[-1]) if sys.argv[-1].isdigit() else 30))


if __name__ == "__main__":
    lines = open(sys.argv[1]).read().split("\n")
    for tag in sys.argv[2:]:
        if not tag.isdigit():
            analyze(lines, tag)
