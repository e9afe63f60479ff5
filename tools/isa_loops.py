#!/usr/bin/env python3
"""Loops of one kernel in a built object: for every backward branch, the instruction mix of the code it spans.

    python tools/isa_loops.py <object|.so> <kernel-name-substring (demangled)> [--min 200] [--dump LO HI]

Shows where scratch spills, divisions and square roots sit relative to the hot loops (build-container tool).
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_resources as kr  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("obj")
    ap.add_argument("kernel")
    ap.add_argument("--min", type=int, default=200)
    ap.add_argument("--dump", type=int, nargs=2, default=None)
    args = ap.parse_args()
    with tempfile.TemporaryDirectory() as td:
        co = kr.unbundle(args.obj, td)
        ks = kr.notes(co)
        names = kr.demangle([k["name"] for k in ks])
        sym = [k["name"] for k, dn in zip(ks, names) if args.kernel in dn]
        if len(sym) != 1:
            raise SystemExit("kernel match is not unique: %s" % [dn for dn in names if args.kernel in dn])
        txt = subprocess.check_output([os.path.join(kr.LLVM, "llvm-objdump"), "-d", "--disassemble-symbols=" + sym[0], co], text=True)
    ins = []
    for ln in txt.splitlines():
        m = re.match(r"\s+([a-z_0-9]+)\s*(.*?)\s*//\s*([0-9A-F]+):", ln)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
    addr_index = {a: k for k, (a, _, _) in enumerate(ins)}
    if args.dump:
        for k in range(args.dump[0], args.dump[1]):
            print(k, ins[k][1], ins[k][2])
        return
    loops = []
    for k, (a, op, arg) in enumerate(ins):
        if op.startswith("s_cbranch") or op == "s_branch":
            try:
                off = int(arg.split()[-1])
            except ValueError:
                continue
            if off >= 32768:
                off -= 65536
            tgt = a + 4 + 4 * off
            if tgt <= a and tgt in addr_index:
                loops.append((addr_index[tgt], k))
    print("%d instructions, %d backward branches" % (len(ins), len(loops)))
    for lo, hi in sorted(loops):
        n = hi - lo + 1
        if n < args.min:
            continue
        c = collections.Counter()
        for (_, op, arg) in ins[lo:hi + 1]:
            c["valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else
              "scratch" if op.startswith("scratch_") else "vmem"] += 1
            if op.startswith("v_div_fmas"):
                c["div"] += 1
            if op.startswith(("v_rsq_f64", "v_sqrt_f64")):
                c["sqrt"] += 1
            if op.startswith(("v_readlane", "v_writelane")):
                c["lane_spill"] += 1
            if op.startswith("v_cndmask"):
                c["cndmask"] += 1
            if op.startswith("s_cbranch"):
                c["branch"] += 1
        print("loop [%6d .. %6d] %5d instr  %s" % (lo, hi, n, dict(c)))


if __name__ == "__main__":
    main()
