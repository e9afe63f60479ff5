#!/usr/bin/env python3
"""Per-kernel resource usage of the built HIP objects (VGPRs, SGPR / VGPR spills, scratch, LDS, static instruction mix).

    python tools/kernel_resources.py [--match step_kernel] [--isa] [objects ...]

Default objects: the seven translation units of marl-mass_amd/csrc (mm_main.o .. mm_split_general.o).  Unbundles the gfx950 code object from the
.hip_fatbin section (llvm-objcopy + clang-offload-bundler), reads the AMDGPU metadata notes (llvm-readelf) and, with
--isa, counts instruction classes in the disassembly (llvm-objdump).  Build-container tool: no GPU needed.
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def unbundle(obj, td):
    fat = os.path.join(td, os.path.basename(obj) + ".fatbin")
    co = os.path.join(td, os.path.basename(obj) + ".co")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                           "--input=" + fat, "--output=" + co, "--unbundle"])
    return co


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return out.splitlines()


def notes(co):
    txt = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)
    kernels, cur = [], None
    for ln in txt.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", ln)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "agpr_count":
            cur = {}
            kernels.append(cur)
        if cur is not None and k in ("agpr_count", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count",
                                     "private_segment_fixed_size", "group_segment_fixed_size", "name", "max_flat_workgroup_size"):
            cur[k] = v if k == "name" else int(v)
    return [k for k in kernels if "name" in k]


def isa_mix(co, sym):
    txt = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--disassemble-symbols=" + sym, co], text=True)
    c = collections.Counter()
    for ln in txt.splitlines():
        m = re.match(r"\s+([a-z_0-9]+)\s", ln)
        if not m:
            continue
        op = m.group(1)
        c["total"] += 1
        if op.startswith("v_"):
            c["valu"] += 1
            if op.startswith(("v_readlane", "v_writelane")):
                c["lane_spill_moves"] += 1
            if "dpp" in ln:
                c["dpp"] += 1
            if op.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")):
                c["trans_f64"] += 1
            if op.startswith("v_div_fmas_f64"):
                c["div_f64"] += 1
            if op.startswith("v_mfma"):
                c["mfma"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
            if op.startswith("s_cbranch"):
                c["branches"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith(("global_", "flat_", "buffer_")):
            c["vmem"] += 1
        elif op.startswith("scratch_"):
            c["scratch"] += 1
    return dict(c)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("objects", nargs="*")
    ap.add_argument("--match", default="step_kernel")
    ap.add_argument("--isa", action="store_true")
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    objs = args.objects or [os.path.join(REPO, "marl-mass_amd", "csrc", n) for n in ("mm_main.o", "mm_general.o", "mm_ipm.o", "mm_lanes.o", "mm_lanes_ipm.o", "mm_split.o", "mm_split_general.o")]
    rows = []
    with tempfile.TemporaryDirectory() as td:
        for obj in objs:
            if not os.path.exists(obj):
                continue
            co = unbundle(obj, td)
            ks = notes(co)
            names = demangle([k["name"] for k in ks])
            for k, dn in zip(ks, names):
                if args.match not in dn:
                    continue
                short = re.sub(r"\(.*$", "", dn).replace("void ", "")
                row = {"object": os.path.basename(obj), "kernel": short, "vgpr": k.get("vgpr_count"), "agpr": k.get("agpr_count"),
                       "sgpr": k.get("sgpr_count"), "vgpr_spill": k.get("vgpr_spill_count"), "sgpr_spill": k.get("sgpr_spill_count"),
                       "scratch_B": k.get("private_segment_fixed_size"), "lds_B": k.get("group_segment_fixed_size")}
                if args.isa:
                    row["isa"] = isa_mix(co, k["name"])
                rows.append(row)
    for r in rows:
        print("%-14s %-62s vgpr %3d agpr %3d sgpr %3d  spill v %3d s %3d  scratch %4d B  lds %6d B%s" % (
            r["object"], r["kernel"], r["vgpr"], r["agpr"], r["sgpr"], r["vgpr_spill"], r["sgpr_spill"], r["scratch_B"], r["lds_B"],
            ("  " + json.dumps(r["isa"])) if args.isa else ""))
    if args.json:
        json.dump(rows, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    sys.exit(main())
