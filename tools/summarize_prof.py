#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel trace / stats / counter collection) into a short text
summary for profiles/.  Only this repository's kernels (namespace list::) are listed by name;
everything else (torch RNG kernels that build the synthetic inputs, copies) is lumped together.

    python tools/summarize_prof.py gpurun_out/prof/kt gpurun_out/prof/fetch gpurun_out/prof/write
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def _mangled(name):
    """`_ZN4list<len><kernel>[I<template args>E]...` -> 'k_name<a, b, ...>' (integer and bool arguments), or None.
    rocprofv3 leaves a name mangled when its demangler does not know a parameter type (`_Float16*` = PDF16_: the
    LDS-window and packed-half scatter kernels of the backward), and a summary keyed on demangled names only would
    silently file those kernels under "other"."""
    m = re.match(r"_ZN4list(\d+)", name)
    if not m:
        return None
    n, i = int(m.group(1)), m.end()
    kernel, i = name[i:i + n], i + n
    if not re.fullmatch(r"k_[a-z_0-9]+", kernel):
        return None
    args = []
    if name[i:i + 1] == "I":
        i += 1
        while i < len(name) and name[i] != "E":
            a = re.match(r"L([a-z])(n?)(\d+)E", name[i:])
            if not a:
                return kernel + "<?>"
            v = ("-" if a.group(2) else "") + a.group(3)
            args.append({"0": "false", "1": "true"}[a.group(3)] if a.group(1) == "b" else v)
            i += a.end()
    return kernel + (f"<{', '.join(args)}>" if args else "")


def short(name):
    name = name.strip('"')
    m = re.search(r"list::(k_[a-z_0-9]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else _mangled(name)


def kernel_trace(d):
    rows = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"]) or "(other: torch/rocclr)"
            dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            # every fc_0 launch is followed by a gated re-run of the same kernel (exact border semantics) that
            # exits at its first instruction on finite inputs: listed on its own line
            if k.startswith("k_gemm_nt") and dur < 15000:
                k += " [gated exit]"
            elif k.startswith("k_gemm"):          # one template serves several layers: tell them apart by their grid
                k += f" grid {r.get('Grid_Size_X', r.get('Grid_Size', ''))}"
            rows[k].append((dur, r))
    return rows


def counters(d):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                if k.startswith("k_gemm"):
                    k += f" grid {r.get('Grid_Size_X', r.get('Grid_Size', ''))}"
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def main():
    out = []
    for d in sys.argv[1:]:
        kt = kernel_trace(d)
        if kt:
            out.append(f"== kernel trace: {d}")
            out.append(f"{'kernel':46s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} "
                       f"{'total_ms':>10s} {'vgpr':>5s} {'lds':>7s} {'grid':>9s} {'wg':>5s}")
            for k, v in sorted(kt.items(), key=lambda kv: -sum(t for t, _ in kv[1])):
                ts = [t for t, _ in v]
                r = v[-1][1]
                out.append(f"{k:46s} {len(ts):6d} {sum(ts)/len(ts)/1e3:10.1f} {min(ts)/1e3:10.1f} "
                           f"{max(ts)/1e3:10.1f} {sum(ts)/1e6:10.3f} {r.get('VGPR_Count',''):>5s} "
                           f"{r.get('LDS_Block_Size',''):>7s} {r.get('Grid_Size_X', r.get('Grid_Size','')):>9s} "
                           f"{r.get('Workgroup_Size_X', r.get('Workgroup_Size','')):>5s}")
        cs = counters(d)
        if cs:
            out.append(f"== counters: {d}  (per-dispatch mean; FETCH_SIZE/WRITE_SIZE in KB as reported)")
            for k, c in sorted(cs.items()):
                for name, vals in sorted(c.items()):
                    out.append(f"{k:46s} {name:14s} n={len(vals):4d} mean={sum(vals)/len(vals):14.1f} "
                               f"max={max(vals):14.1f}")
    print("\n".join(out))


if __name__ == "__main__":
    main()
