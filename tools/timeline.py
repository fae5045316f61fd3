#!/usr/bin/env python3
"""Print one step of a rocprofv3 kernel trace as a timeline (start/end/duration/stream per kernel).

    python tools/timeline.py gpurun_out/prof/kt [anchor kernel = k_head] [which occurrence = middle]
The step runs from one occurrence of the anchor kernel to the next."""
import csv
import glob
import os
import re
import sys

csv.field_size_limit(1 << 30)


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_prof import short as _short  # noqa: E402  (reads the names rocprofv3 left mangled as well)


def short(name):
    return _short(name) or ("(fill)" if "fill" in name.lower() else "(other)")


def main():
    d = sys.argv[1]
    anchor = sys.argv[2] if len(sys.argv) > 2 else "k_head"
    f = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    names = [short(r["Kernel_Name"]) for r in rows]
    idx = [i for i, n in enumerate(names) if n.startswith(anchor)]
    k = int(sys.argv[3]) if len(sys.argv) > 3 else len(idx) // 2
    i, e = idx[k], (idx[k + 1] if k + 1 < len(idx) else len(rows))
    t0 = int(rows[i]["Start_Timestamp"])
    for j in range(i, e):
        st, en = int(rows[j]["Start_Timestamp"]), int(rows[j]["End_Timestamp"])
        print(f"{names[j]:38s} {(st - t0) / 1e3:8.1f} -> {(en - t0) / 1e3:8.1f}  {(en - st) / 1e3:7.1f} us  stream {rows[j]['Stream_Id']}")
    # the forked phase of the backward: from the end of dX (k_gemm_nt_pp / k_gemm_nt16 with the EPI_DX epilogue, template
    # argument 4) to the last kernel in front of the next step's first layout kernel.  `sum / span` is how many kernels
    # are in flight on average: 1.0 = back to back, whatever the number of streams
    dx = [j for j in range(i, e) if re.match(r"k_gemm_nt(_pp|16)?<(4,|\d+, 4,)", names[j])]
    nxt = [j for j in range(i, e) if names[j].startswith(("k_prep_img", "k_transpose_vox"))]
    if dx and nxt and nxt[0] > dx[0]:
        a, b = dx[0], nxt[0]
        t_a = int(rows[a]["End_Timestamp"])
        span = max(int(rows[j]["End_Timestamp"]) for j in range(a + 1, b)) - t_a
        tot = sum(int(rows[j]["End_Timestamp"]) - int(rows[j]["Start_Timestamp"]) for j in range(a + 1, b))
        print(f"== forked phase behind dX: {b - a - 1} kernels, span {span / 1e6:.3f} ms, sum of their durations {tot / 1e6:.3f} ms, "
              f"sum / span = {tot / span:.2f}")
        for pat in ("k_gemm_tn", "k_scatter_vox<16", "k_scatter_vox_h2", "k_scatter_vox1", "k_vs_gather", "k_scatter_vox_win", "k_scatter_vox_box",
                    "k_img_grad_gather", "k_trans_grad", "k_img_grad_level"):
            ds = [(int(rows[j]["End_Timestamp"]) - int(rows[j]["Start_Timestamp"])) / 1e3 for j in range(a + 1, b) if names[j].startswith(pat)]
            if ds:
                print(f"   {pat:22s} " + " ".join(f"{d:7.1f}" for d in ds) + " us")


if __name__ == "__main__":
    main()
