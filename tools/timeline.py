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


def short(name):
    m = re.search(r"list::(k_[a-z_0-9]+)(<[^>]*>)?", name)
    if m:
        return m.group(1) + (m.group(2) or "")
    return "(fill)" if "fill" in name.lower() else "(other)"


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


if __name__ == "__main__":
    main()
