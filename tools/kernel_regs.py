#!/usr/bin/env python3
"""Register / LDS / scratch use of every kernel in one .hip file (compiles to gfx950 asm, reads the
.amdhsa descriptors).  usage: python tools/kernel_regs.py gather_kernels.hip [extra hipcc flags]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "learning-implicitly-from-spatial-transformers-network_amd", "csrc")


def main():
    src = sys.argv[1]
    if not os.path.isabs(src):
        src = os.path.join(CSRC, src)
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        cmd = ["hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", "-I",
               os.path.join(ROOT, "include"), "-I", CSRC, "-S", "--cuda-device-only", src, "-o", out] + sys.argv[2:]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        s = open(out).read()
    names = [m.group(1) for m in re.finditer(r"\.amdhsa_kernel (\S+)", s)]
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
    for m, nice in zip(re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S), dem):
        body = m.group(2)
        g = lambda k: (re.search(r"\.amdhsa_" + k + r" (\d+)", body) or [0, "?"])[1]
        nice = re.sub(r"^void list::", "", nice)
        nice = re.sub(r"\(.*$", "", nice)
        print(f"{nice:48s} vgpr {g('next_free_vgpr'):>4s} (arch {g('accum_offset'):>3s})  sgpr {g('next_free_sgpr'):>3s}"
              f"  lds {g('group_segment_fixed_size'):>6s}  scratch {g('private_segment_fixed_size'):>4s}")


if __name__ == "__main__":
    main()
