"""Build csrc/liblist_hip.so (the C-ABI library of include/list_hip.h) for gfx950 with hipcc.

In-tree build: the .so sits next to the sources so it travels with the repository snapshot.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(CSRC, "liblist_hip.so")
SOURCES = ["prep_kernels.hip", "gather_kernels.hip", "gather_box_kernels.hip", "gemm_kernels.hip", "fused_fc0_kernels.hip", "bwd_mlp_kernels.hip",
           "bwd_scatter_kernels.hip", "bwd_box_kernels.hip", "bwd_box_split_kernels.hip", "list_capi.hip"]
HEADERS = ["list_common.h", "point_math.h", "gather_math.h", "mfma_common.h", "box_partition.h"]
OBJ_DIR = os.path.join(CSRC, "_obj")
ARCH = "gfx950"


STAMP = LIB + ".stamp"


def _fingerprint():
    """SHA-256 over every source, header and the compile flags: what the library was built FROM, independent of
    file times (a checkout or a copy to the GPU box resets those)."""
    import hashlib
    h = hashlib.sha256()
    h.update((ARCH + "|" + os.environ.get("LIST_HIPCC_FLAGS", "")).encode())
    for d in [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.join(INCLUDE, "list_hip.h")]:
        with open(d, "rb") as f:
            h.update(os.path.basename(d).encode() + b"\0" + f.read())
    return h.hexdigest()


def _stale():
    if not (os.path.exists(LIB) and os.path.exists(STAMP)):
        return True
    with open(STAMP) as f:
        return f.read().strip() != _fingerprint()


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build liblist_hip.so")
    # -ffp-contract=off: coordinates and interpolation weights must round exactly like the
    # reference's CPU ops (an fma of "scale*x - floor" skips a rounding); fmaf is explicit where wanted
    flags = ["-O3", "-std=c++17", "-ffp-contract=off"] + os.environ.get("LIST_HIPCC_FLAGS", "").split() + [
        f"--offload-arch={ARCH}", "-fPIC", "-I", INCLUDE, "-I", CSRC]
    os.makedirs(OBJ_DIR, exist_ok=True)
    tag = f".{os.getpid()}"

    def compile_one(src):
        obj = os.path.join(OBJ_DIR, src.replace(".hip", tag + ".o"))
        cmd = [hipcc] + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        return obj, r

    # one translation unit per worker (the files are independent; hipcc itself is single-threaded)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        results = list(ex.map(compile_one, SOURCES))
    objs = [o for o, _ in results]
    tmp = LIB + f".tmp{os.getpid()}"
    try:
        for (_, r), src in zip(results, SOURCES):
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed on {src}:\n" + r.stdout + r.stderr)
        r = subprocess.run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC"] + objs + ["-o", tmp],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
    finally:
        for o in objs:
            if os.path.exists(o):
                os.remove(o)
    os.replace(tmp, LIB)            # atomic: concurrent readers never see a half-written library
    with open(STAMP + f".tmp{os.getpid()}", "w") as f:
        f.write(_fingerprint() + "\n")
    os.replace(STAMP + f".tmp{os.getpid()}", STAMP)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
