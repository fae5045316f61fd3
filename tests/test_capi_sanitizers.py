"""CPU: the C ABI's HOST side under AddressSanitizer + UndefinedBehaviorSanitizer ("-fsanitize=address
host build of the C ABI").  list_capi.hip is compiled for the host alone, the kernel launchers are aborting stubs
(tests/csrc/capi_host_stubs.cpp), and the driver tests/csrc/capi_host_asan.cpp walks the argument validation, the
workspace carving and the chunk arithmetic with hostile and boundary arguments: NULLs, the 256^3 grid, point counts at
2^31 - 1 / 2^31 / 2^31 + 1 / INT64_MAX, 2^31-element maps, misaligned buffers.  No device call is reached (reaching a stub
aborts).  First run of this build found a signed overflow in list_query_workspace_bytes(INT64_MAX, ...) (n + 255)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_capi_host_side_is_clean_under_asan_and_ubsan(tmp_path):
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "capi_host_asan")
    b = subprocess.run(["bash", os.path.join(ROOT, "tools", "build_capi_host_asan.sh"), exe], capture_output=True,
                       text=True, timeout=900)
    assert b.returncode == 0, b.stdout[-2000:] + b.stderr[-4000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "CAPI_HOST_SANITIZERS_OK" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
