"""CPU: the C-ABI library loads and exports every symbol include/list_hip.h declares
(no compute call is made here: there is no GPU in the authoring container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "list_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(list_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_expected_entry_points():
    names = _declared()
    for must in ("list_sdf_query_fwd", "list_prep_img_maps", "list_prep_vox_maps",
                 "list_prep_mlp_weights", "list_percep_pool_fwd", "list_query_workspace_bytes",
                 "list_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from list_amd import hip
    lib = ctypes.CDLL(hip.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), f"{name} declared in list_hip.h but not exported"
    assert set(hip.EXPORTS) == set(_declared())


def test_abi_version_and_error_string_without_gpu():
    from list_amd import hip
    lib = hip.load()
    assert lib.list_abi_version() == 7
    text = open(os.path.join(ROOT, "include", "list_hip.h")).read()
    assert "#define LIST_ABI_VERSION 7" in text
    assert hip.ABI_VERSION == 7                      # header, library and ctypes structs move together
    # argument validation happens before any HIP call: a NULL args struct is rejected cleanly
    assert lib.list_sdf_query_fwd(None, None) == -1
    assert b"NULL" in lib.list_last_error()
    ws = lib.list_query_workspace_bytes(160000, 3610, 512, 256, 256)
    assert ws > 2 * 160000 * 3648 * 2
    # host-side chunk arithmetic: the metric shape is one chunk, a 256^3 grid is 64 chunks of 262144 rows
    assert lib.list_query_chunk_rows(ws, 160000, 3610, 512, 256, 256) == 160000
    big = lib.list_query_workspace_bytes(256 ** 3, 3610, 512, 256, 256)
    assert lib.list_query_chunk_rows(big, 256 ** 3, 3610, 512, 256, 256) == 262144
    assert lib.list_query_chunk_rows(1 << 20, 256 ** 3, 3610, 512, 256, 256) == 0
    # projected perceptual map (ABI 5): 18769 pixels pad to 74 row tiles; fp16 halfs / fp32 floats; fp16 needs no scratch
    assert lib.list_percep_proj_bytes(1, 137, 512, 2) == 18944 * 512 * 2
    assert lib.list_percep_proj_bytes(1, 137, 512, 0) == 18944 * 512 * 4
    assert lib.list_percep_proj_scratch_bytes(1, 137, 1024, 2) == 0
    assert lib.list_percep_proj_scratch_bytes(1, 137, 1024, 0) == 18769 * 1024 * 4


def test_missing_library_fails_loudly(monkeypatch):
    from list_amd import hip
    monkeypatch.setattr(hip, "_lib", None)
    monkeypatch.setattr(hip, "LIB_PATH", "/nonexistent/liblist_hip.so")
    with pytest.raises(RuntimeError, match="no CPU/PyTorch fallback"):
        hip.load()


def test_binding_refuses_a_library_of_another_abi_version(monkeypatch):
    """The argument structs grow at their end between ABI versions (ABI 6: ListQueryArgs.no_activations,
    ListQueryGradArgs.grad_img_map_dtype): a library reading a longer struct than the binding fills would take those
    flags from stray bytes, so load() compares list_abi_version() with the version its structs mirror."""
    import pytest
    from list_amd import hip
    hip.load()                                       # the real pair agrees
    monkeypatch.setattr(hip, "_lib", None)           # force a fresh load against a binding that claims another version
    monkeypatch.setattr(hip, "ABI_VERSION", hip.ABI_VERSION - 1)
    with pytest.raises(RuntimeError, match="speaks ABI 7, this binding ABI 6"):
        hip.load()
    assert hip._lib is None                          # nothing half-loaded is left behind

