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
    assert lib.list_abi_version() == 9
    text = open(os.path.join(ROOT, "include", "list_hip.h")).read()
    assert "#define LIST_ABI_VERSION 9" in text
    assert hip.ABI_VERSION == 9                      # header, library and ctypes structs move together
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
    # projected encoder levels (ABI 8), host arithmetic only: the map is [B][137][137][kept_C + H1]; two kept levels of the
    # 224^2 pyramid = 128 sampled channels; a level count that leaves nothing to project, or a channel count that is
    # not a multiple of 64, is refused with a message
    maps = (hip.ListMap2D * hip.N_IMG_LEVELS)()
    keep = ctypes.create_string_buffer(64)
    for i, (c, r) in enumerate(zip((64, 64, 128, 256, 512), (224, 112, 56, 28, 14))):
        maps[i] = hip.ListMap2D(ctypes.addressof(keep), c, r, r, c * r * r, r * r, r, 1)
    assert lib.list_img_proj_map_bytes(maps, 8, 137, 2, 512, 2) == 8 * 137 * 137 * (128 + 512) * 2
    assert lib.list_img_proj_map_bytes(maps, 8, 137, 2, 512, 0) == 8 * 137 * 137 * (128 + 512) * 4
    assert lib.list_img_proj_map_bytes(maps, 8, 137, 0, 512, 0) == 8 * 137 * 137 * 512 * 4
    fp16_scratch = lib.list_img_proj_scratch_bytes(maps, 8, 2, 512, 2)
    rows = [8 * r * r for r in (56, 28, 14)]
    pad = [(x + 255) // 256 * 256 for x in rows]
    assert fp16_scratch == sum(x * c * 2 + p * 512 * 2 for x, p, c in zip(rows, pad, (128, 256, 512)))
    assert lib.list_img_proj_scratch_bytes(maps, 8, 2, 512, 0) > fp16_scratch
    assert lib.list_img_proj_map_bytes(maps, 8, 137, 5, 512, 0) == 0 and b"n_kept_levels" in lib.list_last_error()
    maps[3].C = 200
    assert lib.list_img_proj_scratch_bytes(maps, 8, 2, 512, 0) == 0 and b"multiples of 64" in lib.list_last_error()


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
    with pytest.raises(RuntimeError, match="speaks ABI 9, this binding ABI 8"):
        hip.load()
    assert hip._lib is None                          # nothing half-loaded is left behind

