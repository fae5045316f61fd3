"""GPU: fc_0 with the perceptual block of its A operand produced on chip (k_fc0_fused, fused_fc0_kernels.hip) against
the unfused path (k_gather_img writes the block into X, k_gemm_nt_pp reads it back): the same products in the same
order, so the SDF is the same BIT FOR BIT -- on every golden case (non-finite inputs and their exact-redo path
included), sorted and unsorted points, inference and training forwards, and on BASELINE configs 2 and 5 at full size.
Reference call sites replaced: network/modules.py:46-53 (bilinear sample), :275-276 (concat + fc_0 + ReLU)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(tmp_path, tag, mode, full, extra_env=None):
    out = os.path.join(tmp_path, f"fused_{tag}.npz")
    env = dict(os.environ)
    env.pop("LIST_FUSED_FC0", None)
    env.pop("LIST_TAIL_RIDE", None)
    if mode is not None:
        env["LIST_FUSED_FC0"] = mode
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(HERE, "_child_fused_fc0.py"), out] + (["full"] if full else []),
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return np.load(out)


def _same(a, b, what):
    keys = sorted(k for k in a.files if not k.endswith("_fused_fc0"))
    assert keys == sorted(k for k in b.files if not k.endswith("_fused_fc0"))
    assert len(keys) >= 10
    for k in keys:
        x, y = a[k], b[k]
        # bit for bit, NaN positions included (array_equal treats NaN as unequal: compare the raw bits)
        assert x.shape == y.shape and np.array_equal(x.view(np.uint32), y.view(np.uint32)), (what, k, float(np.nanmax(np.abs(x - y))))


def test_fused_fc0_equals_the_unfused_path_bit_for_bit(tmp_path):
    fused = _run(str(tmp_path), "on", None, True)
    plain = _run(str(tmp_path), "off", "0", True)
    assert int(fused["config2_fused_fc0"]) == 1 and int(plain["config2_fused_fc0"]) == 0     # (what the library dispatched)
    assert int(fused["tiny_fused_fc0"]) == 1
    assert int(fused["small_bf16_fused_fc0"]) == 1                     # plain bf16 too
    assert int(fused["config2_bf16x3_fused_fc0"]) == 0                 # bf16x3 stays unfused by default (slower: W is twice as long)
    # a forward that keeps its activations for list_sdf_query_bwd materialises the whole feature matrix (d fc_0.weight
    # reads its perceptual columns): never the fused kernel
    assert int(fused["small_train_plan_fused_fc0"]) == 0
    _same(fused, plain, "fused vs unfused")
    assert np.isfinite(fused["config2_sorted"]).all() and np.abs(fused["config2_sorted"]).max() > 1e-3


def test_fused_fc0_in_bf16x3_equals_the_unfused_path_bit_for_bit(tmp_path):
    """LIST_FUSED_FC0=3 forces the fused kernel for the hi / lo split operands too (off by default: measured slower)."""
    forced = _run(str(tmp_path), "x3", "3", False)
    plain = _run(str(tmp_path), "off", "0", False)
    assert int(forced["small_bf16x3_fused_fc0"]) == 1
    _same(forced, plain, "fused bf16x3 vs unfused")


def test_the_128x512_tile_alone_equals_the_256x256_kernel(tmp_path):
    """LIST_FUSED_FC0=x: the same kernel with every K-tile staged from X (the 2-D gather still runs): the tile shape and
    the schedule do not change a bit."""
    tile = _run(str(tmp_path), "x", "x", False)
    plain = _run(str(tmp_path), "off", "0", False)
    _same(tile, plain, "128x512 tile vs 256x256")


def test_scalar_level_riding_with_the_fine_level_equals_the_tail_kernel(tmp_path):
    """The C = 1 occupancy level, xyz and the zero padding written by the 128^3 x 16 level's gather kernel (gather_pair,
    TailRide) against k_gather_tail (LIST_TAIL_RIDE=0): the same arithmetic, the same bits -- every golden case, the
    non-finite ones and their exact redo included, fp16 / bf16x3 / bf16, sorted and unsorted, and a training forward."""
    ride = _run(str(tmp_path), "ride", None, False)
    tail = _run(str(tmp_path), "tail", None, False, {"LIST_TAIL_RIDE": "0"})
    _same(ride, tail, "riding scalar level vs k_gather_tail")
