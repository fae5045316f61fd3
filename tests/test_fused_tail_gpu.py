"""GPU: the single-kernel MLP tail of inference forwards (fc_1 + fc_2 + fc_out, H2 kept in registers; taken when the
caller does not keep the activations: ListQueryArgs.no_activations, fp16 operands) against the two launches a training
forward takes (network/decoders.py:155-163 of the reference: fc_1, fc_2, fc_out on the [B*N, 512] activations).

The fused kernel multiplies the same fp16 products in the same k order and sums the fc_out dot product in the same
tree as the two-launch epilogue, so the bar is bit equality, not a tolerance: a model must answer the same in
eval and under autograd."""
import numpy as np
import pytest
import torch

from oracle import cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import __graft_entry__ as ge
    ge.build()
    from list_amd import hip as h
    h.load()
    return h


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def _prepare(hip, name):
    c = cases.build_case(name)
    md = hip.map_dtype_for("fp16")
    params = {k: dev(v) for k, v in c["weights"].items()}
    img = hip.prep_img_maps([dev(m) for m in c["img_maps"]], dtype=md)
    vox = hip.prep_vox_maps([dev(m) for m in c["vox_maps"]], dtype=md)
    packed = hip.prep_mlp_weights(params, vox.channels, img.channels, "fp16")
    return dev(c["query"]), dev(c["trans_mat"]), img, vox, packed


# tiny: B*N odd and far below one 256-row workgroup; small / real: ragged last workgroup; edge_nan: NaN rows must stay NaN
@pytest.mark.parametrize("sort_points", [True, False])
@pytest.mark.parametrize("name", ["tiny", "small", "real", "edge", "edge_nan"])
def test_fused_tail_equals_the_two_launch_tail_bit_for_bit(hip, monkeypatch, name, sort_points):
    q, tm, img, vox, packed = _prepare(hip, name)
    monkeypatch.setenv("LIST_FUSED_TAIL", "0")
    two = hip.sdf_query(q, tm, img, vox, packed, precision="fp16", sort_points=sort_points).clone()
    monkeypatch.setenv("LIST_FUSED_TAIL", "1")
    one = hip.sdf_query(q, tm, img, vox, packed, precision="fp16", sort_points=sort_points).clone()
    kept, _ctx = hip.sdf_query(q, tm, img, vox, packed, precision="fp16", sort_points=sort_points,
                               save_for_backward=True)
    torch.cuda.synchronize()
    assert torch.equal(torch.isnan(one), torch.isnan(two))
    assert torch.equal(torch.nan_to_num(one), torch.nan_to_num(two))
    assert torch.equal(torch.nan_to_num(one), torch.nan_to_num(kept))


def _bench_inputs(hip, batch):
    """The metric's workload (224^2 images, 128^3 voxels, N = 20 000 uniform queries) at `batch` images."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    inp = bench.make_inputs("list_im2sdf_b8_n20k_224", 0, torch.device("cuda:0"), batch=batch)
    md = hip.map_dtype_for("fp16")
    img = hip.prep_img_maps(inp["img_maps"], inp["map_size"], md)
    vox = hip.prep_vox_maps(inp["vox_maps"], md)
    packed = hip.prep_mlp_weights(inp["weights"], vox.channels, img.channels, "fp16")
    return inp["query"], inp["trans_mat"], img, vox, packed


# B = 1: 79 workgroups of 256 rows, the last one with 32; B = 14: 280 000 rows, two row chunks of the call
# (262 144 + 17 856, hip.query_chunks) -- the fused tail runs once per chunk
@pytest.mark.parametrize("sort_points", [True, False])
@pytest.mark.parametrize("batch", [1, 14])
def test_fused_tail_bit_for_bit_at_the_metric_shapes(hip, monkeypatch, batch, sort_points):
    q, tm, img, vox, packed = _bench_inputs(hip, batch)
    if batch == 14:
        assert hip.query_chunks(q.shape[0] * q.shape[1], packed) == 2
    monkeypatch.setenv("LIST_FUSED_TAIL", "0")
    two = hip.sdf_query(q, tm, img, vox, packed, precision="fp16", sort_points=sort_points).clone()
    monkeypatch.setenv("LIST_FUSED_TAIL", "1")
    one = hip.sdf_query(q, tm, img, vox, packed, precision="fp16", sort_points=sort_points).clone()
    torch.cuda.synchronize()
    assert torch.isfinite(two).all() and float(two.abs().max()) > 0
    assert torch.equal(one, two)


def test_fused_tail_is_the_path_an_inference_forward_takes(hip, monkeypatch):
    """The stage events of a plain fp16 forward show no fc_1 launch (the interval between its two markers is empty:
    ~0.005 ms against ~0.08 ms at B = 8), those of a forward that keeps its activations do."""
    q, tm, img, vox, packed = _bench_inputs(hip, 8)
    monkeypatch.delenv("LIST_FUSED_TAIL", raising=False)
    import ctypes
    from bench import HipEvents
    ev = HipEvents()
    s = hip.STAGE_NAMES.index("fc_1")            # interval s = [event s, event s + 1]
    arr = (ctypes.c_void_p * hip.N_STAGES)(*[ev.create() for _ in range(hip.N_STAGES)])

    def fc1_ms(**kw):
        best = None
        for _ in range(5):
            hip.sdf_query(q, tm, img, vox, packed, precision="fp16", stage_events=arr, **kw)
            torch.cuda.synchronize()
            ms = ev.elapsed_ms(ctypes.c_void_p(arr[s]), ctypes.c_void_p(arr[s + 1]))
            best = ms if best is None else min(best, ms)
        return best

    fused, kept = fc1_ms(), fc1_ms(save_for_backward=True)
    assert fused < 0.5 * kept, (fused, kept)


def test_forward_rejects_a_flag_value_that_is_not_a_flag(hip, monkeypatch):
    """no_activations is 0 or 1.  Anything else is what a struct of another ABI version (or an uninitialised one)
    looks like, and taking it for `true` would run a training forward without writing H1 / H2: refused, loudly."""
    q, tm, img, vox, packed = _prepare(hip, "tiny")
    monkeypatch.setattr(hip, "keeps_no_activations", lambda save_for_backward=False: 2)
    with pytest.raises(hip.ListError, match="no_activations=2") as e:
        hip.sdf_query(q, tm, img, vox, packed, precision="fp16")
    assert e.value.code == hip.ERR_ARG


# The library takes any first hidden width that is a multiple of 256 (the reference's is 2 * h_dim = 512,
# modules.py:196-200); the fused tail streams it in K-tiles of 64: 4, 12 and 16 tiles here instead of 8, around the
# schedule that hands the LDS stages over to W2 behind fc_1's last tiles
@pytest.mark.parametrize("h1", [256, 768, 1024])
def test_fused_tail_bit_for_bit_at_other_first_hidden_widths(hip, monkeypatch, h1):
    from list_amd import synthetic as synth
    c = cases.build_case("small")
    md = hip.map_dtype_for("fp16")
    img = hip.prep_img_maps([dev(m) for m in c["img_maps"]], dtype=md)
    vox = hip.prep_vox_maps([dev(m) for m in c["vox_maps"]], dtype=md)
    F = c["weights"]["fc_0.weight"].shape[1]
    w = dict(synth.make_mlp_weights(91, feature_size=F))
    bound = 1.0 / np.sqrt(h1)
    w["fc_0.weight"] = synth.uniform(9100, (h1, F, 1), -1.0 / np.sqrt(F), 1.0 / np.sqrt(F))
    w["fc_0.bias"] = synth.uniform(9101, (h1,), -1.0 / np.sqrt(F), 1.0 / np.sqrt(F))
    w["fc_1.weight"] = synth.uniform(9102, (256, h1, 1), -bound, bound)
    w["fc_1.bias"] = synth.uniform(9103, (256,), -bound, bound)
    packed = hip.prep_mlp_weights({k: dev(v) for k, v in w.items()}, vox.channels, img.channels, "fp16")
    assert packed.H1 == h1 and packed.H2 == 256
    q, tm = dev(c["query"]), dev(c["trans_mat"])
    monkeypatch.setenv("LIST_FUSED_TAIL", "0")
    two = hip.sdf_query(q, tm, img, vox, packed, precision="fp16").clone()
    monkeypatch.setenv("LIST_FUSED_TAIL", "1")
    one = hip.sdf_query(q, tm, img, vox, packed, precision="fp16").clone()
    torch.cuda.synchronize()
    assert torch.isfinite(two).all() and float(two.abs().max()) > 1e-3
    assert torch.equal(one, two)

