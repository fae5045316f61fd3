"""GPU: the point sort under key pile-up.  An untrained spatial transformer projects most query points onto the clamp
of network/modules.py:43 (bench.py --whole-model: 88 %), i.e. thousands of points share a handful of pixel bins.  The
sort groups equal keys inside a wave into one atomic (gather_kernels.hip, wave_grouped_add); this checks that the orders
it builds are still permutations that group equal keys, and that the sorted forward equals the unsorted one bit for bit,
for all-on-one-pixel, edge-piled and spread projections -- and with a point count that leaves the last wave partly idle."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


@pytest.fixture(scope="module")
def hip():
    import __graft_entry__ as ge
    ge.build()
    from list_amd import hip as h
    return h


@pytest.mark.parametrize("n_points", [5000, 4993])
def test_piled_up_projections_sort_like_spread_ones(hip, n_points):
    from list_amd import synthetic as synth
    seed, B, N = 515, 3, n_points
    q = synth.make_query(seed, B, N)
    T = synth.make_trans_mat(seed, B)
    T[0] = np.array([[0, 0, 0], [0, 0, 0], [0, 0, 0], [-5, -7, 1]], np.float32)          # every point onto pixel (0, 0)
    T[1] = np.array([[400, 0, 0], [0, 3, 0], [0, 0, 0], [68, 68, 1]], np.float32)         # u clamps to 0 / 136, v spreads a little
    md = hip.map_dtype_for("fp16")
    img = hip.prep_img_maps([dev(m) for m in synth.make_img_maps(seed, B, 224)], dtype=md)
    vox = hip.prep_vox_maps([dev(m) for m in synth.make_vox_maps(seed, B, 128)], dtype=md)
    packed = hip.prep_mlp_weights({k: dev(v) for k, v in synth.make_mlp_weights(seed).items()}, vox.channels,
                                  img.channels, "fp16")
    a, keep = hip._fill_query_args(dev(q), (2, 1, 0), 2.0, vox, packed, "fp16", dev(T), img)
    out = torch.full((B, N), float("nan"), device="cuda:0")
    a.sdf = out.data_ptr()
    a.no_activations = 1
    lib = hip.load()
    plain = hip.sdf_query(dev(q), dev(T), img, vox, packed, precision="fp16", sort_points=False)
    # the default inference forward samples the map inside fc_0 and sorts by Morton cell only (no pixel order is built)
    assert lib.list_sdf_query_fwd(C.byref(a), hip._stream()) == 0, lib.list_last_error()
    torch.cuda.synchronize()
    assert torch.isfinite(out).all() and torch.equal(out, plain)
    # the 2-D gather kernel's forward (no_fused_fc0; also every training forward) builds both orders: inspected below
    out.fill_(float("nan"))
    a.no_fused_fc0 = 1
    assert lib.list_sdf_query_fwd(C.byref(a), hip._stream()) == 0, lib.list_last_error()
    torch.cuda.synchronize()
    assert torch.isfinite(out).all() and torch.equal(out, plain)
    # the orders the call left in its workspace: permutations of the points, inverse of each other where they should be
    ws = keep[-1]
    P = B * N
    rows = (P + 255) // 256 * 256
    kp = 3648
    off = 0
    def take(nbytes):
        nonlocal off
        r = off
        off = (off + nbytes + 255) // 256 * 256
        return r
    for nb in (rows * kp * 2, rows * kp * 2, rows * 512 * 2, rows * 512 * 2, rows * 256 * 2, rows * 256 * 2,
               rows * 256 * 2, rows * 256 * 2):
        take(nb)                                       # x, h1, h2, h3 planes (list_common.h, workspace_layout)
    o_order, o_keys, o_order_img, o_row_of = take(rows * 4), take(rows * 4), take(rows * 4), take(rows * 4)
    as_i32 = lambda o: ws[o:o + P * 4].view(torch.int32).cpu().numpy()
    order, order_img, row_of = as_i32(o_order), as_i32(o_order_img), as_i32(o_row_of)
    assert sorted(order.tolist()) == list(range(P)) and sorted(order_img.tolist()) == list(range(P))
    assert (row_of[order] == np.arange(P)).all()
    # pixel order: image 0's points (all on one pixel) occupy one contiguous run of slots
    img0 = np.nonzero(order_img < N)[0]
    assert img0.max() - img0.min() == N - 1
