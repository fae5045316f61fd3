"""Child process of tests/test_launch_order_gpu.py: one pass over the launches whose ORDER is kept by something other
than the stream's queue barrier (hipExtAnyOrderLaunch groups, list_common.h), written to an .npz.  The parent runs it
twice -- default dispatch and LIST_LAUNCH_IN_ORDER=1 (the environment is read when the library makes its first query,
hence a fresh process each) -- and compares the files bit for bit."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cases                     # noqa: E402  (inputs of the parity cases: test infrastructure)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def main(out_path):
    import __graft_entry__ as ge
    ge.build()
    from list_amd import hip
    lib = hip.load()
    c = cases.build_case("small")            # B = 3, N = 200: 600 points = three 256-row chunks
    res = {}
    for precision in ("fp16", "bf16x3"):
        md = hip.map_dtype_for(precision)
        img = hip.prep_img_maps([dev(m) for m in c["img_maps"]], dtype=md)
        vox = hip.prep_vox_maps([dev(m) for m in c["vox_maps"]], dtype=md)
        params = {k: dev(v) for k, v in c["weights"].items()}
        packed = hip.prep_mlp_weights(params, vox.channels, img.channels, precision)
        packed_b = hip.prep_mlp_weights_bwd(params, vox.channels, img.channels, precision)
        q, T = dev(c["query"]), dev(c["trans_mat"])
        # (a) a three-chunk call with the point sort ON, twenty times back to back on one workspace: chunk c + 1's
        #     counter clear and sort behind chunk c's MLP, call i + 1's first chunk behind call i's last
        a, keep = hip._fill_query_args(q, (2, 1, 0), 2.0, vox, packed, precision, T, img)
        small = lib.list_query_workspace_bytes(256, a.F, a.H1, a.H2, a.H3)
        ws = torch.empty((small,), dtype=torch.uint8, device="cuda:0")
        a.workspace, a.workspace_bytes = ws.data_ptr(), small
        outs = [torch.full((q.shape[0], q.shape[1]), float("nan"), device="cuda:0") for _ in range(20)]
        for o in outs:
            a.sdf = o.data_ptr()
            assert lib.list_sdf_query_fwd(C.byref(a), hip._stream()) == 0, lib.list_last_error()
        torch.cuda.synchronize()
        for o in outs[1:]:
            assert torch.equal(o, outs[0]), "back-to-back calls on one workspace differ"
        res[f"{precision}_chunked_sorted"] = outs[0].cpu().numpy()
        # (b) unsorted forward (no point orders: the gathers read the queries directly)
        res[f"{precision}_unsorted"] = hip.sdf_query(q, T, img, vox, packed, precision=precision, sort_points=False).cpu().numpy()
        # (c) the repacks themselves (four / three kernels behind one barrier each)
        #     into zeroed buffers through the C ABI: the 256-B alignment gaps between the sections are never written
        w_struct, w_keep = hip._mlp_weights_struct(params, vox.channels, img.channels, precision)
        for name, size_fn, prep_fn in (("packed", lib.list_packed_mlp_bytes, lib.list_prep_mlp_weights),
                                       ("packed_bwd", lib.list_packed_mlp_bwd_bytes, lib.list_prep_mlp_weights_bwd)):
            need = size_fn(C.byref(w_struct))
            buf = torch.zeros((need,), dtype=torch.uint8, device="cuda:0")
            assert prep_fn(C.byref(w_struct), buf.data_ptr(), need, hip._stream()) == 0, lib.list_last_error()
            res[f"{precision}_{name}"] = buf.cpu().numpy()
        # (d) the optimizer's pattern: weights updated IN PLACE on the stream, re-packed and queried at once, five times
        w = {k: v.clone() for k, v in params.items()}
        seq = []
        for it in range(5):
            for k in w:
                w[k].mul_(1.0 + 0.125 * (it + 1))
            pk = hip.prep_mlp_weights(w, vox.channels, img.channels, precision)
            seq.append(hip.sdf_query(q, T, img, vox, pk, precision=precision))
        torch.cuda.synchronize()
        res[f"{precision}_after_inplace_updates"] = torch.stack(seq).cpu().numpy()
        assert all(np.isfinite(v).all() for v in res.values() if v.dtype.kind == "f")
    np.savez(out_path, **res)


if __name__ == "__main__":
    main(sys.argv[1])
