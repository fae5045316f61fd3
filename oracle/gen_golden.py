"""Generate tests/golden/*.npz by RUNNING THE REFERENCE's own modules (TEST INFRASTRUCTURE).

Run in the authoring container only (needs /root/reference):

    python -m oracle.gen_golden

The reference is imported by path, never copied: /root/reference is put on
sys.path, bytecode writing is disabled, absent third-party modules that the
reference imports at top level (cv2, mcubes, trimesh, torchvision) are replaced
by empty stubs, and torch.Tensor.cuda is an identity while VoxelDecoder2 is
constructed (network/modules.py:214 calls .cuda() unconditionally).
Fixtures hold arrays only (expected outputs + a few intermediates); inputs are
regenerated from oracle/synth.py by whoever consumes the fixture.
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _import_reference():
    import torch
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    for name in ("cv2", "mcubes", "trimesh"):
        sys.modules.setdefault(name, types.ModuleType(name))
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tv.models = tvm
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.models", tvm)
    import network.modules as M          # noqa: E402  (reference)
    import network.losses as L           # noqa: E402  (reference)
    import utils as U                    # noqa: E402  (reference)
    orig = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        dec = M.VoxelDecoder2(3610, 256)
    finally:
        torch.Tensor.cuda = orig
    return torch, M, L, U, dec


def main():
    from . import cases
    torch, M, L, U, dec = _import_reference()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    pool = M.PerceptualPooling()

    for name in cases.CASE_NAMES:
        c = cases.build_case(name)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        dec.load_state_dict({"fc." + k: t(v) for k, v in c["weights"].items()})
        with torch.no_grad():
            query = t(c["query"])
            q = query[:, :, [2, 1, 0]] * 2                       # models.py:91-92 (glue, stated here)
            B, N, _ = q.shape
            img_maps = [t(m) for m in c["img_maps"]]
            vox_maps = [t(m) for m in c["vox_maps"]]
            T = t(c["trans_mat"])
            percep = pool(img_maps, q, T)                         # [B,1024,1,N]
            percep_r = percep.reshape(B, -1, N)
            sdf = dec(q, vox_maps, percep_r)                      # [B,N]
            # intermediates, via the same torch ops the reference calls
            feats = []
            disp = dec.displacments
            pp = torch.cat([q.unsqueeze(1).unsqueeze(1) + d for d in disp], dim=2)
            for f in vox_maps:
                feats.append(torch.nn.functional.grid_sample(f, pp, padding_mode="border",
                                                             align_corners=True))
            vf = torch.cat(feats, dim=1)
            vf = vf.reshape(B, vf.shape[1] * vf.shape[3], vf.shape[4])  # [B,2583,N]
            resized0 = torch.nn.functional.interpolate(img_maps[0], size=137, mode="bilinear",
                                                       align_corners=True)
        sub = slice(0, None, 4)
        np.savez_compressed(
            os.path.join(OUT, f"hotpath_{name}.npz"),
            sdf=sdf.numpy(),
            percep_sub=percep.numpy()[:, :, :, sub],
            voxfeat_sub=vf.numpy()[:, :, sub],
            resized0_sub=resized0.numpy()[:, ::8, ::3, ::3],
            torch_version=np.array(torch.__version__),
        )
        print(name, "sdf", tuple(sdf.shape), float(sdf.abs().max()))

    # a6: grid builder; losses
    grid = U.create_grid_points_from_bounds(-0.5, 0.5, 8)
    rng_o = np.linspace(-1, 1, 2 * 50, dtype=np.float32).reshape(2, 50)
    rng_t = (rng_o[:, ::-1] * 0.7 + 0.1).astype(np.float32).copy()
    loss = L.SDFLoss(2.0)(torch.from_numpy(rng_o), torch.from_numpy(rng_t))
    np.savez_compressed(
        os.path.join(OUT, "aux.npz"),
        grid8=grid, loss_outputs=rng_o, loss_targets=rng_t,
        **{"loss_" + k: v.numpy() for k, v in loss.items()},
        displacements=dec.displacments.numpy(),
    )
    print("aux written")


if __name__ == "__main__":
    main()
