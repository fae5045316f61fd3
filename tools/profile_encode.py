"""Where the per-image stage of LIST.forward (everything before the HIP query path) spends its time.
usage: python tools/profile_encode.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from list_amd import arguments, utils          # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(333)
cfg = arguments.default_config(vox_res=128, train_batch_size=8, precision="fp16", img_res=224)
net = utils.get_class("network.models.LIST")(cfg).to(dev).eval()
img = torch.rand((8, 3, 224, 224), device=dev)


def timed(fn, n=5):
    for _ in range(2):
        out = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


with torch.no_grad():
    img_cl, _ = net._apply_memory_format(img)
    t, (feat_g, _) = timed(lambda: net.im_encoder(img))
    print(f"im_encoder (ResNet-18)        {t:8.3f} ms")
    t, pc = timed(lambda: net.point_decoder([feat_g.unsqueeze(1)]))
    print(f"point_decoder (tree GCN)      {t:8.3f} ms   -> {tuple(pc.shape)}")
    t, occ = timed(lambda: net.create_occ(pc))
    print(f"create_occ (on device)        {t:8.3f} ms   -> {tuple(occ.shape)}")
    t, vox = timed(lambda: net.vox_encoder(occ.contiguous(memory_format=torch.channels_last_3d)
                                           if occ.dim() == 5 else occ))
    print(f"vox_encoder (3-D convs, 128^3){t:8.3f} ms")
    t, (feat_g2, feat_l2) = timed(lambda: net.im_encoder2(img_cl))
    print(f"im_encoder2 (ResNet-18)       {t:8.3f} ms")
    t, _ = timed(lambda: net.point_mlp_coarse(pc))
    print(f"point_mlp_coarse              {t:8.3f} ms")
    t, _ = timed(lambda: net.encode(img))
    print(f"encode() total                {t:8.3f} ms")
    # library knobs for the 3-D encoder (out of the hot path's scope; measured for the hand-off row only)
    for name, dt in (("fp16", torch.float16), ("bf16", torch.bfloat16)):
        def run(dt=dt):
            with torch.autocast("cuda", dtype=dt):
                return net.vox_encoder(occ)
        t, v16 = timed(run)
        err = max(float((a.float() - b).abs().max()) for a, b in zip(v16, vox))
        print(f"vox_encoder autocast {name}     {t:8.3f} ms   max|diff| vs fp32 {err:.3e}  formats "
              f"{[('cl' if a.is_contiguous(memory_format=torch.channels_last_3d) else 'nc') for a in v16]}")
    net2 = utils.get_class("network.models.LIST")(cfg).to(dev).eval()          # plain NCDHW encoder
    net2.vox_encoder.load_state_dict(net.vox_encoder.state_dict())
    t, _ = timed(lambda: net2.vox_encoder(occ))
    print(f"vox_encoder NCDHW fp32        {t:8.3f} ms")
    torch.backends.cudnn.benchmark = True
    t, _ = timed(lambda: net2.vox_encoder(occ), n=5)
    print(f"vox_encoder NCDHW fp32 + benchmark(find) {t:8.3f} ms")
    t, _ = timed(lambda: net.vox_encoder(occ), n=5)
    print(f"vox_encoder NDHWC fp32 + benchmark(find) {t:8.3f} ms")
