"""Dispatch of the SDF query hot path (reference network/models.py:91-97) to the HIP library.

Forward AND backward always run liblist_hip.so (..hip): there is no CPU or PyTorch fallback, tensors that
are not on a HIP device raise.  Three differentiable forms, each a torch.autograd.Function over the C ABI:
  * fused (LIST.forward, VoxelDecoder2.query): list_sdf_query_fwd / list_sdf_query_bwd -> gradients of the
    MLP parameters, the 5 image maps (through list_img_map_grad_to_levels), the 6 voxel maps, trans_mat;
  * pre-pooled (VoxelDecoder2.forward(p, feat, percep_feat), the reference's own call at models.py:97):
    the same pair; the perceptual gradient comes back as d(percep_feat);
  * PerceptualPooling.forward on its own (models.py:94): list_percep_pool_fwd / list_percep_pool_bwd.
Queries above 262 144 points per call are cut along the point axis (one saved workspace per piece).
Query coordinates are data in the reference's training loop (train.py:82-85) and receive no gradient.
"""
import torch

from .. import hip

N_IMG, N_VOX = hip.N_IMG_LEVELS, hip.N_VOX_LEVELS
MLP_KEYS = ("fc_0.weight", "fc_0.bias", "fc_1.weight", "fc_1.bias", "fc_2.weight", "fc_2.bias",
            "fc_out.weight", "fc_out.bias")
_DISPLACEMENT = 0.0722        # network/modules.py:205


def require_hip(t, what):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise RuntimeError(
            f"{what}: the LIST SDF query path runs only on a HIP device through liblist_hip.so "
            f"(got a {t.device if isinstance(t, torch.Tensor) else type(t).__name__} tensor); "
            "there is no CPU/PyTorch fallback")


def _f32(t):
    return t if t.dtype == torch.float32 else t.float()


class _Cache:
    """Remembers the last prepared object for a tuple of source tensors, e.g. across the chunk loop
    of executors.LIST.test.  A hit needs the SAME tensor objects (held strongly here, so neither
    their ids nor their storage can be recycled by a later, different tensor) with unchanged
    version counters, storage addresses, devices and dtypes; anything else -- new encoder outputs, an
    in-place update, an optimizer step, module.to() / .float() (which swap param.data) -- rebuilds.
    Writes through `.data` bump no version counter: trainable parameters are therefore never served from
    a cache while gradients are enabled (sdf_query re-packs them every step), and VoxelDecoder.invalidate()
    drops everything on train() / load_state_dict() / _apply()."""

    def __init__(self):
        self.sources, self.versions, self.value = None, None, None

    def get(self, tensors, make):
        tensors = tuple(tensors)
        versions = tuple((t._version, t.data_ptr(), t.device, t.dtype) for t in tensors)
        hit = (self.sources is not None and len(self.sources) == len(tensors)
               and all(a is b for a, b in zip(self.sources, tensors)) and versions == self.versions)
        if not hit:
            self.sources, self.versions, self.value = None, None, None     # drop the old maps first
            self.value = make()
            self.sources, self.versions = tensors, versions
        return self.value


def stencil_offsets(device, dtype=torch.float32):
    d = _DISPLACEMENT
    return torch.tensor([[0, 0, 0], [-d, 0, 0], [d, 0, 0], [0, -d, 0], [0, d, 0], [0, 0, -d], [0, 0, d]],
                        dtype=dtype, device=device)


HIP_BACKWARD_MAX_POINTS = 262144


def _point_chunks(B, N):
    if B > HIP_BACKWARD_MAX_POINTS:
        raise RuntimeError(f"batch of {B} images exceeds the {HIP_BACKWARD_MAX_POINTS}-point limit of one backward call")
    per = max(1, HIP_BACKWARD_MAX_POINTS // B)
    return [(n0, min(N, n0 + per)) for n0 in range(0, N, per)]


class _SdfQueryHipFn(torch.autograd.Function):
    """HIP forward + HIP backward.  inputs: (state, trans_mat | percep_feat, 5 image maps (fused form only),
    6 voxel maps, 8 MLP tensors).  state["run"](n0, n1) evaluates the points [n0, n1) of every image and
    returns (sdf, hip.QueryContext)."""

    @staticmethod
    def forward(ctx, state, lead, *tensors):
        B, N = state["shape"]
        ctx.state, ctx.lead_shape, ctx.n_img = state, lead.shape, state["n_img"]
        ctx.pieces = []
        parts = []
        for n0, n1 in _point_chunks(B, N):
            sdf, qctx = state["run"](n0, n1)
            parts.append(sdf)
            ctx.pieces.append((n0, n1, qctx))
        ctx.img_like = [t.detach() for t in tensors[:ctx.n_img]]      # shapes/strides of the encoder maps
        ctx.mlp_shapes = [t.shape for t in tensors[ctx.n_img + N_VOX:]]
        ctx.vox_dtypes = [t.dtype for t in tensors[ctx.n_img:ctx.n_img + N_VOX]]
        return parts[0] if len(parts) == 1 else torch.cat(parts, 1)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        needs = ctx.needs_input_grad
        n_img, percep = ctx.n_img, ctx.state["percep"]
        i_img, i_vox, i_mlp = 2, 2 + n_img, 2 + n_img + N_VOX
        want_lead = bool(needs[1])                        # trans_mat (fused) or percep_feat (pre-pooled)
        want_img = any(needs[i_img:i_vox])
        want_vox = any(needs[i_vox:i_mlp])
        want_mlp = any(needs[i_mlp:])
        grad_out = _f32(grad_out)
        total, lead_parts = None, []
        # one piece: the adjoint resize runs inside the call; a cut query sums the map gradient first (it is linear)
        like = ctx.img_like if (len(ctx.pieces) == 1 and want_img and not percep) else None
        for n0, n1, qctx in ctx.pieces:
            out = hip.sdf_query_backward(qctx, grad_out[:, n0:n1], ctx.state["packed_bwd"](), want_mlp=want_mlp,
                                         want_img=(want_lead if percep else want_img), want_vox=want_vox,
                                         want_trans=(want_lead and not percep), img_levels_like=like,
                                         want_img_map=like is None)       # one piece: only the levels are wanted
            if percep and want_lead:
                lead_parts.append(out.pop("percep_feat"))
            if total is None:
                total = out
            else:                                          # further pieces of a long query: sum
                for k, v in out.items():
                    if k == "mlp":
                        for kk in v:
                            total[k][kk] += v[kk]
                    elif k == "vox":
                        for a, b in zip(total[k], v):
                            a += b
                    else:
                        total[k] += v
        grads = [None]
        if percep:
            grads.append((lead_parts[0] if len(lead_parts) == 1 else torch.cat(lead_parts, 2)).reshape(ctx.lead_shape)
                         if want_lead else None)
        else:
            grads.append(total["trans_mat"].reshape(ctx.lead_shape) if want_lead else None)
        if want_img and not percep:
            levels = total["img_levels"] if like is not None else hip.img_map_grad_to_levels(total["img_map"], ctx.img_like)
            grads += [g if n else None for g, n in zip(levels, needs[i_img:i_vox])]
        else:
            grads += [None] * n_img
        if want_vox:       # channels-last buffers seen as [B,C,D,H,W]
            grads += [g.permute(0, 4, 1, 2, 3).to(dt) if n else None
                      for g, n, dt in zip(total["vox"], needs[i_vox:i_mlp], ctx.vox_dtypes)]
        else:
            grads += [None] * N_VOX
        if want_mlp:
            grads += [total["mlp"][k].reshape(shp) if n else None
                      for k, shp, n in zip(MLP_KEYS, ctx.mlp_shapes, needs[i_mlp:])]
        else:
            grads += [None] * len(MLP_KEYS)
        ctx.pieces = None                  # releases the saved workspaces (X, H1, H2)
        return tuple(grads)


class _PercepPoolFn(torch.autograd.Function):
    """PerceptualPooling.forward on its own: list_percep_pool_fwd / list_percep_pool_bwd.
    inputs: (img (prepared), map_size, pc, trans_mat, 5 image maps)."""

    @staticmethod
    def forward(ctx, img, pc, trans_mat, *img_maps):
        ctx.img, ctx.pc, ctx.tm = img, pc.detach(), trans_mat.detach()
        ctx.img_like = [t.detach() for t in img_maps]
        ctx.tm_shape = trans_mat.shape
        return hip.percep_pool(ctx.pc, ctx.tm, img)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        needs = ctx.needs_input_grad
        want_trans, want_img = bool(needs[2]), any(needs[3:])
        out = hip.percep_pool_backward(ctx.pc, ctx.tm, ctx.img, _f32(grad_out).contiguous(), want_img=want_img,
                                       want_trans=want_trans)
        grads = [None, None, out["trans_mat"].reshape(ctx.tm_shape) if want_trans else None]
        if want_img:
            levels = hip.img_map_grad_to_levels(out["img_map"], ctx.img_like)
            grads += [g if n else None for g, n in zip(levels, needs[3:])]
        else:
            grads += [None] * len(ctx.img_like)
        return tuple(grads)


def percep_pool(img_maps, pc, trans_mat, img):
    """[B,Ct,1,N] with gradients to the image maps and trans_mat when they require them."""
    diff = [t for t in (trans_mat, *img_maps) if t.requires_grad]
    if torch.is_grad_enabled() and diff:
        return _PercepPoolFn.apply(img, _f32(pc), _f32(trans_mat), *[_f32(m) for m in img_maps])
    return hip.percep_pool(_f32(pc), _f32(trans_mat), img)


def sdf_query(query, trans_mat, img_maps, vox_maps, mlp_params, *, perm=(2, 1, 0), scale=2.0,
              map_size=137, precision="bf16x3", percep_feat=None, caches=None, ordered_points=False,
              project_percep=None):
    """sdf [B,N] for raw queries; mlp_params: dict with the reference's fc_* keys.
    ordered_points: the queries already come in a spatially coherent order (a raster grid): the forward skips its
    Morton / pixel counting sort (same values; on a 256^3 inference grid the sort is 6 % of the query time and the
    gathers run as fast without it).  Inference only.
    project_percep: take the projected perceptual map (hip.prep_percep_proj) -- None: when ordered_points and the call
    carries at least 4 map_size^2 points per image; a driver that cuts one grid into several calls passes True / False for
    all of them so that every call (and every rank of a sharded grid) computes the same bits."""
    require_hip(query, "query")
    # nn.DataParallel replicas share the module's cache dict (replicate() copies attributes by reference) and run
    # in one thread per device: every device gets its own slots
    caches = (caches if caches is not None else {}).setdefault(("device", query.device.index), {})
    query = _f32(query)
    # a half-precision producer's vector levels stay fp16 (used where they lie); its scalar level is read as fp32
    vox_maps = [v if (v.dtype == torch.float16 and v.shape[1] > 1) else _f32(v) for v in vox_maps]
    mlp = [mlp_params[k] for k in MLP_KEYS]
    md = hip.map_dtype_for(precision)
    img_maps = [_f32(m) for m in img_maps] if percep_feat is None else []
    img_C = sum(m.shape[1] for m in img_maps) if percep_feat is None else percep_feat.shape[1]
    vox_C = [int(v.shape[1]) for v in vox_maps]
    mlp_dict = {k: t.detach() for k, t in zip(MLP_KEYS, mlp)}
    training = torch.is_grad_enabled() and any(t.requires_grad for t in mlp)
    if training:          # parameters move every step (and `.data` writes are invisible to a cache): pack afresh
        packed = hip.prep_mlp_weights(mlp_dict, vox_C, img_C, precision)
    else:
        packed = caches.setdefault("mlp:" + str(precision), _Cache()).get(
            mlp, lambda: hip.prep_mlp_weights(mlp_dict, vox_C, img_C, precision))
    diff = [t for t in (trans_mat, percep_feat, *img_maps, *vox_maps, *mlp)
            if t is not None and t.requires_grad]
    inference = not (torch.is_grad_enabled() and diff)
    # many points per image on fixed maps and weights (an inference grid): the perceptual block of fc_0 is applied
    # to the 137^2 map once (hip.prep_percep_proj, cached with the map and the weights) instead of per point
    want = project_percep if project_percep is not None else (
        ordered_points and percep_feat is None and query.shape[1] >= 4 * map_size * map_size)
    img = None
    if percep_feat is None and inference and not want and hip.img_proj_default(precision) \
            and hip.img_proj_kept_levels(img_maps, map_size) < N_IMG:
        # inference forwards: the low-resolution encoder levels (896 of the 1024 perceptual channels) go through their
        # columns of fc_0 BEFORE the resize, 14^2 .. 56^2 pixels per image instead of one product per query point
        # (hip.prep_img_proj, list_prep_img_proj: F.interpolate and fc_0 are linear and commute)
        def make_proj():
            try:
                return hip.prep_img_proj([m.detach() for m in img_maps], packed, map_size, precision)
            except hip.ListError as e:             # channel counts the projection does not take
                if e.code == hip.ERR_UNSUPPORTED:
                    return False
                raise
        img = caches.setdefault("imgproj:" + str(precision), _Cache()).get(img_maps + [packed.data], make_proj) or None
    if percep_feat is None and img is None:
        img = caches.setdefault("img:" + md, _Cache()).get(
            img_maps, lambda: hip.prep_img_maps([m.detach() for m in img_maps], map_size, md))
    # (the 2-D side first: its kernels are the call's first launches -- channels-last encoders make the 3-D hand-off below
    # a zero-copy with no launch at all, and the host work in between would otherwise sit in front of an idle device)
    vox = caches.setdefault("vox:" + md, _Cache()).get(
        vox_maps, lambda: hip.prep_vox_maps([v.detach() for v in vox_maps], md))

    def run():
        proj = None
        if want and img is not None and not training and not torch.is_grad_enabled():
            def make():
                try:
                    return hip.prep_percep_proj(img, packed, precision)
                except hip.ListError as e:         # shapes / dtype pairs the projection does not take
                    if e.code == hip.ERR_UNSUPPORTED:
                        return False
                    raise
            proj = caches.setdefault("proj:" + str(precision), _Cache()).get([img.data, packed.data], make) or None
        return hip.sdf_query(query.detach(), trans_mat.detach() if trans_mat is not None else None,
                             img, vox, packed, perm=perm, scale=scale, precision=precision,
                             percep_feat=percep_feat.detach() if percep_feat is not None else None,
                             sort_points=not ordered_points, percep_proj=proj)

    if torch.is_grad_enabled() and diff and query.shape[0] * query.shape[1] > 0:
        q_det = query.detach()
        tm_det = trans_mat.detach() if trans_mat is not None else None
        pf_det = percep_feat.detach() if percep_feat is not None else None

        def run_saving(n0, n1):
            return hip.sdf_query(q_det[:, n0:n1], tm_det, img, vox, packed, perm=perm, scale=scale,
                                 precision=precision, save_for_backward=True,
                                 percep_feat=pf_det[:, :, n0:n1] if pf_det is not None else None)

        # the transposed copies for the data gradients are taken NOW, from the same parameter values as `packed`
        # (a lazy build at backward time could see parameters an optimizer or EMA step has moved in between)
        if training:
            packed_b = hip.prep_mlp_weights_bwd(mlp_dict, vox_C, img_C, precision)
        else:
            packed_b = caches.setdefault("mlpT:" + str(precision), _Cache()).get(
                mlp, lambda: hip.prep_mlp_weights_bwd(mlp_dict, vox_C, img_C, precision))

        def packed_bwd():
            return packed_b
        state = {"run": run_saving, "packed_bwd": packed_bwd, "percep": percep_feat is not None,
                 "n_img": len(img_maps), "shape": (query.shape[0], query.shape[1])}
        lead = percep_feat if percep_feat is not None else trans_mat
        return _SdfQueryHipFn.apply(state, lead, *img_maps, *vox_maps, *mlp)
    return run()
