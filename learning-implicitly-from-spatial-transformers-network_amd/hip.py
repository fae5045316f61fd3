"""Host-side binding of the C ABI (include/list_hip.h) for PyTorch-ROCm tensors.

PyTorch is plumbing here: device memory (caching allocator), the current HIP stream and
`torch.distributed`.  Every compute call goes through liblist_hip.so; there is NO fallback:
if the library is missing or a call fails, a RuntimeError is raised.
"""
import ctypes as C
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# LIST_HIP_LIB (diagnostic): another build of the same library, e.g. a compile-time variant for an A/B run
LIB_PATH = os.environ.get("LIST_HIP_LIB") or os.path.join(_HERE, "csrc", "liblist_hip.so")

N_IMG_LEVELS = 5
N_VOX_LEVELS = 6
PREC_BF16X3 = 0
PREC_BF16 = 1
PREC_FP16 = 2
PRECISIONS = {"bf16x3": PREC_BF16X3, "bf16": PREC_BF16, "fp16": PREC_FP16}
MAP_F32, MAP_F16 = 0, 1
MAP_DTYPES = {"f32": MAP_F32, "f16": MAP_F16}


def map_dtype_for(precision):
    """fp16 maps pair with the fp16 MLP (features are rounded to fp16 after interpolation anyway);
    the fp32-grade and bf16 paths keep fp32 maps."""
    return "f16" if precision in ("fp16", PREC_FP16) else "f32"


class ListMap2D(C.Structure):
    _fields_ = [("data", C.c_void_p), ("C", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("sb", C.c_int64), ("sc", C.c_int64), ("sh", C.c_int64), ("sw", C.c_int64)]


class ListMap3D(C.Structure):
    _fields_ = [("data", C.c_void_p), ("C", C.c_int32), ("D", C.c_int32), ("H", C.c_int32),
                ("W", C.c_int32), ("dtype", C.c_int32), ("reserved_", C.c_int32),
                ("sb", C.c_int64), ("sc", C.c_int64), ("sd", C.c_int64),
                ("sh", C.c_int64), ("sw", C.c_int64)]


class ListVoxLevel(C.Structure):
    _fields_ = [("data", C.c_void_p), ("C", C.c_int32), ("D", C.c_int32), ("H", C.c_int32),
                ("W", C.c_int32), ("dtype", C.c_int32), ("reserved_", C.c_int32),
                ("image_stride", C.c_int64)]


class ListMlpWeights(C.Structure):
    _fields_ = [("w0", C.c_void_p), ("b0", C.c_void_p), ("w1", C.c_void_p), ("b1", C.c_void_p),
                ("w2", C.c_void_p), ("b2", C.c_void_p), ("w3", C.c_void_p), ("b3", C.c_void_p),
                ("F", C.c_int32), ("H1", C.c_int32), ("H2", C.c_int32), ("H3", C.c_int32),
                ("vox_C", C.c_int32 * N_VOX_LEVELS), ("img_C", C.c_int32), ("precision", C.c_int32)]


class ListQueryArgs(C.Structure):
    _fields_ = [("B", C.c_int32), ("N", C.c_int32),
                ("query", C.c_void_p), ("q_sb", C.c_int64), ("q_sn", C.c_int64), ("q_sc", C.c_int64),
                ("perm", C.c_int32 * 3), ("scale", C.c_float),
                ("trans_mat", C.c_void_p),
                ("img_map", C.c_void_p), ("img_dtype", C.c_int32), ("map_size", C.c_int32),
                ("img_C", C.c_int32), ("clamp_hi", C.c_float),
                ("percep_feat", C.c_void_p), ("pf_sb", C.c_int64), ("pf_sc", C.c_int64),
                ("pf_sn", C.c_int64),
                ("vox", ListVoxLevel * N_VOX_LEVELS),
                ("packed_mlp", C.c_void_p),
                ("F", C.c_int32), ("H1", C.c_int32), ("H2", C.c_int32), ("H3", C.c_int32),
                ("sdf", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
                ("precision", C.c_int32),
                ("stage_events", C.POINTER(C.c_void_p)), ("no_sort", C.c_int32),
                ("stage_event_sets", C.c_int32), ("percep_proj", C.c_void_p), ("no_activations", C.c_int32),
                ("no_fused_fc0", C.c_int32), ("img_proj", C.c_int32), ("img_kept_C", C.c_int32)]


class ListMlpGrads(C.Structure):
    _fields_ = [("w0", C.c_void_p), ("b0", C.c_void_p), ("w1", C.c_void_p), ("b1", C.c_void_p),
                ("w2", C.c_void_p), ("b2", C.c_void_p), ("w3", C.c_void_p), ("b3", C.c_void_p)]


class ListQueryGradArgs(C.Structure):
    _fields_ = [("fwd", C.POINTER(ListQueryArgs)), ("grad_sdf", C.c_void_p),
                ("packed_mlp_bwd", C.c_void_p), ("mlp", ListMlpGrads),
                ("grad_img_map", C.c_void_p), ("grad_trans_mat", C.c_void_p),
                ("grad_vox", ListVoxLevel * N_VOX_LEVELS),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
                ("stage_events", C.POINTER(C.c_void_p)), ("vox_adjoint", C.c_int32),
                ("grad_percep_feat", C.c_void_p), ("gpf_sb", C.c_int64), ("gpf_sc", C.c_int64),
                ("gpf_sn", C.c_int64), ("aux_streams", C.c_void_p * 3),
                ("grad_img_levels", C.POINTER(ListMap2D)), ("grad_img_map_dtype", C.c_int32)]


VOX_ADJOINT = {"auto": 0, "scatter": 1, "gather": 2}
N_BWD_STAGES = 11
BWD_STAGE_NAMES = ("head", "wgrad_fc2", "dgrad_fc2", "wgrad_fc1", "dgrad_fc1", "dgrad_fc0", "wgrad_fc0",
                   "scatter_vox", "img_map_grad", "trans_mat_grad")

N_STAGES = 13
STAGE_VOX0, STAGE_IMG = 2, 7     # events VOX0 .. IMG lie between two gathers (include/list_hip.h, ListStage)
# interval i = [event i, event i+1]: the kernel (group) that ends at stage i+1 of include/list_hip.h
STAGE_NAMES = ("sort_points", "gather_vox_l1", "gather_vox_l2", "gather_vox_l3", "gather_vox_l4",
               "gather_vox_l5", "gather_img", "gather_tail", "fc_0", "exact_redo", "fc_1", "fc_2_out")


class ListPoolArgs(C.Structure):
    _fields_ = [("B", C.c_int32), ("N", C.c_int32),
                ("pc", C.c_void_p), ("p_sb", C.c_int64), ("p_sn", C.c_int64), ("p_sc", C.c_int64),
                ("trans_mat", C.c_void_p),
                ("img_map", C.c_void_p), ("img_dtype", C.c_int32), ("map_size", C.c_int32),
                ("img_C", C.c_int32), ("clamp_hi", C.c_float),
                ("out", C.c_void_p)]


class ListPoolGradArgs(C.Structure):
    _fields_ = [("fwd", C.POINTER(ListPoolArgs)), ("grad_out", C.c_void_p), ("g_sb", C.c_int64),
                ("g_sc", C.c_int64), ("g_sn", C.c_int64), ("grad_img_map", C.c_void_p),
                ("grad_trans_mat", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t)]


class ListQueryPlan(C.Structure):
    _fields_ = [("rows_per_chunk", C.c_int64), ("chunks", C.c_int32), ("fused_tail", C.c_int32),
                ("fc0_k", C.c_int32), ("box_levels", C.c_int32), ("fused_fc0", C.c_int32),
                ("img_proj", C.c_int32)]


EXPORTS = {
    "list_img_map_bytes": (C.c_size_t, [C.POINTER(ListMap2D), C.c_int32, C.c_int32, C.c_int32]),
    "list_prep_img_maps": (C.c_int, [C.POINTER(ListMap2D), C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                     C.c_size_t, C.c_void_p]),
    "list_vox_pack_bytes": (C.c_size_t, [C.POINTER(ListMap3D), C.c_int32, C.c_int32]),
    "list_prep_vox_maps": (C.c_int, [C.POINTER(ListMap3D), C.c_int32, C.c_int32, C.c_void_p, C.c_size_t,
                                     C.POINTER(ListVoxLevel), C.c_void_p]),
    "list_percep_proj_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "list_percep_proj_scratch_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "list_prep_percep_proj": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_int32,
                                        C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                        C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    "list_img_proj_map_bytes": (C.c_size_t, [C.POINTER(ListMap2D), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "list_img_proj_scratch_bytes": (C.c_size_t, [C.POINTER(ListMap2D), C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "list_prep_img_proj": (C.c_int, [C.POINTER(ListMap2D), C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32),
                                     C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t,
                                     C.c_void_p, C.c_size_t, C.c_void_p]),
    "list_packed_mlp_bytes": (C.c_size_t, [C.POINTER(ListMlpWeights)]),
    "list_prep_mlp_weights": (C.c_int, [C.POINTER(ListMlpWeights), C.c_void_p, C.c_size_t,
                                        C.c_void_p]),
    "list_query_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                                C.c_int32]),
    "list_query_chunk_rows": (C.c_int64, [C.c_size_t, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "list_sdf_query_fwd": (C.c_int, [C.POINTER(ListQueryArgs), C.c_void_p]),
    "list_query_plan": (C.c_int, [C.POINTER(ListQueryArgs), C.POINTER(ListQueryPlan)]),
    "list_percep_pool_fwd": (C.c_int, [C.POINTER(ListPoolArgs), C.c_void_p]),
    "list_gather_features_fwd": (C.c_int, [C.POINTER(ListQueryArgs), C.c_void_p, C.c_void_p]),
    "list_gemm_nt": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                               C.c_void_p]),
    "list_split_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "list_to_fp16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "list_packed_mlp_bwd_bytes": (C.c_size_t, [C.POINTER(ListMlpWeights)]),
    "list_prep_mlp_weights_bwd": (C.c_int, [C.POINTER(ListMlpWeights), C.c_void_p, C.c_size_t, C.c_void_p]),
    "list_query_bwd_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                                    C.c_int32]),
    "list_sdf_query_bwd": (C.c_int, [C.POINTER(ListQueryGradArgs), C.c_void_p]),
    "list_img_map_grad_to_levels": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(ListMap2D),
                                              C.c_void_p]),
    "list_percep_pool_bwd_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32]),
    "list_percep_pool_bwd": (C.c_int, [C.POINTER(ListPoolGradArgs), C.c_void_p]),
    "list_gemm_tn": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_size_t, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "list_last_error": (C.c_char_p, []),
    "list_abi_version": (C.c_int, []),
}

_lib = None
_lock = threading.Lock()


ABI_VERSION = 9          # LIST_ABI_VERSION of the include/list_hip.h these ctypes structs mirror (checked in load())


def load():
    """dlopen liblist_hip.so and bind every symbol of include/list_hip.h.  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` "
                    "(hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback for the "
                    "LIST SDF query path.")
            lib = C.CDLL(LIB_PATH)
            for name, (res, args) in EXPORTS.items():
                fn = getattr(lib, name)
                fn.restype, fn.argtypes = res, args
            # the argument structs grow at their end from one ABI version to the next: a library that reads a longer
            # struct than this binding fills would take flags (no_activations, grad_img_map_dtype) from stray bytes
            got = lib.list_abi_version()
            if got != ABI_VERSION:
                raise RuntimeError(f"{LIB_PATH} speaks ABI {got}, this binding ABI {ABI_VERSION} (include/list_hip.h: "
                                   "LIST_ABI_VERSION): rebuild it with `python __graft_entry__.py`")
            _lib = lib
    return _lib


class ListError(RuntimeError):
    """A call into liblist_hip.so returned a non-zero ListStatus (include/list_hip.h); `code` is that status."""

    def __init__(self, what, code, msg):
        super().__init__(f"{what} failed ({code}): {msg}")
        self.what, self.code = what, int(code)


ERR_ARG, ERR_SHAPE, ERR_WORKSPACE, ERR_HIP, ERR_UNSUPPORTED = -1, -2, -3, -4, -5      # enum ListStatus


def _check(rc, what):
    if rc != 0:
        raise ListError(what, rc, load().list_last_error().decode("utf-8", "replace"))


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32_cuda(t, name):
    if not isinstance(t, torch.Tensor) or t.dtype != torch.float32 or not t.is_cuda:
        raise RuntimeError(f"{name} must be a float32 CUDA/HIP tensor (got {type(t).__name__} "
                           f"{getattr(t, 'dtype', None)} {getattr(t, 'device', None)})")
    return t


# ------------------------------------------------------------------------------------------------
class PreparedImage:
    """Channels-last resized perceptual map [B,ms,ms,Ct], float32 or float16.

    kept_C is not None: the output of prep_img_proj -- [B,ms,ms,kept_C + H1]: the first kept_C channels are the resized
    high-resolution encoder levels, the H1 behind them the low-resolution levels projected through fc_0 (valid for
    `packed` only); `channels` stays the channel count of the feature layout (1024)."""

    def __init__(self, data, map_size, channels, dtype, kept_C=None, packed=None):
        self.data, self.map_size, self.channels, self.dtype = data, map_size, channels, dtype
        self.kept_C, self.packed = kept_C, packed


class PreparedVoxels:
    """Per-image [D][H][W][C] voxel levels; keeps the owning tensors alive."""

    def __init__(self, levels, keepalive):
        self.levels, self._keep = levels, keepalive

    @property
    def channels(self):
        return [int(self.levels[i].C) for i in range(N_VOX_LEVELS)]


class PackedMlp:
    def __init__(self, data, F, H1, H2, H3, vox_C, img_C, fp16):
        self.data, self.F, self.H1, self.H2, self.H3 = data, F, H1, H2, H3
        self.vox_C, self.img_C, self.fp16 = list(vox_C), img_C, fp16


def prep_img_maps(img_featuremaps, map_size=137, dtype="f32"):
    """F.interpolate x5 (+ layout) of the reference, network/modules.py:26-35."""
    lib = load()
    md = MAP_DTYPES[dtype]
    if len(img_featuremaps) != N_IMG_LEVELS:
        raise RuntimeError(f"expected {N_IMG_LEVELS} image feature maps, got {len(img_featuremaps)}")
    maps = (ListMap2D * N_IMG_LEVELS)()
    B = img_featuremaps[0].shape[0]
    Ct = 0
    for i, t in enumerate(img_featuremaps):
        _f32_cuda(t, f"img_featuremaps[{i}]")
        if t.dim() != 4 or t.shape[0] != B:
            raise RuntimeError(f"img_featuremaps[{i}] must be [B,C,H,W]")
        maps[i] = ListMap2D(t.data_ptr(), t.shape[1], t.shape[2], t.shape[3], *t.stride())
        Ct += t.shape[1]
    out = torch.empty((B, map_size, map_size, Ct), dtype=torch.float16 if md == MAP_F16 else torch.float32,
                      device=img_featuremaps[0].device)
    with torch.cuda.device(out.device):
        _check(lib.list_prep_img_maps(maps, B, map_size, md, out.data_ptr(),
                                      out.numel() * out.element_size(), _stream()), "list_prep_img_maps")
    return PreparedImage(out, map_size, Ct, md)


def img_proj_default(precision):
    """Whether inference forwards through the module API (and bench.py's modes) take prep_img_proj.  Measured at BASELINE
    config 2 (B = 8, N = 20 000, 224^2; DESIGN 4 "Round 4b"): every precision gains -- bf16x3 3.50 -> 3.13 ms, bf16
    2.97 -> 2.77 ms, fp16 1.99 -> 1.92 ms (there the kept levels are sampled inside fc_0 and the projected channels in
    its epilogue, k_fc0_fused<0, true>).  LIST_IMG_PROJ=0 / 1 forces it off / on."""
    e = os.environ.get("LIST_IMG_PROJ", "")
    if e in ("0", "1"):
        return e == "1"
    return True


def img_proj_kept_levels(img_featuremaps, map_size=137):
    """How many leading encoder levels prep_img_proj resizes (the rest is projected at its own resolution): a level is
    projected when the resize enlarges it at least 2 x 2 (its H * W * 4 <= map_size^2) and so is every level behind it."""
    n = len(img_featuremaps)
    while n > 0 and img_featuremaps[n - 1].shape[2] * img_featuremaps[n - 1].shape[3] * 4 <= map_size * map_size:
        n -= 1
    return n


def prep_img_proj(img_featuremaps, packed, map_size=137, precision="bf16x3", n_kept_levels=None):
    """list_prep_img_proj: the perceptual map of an INFERENCE forward with the low-resolution encoder levels projected
    through their columns of fc_0 before the resize (F.interpolate and fc_0 are both linear and commute:
    network/modules.py:26-35, 276).  -> PreparedImage for sdf_query(..., save_for_backward=False) with `packed`."""
    lib = load()
    prec = PRECISIONS[precision] if isinstance(precision, str) else int(precision)
    if len(img_featuremaps) != N_IMG_LEVELS:
        raise RuntimeError(f"expected {N_IMG_LEVELS} image feature maps, got {len(img_featuremaps)}")
    if n_kept_levels is None:
        n_kept_levels = img_proj_kept_levels(img_featuremaps, map_size)
    if not 0 <= n_kept_levels < N_IMG_LEVELS:
        raise ListError("prep_img_proj", ERR_UNSUPPORTED, "no encoder level is small enough to project")
    maps = (ListMap2D * N_IMG_LEVELS)()
    B = img_featuremaps[0].shape[0]
    Ct = kept = 0
    for i, t in enumerate(img_featuremaps):
        _f32_cuda(t, f"img_featuremaps[{i}]")
        if t.dim() != 4 or t.shape[0] != B:
            raise RuntimeError(f"img_featuremaps[{i}] must be [B,C,H,W]")
        maps[i] = ListMap2D(t.data_ptr(), t.shape[1], t.shape[2], t.shape[3], *t.stride())
        Ct += t.shape[1]
        kept += t.shape[1] if i < n_kept_levels else 0
    if Ct != packed.img_C:
        raise RuntimeError(f"the encoder levels hold {Ct} channels, the packed weights expect {packed.img_C}")
    nbytes = lib.list_img_proj_map_bytes(maps, B, map_size, n_kept_levels, packed.H1, prec)
    sbytes = lib.list_img_proj_scratch_bytes(maps, B, n_kept_levels, packed.H1, prec)
    if nbytes == 0 or sbytes == 0:
        raise ListError("list_img_proj_map_bytes", ERR_UNSUPPORTED, lib.list_last_error().decode("utf-8", "replace"))
    dev = img_featuremaps[0].device
    f16 = prec == PREC_FP16
    out = torch.empty((B, map_size, map_size, kept + packed.H1), dtype=torch.float16 if f16 else torch.float32, device=dev)
    scratch = torch.empty((sbytes,), dtype=torch.uint8, device=dev)
    vc = (C.c_int32 * N_VOX_LEVELS)(*[int(c) for c in packed.vox_C])
    with torch.cuda.device(dev):
        _check(lib.list_prep_img_proj(maps, B, map_size, n_kept_levels, vc, packed.data.data_ptr(), packed.H1, packed.H2,
                                      packed.H3, prec, out.data_ptr(), out.numel() * out.element_size(),
                                      scratch.data_ptr(), sbytes, _stream()), "list_prep_img_proj")
    return PreparedImage(out, map_size, Ct, MAP_F16 if f16 else MAP_F32, kept_C=kept, packed=packed)


def prep_vox_maps(vox_feat, dtype="f32"):
    """Layout hand-off for the 3-D grid_sample of network/modules.py:263-265.  Levels may be float32 or, from a
    half-precision producer, float16 (channels-last fp16 levels are used where they lie when dtype is 'f16')."""
    lib = load()
    md = MAP_DTYPES[dtype]
    if len(vox_feat) != N_VOX_LEVELS:
        raise RuntimeError(f"expected {N_VOX_LEVELS} voxel feature maps, got {len(vox_feat)}")
    maps = (ListMap3D * N_VOX_LEVELS)()
    B = vox_feat[0].shape[0]
    for i, t in enumerate(vox_feat):
        if not (torch.is_tensor(t) and t.is_cuda and t.dtype in (torch.float32, torch.float16)):
            raise RuntimeError(f"vox_feat[{i}] must be a float32 or float16 CUDA tensor")
        if t.dim() != 5 or t.shape[0] != B:
            raise RuntimeError(f"vox_feat[{i}] must be [B,C,D,H,W]")
        maps[i] = ListMap3D(t.data_ptr(), t.shape[1], t.shape[2], t.shape[3], t.shape[4],
                            MAP_F16 if t.dtype == torch.float16 else MAP_F32, 0, *t.stride())
    need = lib.list_vox_pack_bytes(maps, B, md)
    dev = vox_feat[0].device
    pack = torch.empty((max(need, 16) // 4,), dtype=torch.float32, device=dev)
    levels = (ListVoxLevel * N_VOX_LEVELS)()
    with torch.cuda.device(dev):
        _check(lib.list_prep_vox_maps(maps, B, md, pack.data_ptr(), pack.numel() * 4, levels, _stream()),
               "list_prep_vox_maps")
    return PreparedVoxels(levels, (pack, list(vox_feat)))


def _mlp_weights_struct(params, vox_C, img_C, precision):
    ts = {}
    for name in ("fc_0", "fc_1", "fc_2", "fc_out"):
        w = _f32_cuda(params[name + ".weight"], name + ".weight").detach()
        b = _f32_cuda(params[name + ".bias"], name + ".bias").detach()
        ts[name] = (w.reshape(w.shape[0], -1).contiguous(), b.contiguous())
    w = ListMlpWeights()
    w.w0, w.b0 = ts["fc_0"][0].data_ptr(), ts["fc_0"][1].data_ptr()
    w.w1, w.b1 = ts["fc_1"][0].data_ptr(), ts["fc_1"][1].data_ptr()
    w.w2, w.b2 = ts["fc_2"][0].data_ptr(), ts["fc_2"][1].data_ptr()
    w.w3, w.b3 = ts["fc_out"][0].data_ptr(), ts["fc_out"][1].data_ptr()
    w.F, w.H1 = ts["fc_0"][0].shape[1], ts["fc_0"][0].shape[0]
    w.H2, w.H3 = ts["fc_1"][0].shape[0], ts["fc_2"][0].shape[0]
    if ts["fc_1"][0].shape[1] != w.H1 or ts["fc_2"][0].shape[1] != w.H2 or \
            tuple(ts["fc_out"][0].shape) != (1, w.H3):
        raise RuntimeError("MLP weight shapes do not chain")
    for i, c in enumerate(vox_C):
        w.vox_C[i] = int(c)
    w.img_C = int(img_C)
    w.precision = PRECISIONS[precision] if isinstance(precision, str) else int(precision)
    return w, ts


def prep_mlp_weights(params, vox_C, img_C=1024, precision="bf16x3"):
    """params: dict with fc_0/fc_1/fc_2/fc_out .weight/.bias (reference state_dict names,
    network/modules.py:196-200).  Conv1d weights may be [out,in,1] or [out,in]."""
    lib = load()
    w, ts = _mlp_weights_struct(params, vox_C, img_C, precision)
    need = lib.list_packed_mlp_bytes(C.byref(w))
    if need == 0:
        raise RuntimeError("list_packed_mlp_bytes failed: "
                           + lib.list_last_error().decode("utf-8", "replace"))
    dev = ts["fc_0"][0].device
    packed = torch.empty((need,), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _check(lib.list_prep_mlp_weights(C.byref(w), packed.data_ptr(), need, _stream()),
               "list_prep_mlp_weights")
    # temporaries made by .contiguous() are released stream-ordered by the caching allocator, and
    # every kernel above was enqueued on the same (current) stream: no synchronisation needed
    return PackedMlp(packed, w.F, w.H1, w.H2, w.H3, vox_C, img_C, w.precision == PREC_FP16)


def prep_mlp_weights_bwd(params, vox_C, img_C=1024, precision="bf16x3"):
    """Transposed 16-bit copies of fc_0..fc_2 for the data-gradient GEMMs of list_sdf_query_bwd."""
    lib = load()
    w, ts = _mlp_weights_struct(params, vox_C, img_C, precision)
    need = lib.list_packed_mlp_bwd_bytes(C.byref(w))
    if need == 0:
        raise RuntimeError("list_packed_mlp_bwd_bytes failed: "
                           + lib.list_last_error().decode("utf-8", "replace"))
    dev = ts["fc_0"][0].device
    packed = torch.empty((need,), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _check(lib.list_prep_mlp_weights_bwd(C.byref(w), packed.data_ptr(), need, _stream()),
               "list_prep_mlp_weights_bwd")
    return packed


class PercepProj:
    """The perceptual map projected through the perceptual block of fc_0 (list_prep_percep_proj): valid for the
    prepared map and the packed weights it was made from."""

    def __init__(self, data, img, packed):
        self.data, self.img, self.packed = data, img, packed


def prep_percep_proj(img, packed, precision="bf16x3"):
    """Inference with many points per image (a marching-cubes grid): fc_0 is linear, so its perceptual block applied
    to a bilinear sample of the prepared map equals the bilinear sample of the projected map.  -> PercepProj for
    sdf_query(..., percep_proj=...).  Raises RuntimeError(unsupported) for shapes / dtype pairs the path does not take."""
    lib = load()
    prec = PRECISIONS[precision] if isinstance(precision, str) else int(precision)
    B = img.data.shape[0]
    nbytes = lib.list_percep_proj_bytes(B, img.map_size, packed.H1, prec)
    sbytes = lib.list_percep_proj_scratch_bytes(B, img.map_size, img.channels, prec)
    dev = img.data.device
    out = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    scratch = torch.empty((max(sbytes, 16),), dtype=torch.uint8, device=dev)
    vc = (C.c_int32 * N_VOX_LEVELS)(*[int(c) for c in packed.vox_C])
    with torch.cuda.device(dev):
        _check(lib.list_prep_percep_proj(img.data.data_ptr(), img.dtype, B, img.map_size, vc, img.channels,
                                         packed.data.data_ptr(), packed.H1, packed.H2, packed.H3, prec,
                                         out.data_ptr(), nbytes, scratch.data_ptr() if sbytes else None, sbytes,
                                         _stream()), "list_prep_percep_proj")
    return PercepProj(out, img, packed)       # (the scratch is released stream-ordered by the caching allocator)


# Scratch of the forward (X, H1, H2, point orders: 1.2 - 4.6 GB).  It lives in THREAD-LOCAL storage, keyed by
# (device, stream): two threads, or two streams of one thread, never share one (the kernels of a call keep using
# it after the call has returned), and it cannot outlive its thread -- nn.DataParallel starts fresh threads for
# every forward (reference train.py:126), whose workspaces go back to PyTorch's caching allocator when they end.
_tls = threading.local()


def _workspace(device, nbytes):
    cache = _tls.__dict__.setdefault("workspaces", {})
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = cache.get(key)
    if ws is None or ws.numel() < nbytes:
        cache.pop(key, None)                   # release the smaller one first
        ws = None
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=device)
        cache[key] = ws
    return ws


def release_workspaces():
    """Drop the calling thread's cached workspaces (other threads' die with them) and the idle side streams."""
    _tls.__dict__.pop("workspaces", None)
    with _aux_lock:
        _aux_pool.clear()


def _fill_query_args(query, perm, scale, vox, packed, precision, trans_mat=None, img=None,
                     percep_feat=None, clamp_hi=136.0, private_workspace=False):
    lib = load()
    _f32_cuda(query, "query")
    if query.dim() != 3 or query.shape[2] != 3:
        raise RuntimeError("query must be [B,N,3]")
    B, N, _ = query.shape
    a = ListQueryArgs()
    a.B, a.N = B, N
    a.query = query.data_ptr()
    a.q_sb, a.q_sn, a.q_sc = query.stride()
    for i in range(3):
        a.perm[i] = int(perm[i])
    a.scale = float(scale)
    keep = [query]
    if percep_feat is not None:
        _f32_cuda(percep_feat, "percep_feat")
        if percep_feat.dim() != 3 or percep_feat.shape[0] != B or percep_feat.shape[2] != N:
            raise RuntimeError("percep_feat must be [B,C,N]")
        a.percep_feat = percep_feat.data_ptr()
        a.pf_sb, a.pf_sc, a.pf_sn = percep_feat.stride()
        a.img_C = percep_feat.shape[1]
        keep.append(percep_feat)
    else:
        tm = _f32_cuda(trans_mat, "trans_mat").reshape(B, 4, 3).contiguous()
        a.trans_mat = tm.data_ptr()
        a.img_map = img.data.data_ptr()
        a.img_dtype = img.dtype
        a.map_size, a.img_C = img.map_size, img.channels
        a.clamp_hi = float(clamp_hi)
        if img.kept_C is not None:            # prep_img_proj's map: sampled channels | projected channels
            if img.packed is not packed:
                raise RuntimeError("the projected perceptual map was made for other packed weights")
            a.img_proj, a.img_kept_C = 1, int(img.kept_C)
        keep += [tm, img]
    for i in range(N_VOX_LEVELS):
        a.vox[i] = vox.levels[i]
    a.packed_mlp = packed.data.data_ptr()
    a.F, a.H1, a.H2, a.H3 = packed.F, packed.H1, packed.H2, packed.H3
    a.precision = PRECISIONS[precision] if isinstance(precision, str) else int(precision)
    if (a.precision == PREC_FP16) != bool(packed.fp16):
        raise RuntimeError("packed MLP weights were prepared for a different precision "
                           "(fp16 vs bf16 planes); call prep_mlp_weights(..., precision=...) again")
    nbytes = lib.list_query_workspace_bytes(B * N, a.F, a.H1, a.H2, a.H3)
    # a call whose backward will run later keeps its own workspace (X, H1, H2, point order live there)
    ws = (torch.empty((nbytes,), dtype=torch.uint8, device=query.device) if private_workspace
          else _workspace(query.device, nbytes))
    a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
    keep.append(ws)
    return a, keep


def keeps_no_activations(save_for_backward=False):
    """What sdf_query puts into ListQueryArgs.no_activations: a forward that is not kept for the backward lets the
    library run fc_1 + fc_2 + fc_out as one kernel (fp16 operands; include/list_hip.h).  LIST_FUSED_TAIL=0 in the
    environment keeps the two launches (A/B runs)."""
    return not save_for_backward and os.environ.get("LIST_FUSED_TAIL", "1") != "0"


def query_chunks(n_points, packed):
    """Row chunks list_sdf_query_fwd cuts a query of n_points into (with the workspace sdf_query gives it)."""
    lib = load()
    nbytes = lib.list_query_workspace_bytes(n_points, packed.F, packed.H1, packed.H2, packed.H3)
    rows = lib.list_query_chunk_rows(nbytes, n_points, packed.F, packed.H1, packed.H2, packed.H3)
    if rows <= 0:
        raise RuntimeError("list_query_chunk_rows failed")
    return (n_points + rows - 1) // rows


# diagnostics (bench.py --whole-model): a ctypes array of hipEvent_t handles (N_STAGES per row chunk) recorded by the
# next inference queries that pass no stage_events of their own -- the stage times INSIDE LIST.forward
STAGE_EVENTS_HOOK = None


class QueryContext:
    """What list_sdf_query_bwd needs from a forward call: its argument block (pointers into tensors
    kept alive here) including the private workspace."""

    def __init__(self, args, keep, vox, img):
        self.args, self.keep, self.vox, self.img = args, keep, vox, img


def sdf_query(query, trans_mat, img, vox, packed, perm=(2, 1, 0), scale=2.0, precision="bf16x3",
              percep_feat=None, out=None, stage_events=None, sort_points=True, clamp_hi=136.0,
              save_for_backward=False, percep_proj=None, plan=None, fused_fc0=True):
    """The fused hot path, network/models.py:91-97 -> sdf [B,N] (float32).

    plan: optional dict, filled with what the library dispatches for this call (list_query_plan: chunks,
    rows_per_chunk, fused_tail, fc0_k) -- per-kernel accounting reads it instead of guessing.

    stage_events: optional ctypes array (c_void_p * (n * N_STAGES)) of hipEvent_t handles, one set of N_STAGES
    per row chunk (query_chunks() says how many chunks a query of B*N points takes).
    save_for_backward: return (sdf, QueryContext) for sdf_query_backward."""
    lib = load()
    if query.dim() == 3 and query.shape[0] * query.shape[1] == 0:      # empty query -> empty field
        _f32_cuda(query, "query")
        empty = torch.empty(query.shape[:2], dtype=torch.float32, device=query.device)
        return (empty, None) if save_for_backward else empty
    a, keep = _fill_query_args(query, perm, scale, vox, packed, precision, trans_mat, img, percep_feat,
                               clamp_hi, private_workspace=save_for_backward)
    a.no_sort = 0 if sort_points else 1
    # nothing is kept for a backward: fc_1 / fc_2 / fc_out run as one kernel (fp16 operands), H2 stays in registers
    a.no_activations = int(keeps_no_activations(save_for_backward))
    a.no_fused_fc0 = 0 if fused_fc0 else 1            # (A/B runs: the 2-D gather kernel + k_gemm_nt_pp instead of k_fc0_fused)
    if a.img_proj and (save_for_backward or not a.no_activations or percep_proj is not None):
        raise RuntimeError("a map of prep_img_proj serves inference forwards only (no backward, no percep_proj)")
    if percep_proj is not None:
        if save_for_backward or percep_feat is not None:
            raise RuntimeError("percep_proj is an inference path: no backward, no pre-pooled features")
        if percep_proj.img is not img or percep_proj.packed is not packed:
            raise RuntimeError("percep_proj was made from another prepared map or other packed weights")
        a.percep_proj = percep_proj.data.data_ptr()
        keep.append(percep_proj)
    if stage_events is None and STAGE_EVENTS_HOOK is not None and not save_for_backward:
        stage_events = STAGE_EVENTS_HOOK     # diagnostics: stage times of calls made through the module API (bench.py)
    if stage_events is not None:
        a.stage_events = C.cast(stage_events, C.POINTER(C.c_void_p))
        a.stage_event_sets = max(1, len(stage_events) // N_STAGES)
    B, N = a.B, a.N
    sdf = out if out is not None else torch.empty((B, N), dtype=torch.float32, device=query.device)
    if not sdf.is_contiguous() or sdf.dtype != torch.float32 or tuple(sdf.shape) != (B, N):
        raise RuntimeError("out must be a contiguous float32 [B,N] tensor")
    a.sdf = sdf.data_ptr()
    if plan is not None:
        pl = ListQueryPlan()
        _check(lib.list_query_plan(C.byref(a), C.byref(pl)), "list_query_plan")
        plan.update(chunks=pl.chunks, rows_per_chunk=pl.rows_per_chunk, fused_tail=pl.fused_tail, fc0_k=pl.fc0_k,
                    box_levels=pl.box_levels, fused_fc0=pl.fused_fc0, img_proj=pl.img_proj)
    with torch.cuda.device(query.device):
        _check(lib.list_sdf_query_fwd(C.byref(a), _stream()), "list_sdf_query_fwd")
    if save_for_backward:
        a.stage_events = None
        return sdf, QueryContext(a, keep + [packed, sdf], vox, img)
    return sdf


# Side streams for the forked stages of list_sdf_query_bwd: a per-device pool of pairs, checked out for the
# duration of ONE call (the call joins them back into the caller's stream before it returns, so the next
# borrower's work simply queues behind on them).  Bounded by the number of concurrent callers, not by the
# number of threads that ever ran.
_aux_pool = {}
_aux_lock = threading.Lock()


class _AuxStreams:
    def __init__(self, device):
        self.device = device

    def __enter__(self):
        with _aux_lock:
            free = _aux_pool.setdefault(self.device.index, [])
            self.pair = free.pop() if free else None
        if self.pair is None:
            # the second stream carries dW0 and the 16^3 window level, the longest chain of the forked
            # backward: at high priority its workgroups get compute units first (-0.06 ms, measured)
            # (the third one, ABI 9: the 8^3 window level beside the gathers instead of behind them)
            self.pair = (torch.cuda.Stream(self.device), torch.cuda.Stream(self.device, priority=-1),
                         torch.cuda.Stream(self.device))
        return self.pair

    def __exit__(self, *exc):
        with _aux_lock:
            _aux_pool.setdefault(self.device.index, []).append(self.pair)
        return False


def sdf_query_backward(ctx, grad_sdf, packed_bwd, want_mlp=True, want_img=True, want_vox=True,
                       want_trans=True, stage_events=None, vox_adjoint="auto", overlap=True, img_levels_like=None,
                       want_img_map=True):
    """Backward of sdf_query (list_sdf_query_bwd).  Returns a dict:
      'mlp'       : {fc_0.weight [H1,F,1], fc_0.bias, ..., fc_out.bias} (reference layouts)
      'img_map'   : gradient of the prepared perceptual map, float32 [B,ms,ms,Ct]
      'vox'       : per level float32 [B,D,H,W,C] (channels-last; .permute(0,4,1,2,3) is the NCDHW view)
      'trans_mat' : [B,4,3]
      'img_levels': with img_levels_like (the encoder's 5 maps): their gradients, shaped and strided like them --
                    the adjoint resize then runs inside the call, beside the voxel scatters (else: img_map_grad_to_levels)
                    want_img_map=False (with img_levels_like): only the levels are wanted -- with fp16 operands the map
                    gradient between the two kernels is then kept as halfs at the gradient scale (ABI 6
                    grad_img_map_dtype: 0.6 GB less traffic per step at the metric shape) and 'img_map' is not returned
    For a forward with percep_feat (VoxelDecoder2.forward's own form) 'img_map'/'trans_mat' are replaced by
      'percep_feat': [B,img_C,N], the gradient of the pre-pooled features."""
    lib = load()
    a = ctx.args
    B, N = a.B, a.N
    dev = grad_sdf.device
    g = _f32_cuda(grad_sdf, "grad_sdf").reshape(B, N).contiguous()
    ga = ListQueryGradArgs()
    ga.fwd = C.pointer(a)
    ga.grad_sdf = g.data_ptr()
    ga.packed_mlp_bwd = packed_bwd.data_ptr()
    ga.vox_adjoint = VOX_ADJOINT[vox_adjoint]
    out = {}
    f32 = dict(dtype=torch.float32, device=dev)
    if a.percep_feat:
        want_trans = False
        if want_img:
            out["percep_feat"] = torch.empty((B, a.img_C, N), **f32)
            ga.grad_percep_feat = out["percep_feat"].data_ptr()
            ga.gpf_sb, ga.gpf_sc, ga.gpf_sn = out["percep_feat"].stride()
        want_img = False
    if want_mlp:
        m = {"fc_0.weight": torch.empty((a.H1, a.F, 1), **f32), "fc_0.bias": torch.empty((a.H1,), **f32),
             "fc_1.weight": torch.empty((a.H2, a.H1, 1), **f32), "fc_1.bias": torch.empty((a.H2,), **f32),
             "fc_2.weight": torch.empty((a.H3, a.H2, 1), **f32), "fc_2.bias": torch.empty((a.H3,), **f32),
             "fc_out.weight": torch.empty((1, a.H3, 1), **f32), "fc_out.bias": torch.empty((1,), **f32)}
        ga.mlp.w0, ga.mlp.b0 = m["fc_0.weight"].data_ptr(), m["fc_0.bias"].data_ptr()
        ga.mlp.w1, ga.mlp.b1 = m["fc_1.weight"].data_ptr(), m["fc_1.bias"].data_ptr()
        ga.mlp.w2, ga.mlp.b2 = m["fc_2.weight"].data_ptr(), m["fc_2.bias"].data_ptr()
        ga.mlp.w3, ga.mlp.b3 = m["fc_out.weight"].data_ptr(), m["fc_out.bias"].data_ptr()
        out["mlp"] = m
    keep = []
    if want_img:
        half_map = (not want_img_map and img_levels_like is not None and a.precision == PREC_FP16 and not a.no_sort
                    and B <= 64 and a.map_size * ((a.map_size + 3) // 4) <= 8192 and a.img_C % 4 == 0)
        if half_map:
            keep.append(torch.empty((B, a.map_size, a.map_size, a.img_C), dtype=torch.float16, device=dev))
            ga.grad_img_map, ga.grad_img_map_dtype = keep[-1].data_ptr(), MAP_F16
        else:
            out["img_map"] = torch.empty((B, a.map_size, a.map_size, a.img_C), **f32)
            ga.grad_img_map = out["img_map"].data_ptr()
    if want_img and img_levels_like is not None:
        maps, out["img_levels"] = _level_descriptors(img_levels_like, a.img_C)
        ga.grad_img_levels = maps
    if want_trans:
        out["trans_mat"] = torch.empty((B, 4, 3), **f32)
        ga.grad_trans_mat = out["trans_mat"].data_ptr()
    if want_vox:
        out["vox"] = []
        for i in range(N_VOX_LEVELS):
            lv = a.vox[i]
            t = torch.empty((B, lv.D, lv.H, lv.W, lv.C), **f32)
            out["vox"].append(t)
            ga.grad_vox[i] = ListVoxLevel(t.data_ptr(), lv.C, lv.D, lv.H, lv.W, MAP_F32, 0,
                                          lv.C * lv.D * lv.H * lv.W)
    nbytes = lib.list_query_bwd_workspace_bytes(B * N, a.F, a.H1, a.H2, a.H3, a.precision)
    if nbytes == 0:
        raise RuntimeError(f"list_sdf_query_bwd handles at most 262144 points per call (got {B * N})")
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    ga.workspace, ga.workspace_bytes = ws.data_ptr(), nbytes
    if stage_events is not None:
        ga.stage_events = C.cast(stage_events, C.POINTER(C.c_void_p))
    if overlap:        # dW0 | atomic scatters | window scatters | gathers run side by side (joined before return)
        with _AuxStreams(dev) as (a0, a1, a2), torch.cuda.device(dev):
            ga.aux_streams[0], ga.aux_streams[1] = a0.cuda_stream, a1.cuda_stream
            if os.environ.get("LIST_BWD_WIN2_OWN", "1") != "0":
                ga.aux_streams[2] = a2.cuda_stream
            # The library joins the side streams back into the caller's stream before it returns -- on success AND
            # on every error path -- so whatever runs on the caller's stream afterwards (including the caching
            # allocator handing these buffers to a later allocation on that stream) is ordered behind the side
            # work.  No Tensor.record_stream here: it would park the 4 GB of workspace / gradient buffers behind
            # side-stream events and make every step allocate afresh (measured: training step 8 -> 18 ms).
            _check(lib.list_sdf_query_bwd(C.byref(ga), _stream()), "list_sdf_query_bwd")
        return out
    with torch.cuda.device(dev):
        _check(lib.list_sdf_query_bwd(C.byref(ga), _stream()), "list_sdf_query_bwd")
    return out


def _level_descriptors(like, img_C):
    """Fresh gradients shaped and strided like the encoder's maps + their ListMap2D descriptors."""
    if len(like) != N_IMG_LEVELS:
        raise RuntimeError(f"need {N_IMG_LEVELS} image levels, got {len(like)}")
    maps = (ListMap2D * N_IMG_LEVELS)()
    outs = []
    for i, t in enumerate(like):
        o = torch.empty_like(t, dtype=torch.float32)
        outs.append(o)
        maps[i] = ListMap2D(o.data_ptr(), o.shape[1], o.shape[2], o.shape[3], *o.stride())
    if sum(o.shape[1] for o in outs) != img_C:
        raise RuntimeError("channel counts of `like` do not add up to the prepared map's")
    return maps, outs


def img_map_grad_to_levels(grad_img_map, like):
    """Adjoint of prep_img_maps: gradient of the prepared map [B,ms,ms,Ct] -> one gradient per encoder
    level, shaped (and strided) like the tensors in `like`."""
    lib = load()
    g = _f32_cuda(grad_img_map, "grad_img_map").contiguous()
    B, ms = g.shape[0], g.shape[1]
    maps, outs = _level_descriptors(like, g.shape[3])
    with torch.cuda.device(g.device):
        _check(lib.list_img_map_grad_to_levels(g.data_ptr(), B, ms, maps, _stream()),
               "list_img_map_grad_to_levels")
    return outs


def gemm_tn(a, b, precision="bf16x3"):
    """Diagnostic: out[M,N] = a[P,M]^T @ b[P,N] through the weight-gradient MFMA kernel."""
    lib = load()
    P, M = a.shape
    N = b.shape[1]
    if precision == "fp16":
        a_hi, b_hi = to_fp16(a), to_fp16(b)
        a_lo, b_lo = a_hi, b_hi
    else:
        a_hi, a_lo = split_bf16(a)
        b_hi, b_lo = split_bf16(b)
    out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    slab = torch.empty((128 * M * N,), dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        _check(lib.list_gemm_tn(a_hi.data_ptr(), a_lo.data_ptr(), b_hi.data_ptr(), b_lo.data_ptr(),
                                out.data_ptr(), slab.data_ptr(), slab.numel() * 4, M, N, P,
                                PRECISIONS[precision], _stream()), "list_gemm_tn")
    return out


def gather_features(query, trans_mat, img, vox, packed, perm=(2, 1, 0), scale=2.0,
                    percep_feat=None):
    """Diagnostic: the concatenated feature tensor of network/modules.py:275, [B,F,N]."""
    lib = load()
    a, keep = _fill_query_args(query, perm, scale, vox, packed, "fp16" if packed.fp16 else "bf16x3",
                               trans_mat, img, percep_feat)
    out = torch.zeros((a.B, a.F, a.N), dtype=torch.float32, device=query.device)
    with torch.cuda.device(query.device):
        _check(lib.list_gather_features_fwd(C.byref(a), out.data_ptr(), _stream()),
               "list_gather_features_fwd")
    return out


def percep_pool(pc, trans_mat, img, clamp_hi=136.0):
    """PerceptualPooling.forward on a prepared map -> [B,Ct,1,N] (network/modules.py:37-53)."""
    lib = load()
    _f32_cuda(pc, "pc")
    B, N, _ = pc.shape
    tm = _f32_cuda(trans_mat, "trans_mat").reshape(B, 4, 3).contiguous()
    out = torch.empty((B, img.channels, 1, N), dtype=torch.float32, device=pc.device)
    a = ListPoolArgs()
    a.B, a.N = B, N
    a.pc = pc.data_ptr()
    a.p_sb, a.p_sn, a.p_sc = pc.stride()
    a.trans_mat = tm.data_ptr()
    a.img_map, a.map_size, a.img_C = img.data.data_ptr(), img.map_size, img.channels
    a.img_dtype = img.dtype
    a.clamp_hi = float(clamp_hi)
    a.out = out.data_ptr()
    with torch.cuda.device(pc.device):
        _check(lib.list_percep_pool_fwd(C.byref(a), _stream()), "list_percep_pool_fwd")
    return out


def percep_pool_backward(pc, trans_mat, img, grad_out, want_img=True, want_trans=True, clamp_hi=136.0):
    """Backward of percep_pool (list_percep_pool_bwd): grad_out [B,Ct,1,N] or [B,Ct,N] ->
    {'img_map': [B,ms,ms,Ct] fp32, 'trans_mat': [B,4,3]}."""
    lib = load()
    _f32_cuda(pc, "pc")
    B, N, _ = pc.shape
    tm = _f32_cuda(trans_mat, "trans_mat").reshape(B, 4, 3).contiguous()
    g = _f32_cuda(grad_out, "grad_out").reshape(B, img.channels, N)
    fwd = ListPoolArgs()
    fwd.B, fwd.N = B, N
    fwd.pc = pc.data_ptr()
    fwd.p_sb, fwd.p_sn, fwd.p_sc = pc.stride()
    fwd.trans_mat = tm.data_ptr()
    fwd.img_map, fwd.map_size, fwd.img_C = img.data.data_ptr(), img.map_size, img.channels
    fwd.img_dtype = img.dtype
    fwd.clamp_hi = float(clamp_hi)
    ga = ListPoolGradArgs()
    ga.fwd = C.pointer(fwd)
    ga.grad_out = g.data_ptr()
    ga.g_sb, ga.g_sc, ga.g_sn = g.stride()
    out = {}
    if want_img:
        out["img_map"] = torch.empty((B, img.map_size, img.map_size, img.channels), dtype=torch.float32,
                                     device=pc.device)
        ga.grad_img_map = out["img_map"].data_ptr()
    if want_trans:
        out["trans_mat"] = torch.empty((B, 4, 3), dtype=torch.float32, device=pc.device)
        ga.grad_trans_mat = out["trans_mat"].data_ptr()
    nbytes = lib.list_percep_pool_bwd_workspace_bytes(B * N, img.channels)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=pc.device)
    ga.workspace, ga.workspace_bytes = ws.data_ptr(), nbytes
    with torch.cuda.device(pc.device):
        _check(lib.list_percep_pool_bwd(C.byref(ga), _stream()), "list_percep_pool_bwd")
    return out


def split_bf16(x):
    """float32 -> (hi, lo) bf16 planes as int16 tensors."""
    lib = load()
    x = _f32_cuda(x, "x").contiguous()
    hi = torch.empty(x.shape, dtype=torch.int16, device=x.device)
    lo = torch.empty(x.shape, dtype=torch.int16, device=x.device)
    with torch.cuda.device(x.device):
        _check(lib.list_split_bf16(x.data_ptr(), hi.data_ptr(), lo.data_ptr(), x.numel(), _stream()),
               "list_split_bf16")
    return hi, lo


def to_fp16(x):
    lib = load()
    x = _f32_cuda(x, "x").contiguous()
    out = torch.empty(x.shape, dtype=torch.int16, device=x.device)
    with torch.cuda.device(x.device):
        _check(lib.list_to_fp16(x.data_ptr(), out.data_ptr(), x.numel(), _stream()), "list_to_fp16")
    return out


def interleave_planes(hi, lo):
    """[R,K] hi / lo bf16 planes -> the interleaved layout of X and the packed fc_0 weight: per row and 32-column
    block, 32 hi halfs then 32 lo halfs (one 128-byte line)."""
    R, K = hi.shape
    if K % 32:
        raise RuntimeError("K must be a multiple of 32")
    return torch.stack((hi.reshape(R, K // 32, 32), lo.reshape(R, K // 32, 32)), dim=2).reshape(R, 2 * K).contiguous()


def gemm_nt(a, w, bias=None, relu=False, precision="bf16x3", plain_loop=False, interleaved=False):
    """Diagnostic: out[M,N] = act(a[M,K] @ w[N,K]^T + bias) through the MLP's MFMA kernel.
    plain_loop: force the 2-stage loop where the ping-pong schedule would be taken.
    interleaved (bf16 formats): hand the operands over in the hi / lo interleaved layout fc_0 reads."""
    lib = load()
    M, K = a.shape
    N = w.shape[0]
    if precision == "fp16":
        a_hi, w_hi = to_fp16(a), to_fp16(w)
        a_lo, w_lo = a_hi, w_hi
    else:
        a_hi, a_lo = split_bf16(a)
        w_hi, w_lo = split_bf16(w)
        if interleaved:
            a_hi = a_lo = interleave_planes(a_hi, a_lo)
            w_hi = w_lo = interleave_planes(w_hi, w_lo)
    out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    b = _f32_cuda(bias, "bias").contiguous() if bias is not None else None
    with torch.cuda.device(a.device):
        _check(lib.list_gemm_nt(a_hi.data_ptr(), a_lo.data_ptr(), w_hi.data_ptr(), w_lo.data_ptr(),
                                b.data_ptr() if b is not None else None, out.data_ptr(), M, N, K,
                                int(relu) | (2 if plain_loop else 0) | (4 if interleaved and precision != "fp16" else 0),
                                PRECISIONS[precision], _stream()),
               "list_gemm_nt")
    return out
