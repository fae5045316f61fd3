#!/usr/bin/env python3
"""bench.py -- SDF query-points/sec of the LIST hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic input that is
already resident in HBM in the REFERENCE's layout (NCHW / NCDHW fp32 feature maps, raw queries):
  layout hand-off (resize-to-137^2 + NHWC, NCDHW->NDHWC) -> weight repack -> gathers -> MLP -> sdf,
plus, for N > 1 ranks, the RCCL all-gather of the SDF shards (the loss-reduction exchange).
Each rank owns B images (weak scaling: the batch x query axis is sharded, no other collective).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W_FLOP_PER_PT = 2 * (3610 * 512 + 512 * 256 + 256 * 256 + 256)      # 4 090 368
def P_FLOP_FC0(points):              # algorithmic FLOP of one fc_0 launch (K = 3610, not the padded K)
    return points * 2 * 3610 * 512
W_BYTE_PER_PT = (7 * 8 * 369 + 4 * 1024) * 4 + 16                     # 99 056 fp32 maps
DETAIL_STEPS = 3                 # untimed steps that time each gather on its own (run_config)
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s
PEAK_BF16_TFLOPS = 2500.0        # dense bf16 MFMA
PEAK_ATOMIC_GBS = 1300.0         # MI355X_MICROARCH.md, Global float atomics: chip-wide rate of added bytes (measured)
# stated tolerances of the arithmetic modes against the fp32 reference
TOLERANCE = {"fp16": {"kind": "relative", "bound": 1e-3, "of": "max|sdf| of the compared points"},
             "bf16x3": {"kind": "absolute", "bound": 1e-4}, "bf16": {"kind": "absolute", "bound": 5e-3}}

WORKLOADS = {
    # name: (B per GPU, N, img_res, vox_res, map_size, clamp_hi)
    "list_im2sdf_b8_n20k_224": (8, 20000, 224, 128, 137, 136.0),            # BASELINE configs[1]
    "list_im2sdf_b8_n50k_512": (8, 50000, 512, 128, 274, 273.0),            # BASELINE configs[4]
    "list_grid256_b1": (1, 256 ** 3, 224, 128, 137, 136.0),                 # BASELINE configs[3]
}


class HipEvents:
    """hipEvent_t through ctypes (torch.cuda.Event exposes no stable raw handle before record)."""

    def __init__(self):
        # the HIP runtime instance torch already loaded (never a second copy of the runtime)
        path = next((l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l), "libamdhip64.so")
        self.rt = ctypes.CDLL(path)
        self.rt.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        self.rt.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self.rt.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p,
                                                ctypes.c_void_p]
        self.rt.hipEventDestroy.argtypes = [ctypes.c_void_p]

    def create(self):
        e = ctypes.c_void_p()
        rc = self.rt.hipEventCreate(ctypes.byref(e))
        assert rc == 0, rc
        return e

    def record(self, e):
        rc = self.rt.hipEventRecord(e, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, rc

    def elapsed_ms(self, a, b):
        ms = ctypes.c_float()
        rc = self.rt.hipEventElapsedTime(ctypes.byref(ms), a, b)
        assert rc == 0, rc
        return ms.value


def make_inputs(workload, rank, device, batch=None):
    """Synthetic inputs of SURVEY 8d, generated on the device (torch RNG, seed 333 + rank):
    image maps N(0,1), voxel level 0 U(0,1), levels 1-5 N(0,1), queries U(-0.5,0.5)^3, synthetic camera.
    `batch`: images of this rank (strong scaling: this rank's share of the global batch) instead of the workload's."""
    from list_amd import synthetic as synth        # shapes + exact weight / camera generators (no oracle/ in the GPU leg)
    B, N, img_res, vox_res, map_size, clamp_hi = WORKLOADS[workload]
    if batch is not None:
        B = batch
    g = torch.Generator(device=device)
    g.manual_seed(333 + rank)
    img_maps = [torch.randn(s, generator=g, device=device) for s in synth.img_map_shapes(B, img_res)]
    vshapes = synth.vox_map_shapes(B, vox_res)
    vox_maps = [torch.rand(vshapes[0], generator=g, device=device)]
    vox_maps += [torch.randn(s, generator=g, device=device) for s in vshapes[1:]]
    if workload.startswith("list_grid"):      # inference: the regular query grid of executors.py:191-197
        from list_amd import utils
        res = round(N ** (1 / 3))
        query = utils.grid_points_on_device(-0.5, 0.5, res, device).unsqueeze(0)
    else:
        query = torch.rand((B, N, 3), generator=g, device=device) - 0.5
    trans = torch.from_numpy(synth.make_trans_mat(333 + rank, B)).to(device)
    if map_size != 137:                    # keep projections inside the larger map
        trans = trans * (map_size - 1) / 136.0
        trans[:, :, 2] = trans[:, :, 2] * 136.0 / (map_size - 1)
    weights = {k: torch.from_numpy(v).to(device) for k, v in synth.make_mlp_weights(333).items()}
    # the inference grid arrives in raster order: executors.LIST.predict_grid asks for ordered_points (no point sort)
    return dict(B=B, N=N, img_maps=img_maps, vox_maps=vox_maps, query=query, trans_mat=trans,
                weights=weights, map_size=map_size, clamp_hi=clamp_hi,
                ordered_points=workload.startswith("list_grid"))


# Algorithmic work per query point of every kernel (bytes for the HBM-bound gathers and
# layout kernels at s = 4 B per map element, FLOPs for the MFMA kernels).  B_IMG etc. per step.
VOX_C = [1, 16, 32, 64, 128, 128]


def kernel_table(B, N, img_res, vox_res, map_size, x_bytes_per_feature, map_bytes=4, proj=False, fused_tail=False):
    """name -> (bound, algorithmic units per step, what the units are).
    Layout kernels: HBM bytes that must move (source read once + prepared map written once).
    MFMA kernels: SURVEY 8d FLOPs (K = 3610, not the padded K).
    Gathers: the bytes their taps ask for at the STORED element size (2 B for fp16 maps) plus the X bytes they
    write.  Most taps are L1/L2 hits, so this is a request rate ("l2" bound: the ~10 TB/s the L2s deliver to the
    CUs on this access pattern), never a fraction of the HBM peak; the HBM bytes of a gather
    come from the PMC counters (`hbm_bytes_pmc`, profiles/pmc_traffic.json)."""
    P = B * N
    from list_amd import synthetic as synth
    vox_elems = [int(np.prod(s)) for s in synth.vox_map_shapes(B, vox_res)]
    img_elems_in = sum(int(np.prod(s)) for s in synth.img_map_shapes(B, img_res))
    img_elems_out = B * map_size * map_size * 1024
    # proj (inference grid, list_prep_percep_proj): the perceptual block of fc_0 is applied to the map once per image;
    # the per-point sample reads 4 taps of 512 projected channels and fc_0 runs K = 3610 - 1024 columns: the FLOPs and
    # bytes below are what the kernels then execute
    k0 = 3610 - 1024 if proj else 3610
    img_tap = 512 * map_bytes if proj else 1024 * map_bytes
    img_x = 512 * 4 if proj else 1024 * x_bytes_per_feature
    t = {
        "prep_img_resize_nhwc": ("hbm", 4 * img_elems_in + map_bytes * img_elems_out, "bytes moved"),
        "prep_vox_ndhwc": ("hbm", (4 + map_bytes) * sum(vox_elems[1:]), "bytes moved"),
        "gather_img": ("l2", P * (4 * img_tap + img_x + 12), "tap + X bytes requested"),
        "gather_tail": ("l2", P * (7 * 8 * 4 + 48 * x_bytes_per_feature), "tap + X bytes requested"),
        "fc_0": ("mfma", P * 2 * k0 * 512, "FLOP"),
        # fused_tail (fp16 inference forwards, k_mlp_tail_f16): fc_1, fc_2 and fc_out are ONE launch, timed as the
        # fc_2_out interval; nothing is launched in the fc_1 interval (run_config folds it into fc_2_out)
        "fc_1": ("mfma", 0 if fused_tail else P * 2 * 512 * 256, "FLOP"),
        "fc_2_out": ("mfma", P * 2 * (256 * 256 + 256) + (P * 2 * 512 * 256 if fused_tail else 0), "FLOP"),
    }
    for i, c in enumerate(VOX_C[1:], 1):
        # the two coarsest levels share taps between the 7 stencil samples: 32 distinct taps instead of 56
        taps = 32 if i >= 4 and vox_res // (1 << (i - 1)) <= 16 else 56
        t[f"gather_vox_l{i}"] = ("l2", P * (taps * c * map_bytes + 7 * c * x_bytes_per_feature),
                                 "tap + X bytes requested")
    return t


# LIST_BENCH_FORCE_EXCHANGE=1: a ONE-rank process group takes the N > 1 code path (process group, exchange buffers,
# asynchronous all-gather, exchange check, rank census): the RCCL pre-flight on a one-GPU box
# (tests/test_rccl_preflight_gpu.py, with LIST_FORCE_COLLECTIVES=1 for list_amd.parallel).  Never a scaling number.
FORCE_EXCHANGE = os.environ.get("LIST_BENCH_FORCE_EXCHANGE", "0") == "1"


def run_config(args, precision, steps, warmup, inp, hip, ev, world, device, gather_fn, sustained_steps=0, fused_fc0=True,
               img_proj=None):
    import torch
    import torch.distributed as dist
    multi = world > 1 or FORCE_EXCHANGE
    B, N = inp["B"], inp["N"]
    # img_proj (list_prep_img_proj, ABI 8): the low-resolution encoder levels go through their columns of fc_0 BEFORE the
    # resize (inside the timed step, in place of list_prep_img_maps).  None = what the module API does for an inference
    # forward in this precision (hip.img_proj_default: the bf16 formats; fp16 keeps the 2-D sample inside fc_0)
    if img_proj is None:
        img_proj = hip.img_proj_default(precision)
    img_proj = bool(img_proj) and not inp.get("ordered_points") and \
        hip.img_proj_kept_levels(inp["img_maps"], inp["map_size"]) < hip.N_IMG_LEVELS
    n_ev = hip.N_STAGES
    # one event set per row chunk of the call (ListQueryArgs.stage_event_sets): the intervals of all chunks are
    # summed, so kernel_ms is the time of ALL launches of a kernel in one step
    n_chunks = -(-B * N // 262144)
    # timed region: the seven gathers of a chunk are independent launches, dispatched without the queue barrier between
    # them (include/list_hip.h, ListStage), so no event is recorded between them -- their group is one interval,
    # [SORT, TAIL]; their individual durations come from DETAIL_STEPS further steps after the timed region, with all
    # events (which puts the barriers back)
    apart = set(range(hip.STAGE_VOX0, hip.STAGE_IMG + 1))
    def event_sets(coarse):
        pre = [ev.create() for _ in range(5)]                  # [4]: behind the step's last launch
        arr = (ctypes.c_void_p * (n_ev * n_chunks))(
            *[None if (coarse and i % n_ev in apart) else ev.create() for i in range(n_ev * n_chunks)])
        return pre, arr
    step_events = [event_sets(True) for _ in range(steps)]
    detail_events = [event_sets(False) for _ in range(DETAIL_STEPS)]
    # N > 1: the exchange of step i (one RCCL all-gather of the SDF shards) runs on RCCL's stream beside the kernels
    # of step i+1; two output buffers alternate, and a step first orders itself after the gather that last read its buffer
    # (strong scaling: the global batch may not divide by the ranks -- every rank exchanges a buffer of B_pad =
    # ceil(B_global / world) images, of which it fills its own B)
    B_pad = inp.get("B_pad", B)
    gathered = [torch.empty((world * B_pad, N), dtype=torch.float32, device=device) for _ in range(2)] if multi else None
    sdf_bufs = [torch.zeros((B_pad, N), dtype=torch.float32, device=device) for _ in range(2 if multi else 1)]
    sdfs = [b[:B] for b in sdf_bufs]
    pending = [None, None]
    n_calls = [0]
    overlap_exchange = [True]

    plan = {}                    # what the library dispatched for the step's query (list_query_plan)

    def step(events=None):
        pre, arr = events if events else (None, None)
        k = n_calls[0] % len(sdfs)
        n_calls[0] += 1
        sdf = sdfs[k]
        if pending[k] is not None:
            pending[k].wait()
            pending[k] = None
        if pre: ev.record(pre[0])
        md = hip.map_dtype_for(precision)
        if img_proj:
            # the projection needs the packed weights: hand-off of the voxel levels, weight repack, then the 2-D side
            # (intervals [0,1] = voxel hand-off, [1,2] = weights, [2,3] = resize + projection; renamed below)
            vox = hip.prep_vox_maps(inp["vox_maps"], md)
            if pre: ev.record(pre[1])
            packed = hip.prep_mlp_weights(inp["weights"], vox.channels, sum(m.shape[1] for m in inp["img_maps"]), precision)
            if pre: ev.record(pre[2])
            img = hip.prep_img_proj(inp["img_maps"], packed, inp["map_size"], precision)
        else:
            img = hip.prep_img_maps(inp["img_maps"], inp["map_size"], md)
            if pre: ev.record(pre[1])
            vox = hip.prep_vox_maps(inp["vox_maps"], md)
            if pre: ev.record(pre[2])
            packed = hip.prep_mlp_weights(inp["weights"], vox.channels, img.channels, precision)
        # inference grid (executors.LIST.predict_grid): many points on one image -- the perceptual block of fc_0 is
        # applied to the 137^2 map once (inside the timed step, counted with prep_weights) and sampled per point
        proj = hip.prep_percep_proj(img, packed, precision) if inp.get("ordered_points") else None
        if pre: ev.record(pre[3])
        hip.sdf_query(inp["query"], inp["trans_mat"], img, vox, packed, precision=precision,
                      out=sdf, stage_events=arr, clamp_hi=inp["clamp_hi"],
                      sort_points=not inp.get("ordered_points", False), percep_proj=proj, plan=plan, fused_fc0=fused_fc0)
        if multi:
            if overlap_exchange[0]:
                try:
                    _, pending[k] = gather_fn(sdf_bufs[k], out=gathered[k], async_op=True)
                except (RuntimeError, TypeError) as e:          # no work handle from this backend build: in-line exchange
                    overlap_exchange[0] = False
                    print(f"bench: asynchronous all-gather unavailable ({e}); exchanging in line", file=sys.stderr)
            if not overlap_exchange[0]:
                gather_fn(sdf_bufs[k], out=gathered[k])
        if pre: ev.record(pre[4])
        return sdf

    def drain():
        for k in range(2):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None

    for _ in range(warmup):
        step()
    drain()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(step_events[i])
    drain()                                   # every exchange has completed inside the timed region
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64,
                         device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed_local, elapsed = elapsed, float(t.item())
    else:
        elapsed_local = elapsed
    for ev_set in detail_events:              # untimed: per-gather durations, the gathers one after the other
        step(ev_set)
    drain()
    torch.cuda.synchronize()
    names = ["prep_img_resize_nhwc", "prep_vox_ndhwc", "prep_weights"] + list(hip.STAGE_NAMES)
    acc = np.zeros(len(names))
    group = 0.0
    first = hip.STAGE_VOX0 - 1                  # the event behind the point sort
    def interval(arr, c, s0, s1):
        return ev.elapsed_ms(ctypes.c_void_p(arr[c * n_ev + s0]), ctypes.c_void_p(arr[c * n_ev + s1]))
    for pre, arr in step_events:
        i_img, i_vox, i_w = (2, 0, 1) if img_proj else (0, 1, 2)          # which interval holds which prep (see step())
        acc[0] += ev.elapsed_ms(pre[i_img], pre[i_img + 1])
        acc[1] += ev.elapsed_ms(pre[i_vox], pre[i_vox + 1])
        acc[2] += ev.elapsed_ms(pre[i_w], pre[i_w + 1])
        for c in range(n_chunks):
            group += interval(arr, c, first, hip.STAGE_IMG + 1)
            for s in range(n_ev - 1):
                if s not in apart and s + 1 not in apart:
                    acc[3 + s] += interval(arr, c, s, s + 1)
    acc /= steps
    for pre, arr in detail_events:
        for c in range(n_chunks):
            for s in range(first, hip.STAGE_IMG + 1):
                acc[3 + s] += interval(arr, c, s, s + 1) / DETAIL_STEPS
    kernel_ms = dict(zip(names, acc.tolist()))
    assert plan["chunks"] == n_chunks, (plan, n_chunks)
    if plan["fused_tail"]:
        # one launch for fc_1 + fc_2 + fc_out (as the LIBRARY dispatched it, list_query_plan -- not re-derived from the
        # precision and the environment): the fc_1 interval holds two event records and no kernel
        assert kernel_ms["fc_1"] < 0.02 * n_chunks, kernel_ms["fc_1"]       # (two event records per row chunk)
        kernel_ms["fc_2_out"] += kernel_ms["fc_1"]
        kernel_ms["fc_1"] = 0.0
        kernel_ms["_fused_tail"] = 1
    assert bool(plan.get("img_proj")) == img_proj, (plan, img_proj)
    if img_proj:
        kernel_ms["_img_proj"] = 1          # prep_img_resize_nhwc = resize of the kept levels + projections + their resized sum
        kernel_ms["_fc0_k"] = plan["fc0_k"]  # K columns fc_0 runs over (the projected levels' are left out)
    if plan.get("fused_fc0"):
        # fc_0 produced the perceptual block of its A operand on chip (k_fc0_fused): no 2-D gather kernel ran
        kernel_ms["_fused_fc0"] = 1
    kernel_ms["gathers_back_to_back"] = group / steps       # the same seven launches as they run in the timed region
    kernel_ms["gathers_one_by_one_sum"] = float(sum(acc[3 + s] for s in range(first, hip.STAGE_IMG + 1)))   # (untimed steps)
    kernel_ms["_launches_per_step"] = n_chunks
    # HIP-event time of every timed step (first launch .. behind the last one, on the stream they run on)
    step_ms = sorted(ev.elapsed_ms(pre[0], pre[4]) for pre, _ in step_events)
    q = lambda f: step_ms[min(len(step_ms) - 1, int(f * len(step_ms)))]
    extra = {"elapsed_local_s": elapsed_local}
    if multi:
        # once, outside the timed region: what the exchange delivered holds this rank's shard at this rank's offset, and
        # every other rank's rows are finite numbers (not the zeros the buffer was born with)
        rank = dist.get_rank()
        kk = (n_calls[0] - 1) % len(sdfs)
        got = gathered[kk]
        own_ok = bool(torch.equal(got[rank * B_pad: rank * B_pad + B], sdfs[kk]))
        finite = bool(torch.isfinite(got).all())
        filled = all(bool((got[r * B_pad] != 0).any()) for r in range(world))
        extra["exchange_check"] = {"own_shard_at_own_offset": own_ok, "all_finite": finite,
                                   "every_rank_delivered": filled, "rows_per_rank_in_buffer": B_pad}
        assert own_ok and finite and filled, extra["exchange_check"]
    extra["step_events_ms"] = {"median": q(0.5), "p10": q(0.1), "p90": q(0.9), "min": step_ms[0], "max": step_ms[-1],
                                    "n": len(step_ms)}
    last = sdfs[(n_calls[0] - 1) % len(sdfs)]
    if sustained_steps > 0:
        # untimed-in-headline: the clock the chip settles to under the load (a 20-step burst is 40 ms).  Same step,
        # `sustained_steps` times back to back; stage events on the last 16 steps only
        ring = [event_sets(True) for _ in range(16)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(sustained_steps):
            step(ring[i - (sustained_steps - 16)] if i >= sustained_steps - 16 else None)
        drain()
        torch.cuda.synchronize()
        s_el = time.perf_counter() - t0
        fc0 = sum(interval(arr, c, hip.STAGE_NAMES.index("fc_0"), hip.STAGE_NAMES.index("fc_0") + 1)
                  for _, arr in ring for c in range(n_chunks)) / len(ring)
        st = sorted(ev.elapsed_ms(pre[0], pre[4]) for pre, _ in ring)
        extra["sustained"] = {"steps": sustained_steps, "seconds": s_el, "ms_per_step": s_el / sustained_steps * 1e3,
                                   "value_per_gpu": B * N * sustained_steps / s_el, "fc_0_ms": fc0,
                                   "step_events_median_ms_last16": st[len(st) // 2]}
    return elapsed, kernel_ms, last, extra


def run_train_step(precision, steps, warmup, inp, hip, ev):
    """one training step of the path = layout hand-off + weight repacks + forward (activations
    kept) + list_sdf_query_bwd + adjoint resize, i.e. every gradient the reference's autograd produces for
    the path (MLP parameters, 5 image maps, 6 voxel maps, trans_mat).  d(loss)/d(sdf) is synthetic."""
    B, N = inp["B"], inp["N"]
    md = hip.map_dtype_for(precision)
    g = torch.Generator(device=inp["query"].device)
    g.manual_seed(4242)
    gsdf = torch.randn((B, N), generator=g, device=inp["query"].device) / B
    n_ev = hip.N_BWD_STAGES
    acc = np.zeros(n_ev - 1)
    fwd_ms = bwd_ms = 0.0
    grads = None

    def step(timed):
        nonlocal grads
        arr = (ctypes.c_void_p * n_ev)(*[ev.create() for _ in range(n_ev)]) if timed else None
        e = [ev.create() for _ in range(3)] if timed else None
        img = hip.prep_img_maps(inp["img_maps"], inp["map_size"], md)
        vox = hip.prep_vox_maps(inp["vox_maps"], md)
        packed = hip.prep_mlp_weights(inp["weights"], vox.channels, img.channels, precision)
        packed_b = hip.prep_mlp_weights_bwd(inp["weights"], vox.channels, img.channels, precision)
        if timed: ev.record(e[0])
        sdf, ctx = hip.sdf_query(inp["query"], inp["trans_mat"], img, vox, packed, precision=precision,
                                 save_for_backward=True, clamp_hi=inp["clamp_hi"])
        if timed: ev.record(e[1])
        out = hip.sdf_query_backward(ctx, gsdf, packed_b, stage_events=arr, img_levels_like=inp["img_maps"],
                                     want_img_map=False)      # the training step wants the encoder levels' gradients
        if timed: ev.record(e[2])
        grads = out
        return arr, e

    for _ in range(warmup):
        step(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    recs = [step(True) for _ in range(steps)]
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    for arr, e in recs:
        for s in range(n_ev - 1):
            acc[s] += ev.elapsed_ms(ctypes.c_void_p(arr[s]), ctypes.c_void_p(arr[s + 1]))
        fwd_ms += ev.elapsed_ms(e[0], e[1])
        bwd_ms += ev.elapsed_ms(e[1], e[2])
    kernel_ms = dict(zip(hip.BWD_STAGE_NAMES, (acc / steps).tolist()))
    # per-stage durations with the stages IN LINE on one stream (3 untimed steps): what each kernel group takes on its
    # own -- the forked step above overlaps them, so its main-stream intervals are not kernel times
    inl = np.zeros(n_ev - 1)
    for _ in range(3):
        arr = (ctypes.c_void_p * n_ev)(*[ev.create() for _ in range(n_ev)])
        img = hip.prep_img_maps(inp["img_maps"], inp["map_size"], md)
        vox = hip.prep_vox_maps(inp["vox_maps"], md)
        packed = hip.prep_mlp_weights(inp["weights"], vox.channels, img.channels, precision)
        packed_b = hip.prep_mlp_weights_bwd(inp["weights"], vox.channels, img.channels, precision)
        _, ctx = hip.sdf_query(inp["query"], inp["trans_mat"], img, vox, packed, precision=precision,
                               save_for_backward=True, clamp_hi=inp["clamp_hi"])
        hip.sdf_query_backward(ctx, gsdf, packed_b, stage_events=arr, img_levels_like=inp["img_maps"], overlap=False,
                               want_img_map=False)
        torch.cuda.synchronize()
        for s in range(n_ev - 1):
            inl[s] += ev.elapsed_ms(ctypes.c_void_p(arr[s]), ctypes.c_void_p(arr[s + 1])) / 3
    inline_ms = dict(zip(hip.BWD_STAGE_NAMES, inl.tolist()))
    P = B * N
    flop0 = P * 2 * 3610 * 512
    dom = max(("dgrad_fc0", "wgrad_fc0"), key=lambda k: inline_ms[k])
    products = 3 if precision == "bf16x3" else 1
    ach = flop0 / (inline_ms[dom] * 1e-3) / 1e12
    # bytes the three direct (global-atomic) voxel levels add: 56 taps x C channels per point, fp32 -- the 64^3 x 32 level
    # as packed halfs with fp16 operands; the window levels' flushes and the gathered 32^3 level come on top
    direct = P * 56 * (1 * 4 + 16 * 4 + 32 * (2 if precision == "fp16" else 4))
    roofline = {"kernel": {"dgrad_fc0": "k_gemm_nt_pp / k_gemm_nt16 (dX = dZ1 . W0)", "wgrad_fc0": ("k_gemm_tn" if precision == "fp16" else "k_gemm_tn16") + " (dW0 = dZ1^T . X)"}[dom],   # shape per format
                "bound": "mfma", "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS,
                "traffic": None, "launch_ms": inline_ms[dom], "algorithmic_flop_per_launch": flop0,
                "mfma_products_per_mac": products, "timed": "in line on one stream, 3 untimed steps (stage events)",
                "other_stage": {"stage": "scatter_vox (in line)", "ms": inline_ms["scatter_vox"],
                                "bound": "hbm: global float atomics", "peak": PEAK_ATOMIC_GBS, "unit": "GB/s of added bytes",
                                "direct_levels_atomic_bytes": direct,
                                "achieved_lower_bound": direct / (inline_ms["scatter_vox"] * 1e-3) / 1e9,
                                "frac_lower_bound": direct / (inline_ms["scatter_vox"] * 1e-3) / 1e9 / PEAK_ATOMIC_GBS,
                                "note": "the stage also holds the two coarse levels (matrix-core adjoint, fp16: flush atomics, data dependent) "
                                        "and the voxel-side gather of the 32^3 level: the fraction is a lower bound"}}
    assert 0.0 < roofline["frac"] <= 1.0
    return {"precision": precision, "steps": steps, "ms_per_step": elapsed / steps * 1e3,
            "roofline": roofline, "kernel_ms_inline": inline_ms,
            "value": B * N * steps / elapsed, "unit": "query-points/s (forward + backward)",
            "forward_query_ms": fwd_ms / steps, "backward_ms": bwd_ms / steps,
            "kernel_ms": kernel_ms,
            "kernel_ms_note": "stage intervals on the main stream; dW0, the atomic and the LDS-window voxel levels "
                              "run concurrently on three auxiliary streams; on the main stream the adjoint resize to the 5 "
                              "encoder levels runs between the map gradient and trans_mat_grad (which waits for dW0) "
                              "inside the same call, so the stages overlap and do not add up to backward_ms",
            "outputs": "d fc_0..fc_out (reference layout), d 5 image maps, d 6 voxel maps, d trans_mat"}, grads


def _on_clamp(query, trans_mat, clamp_hi=136.0):
    """Fraction of the query points whose projection (network/modules.py:37-43) sits on the clamp."""
    q = query[:, :, [2, 1, 0]] * 2
    h = torch.cat([q, torch.ones_like(q[..., :1])], -1) @ trans_mat.reshape(-1, 4, 3)
    uv = h[..., :2] / (h[..., 2:3] + 1e-8)
    on = ((uv <= 0) | (uv >= clamp_hi)).any(-1)
    return on.float().mean()


def run_whole_model(workload, precision, device, steps=5, warmup=2):
    """the whole LIST.forward (ResNet encoders, coarse point decoder, on-device create_occ, 3-D
    encoder, spatial transformer AND the HIP query path) on the same B x N, so the weight of the path
    inside the model is visible.  Random-init weights (seed 333), images U[0,1)."""
    from list_amd import arguments, utils
    B, N, img_res, vox_res, _, _ = WORKLOADS[workload]
    g = torch.Generator(device=device)
    g.manual_seed(333)
    img = torch.rand((B, 3, img_res, img_res), generator=g, device=device)
    query = torch.rand((B, N, 3), generator=g, device=device) - 0.5

    def one(vox_encoder_precision, state=None):
        torch.manual_seed(333)
        cfg = arguments.default_config(vox_res=vox_res, train_batch_size=B, precision=precision, img_res=img_res,
                                       vox_encoder_precision=vox_encoder_precision)
        net = utils.get_class("network.models.LIST")(cfg).to(device).eval()
        if state is not None:
            net.load_state_dict(state)
        times = {"encode": 0.0, "query": 0.0}
        with torch.no_grad():
            for it in range(warmup + steps):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                feat_l2, vox_feat, tm, _, _ = net.encode(img)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                sdf = net.query_sdf(query, feat_l2, vox_feat, tm)
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                if it >= warmup:
                    times["encode"] += t1 - t0
                    times["query"] += t2 - t1
        total = (times["encode"] + times["query"]) / steps
        # ---- where the query's wall time goes (2.40 ms here against 1.80 ms for the same B x N in the
        # bench's own channels-last step).  (i) the call timed above starts on an IDLE device behind a synchronize: its
        # first launch waits for the Python dispatch; (ii) the same call with the device kept busy (no synchronize
        # between encode and query: what train.py / test.py do), as GPU time between two events; (iii) its stages
        # (hip.STAGE_EVENTS_HOOK); (iv) the same module call on the SYNTHETIC camera of SURVEY 8d instead of the
        # untrained spatial transformer's output, whose projections pile onto the clamp.
        from list_amd import hip as _hip
        from list_amd import synthetic as _synth
        evs = HipEvents()
        diag = {}
        with torch.no_grad():
            feat_l2, vox_feat, tm, _, _ = net.encode(img)
            tm_syn = torch.from_numpy(_synth.make_trans_mat(333, B)).to(device)
            for name, T in (("model_trans_mat", tm), ("synthetic_camera", tm_syn)):
                gpu, stages = [], np.zeros(_hip.N_STAGES - 1)
                for it in range(6):
                    arr = (ctypes.c_void_p * _hip.N_STAGES)(*[evs.create() for _ in range(_hip.N_STAGES)]) if it >= 3 else None
                    _hip.STAGE_EVENTS_HOOK = arr
                    f2 = [m.clone(memory_format=torch.preserve_format) for m in feat_l2]     # new maps: the caches miss, as in a forward
                    e0, e1 = evs.create(), evs.create()
                    evs.record(e0)
                    net.query_sdf(query, f2, vox_feat, T)
                    evs.record(e1)
                    _hip.STAGE_EVENTS_HOOK = None
                    torch.cuda.synchronize()
                    if it >= 3:
                        gpu.append(evs.elapsed_ms(e0, e1))
                        # (stage events between the gathers put their queue barriers back: +2 % on the group)
                        stages += np.array([evs.elapsed_ms(ctypes.c_void_p(arr[i]), ctypes.c_void_p(arr[i + 1]))
                                            for i in range(_hip.N_STAGES - 1)]) / 3
                st = dict(zip(_hip.STAGE_NAMES, [round(float(x), 4) for x in stages]))
                diag[name] = {"gpu_ms_first_to_last_event": round(float(np.mean(gpu)), 4),
                              "prep_and_dispatch_before_sort_ms": round(float(np.mean(gpu)) - float(stages.sum()), 4),
                              "stage_ms": st, "frac_points_on_clamp": float(_on_clamp(query, T))}
        diag["wall_from_idle_minus_gpu_ms"] = round(times["query"] / steps * 1e3 - diag["model_trans_mat"]["gpu_ms_first_to_last_event"], 4)
        diag["note"] = ("query_sdf_ms is wall clock of one call started on an idle device (synchronize on both sides): "
                        "host dispatch before the first launch is exposed; gpu_ms_first_to_last_event is the same call "
                        "between two events with the host running ahead")
        return net, sdf, {"ms_per_forward": total * 1e3, "value": B * N / total, "query_breakdown": diag,
                          # SURVEY 8d's second run: the query path fed by the model's own (untrained) spatial transformer
                          # and encoders instead of the synthetic camera and maps
                          "query_path_points_per_s": B * N / (times["query"] / steps),
                          "trans_mat": "output of the model's own untrained spatial_transformer",
                          "encode_ms": times["encode"] / steps * 1e3, "query_sdf_ms": times["query"] / steps * 1e3,
                          "finite": bool(torch.isfinite(sdf).all())}

    net, sdf, r = one("fp32")
    out = {"model": "network.models.LIST (random init, eval)", "precision": precision,
           "unit": "query-points/s (whole LIST.forward)", "channels_last_encoders": bool(net.channels_last), **r}
    # SURVEY 8 f2, "optionally half precision": the 3-D encoder under autocast hands fp16 channels-last levels over
    _, sdf_h, rh = one("fp16", net.state_dict())
    rh["max_abs_sdf_diff_vs_fp32_encoder"] = float((sdf_h - sdf).abs().max())
    out["vox_encoder_fp16"] = rh
    return out


def pmc_traffic(precision, workload, batch):
    """HBM bytes per launch from the committed PMC passes (tools/pmc_traffic.py), or {}.  The counters were collected
    on the metric workload at ITS batch per GPU: any other workload, or another number of images on this rank (strong
    scaling: 64 / 32 / 16 images per rank at N = 1 / 2 / 4), has no measured traffic and reports none."""
    tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if workload != "list_im2sdf_b8_n20k_224" or batch != WORKLOADS[workload][0] or not os.path.exists(tf):
        return {}
    data = json.load(open(tf))
    # the counters belong to the kernel sources they were collected on (tools/pmc_traffic.py stamps build.py's SHA-256):
    # stale numbers are dropped, never replayed
    from list_amd import build as _build
    if data.get("_source_fingerprint") != _build._fingerprint():
        return {}
    return data.get(precision, {})


def roofline_of(kernel_ms, table, precision, workload, batch):
    """Roofline entry of the dominant kernel: the longest-running kernel among those with a hardware roof
    (MFMA or HBM).  kernel_ms holds the time of all its launches in a step (one per row chunk); the algorithmic
    units in `table` are per step too, so achieved = units / time whatever the chunking."""
    launches = kernel_ms.get("_launches_per_step", 1)
    roofed = [k for k, (b, _, _) in table.items() if b in ("mfma", "hbm") and not k.startswith("prep_")]
    dom = max(roofed, key=lambda k: kernel_ms.get(k, 0.0))
    bound, units, _ = table[dom]
    secs = kernel_ms[dom] * 1e-3
    traffic = pmc_traffic(precision, workload, batch).get(dom)
    ach = units / secs / 1e12
    # fc_0 runs the ping-pong schedule in every precision (single plane, or hi / lo interleaved for the split formats)
    fc1 = "k_gemm_nt (fc_1 + ReLU)" if precision == "bf16x3" else "k_gemm_nt_pp (fc_1 + ReLU)"
    tail = "k_mlp_tail_f16 (fc_1 + ReLU + fc_2 + ReLU + fc_out)" if kernel_ms.get("_fused_tail") else "k_gemm_nt16 (fc_2 + ReLU + fc_out)"
    fc0 = ("k_fc0_fused (bilinear sample of the perceptual block into LDS + fc_0 + ReLU; the FLOPs are fc_0's)"
           if kernel_ms.get("_fused_fc0") else "k_gemm_nt_pp (fc_0 + ReLU)")
    if kernel_ms.get("_img_proj"):
        # list_prep_img_proj: the low-resolution encoder levels went through their fc_0 columns before the resize, so the
        # kernel EXECUTES fewer products than the reference's fc_0 it replaces (`executed_flop_per_launch`); `achieved`
        # stays the ALGORITHMIC rate (the reference op's FLOPs over the kernel's time), as the task defines it
        fc0 = ("k_fc0_fused<0, true> (kept encoder levels sampled into LDS + fc_0 over the unprojected K-tiles + bilinear "
               "sample of the projected levels in the epilogue + ReLU)" if kernel_ms.get("_fused_fc0")
               else "k_gemm_nt_pp (fc_0 over the unprojected K-tiles + row-vector epilogue + ReLU)")
    r = {"kernel": {"fc_0": fc0, "fc_1": fc1, "fc_2_out": tail}[dom],
         "bound": "mfma", "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
         "frac": ach / PEAK_BF16_TFLOPS, "traffic": traffic, "launches_per_step": launches,
         "launch_ms": kernel_ms[dom] / launches, "algorithmic_flop_per_launch": units / launches,
         "mfma_products_per_mac": 3 if precision == "bf16x3" else 1}
    if dom == "fc_0" and kernel_ms.get("_fc0_k"):
        r["executed_flop_per_launch"] = units / launches * kernel_ms["_fc0_k"] / 3648.0 * (3648.0 / 3610.0)
        r["executed_k_of_3648"] = kernel_ms["_fc0_k"]
    assert 0.0 < r["frac"] <= 1.0, f"roofline fraction {r['frac']} is not physical: accounting bug"
    return r


def path_roofs(value, precision, grid=False):
    """SURVEY 8d whole-path ceilings in query-points/s and where `value` stands against the binding one.

    The gather figure prices every tap REQUEST as HBM bytes (SURVEY 8d's per-point count: scattered queries share few
    taps).  grid=True (the dense 256^3 inference grid on ONE image, BASELINE config 4): neighbouring grid points share
    nearly all their taps and one image's maps stay cache-resident, so that figure is not a roof there -- the path has run
    above it since round 2b -- and is reported as what it is, a no-reuse request rate; the binding roof is the MFMA one,
    on the FLOPs the path executes with the perceptual block of fc_0 applied to the map (K = 3610 - 1024)."""
    map_bytes = 2 if precision == "fp16" else 4
    w_byte = (7 * 8 * 369 + 4 * 1024) * map_bytes + 16
    products = 3 if precision == "bf16x3" else 1
    gather_roof = PEAK_HBM_GBS * 1e9 / w_byte
    flop = W_FLOP_PER_PT - 2 * 1024 * 512 if grid else W_FLOP_PER_PT
    mfma_roof = PEAK_BF16_TFLOPS * 1e12 / (flop * products)
    if grid:
        r = {"no_reuse_tap_request_points_per_s": gather_roof, "gather_bytes_per_point_if_no_tap_were_shared": w_byte,
             "mfma_roof_points_per_s": mfma_roof, "mfma_flop_per_point_executed": flop, "mfma_products_per_mac": products,
             "binding": "mfma (taps of a dense grid on one image are served by the caches)",
             "whole_path_frac_of_binding_roof": value / mfma_roof}
    else:
        bind = min(gather_roof, mfma_roof)
        r = {"gather_roof_points_per_s": gather_roof, "gather_bytes_per_point": w_byte,
             "mfma_roof_points_per_s": mfma_roof, "mfma_products_per_mac": products,
             "binding": "hbm gather" if gather_roof <= mfma_roof else "mfma",
             "whole_path_frac_of_binding_roof": value / bind}
    assert 0.0 < r["whole_path_frac_of_binding_roof"] <= 1.0, \
        f"whole-path fraction {r['whole_path_frac_of_binding_roof']} of its binding roof is not physical: accounting bug"
    return r


def self_launch(n, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks with torch.distributed.run as a child process (the
    driver's own N > 1 command line), pass its output through and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:                       # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this driver
    print("bench: no launcher in the environment, starting " + " ".join(cmd), file=sys.stderr)
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="list_im2sdf_b8_n20k_224", choices=sorted(WORKLOADS))
    ap.add_argument("--precision", default=None, choices=["fp16", "bf16x3", "bf16"],
                    help="MLP arithmetic; default: headline fp16 plus a shorter bf16x3 run reported under 'alt'")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-channels-last-alt", action="store_true")
    ap.add_argument("--no-train-step", action="store_true")
    ap.add_argument("--whole-model", action="store_true",
                    help="also time the whole LIST.forward (encoders included) on the same B x N")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank owns the workload's B images (BASELINE config 2 per GPU); strong: BASELINE "
                         "config 3 -- a global batch of --global-batch images split over the ranks")
    ap.add_argument("--global-batch", type=int, default=64)
    ap.add_argument("--sustained-steps", type=int, default=1000,
                    help="length of the sustained leg behind the timed region (0: none); reported, never the headline")
    ap.add_argument("--cpu-sample-images", type=int, default=8,
                    help="images of the CPU baseline sample (default: the whole batch of the metric workload -- the 2-image sample "
                         "of rounds 1-2 read 1.8x low, profiles/r03_cpu_baseline_8_images.json)")
    ap.add_argument("--plumbing-check", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-sample-points", type=int, default=50000,
                    help="points per image of the CPU baseline sample (bounds the 256^3 grid workload)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started the way the driver starts the N = 1 case (`python bench.py --gpus N`, no launcher): this process
        # becomes the launcher.  Nothing has touched the GPU yet (importing torch does not), the ranks are CHILD
        # processes (never an exec), and rank 0's JSON line and the launcher's return code are relayed.
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world
    import torch.distributed as dist
    multi = world > 1 or FORCE_EXCHANGE
    if args.plumbing_check:
        # tests/test_bench_launch.py (CPU, gloo): launcher -> ranks -> rendezvous -> one collective -> rank 0's line,
        # without a GPU: the argument and environment plumbing of the N > 1 start, nothing else
        if os.environ.get("LIST_BENCH_PLUMBING_FAIL_RANK") == str(rank):
            raise SystemExit(3)                        # (the test of the launcher's exit code)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        seen = [None] * world
        dist.all_gather_object(seen, {"rank": rank, "local_rank": local_rank, "steps": args.steps, "warmup": args.warmup})
        dist.barrier()
        if rank == 0:
            print(json.dumps({"plumbing_check": True, "n_gpus": world, "ranks": seen, "scaling": args.scaling,
                              "workload": args.workload}))
        dist.destroy_process_group()
        return
    # LIST_BENCH_BACKEND=gloo lets several ranks share one GPU to rehearse the control flow of the
    # N > 1 path on a single-GPU box (RCCL refuses duplicate devices); the driver uses nccl (= RCCL).
    backend = os.environ.get("LIST_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:            # (a one-rank group started without a launcher: the pre-flight)
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import __graft_entry__ as ge
    if multi:                       # one builder per node, the others wait for the library
        if local_rank == 0:
            ge.build()
        dist.barrier()
    else:
        ge.build()
    from list_amd import hip
    from list_amd.parallel import gather_sdf_shards, shard_range
    gather_fn = gather_sdf_shards

    b_local = None
    if args.scaling == "strong":
        # BASELINE config 3: B_global images split over the ranks (contiguous, balanced; the first
        # B_global % world ranks own one more)
        if args.workload != "list_im2sdf_b8_n20k_224":
            raise SystemExit("--scaling strong is BASELINE config 3: the list_im2sdf_b8_n20k_224 shapes")
        if args.global_batch < world:
            raise SystemExit(f"--global-batch {args.global_batch} cannot be split over {world} ranks")
        b0, b1 = shard_range(args.global_batch, rank, world)
        b_local = b1 - b0
    inp = make_inputs(args.workload, rank, device, batch=b_local)
    if args.scaling == "strong":
        inp["B_pad"] = -(-args.global_batch // world)
    B, N = inp["B"], inp["N"]
    global_points = args.global_batch * N if args.scaling == "strong" else world * B * N
    _, _, img_res, vox_res, map_size, _ = WORKLOADS[args.workload]
    ev = HipEvents()
    # config 4 (256^3 inference grid) is stated in fp32: its headline is the fp32-grade bf16x3 arithmetic, fp16 the alt
    grid = args.workload.startswith("list_grid")
    headline = args.precision or ("bf16x3" if grid else "fp16")
    alt_prec = "fp16" if headline == "bf16x3" else "bf16x3"
    elapsed, kernel_ms, sdf, extra = run_config(args, headline, args.steps, args.warmup, inp, hip, ev, world,
                                                device, gather_fn, sustained_steps=args.sustained_steps)
    alt_cl = None
    if args.precision is None and not args.no_channels_last_alt:
        # same workload with the maps already in the layout MI355X-first producers emit (SURVEY 8 f2:
        # channels_last / channels_last_3d encoders, as network.models.LIST runs them): the 3-D
        # hand-off becomes zero-copy, the 2-D resize reads NHWC
        inp_cl = dict(inp)
        inp_cl["img_maps"] = [m.contiguous(memory_format=torch.channels_last) for m in inp["img_maps"]]
        inp_cl["vox_maps"] = [m.contiguous(memory_format=torch.channels_last_3d) for m in inp["vox_maps"]]
        c_steps = max(2, args.steps // 2)
        c_el, c_ms, c_sdf, _ = run_config(args, headline, c_steps, min(args.warmup, 2), inp_cl, hip, ev, world,
                                          device, gather_fn)
        alt_cl = {"inputs": "channels_last / channels_last_3d fp32 maps resident in HBM", "precision": headline,
                  "value": global_points * c_steps / c_el, "steps": c_steps, "ms_per_step": c_el / c_steps * 1e3,
                  "kernel_ms": c_ms, "max_abs_diff_vs_headline": float((c_sdf - sdf).abs().max())}
        # the headline keeps the reference's NCHW / NCDHW hand-off inside its timed step (the boundary the reference's
        # encoders emit); the package's own LIST runs channels-last encoders, for which this entry is the query rate
        alt_cl["ratio_to_headline"] = alt_cl["value"] / (global_points * args.steps / elapsed)
        alt_cl["note"] = ("the layout network.models.LIST of this package hands over (channels_last encoders): the "
                          "reference-layout headline pays the NCDHW->NDHWC transposes in every step and under-states "
                          "the in-model query rate by this ratio")
        del inp_cl
    alt = None
    if args.precision is None:
        a_steps = max(2, args.steps // 2)
        a_el, a_ms, a_sdf, a_extra = run_config(args, alt_prec, a_steps, min(args.warmup, 2), inp, hip, ev, world,
                                                device, gather_fn)
        alt = {"precision": alt_prec, "value": global_points * a_steps / a_el, "steps": a_steps,
               "ms_per_step": a_el / a_steps * 1e3, "kernel_ms": a_ms,
               "max_abs_diff_vs_headline": float((a_sdf - sdf).abs().max())}

    alt_unfused = None
    if args.precision is None and kernel_ms.get("_fused_fc0"):
        # the same step with fc_0 as the plain GEMM it was until round 3 (k_gather_img writes the perceptual block into X,
        # k_gemm_nt_pp reads it back): the MFMA roofline of fc_0 ALONE, beside the fused kernel's (whose time also holds
        # the bilinear sampling of 1024 channels per point)
        u_steps = max(2, args.steps // 2)
        u_el, u_ms, u_sdf, _ = run_config(args, headline, u_steps, min(args.warmup, 2), inp, hip, ev, world, device,
                                          gather_fn, fused_fc0=False)
        alt_unfused = {"what": "ListQueryArgs.no_fused_fc0 = 1: k_gather_img + k_gemm_nt_pp instead of k_fc0_fused",
                       "value": global_points * u_steps / u_el, "ms_per_step": u_el / u_steps * 1e3,
                       "fc_0_ms": u_ms["fc_0"], "gather_img_ms": u_ms["gather_img"], "gathers_ms": u_ms["gathers_back_to_back"],
                       "fc_0_frac_of_mfma_peak": P_FLOP_FC0(B * N) / (u_ms["fc_0"] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
                       "bit_identical_to_headline": bool(torch.equal(u_sdf, sdf))}
        assert alt_unfused["bit_identical_to_headline"]
    alt_bf16 = None
    if args.precision is None and "bf16" not in (headline, alt_prec):
        # BASELINE config 2 says "bf16": the plain-bf16 mode (one MFMA product per MAC on bf16 operands, fp32 maps) as
        # a third first-class mode of the line, with its stated tolerance (5e-3 absolute) and both parity legs below
        b_steps = max(2, args.steps // 2)
        b_el, b_ms, b_sdf, _ = run_config(args, "bf16", b_steps, min(args.warmup, 2), inp, hip, ev, world, device, gather_fn)
        alt_bf16 = {"precision": "bf16", "value": global_points * b_steps / b_el, "steps": b_steps,
                    "ms_per_step": b_el / b_steps * 1e3, "fc_0_ms": b_ms["fc_0"],
                    "roofline_frac": P_FLOP_FC0(B * N) / (b_ms["fc_0"] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
                    "max_abs_diff_vs_headline": float((b_sdf - sdf).abs().max())}

    alt_img_proj = None
    if args.precision is None and not grid and alt is not None:
        # list_prep_img_proj (ABI 8) on / off beside what the modes above ran: the fp16 headline with it forced on (the
        # 2-D sample then leaves fc_0 again: k_gather_img + k_gemm_nt_pp with a row-vector epilogue), the bf16x3 mode
        # with it forced off (the round-3 path)
        p_steps = max(2, args.steps // 2)
        on_el, on_ms, on_sdf, _ = run_config(args, headline, p_steps, min(args.warmup, 2), inp, hip, ev, world, device,
                                             gather_fn, img_proj=not kernel_ms.get("_img_proj"))
        off_el, off_ms, off_sdf, _ = run_config(args, alt_prec, p_steps, min(args.warmup, 2), inp, hip, ev, world, device,
                                                gather_fn, img_proj=not alt["kernel_ms"].get("_img_proj"))
        alt_img_proj = {
            "what": "the same two modes with list_prep_img_proj switched the OTHER way (it is on by default for every "
                    "precision, hip.img_proj_default): the low-resolution encoder levels go through their fc_0 columns "
                    "before the resize",
            headline: {"img_proj": bool(on_ms.get("_img_proj")), "ms_per_step": on_el / p_steps * 1e3, "fc_0_ms": on_ms["fc_0"],
                       "prep_img_ms": on_ms["prep_img_resize_nhwc"], "gathers_ms": on_ms["gathers_back_to_back"],
                       "max_abs_diff_vs_mode": float((on_sdf - sdf).abs().max())},
            alt_prec: {"img_proj": bool(off_ms.get("_img_proj")), "ms_per_step": off_el / p_steps * 1e3, "fc_0_ms": off_ms["fc_0"],
                       "prep_img_ms": off_ms["prep_img_resize_nhwc"], "gathers_ms": off_ms["gathers_back_to_back"],
                       "max_abs_diff_vs_mode": float((off_sdf - a_sdf).abs().max())}}

    train = None
    train_grads = None
    if args.precision is None and not args.no_train_step and B * N <= 262144:
        train, train_grads = run_train_step(headline, max(2, args.steps // 2), min(args.warmup, 2), inp, hip, ev)
        # the same training step with fp32-grade gradients (bf16 hi+lo operands): the parity-grade number
        t3, _ = run_train_step("bf16x3", max(2, args.steps // 4), min(args.warmup, 2), inp, hip, ev)
        train["fp32_grade"] = {k: t3[k] for k in ("precision", "steps", "ms_per_step", "value", "forward_query_ms",
                                                  "backward_ms", "kernel_ms", "kernel_ms_inline", "roofline")}
        # the same step with projections piled onto the clamp of network/modules.py:43, the regime training STARTS in (an
        # untrained spatial transformer: 88 % of the points, --whole-model): u, v spread 8x wider around the map centre
        inp_p = dict(inp)
        Tp = inp["trans_mat"].clone()
        Tp[:, :3, :2] *= 8.0
        inp_p["trans_mat"] = Tp
        tp, _ = run_train_step(headline, max(2, args.steps // 4), min(args.warmup, 2), inp_p, hip, ev)
        train["piled_on_clamp"] = {"frac_points_on_clamp": float(_on_clamp(inp["query"], Tp)), "ms_per_step": tp["ms_per_step"],
                                   "backward_ms": tp["backward_ms"], "img_map_grad_ms_inline": tp["kernel_ms_inline"]["img_map_grad"],
                                   "note": "map-side gather of the perceptual-map gradient with its heavy pixel groups cut "
                                           "into chunks (round 4); 8.7 ms / 2.58 ms before"}
        train["gradient_precision_note"] = (
            "fp16 operands flip ~1e-3 of the ReLU masks: gradients carry 2-4 % relative L2 noise against the "
            "reference's autograd (grad_rel_l2_vs_cpu); the bf16x3 step reproduces them to 1.5e-5 of each tensor's "
            "largest entry (tests/test_hip_backward.py)")

    whole = None
    if args.whole_model and rank == 0:
        whole = run_whole_model(args.workload, headline, device)

    ranks = None
    if multi:
        # what the collective backend really spans: every rank's id and timed-region wall clock through an all-gather
        # round trip (nccl = RCCL on device tensors; the gloo rehearsal stages through the host)
        on = device if dist.get_backend() == "nccl" else "cpu"
        mine = torch.tensor([float(rank), extra["elapsed_local_s"], float(B)], dtype=torch.float64, device=on)
        allr = torch.empty((world * 3,), dtype=torch.float64, device=on)
        dist.all_gather_into_tensor(allr, mine)
        allr = allr.cpu().reshape(world, 3)
        ms = allr[:, 1] / args.steps * 1e3
        ranks = {"backend": dist.get_backend(), "rccl_ranks_observed": int(len(set(int(x) for x in allr[:, 0].tolist()))),
                 "world_size": dist.get_world_size(), "images_per_rank": [int(x) for x in allr[:, 2].tolist()],
                 "ms_per_step_per_rank": {"min": float(ms.min()), "max": float(ms.max()),
                                          "all": [round(float(x), 4) for x in ms.tolist()]},
                 "exchange_check": extra.get("exchange_check")}

    if rank != 0:
        if multi:
            dist.destroy_process_group()
        return

    P = B * N
    value = global_points * args.steps / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    xb = 2 if headline != "bf16x3" else 4
    table = kernel_table(B, N, img_res, vox_res, map_size, xb, 2 if headline == "fp16" else 4, proj=bool(inp.get("ordered_points")),
                         fused_tail=bool(kernel_ms.get("_fused_tail")))
    roof = roofline_of(kernel_ms, table, headline, args.workload, B)
    if alt is not None:
        a16 = alt["precision"] == "fp16"
        alt["roofline"] = roofline_of(alt["kernel_ms"], kernel_table(B, N, img_res, vox_res, map_size, 2 if a16 else 4,
                                                                      2 if a16 else 4, proj=bool(inp.get("ordered_points")),
                                                                      fused_tail=bool(alt["kernel_ms"].get("_fused_tail"))),
                                      alt["precision"], args.workload, B)
        alt["path_roofs"] = path_roofs(alt["value"] * P / global_points, alt["precision"], grid=bool(inp.get("ordered_points")))
    gather_ms = kernel_ms["gathers_back_to_back"]       # the seven launches as the timed region runs them
    mlp_ms = kernel_ms["fc_0"] + kernel_ms["fc_1"] + kernel_ms["fc_2_out"]
    pmc = pmc_traffic(headline, args.workload, B)
    path = {
        # the whole path against its binding roof (per GPU)
        **path_roofs(value * P / global_points, headline, grid=bool(inp.get("ordered_points"))),      # rank 0's share
        "mlp_TFLOPs_algorithmic": P * W_FLOP_PER_PT / (mlp_ms * 1e-3) / 1e12,
        "mlp_frac_of_bf16_peak": P * W_FLOP_PER_PT / (mlp_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
        "gather_ms": gather_ms, "mlp_ms": mlp_ms,
    }
    if pmc:
        hbm = sum(v["hbm_bytes"] for k, v in pmc.items() if k in kernel_ms)
        path["hbm_bytes_per_step_pmc"] = hbm
        path["hbm_GBps_whole_step_pmc"] = hbm / (ms_per_step * 1e-3) / 1e9
        path["hbm_frac_of_peak_whole_step_pmc"] = hbm / (ms_per_step * 1e-3) / 1e9 / PEAK_HBM_GBS
    per_kernel = {}
    for k, (b, u, what) in table.items():
        if kernel_ms.get(k, 0) <= 0:
            continue
        e = {"ms": kernel_ms[k], "bound": b, "units": what}
        if k.startswith("gather_"):
            e["timed_apart"] = True        # from the untimed detail steps (barriers between the gathers), not the timed region
        if k == "fc_2_out" and kernel_ms.get("_fused_tail"):
            e["fused"] = "fc_1 + fc_2 + fc_out in one launch (k_mlp_tail_f16)"
        if b == "mfma":
            e["TFLOPs"] = u / (kernel_ms[k] * 1e-3) / 1e12
            e["frac_of_mfma_peak"] = e["TFLOPs"] / PEAK_BF16_TFLOPS
        else:
            e["GBps"] = u / (kernel_ms[k] * 1e-3) / 1e9
            if b == "hbm":
                e["frac_of_hbm_peak"] = e["GBps"] / PEAK_HBM_GBS
        if k in pmc:                       # measured HBM bytes (2 x FETCH_SIZE + WRITE_SIZE) of the same kernel
            e["hbm_bytes_pmc"] = pmc[k]["hbm_bytes"]
            e["hbm_GBps_pmc"] = pmc[k]["hbm_bytes"] / (kernel_ms[k] * 1e-3) / 1e9
            e["hbm_frac_of_peak_pmc"] = e["hbm_GBps_pmc"] / PEAK_HBM_GBS
        per_kernel[k] = e

    # ---- CPU baseline (oracle, torch-op restatement) on a bounded sample ---------------------------
    cpu = None
    parity = None
    parity_leg = None
    modes = {}
    if not args.no_cpu_baseline and not multi:
        from oracle import torch_ops as TO
        ns = min(args.cpu_sample_images, B)
        cq = inp["query"][:ns].cpu()
        ci = [m[:ns].cpu() for m in inp["img_maps"]]
        cv = [m[:ns].cpu() for m in inp["vox_maps"]]
        ct = inp["trans_mat"][:ns].cpu()
        cw = {k: v.cpu() for k, v in inp["weights"].items()}
        if True:
            cores = torch.get_num_threads()
            if cq.shape[1] > args.cpu_sample_points:             # huge queries (the 256^3 grid): a bounded slice
                cq = cq[:, :args.cpu_sample_points].contiguous()
            kw = dict(map_size=inp["map_size"], clamp_hi=inp["clamp_hi"])
            TO.list_query(cq, ci, cv, ct, cw, **kw)                 # warm-up
            times = []
            for _ in range(3):
                c0 = time.perf_counter()
                ref = TO.list_query(cq, ci, cv, ct, cw, **kw)
                times.append(time.perf_counter() - c0)
            med = sorted(times)[1]
            cpu = {"value": ns * cq.shape[1] / med, "unit": "query-points/s", "cores": cores, "kind": "port",
                   "sample": f"{ns} of {B} images x {cq.shape[1]} of {N} points, oracle/torch_ops.py (the reference's torch "
                             f"op sequence, fp32, no_grad), median of 3 after 1 warm-up, torch "
                             f"{torch.__version__}, os.cpu_count()={os.cpu_count()}"}
            # ---- parity, magnitude-aware (the fp16 mode's error is RELATIVE, ~7e-4 of max|sdf|; bf16x3's is fp32-grade) --
            # leg 1: the benched distribution (|sdf| <~ 0.06 with the SURVEY 8d weights); leg 2: every weight and bias of
            # the MLP times 1.74, which takes |sdf| to ~0.5 -- what a trained LIST emits on a unit box (sdf_scale 1.0,
            # arguments.py:54).  Stated tolerances: fp16 1e-3 x max|sdf|, bf16x3 1e-4 absolute; both asserted here.
            md_of = hip.map_dtype_for
            def forward(prec, weights):
                # the path the mode is TIMED on (run_config): the bf16 formats project the low-resolution encoder levels
                # through fc_0 before the resize (list_prep_img_proj), fp16 samples all 1024 channels inside fc_0
                vox_p = hip.prep_vox_maps(inp["vox_maps"], md_of(prec))
                pk = hip.prep_mlp_weights(weights, vox_p.channels, sum(m.shape[1] for m in inp["img_maps"]), prec)
                if hip.img_proj_default(prec) and not inp.get("ordered_points") and \
                        hip.img_proj_kept_levels(inp["img_maps"], inp["map_size"]) < hip.N_IMG_LEVELS:
                    img_p = hip.prep_img_proj(inp["img_maps"], pk, inp["map_size"], prec)
                else:
                    img_p = hip.prep_img_maps(inp["img_maps"], inp["map_size"], md_of(prec))
                pj = hip.prep_percep_proj(img_p, pk, prec) if inp.get("ordered_points") else None
                o = hip.sdf_query(inp["query"], inp["trans_mat"], img_p, vox_p, pk, precision=prec,
                                  clamp_hi=inp["clamp_hi"], sort_points=not inp.get("ordered_points", False), percep_proj=pj)
                torch.cuda.synchronize()
                return o[:ns, :cq.shape[1]].cpu()
            def leg(got, want):
                e, m = float((got - want).abs().max()), float(want.abs().max())
                return {"max_abs_err": e, "max_abs_sdf": m, "rel_err": e / m, "holds_1e-4_absolute": e < 1e-4}
            big_w = {k: v * 1.74 for k, v in inp["weights"].items()}
            ref_big = TO.list_query(cq, ci, cv, ct, {k: v.cpu() for k, v in big_w.items()}, **kw)
            tolerance = TOLERANCE
            modes = {}
            for prec in dict.fromkeys([headline, alt_prec, "bf16"] if args.precision is None else [headline]):
                legs = {"bench_distribution": leg(forward(prec, inp["weights"]), ref),
                        "sdf_about_0.5": leg(forward(prec, big_w), ref_big)}
                tol = tolerance[prec]
                for name, l in legs.items():
                    l["within_stated_tolerance"] = (l["rel_err"] < tol["bound"]) if tol["kind"] == "relative" \
                        else (l["max_abs_err"] < tol["bound"])
                    assert l["within_stated_tolerance"], (prec, name, l)
                modes[prec] = {"stated_tolerance": tol, "parity_vs_cpu": legs}
            parity_leg = leg(sdf[:ns, :cq.shape[1]].cpu(), ref)
            parity = parity_leg["max_abs_err"]
            if train is not None and inp["map_size"] == 137:
                # the same training step through torch autograd on the host (1 image), and the agreement of
                # two gradients that depend on no ReLU mask / on every mask
                g1 = (torch.randn((B, N), generator=torch.Generator(device=device).manual_seed(4242),
                                  device=device) / B)[:1].cpu()
                TO.list_query_grads(cq[:1], [m[:1] for m in ci], [m[:1] for m in cv], ct[:1], cw, g1)
                c0 = time.perf_counter()
                _, cg = TO.list_query_grads(cq[:1], [m[:1] for m in ci], [m[:1] for m in cv], ct[:1], cw, g1)
                dt = time.perf_counter() - c0
                got_T = train_grads["trans_mat"][:1].cpu()
                got_v5 = train_grads["vox"][5][:1].permute(0, 4, 1, 2, 3).cpu()
                train["cpu_baseline"] = {
                    "value": N / dt, "unit": "query-points/s (forward + backward)", "cores": cores, "kind": "port",
                    "sample": f"1 of {B} images x {N} points, torch autograd over oracle/torch_ops.py, 1 run after "
                              "1 warm-up"}
                train["grad_rel_l2_vs_cpu"] = {
                    "d_trans_mat[0]": float((got_T - cg["d_trans_mat"]).norm() / cg["d_trans_mat"].norm()),
                    "d_vox5[0]": float((got_v5 - cg["d_vox5"]).norm() / cg["d_vox5"].norm())}

    perf = {headline: {"value": value, "ms_per_step": ms_per_step, "roofline_frac": roof["frac"]}}
    if alt is not None:
        perf[alt["precision"]] = {"value": alt["value"], "ms_per_step": alt["ms_per_step"],
                                  "roofline_frac": alt["roofline"]["frac"]}
    if alt_bf16 is not None:
        perf["bf16"] = {"value": alt_bf16["value"], "ms_per_step": alt_bf16["ms_per_step"],
                        "roofline_frac": alt_bf16["roofline_frac"]}
    for prec, pf in perf.items():
        modes.setdefault(prec, {"stated_tolerance": TOLERANCE[prec], "parity_vs_cpu": None}).update(pf)
    arith = {"bf16x3": "bf16 hi/lo split operands, 3 MFMA products per MAC, fp32 accumulate",
             "fp16": "fp16 operands (saturating), 1 MFMA product per MAC, fp32 accumulate",
             "bf16": "bf16 operands, 1 MFMA product per MAC, fp32 accumulate"}
    out = {
        "metric": "SDF query-points/sec (B=8, N=20k, 224^2)" if args.workload.endswith("n20k_224")
                  else "SDF query-points/sec",
        "value": value, "unit": "query-points/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": args.scaling, "vs_baseline": None,
        "dtype": {"fp16": "fp16", "bf16": "bf16", "bf16x3": "bf16x3 (bf16 hi+lo operands, fp32-grade)"}[headline],
        "data": "synthetic",
        "config": {"workload": args.workload if args.scaling == "weak" else f"list_im2sdf_b{args.global_batch}_n20k_224 (BASELINE config 3)",
                   "images_per_gpu": B, "points_per_image": N,
                   "global_batch": args.global_batch if args.scaling == "strong" else world * B,
                   "global_points_per_step": global_points, "precision": headline,
                   "point_sort": ("skipped: raster-ordered grid, as executors.LIST.predict_grid queries it"
                                  if inp.get("ordered_points") else "Morton + pixel counting sort inside the step"),
                   "perceptual_block": ("projected through fc_0 once per image (list_prep_percep_proj, inside the step), "
                                        "sampled per point; fc_0 runs the other 2586 columns"
                                        if inp.get("ordered_points") else "1024 sampled features per point in fc_0's K"),
                   "mlp_arithmetic": arith[headline],
                   "gather_arithmetic": "fp32 interpolation; prepared maps stored as "
                                        + ("fp16" if hip.map_dtype_for(headline) == "f16" else "fp32"),
                   "inputs": "reference layout (NCHW/NCDHW fp32) resident in HBM; layout hand-off + weight "
                             "repack inside the timed step",
                   # names the collective that RAN: RCCL on the "nccl" backend, gloo in the one-GPU rehearsals
                   "parallelism": f"batch-sharded x{world}" + (
                       (" + RCCL all-gather of sdf" if dist.get_backend() == "nccl" else
                        f" + {dist.get_backend()} all-gather of sdf (rehearsal, not RCCL)") if multi else "")},
        "ranks": ranks,
        "step_events_ms": extra["step_events_ms"],
        "sustained": extra.get("sustained"),
        "roofline": roof,
        "cpu_baseline": cpu,
        "kernel_ms": kernel_ms,
        "kernel_ms_note": "HIP-event intervals over the timed steps, except gather_vox_l1 .. gather_tail: the seven "
                          "gathers of a step are independent launches and run without queue barriers between them in "
                          "the timed region (gathers_back_to_back is their group, timed there) and with the barriers "
                          f"in {DETAIL_STEPS} further untimed steps, which is where their individual durations come from",
        "per_kernel": per_kernel,
        "path_rates": path,
        "parity_max_abs_err_vs_cpu": parity,
        "parity_max_abs_sdf": parity_leg["max_abs_sdf"] if parity_leg else None,
        "parity_rel_err": parity_leg["rel_err"] if parity_leg else None,
        "parity_bound": ({"fp16": "1e-3 x max|sdf| (relative: the mode rounds features and activations to 11 bits; 1e-4 "
                                  "absolute holds only while |sdf| <~ 0.12 -- see modes.fp16.parity_vs_cpu)",
                          "bf16x3": "1e-4 absolute (fp32-grade at every magnitude)", "bf16": "5e-3 absolute"}[headline]),
        "modes": modes,
        "speedup_vs_cpu": (value / cpu["value"]) if cpu else None,
        "alt": alt,
        "alt_bf16": alt_bf16,
        "alt_unfused_fc0": alt_unfused,
        "alt_img_proj": alt_img_proj,
        "alt_channels_last_inputs": alt_cl,
        "train_step": train,
        "whole_model": whole,
    }
    # the LAST object of the line: what a reviewer needs, short enough for a 2 000-character tail of the output
    # (Mpts = million query-points/s; err = max|sdf - CPU reference| on the benched distribution / at |sdf| ~ 0.5)
    def _mode(pr):
        m = modes.get(pr)
        if not m or "value" not in m:
            return None
        pv = m.get("parity_vs_cpu")
        errs = [float(f"{pv[k]['max_abs_err']:.1e}") for k in ("bench_distribution", "sdf_about_0.5")] if pv else None
        return {"Mpts": round(m["value"] / 1e6, 2), "ms": round(m["ms_per_step"], 3), "err": errs}
    if alt is not None:
        modes[alt["precision"]]["img_proj"] = bool(alt["kernel_ms"].get("_img_proj"))
    modes[headline]["img_proj"] = bool(kernel_ms.get("_img_proj"))
    out["summary"] = {
        "fp16": _mode("fp16"), "bf16x3": _mode("bf16x3"), "bf16": _mode("bf16"),
        "channels_last_Mpts": round(alt_cl["value"] / 1e6, 2) if alt_cl else None,
        "train_ms": [round(train["ms_per_step"], 2), round(train["fp32_grade"]["ms_per_step"], 2)] if train else None,
        "bwd_ms": [round(train["backward_ms"], 2), round(train["fp32_grade"]["backward_ms"], 2)] if train else None,
        "fc0_ms": round(kernel_ms["fc_0"], 4), "fc0_frac": round(roof["frac"], 3),
        "unfused": [round(alt_unfused["ms_per_step"], 3), round(alt_unfused["fc_0_ms"], 4),
                    round(alt_unfused["fc_0_frac_of_mfma_peak"], 3)] if alt_unfused else None,
        # ms per step with list_prep_img_proj [fp16 forced on, bf16x3 forced off] beside the modes' own lines above
        "img_proj_ms": [round(alt_img_proj[headline]["ms_per_step"], 3), round(alt_img_proj[alt_prec]["ms_per_step"], 3)]
                       if alt_img_proj else None,
        "gathers_ms": round(gather_ms, 4), "prep_ms": round(kernel_ms["prep_img_resize_nhwc"] + kernel_ms["prep_vox_ndhwc"], 4),
        "path_frac": round(path["whole_path_frac_of_binding_roof"], 3),
        "hbm_GB_step": round(path["hbm_bytes_per_step_pmc"] / 1e9, 2) if "hbm_bytes_per_step_pmc" in path else None,
        "cpu_kpts": round(cpu["value"] / 1e3, 1) if cpu else None, "cpu_cores": cpu["cores"] if cpu else None,
    }
    assert len(json.dumps(out["summary"])) <= 700, len(json.dumps(out["summary"]))
    print(json.dumps(out))
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
