"""Exact, platform-independent synthetic inputs of the path's shapes: feature-map shapes, the
synthetic camera, Conv1d(k=1)-shaped MLP weights, query points.

Every value is a pure integer hash of (seed, flat index) mapped to a dyadic
rational, so the authoring container, the GPU box and any numpy version produce
the same bits.  No libm call is involved (no Box-Muller): the "normal-ish"
variate is a centred sum of four uniforms, exact in float64.

Lives in the package so that the measurement harness (bench.py's GPU leg, tools/) needs nothing from oracle/;
oracle/synth.py re-exports it for the golden generator and the parity tests, which feed the reference, the
oracle and the HIP path identical inputs.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(z):
    # splitmix64 finaliser on uint64 arrays (wrap-around arithmetic is intended)
    z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def _bits(seed, n, stream=0, chunk=1 << 22):
    out = np.empty(n, dtype=np.uint64)
    base = np.uint64((int(seed) * 0xD1342543DE82EF95 + int(stream) * 0xA0761D6478BD642F)
                     & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        for s in range(0, n, chunk):
            e = min(n, s + chunk)
            idx = np.arange(s, e, dtype=np.uint64)
            out[s:e] = _mix(_mix(idx ^ base) + base)
    return out


def uniform(seed, shape, lo=0.0, hi=1.0):
    """float32 U[lo,hi) on a 2^-24 lattice."""
    n = int(np.prod(shape))
    u = (_bits(seed, n) >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24))
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def normalish(seed, shape, scale=1.0):
    """float32, mean 0, variance scale^2, bell shaped (sum of 4 uniforms)."""
    n = int(np.prod(shape))
    b = _bits(seed, n)
    m16 = np.uint64(0xFFFF)
    s = ((b & m16) + ((b >> np.uint64(16)) & m16) + ((b >> np.uint64(32)) & m16)
         + ((b >> np.uint64(48)) & m16)).astype(np.float64)
    # each term U{0..65535}/65536: mean ~0.5, var 1/12; sum of 4: var 1/3
    z = (s / 65536.0 - 2.0 + 4 * 0.5 / 65536.0) * np.sqrt(3.0)
    return (scale * z).astype(np.float32).reshape(shape)


# ---------------------------------------------------------------------------
# Workload builders (shapes of SURVEY.md section 8a / 8d)
# ---------------------------------------------------------------------------
IMG_CHANNELS = (64, 64, 128, 256, 512)     # ResEncoder maps  (modules.py:1067)
VOX_CHANNELS = (1, 16, 32, 64, 128, 128)   # VoxelEncoder2 maps (modules.py:425-442)


def img_map_shapes(batch, img_res):
    return [(batch, c, max(img_res >> s, 1), max(img_res >> s, 1))
            for c, s in zip(IMG_CHANNELS, range(5))]


def vox_map_shapes(batch, vox_res):
    res = [vox_res, vox_res, vox_res // 2, vox_res // 4, vox_res // 8, vox_res // 16]
    return [(batch, c, max(r, 1), max(r, 1), max(r, 1)) for c, r in zip(VOX_CHANNELS, res)]


def make_img_maps(seed, batch, img_res):
    return [normalish(seed + 11 * i, s) for i, s in enumerate(img_map_shapes(batch, img_res))]


def make_vox_maps(seed, batch, vox_res):
    shapes = vox_map_shapes(batch, vox_res)
    maps = [uniform(seed + 1000, shapes[0])]                      # sigmoid output
    maps += [normalish(seed + 1000 + 13 * i, s) for i, s in enumerate(shapes[1:], 1)]
    return maps


def make_query(seed, batch, n):
    """U(-0.5,0.5)^3, the range of Datasets.py:229."""
    return uniform(seed + 2000, (batch, n, 3), -0.5, 0.5)


def make_trans_mat(seed, batch, jitter=0.05):
    """Synthetic camera of SURVEY 8d: u,v ~ 68 +- 60, Z ~ 1."""
    base = np.array([[60, 0, 0], [0, 60, 0], [0, 0, 0], [68, 68, 1]], dtype=np.float32)
    return (base[None] + normalish(seed + 3000, (batch, 4, 3), jitter)).astype(np.float32)


def make_mlp_weights(seed, feature_size=3610, h_dim=256):
    """Conv1d(k=1)-shaped weights with PyTorch-default-like scale U(+-1/sqrt(fan_in)).

    Keys follow the reference state_dict (modules.py:196-200)."""
    dims = [("fc_0", feature_size, 2 * h_dim), ("fc_1", 2 * h_dim, h_dim),
            ("fc_2", h_dim, h_dim), ("fc_out", h_dim, 1)]
    w = {}
    for i, (name, fin, fout) in enumerate(dims):
        bound = 1.0 / np.sqrt(fin)
        w[name + ".weight"] = uniform(seed + 4000 + 2 * i, (fout, fin, 1), -bound, bound)
        w[name + ".bias"] = uniform(seed + 4001 + 2 * i, (fout,), -bound, bound)
    return w
