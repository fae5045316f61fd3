"""Host utilities that keep the reference's names (utils.py of the reference): dotted-name class
lookup, checkpoint format, grid builders, mesh helpers.  Optional third-party modules (mcubes,
trimesh, scipy KD-tree) are imported lazily so the query path has no such dependency."""
import importlib
import os

import numpy as np
import torch

_PKG = __name__.rsplit(".", 1)[0] if "." in __name__ else None


def print_log(log_fname, logline):
    with open(log_fname, "a") as f:
        f.write(logline + "\n")


def get_class(kls):
    """'network.models.LIST' -> class (reference utils.py:20-26).  Names are resolved inside this
    package first, so the reference's command lines keep working."""
    module, name = kls.rsplit(".", 1)
    last = None
    for cand in ([f"{_PKG}.{module}"] if _PKG else []) + [module]:
        try:
            return getattr(importlib.import_module(cand), name)
        except (ImportError, AttributeError) as e:
            last = e
    raise ImportError(f"cannot resolve {kls}: {last}")


def _unwrap(model):
    return model.module if hasattr(model, "module") else model


def save_checkpoint(epoch, model, optimizer, bestloss, output_filename):
    """Reference format (utils.py:29-34): epoch+1, state_dict of the unwrapped model, optimizer, bestloss."""
    torch.save({"epoch": epoch + 1, "state_dict": _unwrap(model).state_dict(),
                "optimizer": optimizer.state_dict(), "bestloss": bestloss}, output_filename)


def load_checkpoint(cp_filename, model, optimizer=None):
    ck = torch.load(cp_filename, map_location="cpu")
    _unwrap(model).load_state_dict(ck["state_dict"])
    if optimizer is not None:
        optimizer.load_state_dict(ck["optimizer"])
    return ck["epoch"], model, optimizer, ck.get("bestloss", 10000000)


def load_model(cp_filename, model):
    ck = torch.load(cp_filename, map_location="cpu")
    model.load_state_dict(ck["state_dict"])
    return ck["epoch"], model


def switch_grad(model, value):
    for p in model.parameters():
        p.requires_grad = value
    return model


def create_grid_points_from_bounds(minimum, maximum, res, verbose=False):
    """[res^3,3] float64, 'ij' meshgrid with X slowest (reference utils.py:84-95)."""
    if verbose:
        print(f"Generating grid points with bounds {minimum}~{maximum} and res {res}")
    axis = np.linspace(minimum, maximum, res)
    gx, gy, gz = np.meshgrid(axis, axis, axis, indexing="ij")
    return np.stack((gx.ravel(), gy.ravel(), gz.ravel()), axis=1)


def grid_points_on_device(minimum, maximum, res, device, begin=0, end=None):
    """The same grid generated on the device for points [begin, end) -- no host->device copy of
    res^3 x 3 floats.  Values equal np.linspace(...).astype(float32) exactly (computed in float64)."""
    total = res ** 3
    end = total if end is None else end
    idx = torch.arange(begin, end, device=device, dtype=torch.int64)
    step = (maximum - minimum) / (res - 1) if res > 1 else 0.0
    i, j, k = idx // (res * res), (idx // res) % res, idx % res

    def coord(t):
        v = minimum + t.to(torch.float64) * step
        v = torch.where(t == res - 1, torch.full_like(v, float(maximum)), v)   # linspace pins the end
        return v.to(torch.float32)

    return torch.stack((coord(i), coord(j), coord(k)), dim=1)


def get_kdtree(bb_min, bb_max, res):
    from scipy.spatial import cKDTree
    return cKDTree(create_grid_points_from_bounds(bb_min, bb_max, res))


def generate_mesh(pred_values, bb_min, bb_max, as_trimesh_obj=False):
    """Marching cubes of the NEGATED field at level 0 (reference utils.py:172-182)."""
    try:
        import mcubes
    except ImportError as e:
        raise RuntimeError("PyMCubes is required for mesh extraction (not needed for SDF queries)") from e
    verts, tris = mcubes.marching_cubes(-pred_values, 0)
    res = pred_values.shape[0]
    verts = verts * ((bb_max - bb_min) / (res - 1)) + bb_min
    if as_trimesh_obj:
        import trimesh
        return trimesh.Trimesh(verts, tris, process=False)
    return verts, tris


def write_obj(fname, vertices, triangles):
    with open(fname, "w") as f:
        for v in vertices:
            f.write(f"v {v[0]} {v[1]} {v[2]}\n")
        for t in triangles:
            f.write(f"f {t[0] + 1} {t[1] + 1} {t[2] + 1}\n")


def ensure_dir(path):
    os.makedirs(path, exist_ok=True)
    return path
