"""Field transforms applied before / after the network (NumPy, host side).

Same call signature ``f(x, field, z, stats)`` and the same mode names as the reference
(/root/reference/baryon_painter/utils/data_transforms.py:44-119); the split-scale (Gaussian
pyramid) transform of data_transforms.py:14-42 is not part of the CVAE hot path and is omitted.
``stats[field][z] -> {"mean", "var"}`` is tabulated per training redshift and linearly
interpolated in z (clamped at both ends), data_transforms.py:52-64.
"""
import numpy as np


class _Chain:
    def __init__(self, steps):
        self.steps = list(steps)

    def __call__(self, x, field, z, stats):
        for t in self.steps:
            x = t(x, field, z, stats)
        return x


def chain_transformations(transformations):
    """Apply the given transforms one after the other (data_transforms.py:44-49)."""
    return _Chain(transformations)


def interpolate_z(stats_of_field, z):
    """Statistics at redshift z: linear between the bracketing tabulated redshifts,
    the last entry at/after the last tabulated z, the first entry below the first."""
    zs = list(stats_of_field.keys())
    idx = int(np.searchsorted(zs, z, side="right"))
    if idx >= len(zs):
        return stats_of_field[zs[-1]]
    if idx <= 0:
        return stats_of_field[zs[0]]
    lo, hi = zs[idx - 1], zs[idx]
    w = (z - lo) / (hi - lo)
    return {k: w * stats_of_field[hi][k] + (1 - w) * stats_of_field[lo][k] for k in stats_of_field[zs[0]]}


def interpolate_z_many(stats_of_field, zs, key="var"):
    """``interpolate_z(stats_of_field, z)[key]`` for an array of redshifts at once -- the same float64 arithmetic
    element by element (w * hi + (1 - w) * lo, clamped at both ends), without a Python call per tile
    (paint_stream: 100k tiles)."""
    tab = np.array(list(stats_of_field.keys()), dtype=np.float64)
    val = np.array([stats_of_field[z][key] for z in stats_of_field.keys()], dtype=np.float64)
    zs = np.asarray(zs, dtype=np.float64)
    idx = np.searchsorted(tab, zs, side="right")
    i1 = np.clip(idx, 1, len(tab) - 1) if len(tab) > 1 else np.zeros_like(idx)
    i0 = i1 - 1 if len(tab) > 1 else i1
    with np.errstate(divide="ignore", invalid="ignore"):
        w = (zs - tab[i0]) / (tab[i1] - tab[i0]) if len(tab) > 1 else np.zeros_like(zs)
        out = w * val[i1] + (1 - w) * val[i0]
    out = np.where(idx >= len(tab), val[-1], out)
    return np.where(idx <= 0, val[0], out)


# mode -> (forward, inverse); each takes (x, k, std, mean, eps).  Formulas: data_transforms.py:72-108.
_MODES = {
    "log": (lambda x, k, std, mean, eps: np.where(x > 0, np.log(x / std + eps) / k, np.log(eps) / k),
            lambda y, k, std, mean, eps: np.where(y > np.log(eps) / k, (np.exp(y * k) - eps) * std, 0)),
    "shift-log": (lambda x, k, std, mean, eps: np.log(x / std + 1) / k,
                  lambda y, k, std, mean, eps: (np.exp(y * k) - 1) * std),
    "shift-log-2p": (lambda x, k, std, mean, eps: np.log(x / std + k[0]) / k[1],
                     lambda y, k, std, mean, eps: (np.exp(y * k[1]) - k[0]) * std),
    "log-tanh": (lambda x, k, std, mean, eps: np.where(x > 0, np.tanh(np.log(x / std + eps) / k), -1),
                 lambda y, k, std, mean, eps: np.where(y > -1, (np.exp(np.arctanh(y) * k) - eps) * std, 0)),
    "x/(1+x)": (lambda x, k, std, mean, eps: x / (x + std) * k[0] - k[1],
                lambda y, k, std, mean, eps: std / (k[0] / (y + k[1]) - 1)),
    "1/x": (lambda x, k, std, mean, eps: np.where(x / (std * mean * k) > -1, 2 / (x / (std * mean * k) + 1) - 1.001, -1),
            lambda y, k, std, mean, eps: np.where(y >= -1, (2 / (y + 1.001) - 1) * std * mean * k, 0)),
}


class _RangeCompress:
    """One direction of a range-compression transform; a picklable callable ``f(x, field, z, stats)``."""

    def __init__(self, k_values, modes, eps, sqrt_of_mean, direction):
        self.k_values, self.modes, self.eps = dict(k_values), dict(modes), eps
        self.sqrt_of_mean, self.direction = sqrt_of_mean, direction

    def __call__(self, x, field, z, stats):
        mode = self.modes[field]
        if mode.lower() not in _MODES:
            raise ValueError(f"Mode '{mode}' not supported.")
        s = interpolate_z(stats[field], z)
        mean = np.sqrt(s["mean"]) if self.sqrt_of_mean else s["mean"]
        return _MODES[mode.lower()][self.direction](x, self.k_values[field], np.sqrt(s["var"]), mean, self.eps)


def create_range_compress_transforms(k_values, modes={}, eps=1e-3, sqrt_of_mean=False):
    """(transform, inverse) pair for the reference's range-compression modes
    (data_transforms.py:51-110): "log", "shift-log", "shift-log-2p", "log-tanh", "x/(1+x)", "1/x"."""
    return (_RangeCompress(k_values, modes, eps, sqrt_of_mean, 0),
            _RangeCompress(k_values, modes, eps, sqrt_of_mean, 1))


def atleast_3d(x, field, z, stats):
    return x.reshape(1, *x.shape) if x.ndim == 2 else x


def squeeze(x, field, z, stats):
    return x.squeeze()


def as_float32(x, field, z, stats):
    """Keep tiles float32: under NumPy >= 2 promotion ``float32_array / np.float64_scalar`` yields
    float64, whereas the NumPy 1.x the reference was written for kept float32."""
    return np.asarray(x, dtype=np.float32)
