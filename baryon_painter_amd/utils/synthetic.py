"""Deterministic synthetic weights and tiles shared by the golden generator, the
parity tests and bench.py.

Everything is derived from ``numpy.random.Generator(PCG64(seed)).random()``
(uniform doubles), so the GPU box regenerates bit-identical inputs without any
file.  The field statistics are the fiducial ones recovered
from ``trained_models/CVAE/fiducial/model_meta``; the
transform is the reference's ``shift-log`` with k=4
(/root/reference/baryon_painter/utils/data_transforms.py:76,
scripts/CVAE_single_scale.py:34-38).
"""
import math

import numpy as np

# redshifts of the training set (scripts/CVAE_single_scale.py:31)
REDSHIFTS = (0.0, 0.125, 0.25, 0.375, 0.5, 0.75, 1.0, 1.25, 1.5, 1.75, 2.0)

# fiducial field variances at z=0 and z=2 (linear in z in between is close enough
# for *synthetic* inputs; the real tables live in the dataset's stats).
_VAR = {"dm": (1.47251, 0.1165), "pressure": (0.1349, 3.8426e-4)}


def field_sigma(field, z):
    v0, v2 = _VAR[field]
    w = min(max(z / 2.0, 0.0), 1.0)
    return math.sqrt((1 - w) * v0 + w * v2)


def shift_log(x, sigma, k=4.0):
    """data_transforms.py:76: log(x/std + 1)/k."""
    return np.log(x / sigma + 1.0) / k


def fill_params(shapes, seed=0, bn_jitter=True):
    """name->shape (reference state_dict learnable keys, registration order) ->
    name->float32 array.  Conv/linear weights: uniform(+-1/sqrt(fan_in));
    BN gamma in [0.5,1.5), beta in [-0.2,0.2) (so that the affine part is
    exercised); PReLU slope 0.25 +- 0.1."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    for name, shape in shapes.items():
        shape = tuple(shape)
        n = int(np.prod(shape)) if shape else 1
        u = rng.random(n)
        if len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            a = 1.0 / math.sqrt(fan_in)
            # ConvTranspose weights are (in,out,kh,kw): fan computed the same way
            v = (2 * u - 1) * a * 1.7
        elif len(shape) == 1 and shape[0] == 1 and not name.endswith((".bias",)) and _is_prelu(name, shapes):
            v = 0.25 + (u - 0.5) * 0.2
        elif name.endswith("weight"):
            v = (0.5 + u) if bn_jitter else np.ones(n)
        else:
            v = ((u - 0.5) * 0.4) if bn_jitter else np.zeros(n)
        out[name] = v.reshape(shape).astype(np.float32)
    return out


def _is_prelu(name, shapes):
    # a PReLU has a (1,) weight and no sibling ".bias"
    return name.endswith(".weight") and (name[:-6] + "bias") not in shapes


def synthetic_batch(n, h, w, seed=1234, dtype=np.float32):
    """Seeded (x=pressure, y=dm, aux=redshift) batch in the transformed domain:
    raw dm ~ lognormal-ish * mean, raw pressure correlated with dm, then shift-log."""
    rng = np.random.Generator(np.random.PCG64(seed))
    zi = (rng.random(n) * len(REDSHIFTS)).astype(np.int64)
    z = np.array([REDSHIFTS[i] for i in zi], dtype=np.float64)
    # Box-Muller from uniform doubles (stable across numpy versions)
    u1 = rng.random((n, h, w))
    u2 = rng.random((n, h, w))
    g = np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2 * math.pi * u2)
    u3 = rng.random((n, h, w))
    u4 = rng.random((n, h, w))
    g2 = np.sqrt(-2.0 * np.log(1.0 - u3)) * np.cos(2 * math.pi * u4)
    x = np.empty((n, 1, h, w), dtype)
    y = np.empty((n, 1, h, w), dtype)
    for i in range(n):
        sd, sp = field_sigma("dm", z[i]), field_sigma("pressure", z[i])
        dm_raw = np.exp(0.9 * g[i] - 0.4)
        p_raw = 0.05 * np.exp(0.8 * g[i] + 0.6 * g2[i] - 0.5)
        y[i, 0] = shift_log(dm_raw, sd)
        x[i, 0] = shift_log(p_raw, sp)
    return x, y, z.astype(dtype)


def synthetic_eps(shape, seed=99, dtype=np.float32):
    """Standard-normal noise for the reparametrisation sampler (injected in place
    of torch.randn, cvae.py:64)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    u1 = rng.random(shape)
    u2 = rng.random(shape)
    return (np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2 * math.pi * u2)).astype(dtype)


def scaled_architecture(arch, size):
    """The fiducial architecture re-sized to size x size tiles (dim_z = size/32),
    used for small parity cases."""
    import copy
    a = copy.deepcopy(arch)
    a["dim_x"] = (a["dim_x"][0], size, size)
    a["dim_y"] = (a["dim_y"][0], size, size)
    zc = a["dim_z"][0]
    a["dim_z"] = (zc, size // 32, size // 32)

    def fix(seq):
        return [("unflatten", (2, zc, size // 32, size // 32)) if l[0] == "unflatten" else l for l in seq]
    a["q_x_y_out"] = fix(a["q_x_y_out"])
    if "prior_z_y" in a:
        a["prior_z_y"] = fix(a["prior_z_y"])
    return a


def softened_architecture(arch, slope=0.9):
    """The same network with every ReLU (in the sequences, inside residual blocks and at their tails) replaced by
    ``("Leaky ReLU", slope)`` -- same layers, same kernels, same wiring, but the kink of every activation is
    (1 - slope) of a ReLU's.  Gradients of the fiducial net are discontinuous wherever a pre-activation crosses zero,
    and some unit always sits within float32 rounding of zero, which puts a ~1e-2 noise floor under any fp32
    gradient comparison (DESIGN.md "Numerical parity").  With slope 0.9 that floor drops tenfold and a 1 % kernel
    error can no longer hide under it (the WELL-CONDITIONED gradient case of the parity tests; PReLU slopes are set
    to ``slope`` by ``soften_params``)."""
    import copy
    a = copy.deepcopy(arch)

    def soft(seq):
        if seq is None:
            return None
        out = []
        for layer in seq:
            name = layer[0].lower()
            if name == "relu":
                out.append(("Leaky ReLU", slope))
            elif name == "residual block":
                body, tail = layer[1]
                tail = ("Leaky ReLU", slope) if tail[0] is not None and tail[0].lower() == "relu" else tail
                out.append((layer[0], (soft(body), tail)))
            else:
                out.append(layer)
        return out
    for k in ("prior_z_y", "q_x_in", "q_y_in", "q_x_y_out", "p_y_in", "p_z_in", "p_y_z_in"):
        if k in a:
            a[k] = soft(a[k])
    a["p_y_z_out"] = tuple(soft(h) for h in a["p_y_z_out"])
    return a


def soften_params(params, slope=0.9):
    """PReLU slopes (the one-element ``.weight`` parameters) of a ``fill_params`` result set to ``slope``."""
    shapes = {k: v.shape for k, v in params.items()}
    out = dict(params)
    for k, v in params.items():
        if tuple(v.shape) == (1,) and k.endswith("weight") and _is_prelu(k, shapes):
            out[k] = np.full((1,), slope, dtype=v.dtype)
    return out
