"""Compact tensor summaries used by the golden fixtures (tests/golden/*.npz).

A tensor is stored either whole (<= FULL_LIMIT elements) or as
(shape, l2 norm, sum, values at NSAMPLE fixed pseudo-random flat indices).
"""
import numpy as np

FULL_LIMIT = 4096
NSAMPLE = 96


def sample_indices(n, tag=0):
    rng = np.random.Generator(np.random.PCG64(1000003 + tag))
    return (rng.random(NSAMPLE) * n).astype(np.int64)


def summarize(prefix, arr, out):
    a = np.asarray(arr, dtype=np.float64)
    out[prefix + "/shape"] = np.array(a.shape, dtype=np.int64)
    if a.size <= FULL_LIMIT:
        out[prefix + "/full"] = a.astype(np.float32)
    else:
        flat = a.reshape(-1)
        out[prefix + "/l2"] = np.array(np.sqrt((flat ** 2).sum()))
        out[prefix + "/sum"] = np.array(flat.sum())
        out[prefix + "/samples"] = flat[sample_indices(flat.size)].astype(np.float32)


# Contiguous full-resolution crops of a large (N,C,H,W) tensor: two opposite corners (padding handling on all four
# borders) of every sample, stored whole, so that a TRUE relative L2 distance ||a-b|| / ||b|| can be computed.
CROP = 96


def crop_views(arr):
    a = np.asarray(arr)
    return {"tl": a[..., :CROP, :CROP], "br": a[..., -CROP:, -CROP:]}


def store_crops(prefix, arr, out):
    for name, v in crop_views(arr).items():
        out[f"{prefix}/crop_{name}"] = np.ascontiguousarray(v, dtype=np.float32)


def crop_rel_l2(prefix, arr, gold):
    """||ours - ref||_2 / ||ref||_2 over the stored crops."""
    num = den = 0.0
    for name, v in crop_views(arr).items():
        g = gold[f"{prefix}/crop_{name}"].astype(np.float64)
        num += ((np.asarray(v, np.float64) - g) ** 2).sum()
        den += (g ** 2).sum()
    return float(np.sqrt(num / max(den, 1e-300)))


def check(prefix, arr, gold, rtol, atol_frac=1e-6, what=""):
    """Assert ``arr`` matches the stored summary.  ``rtol`` is relative to the
    tensor's overall scale (l2/sqrt(n) or max|.|), which is the meaningful
    measure for conv outputs/gradients."""
    a = np.asarray(arr, dtype=np.float64)
    shape = tuple(int(v) for v in gold[prefix + "/shape"])
    assert a.shape == shape, f"{what}{prefix}: shape {a.shape} vs golden {shape}"
    if prefix + "/full" in gold:
        g = gold[prefix + "/full"].astype(np.float64)
        scale = max(np.abs(g).max(), 1e-30)
        err = np.abs(a - g).max() / scale
        assert err <= rtol, f"{what}{prefix}: max err/scale {err:.3e} > {rtol:.1e}"
        return err
    flat = a.reshape(-1)
    gl2 = float(gold[prefix + "/l2"])
    rms = max(gl2 / np.sqrt(flat.size), 1e-30)
    l2 = np.sqrt((flat ** 2).sum())
    e1 = abs(l2 - gl2) / max(gl2, 1e-30)
    gs = gold[prefix + "/samples"].astype(np.float64)
    e2 = np.abs(flat[sample_indices(flat.size)] - gs).max() / max(np.abs(gs).max(), rms)
    assert e1 <= rtol, f"{what}{prefix}: l2 rel err {e1:.3e} > {rtol:.1e}"
    assert e2 <= rtol * 4, f"{what}{prefix}: sample err/scale {e2:.3e} > {4*rtol:.1e}"
    return max(e1, e2)


def summary_distance(prefix_a, prefix_b, gold):
    """Distance between two stored summaries of the same tensor (same measure as check())."""
    if prefix_a + "/full" in gold:
        a = gold[prefix_a + "/full"].astype(np.float64)
        b = gold[prefix_b + "/full"].astype(np.float64)
        return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
    la, lb = float(gold[prefix_a + "/l2"]), float(gold[prefix_b + "/l2"])
    sa = gold[prefix_a + "/samples"].astype(np.float64)
    sb = gold[prefix_b + "/samples"].astype(np.float64)
    n = int(np.prod(gold[prefix_b + "/shape"]))
    rms = max(lb / np.sqrt(n), 1e-30)
    return max(abs(la - lb) / max(lb, 1e-30), np.abs(sa - sb).max() / max(np.abs(sb).max(), rms))


def distance(prefix, arr, gold):
    """Distance of ``arr`` from a stored summary, without asserting."""
    a = np.asarray(arr, dtype=np.float64)
    if prefix + "/full" in gold:
        g = gold[prefix + "/full"].astype(np.float64).reshape(a.shape)
        return np.abs(a - g).max() / max(np.abs(g).max(), 1e-30)
    flat = a.reshape(-1)
    gl2 = float(gold[prefix + "/l2"])
    rms = max(gl2 / np.sqrt(flat.size), 1e-30)
    gs = gold[prefix + "/samples"].astype(np.float64)
    e1 = abs(np.sqrt((flat ** 2).sum()) - gl2) / max(gl2, 1e-30)
    e2 = np.abs(flat[sample_indices(flat.size)] - gs).max() / max(np.abs(gs).max(), rms)
    return max(e1, e2)


def distance_arrays(arr, ref):
    """``distance`` between two arrays in hand: the measure a stored summary of ``ref`` would give."""
    out = {}
    summarize("t", ref, out)
    return distance("t", arr, out)
