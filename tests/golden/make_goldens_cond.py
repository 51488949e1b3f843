"""Conditioning of the gradient fixtures: how far the float64 TRUE gradient of each test case moves when every
parameter is perturbed by a few units in the last place of float32.

Many gradients of this model are 1000:1 cancelling sums behind ReLU / PReLU masks (cvae.py:26-45 builds p_z_in as
ConvTranspose2d -> BatchNorm -> ReLU on ONE channel: a single latent-level unit carries 1/64 of the map at 128^2).
Where a pre-activation sits within a few ulp of zero, the true gradient is discontinuous within float32 rounding: ANY
float32 evaluation -- the reference's included, see ``grad_variant_dist`` in model.npz -- lands on either side at
random, and no implementation can be closer to the truth than that jump.  This script measures it with the NumPy
float64 oracle alone (no reference import; the oracle is pinned to the reference by tests/test_oracle_golden.py):

    cond[d][k] = distance(grad64 with parameters * (1 + DELTA * u_d), grad64),  u_d ~ U(-1, 1) per element

DELTA = 2^-21 (8 ulp of float32: the accumulated rounding of a float32 dot product of a few hundred terms).
The GPU tests take max(reference executions, cond) as the float32 noise floor of a gradient (tests/test_gpu_model.py).

Run:  python tests/golden/make_goldens_cond.py   -> tests/golden/cond.npz   (about 10 minutes on 8 cores)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from baryon_painter_amd.models import arch as A                 # noqa: E402
from baryon_painter_amd.utils import synthetic as syn           # noqa: E402
from golden_util import distance_arrays                         # noqa: E402
from oracle.cvae_oracle import CVAEOracle                       # noqa: E402

DELTA = 2.0 ** -21
DRAWS = 3
CASES = [("fid128_n2", 128, 2), ("fid256_n4", 256, 4), ("fid512_n2", 512, 2)]


def gradient(arch, n, size, delta, draw, seed_w=7, seed_d=1234, seed_eps=99, alpha=None):
    m = CVAEOracle(arch, dtype=np.float64)
    if alpha is not None:
        m.alpha_var = alpha               # (two-head cases of tests/test_gpu_model.py: the variance head's weight)
    P = syn.fill_params(m.param_shapes(), seed_w)
    if delta > 0:
        rng = np.random.default_rng(1000 + draw)
        P = {k: v.astype(np.float64) * (1.0 + delta * rng.uniform(-1, 1, v.shape)) for k, v in P.items()}
    m.load_params(P)
    x, y, aux = syn.synthetic_batch(n, size, size, seed=seed_d)
    eps = syn.synthetic_eps((arch.get("L", 1), n, *arch["dim_z"]), seed=seed_eps)
    m.forward(x, y, aux, eps)
    return m.backward(seed=-1.0)


def main():
    fid = A.fiducial_architecture(512)
    out = {"delta": np.array(DELTA), "draws": np.array(DRAWS)}
    for tag, size, n in CASES:
        arch = fid if size == 512 else syn.scaled_architecture(fid, size)
        g0 = gradient(arch, n, size, 0.0, 0)
        names = list(g0)
        dist = []
        for d in range(DRAWS):
            g = gradient(arch, n, size, DELTA, d)
            dist.append([distance_arrays(g[k], g0[k]) for k in names])
            print(tag, "draw", d, "largest:", sorted(zip(dist[-1], names), reverse=True)[:4], flush=True)
        out[f"{tag}/grad_cond_params"] = np.array(",".join(names))
        out[f"{tag}/grad_cond_dist"] = np.array(dist)
    np.savez_compressed(os.path.join(HERE, "cond.npz"), **out)
    print("cond.npz", os.path.getsize(os.path.join(HERE, "cond.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
