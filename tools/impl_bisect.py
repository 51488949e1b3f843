"""Debug aid: run the same step with MFMA and direct kernels, report per-unit d_raw / weight-grad differences."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from baryon_painter_amd.models import arch as A
from baryon_painter_amd.models.cvae import CVAE
from baryon_painter_amd.utils import synthetic as syn
size, n = 128, 2
arch = A.fiducial_architecture(size)
x, y, aux = syn.synthetic_batch(n, size, size, seed=1234)
eps = syn.synthetic_eps((1, n, *arch["dim_z"]), seed=99)
models = {}
for impl in ("mfma", "direct"):
    m = CVAE(arch, "cuda:0", impl=impl)
    P = syn.fill_params({k: tuple(p.shape) for k, p in m.named_parameters()}, 7)
    with torch.no_grad():
        for k, p in m.named_parameters(): p.copy_(torch.from_numpy(P[k]))
    m._eps_override = eps
    e = m(torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(aux)); (-e).backward()
    models[impl] = m
def units(plan):
    out = []
    def walk(us):
        for u in us:
            if hasattr(u, "body"):
                walk(u.body); out.append((u.name + ".tail", u.out))
            else:
                out.append((u.name, u.out))
    for us in plan.q_units: walk(us)
    walk(plan.p_units)
    for us in plan.g_units: walk(us)
    walk(plan.mu_units)
    return out
ua, ub = units(models["mfma"]._last), units(models["direct"]._last)
def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))
for (na, sa), (nb, sb) in reversed(list(zip(ua, ub))):
    ga = sa.grad_buf[..., sa.coff:sa.coff + sa.c]; gb = sb.grad_buf[..., sb.coff:sb.coff + sb.c]
    ra = sa.buf[..., sa.coff:sa.coff + sa.c]; rb = sb.buf[..., sb.coff:sb.coff + sb.c]
    print(f"{na:30s} raw diff {rel(ra, rb):.2e}   d_raw diff {rel(ga, gb):.2e}")
