"""Debug aid: the bf16 (configs[3]) step against the fp32 step of the same model and inputs: losses, x_mu, every
gradient (relative L2), sample_P; and a quick timing.   usage: bf16_report.py [size] [n]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from baryon_painter_amd.models import arch as A
from baryon_painter_amd.models.cvae import CVAE
from baryon_painter_amd.utils import synthetic as syn

size, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (128, 2)
arch = A.fiducial_architecture(size)
x, y, aux = [torch.from_numpy(t) for t in syn.synthetic_batch(n, size, size, seed=1234)]
eps = syn.synthetic_eps((1, n, *arch["dim_z"]), seed=99)
out = {}
for dt in ("f32", "bf16"):
    m = CVAE(arch, "cuda:0", dtype=dt)
    P = syn.fill_params({k: tuple(p.shape) for k, p in m.named_parameters()}, 7)
    with torch.no_grad():
        for k, p in m.named_parameters():
            p.copy_(torch.from_numpy(P[k]))
    m._eps_override = eps
    elbo = m(x, y, aux)
    (-elbo).backward()
    torch.cuda.synchronize()
    plan = m._last
    nb = sum(1 for u in plan.pack_batch.units if u.bf16)
    out[dt] = dict(stats=np.array(m.get_stats()), x_mu=m.x_mu.cpu().double(), grads={k: p.grad.cpu().double().clone() for k, p in m.named_parameters()})
    m.train(False)
    out[dt]["sample"] = m.sample_P(y, aux_label=aux, z=syn.synthetic_eps((n, *arch["dim_z"]), seed=101)).cpu().double()
    m.train(True)
    # timing: 5 steps
    for _ in range(2):
        (-m(x, y, aux)).backward()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(5):
        (-m(x, y, aux)).backward()
    torch.cuda.synchronize()
    print(dt, "bf16 units:", nb, "stats", out[dt]["stats"], "ms/step (fwd+bwd, eager)", (time.time() - t0) / 5 * 1e3, flush=True)
a, b = out["f32"], out["bf16"]
rl2 = lambda u, v: float((u - v).norm() / v.norm().clamp_min(1e-300))
print("stats rel diff", np.abs(b["stats"] - a["stats"]) / np.abs(a["stats"]))
print("x_mu rel-L2", rl2(b["x_mu"], a["x_mu"]), "sample_P rel-L2", rl2(b["sample"], a["sample"]))
rows = sorted(((rl2(b["grads"][k], a["grads"][k]), k) for k in a["grads"]), reverse=True)
for r in rows[:12]:
    print("grad rel-L2 %.3e  %s" % r)
print("median grad rel-L2", np.median([r[0] for r in rows]))
