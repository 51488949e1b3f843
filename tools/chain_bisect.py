"""Debug aid: walk the whole backward chain and compare every convolution's raw output and d(loss)/d(raw) of the
HIP path with the float64 truth (stock torch.nn.functional on the CPU in double), next to the distance of the same
graph evaluated in float32 on the CPU (what the reference computes).  The first tensor whose ratio jumps is where
the HIP path loses accuracy.   usage: chain_bisect.py [size] [n]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from baryon_painter_amd.models import arch as A
from baryon_painter_amd.models.cvae import CVAE
from baryon_painter_amd.utils import synthetic as syn
from oracle.torch_ref import TorchRefCVAE

size, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 2)
arch = A.fiducial_architecture(size)
m = CVAE(arch, "cuda:0")
P = syn.fill_params({k: tuple(p.shape) for k, p in m.named_parameters()}, 7)
with torch.no_grad():
    for k, p in m.named_parameters():
        p.copy_(torch.from_numpy(P[k]))
x, y, aux = syn.synthetic_batch(n, size, size, seed=1234)
eps = syn.synthetic_eps((1, n, *arch["dim_z"]), seed=99)
m._eps_override = eps
elbo = m(torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(aux))
(-elbo).backward()
torch.cuda.synchronize()
plan = m._last

refs = {}
for name, dt in (("f64", torch.float64), ("f32", torch.float32)):
    tap = {}
    r = TorchRefCVAE(arch, P, dtype=dt, tap=tap)
    (-r.forward(x, y, aux, eps)).backward()
    refs[name] = (r, tap)
    print(name, "ELBO", float(r.ELBO), flush=True)
print("hip  ELBO", float(elbo))


def units(plan):
    out = []

    def walk(us):
        for u in us:
            if hasattr(u, "body"):
                walk(u.body)
            else:
                out.append(u)
    for us in plan.q_units:
        walk(us)
    walk(plan.p_units)
    for us in plan.g_units:
        walk(us)
    walk(plan.mu_units)
    return out


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-300)), \
        float((a - b).norm() / b.norm().clamp_min(1e-300))


print(f"{'unit':32s} {'raw hip':>9s} {'raw f32':>9s} | {'d_raw hip (max, l2)':>21s} {'d_raw f32 (max, l2)':>21s}  ratio(l2)")
for u in units(plan):
    key = u.name + "."
    t64, t32 = refs["f64"][1][key], refs["f32"][1][key]
    s = u.out
    raw = s.buf[..., s.coff:s.coff + s.c].permute(0, 3, 1, 2).cpu()
    g = s.grad_buf[..., s.coff:s.coff + s.c].permute(0, 3, 1, 2).cpu()
    r_h, r_32 = rel(raw, t64.detach())[1], rel(t32.detach(), t64.detach())[1]
    if t64.grad is None:
        print(f"{u.name:32s} {r_h:9.2e} {r_32:9.2e} | (no gradient)")
        continue
    gh, g32 = rel(g, t64.grad), rel(t32.grad, t64.grad)
    print(f"{u.name:32s} {r_h:9.2e} {r_32:9.2e} | {gh[0]:9.2e} {gh[1]:9.2e}   {g32[0]:9.2e} {g32[1]:9.2e}   "
          f"{gh[1] / max(g32[1], 1e-30):7.2f}", flush=True)

print("\nparameter gradients: distance from float64 (max/scale), ours vs f32 CPU")
rows = []
for k, p in m.named_parameters():
    t = refs["f64"][0].P[k].grad
    o = rel(p.grad.cpu(), t)[0]
    r = rel(refs["f32"][0].P[k].grad, t)[0]
    rows.append((o / max(r, 1e-12), o, r, k))
for r in sorted(rows, reverse=True):
    print("ratio %8.2f  ours %.2e  f32 %.2e  %s" % r)
out_dir = os.path.join(ROOT, "gpurun_out")
if os.path.isdir(out_dir):
    np.savez_compressed(os.path.join(out_dir, f"grads_hip_{size}_n{n}.npz"),
                        **{k: p.grad.cpu().numpy() for k, p in m.named_parameters()},
                        **{"f64/" + k: refs["f64"][0].P[k].grad.numpy() for k, _ in m.named_parameters()},
                        **{"f32/" + k: refs["f32"][0].P[k].grad.numpy() for k, _ in m.named_parameters()})
