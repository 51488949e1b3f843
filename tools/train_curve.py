"""Do the fp32 (parity) and the bf16 (throughput) modes train alike?  The same initial weights, the same synthetic
tiles in the same order, FlatAdam(lr 1e-3), N steps each; prints the ELBO every `every` steps side by side.
usage: train_curve.py [steps] [batch] [tile]"""
import contextlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from baryon_painter_amd.models import arch as A
from baryon_painter_amd.models.cvae import CVAE
from baryon_painter_amd.optim import FlatAdam
from baryon_painter_amd.utils import synthetic as syn

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
tile = int(sys.argv[3]) if len(sys.argv) > 3 else 256
every = 20
arch = A.fiducial_architecture(tile)
pool = 8
data = [syn.synthetic_batch(batch, tile, tile, seed=100 + i) for i in range(pool)]
curves = {}
for dtype in ("f32", "bf16"):
    torch.manual_seed(7)
    with contextlib.redirect_stdout(sys.stderr):
        m = CVAE(arch, "cuda:0", dtype=dtype)
    opt = FlatAdam(m, lr=1e-3)
    gen = torch.Generator(device="cuda").manual_seed(11)
    out = []
    for it in range(steps):
        x, y, aux = data[it % pool]
        m._eps_override = torch.randn((1, batch, *arch["dim_z"]), device="cuda", generator=gen).cpu().numpy()
        elbo = m(torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(aux))
        opt.zero_grad()
        (-elbo).backward()
        opt.step()
        if it % every == 0 or it == steps - 1:
            out.append((it, float(elbo.detach())))
    curves[dtype] = out
    del m, opt
    torch.cuda.empty_cache()
print(f"# CVAE fiducial at {tile}^2, batch {batch}, FlatAdam lr 1e-3, {pool} synthetic batches cycled, same init / noise")
print(f"{'step':>6s} {'ELBO fp32':>14s} {'ELBO bf16':>14s} {'rel diff':>10s}")
worst = 0.0
for (it, a), (_, b) in zip(curves["f32"], curves["bf16"]):
    rel = abs(a - b) / max(abs(a), 1e-30)
    worst = max(worst, rel)
    print(f"{it:6d} {a:14.2f} {b:14.2f} {rel:10.2e}")
print(f"# ELBO improved by a factor {curves['f32'][0][1] / curves['f32'][-1][1]:.1f} (fp32) / "
      f"{curves['bf16'][0][1] / curves['bf16'][-1][1]:.1f} (bf16); largest relative gap between the two curves {worst:.2e}")
