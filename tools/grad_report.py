"""Debug aid: per-parameter distance of the HIP gradients from the float64 truth, next to the
fp32 reference's own distance (fixtures: tests/golden/model.npz)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from baryon_painter_amd.models import arch as A
from baryon_painter_amd.models.cvae import CVAE
from baryon_painter_amd.utils import synthetic as syn
from golden_util import distance, summary_distance

tag, size, n = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else ("fid512_n2", 512, 2)
impl = sys.argv[4] if len(sys.argv) > 4 else "auto"
gold = np.load(os.path.join(ROOT, "tests", "golden", "model.npz"))
arch = A.fiducial_architecture(size)
m = CVAE(arch, "cuda:0", impl=impl)
P = syn.fill_params({k: tuple(p.shape) for k, p in m.named_parameters()}, 7)
with torch.no_grad():
    for k, p in m.named_parameters():
        p.copy_(torch.from_numpy(P[k]))
x, y, aux = syn.synthetic_batch(n, size, size, seed=1234)
m._eps_override = syn.synthetic_eps((1, n, *arch["dim_z"]), seed=99)
elbo = m(torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(aux))
(-elbo).backward()
rows = []
for k, p in m.named_parameters():
    ours = distance(f"{tag}/grad64/{k}", p.grad.cpu().numpy(), gold)
    ref = summary_distance(f"{tag}/grad/{k}", f"{tag}/grad64/{k}", gold)
    rows.append((ours / max(ref, 1e-9), ours, ref, k))
for r in sorted(rows, reverse=True)[:25]:
    print("ratio %8.2f  ours %.2e  ref %.2e  %s" % r)
print("forward: stats", distance(f"{tag}/stats", np.array(m.get_stats()), gold), "x_mu", distance(f"{tag}/x_mu", m.x_mu.cpu().numpy(), gold),
      "z_mu", distance(f"{tag}/z_mu", m.z_mu.cpu().numpy(), gold), "z_lv", distance(f"{tag}/z_log_var", m.z_log_var.cpu().numpy(), gold))
for k, b in m.named_buffers():
    if "running" in k and ("p_mu" in k or "p_z_in" in k or "p_y_z_in.1." in k or "p_y_z_in.23" in k):
        print(k, distance(f"{tag}/buf/{k}", b.cpu().numpy(), gold))
