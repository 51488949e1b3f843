import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from baryon_painter_amd.models import arch as A
from baryon_painter_amd.models.cvae import CVAE
from baryon_painter_amd.utils import synthetic as syn
gold = np.load(os.path.join(ROOT, "tests", "golden", "model.npz"))
size, n = 128, 2
arch = A.fiducial_architecture(size)
m = CVAE(arch, "cuda:0", impl="mfma")
P = syn.fill_params({k: tuple(p.shape) for k, p in m.named_parameters()}, 7)
with torch.no_grad():
    for k, p in m.named_parameters(): p.copy_(torch.from_numpy(P[k]))
x, y, aux = syn.synthetic_batch(n, size, size, seed=1234)
m._eps_override = syn.synthetic_eps((1, n, *arch["dim_z"]), seed=99)
for rep in range(2):
    for p in m.parameters(): p.grad = None
    e = m(torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(aux)); (-e).backward()
    for k in ("p_z_in.4.bias", "p_z_in.7.weight", "p_mu_out.1.weight", "p_z_in.1.bias"):
        print(rep, k, "gpu %.9e" % m.get_parameter(k).grad.item(), "truth %.9e" % float(gold[f"fid128_n2/grad64/{k}/full"].ravel()[0]),
              "ref %.9e" % float(gold[f"fid128_n2/grad/{k}/full"].ravel()[0]))
    print("elbo %.9e" % float(e.detach()), "running_mean p_z_in.1", m.get_buffer("p_z_in.1.running_mean").item())
