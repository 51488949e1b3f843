"""Debug aid: bisect where the p_z_in.7 batch-norm bias gradient (a heavily cancelling sum of the
data-gradient of p_y_z_in.0) loses accuracy at 512^2."""
import ctypes as C, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from baryon_painter_amd import _lib as L
from baryon_painter_amd.models import arch as A
from baryon_painter_amd.models.cvae import CVAE
from baryon_painter_amd.utils import synthetic as syn
from oracle import ops
size, n = int(sys.argv[1]), int(sys.argv[2])
arch = A.fiducial_architecture(size)
m = CVAE(arch, "cuda:0")
P = syn.fill_params({k: tuple(p.shape) for k, p in m.named_parameters()}, 7)
with torch.no_grad():
    for k, p in m.named_parameters(): p.copy_(torch.from_numpy(P[k]))
x, y, aux = syn.synthetic_batch(n, size, size, seed=1234)
m._eps_override = syn.synthetic_eps((1, n, *arch["dim_z"]), seed=99)
e = m(torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(aux)); (-e).backward()
plan, lib = m._last, m._lib
u = plan.g_units[1][0]                                   # p_y_z_in.0 (+BN+ReLU)
uz = plan.g_units[0][-1]                                 # p_z_in.6 (+BN p_z_in.7 + ReLU)
draw = u.out.grad_buf.clone()                            # d_raw of p_y_z_in.0 (16 ch)
w = u.holder.weight.detach()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
out = {}
for name, impl in (("mfma", L.IMPL_MFMA), ("direct", L.IMPL_DIRECT)):
    dx = torch.zeros_like(plan.p_in.buf)
    v = L.View(dx.data_ptr(), plan.p_in.n, plan.p_in.h, plan.p_in.w, plan.p_in.c, plan.p_in.cstride, 0)
    gv = L.View(draw.data_ptr(), u.out.n, u.out.h, u.out.w, u.out.c, u.out.cstride, 0)
    L.check(lib.bp_conv_backward_data(C.byref(u.cv), C.byref(gv), L.ptr(u.packed_bwd), L.ptr(w), C.byref(v), impl, st))
    out[name] = dx[..., 0].cpu().numpy().astype(np.float64)
d64 = ops.conv2d_bwd_data(draw.cpu().numpy().astype(np.float64).transpose(0, 3, 1, 2), w.cpu().numpy().astype(np.float64),
                          1, 2, size, size)[:, 0]
raw = plan.p_in.buf[..., 0].cpu().numpy().astype(np.float64)
sc, sh = uz.out_pw.scale.item(), uz.out_pw.shift.item()
mask = (raw * sc + sh) > 0
print("masked fraction", mask.mean())
for k, d in out.items():
    err = d - d64
    print(f"{k:7s} rms {np.sqrt((d64**2).mean()):.3e} rms err {np.sqrt((err**2).mean()):.3e} mean err {err.mean():.3e} "
          f"S0 {(d*mask).sum():.6f} vs f64 {(d64*mask).sum():.6f}")
print("dbeta reported", m.get_parameter("p_z_in.7.bias").grad.item())
