"""Debug aid: forward error of one in-model convolution (realistic activations) per kernel family."""
import ctypes as C, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from baryon_painter_amd import _lib as L
from baryon_painter_amd.models import arch as A
from baryon_painter_amd.models.cvae import CVAE
from baryon_painter_amd.utils import synthetic as syn
from oracle import ops
size, n = 128, 2
arch = A.fiducial_architecture(size)
m = CVAE(arch, "cuda:0", impl="mfma")
P = syn.fill_params({k: tuple(p.shape) for k, p in m.named_parameters()}, 7)
with torch.no_grad():
    for k, p in m.named_parameters(): p.copy_(torch.from_numpy(P[k]))
x, y, aux = syn.synthetic_batch(n, size, size, seed=1234)
m._eps_override = syn.synthetic_eps((1, n, *arch["dim_z"]), seed=99)
with torch.no_grad():
    m(torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(aux))
plan, lib = m._last, m._lib
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
res = plan.g_units[1][5]                       # second residual block
for u in (res.body[1], plan.g_units[0][2]):
    inp = u.inp
    raw_in = inp.buf[..., inp.coff:inp.coff + inp.c].double().cpu().numpy()
    if inp.pw is not None:
        t = raw_in * inp.pw.scale.double().cpu().numpy() + inp.pw.shift.double().cpu().numpy()
        act = np.where(t > 0, t, t * inp.pw.slope.double().cpu().numpy())
    else:
        act = raw_in
    w = u.holder.weight.detach().double().cpu().numpy()
    a_nchw = act.transpose(0, 3, 1, 2)
    if u.cv.transposed:
        ref = ops.convT2d_fwd(a_nchw, w, u.cv.stride, u.cv.pad)
    else:
        ref = ops.conv2d_fwd(a_nchw, w, u.cv.stride, u.cv.pad)
    print(u.name, "input: frac zeros %.3f, mean %.3e, rms %.3e" % ((act == 0).mean(), act.mean(), np.sqrt((act**2).mean())))
    for name, impl in (("mfma", 2), ("direct", 1)):
        o = torch.zeros_like(u.out.buf)
        ov = L.View(o.data_ptr(), u.out.n, u.out.h, u.out.w, u.out.c, u.out.cstride, u.out.coff)
        L.check(lib.bp_conv_forward(C.byref(u.cv), C.byref(inp.view), inp.pw_struct(), L.ptr(u.packed_fwd),
                                    L.ptr(u.holder.weight), None, C.byref(ov), impl, st))
        got = o[..., u.out.coff:u.out.coff + u.out.c].double().cpu().numpy().transpose(0, 3, 1, 2)
        e = got - ref
        print(f"   {name:6s} err mean {e.mean():.3e} rms {np.sqrt((e**2).mean()):.3e} max {np.abs(e).max():.3e}  ref rms {np.sqrt((ref**2).mean()):.3e}")
