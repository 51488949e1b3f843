"""Debug aid: direct vs MFMA kernels vs float64 NumPy on one larger layer."""
import ctypes as C, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from baryon_painter_amd import _lib as L
from oracle import ops
import gpu_util as G
lib = L.load()
rng = np.random.default_rng(0)
for (ci, co, k, s, p, n, h, w) in [(128, 128, 3, 1, 1, 2, 32, 32), (16, 32, 4, 2, 1, 2, 64, 64), (16, 8, 7, 1, 3, 2, 64, 64)]:
    x = rng.standard_normal((n, ci, h, w)).astype(np.float32)
    wt = (rng.standard_normal((co, ci, k, k)) / np.sqrt(ci * k * k)).astype(np.float32)
    cv = L.Conv(0, ci, co, k, s, p, 0)
    y64 = ops.conv2d_fwd(x.astype(np.float64), wt.astype(np.float64), s, p)
    dy = rng.standard_normal(y64.shape).astype(np.float32)
    dx64 = ops.conv2d_bwd_data(dy.astype(np.float64), wt.astype(np.float64), s, p, h, w)
    dw64 = ops.conv2d_bwd_weight(x.astype(np.float64), dy.astype(np.float64), s, p, k, k)
    st = G.stream()
    xb, xv = G.to_nhwc(x); dyb, dyv = G.to_nhwc(dy)
    wd = G.dev(wt)
    pf = torch.zeros(lib.bp_conv_packed_floats(C.byref(cv), 0), device="cuda"); pb = torch.zeros(lib.bp_conv_packed_floats(C.byref(cv), 1), device="cuda")
    L.check(lib.bp_conv_pack(C.byref(cv), 0, L.ptr(wd), L.ptr(pf), st)); L.check(lib.bp_conv_pack(C.byref(cv), 1, L.ptr(wd), L.ptr(pb), st))
    for name, impl in (("mfma", 2), ("direct", 1)):
        yb, yv = G.empty_nhwc(n, y64.shape[2], y64.shape[3], co)
        L.check(lib.bp_conv_forward(C.byref(cv), C.byref(xv), None, L.ptr(pf), L.ptr(wd), None, C.byref(yv), impl, st))
        dxb, dxv = G.empty_nhwc(n, h, w, ci)
        L.check(lib.bp_conv_backward_data(C.byref(cv), C.byref(dyv), L.ptr(pb), L.ptr(wd), C.byref(dxv), impl, st))
        ws = torch.zeros(lib.bp_conv_backward_weight_workspace(C.byref(cv), C.byref(xv), C.byref(dyv)) // 8 + 8, dtype=torch.float64, device="cuda")
        dw = torch.zeros(wt.shape, device="cuda")
        L.check(lib.bp_conv_backward_weight(C.byref(cv), C.byref(xv), None, C.byref(dyv), L.ptr(dw), None, L.ptr(ws), ws.numel() * 8, impl, st))
        print(f"{ci}->{co} k{k}s{s} {name:6s} fwd {G.rel_err(G.from_nhwc(yb, co), y64):.2e} dgrad {G.rel_err(G.from_nhwc(dxb, ci), dx64):.2e} wgrad {G.rel_err(dw.cpu().numpy(), dw64):.2e}")
        if name:
            e = G.from_nhwc(yb, co).astype(np.float64) - y64
            print(f"      fwd error: mean {e.mean():.3e} rms {np.sqrt((e**2).mean()):.3e} max {np.abs(e).max():.3e}  corr(e,y) {np.corrcoef(e.ravel(), y64.ravel())[0,1]:.3e}")
