"""Time single convolution launches through the C ABI: python tools/conv_bench.py "t,ci,co,k,s,p,n,h,w" ..."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from baryon_painter_amd import _lib as L
import gpu_util as G
lib = L.load()
st = G.stream()
IMPL = L.IMPL_DIRECT if os.environ.get('BENCH_IMPL') == 'direct' else L.IMPL_MFMA
for spec in sys.argv[1:]:
    tr, ci, co, k, s, p, n, h, w = map(int, spec.split(","))
    cv = L.Conv(tr, ci, co, k, s, p, 0)
    ho = (h - 1) * s - 2 * p + k if tr else (h + 2 * p - k) // s + 1
    wo = (w - 1) * s - 2 * p + k if tr else (w + 2 * p - k) // s + 1
    xb = torch.randn((n, h, w, ci), device="cuda"); xv = L.View(xb.data_ptr(), n, h, w, ci, ci, 0)
    yb = torch.empty((n, ho, wo, co), device="cuda"); yv = L.View(yb.data_ptr(), n, ho, wo, co, co, 0)
    dxb = torch.empty_like(xb); dxv = L.View(dxb.data_ptr(), n, h, w, ci, ci, 0)
    wd = torch.randn(((ci, co) if tr else (co, ci)) + (k, k), device="cuda") * 0.05
    keep, pw = G.pointwise(np.ones(ci, np.float32), np.zeros(ci, np.float32), np.full(ci, 0.2, np.float32))
    pwp = None if os.environ.get('BENCH_NOPW') else C.byref(pw)
    pf = torch.zeros(lib.bp_conv_packed_floats(C.byref(cv), L.PACK_FWD), device="cuda")
    pb = torch.zeros(lib.bp_conv_packed_floats(C.byref(cv), L.PACK_BWD), device="cuda")
    L.check(lib.bp_conv_pack(C.byref(cv), L.PACK_FWD, L.ptr(wd), L.ptr(pf), st))
    L.check(lib.bp_conv_pack(C.byref(cv), L.PACK_BWD, L.ptr(wd), L.ptr(pb), st))
    wsb = lib.bp_conv_backward_weight_workspace(C.byref(cv), C.byref(xv), C.byref(yv))
    ws = torch.zeros(wsb // 8 + 8, dtype=torch.float64, device="cuda")
    dw = torch.zeros_like(wd)
    dense = (n * h * w) if tr else (n * ho * wo)
    flop = 2.0 * dense * k * k * ci * co
    def run(kind):
        if kind == "fwd":
            L.check(lib.bp_conv_forward(C.byref(cv), C.byref(xv), pwp, L.ptr(pf), L.ptr(wd), None, C.byref(yv), IMPL, st))
        elif kind == "dgrad":
            L.check(lib.bp_conv_backward_data(C.byref(cv), C.byref(yv), L.ptr(pb), L.ptr(wd), C.byref(dxv), IMPL, st))
        else:
            L.check(lib.bp_conv_backward_weight(C.byref(cv), C.byref(xv), C.byref(pw), C.byref(yv), L.ptr(dw), None, L.ptr(ws), ws.numel() * 8, IMPL, st))
    out = []
    for kind in ("fwd", "dgrad", "wgrad"):
        for _ in range(2):
            run(kind)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run(kind)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        out.append("%s %.3f ms %.1f TF" % (kind, ms, flop / ms / 1e9))
    ids = [lib.bp_conv_kernel_id(C.byref(cv), d) for d in (L.PACK_FWD, L.PACK_BWD)]
    print(spec, ids, " | ".join(out), flush=True)
