"""Micro-benchmark of one bf16 convolution layer through the C ABI (forward, data gradient, weight gradient).
usage: conv_bench_bf16.py transposed cin cout k stride pad n h w [reps]
YF32=1: the produced tensor and its gradient are fp32 (the heads' first layer: bf16 trunk in, fp32 out); PWIN=1: a
pending batch-norm + ReLU on the input (as in the network); STATS=1: also the forward with the statistics epilogue."""
import ctypes as C, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from baryon_painter_amd import _lib as L
tr, ci, co, k, s, p, n, h, w = map(int, sys.argv[1:10])
reps = int(sys.argv[10]) if len(sys.argv) > 10 else 20
lib = L.load()
cv = L.Conv(tr, ci, co, k, s, p, 0)
ho = (h - 1) * s - 2 * p + k if tr else (h + 2 * p - k) // s + 1
wo = (w - 1) * s - 2 * p + k if tr else (w + 2 * p - k) // s + 1
xdt = torch.float32 if os.environ.get("XF32") == "1" else torch.bfloat16     # XF32=1: fp32 input (the stem); XCS: its channel stride
XT = L.F32 if xdt == torch.float32 else L.BF16
xcs = int(os.environ.get("XCS", ci))
x = torch.randn((n, h, w, xcs), device="cuda").to(xdt)
ydt = torch.float32 if os.environ.get("YF32") == "1" else torch.bfloat16
YT = L.F32 if ydt == torch.float32 else L.BF16
y = torch.empty((n, ho, wo, co), device="cuda", dtype=ydt)
dy = torch.randn((n, ho, wo, co), device="cuda").to(ydt)
dx = torch.empty_like(x)
wt = torch.randn(((ci, co) if tr else (co, ci)) + (k, k), device="cuda") * 0.05
xv = L.View(x.data_ptr(), n, h, w, ci, xcs, 0, XT)
yv = L.View(y.data_ptr(), n, ho, wo, co, co, 0, YT)
dyv = L.View(dy.data_ptr(), n, ho, wo, co, co, 0, YT)
pwk = [torch.rand(ci, device="cuda") + 0.5, torch.rand(ci, device="cuda") - 0.5, torch.zeros(ci, device="cuda")]
pw = C.byref(L.Pointwise(*[t.data_ptr() for t in pwk])) if os.environ.get("PWIN") == "1" else None
dxv = L.View(dx.data_ptr(), n, h, w, ci, xcs, 0, XT)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
pf = torch.zeros(lib.bp_conv_bf16_packed_elems(C.byref(cv), 0), device="cuda", dtype=torch.bfloat16)
pb = torch.zeros(lib.bp_conv_bf16_packed_elems(C.byref(cv), 1), device="cuda", dtype=torch.bfloat16)
L.check(lib.bp_conv_bf16_pack(C.byref(cv), 0, L.ptr(wt), L.ptr(pf), st)); L.check(lib.bp_conv_bf16_pack(C.byref(cv), 1, L.ptr(wt), L.ptr(pb), st))
wsb = lib.bp_conv_backward_weight_workspace(C.byref(cv), C.byref(xv), C.byref(dyv))
ws = torch.zeros(wsb // 8 + 8, dtype=torch.float64, device="cuda")
dw = torch.zeros_like(wt)
flop = 2.0 * n * (h * w if tr else ho * wo) * k * k * ci * co
def run(name, fn):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print(f"{name:8s} {dt*1e6:9.1f} us  {flop/dt/1e12:8.1f} TF/s  {(x.numel()*x.element_size()+y.numel()*y.element_size())/dt/1e9:8.1f} GB/s", flush=True)
which = os.environ.get("WHICH", "fdw")
if os.environ.get("STATS") == "1":      # forward with the batch-norm sums from the epilogue (bp_conv_forward_stats)
    nb = lib.bp_conv_stats_workspace(C.byref(cv), 0, C.byref(xv), C.byref(yv), L.IMPL_BF16)
    sums = torch.zeros(2 * co, dtype=torch.float64, device="cuda")
    wss = torch.zeros(nb // 8 + 8, dtype=torch.float64, device="cuda")
    run("fwd+stat", lambda: L.check(lib.bp_conv_forward_stats(C.byref(cv), C.byref(xv), pw, L.ptr(pf), C.byref(yv), L.ptr(sums), L.ptr(wss), nb, L.IMPL_BF16, st)))
if "f" in which: run("forward", lambda: L.check(lib.bp_conv_forward(C.byref(cv), C.byref(xv), pw, L.ptr(pf), L.ptr(wt), None, C.byref(yv), L.IMPL_BF16, st)))
if "d" in which: run("dgrad", lambda: L.check(lib.bp_conv_backward_data(C.byref(cv), C.byref(dyv), L.ptr(pb), L.ptr(wt), C.byref(dxv), L.IMPL_BF16, st)))
if "w" in which: run("wgrad", lambda: L.check(lib.bp_conv_backward_weight(C.byref(cv), C.byref(xv), pw, C.byref(dyv), L.ptr(dw), None, L.ptr(ws), ws.numel() * 8, L.IMPL_BF16, st)))
