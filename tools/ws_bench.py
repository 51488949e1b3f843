"""A/B of the weights-stationary trunk kernel (csrc/conv_bf16_ws.hip) against the tiled igemm_bf16_kernel, in ONE process,
interleaved rounds, random data (cdna_hip_programming.md rule 24 / 25).  usage: ws_bench.py [n h w [rounds]]"""
import ctypes as C, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from baryon_painter_amd import _lib as L

n, h, w = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 64, 64)
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 15
LAYER = os.environ.get("LAYER", "k3")        # k3: 128->128 k3 (h, w = the layer's grid); k4s2 / t4s2: h, w = the COARSE grid
#                                              k3f32: the fp32 trunk layer (csrc/conv_ws_f32.hip against igemm_dma_kernel)
lib = L.load()
F32 = LAYER == "k3f32"
tr, ci, co, k, s_, p_ = {"k3": (0, 128, 128, 3, 1, 1), "k3f32": (0, 128, 128, 3, 1, 1), "k4s2": (0, 64, 128, 4, 2, 1),
                         "t4s2": (1, 128, 64, 4, 2, 1)}[LAYER]
cv = L.Conv(tr, ci, co, k, s_, p_, 0)
hi, wi = (h, w) if LAYER != "k4s2" else (2 * h, 2 * w)          # module input
ho, wo = (h, w) if LAYER != "t4s2" else (2 * h, 2 * w)          # module output
DT, VT = (torch.float32, L.F32) if F32 else (torch.bfloat16, L.BF16)
x = torch.randn((n, hi, wi, ci), device="cuda").to(DT)
y = torch.empty((n, ho, wo, co), device="cuda", dtype=DT)
dy = torch.randn((n, ho, wo, co), device="cuda").to(DT)
dx = torch.empty_like(x)
wt = torch.randn(((ci, co) if tr else (co, ci)) + (k, k), device="cuda") * 0.05
xv = L.View(x.data_ptr(), n, hi, wi, ci, ci, 0, VT)
yv = L.View(y.data_ptr(), n, ho, wo, co, co, 0, VT)
dyv = L.View(dy.data_ptr(), n, ho, wo, co, co, 0, VT)
dxv = L.View(dx.data_ptr(), n, hi, wi, ci, ci, 0, VT)
pwk = [torch.rand(ci, device="cuda") + 0.5, torch.rand(ci, device="cuda") - 0.5, torch.zeros(ci, device="cuda")]
pw = L.Pointwise(*[t.data_ptr() for t in pwk])
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
IMPL = L.IMPL_MFMA if F32 else L.IMPL_BF16
OPT = b"f32_ws" if F32 else b"bf16_ws"
if F32:
    pf = torch.zeros(lib.bp_conv_packed_floats(C.byref(cv), 0), device="cuda")
    pb = torch.zeros(lib.bp_conv_packed_floats(C.byref(cv), 1), device="cuda")
    L.check(lib.bp_conv_pack(C.byref(cv), 0, L.ptr(wt), L.ptr(pf), st))
    L.check(lib.bp_conv_pack(C.byref(cv), 1, L.ptr(wt), L.ptr(pb), st))
else:
    pf = torch.zeros(lib.bp_conv_bf16_packed_elems(C.byref(cv), 0), device="cuda", dtype=torch.bfloat16)
    pb = torch.zeros(lib.bp_conv_bf16_packed_elems(C.byref(cv), 1), device="cuda", dtype=torch.bfloat16)
    L.check(lib.bp_conv_bf16_pack(C.byref(cv), 0, L.ptr(wt), L.ptr(pf), st))
    L.check(lib.bp_conv_bf16_pack(C.byref(cv), 1, L.ptr(wt), L.ptr(pb), st))
lib.bp_set_option(OPT, 1)
nb = max(lib.bp_conv_stats_workspace(C.byref(cv), 0, C.byref(xv), C.byref(yv), IMPL), 8)
sums = torch.zeros(2 * co, dtype=torch.float64, device="cuda")
wss = torch.zeros(nb // 8 + 8, dtype=torch.float64, device="cuda")
flop = 2.0 * n * h * w * (9 if LAYER in ("k3", "k3f32") else 16) * ci * co
PEAK = 157.3 if F32 else 2500.0

legs = {
    "fwd": lambda: lib.bp_conv_forward(C.byref(cv), C.byref(xv), None, L.ptr(pf), L.ptr(wt), None, C.byref(yv), IMPL, st),
    "fwd+act": lambda: lib.bp_conv_forward(C.byref(cv), C.byref(xv), C.byref(pw), L.ptr(pf), L.ptr(wt), None, C.byref(yv), IMPL, st),
    "fwd+act+stats": lambda: lib.bp_conv_forward_stats(C.byref(cv), C.byref(xv), C.byref(pw), L.ptr(pf), C.byref(yv), L.ptr(sums), L.ptr(wss), nb, IMPL, st),
    "dgrad": lambda: lib.bp_conv_backward_data(C.byref(cv), C.byref(dyv), L.ptr(pb), L.ptr(wt), C.byref(dxv), IMPL, st),
}
times = {(k, o): [] for k in legs for o in (1, 0)}
REP = 10
for r in range(rounds + 2):
    for o in (1, 0):
        lib.bp_set_option(OPT, o)
        for k, fn in legs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(REP):
                L.check(fn())
            e1.record()
            torch.cuda.synchronize()
            if r >= 2:
                times[(k, o)].append(e0.elapsed_time(e1) * 1e3 / REP)
lib.bp_set_option(OPT, -1)
print(f"{LAYER} bf16, batch {n} of {h}x{w}: us per launch incl. its reductions (median / min over {rounds} rounds of {REP})")
for k in legs:
    a, b = np.array(times[(k, 1)]), np.array(times[(k, 0)])
    print(f"  {k:14s} stationary {np.median(a):7.1f} / {a.min():7.1f} us = {flop/np.median(a)/1e6:7.1f} TF/s ({flop/np.median(a)/1e6/PEAK:.3f} of the {PEAK:g} TF peak)"
          f"   tiled {np.median(b):7.1f} / {b.min():7.1f} us = {flop/np.median(b)/1e6:7.1f} TF/s", flush=True)
