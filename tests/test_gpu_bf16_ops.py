"""GPU: the bf16 matrix-core convolutions (csrc/conv_bf16.hip, impl BP_IMPL_BF16) through the C ABI.

The kernels multiply bf16 x bf16 exactly and accumulate in fp32, so against a float64 convolution of the SAME
bf16-rounded operands (activation applied in fp32, then rounded, as the kernel does) the only differences are the
fp32 accumulation order and, for bf16 outputs, one final rounding:
    fp32 output  <= 2e-5 of the tensor's scale        bf16 output  <= 2^-8 (4e-3) of the tensor's scale.
Every trunk layer class of the fiducial CVAE generator, forward, data gradient and weight
gradient, with each side fp32 or bf16, at sizes that leave ragged tiles."""
import ctypes as C

import numpy as np
import pytest
import torch

from baryon_painter_amd import _lib as L
from oracle import ops

import gpu_util as G

pytestmark = pytest.mark.gpu


def bf16_round(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(torch.bfloat16).to(torch.float32).numpy()


def to_view(x_nchw, bf16, cstride=None, coff=0):
    n, c, h, w = x_nchw.shape
    cs = c if cstride is None else cstride
    buf = torch.full((n, h, w, cs), 7.5, dtype=torch.float32, device="cuda")
    buf[..., coff:coff + c] = torch.from_numpy(np.ascontiguousarray(x_nchw.transpose(0, 2, 3, 1))).cuda()
    if bf16:
        buf = buf.to(torch.bfloat16)
    return buf, L.View(buf.data_ptr(), n, h, w, c, cs, coff, L.BF16 if bf16 else L.F32)


def empty_view(n, h, w, c, bf16, cstride=None, coff=0):
    cs = c if cstride is None else cstride
    buf = torch.full((n, h, w, cs), float("nan"), dtype=torch.bfloat16 if bf16 else torch.float32, device="cuda")
    return buf, L.View(buf.data_ptr(), n, h, w, c, cs, coff, L.BF16 if bf16 else L.F32)


def from_view(buf, c, coff=0):
    return buf[..., coff:coff + c].to(torch.float32).permute(0, 3, 1, 2).contiguous().cpu().numpy()


# (transposed, cin, cout, k, stride, pad, (n, h, w), in_cstride): the generator trunk + heads
CASES = [
    (0, 3, 16, 5, 1, 2, (2, 21, 37), 4),            # p_y_z_in.0 (the 3(+1)-channel stem; CC = 4, 8 taps per MFMA)
    (0, 16, 32, 4, 2, 1, (2, 22, 38), None),        # p_y_z_in.3  (CC = 16, tap pairs in the stride planes)
    (0, 32, 64, 4, 2, 1, (2, 14, 22), None),
    (0, 64, 128, 4, 2, 1, (2, 10, 38), None),
    (0, 128, 128, 3, 1, 1, (2, 13, 35), None),      # residual trunk
    (0, 128, 128, 3, 1, 1, (2, 9, 16), None),       # ... at widths the weights-stationary kernel takes (conv_bf16_ws.hip:
    (0, 128, 128, 3, 1, 1, (3, 7, 32), None),       #     bf16 on both sides; the other element types stay on the tiled one)
    (0, 128, 128, 3, 1, 1, (2, 21, 64), None),
    (1, 128, 64, 4, 2, 1, (2, 7, 18), None),        # decoder
    (1, 64, 32, 4, 2, 1, (3, 5, 9), None),
    (1, 32, 16, 4, 2, 1, (2, 9, 21), None),
    (0, 16, 8, 7, 1, 3, (2, 19, 35), None),         # p_mu_out.0 (data gradient gathers 8 channels: CC = 8)
    (0, 16, 8, 7, 1, 3, (3, 37, 150), None),        # ... several 64 x 16 tiles of the flattened-K kernel (conv_bf16_flat.hip:
    #                                                 bf16 -> fp32 / bf16 forward, fp32 / bf16 -> bf16 data gradient), ragged both ways
    (0, 16, 8, 7, 1, 3, (66, 19, 130), None),       # ... and enough 64-column strips for the row-walking forward (flatr_k7_kernel):
    #                                                 ragged last strip, a height that is not a multiple of its 8-row step
    # (p_mu_out.2, 8 -> 1 k5, exists with one element-type pattern only: test_head_tail_layer_on_matrix_cores)
]


@pytest.mark.parametrize("relu", [False, True], ids=["leaky", "relu"])
@pytest.mark.parametrize("io", [(True, True), (False, True), (True, False), (False, False)],
                         ids=["bf16-bf16", "f32-bf16", "bf16-f32", "f32-f32"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "%s%d_%d_k%ds%d_%dx%d" % ("T" if c[0] else "C", *c[1:5], *c[6][1:]))
def test_bf16_convolution_forward_dgrad_wgrad(case, io, relu):
    lib = L.load()
    tr, ci, co, k, s, p, (n, h, w), in_cs = case
    in_bf, out_bf = io
    rng = np.random.default_rng(ci * 11 + co + k + 2 * in_bf + out_bf)
    x = rng.standard_normal((n, ci, h, w)).astype(np.float32)
    wt = (rng.standard_normal(((ci, co) if tr else (co, ci)) + (k, k)) * 0.1).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, ci).astype(np.float32)
    shift = rng.uniform(0.2, 0.6, ci).astype(np.float32)          # act(0) != 0: padding must stay 0
    slope = rng.uniform(0.0, 0.3, ci).astype(np.float32)
    if relu:
        slope[:] = 0.0                 # batch-norm + ReLU: the staging code's fma + max path (every trunk layer)
    xin = bf16_round(x) if in_bf else x
    # the kernel evaluates fmaf(x, scale, shift) in fp32: product exact in float64, one rounding
    t = (xin.astype(np.float64) * scale[None, :, None, None].astype(np.float64)
         + shift[None, :, None, None].astype(np.float64)).astype(np.float32)
    xa = bf16_round(np.where(t > 0, t, t * slope[None, :, None, None]).astype(np.float32)).astype(np.float64)
    w64 = bf16_round(wt).astype(np.float64)
    y_ref = ops.convT2d_fwd(xa, w64, s, p, 0) if tr else ops.conv2d_fwd(xa, w64, s, p)
    _, _, ho, wo = y_ref.shape
    cv = L.Conv(tr, ci, co, k, s, p, 0)
    st = G.stream()
    xb, xv = to_view(xin, in_bf, cstride=in_cs)
    yb, yv = empty_view(n, ho, wo, co, out_bf, cstride=co + 8, coff=8)
    assert lib.bp_conv_bf16_supported(C.byref(cv), L.PACK_FWD, C.byref(xv), C.byref(yv)) == 1
    keep, pw = G.pointwise(scale, shift, slope)
    wd = G.dev(wt)
    pf = torch.zeros(lib.bp_conv_bf16_packed_elems(C.byref(cv), L.PACK_FWD), device="cuda", dtype=torch.bfloat16)
    pb = torch.zeros(lib.bp_conv_bf16_packed_elems(C.byref(cv), L.PACK_BWD), device="cuda", dtype=torch.bfloat16)
    L.check(lib.bp_conv_bf16_pack(C.byref(cv), L.PACK_FWD, L.ptr(wd), L.ptr(pf), st))
    L.check(lib.bp_conv_bf16_pack(C.byref(cv), L.PACK_BWD, L.ptr(wd), L.ptr(pb), st))
    L.check(lib.bp_conv_forward(C.byref(cv), C.byref(xv), C.byref(pw), L.ptr(pf), L.ptr(wd), None, C.byref(yv),
                                L.IMPL_BF16, st), "forward")
    tol = 4e-3 if out_bf else 2e-5
    got = from_view(yb, co, coff=8)
    assert G.rel_err(got, y_ref) < tol, "forward"
    raw = yb.to(torch.float32)
    assert torch.isnan(raw[..., :8]).all(), "stores outside the view"

    # ---- the same forward with the batch-norm sums taken in the epilogue: the sums of the tensor AS STORED
    nb = lib.bp_conv_stats_workspace(C.byref(cv), L.PACK_FWD, C.byref(xv), C.byref(yv), L.IMPL_BF16)
    if co & (co - 1) == 0:
        assert nb > 0
        yb2, yv2 = empty_view(n, ho, wo, co, out_bf, cstride=co + 8, coff=8)
        sums = torch.full((2 * co,), float("nan"), dtype=torch.float64, device="cuda")
        wss = torch.full((nb // 8 + 8,), float("nan"), dtype=torch.float64, device="cuda")
        L.check(lib.bp_conv_forward_stats(C.byref(cv), C.byref(xv), C.byref(pw), L.ptr(pf), C.byref(yv2), L.ptr(sums),
                                          L.ptr(wss), nb, L.IMPL_BF16, st), "forward + statistics")
        got2 = from_view(yb2, co, coff=8)
        assert np.array_equal(got2, got), "the statistics epilogue must not change the output"
        g64 = got.astype(np.float64)
        sm = sums.cpu().numpy()
        # (a lane's <= 256 terms are summed in fp32 before the double reductions: 1e-6 of the sum of magnitudes)
        assert (np.abs(sm[:co] - g64.sum(axis=(0, 2, 3))) <= 1e-6 * np.abs(g64).sum(axis=(0, 2, 3)) + 1e-12).all()
        assert np.allclose(sm[co:], (g64 ** 2).sum(axis=(0, 2, 3)), rtol=1e-6, atol=1e-12)
    else:
        assert nb == 0

    # ---- data gradient (no pending activation on dy)
    dy = rng.standard_normal(y_ref.shape).astype(np.float32)
    dyin = bf16_round(dy) if in_bf else dy
    dy64 = bf16_round(dyin).astype(np.float64)
    dyb, dyv = to_view(dyin, in_bf)
    dxb, dxv = empty_view(n, h, w, ci, out_bf and ci % 8 == 0, cstride=None if ci != 3 else 4)
    L.check(lib.bp_conv_backward_data(C.byref(cv), C.byref(dyv), L.ptr(pb), L.ptr(wd), C.byref(dxv), L.IMPL_BF16, st),
            "backward_data")
    dx_ref = ops.convT2d_bwd_data(dy64, w64, s, p) if tr else ops.conv2d_bwd_data(dy64, w64, s, p, h, w)
    assert G.rel_err(from_view(dxb, ci), dx_ref) < (4e-3 if dxv.dtype == L.BF16 else 2e-5), "backward_data"

    # ---- weight gradient: fp32 result from bf16 products
    ws_bytes = lib.bp_conv_backward_weight_workspace(C.byref(cv), C.byref(xv), C.byref(dyv))
    assert ws_bytes > 0
    ws = torch.zeros(ws_bytes // 8 + 8, dtype=torch.float64, device="cuda")
    dw = torch.full(wt.shape, float("nan"), device="cuda")
    L.check(lib.bp_conv_backward_weight(C.byref(cv), C.byref(xv), C.byref(pw), C.byref(dyv), L.ptr(dw), None,
                                        L.ptr(ws), ws.numel() * 8, L.IMPL_BF16, st), "backward_weight")
    dw_ref = ops.convT2d_bwd_weight(xa, dy64, s, p, k, k) if tr else ops.conv2d_bwd_weight(xa, dy64, s, p, k, k)
    assert G.rel_err(dw.cpu().numpy(), dw_ref) < 1e-4, "backward_weight"


# (cin, cout, k, stride, pad, dy is bf16): Conv2d layers whose DATA GRADIENT runs on a flattened-K kernel that can take
# the producer's batch-norm backward sums in its epilogue (conv_bf16_flat.hip STATS == 2)
BWD_STATS_CASES = [(0, 16, 8, 7, 1, 3, False), (0, 16, 32, 4, 2, 1, True), (0, 32, 64, 4, 2, 1, True), (1, 32, 16, 4, 2, 1, True)]


@pytest.mark.parametrize("shape", [(2, 22, 38), (3, 64, 130)], ids=["ragged", "tiles"])
@pytest.mark.parametrize("case", BWD_STATS_CASES, ids=lambda c: "%s%d_%d_k%ds%d" % ("T" if c[0] else "C", *c[1:5]))
def test_bf16_data_gradient_with_activation_sums(case, shape):
    """bp_conv_backward_data_stats on bf16 views: dx as bp_conv_backward_data writes it (bit for bit) and
    {sum g, sum g*raw}, g = dx * act'(pw(raw)), equal to what bp_act_backward computes from the stored bf16 dx."""
    lib = L.load()
    tr, ci, co, k, s, p, dy_bf = case
    n, h, w = shape
    if tr:
        h, w = h // 2, w // 2                      # (the transposed layer's input is the coarse grid)
    rng = np.random.default_rng(ci + co + k)
    wt = (rng.standard_normal(((ci, co) if tr else (co, ci)) + (k, k)) * 0.1).astype(np.float32)
    ho, wo = ((h - 1) * s - 2 * p + k, (w - 1) * s - 2 * p + k) if tr else ((h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1)
    dy = rng.standard_normal((n, co, ho, wo)).astype(np.float32)
    raw = bf16_round(rng.standard_normal((n, ci, h, w)).astype(np.float32))
    scale = rng.uniform(0.5, 1.5, ci).astype(np.float32)
    shift = rng.uniform(-0.4, 0.4, ci).astype(np.float32)
    slope = rng.uniform(0.0, 0.3, ci).astype(np.float32)
    slope[::3] = 0.0
    cv = L.Conv(tr, ci, co, k, s, p, 0)
    st = G.stream()
    dyb, dyv = to_view(bf16_round(dy) if dy_bf else dy, dy_bf)
    rb, rv = to_view(raw, True, cstride=ci + 8, coff=8 if ci >= 32 else 4)
    keep, pw = G.pointwise(scale, shift, slope)
    wd = G.dev(wt)
    pb = torch.zeros(lib.bp_conv_bf16_packed_elems(C.byref(cv), L.PACK_BWD), device="cuda", dtype=torch.bfloat16)
    L.check(lib.bp_conv_bf16_pack(C.byref(cv), L.PACK_BWD, L.ptr(wd), L.ptr(pb), st))
    dxb, dxv = empty_view(n, h, w, ci, True)
    L.check(lib.bp_conv_backward_data(C.byref(cv), C.byref(dyv), L.ptr(pb), L.ptr(wd), C.byref(dxv), L.IMPL_BF16, st))
    nb = lib.bp_conv_stats_workspace(C.byref(cv), L.PACK_BWD, C.byref(dxv), C.byref(dyv), L.IMPL_BF16)
    assert nb > 0
    dx2b, dx2v = empty_view(n, h, w, ci, True)
    sums = torch.full((2 * ci,), float("nan"), dtype=torch.float64, device="cuda")
    ws = torch.full((nb // 8 + 8,), float("nan"), dtype=torch.float64, device="cuda")
    L.check(lib.bp_conv_backward_data_stats(C.byref(cv), C.byref(dyv), L.ptr(pb), C.byref(dx2v), C.byref(rv), C.byref(pw),
                                            L.ptr(sums), L.ptr(ws), nb, st), "backward_data + activation sums")
    assert torch.equal(dx2b, dxb), "the epilogue must not change the data gradient"
    dx = from_view(dxb, ci).astype(np.float64)
    t = (raw.astype(np.float64) * scale[None, :, None, None] + shift[None, :, None, None]).astype(np.float32)
    g = np.where(t > 0, dx, dx * slope[None, :, None, None].astype(np.float64))
    got = sums.cpu().numpy()
    r64 = raw.astype(np.float64)
    assert (np.abs(got[:ci] - g.sum(axis=(0, 2, 3))) <= 2e-6 * np.abs(g).sum(axis=(0, 2, 3)) + 1e-12).all()
    assert (np.abs(got[ci:] - (g * r64).sum(axis=(0, 2, 3))) <= 2e-6 * np.abs(g * r64).sum(axis=(0, 2, 3)) + 1e-12).all()
    # ... and the separate pass over the stored gradient
    ref = torch.zeros(3 * ci, dtype=torch.float64, device="cuda")
    r2b, r2v = to_view(raw, True)               # (the streaming pass wants views of one geometry)
    nb2 = lib.bp_act_backward_workspace(C.byref(r2v))
    ws2 = torch.zeros(nb2 // 8 + 8, dtype=torch.float64, device="cuda")
    L.check(lib.bp_act_backward(C.byref(dxv), None, C.byref(r2v), C.byref(pw), None, None, L.ptr(ref), L.ptr(ws2), nb2, st))
    ref = ref.cpu().numpy()
    assert (np.abs(got[:ci] - ref[:ci]) <= 2e-6 * np.abs(g).sum(axis=(0, 2, 3)) + 1e-12).all()
    assert (np.abs(got[ci:] - ref[ci:2 * ci]) <= 2e-6 * np.abs(g * r64).sum(axis=(0, 2, 3)) + 1e-12).all()
    assert lib.bp_conv_backward_data_stats(C.byref(cv), C.byref(dyv), L.ptr(pb), C.byref(dx2v), C.byref(rv), C.byref(pw),
                                           L.ptr(sums), L.ptr(ws), nb - 8, st) == -4
    # the generic bf16 kernels have no such epilogue
    cv2 = L.Conv(0, 128, 128, 3, 1, 1, 0)
    a_b, a_v = empty_view(1, 8, 8, 128, True)
    b_b, b_v = empty_view(1, 8, 8, 128, True)
    assert lib.bp_conv_stats_workspace(C.byref(cv2), L.PACK_BWD, C.byref(a_v), C.byref(b_v), L.IMPL_BF16) == 0


def test_fp32_entry_points_refuse_bf16_views():
    lib = L.load()
    buf, v = empty_view(1, 4, 4, 8, True)
    sums = torch.zeros(16, dtype=torch.float64, device="cuda")
    ws = torch.zeros(4096, dtype=torch.float64, device="cuda")
    cv = L.Conv(0, 8, 8, 3, 1, 1, 0)
    out, ov = empty_view(1, 4, 4, 8, False)
    pf = torch.zeros(lib.bp_conv_packed_floats(C.byref(cv), L.PACK_FWD), device="cuda")
    assert lib.bp_conv_forward(C.byref(cv), C.byref(v), None, L.ptr(pf), None, None, C.byref(ov), L.IMPL_MFMA,
                               G.stream()) == -1            # BP_EINVAL: the fp32 kernels take fp32 views only


# ---- the weights-stationary kernels (csrc/conv_bf16_ws.hip) against the tiled kernels they replace
@pytest.mark.parametrize("shape", [(2, 21, 37), (3, 37, 150), (1, 64, 128), (5, 130, 200)], ids=["ragged", "tiles3", "whole", "fold"])
def test_head_tail_layer_on_matrix_cores(shape):
    """The heads' 8 -> 1 k5 layer between the 8-channel bf16 slot and the fp32 one-channel tail (throughput-mode policy of
    models/cvae.py): forward (conv_bf16_flat.hip kind 9), data gradient with the producer's PReLU backward in the epilogue
    (conv_bf16_head.hip: g as bf16 and the three sums of bp_act_backward) and weight gradient (wgrad_head_kernel), against
    the float64 oracle on the same bf16-rounded operands; views that are channel slices of wider buffers."""
    lib = L.load()
    n, h, w = shape
    ci, co, k, p = 8, 1, 5, 2
    rng = np.random.default_rng(h * 7 + w)
    raw = bf16_round(rng.standard_normal((n, ci, h, w)).astype(np.float32))
    wt = (rng.standard_normal((co, ci, k, k)) * 0.1).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, ci).astype(np.float32)
    shift = rng.uniform(-0.4, 0.4, ci).astype(np.float32)
    slope = rng.uniform(0.05, 0.3, ci).astype(np.float32)
    dy = rng.standard_normal((n, co, h, w)).astype(np.float32)
    t = (raw.astype(np.float64) * scale[None, :, None, None].astype(np.float64)
         + shift[None, :, None, None].astype(np.float64)).astype(np.float32)
    xa = bf16_round(np.where(t > 0, t, t * slope[None, :, None, None]).astype(np.float32)).astype(np.float64)
    w64 = bf16_round(wt).astype(np.float64)
    dy64 = bf16_round(dy).astype(np.float64)
    cv = L.Conv(0, ci, co, k, 1, p, 0)
    st = G.stream()
    rb, rv = to_view(raw, True, cstride=ci + 8, coff=8)
    dyb, dyv = to_view(dy, False, cstride=3, coff=1)
    keep, pw = G.pointwise(scale, shift, slope)
    wd = G.dev(wt)
    pf = torch.zeros(lib.bp_conv_bf16_packed_elems(C.byref(cv), L.PACK_FWD), device="cuda", dtype=torch.bfloat16)
    pb = torch.zeros(lib.bp_conv_bf16_packed_elems(C.byref(cv), L.PACK_BWD), device="cuda", dtype=torch.bfloat16)
    L.check(lib.bp_conv_bf16_pack(C.byref(cv), L.PACK_FWD, L.ptr(wd), L.ptr(pf), st))
    L.check(lib.bp_conv_bf16_pack(C.byref(cv), L.PACK_BWD, L.ptr(wd), L.ptr(pb), st))
    # ---- forward
    yb, yv = empty_view(n, h, w, co, False, cstride=3, coff=2)
    L.check(lib.bp_conv_forward(C.byref(cv), C.byref(rv), C.byref(pw), L.ptr(pf), L.ptr(wd), None, C.byref(yv),
                                L.IMPL_BF16, st), "forward")
    assert G.rel_err(from_view(yb, co, coff=2), ops.conv2d_fwd(xa, w64, 1, p)) < 2e-5
    assert torch.isnan(yb[..., :2]).all(), "stores outside the view"
    # ---- data gradient + PReLU backward of the slot
    d_ref = ops.conv2d_bwd_data(dy64, w64, 1, p, h, w)
    g_ref = np.where(t > 0, d_ref, d_ref * slope[None, :, None, None].astype(np.float64))
    gb, gv = empty_view(n, h, w, ci, True, cstride=ci + 8, coff=8)
    nb = lib.bp_conv_backward_data_act_workspace(C.byref(cv), C.byref(dyv), C.byref(gv))
    assert nb > 0
    sums = torch.full((3 * ci,), float("nan"), dtype=torch.float64, device="cuda")
    ws = torch.full((nb // 8 + 8,), float("nan"), dtype=torch.float64, device="cuda")
    L.check(lib.bp_conv_backward_data_act(C.byref(cv), C.byref(dyv), L.ptr(pb), C.byref(gv), C.byref(rv), C.byref(pw),
                                          L.ptr(sums), L.ptr(ws), nb, st), "backward_data + activation backward")
    assert G.rel_err(from_view(gb, ci, coff=8), g_ref) < 4e-3
    assert torch.isnan(gb.to(torch.float32)[..., :8]).all(), "stores outside the view"
    got = sums.cpu().numpy()
    r64 = raw.astype(np.float64)
    refs = [g_ref, g_ref * r64, np.where(t > 0, 0.0, d_ref * t.astype(np.float64))]
    for q in range(3):
        ref_q, mag_q = refs[q].sum(axis=(0, 2, 3)), np.abs(refs[q]).sum(axis=(0, 2, 3))
        assert (np.abs(got[q * ci:(q + 1) * ci] - ref_q) <= 2e-5 * mag_q + 1e-12).all(), q
    assert lib.bp_conv_backward_data_act(C.byref(cv), C.byref(dyv), L.ptr(pb), C.byref(gv), C.byref(rv), C.byref(pw),
                                         L.ptr(sums), L.ptr(ws), nb - 8, st) == -4
    # ... and without the epilogue: d itself
    dxb, dxv = empty_view(n, h, w, ci, True)
    L.check(lib.bp_conv_backward_data(C.byref(cv), C.byref(dyv), L.ptr(pb), L.ptr(wd), C.byref(dxv), L.IMPL_BF16, st))
    assert G.rel_err(from_view(dxb, ci), d_ref) < 4e-3
    # ---- weight gradient
    ws_bytes = lib.bp_conv_backward_weight_workspace(C.byref(cv), C.byref(rv), C.byref(dyv))
    assert ws_bytes > 0
    wsw = torch.zeros(ws_bytes // 8 + 8, dtype=torch.float64, device="cuda")
    dw = torch.full(wt.shape, float("nan"), device="cuda")
    L.check(lib.bp_conv_backward_weight(C.byref(cv), C.byref(rv), C.byref(pw), C.byref(dyv), L.ptr(dw), None,
                                        L.ptr(wsw), wsw.numel() * 8, L.IMPL_BF16, st), "backward_weight")
    assert G.rel_err(dw.cpu().numpy(), ops.conv2d_bwd_weight(xa, dy64, 1, p, k, k)) < 1e-4


WS_LAYERS = {
    # (transposed, cin, cout, k, stride, pad), input shapes (n, h, w) the stationary kernel takes in at least one direction
    "k3": ((0, 128, 128, 3, 1, 1), [(2, 9, 16), (3, 20, 32), (1, 64, 64), (5, 16, 64), (70, 8, 16)]),
    # forward: the strided gather (kind 4), data gradient: the transposed gather (kind 5)
    "k4s2": ((0, 64, 128, 4, 2, 1), [(2, 16, 32), (3, 12, 64), (2, 40, 128), (66, 8, 32)]),
    # forward: the transposed gather, data gradient: the strided one
    "t4s2": ((1, 128, 64, 4, 2, 1), [(2, 7, 16), (2, 9, 32), (3, 20, 64), (40, 5, 16)]),
}


@pytest.mark.parametrize("act", ["none", "relu", "leaky"])
@pytest.mark.parametrize("layer,shape", [(k, sh) for k, (_, shs) in WS_LAYERS.items() for sh in shs],
                         ids=lambda v: v if isinstance(v, str) else "%dx%dx%d" % v)
def test_weights_stationary_kernels(layer, shape, act):
    """Forward (with and without the batch-norm sums) and data gradient: each kernel within 2^-8 of the float64
    convolution of the same bf16 operands, the two kernels' sums equal to 1e-6 of the sum of magnitudes; views that are
    channel slices of wider buffers; bands that end inside the image (n * h chosen so that several band sizes occur)."""
    lib = L.load()
    n, h, w = shape
    tr, ci, co, k, st_, p = WS_LAYERS[layer][0]
    rng = np.random.default_rng(n * 1000 + h * 10 + w + k)
    x = bf16_round(rng.standard_normal((n, ci, h, w)).astype(np.float32))
    wt = (rng.standard_normal(((ci, co) if tr else (co, ci)) + (k, k)) * 0.05).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, ci).astype(np.float32)
    shift = rng.uniform(0.2, 0.6, ci).astype(np.float32)
    slope = rng.uniform(0.0, 0.3, ci).astype(np.float32)
    if act == "relu":
        slope[:] = 0.0
    if act == "none":
        xa = x.astype(np.float64)
    else:
        t = (x.astype(np.float64) * scale[None, :, None, None].astype(np.float64)
             + shift[None, :, None, None].astype(np.float64)).astype(np.float32)
        xa = bf16_round(np.where(t > 0, t, t * slope[None, :, None, None]).astype(np.float32)).astype(np.float64)
    w64 = bf16_round(wt).astype(np.float64)
    y_ref = ops.convT2d_fwd(xa, w64, st_, p, 0) if tr else ops.conv2d_fwd(xa, w64, st_, p)
    _, _, ho, wo = y_ref.shape
    cv = L.Conv(tr, ci, co, k, st_, p, 0)
    st = G.stream()
    xb, xv = to_view(x, True, cstride=ci + 16, coff=8)
    keep, pw = G.pointwise(scale, shift, slope)
    pwp = None if act == "none" else C.byref(pw)
    wd = G.dev(wt)
    pf = torch.zeros(lib.bp_conv_bf16_packed_elems(C.byref(cv), L.PACK_FWD), device="cuda", dtype=torch.bfloat16)
    pb = torch.zeros(lib.bp_conv_bf16_packed_elems(C.byref(cv), L.PACK_BWD), device="cuda", dtype=torch.bfloat16)
    L.check(lib.bp_conv_bf16_pack(C.byref(cv), L.PACK_FWD, L.ptr(wd), L.ptr(pf), st))
    L.check(lib.bp_conv_bf16_pack(C.byref(cv), L.PACK_BWD, L.ptr(wd), L.ptr(pb), st))
    dy = bf16_round(rng.standard_normal(y_ref.shape).astype(np.float32))
    dyb, dyv = to_view(dy, True)
    dx_ref = ops.convT2d_bwd_data(dy.astype(np.float64), w64, st_, p) if tr else \
        ops.conv2d_bwd_data(dy.astype(np.float64), w64, st_, p, h, w)
    res = {}
    try:
        for ws_on in (1, 0):
            assert lib.bp_set_option(b"bf16_ws", ws_on) == 0
            yb, yv = empty_view(n, ho, wo, co, True, cstride=co + 8, coff=8)
            L.check(lib.bp_conv_forward(C.byref(cv), C.byref(xv), pwp, L.ptr(pf), L.ptr(wd), None, C.byref(yv),
                                        L.IMPL_BF16, st), "forward")
            got = from_view(yb, co, coff=8)
            assert G.rel_err(got, y_ref) < 4e-3, f"forward ws={ws_on}"
            assert torch.isnan(yb.to(torch.float32)[..., :8]).all(), "stores outside the view"
            nb = lib.bp_conv_stats_workspace(C.byref(cv), L.PACK_FWD, C.byref(xv), C.byref(yv), L.IMPL_BF16)
            assert nb > 0
            yb2, yv2 = empty_view(n, ho, wo, co, True, cstride=co + 8, coff=8)
            sums = torch.full((2 * co,), float("nan"), dtype=torch.float64, device="cuda")
            wss = torch.full((nb // 8 + 8,), float("nan"), dtype=torch.float64, device="cuda")
            L.check(lib.bp_conv_forward_stats(C.byref(cv), C.byref(xv), pwp, L.ptr(pf), C.byref(yv2), L.ptr(sums),
                                              L.ptr(wss), nb, L.IMPL_BF16, st), "forward + statistics")
            assert np.array_equal(from_view(yb2, co, coff=8), got), "the statistics epilogue must not change the output"
            g64 = got.astype(np.float64)
            sm = sums.cpu().numpy()
            assert (np.abs(sm[:co] - g64.sum(axis=(0, 2, 3))) <= 1e-6 * np.abs(g64).sum(axis=(0, 2, 3)) + 1e-12).all()
            assert np.allclose(sm[co:], (g64 ** 2).sum(axis=(0, 2, 3)), rtol=1e-6, atol=1e-12)
            dxb, dxv = empty_view(n, h, w, ci, True, cstride=ci + 8, coff=0)
            L.check(lib.bp_conv_backward_data(C.byref(cv), C.byref(dyv), L.ptr(pb), L.ptr(wd), C.byref(dxv), L.IMPL_BF16, st),
                    "backward_data")
            dx = from_view(dxb, ci)
            assert G.rel_err(dx, dx_ref) < 4e-3, f"backward_data ws={ws_on}"
            assert torch.isnan(dxb.to(torch.float32)[..., ci:]).all(), "stores outside the view"
            res[ws_on] = (got, dx)
    finally:
        lib.bp_set_option(b"bf16_ws", -1)
    # the two kernels differ by accumulation order only: a bf16 ulp (2^-8 of the value, at most 2^-7 of the maximum) here and there
    assert G.rel_err(res[1][0], res[0][0]) < 8e-3 and G.rel_err(res[1][1], res[0][1]) < 8e-3
    assert np.mean(res[1][0] != res[0][0]) < 0.05 and np.mean(res[1][1] != res[0][1]) < 0.05
    # ... and the stationary kernel did run where it should (otherwise the two results are bit-equal)
    assert not np.array_equal(res[1][0], res[0][0]), "the forward never reached the weights-stationary kernel"
    assert not np.array_equal(res[1][1], res[0][1]), "the data gradient never reached the weights-stationary kernel"


@pytest.mark.parametrize("ws_on", [1, 0], ids=["stationary", "tiled"])
def test_bf16_relu_staging_propagates_nan(ws_on):
    """torch.relu(NaN) is NaN (the reference's ReLU, utils.py:140): a NaN activation must reach every output its
    3 x 3 footprint covers, through the batch-norm + ReLU staging of the forward kernels and of the weight gradient --
    fmaxf(NaN, 0) would turn it into 0 and a diverged run would keep painting finite tiles."""
    lib = L.load()
    n, h, w, ci, co = 2, 8, 16, 128, 128
    rng = np.random.default_rng(5)
    x = bf16_round(rng.standard_normal((n, ci, h, w)).astype(np.float32))
    x[1, 37, 4, 9] = np.nan
    wt = (rng.standard_normal((co, ci, 3, 3)) * 0.05).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, ci).astype(np.float32)
    shift = rng.uniform(0.2, 0.6, ci).astype(np.float32)
    slope = np.zeros(ci, np.float32)
    cv = L.Conv(0, ci, co, 3, 1, 1, 0)
    st = G.stream()
    xb, xv = to_view(x, True)
    keep, pw = G.pointwise(scale, shift, slope)
    wd = G.dev(wt)
    pf = torch.zeros(lib.bp_conv_bf16_packed_elems(C.byref(cv), L.PACK_FWD), device="cuda", dtype=torch.bfloat16)
    L.check(lib.bp_conv_bf16_pack(C.byref(cv), L.PACK_FWD, L.ptr(wd), L.ptr(pf), st))
    try:
        assert lib.bp_set_option(b"bf16_ws", ws_on) == 0
        yb, yv = empty_view(n, h, w, co, True)
        L.check(lib.bp_conv_forward(C.byref(cv), C.byref(xv), C.byref(pw), L.ptr(pf), L.ptr(wd), None, C.byref(yv),
                                    L.IMPL_BF16, st), "forward")
    finally:
        lib.bp_set_option(b"bf16_ws", -1)
    y = from_view(yb, co)
    bad = np.isnan(y)
    assert bad[1, :, 3:6, 8:11].all() and bad.sum() == co * 9
    dy = bf16_round(rng.standard_normal(y.shape).astype(np.float32))
    dyb, dyv = to_view(dy, True)
    ws_bytes = lib.bp_conv_backward_weight_workspace(C.byref(cv), C.byref(xv), C.byref(dyv))
    ws = torch.zeros(ws_bytes // 8 + 8, dtype=torch.float64, device="cuda")
    dw = torch.zeros(wt.shape, device="cuda")
    L.check(lib.bp_conv_backward_weight(C.byref(cv), C.byref(xv), C.byref(pw), C.byref(dyv), L.ptr(dw), None,
                                        L.ptr(ws), ws.numel() * 8, L.IMPL_BF16, st), "backward_weight")
    dwn = torch.isnan(dw).cpu().numpy()
    assert dwn[:, 37].all() and not np.delete(dwn, 37, axis=1).any()


@pytest.mark.parametrize("act", ["none", "relu", "leaky"])
@pytest.mark.parametrize("shape", [(2, 9, 32), (2, 64, 64), (3, 21, 64), (66, 8, 32)], ids=lambda s: "%dx%dx%d" % s)
def test_output_stationary_bf16_trunk_weight_gradient(shape, act):
    """bp_conv_backward_weight (impl BP_IMPL_BF16) of Conv2d(128, 128, 3, 1, 1) on bf16 views through the output-stationary
    kernel (csrc/conv_wgrad_ws_bf16.hip) and through the tiled one (bp_set_option("bf16_wgrad_ws", 1 / 0)): exact bf16
    products, fp32 accumulation -- both within 1e-4 of the float64 correlation of the same bf16-rounded operands (activation
    applied in fp32, then rounded, as the staging does); views that are channel slices; a NaN in X reaches exactly the
    gradients of its input channel."""
    lib = L.load()
    n, h, w = shape
    ci = co = 128
    rng = np.random.default_rng(n + 7 * h + w)
    x = bf16_round(rng.standard_normal((n, ci, h, w)).astype(np.float32))
    dy = bf16_round(rng.standard_normal((n, co, h, w)).astype(np.float32))
    scale = rng.uniform(0.5, 1.5, ci).astype(np.float32)
    shift = rng.uniform(-0.3, 0.6, ci).astype(np.float32)
    slope = rng.uniform(0.0, 0.3, ci).astype(np.float32)
    if act == "relu":
        slope[:] = 0.0
    if act == "none":
        xa = x.astype(np.float64)
    else:
        t = (x.astype(np.float64) * scale[None, :, None, None].astype(np.float64)
             + shift[None, :, None, None].astype(np.float64)).astype(np.float32)
        xa = bf16_round(np.where(t > 0, t, t * slope[None, :, None, None]).astype(np.float32)).astype(np.float64)
    dw_ref = ops.conv2d_bwd_weight(xa, dy.astype(np.float64), 1, 1, 3, 3)
    cv = L.Conv(0, ci, co, 3, 1, 1, 0)
    st = G.stream()
    xb, xv = to_view(x, True, cstride=ci + 16, coff=8)
    dyb, dyv = to_view(dy, True, cstride=co + 8, coff=0)
    keep, pw = G.pointwise(scale, shift, slope)
    pwp = None if act == "none" else C.byref(pw)
    res = {}
    try:
        for on in (1, 0):
            assert lib.bp_set_option(b"bf16_wgrad_ws", on) == 0
            nb = lib.bp_conv_backward_weight_workspace(C.byref(cv), C.byref(xv), C.byref(dyv))
            assert nb > 0
            ws = torch.full((nb // 8 + 8,), float("nan"), dtype=torch.float64, device="cuda")
            dw = torch.full((co, ci, 3, 3), float("nan"), device="cuda")
            L.check(lib.bp_conv_backward_weight(C.byref(cv), C.byref(xv), pwp, C.byref(dyv), L.ptr(dw), None, L.ptr(ws),
                                                ws.numel() * 8, L.IMPL_BF16, st), "backward_weight")
            got = dw.cpu().numpy()
            assert G.rel_err(got, dw_ref) < 1e-4, f"ws={on}"
            res[on] = got
        assert not np.array_equal(res[1], res[0]), "the output-stationary kernel never ran (both results bit-equal)"
        assert lib.bp_set_option(b"bf16_wgrad_ws", 1) == 0
        x2 = x.copy()
        x2[n - 1, 77, h // 2, w // 2] = np.nan
        xb2, xv2 = to_view(x2, True, cstride=ci + 16, coff=8)
        dw = torch.zeros((co, ci, 3, 3), device="cuda")
        L.check(lib.bp_conv_backward_weight(C.byref(cv), C.byref(xv2), pwp, C.byref(dyv), L.ptr(dw), None, L.ptr(ws),
                                            ws.numel() * 8, L.IMPL_BF16, st), "backward_weight")
        bad = torch.isnan(dw).cpu().numpy()
        assert bad[:, 77].all() and not np.delete(bad, 77, axis=1).any()
    finally:
        lib.bp_set_option(b"bf16_wgrad_ws", -1)
