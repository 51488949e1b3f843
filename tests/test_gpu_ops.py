"""GPU parity of every operator entry point of the C ABI against the NumPy oracle and the
golden fixtures generated from the reference (tests/golden/ops.npz).

Tolerance: fp32 matrix-core FMA chains vs float64 oracle, relative to each tensor's scale:
2e-5 for forward/data-gradient, 1e-4 for weight gradients (longer reductions)."""
import ctypes as C

import numpy as np
import pytest
import torch

from baryon_painter_amd import _lib as L
from baryon_painter_amd.utils import synthetic as syn
from golden.make_goldens_cases import OP_CASES
from golden_util import check
from oracle import ops

import gpu_util as G

pytestmark = pytest.mark.gpu


def _case(idx):
    name, kind, cfg, (n, h, w) = OP_CASES[idx]
    cfg = dict(cfg)
    cfg.setdefault("bias", False)
    tr = kind == "transp conv"
    wshape = ((cfg["in_channels"], cfg["out_channels"]) if tr else (cfg["out_channels"], cfg["in_channels"])) \
        + (cfg["kernel_size"],) * 2
    shapes = {"0.weight": wshape}
    if cfg["bias"]:
        shapes["0.bias"] = (cfg["out_channels"],)
    P = syn.fill_params(shapes, 100 + idx)
    x = syn.synthetic_eps((n, cfg["in_channels"], h, w), seed=200 + idx)
    cv = L.Conv(1 if tr else 0, cfg["in_channels"], cfg["out_channels"], cfg["kernel_size"], cfg["stride"],
                cfg["padding"], cfg.get("output_padding", 0))
    return name, cfg, tr, P, x, cv


@pytest.mark.parametrize("impl", [L.IMPL_DIRECT, L.IMPL_MFMA], ids=["direct", "mfma"])
@pytest.mark.parametrize("idx", range(len(OP_CASES)))
def test_conv_forward_backward(idx, impl, golden_ops):
    lib = L.load()
    name, cfg, tr, P, x, cv = _case(idx)
    w = P["0.weight"]
    bias = P.get("0.bias")
    s, p, op = cfg["stride"], cfg["padding"], cfg.get("output_padding", 0)
    x64, w64 = x.astype(np.float64), w.astype(np.float64)
    y_ref = ops.convT2d_fwd(x64, w64, s, p, op) if tr else ops.conv2d_fwd(x64, w64, s, p)
    if bias is not None:
        y_ref = y_ref + bias[None, :, None, None]
    n, co, ho, wo = y_ref.shape
    dy = syn.synthetic_eps(y_ref.shape, seed=300 + idx)
    st = G.stream()
    # odd channel strides/offsets on purpose when the channel count is small
    cs_in = x.shape[1] + (3 if x.shape[1] < 8 else 0)
    xb, xv = G.to_nhwc(x, cstride=cs_in, coff=(1 if x.shape[1] < 8 else 0))
    yb, yv = G.empty_nhwc(n, ho, wo, co)
    wd = G.dev(w)
    bd = None if bias is None else G.dev(bias)
    nf = lib.bp_conv_packed_floats(C.byref(cv), L.PACK_FWD)
    nb = lib.bp_conv_packed_floats(C.byref(cv), L.PACK_BWD)
    assert nf > 0 and nb > 0
    pf = torch.zeros(nf, device="cuda")
    pb = torch.zeros(nb, device="cuda")
    L.check(lib.bp_conv_pack(C.byref(cv), L.PACK_FWD, L.ptr(wd), L.ptr(pf), st))
    L.check(lib.bp_conv_pack(C.byref(cv), L.PACK_BWD, L.ptr(wd), L.ptr(pb), st))
    L.check(lib.bp_conv_forward(C.byref(cv), C.byref(xv), None, L.ptr(pf), L.ptr(wd), L.ptr(bd), C.byref(yv),
                                impl, st), "forward")
    y = G.from_nhwc(yb, co)
    assert G.rel_err(y, y_ref) < 2e-5
    check(f"{name}/y", y, golden_ops, 5e-5)

    # data gradient
    dyb, dyv = G.to_nhwc(dy)
    dxb, dxv = G.empty_nhwc(*[x.shape[i] for i in (0, 2, 3, 1)])
    L.check(lib.bp_conv_backward_data(C.byref(cv), C.byref(dyv), L.ptr(pb), L.ptr(wd), C.byref(dxv), impl, st),
            "backward_data")
    dx = G.from_nhwc(dxb, x.shape[1])
    dy64 = dy.astype(np.float64)
    dx_ref = ops.convT2d_bwd_data(dy64, w64, s, p) if tr else ops.conv2d_bwd_data(dy64, w64, s, p, *x.shape[2:])
    assert G.rel_err(dx, dx_ref) < 2e-5
    check(f"{name}/dx", dx, golden_ops, 5e-5)

    # weight (+bias) gradient
    ws_bytes = lib.bp_conv_backward_weight_workspace(C.byref(cv), C.byref(xv), C.byref(dyv))
    ws = torch.zeros(ws_bytes // 8 + 8, dtype=torch.float64, device="cuda")
    dw = torch.full(w.shape, float("nan"), device="cuda")
    db = None if bias is None else torch.full(bias.shape, float("nan"), device="cuda")
    L.check(lib.bp_conv_backward_weight(C.byref(cv), C.byref(xv), None, C.byref(dyv), L.ptr(dw), L.ptr(db),
                                        L.ptr(ws), ws.numel() * 8, impl, st), "backward_weight")
    k = cfg["kernel_size"]
    dw_ref = ops.convT2d_bwd_weight(x64, dy64, s, p, k, k) if tr else ops.conv2d_bwd_weight(x64, dy64, s, p, k, k)
    assert G.rel_err(dw.cpu().numpy(), dw_ref) < 1e-4
    check(f"{name}/d_0.weight", dw.cpu().numpy(), golden_ops, 2e-4)
    if bias is not None:
        check(f"{name}/d_0.bias", db.cpu().numpy(), golden_ops, 2e-4)


@pytest.mark.parametrize("impl", [L.IMPL_DIRECT, L.IMPL_MFMA], ids=["direct", "mfma"])
def test_conv_lazy_activation_and_zero_padding(impl):
    """The consumer applies the producer's affine+leaky-ReLU while loading; padding is zero in
    the ACTIVATED domain (torch pads after the activation)."""
    lib = L.load()
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 16, 9, 11)).astype(np.float32)
    w = (rng.standard_normal((32, 16, 3, 3)) * 0.2).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, 16).astype(np.float32)
    shift = rng.uniform(0.2, 0.6, 16).astype(np.float32)      # act(0) != 0: padding must stay 0
    slope = np.full(16, 0.1, np.float32)
    t = x * scale[None, :, None, None] + shift[None, :, None, None]
    xa = np.where(t > 0, t, t * slope[None, :, None, None]).astype(np.float64)
    y_ref = ops.conv2d_fwd(xa, w.astype(np.float64), 1, 1)
    cv = L.Conv(0, 16, 32, 3, 1, 1, 0)
    st = G.stream()
    xb, xv = G.to_nhwc(x)
    yb, yv = G.empty_nhwc(2, 9, 11, 32)
    keep, pw = G.pointwise(scale, shift, slope)
    wd = G.dev(w)
    pf = torch.zeros(lib.bp_conv_packed_floats(C.byref(cv), L.PACK_FWD), device="cuda")
    L.check(lib.bp_conv_pack(C.byref(cv), L.PACK_FWD, L.ptr(wd), L.ptr(pf), st))
    L.check(lib.bp_conv_forward(C.byref(cv), C.byref(xv), C.byref(pw), L.ptr(pf), L.ptr(wd), None, C.byref(yv),
                                impl, st))
    assert G.rel_err(G.from_nhwc(yb, 32), y_ref) < 2e-5
    # weight gradient sees the activated input too
    dy = rng.standard_normal(y_ref.shape).astype(np.float32)
    dyb, dyv = G.to_nhwc(dy)
    ws_bytes = lib.bp_conv_backward_weight_workspace(C.byref(cv), C.byref(xv), C.byref(dyv))
    ws = torch.zeros(ws_bytes // 8 + 8, dtype=torch.float64, device="cuda")
    dw = torch.zeros(w.shape, device="cuda")
    L.check(lib.bp_conv_backward_weight(C.byref(cv), C.byref(xv), C.byref(pw), C.byref(dyv), L.ptr(dw), None,
                                        L.ptr(ws), ws.numel() * 8, impl, st))
    dw_ref = ops.conv2d_bwd_weight(xa, dy.astype(np.float64), 1, 1, 3, 3)
    assert G.rel_err(dw.cpu().numpy(), dw_ref) < 1e-4


# (transposed, cin, cout, k, stride, pad, (n, h, w)): layers wide enough for the LDS-DMA pipelined kernels
# (igemm_dma_kernel, wgrad_tiles_dma_kernel), at sizes that leave ragged tiles and partial channel blocks.
DMA_CASES = [
    (0, 48, 80, 3, 1, 1, (3, 13, 19)),
    (0, 128, 128, 3, 1, 1, (2, 22, 37)),
    (0, 32, 64, 4, 2, 1, (2, 14, 22)),
    (0, 64, 128, 4, 2, 1, (2, 10, 38)),
    (1, 64, 32, 4, 2, 1, (3, 5, 9)),
    (1, 128, 64, 4, 2, 1, (2, 7, 18)),
    (0, 16, 32, 4, 2, 1, (3, 30, 70)),        # weights-resident persistent igemm (stride-2 gather)
    (0, 16, 8, 7, 1, 3, (3, 21, 45)),         # forward: conv_flat.hip flat_h7 (K split over two waves); weight gradient: conv_wgrad_flat.hip
    (0, 16, 8, 7, 1, 3, (9, 64, 250)),        # ... more tiles than workgroups (grid-stride walk)
    (0, 8, 16, 7, 1, 3, (3, 21, 45)),         # conv_flat.hip forward (weights in registers); its data gradient: wres
    (0, 16, 8, 7, 1, 3, (40, 64, 130)),       # conv_flat.hip data gradient with more tiles than workgroups
    (1, 32, 16, 4, 2, 1, (3, 9, 13)),         # conv_flat.hip four-phase transposed form (forward); gather: wres
    (1, 32, 16, 4, 2, 1, (24, 40, 80)),       # ... more tiles than workgroups
    (0, 32, 64, 4, 2, 1, (40, 64, 96)),       # conv_flat.hip stride-2 gather 32 -> 64: more tiles than workgroups
    (1, 64, 32, 4, 2, 1, (5, 31, 50)),        # ... as the data gradient of the transposed layer, ragged tiles
    (0, 8, 16, 8, 4, 2, (3, 37, 70)),         # conv_enc.hip: k8 stride-4 layer of the recognition / prior networks
    (0, 8, 16, 8, 4, 2, (100, 64, 128)),      # ... more tiles than workgroups (forward)
    (1, 16, 8, 8, 4, 2, (3, 9, 17)),          # ... the same two kernels with the roles swapped (transposed layer)
]


@pytest.mark.parametrize("case", DMA_CASES, ids=lambda c: "%s%d_%d_k%ds%d" % ("T" if c[0] else "C", *c[1:5]))
def test_dma_pipelined_kernels_ragged_views_and_activation(case):
    """Forward, data gradient and weight gradient of the wide layers, reading the input as a channel slice of
    a wider buffer (16-byte aligned, as the networks' concat buffers are) with a pending affine + leaky-ReLU
    whose act(0) != 0, against the float64 oracle."""
    lib = L.load()
    tr, ci, co, k, s, p, (n, h, w) = case
    rng = np.random.default_rng(ci * 7 + co + k)
    x = rng.standard_normal((n, ci, h, w)).astype(np.float32)
    wt = (rng.standard_normal(((ci, co) if tr else (co, ci)) + (k, k)) * 0.1).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, ci).astype(np.float32)
    shift = rng.uniform(0.2, 0.6, ci).astype(np.float32)
    slope = rng.uniform(0.0, 0.3, ci).astype(np.float32)
    t = x * scale[None, :, None, None] + shift[None, :, None, None]
    xa = np.where(t > 0, t, t * slope[None, :, None, None]).astype(np.float64)
    w64 = wt.astype(np.float64)
    y_ref = ops.convT2d_fwd(xa, w64, s, p, 0) if tr else ops.conv2d_fwd(xa, w64, s, p)
    _, _, ho, wo = y_ref.shape
    cv = L.Conv(tr, ci, co, k, s, p, 0)
    assert max(lib.bp_conv_kernel_id(C.byref(cv), d) for d in (L.PACK_FWD, L.PACK_BWD)) >= 100000, \
        "case is meant for the DMA kernels"
    st = G.stream()
    xb, xv = G.to_nhwc(x, cstride=ci + 24, coff=8)
    yb, yv = G.empty_nhwc(n, ho, wo, co, cstride=co + 12, coff=4)
    keep, pw = G.pointwise(scale, shift, slope)
    wd = G.dev(wt)
    pf = torch.zeros(lib.bp_conv_packed_floats(C.byref(cv), L.PACK_FWD), device="cuda")
    pb = torch.zeros(lib.bp_conv_packed_floats(C.byref(cv), L.PACK_BWD), device="cuda")
    L.check(lib.bp_conv_pack(C.byref(cv), L.PACK_FWD, L.ptr(wd), L.ptr(pf), st))
    L.check(lib.bp_conv_pack(C.byref(cv), L.PACK_BWD, L.ptr(wd), L.ptr(pb), st))
    L.check(lib.bp_conv_forward(C.byref(cv), C.byref(xv), C.byref(pw), L.ptr(pf), L.ptr(wd), None, C.byref(yv),
                                L.IMPL_MFMA, st), "forward")
    assert G.rel_err(G.from_nhwc(yb, co, coff=4), y_ref) < 2e-5
    assert torch.isnan(yb[..., :4]).all() and torch.isnan(yb[..., 4 + co:]).all(), "stores outside the view"

    dy = rng.standard_normal(y_ref.shape).astype(np.float32)
    dy64 = dy.astype(np.float64)
    dyb, dyv = G.to_nhwc(dy, cstride=co + 4, coff=4)
    dxb, dxv = G.empty_nhwc(n, h, w, ci)
    L.check(lib.bp_conv_backward_data(C.byref(cv), C.byref(dyv), L.ptr(pb), L.ptr(wd), C.byref(dxv), L.IMPL_MFMA,
                                      st), "backward_data")
    dx_ref = ops.convT2d_bwd_data(dy64, w64, s, p) if tr else ops.conv2d_bwd_data(dy64, w64, s, p, h, w)
    assert G.rel_err(G.from_nhwc(dxb, ci), dx_ref) < 2e-5

    ws_bytes = lib.bp_conv_backward_weight_workspace(C.byref(cv), C.byref(xv), C.byref(dyv))
    ws = torch.zeros(ws_bytes // 8 + 8, dtype=torch.float64, device="cuda")
    dw = torch.full(wt.shape, float("nan"), device="cuda")
    L.check(lib.bp_conv_backward_weight(C.byref(cv), C.byref(xv), C.byref(pw), C.byref(dyv), L.ptr(dw), None,
                                        L.ptr(ws), ws.numel() * 8, L.IMPL_MFMA, st), "backward_weight")
    dw_ref = ops.convT2d_bwd_weight(xa, dy64, s, p, k, k) if tr else ops.conv2d_bwd_weight(xa, dy64, s, p, k, k)
    assert G.rel_err(dw.cpu().numpy(), dw_ref) < 1e-4


# unit-stride layers with cin*cout <= 16 served by the vector-ALU kernel (conv_small.hip) in at least one direction
SMALL_CASES = [(8, 1, 5, 2), (1, 8, 5, 2), (1, 16, 5, 2), (1, 1, 3, 1), (32, 1, 9, 4), (2, 32, 9, 4), (64, 1, 4, 1)]


@pytest.mark.parametrize("case", SMALL_CASES, ids=lambda c: "%d_%d_k%d" % c[:3])
def test_small_channel_kernels_ragged_views_and_activation(case):
    lib = L.load()
    ci, co, k, p = case
    n, h, w = 3, 37, 70                       # ragged against the 16 x 64 tiles
    rng = np.random.default_rng(ci * 13 + co)
    x = rng.standard_normal((n, ci, h, w)).astype(np.float32)
    wt = (rng.standard_normal((co, ci, k, k)) * 0.2).astype(np.float32)
    bias = rng.standard_normal(co).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, ci).astype(np.float32)
    shift = rng.uniform(0.2, 0.6, ci).astype(np.float32)
    slope = rng.uniform(0.0, 0.3, ci).astype(np.float32)
    t = x * scale[None, :, None, None] + shift[None, :, None, None]
    xa = np.where(t > 0, t, t * slope[None, :, None, None]).astype(np.float64)
    w64 = wt.astype(np.float64)
    y_ref = ops.conv2d_fwd(xa, w64, 1, p) + bias[None, :, None, None]
    cv = L.Conv(0, ci, co, k, 1, p, 0)
    ids = [lib.bp_conv_kernel_id(C.byref(cv), d) for d in (L.PACK_FWD, L.PACK_BWD)]
    assert max(ids) >= 900000, "case is meant for small_conv_kernel"
    st = G.stream()
    xb, xv = G.to_nhwc(x, cstride=ci + 4, coff=4)
    ho, wo = y_ref.shape[2:]
    yb, yv = G.empty_nhwc(n, ho, wo, co, cstride=co + 3, coff=2)
    keep, pw = G.pointwise(scale, shift, slope)
    wd, bd = G.dev(wt), G.dev(bias)
    pf = torch.zeros(lib.bp_conv_packed_floats(C.byref(cv), L.PACK_FWD), device="cuda")
    pb = torch.zeros(lib.bp_conv_packed_floats(C.byref(cv), L.PACK_BWD), device="cuda")
    L.check(lib.bp_conv_pack(C.byref(cv), L.PACK_FWD, L.ptr(wd), L.ptr(pf), st))
    L.check(lib.bp_conv_pack(C.byref(cv), L.PACK_BWD, L.ptr(wd), L.ptr(pb), st))
    L.check(lib.bp_conv_forward(C.byref(cv), C.byref(xv), C.byref(pw), L.ptr(pf), L.ptr(wd), L.ptr(bd), C.byref(yv),
                                L.IMPL_MFMA, st), "forward")
    assert G.rel_err(G.from_nhwc(yb, co, coff=2), y_ref) < 2e-5
    assert torch.isnan(yb[..., :2]).all() and torch.isnan(yb[..., 2 + co:]).all(), "stores outside the view"
    dy = rng.standard_normal(y_ref.shape).astype(np.float32)
    dyb, dyv = G.to_nhwc(dy, cstride=co + 1, coff=1)
    dxb, dxv = G.empty_nhwc(n, h, w, ci, cstride=ci + 2, coff=1)
    L.check(lib.bp_conv_backward_data(C.byref(cv), C.byref(dyv), L.ptr(pb), L.ptr(wd), C.byref(dxv), L.IMPL_MFMA,
                                      st), "backward_data")
    dx_ref = ops.conv2d_bwd_data(dy.astype(np.float64), w64, 1, p, h, w)
    assert G.rel_err(G.from_nhwc(dxb, ci, coff=1), dx_ref) < 2e-5
    assert torch.isnan(dxb[..., :1]).all() and torch.isnan(dxb[..., 1 + ci:]).all(), "stores outside the view"

@pytest.mark.parametrize("tr", [0, 1], ids=["conv", "transposed"])
def test_encoder_k8s4_kernels_unaligned_views_and_bias(tr):
    """conv_enc.hip through views whose channel slices are not 16-byte aligned (scalar loads / stores) and with a
    bias, forward and data gradient, against the float64 oracle."""
    lib = L.load()
    ci, co, k, s, p = (16, 8, 8, 4, 2) if tr else (8, 16, 8, 4, 2)
    n, h, w = (2, 7, 19) if tr else (2, 30, 77)
    rng = np.random.default_rng(5 + tr)
    x = rng.standard_normal((n, ci, h, w)).astype(np.float32)
    wt = (rng.standard_normal(((ci, co) if tr else (co, ci)) + (k, k)) * 0.1).astype(np.float32)
    bias = rng.standard_normal(co).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, ci).astype(np.float32)
    shift = rng.uniform(-0.2, 0.6, ci).astype(np.float32)
    slope = rng.uniform(0.0, 0.3, ci).astype(np.float32)
    t = x * scale[None, :, None, None] + shift[None, :, None, None]
    xa = np.where(t > 0, t, t * slope[None, :, None, None]).astype(np.float64)
    w64 = wt.astype(np.float64)
    y_ref = (ops.convT2d_fwd(xa, w64, s, p, 0) if tr else ops.conv2d_fwd(xa, w64, s, p)) + bias[None, :, None, None]
    _, _, ho, wo = y_ref.shape
    cv = L.Conv(tr, ci, co, k, s, p, 0)
    assert sorted(lib.bp_conv_kernel_id(C.byref(cv), d) for d in (L.PACK_FWD, L.PACK_BWD)) == [760000, 770000]
    st = G.stream()
    xb, xv = G.to_nhwc(x, cstride=ci + 3, coff=1)
    yb, yv = G.empty_nhwc(n, ho, wo, co, cstride=co + 3, coff=2)
    keep, pw = G.pointwise(scale, shift, slope)
    wd, bd = G.dev(wt), G.dev(bias)
    pf = torch.zeros(lib.bp_conv_packed_floats(C.byref(cv), L.PACK_FWD), device="cuda")
    pb = torch.zeros(lib.bp_conv_packed_floats(C.byref(cv), L.PACK_BWD), device="cuda")
    L.check(lib.bp_conv_pack(C.byref(cv), L.PACK_FWD, L.ptr(wd), L.ptr(pf), st))
    L.check(lib.bp_conv_pack(C.byref(cv), L.PACK_BWD, L.ptr(wd), L.ptr(pb), st))
    L.check(lib.bp_conv_forward(C.byref(cv), C.byref(xv), C.byref(pw), L.ptr(pf), L.ptr(wd), L.ptr(bd), C.byref(yv),
                                L.IMPL_MFMA, st), "forward")
    assert G.rel_err(G.from_nhwc(yb, co, coff=2), y_ref) < 2e-5
    assert torch.isnan(yb[..., :2]).all() and torch.isnan(yb[..., 2 + co:]).all(), "stores outside the view"
    dy = rng.standard_normal(y_ref.shape).astype(np.float32)
    dyb, dyv = G.to_nhwc(dy, cstride=co + 1, coff=1)
    dxb, dxv = G.empty_nhwc(n, h, w, ci, cstride=ci + 2, coff=1)
    L.check(lib.bp_conv_backward_data(C.byref(cv), C.byref(dyv), L.ptr(pb), L.ptr(wd), C.byref(dxv), L.IMPL_MFMA,
                                      st), "backward_data")
    dx_ref = ops.convT2d_bwd_data(dy.astype(np.float64), w64, s, p) if tr \
        else ops.conv2d_bwd_data(dy.astype(np.float64), w64, s, p, h, w)
    assert G.rel_err(G.from_nhwc(dxb, ci, coff=1), dx_ref) < 2e-5
    assert torch.isnan(dxb[..., :1]).all() and torch.isnan(dxb[..., 1 + ci:]).all(), "stores outside the view"
    ws_bytes = lib.bp_conv_backward_weight_workspace(C.byref(cv), C.byref(xv), C.byref(dyv))
    ws = torch.zeros(ws_bytes // 8 + 8, dtype=torch.float64, device="cuda")
    dw = torch.full(wt.shape, float("nan"), device="cuda")
    L.check(lib.bp_conv_backward_weight(C.byref(cv), C.byref(xv), C.byref(pw), C.byref(dyv), L.ptr(dw), None,
                                        L.ptr(ws), ws.numel() * 8, L.IMPL_MFMA, st), "backward_weight")
    dw_ref = ops.convT2d_bwd_weight(xa, dy.astype(np.float64), s, p, k, k) if tr \
        else ops.conv2d_bwd_weight(xa, dy.astype(np.float64), s, p, k, k)
    assert G.rel_err(dw.cpu().numpy(), dw_ref) < 2e-5


# weight gradients of unit-stride layers that end in ONE channel (conv_wgrad_thin.hip): the (ky, gx)-column MFMA kernel
# for 5..8 -> 1 k5 and the register-band kernel for 1 -> 1 k3 / k5; a transposed layer swaps the roles (its input is
# the one-channel "Y" of the kernel and carries the pending activation)
THIN_CASES = [(0, 8, 1, 5, 2), (0, 6, 1, 5, 2), (0, 1, 1, 3, 1), (0, 1, 1, 5, 2), (1, 1, 1, 3, 1), (1, 1, 7, 5, 2)]


@pytest.mark.parametrize("shape", [(3, 37, 70), (2, 64, 96)], ids=["ragged", "tiles"])
@pytest.mark.parametrize("case", THIN_CASES, ids=lambda c: "%s%d_%d_k%d" % ("T" if c[0] else "C", *c[1:4]))
def test_one_channel_tail_weight_gradients(case, shape):
    lib = L.load()
    tr, ci, co, k, p = case
    n, h, w = shape
    rng = np.random.default_rng(ci * 17 + co + k)
    x = rng.standard_normal((n, ci, h, w)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, ci).astype(np.float32)
    shift = rng.uniform(-0.3, 0.6, ci).astype(np.float32)
    slope = rng.uniform(0.0, 0.3, ci).astype(np.float32)
    t = x * scale[None, :, None, None] + shift[None, :, None, None]
    xa = np.where(t > 0, t, t * slope[None, :, None, None]).astype(np.float64)
    dy = rng.standard_normal((n, co, h, w)).astype(np.float32)
    cv = L.Conv(tr, ci, co, k, 1, p, 0)
    st = G.stream()
    for cs_extra, coff in ((0, 0), (3, 2)):
        xb, xv = G.to_nhwc(x, cstride=ci + cs_extra, coff=coff if cs_extra else 0)
        dyb, dyv = G.to_nhwc(dy, cstride=co + cs_extra, coff=1 if cs_extra else 0)
        keep, pw = G.pointwise(scale, shift, slope)
        ws_bytes = lib.bp_conv_backward_weight_workspace(C.byref(cv), C.byref(xv), C.byref(dyv))
        ws = torch.zeros(ws_bytes // 8 + 8, dtype=torch.float64, device="cuda")
        shape_w = (ci, co, k, k) if tr else (co, ci, k, k)
        dw = torch.full(shape_w, float("nan"), device="cuda")
        L.check(lib.bp_conv_backward_weight(C.byref(cv), C.byref(xv), C.byref(pw), C.byref(dyv), L.ptr(dw), None,
                                            L.ptr(ws), ws.numel() * 8, L.IMPL_MFMA, st), "backward_weight")
        dw_ref = ops.convT2d_bwd_weight(xa, dy.astype(np.float64), 1, p, k, k) if tr \
            else ops.conv2d_bwd_weight(xa, dy.astype(np.float64), 1, p, k, k)
        assert G.rel_err(dw.cpu().numpy(), dw_ref) < 2e-5, (cs_extra, coff)


# strided layers with one or two channels on one side, served by the per-pixel kernels of conv_small.hip
# (transposed, cin, cout, k, stride, pad)
TINY_CASES = [(0, 1, 8, 4, 2, 1), (0, 2, 8, 4, 2, 1), (1, 1, 1, 8, 4, 2), (1, 1, 1, 4, 2, 1), (0, 1, 1, 8, 4, 2),
              (0, 1, 1, 4, 2, 1)]


@pytest.mark.parametrize("case", TINY_CASES, ids=lambda c: "%s%d_%d_k%ds%d" % ("T" if c[0] else "C", *c[1:5]))
def test_strided_few_channel_kernels_ragged_views_and_activation(case):
    """Forward and data gradient of the 512^2 ends of the recognition / prior networks and of the latent up-sampler
    (cvae.py:26-45 via arch q_x_in / q_y_in / prior_z_y / p_z_in) on views that are channel slices, with a pending
    activation whose act(0) != 0 and a bias, against the float64 oracle."""
    lib = L.load()
    tr, ci, co, k, s, p = case
    n, h, w = 3, 21, 38
    rng = np.random.default_rng(ci * 17 + co + k)
    x = rng.standard_normal((n, ci, h, w)).astype(np.float32)
    wt = (rng.standard_normal(((ci, co) if tr else (co, ci)) + (k, k)) * 0.2).astype(np.float32)
    bias = rng.standard_normal(co).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, ci).astype(np.float32)
    shift = rng.uniform(0.2, 0.6, ci).astype(np.float32)
    slope = rng.uniform(0.0, 0.3, ci).astype(np.float32)
    t = x * scale[None, :, None, None] + shift[None, :, None, None]
    xa = np.where(t > 0, t, t * slope[None, :, None, None]).astype(np.float64)
    w64 = wt.astype(np.float64)
    y_ref = (ops.convT2d_fwd(xa, w64, s, p, 0) if tr else ops.conv2d_fwd(xa, w64, s, p)) + bias[None, :, None, None]
    cv = L.Conv(tr, ci, co, k, s, p, 0)
    ids = [lib.bp_conv_kernel_id(C.byref(cv), d) for d in (L.PACK_FWD, L.PACK_BWD)]
    # ({1,2} -> 8 forward: the MFMA kernel of conv_enc.hip since round 3; everything else here: per-pixel kernels)
    assert 800000 <= ids[0] < 900000 or ids[0] in (780001, 780002), "case is meant for the few-channel strided kernels"
    st = G.stream()
    xb, xv = G.to_nhwc(x, cstride=ci + 4, coff=3)
    ho, wo = y_ref.shape[2:]
    yb, yv = G.empty_nhwc(n, ho, wo, co, cstride=co + 4, coff=4)
    keep, pw = G.pointwise(scale, shift, slope)
    wd, bd = G.dev(wt), G.dev(bias)
    pf = torch.zeros(lib.bp_conv_packed_floats(C.byref(cv), L.PACK_FWD), device="cuda")
    pb = torch.zeros(lib.bp_conv_packed_floats(C.byref(cv), L.PACK_BWD), device="cuda")
    L.check(lib.bp_conv_pack(C.byref(cv), L.PACK_FWD, L.ptr(wd), L.ptr(pf), st))
    L.check(lib.bp_conv_pack(C.byref(cv), L.PACK_BWD, L.ptr(wd), L.ptr(pb), st))
    L.check(lib.bp_conv_forward(C.byref(cv), C.byref(xv), C.byref(pw), L.ptr(pf), L.ptr(wd), L.ptr(bd), C.byref(yv),
                                L.IMPL_MFMA, st), "forward")
    assert G.rel_err(G.from_nhwc(yb, co, coff=4), y_ref) < 2e-5
    assert torch.isnan(yb[..., :4]).all() and torch.isnan(yb[..., 4 + co:]).all(), "stores outside the view"
    dy = rng.standard_normal(y_ref.shape).astype(np.float32)
    dy64 = dy.astype(np.float64)
    dyb, dyv = G.to_nhwc(dy, cstride=co + 1, coff=1)
    dxb, dxv = G.empty_nhwc(n, h, w, ci, cstride=ci + 2, coff=1)
    L.check(lib.bp_conv_backward_data(C.byref(cv), C.byref(dyv), L.ptr(pb), L.ptr(wd), C.byref(dxv), L.IMPL_MFMA,
                                      st), "backward_data")
    dx_ref = ops.convT2d_bwd_data(dy64, w64, s, p) if tr else ops.conv2d_bwd_data(dy64, w64, s, p, h, w)
    assert G.rel_err(G.from_nhwc(dxb, ci, coff=1), dx_ref) < 2e-5
    assert torch.isnan(dxb[..., :1]).all() and torch.isnan(dxb[..., 1 + ci:]).all(), "stores outside the view"


def test_stem_kernel_padded_slot_bias_and_activation():
    """Conv2d 3 -> 16 k5 (arch p_y_z_in.0) through conv_stem.hip: input as the model holds it (3 channels in a
    4-float pixel: 16-byte loads) and as a ragged slice (scalar loads), pending activation, bias, enough tiles that
    workgroups walk several of them (grid stride + double-buffered staging)."""
    lib = L.load()
    rng = np.random.default_rng(316)
    cv = L.Conv(0, 3, 16, 5, 1, 2, 0)
    assert lib.bp_conv_kernel_id(C.byref(cv), L.PACK_FWD) == 700000
    wt = (rng.standard_normal((16, 3, 5, 5)) * 0.2).astype(np.float32)
    bias = rng.standard_normal(16).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, 3).astype(np.float32)
    shift = rng.uniform(0.2, 0.6, 3).astype(np.float32)
    slope = rng.uniform(0.0, 0.3, 3).astype(np.float32)
    st = G.stream()
    wd, bd = G.dev(wt), G.dev(bias)
    keep, pw = G.pointwise(scale, shift, slope)
    pf = torch.zeros(lib.bp_conv_packed_floats(C.byref(cv), L.PACK_FWD), device="cuda")
    L.check(lib.bp_conv_pack(C.byref(cv), L.PACK_FWD, L.ptr(wd), L.ptr(pf), st))
    for (n, h, w), (cs, co_) in (((2, 19, 150), (4, 0)), ((3, 37, 70), (7, 2)), ((40, 130, 200), (4, 0))):
        x = rng.standard_normal((n, 3, h, w)).astype(np.float32)
        t = x * scale[None, :, None, None] + shift[None, :, None, None]
        xa = np.where(t > 0, t, t * slope[None, :, None, None]).astype(np.float64)
        y_ref = ops.conv2d_fwd(xa, wt.astype(np.float64), 1, 2) + bias[None, :, None, None]
        xb, xv = G.to_nhwc(x, cstride=cs, coff=co_)
        yb, yv = G.empty_nhwc(n, h, w, 16, cstride=20, coff=4)
        L.check(lib.bp_conv_forward(C.byref(cv), C.byref(xv), C.byref(pw), L.ptr(pf), L.ptr(wd), L.ptr(bd),
                                    C.byref(yv), L.IMPL_MFMA, st), "stem forward")
        assert G.rel_err(G.from_nhwc(yb, 16, coff=4), y_ref) < 2e-5
        assert torch.isnan(yb[..., :4]).all(), "stores outside the view"
        # weight gradient (stem_wgrad_kernel): sees the activated input, dy as a channel slice
        dy = rng.standard_normal(y_ref.shape).astype(np.float32)
        dyb, dyv = G.to_nhwc(dy, cstride=20, coff=4)
        ws_bytes = lib.bp_conv_backward_weight_workspace(C.byref(cv), C.byref(xv), C.byref(dyv))
        ws = torch.full((ws_bytes // 8 + 8,), float("nan"), dtype=torch.float64, device="cuda")
        dw = torch.full(wt.shape, float("nan"), device="cuda")
        db = torch.full((16,), float("nan"), device="cuda")
        L.check(lib.bp_conv_backward_weight(C.byref(cv), C.byref(xv), C.byref(pw), C.byref(dyv), L.ptr(dw), L.ptr(db),
                                            L.ptr(ws), ws.numel() * 8, L.IMPL_MFMA, st), "stem backward_weight")
        dw_ref = ops.conv2d_bwd_weight(xa, dy.astype(np.float64), 1, 2, 5, 5)
        assert G.rel_err(dw.cpu().numpy(), dw_ref) < 1e-4
        assert G.rel_err(db.cpu().numpy(), dy.astype(np.float64).sum(axis=(0, 2, 3))) < 1e-5


def test_conv_rejects_bad_shapes():
    lib = L.load()
    cv = L.Conv(0, 16, 32, 3, 1, 1, 0)
    xb, xv = G.empty_nhwc(2, 8, 8, 16)
    yb, yv = G.empty_nhwc(2, 7, 8, 32)          # wrong height
    assert lib.bp_conv_forward(C.byref(cv), C.byref(xv), None, None, L.ptr(xb), None, C.byref(yv), L.IMPL_DIRECT,
                               G.stream()) == -1
    bad = L.Conv(0, 0, 32, 3, 1, 1, 0)
    assert lib.bp_conv_packed_floats(C.byref(bad), L.PACK_FWD) == -1


def test_batchnorm_relu_chain(golden_ops):
    """channel sums -> finalize -> (lazy apply) ; act backward -> bn backward."""
    lib = L.load()
    st = G.stream()
    Pm = syn.fill_params({"0.weight": (5,), "0.bias": (5,)}, 400)
    gamma, beta = Pm["0.weight"], Pm["0.bias"]
    x = (syn.synthetic_eps((3, 5, 6, 7), seed=401) * 1.7 + 0.3).astype(np.float32)
    dy = syn.synthetic_eps((3, 5, 6, 7), seed=402)
    xb, xv = G.to_nhwc(x, cstride=8, coff=2)
    sums = torch.zeros(15, dtype=torch.float64, device="cuda")
    ws = torch.zeros(1 << 16, dtype=torch.float64, device="cuda")
    L.check(lib.bp_channel_sums(C.byref(xv), L.ptr(sums), L.ptr(ws), ws.numel() * 8, st))
    x64 = x.astype(np.float64)
    assert G.rel_err(sums[:5].cpu().numpy(), x64.sum(axis=(0, 2, 3))) < 1e-12
    assert G.rel_err(sums[5:10].cpu().numpy(), (x64 ** 2).sum(axis=(0, 2, 3))) < 1e-12
    g, b = G.dev(gamma), G.dev(beta)
    rm, rv = torch.zeros(5, device="cuda"), torch.ones(5, device="cuda")
    nbt = torch.zeros((), dtype=torch.int64, device="cuda")
    scale, shift = torch.zeros(5, device="cuda"), torch.zeros(5, device="cuda")
    mean = torch.zeros(5, device="cuda", dtype=torch.float64)
    inv = torch.zeros(5, device="cuda", dtype=torch.float64)
    cnt = float(3 * 6 * 7)
    L.check(lib.bp_bn_finalize(L.ptr(sums), cnt, 5, L.ptr(g), L.ptr(b), 1e-5, 0.1, L.ptr(rm), L.ptr(rv),
                               L.ptr(nbt), L.ptr(scale), L.ptr(shift), L.ptr(mean), L.ptr(inv), st))
    check("bn_relu/running_mean", rm.cpu().numpy(), golden_ops, 1e-6)
    check("bn_relu/running_var", rv.cpu().numpy(), golden_ops, 1e-6)
    assert int(nbt.item()) == 1
    slope = torch.zeros(5, device="cuda")
    pw = L.Pointwise(scale.data_ptr(), shift.data_ptr(), slope.data_ptr())
    out = torch.empty((3, 5, 6, 7), device="cuda")
    L.check(lib.bp_view_to_nchw(C.byref(xv), C.byref(pw), 0, L.ptr(out), st))
    check("bn_relu/y", out.cpu().numpy(), golden_ops, 1e-5)
    # backward
    dyb, dyv = G.to_nhwc(dy)
    gb, gv = G.empty_nhwc(3, 6, 7, 5)
    L.check(lib.bp_act_backward(C.byref(dyv), None, C.byref(xv), C.byref(pw), None, C.byref(gv), L.ptr(sums),
                                L.ptr(ws), ws.numel() * 8, st))
    dgam, dbet = torch.zeros(5, device="cuda"), torch.zeros(5, device="cuda")
    abc = torch.zeros(20, device="cuda", dtype=torch.float64)
    L.check(lib.bp_bn_backward_finalize(L.ptr(sums), cnt, 5, L.ptr(g), L.ptr(mean), L.ptr(inv), 1.0, L.ptr(dgam),
                                        L.ptr(dbet), L.ptr(abc), st))
    L.check(lib.bp_bn_backward_apply(C.byref(gv), C.byref(xv), L.ptr(abc), C.byref(gv), st))
    check("bn_relu/dx", G.from_nhwc(gb, 5), golden_ops, 2e-5)
    check("bn_relu/d_0.weight", dgam.cpu().numpy(), golden_ops, 2e-5)
    check("bn_relu/d_0.bias", dbet.cpu().numpy(), golden_ops, 2e-5)
    # eval-mode pointwise from running stats
    L.check(lib.bp_bn_eval_pointwise(5, L.ptr(g), L.ptr(b), L.ptr(rm), L.ptr(rv), 1e-5, L.ptr(scale),
                                     L.ptr(shift), st))
    L.check(lib.bp_view_to_nchw(C.byref(xv), C.byref(pw), 0, L.ptr(out), st))
    check("bn_relu/y_eval", out.cpu().numpy(), golden_ops, 1e-5)


def test_prelu_and_softplus(golden_ops):
    lib = L.load()
    st = G.stream()
    x = (syn.synthetic_eps((2, 3, 5, 5), seed=410) * 8.0)
    x[0, 0, 0, 0] = 25.0
    dy = syn.synthetic_eps((2, 3, 5, 5), seed=411)
    xb, xv = G.to_nhwc(x)
    keep, pw = G.pointwise(np.ones(3), np.zeros(3), np.full(3, 0.25))
    out = torch.empty((2, 3, 5, 5), device="cuda")
    L.check(lib.bp_view_to_nchw(C.byref(xv), C.byref(pw), 0, L.ptr(out), st))
    check("act_prelu/y", out.cpu().numpy(), golden_ops, 1e-6)
    L.check(lib.bp_view_to_nchw(C.byref(xv), None, 1, L.ptr(out), st))
    check("act_softplus/y", out.cpu().numpy(), golden_ops, 1e-6)
    dyb, dyv = G.to_nhwc(dy)
    gb, gv = G.empty_nhwc(2, 5, 5, 3)
    sums = torch.zeros(9, dtype=torch.float64, device="cuda")
    ws = torch.zeros(1 << 14, dtype=torch.float64, device="cuda")
    L.check(lib.bp_act_backward(C.byref(dyv), None, C.byref(xv), C.byref(pw), None, C.byref(gv), L.ptr(sums),
                                L.ptr(ws), ws.numel() * 8, st))
    check("act_prelu/dx", G.from_nhwc(gb, 3), golden_ops, 1e-6)
    da = torch.zeros(1, device="cuda")
    L.check(lib.bp_prelu_slope_grad(L.ptr(sums), 3, L.ptr(da), st))
    check("act_prelu/d_0.weight", da.cpu().numpy(), golden_ops, 1e-5)


def test_merge_aux_label_layout(golden_ops):
    lib = L.load()
    yv = syn.synthetic_eps((3, 1, 4, 5), seed=430)
    aux = np.array([0.0, 0.5, 2.0], np.float32)
    ob, ov = G.empty_nhwc(3, 4, 5, 2, cstride=4, coff=1)
    yd, ad = G.dev(yv), G.dev(aux)        # keep the device buffers alive across the call
    L.check(lib.bp_nchw_to_view(L.ptr(yd), 1, L.ptr(ad), 1, C.byref(ov), G.stream()))
    got = G.from_nhwc(ob, 2, coff=1)
    assert np.array_equal(got, golden_ops["merge_aux/out/full"].reshape(got.shape))   # bit-exact copy


@pytest.mark.parametrize("c,caux,cs,coff", [(1, 1, 2, 0), (1, 1, 4, 1), (1, 0, 1, 0), (2, 1, 4, 0), (3, 1, 6, 1), (1, 1, 3, 1)])
@pytest.mark.parametrize("hw", [(8, 12), (5, 7)], ids=["quads", "ragged"])
def test_nchw_to_view_all_forms(c, caux, cs, coff, hw):
    """bp_nchw_to_view: the four-pixels-per-thread kernel (h*w a multiple of 4) and the element-per-thread one must
    both be the bit-exact channel-last copy with the per-tile aux values broadcast over the pixels."""
    lib = L.load()
    n, (h, w) = 3, hw
    rng = np.random.default_rng(c * 7 + caux + cs)
    y = rng.standard_normal((n, c, h, w)).astype(np.float32)
    aux = rng.standard_normal((n, max(caux, 1))).astype(np.float32)
    ob, ov = G.empty_nhwc(n, h, w, c + caux, cstride=cs, coff=coff)
    yd, ad = G.dev(y), G.dev(aux)
    L.check(lib.bp_nchw_to_view(L.ptr(yd), c, L.ptr(ad) if caux else None, caux, C.byref(ov), G.stream()))
    got = G.from_nhwc(ob, c + caux, coff=coff)
    ref = np.concatenate([y] + ([np.broadcast_to(aux[:, :caux, None, None], (n, caux, h, w))] if caux else []), axis=1)
    assert np.array_equal(got, ref)
    rest = torch.ones(cs, dtype=torch.bool); rest[coff:coff + c + caux] = False
    assert torch.isnan(ob[..., rest.cuda()]).all(), "stores outside the view"


def test_adam_matches_torch():
    lib = L.load()
    rng = np.random.default_rng(0)
    p0 = rng.standard_normal(10007).astype(np.float32)
    ref = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.Adam([ref], lr=1e-3)
    p = G.dev(p0)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for step in range(1, 4):
        g = rng.standard_normal(10007).astype(np.float32)
        ref.grad = torch.from_numpy(g.copy())
        opt.step()
        gd = G.dev(g)
        L.check(lib.bp_adam_step(L.ptr(p), L.ptr(gd), L.ptr(m), L.ptr(v), p.numel(), 1e-3, 0.9, 0.999, 1e-8,
                                 step, G.stream()))
        torch.cuda.synchronize()
    assert G.rel_err(p.cpu().numpy(), ref.detach().numpy()) < 1e-6


@pytest.mark.parametrize("case", [(0, 32, 64, 4, 2, 1, (2, 14, 22)), (1, 64, 32, 4, 2, 1, (2, 5, 9)), (0, 16, 8, 7, 1, 3, (2, 11, 21)),
                                  (1, 32, 16, 4, 2, 1, (2, 9, 13))],
                         ids=lambda c: "%s%d_%d_k%ds%d" % ("T" if c[0] else "C", *c[1:5]))
def test_auto_takes_the_direct_kernel_for_views_the_register_resident_kernels_refuse(case):
    """conv_flat.hip stores 16 bytes per lane: a produced view at an odd channel offset is refused by name
    (BP_IMPL_MFMA -> BP_EUNSUPPORTED) and served by the direct kernel under BP_IMPL_AUTO."""
    lib = L.load()
    tr, ci, co, k, s, p, (n, h, w) = case
    rng = np.random.default_rng(ci + 3 * co + k)
    x = rng.standard_normal((n, ci, h, w)).astype(np.float32)
    wt = (rng.standard_normal(((ci, co) if tr else (co, ci)) + (k, k)) * 0.1).astype(np.float32)
    y_ref = ops.convT2d_fwd(x.astype(np.float64), wt.astype(np.float64), s, p, 0) if tr else \
        ops.conv2d_fwd(x.astype(np.float64), wt.astype(np.float64), s, p)
    _, _, ho, wo = y_ref.shape
    cv = L.Conv(tr, ci, co, k, s, p, 0)
    st = G.stream()
    xb, xv = G.to_nhwc(x)
    yb, yv = G.empty_nhwc(n, ho, wo, co, cstride=co + 3, coff=1)
    wd = G.dev(wt)
    pf = torch.zeros(lib.bp_conv_packed_floats(C.byref(cv), L.PACK_FWD), device="cuda")
    L.check(lib.bp_conv_pack(C.byref(cv), L.PACK_FWD, L.ptr(wd), L.ptr(pf), st))
    assert lib.bp_conv_forward(C.byref(cv), C.byref(xv), None, L.ptr(pf), L.ptr(wd), None, C.byref(yv), L.IMPL_MFMA, st) == -2
    L.check(lib.bp_conv_forward(C.byref(cv), C.byref(xv), None, L.ptr(pf), L.ptr(wd), None, C.byref(yv), L.IMPL_AUTO, st))
    assert G.rel_err(G.from_nhwc(yb, co, coff=1), y_ref) < 2e-5


# ---- the weights-stationary fp32 kernel of the 128 -> 128 k3 trunk (csrc/conv_ws_f32.hip) against the tiled kernel
@pytest.mark.parametrize("act", ["none", "relu", "leaky"])
@pytest.mark.parametrize("shape", [(2, 9, 16), (3, 20, 32), (2, 64, 64), (70, 8, 16), (2, 11, 128), (1, 6, 192)],
                         ids=lambda s: "%dx%dx%d" % s)        # (128, 192: column strips of 64 pixels, the CGAN generator's trunk)
def test_weights_stationary_fp32_trunk_kernel(shape, act):
    """Forward (+ the batch-norm sums from the epilogue) and data gradient of Conv2d(128, 128, 3, 1, 1) through both fp32
    kernels (bp_set_option("f32_ws", 1 / 0)): each within 2e-5 of the float64 convolution, the sums within 1e-10 of the sum of
    magnitudes of the tensor as stored; views that are channel slices of wider buffers; a NaN activation reaches its
    3 x 3 footprint through the fused batch-norm + ReLU staging (torch.relu(NaN) is NaN)."""
    lib = L.load()
    n, h, w = shape
    ci = co = 128
    rng = np.random.default_rng(n * 100 + h + w)
    x = rng.standard_normal((n, ci, h, w)).astype(np.float32)
    wt = (rng.standard_normal((co, ci, 3, 3)) * 0.05).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, ci).astype(np.float32)
    shift = rng.uniform(0.2, 0.6, ci).astype(np.float32)
    slope = rng.uniform(0.0, 0.3, ci).astype(np.float32)
    if act == "relu":
        slope[:] = 0.0
    if act == "none":
        xa = x.astype(np.float64)
    else:
        t = x * scale[None, :, None, None] + shift[None, :, None, None]
        xa = np.where(t > 0, t, t * slope[None, :, None, None]).astype(np.float64)
    w64 = wt.astype(np.float64)
    y_ref = ops.conv2d_fwd(xa, w64, 1, 1)
    cv = L.Conv(0, ci, co, 3, 1, 1, 0)
    st = G.stream()
    xb, xv = G.to_nhwc(x, cstride=ci + 8, coff=4)
    keep, pw = G.pointwise(scale, shift, slope)
    pwp = None if act == "none" else C.byref(pw)
    wd = G.dev(wt)
    pf = torch.zeros(lib.bp_conv_packed_floats(C.byref(cv), L.PACK_FWD), device="cuda")
    pb = torch.zeros(lib.bp_conv_packed_floats(C.byref(cv), L.PACK_BWD), device="cuda")
    L.check(lib.bp_conv_pack(C.byref(cv), L.PACK_FWD, L.ptr(wd), L.ptr(pf), st))
    L.check(lib.bp_conv_pack(C.byref(cv), L.PACK_BWD, L.ptr(wd), L.ptr(pb), st))
    dy = rng.standard_normal(y_ref.shape).astype(np.float32)
    dyb, dyv = G.to_nhwc(dy)
    dx_ref = ops.conv2d_bwd_data(dy.astype(np.float64), w64, 1, 1, h, w)
    res = {}
    try:
        for on in (1, 0):
            assert lib.bp_set_option(b"f32_ws", on) == 0
            yb, yv = G.empty_nhwc(n, h, w, co, cstride=co + 12, coff=4)
            assert lib.bp_conv_ws_kind(C.byref(cv), L.PACK_FWD, C.byref(xv), C.byref(yv)) == (3 if on else 0)
            L.check(lib.bp_conv_forward(C.byref(cv), C.byref(xv), pwp, L.ptr(pf), L.ptr(wd), None, C.byref(yv),
                                        L.IMPL_MFMA, st), "forward")
            got = G.from_nhwc(yb, co, coff=4)
            assert G.rel_err(got, y_ref) < 2e-5, f"forward ws={on}"
            assert torch.isnan(yb[..., :4]).all() and torch.isnan(yb[..., 4 + co:]).all(), "stores outside the view"
            nb = lib.bp_conv_stats_workspace(C.byref(cv), L.PACK_FWD, C.byref(xv), C.byref(yv), L.IMPL_MFMA)
            assert nb > 0
            yb2, yv2 = G.empty_nhwc(n, h, w, co, cstride=co + 12, coff=4)
            sums = torch.full((2 * co,), float("nan"), dtype=torch.float64, device="cuda")
            wss = torch.full((nb // 8 + 8,), float("nan"), dtype=torch.float64, device="cuda")
            L.check(lib.bp_conv_forward_stats(C.byref(cv), C.byref(xv), pwp, L.ptr(pf), C.byref(yv2), L.ptr(sums),
                                              L.ptr(wss), nb, L.IMPL_MFMA, st), "forward + statistics")
            assert np.array_equal(G.from_nhwc(yb2, co, coff=4), got), "the statistics epilogue must not change the output"
            g64 = got.astype(np.float64)
            sm = sums.cpu().numpy()
            assert (np.abs(sm[:co] - g64.sum(axis=(0, 2, 3))) <= 1e-10 * np.abs(g64).sum(axis=(0, 2, 3)) + 1e-12).all()
            assert np.allclose(sm[co:], (g64 ** 2).sum(axis=(0, 2, 3)), rtol=1e-10, atol=1e-12)
            dxb, dxv = G.empty_nhwc(n, h, w, ci, cstride=ci + 4, coff=0)
            L.check(lib.bp_conv_backward_data(C.byref(cv), C.byref(dyv), L.ptr(pb), L.ptr(wd), C.byref(dxv), L.IMPL_MFMA, st),
                    "backward_data")
            dx = G.from_nhwc(dxb, ci)
            assert G.rel_err(dx, dx_ref) < 2e-5, f"backward_data ws={on}"
            assert torch.isnan(dxb[..., ci:]).all(), "stores outside the view"
            res[on] = (got, dx)
        assert not np.array_equal(res[1][0], res[0][0]) and not np.array_equal(res[1][1], res[0][1]), \
            "the weights-stationary kernel never ran (both results bit-equal)"
        # NaN propagation through the staging's activation
        if act == "relu":
            assert lib.bp_set_option(b"f32_ws", 1) == 0
            x2 = x.copy()
            x2[n - 1, 37, h // 2, w // 2] = np.nan
            xb2, xv2 = G.to_nhwc(x2, cstride=ci + 8, coff=4)
            yb, yv = G.empty_nhwc(n, h, w, co)
            L.check(lib.bp_conv_forward(C.byref(cv), C.byref(xv2), pwp, L.ptr(pf), L.ptr(wd), None, C.byref(yv),
                                        L.IMPL_MFMA, st), "forward")
            bad = np.isnan(G.from_nhwc(yb, co))
            assert bad[n - 1, :, h // 2 - 1:h // 2 + 2, w // 2 - 1:w // 2 + 2].all() and bad.sum() == co * 9
    finally:
        lib.bp_set_option(b"f32_ws", -1)


@pytest.mark.parametrize("act", ["none", "relu"])
@pytest.mark.parametrize("shape", [(2, 9, 16), (3, 20, 32), (2, 64, 64), (70, 8, 16), (2, 11, 128), (1, 6, 192), (2, 128, 128)],
                         ids=lambda s: "%dx%dx%d" % s)
def test_output_stationary_fp32_trunk_weight_gradient(shape, act):
    """bp_conv_backward_weight of Conv2d(128, 128, 3, 1, 1) through the output-stationary kernel (csrc/conv_wgrad_ws_f32.hip)
    and through the tap-blocked tiled kernel (bp_set_option("f32_wgrad_ws", 1 / 0)): both within 1e-4 of the float64
    correlation (fp32 products, fp32 partial sums per band, double across bands); views that are channel slices; a NaN in X
    reaches exactly the gradients of its input channel."""
    lib = L.load()
    n, h, w = shape
    ci = co = 128
    rng = np.random.default_rng(n + 10 * h + w)
    x = rng.standard_normal((n, ci, h, w)).astype(np.float32)
    dy = rng.standard_normal((n, co, h, w)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, ci).astype(np.float32)
    shift = rng.uniform(-0.3, 0.6, ci).astype(np.float32)
    slope = np.zeros(ci, np.float32) if w < 128 else np.full(ci, 0.2, np.float32)      # (the CGAN trunk: LeakyReLU(0.2))
    if act == "none":
        xa = x.astype(np.float64)
    else:
        t = x * scale[None, :, None, None] + shift[None, :, None, None]
        xa = np.where(t > 0, t, t * slope[None, :, None, None]).astype(np.float64)
    dw_ref = ops.conv2d_bwd_weight(xa, dy.astype(np.float64), 1, 1, 3, 3)
    cv = L.Conv(0, ci, co, 3, 1, 1, 0)
    st = G.stream()
    xb, xv = G.to_nhwc(x, cstride=ci + 8, coff=4)
    dyb, dyv = G.to_nhwc(dy, cstride=co + 4, coff=0)
    keep, pw = G.pointwise(scale, shift, slope)
    pwp = None if act == "none" else C.byref(pw)
    res = {}
    try:
        for on in (1, 0):
            assert lib.bp_set_option(b"f32_wgrad_ws", on) == 0
            nb = lib.bp_conv_backward_weight_workspace(C.byref(cv), C.byref(xv), C.byref(dyv))
            assert nb > 0
            ws = torch.full((nb // 8 + 8,), float("nan"), dtype=torch.float64, device="cuda")
            dw = torch.full((co, ci, 3, 3), float("nan"), device="cuda")
            L.check(lib.bp_conv_backward_weight(C.byref(cv), C.byref(xv), pwp, C.byref(dyv), L.ptr(dw), None, L.ptr(ws),
                                                ws.numel() * 8, L.IMPL_MFMA, st), "backward_weight")
            got = dw.cpu().numpy()
            assert G.rel_err(got, dw_ref) < 1e-4, f"ws={on}"
            res[on] = got
        assert not np.array_equal(res[1], res[0]), "the output-stationary kernel never ran (both results bit-equal)"
        assert G.rel_err(res[1], res[0]) < 1e-4
        assert lib.bp_set_option(b"f32_wgrad_ws", 1) == 0
        x2 = x.copy()
        x2[0, 77, h // 2, w // 2] = np.nan
        xb2, xv2 = G.to_nhwc(x2, cstride=ci + 8, coff=4)
        dw = torch.zeros((co, ci, 3, 3), device="cuda")
        L.check(lib.bp_conv_backward_weight(C.byref(cv), C.byref(xv2), pwp, C.byref(dyv), L.ptr(dw), None, L.ptr(ws),
                                            ws.numel() * 8, L.IMPL_MFMA, st), "backward_weight")
        bad = torch.isnan(dw).cpu().numpy()
        assert bad[:, 77].all() and not np.delete(bad, 77, axis=1).any()
    finally:
        lib.bp_set_option(b"f32_wgrad_ws", -1)
