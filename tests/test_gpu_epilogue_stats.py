"""GPU: convolution + per-channel sums in one launch (bp_conv_forward_stats / bp_conv_backward_data_stats).

The sums taken from the accumulators in the igemm epilogue must be the ones the separate streaming passes
(bp_channel_sums, bp_act_backward) produce: the training-mode batch-norm statistics of utils.py:146-147 and the two
reductions of its backward.  Checked against float64 sums of the oracle's convolution on ragged sizes, for every kernel
family (plain, LDS-DMA, LDS-DMA with fused phases), both on views that are channel slices of wider buffers.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from baryon_painter_amd import _lib as L
from oracle import ops

import gpu_util as G

pytestmark = pytest.mark.gpu

# (transposed, cin, cout, k, stride, pad, (n, h, w))
CASES = [
    (0, 32, 32, 3, 1, 1, (3, 13, 19)),
    (0, 32, 32, 3, 1, 1, (4, 96, 96)),        # > 128 workgroup rows: the first fold runs
    (0, 128, 128, 3, 1, 1, (2, 22, 37)),
    (0, 32, 64, 4, 2, 1, (2, 14, 22)),
    (0, 64, 128, 4, 2, 1, (2, 10, 38)),
    (1, 64, 32, 4, 2, 1, (3, 5, 9)),
    (1, 128, 64, 4, 2, 1, (2, 7, 18)),
    (1, 32, 16, 4, 2, 1, (2, 9, 11)),
    (0, 8, 16, 5, 1, 2, (2, 17, 23)),
    (0, 16, 16, 3, 1, 1, (2, 9, 33)),
    (0, 4, 16, 3, 1, 1, (2, 11, 21)),
    (0, 3, 16, 5, 1, 2, (3, 37, 70)),         # the generator's stem (conv_stem.hip), ragged against its 8 x 64 tiles
    (0, 16, 32, 4, 2, 1, (3, 30, 70)),         # weights-resident persistent kernel (igemm_wres_kernel): stride 2
    (0, 16, 8, 7, 1, 3, (2, 19, 45)),          # ... 49 taps, two pixels per MFMA column block
    (0, 8, 16, 7, 1, 3, (2, 19, 45)),
    (0, 16, 32, 4, 2, 1, (70, 64, 64)),        # ... more tiles than workgroups
    (1, 32, 16, 4, 2, 1, (3, 9, 11)),          # conv_flat.hip transposed 32 -> 16: sums kept per lane over the tiles
    (1, 32, 16, 4, 2, 1, (24, 40, 80)),
    (0, 32, 64, 4, 2, 1, (40, 64, 96)),        # conv_flat.hip stride-2 gather 32 -> 64: sums kept per lane in LDS
    (1, 64, 32, 4, 2, 1, (11, 40, 70)),        # ... and the transposed form 64 -> 32 (four phases x two channel blocks)
    (0, 3, 16, 5, 1, 2, (40, 130, 200)),      # ... with more tiles than workgroups (grid-stride walk)
    (0, 8, 16, 8, 4, 2, (3, 37, 70)),         # conv_enc.hip: sums kept per thread over its tiles, one row per workgroup
    (0, 8, 16, 8, 4, 2, (100, 64, 128)),      # ... more tiles than workgroups
    (0, 1, 8, 4, 2, 1, (3, 37, 70)),          # conv_enc.hip enc0_fwd_kernel: the {1,2} -> 8 stems of the encoders
    (0, 2, 8, 4, 2, 1, (3, 37, 70)),
    (0, 2, 8, 4, 2, 1, (40, 128, 256)),       # ... more tiles than workgroups
    (1, 1, 1, 8, 4, 2, (3, 9, 11)),           # conv_small.hip per-pixel transposed kernel (the latent up-sampler)
    (1, 1, 1, 4, 2, 1, (3, 9, 11)),
    (1, 1, 1, 8, 4, 2, (5, 32, 32)),          # ... more than 2048 workgroup rows: the fold runs
]


def _ids(c):
    return "%s%d_%d_k%ds%d_%dx%dx%d" % ("T" if c[0] else "C", c[1], c[2], c[3], c[4], *c[6])


def _close(got, ref, mag, tol=2e-5):
    """|got - ref| <= tol * (sum of the magnitudes of the terms): the error model of a float32 partial sum."""
    return np.all(np.abs(got - ref) <= tol * mag + 1e-30)


@pytest.mark.parametrize("case", CASES, ids=_ids)
def test_forward_statistics_from_the_epilogue(case):
    lib = L.load()
    tr, ci, co, k, s, p, (n, h, w) = case
    rng = np.random.default_rng(ci * 5 + co + k)
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
    st = G.stream()
    xb, xv = G.to_nhwc(x, cstride=ci + 8, coff=4)
    yb, yv = G.empty_nhwc(n, ho, wo, co, cstride=co + 12, coff=4)
    keep, pw = G.pointwise(scale, shift, slope)
    wd = G.dev(wt)
    pf = torch.zeros(lib.bp_conv_packed_floats(C.byref(cv), L.PACK_FWD), device="cuda")
    L.check(lib.bp_conv_pack(C.byref(cv), L.PACK_FWD, L.ptr(wd), L.ptr(pf), st))
    nb = lib.bp_conv_stats_workspace(C.byref(cv), L.PACK_FWD, C.byref(xv), C.byref(yv), L.IMPL_MFMA)
    sums = torch.full((2 * co,), float("nan"), dtype=torch.float64, device="cuda")
    ws = torch.full((nb // 8 + 8,), float("nan"), dtype=torch.float64, device="cuda")
    rc = lib.bp_conv_forward_stats(C.byref(cv), C.byref(xv), C.byref(pw), L.ptr(pf), C.byref(yv), L.ptr(sums),
                                   L.ptr(ws), nb, L.IMPL_MFMA, st)
    if nb == 0:
        assert rc == -2
        return
    L.check(rc, "forward + statistics")
    assert G.rel_err(G.from_nhwc(yb, co, coff=4), y_ref) < 2e-5
    assert torch.isnan(yb[..., :4]).all() and torch.isnan(yb[..., 4 + co:]).all(), "stores outside the view"
    got = sums.cpu().numpy()
    assert _close(got[:co], y_ref.sum(axis=(0, 2, 3)), np.abs(y_ref).sum(axis=(0, 2, 3)))
    assert _close(got[co:], (y_ref ** 2).sum(axis=(0, 2, 3)), (y_ref ** 2).sum(axis=(0, 2, 3)))
    # the same numbers as the streaming pass over the tensor just written (to float32 rounding of the elements)
    ref = torch.zeros(2 * co, dtype=torch.float64, device="cuda")
    nb2 = lib.bp_channel_sums_workspace(C.byref(yv))
    ws2 = torch.zeros(nb2 // 8 + 8, dtype=torch.float64, device="cuda")
    L.check(lib.bp_channel_sums(C.byref(yv), L.ptr(ref), L.ptr(ws2), nb2, st))
    ref = ref.cpu().numpy()
    assert _close(got[:co], ref[:co], np.abs(y_ref).sum(axis=(0, 2, 3)), 1e-6)
    assert _close(got[co:], ref[co:], ref[co:], 1e-6)
    # too small a workspace is refused, not overrun
    assert lib.bp_conv_forward_stats(C.byref(cv), C.byref(xv), C.byref(pw), L.ptr(pf), C.byref(yv), L.ptr(sums),
                                     L.ptr(ws), nb - 8, L.IMPL_MFMA, st) == -4


@pytest.mark.parametrize("case", CASES, ids=_ids)
def test_activation_backward_sums_from_the_data_gradient_epilogue(case):
    lib = L.load()
    tr, ci, co, k, s, p, (n, h, w) = case
    rng = np.random.default_rng(ci * 3 + co + k)
    wt = (rng.standard_normal(((ci, co) if tr else (co, ci)) + (k, k)) * 0.1).astype(np.float32)
    w64 = wt.astype(np.float64)
    ho = (h - 1) * s - 2 * p + k if tr else (h + 2 * p - k) // s + 1
    wo = (w - 1) * s - 2 * p + k if tr else (w + 2 * p - k) // s + 1
    dy = rng.standard_normal((n, co, ho, wo)).astype(np.float32)
    dy64 = dy.astype(np.float64)
    dx_ref = ops.convT2d_bwd_data(dy64, w64, s, p) if tr else ops.conv2d_bwd_data(dy64, w64, s, p, h, w)
    raw = rng.standard_normal((n, ci, h, w)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, ci).astype(np.float32)
    shift = rng.uniform(-0.4, 0.4, ci).astype(np.float32)
    slope = rng.uniform(0.0, 0.3, ci).astype(np.float32)
    slope[::3] = 0.0                                       # relu channels
    t = raw * scale[None, :, None, None] + shift[None, :, None, None]
    g_ref = np.where(t > 0, dx_ref, dx_ref * slope[None, :, None, None].astype(np.float64))
    cv = L.Conv(tr, ci, co, k, s, p, 0)
    st = G.stream()
    dyb, dyv = G.to_nhwc(dy, cstride=co + 4, coff=4)
    dxb, dxv = G.empty_nhwc(n, h, w, ci)
    rb, rv = G.to_nhwc(raw, cstride=ci + 8, coff=4)
    keep, pw = G.pointwise(scale, shift, slope)
    wd = G.dev(wt)
    pb = torch.zeros(lib.bp_conv_packed_floats(C.byref(cv), L.PACK_BWD), device="cuda")
    L.check(lib.bp_conv_pack(C.byref(cv), L.PACK_BWD, L.ptr(wd), L.ptr(pb), st))
    nb = lib.bp_conv_stats_workspace(C.byref(cv), L.PACK_BWD, C.byref(dxv), C.byref(dyv), L.IMPL_MFMA)
    sums = torch.full((3 * ci,), float("nan"), dtype=torch.float64, device="cuda")
    ws = torch.full((nb // 8 + 8,), float("nan"), dtype=torch.float64, device="cuda")
    rc = lib.bp_conv_backward_data_stats(C.byref(cv), C.byref(dyv), L.ptr(pb), C.byref(dxv), C.byref(rv),
                                         C.byref(pw), L.ptr(sums), L.ptr(ws), nb, st)
    if nb == 0:
        assert rc == -2
        return
    L.check(rc, "backward_data + activation sums")
    assert G.rel_err(G.from_nhwc(dxb, ci), dx_ref) < 2e-5
    got = sums.cpu().numpy()
    raw64 = raw.astype(np.float64)
    assert _close(got[:ci], g_ref.sum(axis=(0, 2, 3)), np.abs(g_ref).sum(axis=(0, 2, 3)))
    assert _close(got[ci:2 * ci], (g_ref * raw64).sum(axis=(0, 2, 3)), np.abs(g_ref * raw64).sum(axis=(0, 2, 3)))
    # bp_act_backward on the gradient just written gives the same two sums
    ref = torch.zeros(3 * ci, dtype=torch.float64, device="cuda")
    nb2 = lib.bp_act_backward_workspace(C.byref(rv))
    ws2 = torch.zeros(nb2 // 8 + 8, dtype=torch.float64, device="cuda")
    L.check(lib.bp_act_backward(C.byref(dxv), None, C.byref(rv), C.byref(pw), None, None, L.ptr(ref), L.ptr(ws2),
                                nb2, st))
    ref = ref.cpu().numpy()
    assert _close(got[:ci], ref[:ci], np.abs(g_ref).sum(axis=(0, 2, 3)), 1e-6)
    assert _close(got[ci:2 * ci], ref[ci:2 * ci], np.abs(g_ref * raw64).sum(axis=(0, 2, 3)), 1e-6)


@pytest.mark.parametrize("shape", [(3, 37, 70), (2, 64, 128), (5, 130, 200)], ids=["ragged", "tiles", "fold"])
@pytest.mark.parametrize("aligned", [True, False], ids=["vec", "scalar"])
@pytest.mark.parametrize("layer", [(8, 5, 2), (1, 3, 1)], ids=["8to1_k5", "1to1_k3"])
def test_data_gradient_with_the_whole_activation_backward(shape, aligned, layer):
    """bp_conv_backward_data_act (the heads' 8 -> 1 k5 and 1 -> 1 k3 layers): g = dx * act'(pw(raw)) stored in place
    of dx and the three sums of bp_act_backward, against the two separate launches and the float64 oracle."""
    lib = L.load()
    ci, k, p = layer
    co, s = 1, 1
    n, h, w = shape
    rng = np.random.default_rng(11)
    wt = (rng.standard_normal((co, ci, k, k)) * 0.1).astype(np.float32)
    dy = rng.standard_normal((n, co, h, w)).astype(np.float32)
    raw = rng.standard_normal((n, ci, h, w)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, ci).astype(np.float32)
    shift = rng.uniform(-0.4, 0.4, ci).astype(np.float32)
    slope = rng.uniform(0.05, 0.3, ci).astype(np.float32)
    dx_ref = ops.conv2d_bwd_data(dy.astype(np.float64), wt.astype(np.float64), s, p, h, w)
    t = raw.astype(np.float64) * scale[None, :, None, None] + shift[None, :, None, None]
    g_ref = np.where(t > 0, dx_ref, dx_ref * slope[None, :, None, None])
    cv = L.Conv(0, ci, co, k, s, p, 0)
    st = G.stream()
    dyb, dyv = G.to_nhwc(dy)
    rb, rv = G.to_nhwc(raw) if aligned else G.to_nhwc(raw, cstride=ci + 3, coff=1)
    gb, gv = G.empty_nhwc(n, h, w, ci) if aligned else G.empty_nhwc(n, h, w, ci, cstride=ci + 2, coff=1)
    keep, pw = G.pointwise(scale, shift, slope)
    wd = G.dev(wt)
    pb = torch.zeros(lib.bp_conv_packed_floats(C.byref(cv), L.PACK_BWD), device="cuda")
    L.check(lib.bp_conv_pack(C.byref(cv), L.PACK_BWD, L.ptr(wd), L.ptr(pb), st))
    nb = lib.bp_conv_backward_data_act_workspace(C.byref(cv), C.byref(dyv), C.byref(gv))
    assert nb > 0
    sums = torch.full((3 * ci,), float("nan"), dtype=torch.float64, device="cuda")
    ws = torch.full((nb // 8 + 8,), float("nan"), dtype=torch.float64, device="cuda")
    L.check(lib.bp_conv_backward_data_act(C.byref(cv), C.byref(dyv), L.ptr(pb), C.byref(gv), C.byref(rv), C.byref(pw),
                                          L.ptr(sums), L.ptr(ws), nb, st), "backward_data + activation backward")
    coff = 0 if aligned else 1
    assert G.rel_err(G.from_nhwc(gb, ci, coff=coff), g_ref) < 2e-5
    if not aligned:
        assert torch.isnan(gb[..., :1]).all() and torch.isnan(gb[..., 1 + ci:]).all(), "stores outside the view"
    got = sums.cpu().numpy()
    raw64 = raw.astype(np.float64)
    refs = [g_ref.sum(axis=(0, 2, 3)), (g_ref * raw64).sum(axis=(0, 2, 3)), np.where(t > 0, 0.0, dx_ref * t).sum(axis=(0, 2, 3))]
    mags = [np.abs(g_ref).sum(axis=(0, 2, 3)), np.abs(g_ref * raw64).sum(axis=(0, 2, 3)),
            np.abs(np.where(t > 0, 0.0, dx_ref * t)).sum(axis=(0, 2, 3))]
    for q in range(3):
        assert _close(got[q * ci:(q + 1) * ci], refs[q], mags[q]), q
    # the two separate launches: data gradient, then bp_act_backward writing g over it
    dxb, dxv = G.empty_nhwc(n, h, w, ci)
    L.check(lib.bp_conv_backward_data(C.byref(cv), C.byref(dyv), L.ptr(pb), L.ptr(wd), C.byref(dxv), L.IMPL_MFMA, st))
    ref = torch.zeros(3 * ci, dtype=torch.float64, device="cuda")
    r2b, r2v = G.to_nhwc(raw)
    nb2 = lib.bp_act_backward_workspace(C.byref(r2v))
    ws2 = torch.zeros(nb2 // 8 + 8, dtype=torch.float64, device="cuda")
    L.check(lib.bp_act_backward(C.byref(dxv), None, C.byref(r2v), C.byref(pw), None, C.byref(dxv), L.ptr(ref),
                                L.ptr(ws2), nb2, st))
    assert torch.equal(G_from(dxb), G_from(gb[..., coff:coff + ci])), "g differs from the two-pass result"
    ref = ref.cpu().numpy()
    for q in range(3):
        assert _close(got[q * ci:(q + 1) * ci], ref[q * ci:(q + 1) * ci], mags[q], 1e-6), q
    assert lib.bp_conv_backward_data_act(C.byref(cv), C.byref(dyv), L.ptr(pb), C.byref(gv), C.byref(rv), C.byref(pw),
                                         L.ptr(sums), L.ptr(ws), nb - 8, st) == -4
    # a layer whose kernel has no such epilogue says so
    cv2 = L.Conv(0, 16, 8, 7, 1, 3, 0)
    d2b, d2v = G.empty_nhwc(n, h, w, 8)
    g2b, g2v = G.empty_nhwc(n, h, w, 16)
    assert lib.bp_conv_backward_data_act_workspace(C.byref(cv2), C.byref(d2v), C.byref(g2v)) == 0


def G_from(t):
    return t.contiguous()


def test_statistics_are_refused_where_the_kernel_has_none():
    """Pixel-packed heads (<= 8 produced channels) and the vector-ALU layers keep the separate passes."""
    lib = L.load()
    for cv, (n, h, w) in ((L.Conv(0, 16, 8, 3, 1, 1, 0), (2, 9, 33)), (L.Conv(0, 1, 8, 5, 1, 2, 0), (2, 9, 33)),
                          (L.Conv(0, 16, 24, 3, 1, 1, 0), (2, 9, 33))):
        xb, xv = G.empty_nhwc(n, h, w, cv.cin)
        yb, yv = G.empty_nhwc(n, h, w, cv.cout)
        assert lib.bp_conv_stats_workspace(C.byref(cv), L.PACK_FWD, C.byref(xv), C.byref(yv), L.IMPL_MFMA) == 0


@pytest.mark.parametrize("case", [(0, 32, 32, 3, 1, 1, (4, 96, 96)), (0, 32, 64, 4, 2, 1, (5, 30, 44)), (0, 3, 16, 5, 1, 2, (3, 37, 70)),
                                  (1, 32, 16, 4, 2, 1, (3, 9, 11))], ids=_ids)
def test_forward_with_the_batch_norm_finalize_folded_in(case):
    """bp_conv_forward_bn == bp_conv_forward_stats followed by bp_bn_finalize, bit for bit (sums, scale / shift,
    saved statistics, running statistics, the batch counter)."""
    lib = L.load()
    tr, ci, co, k, s, p, (n, h, w) = case
    rng = np.random.default_rng(ci + co + k)
    x = rng.standard_normal((n, ci, h, w)).astype(np.float32)
    wt = (rng.standard_normal(((ci, co) if tr else (co, ci)) + (k, k)) * 0.1).astype(np.float32)
    ho = (h - 1) * s - 2 * p + k if tr else (h + 2 * p - k) // s + 1
    wo = (w - 1) * s - 2 * p + k if tr else (w + 2 * p - k) // s + 1
    cv = L.Conv(tr, ci, co, k, s, p, 0)
    st = G.stream()
    xb, xv = G.to_nhwc(x)
    wd = G.dev(wt)
    pf = torch.zeros(lib.bp_conv_packed_floats(C.byref(cv), L.PACK_FWD), device="cuda")
    L.check(lib.bp_conv_pack(C.byref(cv), L.PACK_FWD, L.ptr(wd), L.ptr(pf), st))
    gamma = torch.from_numpy(rng.uniform(0.5, 1.5, co).astype(np.float32)).cuda()
    beta = torch.from_numpy(rng.uniform(-0.5, 0.5, co).astype(np.float32)).cuda()
    out = []
    for fused in (False, True):
        yb, yv = G.empty_nhwc(n, ho, wo, co)
        nb = lib.bp_conv_stats_workspace(C.byref(cv), L.PACK_FWD, C.byref(xv), C.byref(yv), L.IMPL_MFMA)
        assert nb > 0
        ws = torch.zeros(nb // 8 + 8, dtype=torch.float64, device="cuda")
        sums = torch.zeros(3 * co, dtype=torch.float64, device="cuda")
        rm, rv = torch.full((co,), 0.25, device="cuda"), torch.full((co,), 0.75, device="cuda")
        nbt = torch.full((1,), 7, dtype=torch.int64, device="cuda")
        scale, shift = torch.zeros(co, device="cuda"), torch.zeros(co, device="cuda")
        sm, si = torch.zeros(co, dtype=torch.float64, device="cuda"), torch.zeros(co, dtype=torch.float64, device="cuda")
        cnt = float(n * ho * wo)
        if fused:
            bt = L.BnTrain(cnt, gamma.data_ptr(), beta.data_ptr(), 1e-5, 0.1, rm.data_ptr(), rv.data_ptr(), nbt.data_ptr(),
                           scale.data_ptr(), shift.data_ptr(), sm.data_ptr(), si.data_ptr())
            L.check(lib.bp_conv_forward_bn(C.byref(cv), C.byref(xv), None, L.ptr(pf), C.byref(yv), L.ptr(sums), C.byref(bt),
                                           L.ptr(ws), nb, L.IMPL_MFMA, st))
        else:
            L.check(lib.bp_conv_forward_stats(C.byref(cv), C.byref(xv), None, L.ptr(pf), C.byref(yv), L.ptr(sums),
                                              L.ptr(ws), nb, L.IMPL_MFMA, st))
            L.check(lib.bp_bn_finalize(L.ptr(sums), cnt, co, L.ptr(gamma), L.ptr(beta), 1e-5, 0.1, L.ptr(rm), L.ptr(rv),
                                       L.ptr(nbt), L.ptr(scale), L.ptr(shift), L.ptr(sm), L.ptr(si), st))
        torch.cuda.synchronize()
        out.append([t.cpu().numpy().copy() for t in (yb, sums[:2 * co], scale, shift, sm, si, rm, rv, nbt)])
    for a, b in zip(*out):
        assert np.array_equal(a, b)
    assert out[1][8][0] == 8
