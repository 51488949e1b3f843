"""GPU: deferred split-K reductions of the fp32 weight gradients (bp_wgrad_defer_begin / _flush, csrc/conv_wgrad.hip).

The feature moves ~30 small reduce launches of a backward pass into two; each dW must come out bit for bit as without
deferral, for any number of recorded jobs (the batch kernel takes 40 per launch), and a flush on a thread that never began
a deferral must fail loudly instead of leaving gradients unreduced."""
import ctypes as C
import os
import threading

import numpy as np
import pytest
import torch

from baryon_painter_amd import _lib as L
from baryon_painter_amd.models import arch as A
from baryon_painter_amd.utils import synthetic as syn

import gpu_util as G

pytestmark = pytest.mark.gpu


def _grads(defer):
    from baryon_painter_amd.models.cvae import CVAE
    old = os.environ.get("BP_DEFER_REDUCE")
    os.environ["BP_DEFER_REDUCE"] = defer
    try:
        arch = A.fiducial_architecture(64)
        m = CVAE(arch, "cuda:0")
        P = syn.fill_params({k: tuple(p.shape) for k, p in m.named_parameters()}, 7)
        with torch.no_grad():
            for k, p in m.named_parameters():
                p.copy_(torch.from_numpy(P[k]))
        x, y, aux = syn.synthetic_batch(4, 64, 64, seed=5)
        m._eps_override = syn.synthetic_eps((1, 4, *arch["dim_z"]), seed=6)
        m.train(True)
        e = m(torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(aux))
        (-e).backward()
        torch.cuda.synchronize()
        assert m._last.defer_reduce == (defer == "1")
        return m._flat_grads.clone()
    finally:
        if old is None:
            os.environ.pop("BP_DEFER_REDUCE", None)
        else:
            os.environ["BP_DEFER_REDUCE"] = old


def test_model_gradients_equal_with_and_without_deferral():
    a, b = _grads("1"), _grads("0")
    assert torch.isfinite(a).all() and float(a.abs().max()) > 0
    assert torch.equal(a, b), "deferred reductions must reproduce every weight gradient bit for bit"


def test_more_deferred_jobs_than_one_batch_launch_takes():
    """97 flagged calls (the batch kernel takes 40 jobs per launch) on the trunk layer's shape with different data: each dW equals
    the dW of the same call without deferral, bit for bit; nothing is reduced before the flush."""
    lib = L.load()
    st = G.stream()
    cv = L.Conv(0, 128, 128, 3, 1, 1, 0)          # the trunk layer: tap-blocked kernel + wgrad_reduce (the deferrable path)
    ci, co, k = 128, 128, 3
    n, h, w = 1, 10, 12
    ho, wo = h, w
    rng = np.random.default_rng(3)
    njob = 97
    xs = [G.to_nhwc(rng.standard_normal((n, ci, h, w)).astype(np.float32)) for _ in range(3)]
    dys = [G.to_nhwc(rng.standard_normal((n, co, ho, wo)).astype(np.float32)) for _ in range(3)]
    nb = lib.bp_conv_backward_weight_workspace(C.byref(cv), C.byref(xs[0][1]), C.byref(dys[0][1]))
    assert nb > 0
    want = {}
    ws0 = torch.zeros(nb // 8 + 8, dtype=torch.float64, device="cuda")
    for i in range(3):
        for j in range(3):
            dw = torch.full((co, ci, k, k), float("nan"), device="cuda")
            L.check(lib.bp_conv_backward_weight(C.byref(cv), C.byref(xs[i][1]), None, C.byref(dys[j][1]), L.ptr(dw), None,
                                                L.ptr(ws0), ws0.numel() * 8, L.IMPL_MFMA, st))
            want[(i, j)] = dw
    torch.cuda.synchronize()
    wss = [torch.zeros(nb // 8 + 8, dtype=torch.float64, device="cuda") for _ in range(njob)]
    dws = [torch.full((co, ci, k, k), float("nan"), device="cuda") for _ in range(njob)]
    assert lib.bp_wgrad_defer_begin() == 0
    for q in range(njob):
        i, j = q % 3, (q // 3) % 3
        L.check(lib.bp_conv_backward_weight(C.byref(cv), C.byref(xs[i][1]), None, C.byref(dys[j][1]), L.ptr(dws[q]), None,
                                            L.ptr(wss[q]), wss[q].numel() * 8, L.IMPL_MFMA | L.IMPL_DEFER, st))
    torch.cuda.synchronize()
    assert all(torch.isnan(d).all() for d in dws), "a deferred reduction ran before the flush"
    assert lib.bp_wgrad_defer_flush(1, st) == 0
    torch.cuda.synchronize()
    for q in range(njob):
        assert torch.equal(dws[q], want[(q % 3, (q // 3) % 3)]), q


def test_flush_without_begin_is_refused_per_thread():
    lib = L.load()
    st = G.stream()
    assert lib.bp_wgrad_defer_flush(1, st) == L.BP_EINVAL
    assert lib.bp_wgrad_defer_begin() == 0
    rc = []
    t = threading.Thread(target=lambda: rc.append(lib.bp_wgrad_defer_flush(1, None)))      # another host thread: no deferral there
    t.start()
    t.join()
    assert rc == [L.BP_EINVAL]
    assert lib.bp_wgrad_defer_flush(1, st) == 0        # the owning thread ends its (empty) deferral
    assert lib.bp_wgrad_defer_flush(-1, None) == 0     # abandoning is always allowed
