#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING THE REAL REFERENCE (build container only).

    python tests/golden/make_goldens.py

Needs /root/reference (absent on the GPU box -- the fixtures travel instead).
Nothing from the reference is copied: the script imports
``baryon_painter.models.cvae`` / ``models.utils`` from where they lie, feeds them
seeded inputs and weights from ``baryon_painter_amd.utils.synthetic`` and stores
output *summaries* (tests/golden_util.py).  ``torch.randn`` is replaced by an
injected eps for the duration of a forward (cvae.py:64) so the sampler is
deterministic.
"""
import ast
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")

from baryon_painter.models import cvae as ref_cvae          # noqa: E402  (the reference)
from baryon_painter.models import utils as ref_utils        # noqa: E402
from baryon_painter_amd.models import arch as our_arch      # noqa: E402
from baryon_painter_amd.utils import synthetic as syn       # noqa: E402
from golden_util import distance, summarize, store_crops              # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)


class inject_eps:
    def __init__(self, eps):
        self.eps = eps

    def __enter__(self):
        self._orig = torch.randn
        eps = self.eps

        def fake(size=None, device=None, **kw):
            t = torch.as_tensor(eps, dtype=torch.float32)
            assert tuple(t.shape) == tuple(size), (t.shape, size)
            return t
        torch.randn = fake

    def __exit__(self, *a):
        torch.randn = self._orig


def load_params(model, seed):
    shapes = {k: tuple(v.shape) for k, v in model.named_parameters()}
    vals = syn.fill_params(shapes, seed)
    with torch.no_grad():
        for k, p in model.named_parameters():
            p.copy_(torch.from_numpy(vals[k]))
    return shapes


from make_goldens_cases import OP_CASES                     # noqa: E402


def op_goldens():
    out = {}
    for i, (name, kind, cfg, (n, h, w)) in enumerate(OP_CASES):
        cfg = dict(cfg)
        cfg.setdefault("bias", False)
        seq = ref_utils.build_sequential([(kind, cfg)])          # the reference's builder
        shapes = {k: tuple(v.shape) for k, v in seq.named_parameters()}
        vals = syn.fill_params(shapes, seed=100 + i, bn_jitter=True)
        with torch.no_grad():
            for k, p in seq.named_parameters():
                p.copy_(torch.from_numpy(vals[k]))
        x = syn.synthetic_eps((n, cfg["in_channels"], h, w), seed=200 + i)
        xt = torch.tensor(x, requires_grad=True)
        y = seq(xt)
        dy = syn.synthetic_eps(tuple(y.shape), seed=300 + i)
        y.backward(torch.from_numpy(dy))
        summarize(f"{name}/y", y.detach().numpy(), out)
        summarize(f"{name}/dx", xt.grad.numpy(), out)
        for k, p in seq.named_parameters():
            summarize(f"{name}/d_{k}", p.grad.numpy(), out)
    # batchnorm (train + eval), activations, residual block through the reference builder
    bn_arch = [("batchnorm", {"num_features": 5}), ("ReLU",)]
    seq = ref_utils.build_sequential(bn_arch)
    shapes = {k: tuple(v.shape) for k, v in seq.named_parameters()}
    vals = syn.fill_params(shapes, seed=400)
    with torch.no_grad():
        for k, p in seq.named_parameters():
            p.copy_(torch.from_numpy(vals[k]))
    x = syn.synthetic_eps((3, 5, 6, 7), seed=401) * 1.7 + 0.3
    xt = torch.tensor(x, requires_grad=True)
    y = seq(xt)
    dy = syn.synthetic_eps(tuple(y.shape), seed=402)
    y.backward(torch.from_numpy(dy))
    summarize("bn_relu/y", y.detach().numpy(), out)
    summarize("bn_relu/dx", xt.grad.numpy(), out)
    for k, p in seq.named_parameters():
        summarize(f"bn_relu/d_{k}", p.grad.numpy(), out)
    summarize("bn_relu/running_mean", seq[0].running_mean.numpy(), out)
    summarize("bn_relu/running_var", seq[0].running_var.numpy(), out)
    seq.train(False)
    summarize("bn_relu/y_eval", seq(torch.from_numpy(x)).detach().numpy(), out)

    for act in ("prelu", "softplus", "tanh", "sigmoid", ("Leaky ReLU", 0.2)):
        layer = (act,) if isinstance(act, str) else act
        seq = ref_utils.build_sequential([layer])
        tag = layer[0].lower().replace(" ", "_")
        x = syn.synthetic_eps((2, 3, 5, 5), seed=410) * 8.0
        x[0, 0, 0, 0] = 25.0      # beyond the softplus threshold
        xt = torch.tensor(x, requires_grad=True)
        y = seq(xt * 1.0)        # non-leaf: the reference's LeakyReLU is in-place
        dy = syn.synthetic_eps(tuple(y.shape), seed=411)
        y.backward(torch.from_numpy(dy))
        summarize(f"act_{tag}/y", y.detach().numpy(), out)
        summarize(f"act_{tag}/dx", xt.grad.numpy(), out)
        for k, p in seq.named_parameters():
            summarize(f"act_{tag}/d_{k}", p.grad.numpy(), out)

    seq = ref_utils.build_sequential([("residual block", ref_utils.res_block(8))])
    shapes = {k: tuple(v.shape) for k, v in seq.named_parameters()}
    vals = syn.fill_params(shapes, seed=420)
    with torch.no_grad():
        for k, p in seq.named_parameters():
            p.copy_(torch.from_numpy(vals[k]))
    x = syn.synthetic_eps((2, 8, 6, 6), seed=421)
    xt = torch.tensor(x, requires_grad=True)
    y = seq(xt)
    dy = syn.synthetic_eps(tuple(y.shape), seed=422)
    y.backward(torch.from_numpy(dy))
    summarize("resblock/y", y.detach().numpy(), out)
    summarize("resblock/dx", xt.grad.numpy(), out)
    for k, p in seq.named_parameters():
        summarize(f"resblock/d_{k}", p.grad.numpy(), out)

    # merge_aux_label (utils.py:159-182)
    yv = syn.synthetic_eps((3, 1, 4, 5), seed=430)
    aux = np.array([0.0, 0.5, 2.0], np.float32)
    summarize("merge_aux/out", ref_utils.merge_aux_label(torch.from_numpy(yv), torch.from_numpy(aux)).numpy(), out)
    return out


# --------------------------------------------------------------- full-model cases
# Equally valid executions of the SAME reference code on a CPU: thread count and the oneDNN switch change the
# blocking / summation order inside ATen's convolutions and reductions, nothing else.  The spread of their results
# around the float64 truth is the reference's own fp32 noise floor for each gradient.
REF_VARIANTS = [("t1", 1, True), ("t3", 3, True), ("t8nomkl", 8, False), ("t1nomkl", 1, False)]


def reference_variants(tag, arch, n, size, out, seed_w=7, seed_d=1234, seed_eps=99):
    x, y, aux = syn.synthetic_batch(n, size, size, seed=seed_d)
    eps = syn.synthetic_eps((arch.get("L", 1), n, *arch["dim_z"]), seed=seed_eps)
    dist = []
    for name, threads, mkldnn in REF_VARIANTS:
        torch.set_num_threads(threads)
        with torch.backends.mkldnn.flags(enabled=mkldnn):
            model = ref_cvae.CVAE(arch, "cpu")
            load_params(model, seed_w)
            model.train(True)
            with inject_eps(eps):
                elbo = model(torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(aux))
            (-elbo).backward()
        names = [k for k, _ in model.named_parameters()]
        dist.append([distance(f"{tag}/grad64/{k}", p.grad.numpy(), out) for k, p in model.named_parameters()])
        print(tag, name, "ELBO", float(elbo))
    torch.set_num_threads(8)
    # distance of each execution's gradient from the float64 truth (tests/golden_util.distance), [variant][parameter]
    out[f"{tag}/grad_variants"] = np.array(",".join(v[0] for v in REF_VARIANTS))
    out[f"{tag}/grad_variant_params"] = np.array(",".join(names))
    out[f"{tag}/grad_variant_dist"] = np.array(dist)


def model_golden(tag, arch, n, size, out, seed_w=7, seed_d=1234, seed_eps=99, adam=False, alpha_var=None,
                 crops=False):
    model = ref_cvae.CVAE(arch, "cpu")
    load_params(model, seed_w)
    if alpha_var is not None:
        model.alpha_var = alpha_var
    x, y, aux = syn.synthetic_batch(n, size, size, seed=seed_d)
    L = arch.get("L", 1)
    eps = syn.synthetic_eps((L, n, *arch["dim_z"]), seed=seed_eps)
    xt, yt, at = torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(aux)
    model.train(True)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    with inject_eps(eps):
        elbo = model(xt, yt, at)
    opt.zero_grad()
    (-elbo).backward()
    summarize(f"{tag}/stats", np.array(model.get_stats(), np.float64), out)
    summarize(f"{tag}/x_mu", model.x_mu.detach().numpy(), out)
    if crops:
        store_crops(f"{tag}/x_mu", model.x_mu.detach().numpy(), out)
    summarize(f"{tag}/z_mu", model.z_mu.detach().numpy(), out)
    summarize(f"{tag}/z_log_var", model.z_log_var.detach().numpy(), out)
    for k, p in model.named_parameters():
        summarize(f"{tag}/grad/{k}", p.grad.numpy(), out)       # d(-ELBO)/dp
    for k, b in model.named_buffers():
        summarize(f"{tag}/buf/{k}", b.numpy(), out)
    # validation-style forward: train mode, no_grad (painter.py:306-314) -- second BN update
    # paint-style sampling (cvae.py:149-162) in eval mode, fixed eps and fixed z
    model.train(False)
    eps1 = syn.synthetic_eps((1, n, *arch["dim_z"]), seed=seed_eps + 1)
    with inject_eps(eps1):
        s = model.sample_P(yt, aux_label=at)
    summarize(f"{tag}/sample_P_eval", s.numpy(), out)
    if crops:
        store_crops(f"{tag}/sample_P_eval", s.numpy(), out)
    zfix = syn.synthetic_eps((n, *arch["dim_z"]), seed=seed_eps + 2)
    s = model.sample_P(yt, aux_label=at, z=zfix)
    summarize(f"{tag}/sample_P_eval_zfix", s.numpy(), out)
    if crops:
        store_crops(f"{tag}/sample_P_eval_zfix", s.numpy(), out)
    if len(arch["p_y_z_out"]) > 1:
        mu, var = model.sample_P(yt, aux_label=at, z=zfix, return_var=True)
        summarize(f"{tag}/sample_P_eval_var", var.numpy(), out)
    if adam:      # one torch.optim.Adam step, lr 1e-3 (painter.py:93,228)
        opt.step()
        for k, p in model.named_parameters():
            summarize(f"{tag}/adam/{k}", p.detach().numpy(), out)
    out[f"{tag}/n_params"] = np.array(model.count_parameters())
    out[f"{tag}/stats_labels"] = np.array(",".join(model.get_stats_labels()))
    out[f"{tag}/state_keys"] = np.array(",".join(model.state_dict().keys()))
    print(tag, "ELBO", float(elbo), "stats", model.get_stats())


def truth64(tag, arch, n, size, out, seed_w=7, seed_d=1234, seed_eps=99):
    """Float64 "true value" of the same case from the NumPy oracle (itself pinned to the
    reference by tests/test_oracle_golden.py).  Lets the GPU tests state their gradient
    tolerance relative to the fp32 reference's OWN distance from the true value."""
    from oracle.cvae_oracle import CVAEOracle
    m = CVAEOracle(arch, dtype=np.float64)
    m.load_params(syn.fill_params(m.param_shapes(), seed_w))
    x, y, aux = syn.synthetic_batch(n, size, size, seed=seed_d)
    eps = syn.synthetic_eps((arch.get("L", 1), n, *arch["dim_z"]), seed=seed_eps)
    m.forward(x, y, aux, eps)
    g = m.backward(seed=-1.0)
    summarize(f"{tag}/stats64", np.array(m.get_stats(), np.float64), out)
    for k, v in g.items():
        summarize(f"{tag}/grad64/{k}", v, out)


def main():
    fid_txt = open("/root/reference/trained_models/CVAE/fiducial/architecture.txt").read()
    fid = ast.literal_eval(fid_txt)
    ours = our_arch.fiducial_architecture(512)
    assert repr(ours) == repr(fid), "fiducial_architecture() differs from the reference's architecture.txt"

    ops_out = op_goldens()
    np.savez_compressed(os.path.join(HERE, "ops.npz"), **ops_out)

    out = {}
    out["fiducial_arch_repr"] = np.array(repr(fid))
    a64 = syn.scaled_architecture(fid, 64)
    model_golden("fid64_n3", a64, 3, 64, out, adam=True)
    a128 = syn.scaled_architecture(fid, 128)
    model_golden("fid128_n2", a128, 2, 128, out)
    two = our_arch.fiducial_architecture(64, predict_var=True)
    # the checked-in script's two-head network (scripts/CVAE_single_scale.py:97-138)
    model_golden("twohead64_n2", two, 2, 64, out, alpha_var=0.3)
    model_golden("fid512_n2", fid, 2, 512, out, crops=True)
    truth64("fid512_n2", fid, 2, 512, out)
    truth64("fid128_n2", a128, 2, 128, out)
    reference_variants("fid512_n2", fid, 2, 512, out)
    reference_variants("fid128_n2", a128, 2, 128, out)
    # BASELINE.json configs[0] geometry: batch 4 of 256x256 tiles (dim_z 1x8x8)
    a256 = syn.scaled_architecture(fid, 256)
    model_golden("fid256_n4", a256, 4, 256, out, crops=True)
    truth64("fid256_n4", a256, 4, 256, out)
    reference_variants("fid256_n4", a256, 4, 256, out)
    np.savez_compressed(os.path.join(HERE, "model.npz"), **out)
    for f in ("ops.npz", "model.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
