"""CPU: pin the NumPy oracle (oracle/) to the fixtures generated from the real
reference (tests/golden/make_goldens.py).  Tolerances are relative to each
tensor's scale; the goldens are fp32 torch/oneDNN results, the oracle runs in
float64, so the residual is the reference's own fp32 rounding."""
import ast

import numpy as np
import pytest

from baryon_painter_amd.models import arch as A
from baryon_painter_amd.utils import synthetic as syn
from golden_util import check
from oracle import cvae_oracle as O
from oracle import ops

from golden.make_goldens_cases import OP_CASES


def _fill(seq, prefix, seed):
    shapes = {}
    for p, layer in seq.named_layers(prefix):
        for name in layer.params:
            if name in layer.shapes:
                shapes[p + name] = layer.shapes[name]
    return syn.fill_params(shapes, seed)


@pytest.mark.parametrize("idx", range(len(OP_CASES)))
def test_conv_ops_match_reference(idx, golden_ops):
    name, kind, cfg, (n, h, w) = OP_CASES[idx]
    cfg = dict(cfg)
    cfg.setdefault("bias", False)
    seq = O.Sequential([(kind, cfg)])
    P = {k: v.astype(np.float64) for k, v in _fill(seq, "", 100 + idx).items()}
    x = syn.synthetic_eps((n, cfg["in_channels"], h, w), seed=200 + idx).astype(np.float64)
    y = seq.forward(x, P, "", True)
    dy = syn.synthetic_eps(y.shape, seed=300 + idx).astype(np.float64)
    g = {}
    dx = seq.backward(dy, g, "")
    check(f"{name}/y", y, golden_ops, 2e-5)
    check(f"{name}/dx", dx, golden_ops, 2e-5)
    for k, v in g.items():
        check(f"{name}/d_{k}", v, golden_ops, 5e-5)


def test_batchnorm_relu(golden_ops):
    seq = O.Sequential([("batchnorm", {"num_features": 5}), ("ReLU",)])
    P = {k: v.astype(np.float64) for k, v in _fill(seq, "", 400).items()}
    P["0.running_mean"] = np.zeros(5)
    P["0.running_var"] = np.ones(5)
    P["0.num_batches_tracked"] = np.array(0)
    x = (syn.synthetic_eps((3, 5, 6, 7), seed=401) * 1.7 + 0.3).astype(np.float64)
    y = seq.forward(x, P, "", True)
    dy = syn.synthetic_eps(y.shape, seed=402).astype(np.float64)
    g = {}
    dx = seq.backward(dy, g, "")
    check("bn_relu/y", y, golden_ops, 1e-5)
    check("bn_relu/dx", dx, golden_ops, 1e-5)
    check("bn_relu/d_0.weight", g["0.weight"], golden_ops, 1e-5)
    check("bn_relu/d_0.bias", g["0.bias"], golden_ops, 1e-5)
    check("bn_relu/running_mean", P["0.running_mean"], golden_ops, 1e-6)
    check("bn_relu/running_var", P["0.running_var"], golden_ops, 1e-6)
    check("bn_relu/y_eval", seq.forward(x, P, "", False), golden_ops, 1e-5)


@pytest.mark.parametrize("layer", [("prelu",), ("softplus",), ("tanh",), ("sigmoid",), ("Leaky ReLU", 0.2)])
def test_activations(layer, golden_ops):
    tag = layer[0].lower().replace(" ", "_")
    seq = O.Sequential([layer])
    P = {"0.weight": np.array([0.25])}
    x = (syn.synthetic_eps((2, 3, 5, 5), seed=410) * 8.0)
    x[0, 0, 0, 0] = 25.0
    x = x.astype(np.float64)
    y = seq.forward(x, P, "", True)
    dy = syn.synthetic_eps(y.shape, seed=411).astype(np.float64)
    g = {}
    dx = seq.backward(dy, g, "")
    check(f"act_{tag}/y", y, golden_ops, 1e-6)
    check(f"act_{tag}/dx", dx, golden_ops, 1e-6)
    if tag == "prelu":
        check("act_prelu/d_0.weight", g["0.weight"], golden_ops, 1e-5)


def test_residual_block(golden_ops):
    seq = O.Sequential([("residual block", A.res_block(8))])
    P = {k: v.astype(np.float64) for k, v in _fill(seq, "", 420).items()}
    for k in ("0.res_block.1.", "0.res_block.4."):
        P[k + "running_mean"], P[k + "running_var"] = np.zeros(8), np.ones(8)
        P[k + "num_batches_tracked"] = np.array(0)
    x = syn.synthetic_eps((2, 8, 6, 6), seed=421).astype(np.float64)
    y = seq.forward(x, P, "", True)
    dy = syn.synthetic_eps(y.shape, seed=422).astype(np.float64)
    g = {}
    dx = seq.backward(dy, g, "")
    check("resblock/y", y, golden_ops, 2e-5)
    check("resblock/dx", dx, golden_ops, 5e-5)
    for k, v in g.items():
        check(f"resblock/d_{k}", v, golden_ops, 5e-5)


def test_merge_aux_label(golden_ops):
    yv = syn.synthetic_eps((3, 1, 4, 5), seed=430)
    aux = np.array([0.0, 0.5, 2.0], np.float32)
    got = ops.merge_aux_label(yv, aux)
    assert np.array_equal(got, golden_ops["merge_aux/out/full"].reshape(got.shape))
    with pytest.raises(ValueError):
        ops.merge_aux_label(yv, np.zeros(2, np.float32))


def test_fiducial_architecture_matches_reference_file(golden_model):
    assert repr(A.fiducial_architecture(512)) == str(golden_model["fiducial_arch_repr"])


CASES = [("fid64_n3", 64, 3, False, None), ("fid128_n2", 128, 2, False, None),
         ("twohead64_n2", 64, 2, True, 0.3)]


@pytest.mark.parametrize("tag,size,n,two,alpha", CASES)
def test_full_model_matches_reference(tag, size, n, two, alpha, golden_model):
    arch = A.fiducial_architecture(size, predict_var=two)
    m = O.CVAEOracle(arch, dtype=np.float64)
    shapes = m.param_shapes()
    assert ",".join(k for k in str(golden_model[f"{tag}/state_keys"]).split(",")
                    if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))) == ",".join(shapes)
    assert sum(int(np.prod(s)) for s in shapes.values()) == int(golden_model[f"{tag}/n_params"])
    m.load_params(syn.fill_params(shapes, 7))
    if alpha is not None:
        m.alpha_var = alpha
    x, y, aux = syn.synthetic_batch(n, size, size, seed=1234)
    eps = syn.synthetic_eps((1, n, *arch["dim_z"]), seed=99)
    m.forward(x, y, aux, eps)
    check(f"{tag}/stats", np.array(m.get_stats()), golden_model, 2e-5)
    check(f"{tag}/x_mu", m.x_mu, golden_model, 5e-5)
    check(f"{tag}/z_mu", m.z_mu, golden_model, 5e-5)
    check(f"{tag}/z_log_var", m.z_log_var, golden_model, 5e-5)
    g = m.backward(seed=-1.0)          # the reference back-propagates -ELBO
    for k in shapes:
        check(f"{tag}/grad/{k}", g[k], golden_model, 1e-3, what="grad ")   # fp32 reference rounding
    for k in m.buffer_shapes():
        check(f"{tag}/buf/{k}", m.P[k], golden_model, 2e-5)
    m.training = False
    eps1 = syn.synthetic_eps((1, n, *arch["dim_z"]), seed=100)
    check(f"{tag}/sample_P_eval", m.sample_P(y, aux, eps=eps1), golden_model, 5e-5)
    zfix = syn.synthetic_eps((n, *arch["dim_z"]), seed=101)
    check(f"{tag}/sample_P_eval_zfix", m.sample_P(y, aux, z=zfix), golden_model, 5e-5)
    if two:
        _, var = m.sample_P(y, aux, z=zfix, return_var=True)
        check(f"{tag}/sample_P_eval_var", var, golden_model, 5e-5)


@pytest.mark.parametrize("tag,size,n,two,alpha", CASES[:1] + CASES[2:] + [("fid256_n4", 256, 4, False, None)])
def test_torch_cpu_restatement_matches_reference(tag, size, n, two, alpha, golden_model):
    """oracle/torch_ref.py (the timed CPU baseline of bench.py) against the same fixtures; fid256_n4 is the geometry
    of BASELINE.json configs[0] (batch 4 of 256x256 tiles, dim_z 1x8x8)."""
    import torch
    from oracle.torch_ref import TorchRefCVAE
    arch = A.fiducial_architecture(size, predict_var=two)
    shapes = O.CVAEOracle(arch).param_shapes()
    m = TorchRefCVAE(arch, syn.fill_params(shapes, 7))
    if alpha is not None:
        m.alpha_var = alpha
    x, y, aux = syn.synthetic_batch(n, size, size, seed=1234)
    eps = syn.synthetic_eps((1, n, *arch["dim_z"]), seed=99)
    elbo = m.forward(x, y, aux, eps)
    (-elbo).backward()
    check(f"{tag}/stats", np.array(m.get_stats()), golden_model, 1e-6)
    check(f"{tag}/x_mu", m.x_mu.detach().numpy(), golden_model, 1e-6)
    for k in shapes:
        # (at 256^2 the latent-path gradients move by 1e-3..1e-2 with the thread count of the host: fixtures'
        #  grad_variant_dist; the same arithmetic on another core count is not bit-equal)
        check(f"{tag}/grad/{k}", m.P[k].grad.numpy(), golden_model, 1e-5 if size <= 128 else 5e-2, what="grad ")
    m.training = False
    zfix = syn.synthetic_eps((n, *arch["dim_z"]), seed=101)
    s = m.sample_P(y, aux, z=zfix).numpy()
    check(f"{tag}/sample_P_eval_zfix", s, golden_model, 1e-6)
    if f"{tag}/x_mu/crop_tl" in golden_model:
        from golden_util import crop_rel_l2
        assert crop_rel_l2(f"{tag}/x_mu", m.x_mu.detach().numpy(), golden_model) <= 1e-6
        assert crop_rel_l2(f"{tag}/sample_P_eval_zfix", s, golden_model) <= 1e-6


def test_oracle_on_softened_case(golden_model_r3):
    """The float64 NumPy oracle against the reference's fp32 run of the WELL-CONDITIONED case (every ReLU softened to
    LeakyReLU(0.9): tests/golden/make_goldens_r3.py).  Without ReLU-flip noise the reference's fp32 gradients and
    the oracle's float64 ones agree to 1e-4 of each tensor's scale (the ReLU cases above: 1e-3)."""
    gold, tag, size, n = golden_model_r3, "soft128_n4", 128, 4
    arch = syn.softened_architecture(A.fiducial_architecture(size), 0.9)
    m = O.CVAEOracle(arch, dtype=np.float64)
    shapes = m.param_shapes()
    m.load_params(syn.soften_params(syn.fill_params(shapes, 7), 0.9))
    x, y, aux = syn.synthetic_batch(n, size, size, seed=1234)
    eps = syn.synthetic_eps((1, n, *arch["dim_z"]), seed=99)
    m.forward(x, y, aux, eps)
    check(f"{tag}/stats", np.array(m.get_stats()), gold, 2e-5)
    check(f"{tag}/x_mu", m.x_mu, gold, 5e-5)
    g = m.backward(seed=-1.0)
    for k in shapes:
        check(f"{tag}/grad/{k}", g[k], gold, 1e-4, what="grad ")
        check(f"{tag}/grad64/{k}", g[k], gold, 1e-6, what="grad64 ")       # (the stored truth IS this oracle's)
    for k in m.buffer_shapes():
        check(f"{tag}/buf/{k}", m.P[k], gold, 2e-5)


@pytest.mark.parametrize("tag,alpha", [("twohead512_n2", 0.3), ("twohead512_n2_a1", 1.0)])
def test_torch_cpu_restatement_two_heads_at_512(tag, alpha, golden_model_r3):
    """oracle/torch_ref.py on the training script's two-head network at its real geometry against the reference's run
    (forward quantities to fp32 rounding; gradients as in the other >= 256^2 cases)."""
    from golden_util import crop_rel_l2
    from oracle.torch_ref import TorchRefCVAE
    gold = golden_model_r3
    arch = A.fiducial_architecture(512, predict_var=True)
    shapes = O.CVAEOracle(arch).param_shapes()
    m = TorchRefCVAE(arch, syn.fill_params(shapes, 7))
    m.alpha_var = alpha
    x, y, aux = syn.synthetic_batch(2, 512, 512, seed=1234)
    eps = syn.synthetic_eps((1, 2, *arch["dim_z"]), seed=99)
    elbo = m.forward(x, y, aux, eps)
    (-elbo).backward()
    check(f"{tag}/stats", np.array(m.get_stats()), gold, 1e-6)
    assert crop_rel_l2(f"{tag}/x_mu", m.x_mu.detach().numpy(), gold) <= 1e-6
    for k in shapes:
        check(f"{tag}/grad/{k}", m.P[k].grad.numpy(), gold, 5e-2, what="grad ")


def test_conditioning_fixture_is_what_its_script_makes():
    """tests/golden/cond.npz (the float32 noise floor the GPU gradient tests use) against a fresh evaluation of one
    draw at 128^2 by tests/golden/make_goldens_cond.py: the fixture is data of the committed script, and the true
    gradient of this case really jumps by ~1e-2 under an 8-ulp perturbation (a latent-level ReLU sits at zero)."""
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_goldens_cond as mc
    from golden_util import distance_arrays
    cond = np.load(os.path.join(here, "golden", "cond.npz"))
    assert float(cond["delta"]) == mc.DELTA and int(cond["draws"]) == mc.DRAWS
    for tag, _, _ in mc.CASES:
        assert cond[f"{tag}/grad_cond_dist"].shape[0] == mc.DRAWS
    arch = syn.scaled_architecture(A.fiducial_architecture(512), 128)
    g0 = mc.gradient(arch, 2, 128, 0.0, 0)
    g1 = mc.gradient(arch, 2, 128, mc.DELTA, 1)
    names = str(cond["fid128_n2/grad_cond_params"]).split(",")
    assert names == list(g0)
    fresh = np.array([distance_arrays(g1[k], g0[k]) for k in names])
    stored = cond["fid128_n2/grad_cond_dist"][1]
    assert np.allclose(fresh, stored, rtol=1e-3, atol=1e-7)
    assert 1e-3 < stored.max() < 1e-1
