"""GPU parity of the whole CVAE path (forward, losses, every parameter gradient, batch-norm
running statistics, sample_P) against the golden fixtures generated from the reference and
against the float64 NumPy oracle.

Stated fp32 tolerances (relative to each tensor's scale): losses 2e-5 vs goldens, x_mu / samples
1e-4 rel-L2 class, gradients 5e-3 vs the reference's own fp32 results (1e-3 is the spread between
the fp32 reference and the float64 oracle; the recognition-net gradients are ill-conditioned:
they flow through batch-norm cancellations of the tiny KL/latent terms) and 2e-3 vs the oracle."""
import os

import numpy as np
import pytest
import torch

from baryon_painter_amd.models import arch as A
from baryon_painter_amd.utils import synthetic as syn
from golden_util import check, crop_rel_l2, distance, summary_distance
from oracle.cvae_oracle import CVAEOracle

import gpu_util as G

pytestmark = pytest.mark.gpu

CASES = [("fid64_n3", 64, 3, False, None), ("fid128_n2", 128, 2, False, None),
         ("twohead64_n2", 64, 2, True, 0.3), ("fid256_n4", 256, 4, False, None)]


_COND = None


def conditioning(tag, k):
    """How far the float64 TRUE gradient itself moves under an 8-ulp (float32) perturbation of the parameters
    (tests/golden/cond.npz, made by tests/golden/make_goldens_cond.py with the float64 oracle): where a latent-level
    pre-activation sits within rounding of zero the gradient is discontinuous there, and NO float32 evaluation can be
    expected closer to the truth than that jump.  0 when the fixture has no entry."""
    global _COND
    if _COND is None:
        _COND = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cond.npz"))
    if f"{tag}/grad_cond_dist" not in _COND:
        return 0.0
    names = str(_COND[f"{tag}/grad_cond_params"]).split(",")
    return float(_COND[f"{tag}/grad_cond_dist"][:, names.index(k)].max())


def reference_noise_floor(tag, k, gold):
    """The float32 noise floor of one gradient: the largest of
      * how far the REFERENCE's own fp32 gradient lies from the float64 truth over the executions of the reference
        stored in the fixtures -- the golden run (8 threads, oneDNN) and the variants of tests/golden/make_goldens.py
        REF_VARIANTS (1 / 3 threads, oneDNN off), which change nothing but the summation order inside ATen;
      * the conditioning of the true gradient (``conditioning``).
    Many of these gradients are sums with 1000:1 cancellation behind ReLU / PReLU masks that flip with the last bit
    of the forward pass; the reference's result moves by factors of 3-10 between such executions (measured:
    p_z_in.7.bias at 512^2 is 1.2e-3 ... 3.0e-3 from the truth in the build container and 8e-3 on the GPU box's
    host; the truth itself moves by 1.3e-2 under an 8-ulp perturbation; DESIGN.md "Numerical parity").  The
    reference's executions share most of their rounding (same ATen kernels), so their spread alone understates what
    an independent float32 implementation sees.  No per-tensor exemptions."""
    errs = [summary_distance(f"{tag}/grad/{k}", f"{tag}/grad64/{k}", gold), conditioning(tag, k)]
    if f"{tag}/grad_variant_dist" in gold:
        names = str(gold[f"{tag}/grad_variant_params"]).split(",")
        errs += list(gold[f"{tag}/grad_variant_dist"][:, names.index(k)])
    return float(max(errs))


def _check_grad(tag, k, grad, gold):
    """Gradient criterion.  Where the fixtures hold the float64 true value (grad64), the HIP gradient may be at most
    4x as far from it as the float32 noise floor of that gradient (``reference_noise_floor``), and never needs to be
    closer than 2e-3 of the tensor's scale (the floor of the fp32 comparison everywhere else in this suite; a wrong
    kernel shows up as >= 1e-1 on the well-conditioned gradients, which are most of them).  Otherwise 5e-3 against
    the fp32 reference."""
    if f"{tag}/grad64/{k}/shape" in gold:
        ref_err = reference_noise_floor(tag, k, gold)
        ours = distance(f"{tag}/grad64/{k}", grad, gold)
        assert ours <= max(4 * ref_err, 2e-3), f"grad {k}: {ours:.2e} from truth, noise floor {ref_err:.2e}"
    else:
        check(f"{tag}/grad/{k}", grad, gold, 5e-3, what="grad ")


def _model(arch, impl="auto"):
    from baryon_painter_amd.models.cvae import CVAE
    m = CVAE(arch, "cuda:0", impl=impl)
    shapes = {k: tuple(p.shape) for k, p in m.named_parameters()}
    P = syn.fill_params(shapes, 7)
    with torch.no_grad():
        for k, p in m.named_parameters():
            p.copy_(torch.from_numpy(P[k]))
    return m, P


@pytest.mark.parametrize("impl", ["mfma", "direct"])
@pytest.mark.parametrize("tag,size,n,two,alpha", CASES)
def test_model_matches_reference_goldens(tag, size, n, two, alpha, impl, golden_model):
    if impl == "direct" and size > 128:
        pytest.skip("the vector-ALU second-opinion kernels are exercised at 64^2 / 128^2")
    arch = A.fiducial_architecture(size, predict_var=two)
    m, P = _model(arch, impl)
    assert ",".join(m.state_dict().keys()) == str(golden_model[f"{tag}/state_keys"])
    assert m.count_parameters() == int(golden_model[f"{tag}/n_params"])
    assert ",".join(m.get_stats_labels()) == str(golden_model[f"{tag}/stats_labels"])
    if alpha is not None:
        m.alpha_var = alpha
    x, y, aux = syn.synthetic_batch(n, size, size, seed=1234)
    eps = syn.synthetic_eps((1, n, *arch["dim_z"]), seed=99)
    m._eps_override = eps
    m.train(True)
    elbo = m(torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(aux))
    (-elbo).backward()
    torch.cuda.synchronize()
    check(f"{tag}/stats", np.array(m.get_stats()), golden_model, 2e-5)
    check(f"{tag}/x_mu", m.x_mu.cpu().numpy(), golden_model, 1e-4)
    check(f"{tag}/z_mu", m.z_mu.cpu().numpy(), golden_model, 1e-4)
    check(f"{tag}/z_log_var", m.z_log_var.cpu().numpy(), golden_model, 1e-4)
    if f"{tag}/x_mu/crop_tl" in golden_model:
        assert crop_rel_l2(f"{tag}/x_mu", m.x_mu.cpu().numpy(), golden_model) <= 1e-4
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        _check_grad(tag, k, p.grad.cpu().numpy(), golden_model)
    for k, b in m.named_buffers():
        check(f"{tag}/buf/{k}", b.cpu().numpy(), golden_model, 2e-5)
    if size <= 128 and impl == "mfma":
        # float64 NumPy oracle in-process, every gradient in full (the fixtures hold summaries of the large ones).
        # (Run for the matrix-core kernels only: the oracle passes are the slow part of this test.)
        ora = CVAEOracle(arch, dtype=np.float64)
        ora.load_params(P)
        if alpha is not None:
            ora.alpha_var = alpha
        ora.forward(x, y, aux, eps)
        g = ora.backward(seed=-1.0)
        # The float32 noise floor of every gradient IN FULL (the fixtures hold it for summaries): how far the true
        # gradient itself moves under the 8-ulp parameter perturbations of tests/golden/make_goldens_cond.py.  One
        # ReLU unit of ~1.5 million sits within 1e-6 of zero in any such case, and flipping it moves these (spiky,
        # two-tile) gradients by up to 1e-2; which side a float32 evaluation lands on is chance.
        floor = {k: reference_noise_floor(tag, k, golden_model) if f"{tag}/grad64/{k}/shape" in golden_model else 0.0
                 for k in g}
        # (... for the two-head cases too: their fixtures hold no conditioning draws, and a kernel that merely adds in a
        #  different order -- the MFMA form of the recognition / prior stems -- moved p_z_in.7.bias by 1.5 of the old
        #  limit at 64^2 through one such unit)
        from golden import make_goldens_cond as mc
        for draw in range(mc.DRAWS):
            gd = mc.gradient(arch, n, size, mc.DELTA, draw, alpha=alpha)
            for k in g:
                floor[k] = max(floor[k], G.rel_err(gd[k], g[k]))
        errs = sorted(((G.rel_err(p.grad.cpu().numpy(), g[k]) / max(4 * floor[k], 2e-3), k)
                       for k, p in m.named_parameters()), reverse=True)
        print("worst gradient errors vs float64 oracle, in units of max(4 x noise floor, 2e-3):", errs[:5])
        assert errs[0][0] < 1.0, errs[:5]
    # paint-style sampling in eval mode (running statistics)
    m.train(False)
    m._eps_override = syn.synthetic_eps((1, n, *arch["dim_z"]), seed=100)
    s = m.sample_P(torch.from_numpy(y), aux_label=torch.from_numpy(aux))
    check(f"{tag}/sample_P_eval", s.cpu().numpy(), golden_model, 1e-4)
    zfix = syn.synthetic_eps((n, *arch["dim_z"]), seed=101)
    s = m.sample_P(torch.from_numpy(y), aux_label=torch.from_numpy(aux), z=zfix)
    check(f"{tag}/sample_P_eval_zfix", s.cpu().numpy(), golden_model, 1e-4)
    if f"{tag}/sample_P_eval_zfix/crop_tl" in golden_model:
        assert crop_rel_l2(f"{tag}/sample_P_eval_zfix", s.cpu().numpy(), golden_model) <= 1e-4
    if two:
        _, var = m.sample_P(torch.from_numpy(y), aux_label=torch.from_numpy(aux), z=zfix, return_var=True)
        check(f"{tag}/sample_P_eval_var", var.cpu().numpy(), golden_model, 1e-4)


def test_fiducial_512_matches_reference_goldens(golden_model):
    """BASELINE config geometry (512x512 tiles), N=2: scalars, x_mu and every gradient."""
    arch = A.fiducial_architecture(512)
    m, P = _model(arch)
    x, y, aux = syn.synthetic_batch(2, 512, 512, seed=1234)
    m._eps_override = syn.synthetic_eps((1, 2, *arch["dim_z"]), seed=99)
    elbo = m(torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(aux))
    (-elbo).backward()
    tag = "fid512_n2"
    check(f"{tag}/stats", np.array(m.get_stats()), golden_model, 2e-5)
    check(f"{tag}/x_mu", m.x_mu.cpu().numpy(), golden_model, 1e-4)
    # north_star: "within 1e-4 rel-L2 of reference" as a true ||a-b||/||b|| over full-resolution crops
    assert crop_rel_l2(f"{tag}/x_mu", m.x_mu.cpu().numpy(), golden_model) <= 1e-4
    for k, p in m.named_parameters():
        _check_grad(tag, k, p.grad.cpu().numpy(), golden_model)
    for k, b in m.named_buffers():
        check(f"{tag}/buf/{k}", b.cpu().numpy(), golden_model, 2e-5)
    m.train(False)
    yt, at = torch.from_numpy(y), torch.from_numpy(aux)
    zfix = syn.synthetic_eps((2, *arch["dim_z"]), seed=101)
    s = m.sample_P(yt, aux_label=at, z=zfix)
    check(f"{tag}/sample_P_eval_zfix", s.cpu().numpy(), golden_model, 1e-4)
    assert crop_rel_l2(f"{tag}/sample_P_eval_zfix", s.cpu().numpy(), golden_model) <= 1e-4
    m._eps_override = syn.synthetic_eps((1, 2, *arch["dim_z"]), seed=100)
    s = m.sample_P(yt, aux_label=at)
    check(f"{tag}/sample_P_eval", s.cpu().numpy(), golden_model, 1e-4)
    assert crop_rel_l2(f"{tag}/sample_P_eval", s.cpu().numpy(), golden_model) <= 1e-4
    # the hipGraph-captured paint forward (the path bench.py's paint leg times) against the same golden
    m._eps_override = None
    s = m.sample_P_graphed(yt, aux_label=at, z=zfix)
    check(f"{tag}/sample_P_eval_zfix", s.cpu().numpy(), golden_model, 1e-4)
    assert crop_rel_l2(f"{tag}/sample_P_eval_zfix", s.cpu().numpy(), golden_model) <= 1e-4


def test_adam_step_matches_reference(golden_model):
    arch = A.fiducial_architecture(64)
    m, P = _model(arch)
    x, y, aux = syn.synthetic_batch(3, 64, 64, seed=1234)
    m._eps_override = syn.synthetic_eps((1, 3, *arch["dim_z"]), seed=99)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    elbo = m(torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(aux))
    opt.zero_grad()
    (-elbo).backward()
    # the golden's Adam step was taken after two further eval-mode forwards, which do not touch
    # parameters; BN buffers are not part of this comparison
    opt.step()
    bad = []
    for k, p in m.named_parameters():
        try:
            check(f"fid64_n3/adam/{k}", p.detach().cpu().numpy(), golden_model, 2e-3)
        except AssertionError as e:      # Adam's first step is +-lr*sign(g): tiny gradients may flip
            bad.append(str(e))
    assert len(bad) <= 2, bad


def test_flat_adam_equals_torch_adam():
    """FlatAdam (one fused kernel over the flat parameter buffer) against torch.optim.Adam on the same gradients:
    after ONE step from the same state the parameters agree to float32 rounding (the arithmetic is the same); over
    three steps with a decaying learning rate the two trajectories stay together to a fraction of a step (they are
    not bitwise comparable: a last-bit difference after step one moves the next gradient through the ReLU masks,
    see ``conditioning``, and Adam normalises every gradient to a step of order lr)."""
    from baryon_painter_amd.optim import FlatAdam
    arch = A.fiducial_architecture(64)
    x, y, aux = syn.synthetic_batch(2, 64, 64, seed=3)
    eps = syn.synthetic_eps((1, 2, *arch["dim_z"]), seed=4)
    finals, first = [], []
    for kind in ("torch", "flat"):
        m, _ = _model(arch)
        m._eps_override = eps
        opt = torch.optim.Adam(m.parameters(), lr=1e-3) if kind == "torch" else FlatAdam(m, lr=1e-3)
        sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda e: 0.5 ** e)
        for it in range(3):
            elbo = m(torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(aux))
            opt.zero_grad()
            (-elbo).backward()
            opt.step()
            sched.step()
            if it == 0:
                first.append((float(elbo.detach()), m._flat_params.clone()))
        finals.append((float(elbo.detach()), m._flat_params.clone()))
    assert first[0][0] == first[1][0]
    assert G.rel_err(first[1][1].cpu().numpy(), first[0][1].cpu().numpy()) < 1e-6
    assert abs(finals[0][0] - finals[1][0]) <= 1e-4 * abs(finals[0][0])
    # Adam moves a parameter by ~lr per step whatever the size of its gradient, so a parameter whose gradient is at the
    # rounding floor may take steps two and three (0.5e-3 + 0.25e-3) in opposite directions in the two runs: 1.5e-3 is
    # the bound for any single parameter, and all but a handful stay within half of the first (largest) step
    d = (finals[1][1] - finals[0][1]).abs()
    assert d.max().item() < 1.5e-3
    assert (d > 0.5e-3).float().mean().item() < 1e-4


def test_forward_is_deterministic_and_shape_checked():
    arch = A.fiducial_architecture(64)
    m, _ = _model(arch)
    x, y, aux = syn.synthetic_batch(2, 64, 64, seed=3)
    m._eps_override = syn.synthetic_eps((1, 2, *arch["dim_z"]), seed=4)
    with torch.no_grad():
        a = float(m(torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(aux)))
        st = m.state_dict()
        b = float(m(torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(aux)))
    assert a == b                                    # bitwise: no float atomics anywhere
    with pytest.raises(ValueError):
        m(torch.from_numpy(x[:, :, :32]), torch.from_numpy(y), torch.from_numpy(aux))
    with pytest.raises(ValueError):
        m(torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(aux[:1]))


def test_backward_is_bitwise_reproducible_with_overlapped_streams():
    """Weight gradients run on a second stream and the q_y_in / prior branches on their own streams: repeating the
    same step from the same state must give the same bits every time (a workspace or buffer shared across
    streams would show up here), and the same bits as the single-stream schedule."""
    arch = A.fiducial_architecture(128)
    m, _ = _model(arch)
    x, y, aux = [torch.from_numpy(t) for t in syn.synthetic_batch(4, 128, 128, seed=5)]
    m._eps_override = syn.synthetic_eps((1, 4, *arch["dim_z"]), seed=6)
    state = {k: v.clone() for k, v in m.state_dict().items()}

    def grads():
        m.load_state_dict(state)
        m._bump_param_versions()
        m.zero_grad()
        (-m(x, y, aux)).backward()
        return m._flat_grads.clone()

    ref = grads()
    assert m._last.side is not None and m._last.branch is not None, "the overlapped schedule is the default"
    for _ in range(5):
        assert torch.equal(grads(), ref)
    m.overlap_weight_gradients(False)
    assert torch.equal(grads(), ref)
    m.overlap_weight_gradients(True)


@pytest.mark.parametrize("n,L,train", [(1, 1, True), (5, 1, True), (2, 2, True), (3, 1, False)])
def test_batch_sizes_L_and_eval_mode_against_oracle(n, L, train):
    """Shapes the fixtures do not cover (batch 1, odd batch, L = 2 samples per datum, eval-mode
    forward) against the float64 NumPy oracle."""
    arch = A.fiducial_architecture(64)
    arch["L"] = L
    m, P = _model(arch)
    ora = CVAEOracle(arch, dtype=np.float64)
    ora.load_params(P)
    x, y, aux = syn.synthetic_batch(n, 64, 64, seed=40 + n)
    eps = syn.synthetic_eps((L, n, *arch["dim_z"]), seed=41)
    if not train:      # give the running statistics non-trivial values first (one train-mode forward each)
        m._eps_override = eps
        with torch.no_grad():
            m(torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(aux))
        ora.forward(x, y, aux, eps)
    m.train(train)
    ora.training = train
    m._eps_override = eps
    elbo = m(torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(aux))
    ora.forward(x, y, aux, eps)
    got, ref = np.array(m.get_stats()), np.array(ora.get_stats())
    assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max(), (got, ref)
    assert G.rel_err(m.x_mu.cpu().numpy(), ora.x_mu) < 1e-4
    if not train:
        with pytest.raises(RuntimeError, match="eval-mode"):
            (-elbo).backward()
        return
    (-elbo).backward()
    g = ora.backward(seed=-1.0)
    # float32 noise floor of these gradients: how far the TRUE gradient moves when the parameters are perturbed by
    # 2^-20 relative -- the size of the difference between any float32 forward pass and the float64 one (activations
    # agree to ~1e-6 of their scale).  At batch 1 a single ReLU unit at zero moves these gradients by percents (see
    # ``conditioning``); which side of it a float32 evaluation lands on changes with the summation order of any kernel.
    floor = {k: 0.0 for k in g}
    rng = np.random.default_rng(7)
    for _ in range(4):
        pert = CVAEOracle(arch, dtype=np.float64)
        pert.load_params({k: np.asarray(v, np.float64) * (1.0 + 2.0 ** -20 * rng.uniform(-1, 1, np.shape(v)))
                          for k, v in P.items()})
        pert.forward(x, y, aux, eps)
        gp = pert.backward(seed=-1.0)
        for k in g:
            floor[k] = max(floor[k], G.rel_err(gp[k], g[k]))
    errs = sorted(((G.rel_err(p.grad.cpu().numpy(), g[k]) / max(4 * floor[k], 5e-3), k) for k, p in m.named_parameters()),
                  reverse=True)
    assert errs[0][0] < 1.0, errs[:4]
