"""GPU: the bf16 throughput mode (CVAE(dtype="bf16"), BASELINE.json configs[3]) against the float64 oracle / the fp32
fixtures generated from the reference.

Stated bf16 tolerances (bf16 = 8 significant bits; every trunk activation and gradient is rounded once when stored):
    losses (ELBO, KL, log-likelihood)   <= 1e-3 relative
    x_mu, sample_P                      <= 1e-2 relative L2
    whole gradient (flat, 1.66 M params) cosine with the fp32 path's gradient >= 0.995, relative L2 <= 0.1
                                         (the fp32 path is pinned to the reference by tests/test_gpu_model.py)
Single tensors are not asserted: many gradients of this network are sums with 1000:1 cancellation (fp32 noise floor
1e-3..1e-2, tests/test_gpu_model.py) that the rounding noise of bf16 gradients dominates at small batch; measured
23 % on one trunk weight tensor at batch 2 while the whole-gradient cosine is 0.9999."""
import numpy as np
import pytest
import torch

from baryon_painter_amd.models import arch as A
from baryon_painter_amd.utils import synthetic as syn
from golden_util import check, crop_rel_l2, distance

pytestmark = pytest.mark.gpu


def _model(arch, dtype):
    from baryon_painter_amd.models.cvae import CVAE
    m = CVAE(arch, "cuda:0", dtype=dtype)
    P = syn.fill_params({k: tuple(p.shape) for k, p in m.named_parameters()}, 7)
    with torch.no_grad():
        for k, p in m.named_parameters():
            p.copy_(torch.from_numpy(P[k]))
    return m, P


@pytest.mark.parametrize("tag,size,n", [("fid128_n2", 128, 2), ("fid512_n2", 512, 2)])
def test_bf16_step_against_reference_fixtures(tag, size, n, golden_model):
    arch = A.fiducial_architecture(size)
    m, P = _model(arch, "bf16")
    x, y, aux = syn.synthetic_batch(n, size, size, seed=1234)
    m._eps_override = syn.synthetic_eps((1, n, *arch["dim_z"]), seed=99)
    elbo = m(torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(aux))
    (-elbo).backward()
    plan = m._last
    assert sum(1 for u in plan.pack_batch.units if u.bf16) >= 16, "the generator trunk must run on the bf16 kernels"
    assert plan.h.buf.dtype == torch.bfloat16
    check(f"{tag}/stats", np.array(m.get_stats()), golden_model, 1e-3)
    check(f"{tag}/x_mu", m.x_mu.cpu().numpy(), golden_model, 1e-2)
    if f"{tag}/x_mu/crop_tl" in golden_model:
        assert crop_rel_l2(f"{tag}/x_mu", m.x_mu.cpu().numpy(), golden_model) <= 1e-2
    for k, b in m.named_buffers():
        check(f"{tag}/buf/{k}", b.cpu().numpy(), golden_model, 5e-3)
    m.train(False)
    zfix = syn.synthetic_eps((n, *arch["dim_z"]), seed=101)
    s = m.sample_P(torch.from_numpy(y), aux_label=torch.from_numpy(aux), z=zfix)
    check(f"{tag}/sample_P_eval_zfix", s.cpu().numpy(), golden_model, 1e-2)
    g = m.sample_P_graphed(torch.from_numpy(y), aux_label=torch.from_numpy(aux), z=zfix)
    assert torch.equal(g, s)


@pytest.mark.parametrize("size,n,steps", [(256, 8, 12), (512, 4, 0)])
def test_bf16_gradient_direction_and_training_progress(size, n, steps):
    """The bf16 gradient points where the fp32 gradient points (cosine over all 1.66 M parameters), and a few Adam
    steps in bf16 lower the loss like the same steps in fp32."""
    from baryon_painter_amd.optim import FlatAdam
    arch = A.fiducial_architecture(size)
    x, y, aux = [torch.from_numpy(t) for t in syn.synthetic_batch(n, size, size, seed=21)]
    eps = syn.synthetic_eps((1, n, *arch["dim_z"]), seed=22)
    flat, curves = {}, {}
    for dt in ("f32", "bf16"):
        m, _ = _model(arch, dt)
        m._eps_override = eps
        elbo = m(x, y, aux)
        (-elbo).backward()
        flat[dt] = m._flat_grads.double().clone()
        opt = FlatAdam(m, lr=1e-3)
        curve = []
        for _ in range(steps):
            elbo = m(x, y, aux)
            opt.zero_grad()
            (-elbo).backward()
            opt.step()
            curve.append(float(elbo.detach()))
        curves[dt] = curve
    cos = float(torch.dot(flat["f32"], flat["bf16"]) / (flat["f32"].norm() * flat["bf16"].norm()))
    rel = float((flat["bf16"] - flat["f32"]).norm() / flat["f32"].norm())
    print("cosine", cos, "relative L2", rel, "curves", curves)
    assert cos >= 0.995 and rel <= 0.1, (cos, rel)
    if not steps:
        return
    f, b = curves["f32"], curves["bf16"]
    assert b[-1] > b[0] and f[-1] > f[0]                                   # the ELBO rises in both
    assert abs(b[-1] - f[-1]) <= 0.05 * abs(f[-1] - f[0]) + 1e-3 * abs(f[-1])   # and by about as much
