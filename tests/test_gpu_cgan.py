"""GPU: CGAN (generator + spectrally-normalised PatchGAN discriminator, alternating step) on the HIP
kernels vs this repository's own torch CPU restatement (oracle/cgan_torch.py).  The reference has no
CGAN code, so this parity is "vs. own restatement" (unpinned)."""
import numpy as np
import pytest
import torch

from baryon_painter_amd.utils import synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("size,n_res,n", [(64, 2, 2), (512, 9, 2)])
def test_cgan_iteration_matches_torch_restatement(size, n_res, n):
    """(64, 2 blocks): quick case.  (512, 9 blocks): the fiducial CGAN of BASELINE.json configs[2] at its real
    geometry -- the k9 stem / head at 512^2, nine residual blocks at 128^2, the 256- and 512-channel PatchGAN
    layers -- one alternating D + G iteration, every loss and gradient.

    Gradient criterion (the CVAE's, tests/test_gpu_model.py): the restatement evaluated in FLOAT64 is the true value;
    the float32 noise floor of each gradient is how far float32 executions of the same restatement (default / 4 / 2
    threads: nothing but the summation order inside ATen changes) land from it, or how far the true gradient itself
    moves under an 8-ulp perturbation of the parameters, whichever is larger; the HIP gradient may be at most 4x
    that far away, and never needs to be closer than 2e-3 of the tensor's scale."""
    from baryon_painter_amd.models.cgan import CGAN
    from oracle.cgan_torch import TorchCGAN
    torch.manual_seed(0)
    m = CGAN(tile_size=size, device="cuda:0", n_res=n_res)
    m.train(True)
    state = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    ref = TorchCGAN(m.g_arch, m.d_arch, state, m.lambda_perceptual)
    x, y, z = syn.synthetic_batch(n, size, size, seed=5)
    x = np.tanh(3 * x - 0.5).astype(np.float32)            # real field in the tanh domain
    opt_g = torch.optim.Adam(m.g_parameters(), lr=5e-5, betas=(0.5, 0.999))
    opt_d = torch.optim.Adam(m.d_parameters(), lr=5e-5, betas=(0.5, 0.999))
    cap = {}
    losses = m.train_step(torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(z), opt_g, opt_d, capture=cap)
    rl, gd, gg, fake = ref.iteration(x, y, z)
    for k in ("D", "G_adv", "G_perceptual"):
        assert abs(float(losses[k]) - rl[k]) <= 2e-5 * max(1.0, abs(rl[k])), (k, float(losses[k]), rl[k])
    truth = TorchCGAN(m.g_arch, m.d_arch, state, m.lambda_perceptual, dtype=torch.float64)
    _, td, tg, _ = truth.iteration(x, y, z)
    variants = [(gd, gg)]
    threads = torch.get_num_threads()
    for t in (4, 2):
        torch.set_num_threads(t)
        v = TorchCGAN(m.g_arch, m.d_arch, state, m.lambda_perceptual)
        _, vd, vg, _ = v.iteration(x, y, z)
        variants.append((vd, vg))
    torch.set_num_threads(threads)
    # conditioning of the true gradient (as tests/golden/make_goldens_cond.py does for the CVAE): how far the FLOAT64
    # gradient itself moves when every parameter is perturbed by 8 ulp of float32 -- LeakyReLU units within rounding
    # of zero make it discontinuous there, and any float32 evaluation lands on either side by chance
    cond = []
    gen = torch.Generator().manual_seed(11)
    for _ in range(2):
        pert = {k: (v.double() * (1.0 + 2.0 ** -21 * (2 * torch.rand(v.shape, generator=gen, dtype=torch.float64) - 1))
                    if v.dtype.is_floating_point and not k.endswith(("running_mean", "running_var", "weight_u", "weight_v"))
                    else v) for k, v in state.items()}
        c = TorchCGAN(m.g_arch, m.d_arch, pert, m.lambda_perceptual, dtype=torch.float64)
        _, cd, cg, _ = c.iteration(x, y, z)
        cond.append((cd, cg))
    errs = []
    for idx, (net, got, want) in enumerate((("discriminator.", cap["d"], td), ("generator.", cap["g"], tg))):
        # a conv bias in front of a batch-norm has an exactly-zero gradient (rounding noise on both
        # sides): errors are measured against the larger of the tensor's and 1e-4 of the net's scale
        scale0 = 1e-4 * max(float(v.abs().max()) for v in want.values() if v.numel() > 1)

        def dist(a, w):
            return float((a.cpu().double() - w).abs().max() / max(float(w.abs().max()), scale0))
        for k, g in got.items():
            w = want[net + k].double()
            floor = max(dist(var[idx][net + k], w) for var in variants + cond)
            errs.append((dist(g, w) / max(4 * floor, 2e-3), dist(g, w), floor, net + k))
    errs.sort(reverse=True)
    print("worst CGAN gradients (distance from float64 / limit, distance, fp32 floor):", errs[:4])
    assert errs[0][0] < 1.0, errs[:6]
    # spectral-norm power-iteration state after the two discriminator forwards
    after = m.state_dict()
    for k, t in ref.P.items():
        if k.endswith(("weight_u", "weight_v")):
            assert torch.allclose(after[k].cpu(), t, atol=1e-5), k
    # generated field
    g = m.generate(torch.from_numpy(y), torch.from_numpy(z))
    assert g.shape == (n, 1, size, size) and torch.isfinite(g).all() and float(g.abs().max()) <= 1.0


_CGAN_DP_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from baryon_painter_amd.dist import Sync
from baryon_painter_amd.models.cgan import CGAN
from baryon_painter_amd.utils import synthetic as syn
dist.init_process_group("gloo")            # 2 ranks share the one GPU of the box; RCCL needs distinct devices
r, w = dist.get_rank(), dist.get_world_size()
size, n = 64, 4
x, y, z = syn.synthetic_batch(n, size, size, seed=5)
x = np.tanh(3 * x - 0.5).astype(np.float32)
def run(sync, sl):
    torch.manual_seed(0)
    m = CGAN(tile_size=size, device="cuda:0", n_res=2, sync=sync)
    m.train(True)
    og = torch.optim.Adam(m.g_parameters(), lr=5e-5, betas=(0.5, 0.999))
    od = torch.optim.Adam(m.d_parameters(), lr=5e-5, betas=(0.5, 0.999))
    cap = {}
    losses = m.train_step(torch.from_numpy(x[sl]), torch.from_numpy(y[sl]), torch.from_numpy(z[sl]), og, od, capture=cap)
    return m, cap, {k: float(v) for k, v in losses.items()}
h = n // w
dp, cap_dp, l_dp = run(Sync(), slice(r * h, (r + 1) * h))
t = torch.tensor([l_dp["D"], l_dp["G_adv"], l_dp["G_perceptual"]], dtype=torch.float64); dist.all_reduce(t); t /= w
if r == 0:
    ref, cap_ref, l_ref = run(None, slice(0, n))
    for v, k in zip(t.tolist(), ("D", "G_adv", "G_perceptual")):
        assert abs(v - l_ref[k]) <= 2e-5 * max(1.0, abs(l_ref[k])), (k, v, l_ref[k])
    errs = []
    for net in ("d", "g"):
        floor = 1e-4 * max(float(v.abs().max()) for v in cap_ref[net].values() if v.numel() > 1)
        for k, g in cap_dp[net].items():
            b = cap_ref[net][k].double()
            errs.append((float((g.double() - b).abs().max() / max(float(b.abs().max()), floor)), net + "." + k))
    errs.sort(reverse=True)
    print("worst:", errs[:5])
    assert errs[0][0] < 5e-4, errs[:5]
    for (k, a), (_, b) in zip(dp.state_dict().items(), ref.state_dict().items()):
        # parameters after the Adam steps; atol = 2 lr: a bias in front of a batch-norm has a pure-noise gradient,
        # whose sign decides Adam's first step
        assert torch.allclose(a.double(), b.double(), rtol=2e-4, atol=1.1e-4), k
    open(os.path.join(sys.argv[2], "dp.ok"), "w").write(str(errs[0][0]))
dist.barrier()
dist.destroy_process_group()
"""


def test_cgan_two_rank_data_parallel_equals_single_device(tmp_path):
    """Sharded batch, generator batch-norm on global statistics, averaged D and G gradient buffers == one device on
    the whole batch (losses, every gradient, parameters after the alternating step)."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "cgan_dp_worker.py"
    script.write_text(_CGAN_DP_WORKER)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), root, str(tmp_path)],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert (tmp_path / "dp.ok").exists()


def test_cgan_painter_api(tmp_path):
    from baryon_painter_amd.painter import CGANPainter
    from baryon_painter_amd.utils import datasets as D
    tile = 64
    ds = D.SyntheticTileDataset(n_sample=16, tile_size=tile, seed=3)
    p = CGANPainter(training_data_set=ds, tile_size=tile, compute_device="cuda:0", n_res=1)
    log = p.train(n_iter=3, batch_size=2)
    assert len(log) == 3 and all(np.isfinite(v) for row in log for v in row.values())
    dm, pr, z = ds.raw_fields(1)
    out = p.paint(dm, z=z)
    assert out.shape == (tile, tile) and np.isfinite(out).all()
    raw = p.paint(dm, z=z, inverse_transform=False)
    assert raw.shape == (1, 1, tile, tile)
    with pytest.raises(ValueError):
        p.paint(dm[:10], z=z)



def test_cgan_gradients_stay_in_the_flat_buffers_after_zero_grad():
    """Data parallelism averages the two flat gradient buffers: every parameter's gradient -- the spectrally normalised
    weights' too, which are finished by torch glue -- must be a view of them after ``zero_grad()`` (set_to_none is
    torch's default) as well."""
    from baryon_painter_amd.models.cgan import CGAN
    torch.manual_seed(0)
    m = CGAN(tile_size=64, device="cuda:0", n_res=1)
    m.train(True)
    x, y, z = syn.synthetic_batch(2, 64, 64, seed=5)
    x = np.tanh(3 * x - 0.5).astype(np.float32)
    opt_g = torch.optim.Adam(m.g_parameters(), lr=5e-5, betas=(0.5, 0.999))
    opt_d = torch.optim.Adam(m.d_parameters(), lr=5e-5, betas=(0.5, 0.999))
    args = (torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(z), opt_g, opt_d)
    m.train_step(*args)
    m.zero_grad()
    opt_g.zero_grad()
    opt_d.zero_grad()
    assert all(p.grad is None for p in m.parameters())
    m.train_step(*args)
    for name, net in (("d", m.discriminator), ("g", m.generator)):
        flat = m._flat[name]
        lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * 4
        for k, p in net.named_parameters():
            assert p.grad is not None and lo <= p.grad.data_ptr() < hi, (name, k)
        assert torch.isfinite(flat).all() and float(flat.abs().sum()) > 0
    sn = m.sn_layers[0]
    assert float(sn.weight_orig.grad.abs().sum()) > 0
