"""GPU: the drop-in boundary (CVAEPainter train / validate / paint / checkpoint files) and the
data-parallel arithmetic (2 ranks, global-batch batch-norm) on the HIP path."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from baryon_painter_amd.models import arch as A
from baryon_painter_amd.utils import datasets as D
from baryon_painter_amd.utils import synthetic as syn

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_train_checkpoint_reload_paint(tmp_path):
    from baryon_painter_amd.painter import CVAEPainter
    tile = 64
    train = D.SyntheticTileDataset(n_sample=64, tile_size=tile, seed=1)
    test = D.SyntheticTileDataset(n_sample=16, tile_size=tile, seed=2)
    arch = A.fiducial_architecture(tile, predict_var=True)
    torch.manual_seed(0)
    p = CVAEPainter(training_data_set=train, test_data_set=test, architecture=arch, compute_device="cuda:0")
    out = tmp_path / "run"
    ts, vs = p.train(n_epoch=1, n_pepoch=3, learning_rate=1e-3, batch_size=4,
                     adaptive_batch_size=lambda pe: 4 if pe < 2 else 8,
                     adaptive_learning_rate=lambda pe: 1.0 if pe < 1 else 0.5,
                     pepoch_size=16, validation_pepochs=[0, 1], validation_batch_size=2,
                     validation_loss_frequency=8, validation_loss_batch_size=4, checkpoint_frequency=24,
                     statistics_report_frequency=16, output_path=str(out), verbose=False, show_plots=False)
    assert ts.n_batches >= 8 and vs.n_batches >= 2
    assert all(np.isfinite(v) for v in ts.loss_terms["ELBO"]["all"])
    assert ts.loss_terms["batch_size"]["all"][0] == 4 and ts.loss_terms["batch_size"]["all"][-1] == 8
    assert ts.loss_terms["lr"]["all"][-1] == pytest.approx(5e-4)
    header = (out / "training_stats.txt").read_text().splitlines()[0]
    assert header == ("# Batch nr, sample nr, ELBO, KL_term, log_likelihood_pressure_0, "
                      "log_likelihood_fixed_var_pressure_0, log_likelihood_free_var_pressure_0, lr, batch_size")
    files = sorted(os.listdir(out))
    assert "model_state" in files and "model_meta" in files and "validation_stats.txt" in files
    assert any(f.startswith("checkpoint_sample") and f.endswith("_final_state") for f in files)
    # the loss goes down on the training distribution
    first, last = np.mean(ts.loss_terms["ELBO"]["all"][:3]), np.mean(ts.loss_terms["ELBO"]["all"][-3:])
    assert last > first

    q = CVAEPainter(filename=(str(out / "model_state"), str(out / "model_meta")), compute_device="cuda:0")
    for (k, a), (k2, b) in zip(p.model.state_dict().items(), q.model.state_dict().items()):
        assert k == k2 and torch.equal(a, b), k
    dm, pr, z = test.raw_fields(5)
    eps = syn.synthetic_eps((1, 1, *arch["dim_z"]), seed=8)
    p.input_field, p.label_fields = q.input_field, q.label_fields
    p.transform, p.inverse_transform = q.transform, q.inverse_transform
    p.model._eps_override = q.model._eps_override = eps
    a = p.paint(dm, z=z)
    b = q.paint(dm, z=z)
    assert a.shape == (tile, tile) and np.array_equal(a, b) and np.isfinite(a).all() and (a >= -1e-6).all()
    raw = q.paint(dm, z=z, inverse_transform=False)
    assert raw.shape == (1, 1, tile, tile)
    # batched paint == per-tile paint (same eps, eval-mode batch-norm is per sample)
    q.model._eps_override = np.repeat(eps, 3, axis=1)
    many = q.paint_batch(np.stack([dm, dm, dm]), z=z)
    assert many.shape == (3, tile, tile) and np.allclose(many[1], a, rtol=1e-5, atol=1e-7)
    with pytest.raises(ValueError):
        q.paint(dm[:32], z=z)
    with pytest.raises(ValueError):
        CVAEPainter(filename=str(out / "model_state"))
    with pytest.raises(RuntimeError):
        CVAEPainter(architecture=arch, compute_device="cuda:0").train()


_DP_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from baryon_painter_amd.dist import Sync
from baryon_painter_amd.models import arch as A
from baryon_painter_amd.models.cvae import CVAE
from baryon_painter_amd.utils import synthetic as syn
backend, dtype, early = sys.argv[3], sys.argv[4], sys.argv[5]
r = int(os.environ["RANK"])
# gloo: 2 ranks share the one GPU of the box; nccl (= RCCL) needs one device per rank
dev = "cuda:%d" % (r if backend == "nccl" else 0)
torch.cuda.set_device(dev)
if backend == "nccl":
    dist.init_process_group("nccl")
else:
    dist.init_process_group("gloo")
w = dist.get_world_size()
size, n = 64, 4
arch = A.fiducial_architecture(size)
x, y, aux = syn.synthetic_batch(n, size, size, seed=21)
eps = syn.synthetic_eps((1, n, *arch["dim_z"]), seed=22)
def run(sync, sl):
    m = CVAE(arch, dev, sync=sync, dtype=dtype)
    P = syn.fill_params({k: tuple(p.shape) for k, p in m.named_parameters()}, 7)
    with torch.no_grad():
        for k, p in m.named_parameters(): p.copy_(torch.from_numpy(P[k]))
    m._eps_override = eps[:, sl]
    e = m(torch.from_numpy(x[sl]), torch.from_numpy(y[sl]), torch.from_numpy(aux[sl]))
    (-e).backward()
    torch.cuda.synchronize()
    return m, float(e.detach())
h = n // w
sync = Sync(grad_group="new" if early == "1" else None)
assert sync.overlap == (early == "1")
if early == "1":
    os.environ["BP_EARLY_ALLREDUCE"] = "1"
dp, e_dp = run(sync, slice(r * h, (r + 1) * h))
assert sync.n_grad == (3 if early == "1" else 1), sync.n_grad
# the batch-norm statistics travelled through the peer-memory kernel (csrc/peer_comm.hip), not the process group
if os.environ.get("BP_PEER_SYNC", "1") != "0":
    assert sync.peer is not None and sync.peer.world == w, "peer-memory all-reduce was not set up"
    assert (sync.n_fused > 0) == (os.environ.get("BP_PEER_FUSED", "1") != "0"), sync.n_fused
else:
    assert sync.peer is None and sync.n_fused == 0
sync.check()
t = torch.tensor([e_dp], dtype=torch.float64, device=dev if backend == "nccl" else "cpu"); dist.all_reduce(t); e_mean = t.item() / w
if r == 0:
    ref, e_ref = run(None, slice(0, n))
    assert abs(e_mean - e_ref) <= 2e-6 * abs(e_ref), (e_mean, e_ref)
    errs = []
    for (k, a), (_, b) in zip(dp.named_parameters(), ref.named_parameters()):
        ga, gb = a.grad.double(), b.grad.double()
        errs.append((float((ga - gb).abs().max() / gb.abs().max().clamp_min(1e-30)), k))
    errs.sort(reverse=True)
    print("worst:", errs[:6])
    worst = errs[0][0]
    for (k, a), (_, b) in zip(dp.named_buffers(), ref.named_buffers()):
        assert torch.allclose(a.double(), b.double(), rtol=1e-5, atol=1e-7), k
    # fp32: the sharded sums differ from the single-device ones in summation order only.  bf16: the same holds (every
    # rank rounds the same values to bf16: the statistics are global before any activation is rounded), but the
    # gradients are 1000:1 cancelling sums of bf16-rounded terms whose fp32 partial sums are grouped per rank
    assert worst < (2e-4 if dtype == "f32" else 5e-2), worst
    open(os.path.join(sys.argv[2], "dp.ok"), "w").write(str(worst))
dist.barrier()
dist.destroy_process_group()
"""


def _run_dp_worker(tmp_path, backend, dtype, early, transport="peer"):
    import socket
    script = tmp_path / "dp_worker.py"
    script.write_text(_DP_WORKER)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("BP_EARLY_ALLREDUCE", None)
    # batch-norm statistics: inside the finalize kernels over peer memory (default) / one peer-memory kernel per
    # collective (BP_PEER_FUSED=0) / the process group's all-reduce with the branch networks in lock step (BP_PEER_SYNC=0)
    env.pop("BP_PEER_SYNC", None); env.pop("BP_PEER_FUSED", None)
    if transport == "peer-unfused":
        env["BP_PEER_FUSED"] = "0"
    elif transport == "group":
        env["BP_PEER_SYNC"] = "0"
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), ROOT, str(tmp_path),
                          backend, dtype, early],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert (tmp_path / "dp.ok").exists()


@pytest.mark.parametrize("dtype,early,transport", [("f32", "0", "peer"), ("f32", "1", "peer"), ("bf16", "0", "peer"),
                                                   ("f32", "0", "group"), ("bf16", "0", "peer-unfused")])
def test_two_rank_data_parallel_equals_single_device(tmp_path, dtype, early, transport):
    """Sharded batch + global batch-norm statistics + averaged gradients == the single-device global batch (the
    reference's arithmetic); gloo process group, both ranks on this GPU.  ``early``: the opt-in schedule with the
    trunk's gradients reduced on the weight-gradient stream through a second communicator.  bf16: the data-parallel
    + bf16 combination of BASELINE.json configs[3].  ``transport``: how the statistics travel (see _run_dp_worker): the
    fallbacks stay covered."""
    _run_dp_worker(tmp_path, "gloo", dtype, early, transport)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL wants one device per rank")
@pytest.mark.parametrize("dtype,early", [("f32", "0"), ("bf16", "0"), ("f32", "1")])
def test_two_gpu_rccl_data_parallel(tmp_path, dtype, early):
    """The same check with backend nccl (= RCCL over xGMI) on two GPUs: the run that must pass before
    BP_EARLY_ALLREDUCE=1 (two communicators in flight) may become the default.  Skipped on one-GPU boxes."""
    _run_dp_worker(tmp_path, "nccl", dtype, early)


_RCCL_ONE_RANK_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
os.environ["BP_SYNC_FORCE"] = "1"           # issue every collective although there is one rank
import numpy as np, torch, torch.distributed as dist
from baryon_painter_amd.dist import Sync
from baryon_painter_amd.models import arch as A
from baryon_painter_amd.models.cvae import CVAE
from baryon_painter_amd.utils import synthetic as syn
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
size, n = 64, 4
arch = A.fiducial_architecture(size)
x, y, aux = syn.synthetic_batch(n, size, size, seed=21)
eps = syn.synthetic_eps((1, n, *arch["dim_z"]), seed=22)
def run(sync):
    m = CVAE(arch, "cuda:0", sync=sync)
    P = syn.fill_params({k: tuple(p.shape) for k, p in m.named_parameters()}, 7)
    with torch.no_grad():
        for k, p in m.named_parameters(): p.copy_(torch.from_numpy(P[k]))
    m._eps_override = eps
    e = m(torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(aux))
    (-e).backward()
    torch.cuda.synchronize()
    return m, float(e.detach())
ref, e_ref = run(None)
worst = 0.0
for early in ("0", "1"):
    os.environ["BP_EARLY_ALLREDUCE"] = early
    sync = Sync()
    assert sync.active and sync.overlap == (early == "1") and (sync.grad_group is not sync.group) == (early == "1")
    dp, e_dp = run(sync)
    # 44 batch-norm statistics exchanges per step: separate collectives, or -- over peer memory -- inside the kernels that
    # finalize the statistics (dist.Sync.fused) for the layers whose sums come out of a fused epilogue / reduction
    # (fused: per LAYER -- the lock-step levels of the three branch networks share one separate collective per level, but
    #  each of their layers finalizes its own statistics: 57 in-kernel exchanges + 1 collective instead of 44 collectives)
    fused = sync.peer is not None and os.environ.get("BP_PEER_FUSED", "1") != "0"
    assert sync.n_grad == (3 if early == "1" else 1), sync.n_grad
    assert (sync.n_small, sync.n_fused) == ((1, 57) if fused else (44, 0)), (sync.n_small, sync.n_fused)
    assert (sync.peer is not None) == (os.environ.get("BP_PEER_SYNC", "1") != "0")
    sync.check()
    assert abs(e_dp - e_ref) <= 1e-6 * abs(e_ref), (e_dp, e_ref)
    for (k, a), (_, b) in zip(dp.named_parameters(), ref.named_parameters()):
        worst = max(worst, float((a.grad.double() - b.grad.double()).abs().max() / b.grad.double().abs().max().clamp_min(1e-30)))
    assert worst < 2e-4, worst
open(os.path.join(sys.argv[2], "rccl.ok"), "w").write(str(worst))
dist.destroy_process_group()
"""


def test_data_parallel_schedule_through_rccl_with_one_rank(tmp_path):
    """What a one-GPU box can check of the N > 1 path: with BP_SYNC_FORCE=1 one rank issues every collective of the
    data-parallel step through RCCL (backend nccl): float64 statistics and the flat gradient buffer on one communicator
    (default), then the opt-in schedule with gradient slices on the weight-gradient stream's second communicator; the
    result is the single-device one either way."""
    import socket
    script = tmp_path / "rccl_worker.py"
    script.write_text(_RCCL_ONE_RANK_WORKER)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, str(script), ROOT, str(tmp_path)], capture_output=True, text=True, env=env,
                         timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert (tmp_path / "rccl.ok").exists()


_PEER_WORKER = r"""
import os, sys, time
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from baryon_painter_amd.dist import PeerAllReduce
torch.cuda.set_device(0)
dist.init_process_group("gloo")
r, w = dist.get_rank(), dist.get_world_size()
peer = PeerAllReduce.create(None, torch.device("cuda:0"))
assert peer is not None, "the peer-memory path did not come up"
rng = np.random.default_rng(100 + r)
worst = 0
for it in range(400):
    n = int(np.random.default_rng(it).integers(1, peer.max_doubles + 1))       # the same length on every rank
    mine = torch.from_numpy(rng.standard_normal(n))
    want = mine.clone()
    dist.all_reduce(want)                                                       # gloo, on the host
    t = mine.cuda()
    if (it + r) % 7 == 0:
        time.sleep(0.002 * (1 + r))                                             # uneven arrival
    peer.all_reduce_sum(t)
    got = t.cpu()
    # rank-order sum on the device vs gloo's order: equal to rounding; and bitwise equal across ranks
    assert torch.allclose(got, want, rtol=1e-13, atol=1e-13), (it, n)
    ref = got.clone()
    dist.broadcast(ref, 0)
    assert torch.equal(got, ref), ("ranks disagree", it)
assert peer.timeouts() == 0
if r == 0:
    open(os.path.join(sys.argv[2], "peer.ok"), "w").write("ok")
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("world", [2, 3])
def test_peer_memory_all_reduce_between_processes(tmp_path, world):
    """csrc/peer_comm.hip between `world` processes that share this GPU (IPC-mapped fine-grained buffers, system-scope
    stores / flags / loads): 400 all-reduces of random lengths with ranks arriving unevenly -- the sums equal gloo's to
    rounding and are BITWISE equal on every rank (fixed rank-order sum), no spin ever times out.  What this cannot show
    on a one-GPU box is the xGMI hop itself; the protocol, the ring of slots and the IPC mapping are the same."""
    import socket
    script = tmp_path / "peer_worker.py"
    script.write_text(_PEER_WORKER)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), ROOT, str(tmp_path)],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert (tmp_path / "peer.ok").exists()


def test_graphed_paint_matches_eager_statistics():
    """The hipGraph-captured eval forward draws its own noise; with the prior variance forced to ~0
    (z_log_var -> very negative is impossible after ReLU, so compare through fixed z instead):
    graph replay must be deterministic given the RNG state and finite, and equal the eager path when
    both consume the same generator state."""
    from baryon_painter_amd.models.cvae import CVAE
    arch = A.fiducial_architecture(64)
    torch.manual_seed(3)
    m = CVAE(arch, "cuda:0")
    m.train(False)
    x, y, aux = syn.synthetic_batch(4, 64, 64, seed=9)
    yt, at = torch.from_numpy(y).cuda(), torch.from_numpy(aux).cuda()
    torch.manual_seed(11)
    eager = m.sample_P(yt, aux_label=at)
    m.sample_P_graphed(yt, aux_label=at)               # capture (consumes RNG in warm-up + capture)
    torch.manual_seed(11)
    a = m.sample_P_graphed(yt, aux_label=at)
    torch.manual_seed(11)
    b = m.sample_P_graphed(yt, aux_label=at)
    assert torch.equal(a, b) and torch.isfinite(a).all()
    # same prior, different noise draw: outputs agree in distribution (mean level), not bitwise
    assert abs(float(a.mean()) - float(eager.mean())) < 0.05 * abs(float(eager.mean())) + 1e-3
    with pytest.raises(RuntimeError):
        m.train(True)
        m.sample_P_graphed(yt, aux_label=at)


def test_graphed_paint_four_streams_equals_eager_given_z():
    """The graph that bench.py's paint leg replays paints four sub-batches on four streams; with the latent given
    (``sample_P(..., z=z)``, cvae.py:149-154) it must reproduce the eager single-stream forward -- which the golden
    tests pin to the reference -- bit for bit (eval-mode layers do not couple the tiles of a batch)."""
    from baryon_painter_amd.models.cvae import CVAE
    arch = A.fiducial_architecture(128)
    torch.manual_seed(4)
    m = CVAE(arch, "cuda:0")
    with torch.no_grad():                               # non-trivial running statistics
        x, y, aux = syn.synthetic_batch(4, 128, 128, seed=12)
        m(torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(aux))
    m.train(False)
    n = 32
    x, y, aux = syn.synthetic_batch(n, 128, 128, seed=13)
    yt, at = torch.from_numpy(y).cuda(), torch.from_numpy(aux).cuda()
    z = syn.synthetic_eps((n, *arch["dim_z"]), seed=14)
    eager = m.sample_P(yt, aux_label=at, z=z)
    graphed = m.sample_P_graphed(yt, aux_label=at, z=z)
    assert len(m._graphs[(n, "z")]["plans"]) == 4
    assert torch.equal(eager, graphed)
    again = m.sample_P_graphed(yt.flip(0), aux_label=at.flip(0), z=z[::-1].copy())
    assert torch.equal(again.flip(0), eager)


def test_device_tile_assembler_matches_host_dataset():
    """GPU batch assembly == BAHAMASDataset[idx] (index arithmetic bit-exact; transform to 1 ulp)."""
    from baryon_painter_amd.utils import data_transforms as T
    rng = np.random.default_rng(3)
    data = {}
    for field, amp in (("dm", 5.0e3), ("pressure", 0.05)):
        data[field] = {}
        for z in (0.0, 0.5, 1.0):
            data[field][z] = {"100": (rng.random((5, 64, 64), dtype=np.float32) * amp),
                              "150": (rng.random((5, 64, 64), dtype=np.float32) * amp),
                              "mean_100": 0.5 * amp, "mean_150": 0.5 * amp,
                              "var_100": amp * amp / 12, "var_150": amp * amp / 12}
    fwd, inv = T.create_range_compress_transforms({"dm": 4.0, "pressure": 4}, {"dm": "shift-log", "pressure": "shift-log"})
    tr = T.chain_transformations([fwd, T.atleast_3d, T.as_float32])
    for fixed in (False, True):
        ds = D.BAHAMASDataset(data=data, redshifts=[0.0, 0.5, 1.0], label_fields=["pressure"], n_stack=4,
                              stack_offset=1, n_tile=4, tile_permutations=True, transform=tr,
                              inverse_transform=T.chain_transformations([T.squeeze, inv]), fixed_indexing=fixed)
        asm = D.DeviceTileAssembler(ds, "cuda:0", k_values={"dm": 4.0, "pressure": 4})
        idx = [0, 5, 70, ds.n_sample - 1, ds.n_sample + 3, 2 * ds.n_sample + 12345, len(ds) - 1] + \
              [int(i) for i in rng.integers(0, len(ds), 9)]
        x, y, z = asm.get_batch(idx)
        for n, i in enumerate(idx):
            (dm, pr), _, zz = ds[i]
            assert float(z[n]) == np.float32(zz)
            for got, ref in ((y[n].cpu().numpy(), dm), (x[n].cpu().numpy(), pr)):
                assert got.shape == ref.shape
                ulp = np.abs(got - ref).max() / np.spacing(np.abs(ref).max())
                assert ulp <= 1.0, (i, ulp)
    raw = D.DeviceTileAssembler(D.BAHAMASDataset(data=data, redshifts=[0.0], label_fields=["pressure"], n_stack=4,
                                                 tile_permutations=True, scale_to_SLICS=False), "cuda:0", mode=None)
    ds0 = raw.ds
    x, y, z = raw.get_batch([3, 200000])
    assert np.array_equal(y[1, 0].cpu().numpy(), ds0[200000][0][0])          # bit-exact without transform


def test_graphed_train_step_equals_eager_steps():
    """CVAE.make_graphed_train_step replays exactly the launches of model(x,y,aux) / backward / FlatAdam.step:
    feeding the eager model the noise the graph drew, parameters, Adam moments and batch-norm buffers stay
    bitwise identical over several steps (with a learning-rate change in between)."""
    from baryon_painter_amd.models.cvae import CVAE
    from baryon_painter_amd.optim import FlatAdam
    tile, n = 64, 4
    arch = A.fiducial_architecture(tile)
    torch.manual_seed(3)
    ma = CVAE(arch, "cuda:0")
    mb = CVAE(arch, "cuda:0")
    mb.load_state_dict(ma.state_dict())
    mb._bump_param_versions()
    oa, ob = FlatAdam(ma, lr=1e-3), FlatAdam(mb, lr=1e-3)
    ma.train(True); mb.train(True)
    step = ma.make_graphed_train_step(oa, n)
    for k, (pa, pb) in enumerate(zip(ma.parameters(), mb.parameters())):
        assert torch.equal(pa, pb), "capturing the graph must not change the training state"
    for it in range(3):
        x, y, aux = syn.synthetic_batch(n, tile, tile, seed=40 + it)
        x, y, aux = torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(aux)
        if it == 2:
            for o in (oa, ob):
                o.param_groups[0]["lr"] = 3e-4
        elbo_a = step(x, y, aux)
        mb._eps_override = step.last_eps().clone()
        elbo_b = mb(x, y, aux)
        ob.zero_grad()
        (-elbo_b).backward()
        ob.step()
        assert torch.equal(elbo_a.cpu(), elbo_b.detach().cpu())
        assert ma.get_stats() == mb.get_stats()
    assert oa.n_steps == ob.n_steps == 3
    for (ka, pa), (kb, pb) in zip(ma.state_dict().items(), mb.state_dict().items()):
        assert ka == kb and torch.equal(pa, pb), ka
    assert torch.equal(oa.exp_avg, ob.exp_avg) and torch.equal(oa.exp_avg_sq, ob.exp_avg_sq)
    for pa, pb in zip(ma.parameters(), mb.parameters()):
        assert torch.equal(pa.grad, pb.grad)
    # the eager paths see the updated weights after a replay
    ma.train(False); mb.train(False)
    z = torch.zeros((n, *arch["dim_z"]))
    assert torch.equal(ma.sample_P(y, aux_label=aux, z=z), mb.sample_P(y, aux_label=aux, z=z))


def test_painter_train_with_graph_step(tmp_path):
    from baryon_painter_amd.painter import CVAEPainter
    tile = 64
    train = D.SyntheticTileDataset(n_sample=32, tile_size=tile, seed=1)
    test = D.SyntheticTileDataset(n_sample=8, tile_size=tile, seed=2)
    torch.manual_seed(0)
    p = CVAEPainter(training_data_set=train, test_data_set=test, architecture=A.fiducial_architecture(tile),
                    compute_device="cuda:0")
    ts, vs = p.train(n_epoch=1, n_pepoch=2, learning_rate=1e-3, batch_size=4, pepoch_size=16,
                     adaptive_learning_rate=lambda pe: 1.0 if pe < 1 else 0.5, validation_pepochs=[],
                     validation_loss_frequency=16, validation_loss_batch_size=4, statistics_report_frequency=0,
                     verbose=False, graph_step=True)
    elbo = np.asarray(ts.loss_terms["ELBO"]["all"])
    assert len(elbo) >= 8 and np.isfinite(elbo).all()
    assert elbo[-4:].mean() > elbo[:4].mean(), "ELBO should improve over the first steps"


def test_training_script_runs_end_to_end(tmp_path):
    """scripts/CVAE_single_scale.py (the reference's training entry point, same constants and painter.train call)
    on the synthetic fallback dataset, shortened through its environment overrides."""
    env = dict(os.environ, BP_TILE="64", BP_N_PEPOCH="1", BP_OUTPUT_PATH=str(tmp_path / "out"), BP_DEVICE="cuda:0",
               BP_DATA_PATH=str(tmp_path / "no_such_stacks"))
    r = subprocess.run([sys.executable, "CVAE_single_scale.py"], cwd=os.path.join(ROOT, "scripts"), env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    runs = list((tmp_path / "out").iterdir())
    assert len(runs) == 1
    files = {p.name for p in runs[0].iterdir()}
    assert {"model_state", "model_meta", "training_stats.txt", "validation_stats.txt"} <= files, files


def test_bench_two_ranks_rehearsal():
    """`python bench.py --gpus 2` started plainly: the script launches torch.distributed.run itself as a child process
    (two ranks, gloo collectives, both on this GPU).  The multi-rank code path of the bench line - sharded batch,
    global batch-norm statistics, ONE flat gradient all-reduce, the serial profiling pass on every rank, the bf16
    secondary leg - runs and reports the aggregate."""
    import json
    env = dict(os.environ, BP_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BP_EARLY_ALLREDUCE"):
        env.pop(k, None)
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2", "--tile", "64",
           "--no-paint", "--secondary-steps", "2"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 4 and out["value"] > 0 and out["dtype"] == "f32"
    assert out["config"]["batch_norm"].startswith("global")
    # (at 64 x 64 tiles the launch with the largest time may be one of micro-FLOP size: its rate can round to 0.00 TFLOP/s)
    assert out["roofline"]["achieved"] >= 0 and out["roofline"]["kernel"] and out["roofline"]["avg_launch_ms"] > 0
    c = out["config"]["collectives_per_step"]
    # (44 separate statistics collectives, or 1 + 57 exchanges inside the finalize kernels over peer memory)
    assert (c["batch_norm_statistics"], c["batch_norm_statistics_fused_into_finalize"]) in ((44, 0), (1, 57))
    assert c["gradient_buffers"] == 1
    assert out["config"]["gradient_bytes_per_step"] == 4 * 1662961
    inside = out["config"]["inside_collectives"]
    assert set(inside) == {"bn", "grad"} and inside["bn"]["per_step"] == c["batch_norm_statistics"] and inside["bn"]["min_us"] > 0
    b = out["bf16"]
    assert b["dtype"] == "bf16" and b["n_gpus"] == 2 and b["value"] > 0 and b["roofline"]["bound"] == "hbm"
    assert out["config"]["other_configs"]["bf16 (configs[3] per GPU)"]["value"] == b["value"]
    assert "cgan" not in out                      # (single-GPU leg)


def test_training_script_two_ranks_rehearsal(tmp_path):
    """The training script under torch.distributed.run with two ranks (gloo collectives, both ranks on this GPU):
    sharded loader, global batch-norm statistics, flat gradient all-reduce, rank-0-only output files."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, BP_TILE="64", BP_N_PEPOCH="1", BP_OUTPUT_PATH=str(tmp_path / "out"),
               BP_DATA_PATH=str(tmp_path / "no_such_stacks"), BP_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), "CVAE_single_scale.py"]
    r = subprocess.run(cmd, cwd=os.path.join(ROOT, "scripts"), env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    runs = list((tmp_path / "out").iterdir())
    assert len(runs) == 1
    files = {p.name for p in runs[0].iterdir()}
    assert {"model_state", "model_meta", "training_stats.txt"} <= files, files
