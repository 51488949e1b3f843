"""CPU tests of the host side: C ABI symbols, architecture language, datasets / transforms /
checkpoint-format helpers, data-parallel plumbing over gloo (world_size 2)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    """Every function declared in include/bp_hip.h resolves in the built .so (no compute calls)."""
    from baryon_painter_amd import _lib as L
    lib = L.load()
    header = open(os.path.join(ROOT, "include", "bp_hip.h")).read()
    names = set(re.findall(r"\b(bp_[a-z0-9_]+)\s*\(", header))
    assert len(names) >= 25
    for n in sorted(names):
        assert hasattr(lib, n), f"{n} declared in bp_hip.h but not exported"
        assert n in L.SIGNATURES, f"{n} has no ctypes signature"
    assert lib.bp_version() >= 100
    assert b"invalid" in lib.bp_strerror(-1)
    cv = L.Conv(0, 128, 128, 3, 1, 1, 0)
    assert lib.bp_conv_packed_floats(ctypes.byref(cv), L.PACK_FWD) == 9 * 128 * 128
    assert lib.bp_conv_kernel_id(ctypes.byref(cv), L.PACK_FWD) == 216424   # LDS-DMA igemm with 8 waves, CC 16, NT 4, WN 2, MT 4


def test_model_refuses_cpu_and_missing_library(monkeypatch, tmp_path):
    from baryon_painter_amd.models import arch as A
    from baryon_painter_amd.models.cvae import CVAE
    with pytest.raises(RuntimeError, match="GPU only"):
        CVAE(A.fiducial_architecture(64), "cpu")
    from baryon_painter_amd import _lib as L
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        L.load()


def test_architecture_helpers_error_behaviour():
    from baryon_painter_amd.models import arch as A
    from baryon_painter_amd.models.graph import build_holders
    with pytest.raises(ValueError):
        A.conv_block(1, 1, kernel=4)                       # utils.py:42-43
    with pytest.raises(NotImplementedError):
        A.conv_block(1, 1, scale=3)
    with pytest.raises(NotImplementedError):
        A.conv_block(1, 1, activation="swish")
    with pytest.raises(NotImplementedError):
        build_holders([("maxpool", {})])
    with pytest.raises(RuntimeError):
        build_holders([("conv", {}, 1)])
    seq = build_holders(A.fiducial_architecture(64)["p_y_z_in"])
    keys = list(seq.state_dict().keys())
    assert "12.res_block.0.weight" in keys and "1.num_batches_tracked" in keys
    assert build_holders(None) is None


def test_transform_roundtrip_within_reference_tolerance():
    """transform o inverse == id within 2e-5*sigma (the reference's tests/test_dataset.py:80-83)."""
    from baryon_painter_amd.utils import data_transforms as T
    stats = {"dm": {0.0: {"mean": 1.0, "var": 1.47}, 1.0: {"mean": 1.0, "var": 0.4}, 2.0: {"mean": 1.0, "var": 0.11}}}
    rng = np.random.default_rng(0)
    x = rng.lognormal(0, 1, (64, 64)).astype(np.float32)
    for mode, k in (("shift-log", 4.0), ("log", 2.0), ("shift-log-2p", (0.01, 4.0)), ("x/(1+x)", (2, 1))):
        fwd, inv = T.create_range_compress_transforms({"dm": k}, {"dm": mode}, eps=1e-4)
        for z in (0.0, 0.3, 1.0, 2.5):
            back = inv(fwd(x, "dm", z, stats), "dm", z, stats)
            sigma = np.sqrt(T.interpolate_z(stats["dm"], z)["var"])
            assert np.abs(back - x).max() <= 2e-5 * sigma * max(1.0, x.max())
    assert T.interpolate_z(stats["dm"], 0.5)["var"] == pytest.approx(0.5 * 1.47 + 0.5 * 0.4)
    assert T.interpolate_z(stats["dm"], 9.0)["var"] == 0.11 and T.interpolate_z(stats["dm"], -1.0)["var"] == 1.47
    with pytest.raises(ValueError):
        T.create_range_compress_transforms({"dm": 1}, {"dm": "nope"})[0](x, "dm", 0.0, stats)


def _fake_stacks(n_stack=5, n_grid=32):
    rng = np.random.default_rng(1)
    data = {}
    for field in ("dm", "pressure"):
        data[field] = {}
        for z in (0.0, 1.0):
            data[field][z] = {"100": rng.random((n_stack, n_grid, n_grid), dtype=np.float32),
                              "150": rng.random((n_stack, n_grid, n_grid), dtype=np.float32),
                              "mean_100": 0.5, "mean_150": 0.5, "var_100": 0.08, "var_150": 0.08}
    return data


def test_bahamas_index_mapping_quirks():
    from baryon_painter_amd.utils.datasets import BAHAMASDataset
    data = _fake_stacks()
    ds = BAHAMASDataset(data=data, redshifts=[0.0, 1.0], label_fields=["pressure"], n_stack=4, stack_offset=1,
                        n_tile=4, tile_permutations=True, scale_to_SLICS=True)
    assert ds.n_tile_permutation == 8 and ds.tile_size == 8
    assert ds.n_sample == 4 ** 2 * 4 ** 4 * 64 and len(ds) == 2 * ds.n_sample
    # only idx % 64 selects the (stack, tile) combination (datasets.py:327)
    assert ds.sample_idx_to_tile(5) == ds.sample_idx_to_tile(5 + 64) == (1, 0, 0, 1, 1, 1)
    assert ds.sample_idx_to_tile(63) == (1, 0, 0, 4, 3, 3)
    assert ds.sample_idx_to_tile_permutation(0) == (0, 0)
    assert ds.sample_idx_to_tile_permutation(ds.n_sample - 1) == (7, 7)
    assert ds.sample_idx_to_redshift(ds.n_sample) == 1.0
    t = np.arange(16.0).reshape(4, 4)
    assert np.array_equal(ds.apply_tile_permutation(t, 3), t)              # flip code 3: no-op
    assert np.array_equal(ds.apply_tile_permutation(t, 1), t[:, ::-1])
    assert np.array_equal(ds.apply_tile_permutation(t, 6), np.rot90(t, 1)[::-1])
    sample, idx, z = ds[70]
    assert idx == 70 and z == 0.0 and sample[0].shape == (8, 8) and len(sample) == 2
    expected = (data["dm"][0.0]["100"][1][0:8, 0:8] + data["dm"][0.0]["150"][1][8:16, 16:24]) * (1 / (32 / 8 * 5) * 0.2793 / (0.2793 - 0.0463))
    assert np.allclose(sample[0], expected)
    fixed = BAHAMASDataset(data=data, redshifts=[0.0, 1.0], label_fields=["pressure"], n_stack=4, stack_offset=1,
                           tile_permutations=True, fixed_indexing=True)
    assert fixed.sample_idx_to_tile(5 + 64) != fixed.sample_idx_to_tile(5)
    with pytest.raises(ValueError):
        BAHAMASDataset(data=data, n_stack=6)
    with pytest.raises(ValueError):
        BAHAMASDataset()


def test_training_stats_text_format(tmp_path):
    from baryon_painter_amd.painter import TrainingStats
    f = tmp_path / "training_stats.txt"
    st = TrainingStats(["ELBO", "KL_term", "log_likelihood_pressure_0", "lr", "batch_size"], 2,
                       dump_to_file_frequency=2, stats_filename=str(f))
    st.push_loss(4, -10.0, -1.0, -9.0, 1e-3, 4)
    st.push_loss(8, -8.0, -0.5, -7.5, 1e-3, 4)
    st.push_loss(12, -6.0, -0.5, -5.5, 1e-3, 4)
    st.flush_to_file()
    lines = f.read_text().splitlines()
    assert lines[0] == "# Batch nr, sample nr, ELBO, KL_term, log_likelihood_pressure_0, lr, batch_size"
    assert lines[1] == "0 4 -10.0 -1.0 -9.0 0.001 4 "
    assert len(lines) == 4 and st.loss_terms["ELBO"]["mavg"][-1] == -7.0
    assert "ELBO" in st.get_pretty_str()


def test_shard_indices_reproduce_global_batches():
    from baryon_painter_amd.dist import shard_indices
    perm = np.random.default_rng(0).permutation(103)
    shards = [shard_indices(perm, r, 4, 6) for r in range(4)]
    assert all(len(s) == 103 // 24 for s in shards)
    for b in range(len(shards[0])):
        merged = sum((shards[r][b] for r in range(4)), [])
        assert merged == list(perm[b * 24:(b + 1) * 24])


_GLOO_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from baryon_painter_amd.dist import Sync
dist.init_process_group("gloo")
s = Sync()
r = s.rank
sums = torch.tensor([1.0 + r, 10.0 * (r + 1)], dtype=torch.float64)
s.all_reduce_sum(sums)
assert sums.tolist() == [3.0, 30.0], sums
flat = torch.full((1000,), float(r + 1))
s.all_reduce_mean(flat)
assert torch.allclose(flat, torch.full((1000,), 1.5))
# default: ONE communicator for statistics and gradients (program order = the same order on every rank)
assert not s.overlap and s.grad_group is s.group
# opt-in: gradients on their own communicator (may overlap the statistics' one on another stream), slice-wise
s2 = Sync(grad_group="new")
assert s2.overlap and s2.grad_group is not s2.group
part = torch.arange(10.0) * (r + 1)
s2.all_reduce_mean(part[2:5])
s.n_grad += s2.n_grad; s.bytes_grad += s2.bytes_grad
assert part[:2].tolist() == [0.0, 1.0 * (r + 1)] and torch.allclose(part[2:5], torch.tensor([3.0, 4.5, 6.0]))
assert s.n_grad == 2 and s.bytes_grad == 4000 + 12 and s.n_small == 1
local = Sync(sync_bn=False)
t = torch.tensor([float(r)], dtype=torch.float64)
local.all_reduce_sum(t)
assert t.item() == float(r)                      # batch-norm statistics stay local
# two-rank batch-norm statistics == single-device statistics of the concatenated batch
g = torch.Generator().manual_seed(0)
full = torch.randn(8, 5, 4, 4, generator=g, dtype=torch.float64)
mine = full[r * 4:(r + 1) * 4]
st = torch.cat([mine.sum(dim=(0, 2, 3)), (mine ** 2).sum(dim=(0, 2, 3))])
s.all_reduce_sum(st)
n = full.numel() / 5
mean = st[:5] / n
var = st[5:] / n - mean ** 2
assert torch.allclose(mean, full.mean(dim=(0, 2, 3))) and torch.allclose(var, full.var(dim=(0, 2, 3), unbiased=False))
dist.destroy_process_group()
open(os.path.join(sys.argv[2], f"rank{r}.ok"), "w").write("ok")      # stdout of the ranks interleaves
"""


def test_sync_collectives_over_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_WORKER)
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), ROOT, str(tmp_path)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert (tmp_path / "rank0.ok").exists() and (tmp_path / "rank1.ok").exists()


def test_lightcone_tiling_known_answers():
    """The four known answers of the reference's tests/test_SLICS_tiling.py:72-81 + coverage."""
    from baryon_painter_amd.lightcone import generate_tiling, get_tile, make_weight_map
    assert len(generate_tiling(512, 256, min_tile_overlap=0.0)[0]) == 2
    assert len(generate_tiling(512, 250, min_tile_overlap=0.0)[0]) == 3
    assert len(generate_tiling(512, 256, min_tile_overlap=0.5)[0]) == 3
    assert len(generate_tiling(512, 128, min_tile_overlap=0.0)[0]) == 4
    _, tiles = generate_tiling(512, 32, min_tile_overlap=0.33)
    cover = np.zeros((512, 512))
    for row in tiles:
        for s in row:
            cover[s] += 1
    assert cover.min() >= 1 and cover[:32, :32].max() <= 4
    m = np.arange(64.0).reshape(8, 8)
    t = get_tile(m, (0.75, 0.5), 0.5)
    assert t.shape == (4, 4) and t[0, 0] == m[6, 4] and t[3, 3] == m[1, 7]      # wraps around
    with pytest.raises(ValueError):
        get_tile(m, (0, 0), 0.5, expansion_factor=0.5)
    w = make_weight_map((100, 100), falloff=0.05, sigma=0.5)
    assert w[50, 50] == 1.0 and w[0, 50] == pytest.approx(np.exp(-0.5 * 25 / 6.25)) and np.allclose(w, w.T)


def test_device_loader_reproduces_dataloader_shuffle():
    import torch
    from baryon_painter_amd.painter import dataloader_shuffle_order

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return 37

        def __getitem__(self, i):
            return i
    torch.manual_seed(5)
    ref = [int(v) for b in torch.utils.data.DataLoader(DS(), batch_size=4, shuffle=True) for v in b]
    torch.manual_seed(5)
    assert dataloader_shuffle_order(37) == ref


def test_cgan_field_and_redshift_transforms_known_answers():
    """g4: the CGAN's own conditioning transforms, pinned by hand-computed values.  Field transform mode
    "shift-log-cam" of trained_models/CGAN/fiducial/transform.pickle (disassembled in SURVEY.md 8a g4):
    t(x) = log(x / sigma + 1) / k0 - k1 with k = [4.0, 1.0]; redshift map f(z) = z - 1
    (/root/reference/trained_models/README.md:99)."""
    import torch
    from baryon_painter_amd.painter import CGANPainter
    from baryon_painter_amd.models.cgan import CGAN
    p = CGANPainter.__new__(CGANPainter)
    # sigma^2 tabulated at z = 0 and z = 1; sigma(0) = 2, sigma(1) = 4, linear IN THE VARIANCE in between
    p.stats = {"dm": {0.0: {"mean": 1.0, "var": 4.0}, 1.0: {"mean": 1.0, "var": 16.0}},
               "pressure": {0.0: {"mean": 0.1, "var": 0.25}, 1.0: {"mean": 0.1, "var": 0.25}}}
    assert CGANPainter.K == (4.0, 1.0)
    e = np.e
    x = np.array([0.0, 2.0 * (e - 1), 2.0 * (e ** 4 - 1), 2.0 * (e ** 8 - 1)])
    # log(x/2 + 1) = 0, 1, 4, 8  ->  /4 - 1  =  -1, -0.75, 0, 1: the tanh range of the generator
    assert np.allclose(p.transform(x, "dm", 0.0), [-1.0, -0.75, 0.0, 1.0], atol=1e-6)
    # z = 0.5: var = 10, sigma = sqrt(10); x = sigma * (e^2 - 1) -> 2/4 - 1 = -0.5
    assert np.allclose(p.transform(np.array([np.sqrt(10.0) * (e ** 2 - 1)]), "dm", 0.5), [-0.5], atol=1e-6)
    # beyond the last tabulated redshift the last entry holds (data_transforms.py:52-64)
    assert np.allclose(p.transform(np.array([4.0 * (e ** 4 - 1)]), "dm", 3.0), [0.0], atol=1e-6)
    # inverse: (exp((y + k1) * k0) - 1) * sigma; y = -1 is zero pressure, y = -0.75 is sigma * (e - 1)
    assert np.allclose(p.inverse_transform(np.array([-1.0, -0.75]), "pressure", 0.0), [0.0, 0.5 * (e - 1)], atol=1e-12)
    y = np.array([0.0, 0.03, 1.7, 250.0])
    assert np.allclose(p.inverse_transform(p.transform(y, "pressure", 0.3), "pressure", 0.3), y, rtol=2e-5, atol=2e-6)
    assert CGAN.z_transform(0.0) == -1.0 and CGAN.z_transform(2.0) == 1.0
    assert torch.equal(CGAN.z_transform(torch.tensor([0.0, 0.5, 1.0])), torch.tensor([-1.0, -0.5, 0.0]))


def test_vectorised_redshift_interpolation_equals_the_scalar_one():
    """paint_stream's per-tile sigma table (data_transforms.interpolate_z_many) is bit for bit the reference's
    per-call interpolation (data_transforms.py:52-64: searchsorted side="right", clamped at both ends)."""
    from baryon_painter_amd.utils import data_transforms as T
    rng = np.random.default_rng(0)
    st = {0.0: {"mean": 1.0, "var": 1.47251}, 0.125: {"mean": 1.0, "var": 1.3}, 0.5: {"mean": 1.0, "var": 0.9},
          2.0: {"mean": 1.0, "var": 0.1165}}
    zs = np.concatenate([rng.uniform(-0.5, 2.5, 500), [0.0, 0.125, 0.5, 2.0, 1.9999999, 2.0000001, -1e-9]])
    for key in ("var", "mean"):
        assert np.array_equal(np.array([T.interpolate_z(st, float(z))[key] for z in zs]), T.interpolate_z_many(st, zs, key))
    one = {0.3: {"mean": 1.0, "var": 2.0}}
    assert np.array_equal(T.interpolate_z_many(one, zs), np.full(len(zs), 2.0))


def test_dense_gradient_buffer_of_a_channel_slice():
    """graph.Slot: a channel slice normally keeps its gradient inside the wide slot's gradient buffer (same stride);
    with ``dense_grad`` (the latent channel of the generator's 4-channel input, cvae._Plan) it owns a dense buffer --
    host bookkeeping only, no kernel."""
    import torch
    from baryon_painter_amd.models.graph import Slot
    wide = Slot.new(2, 4, 6, 3, "cpu", cstride=4)
    z = wide.sub(0, 1, dense_grad=True)
    ya = wide.sub(1, 3)
    gz, gy = z.ensure_grad(), ya.ensure_grad()
    assert (gz.c, gz.cstride, gz.coff) == (1, 1, 0) and tuple(z.grad_buf.shape) == (2, 4, 6, 1)
    assert (gy.c, gy.cstride, gy.coff) == (2, 4, 1) and ya.grad_buf is wide.grad_buf
    assert z.grad_buf.data_ptr() != wide.grad_buf.data_ptr()
    assert z.ensure_grad() is gz                           # allocated once
    assert torch.count_nonzero(z.grad_buf) == 0
