"""GPU: the paint() throughput pipeline (BASELINE.json configs[4]; SURVEY.md 8d metric (B), 8e, 8f-2, 8f-3):
device-side transforms, counter-based per-tile noise, pinned double-buffered streaming, tile sharding, and the
light-cone caller -- against the per-tile ``paint`` of the drop-in API (which tests/test_gpu_reference_paint.py pins
to the reference's painter end to end)."""
import numpy as np
import pytest
import torch

import host_cases as HC
from baryon_painter_amd import _lib as L
from baryon_painter_amd.models import arch as A
from baryon_painter_amd.utils import data_transforms as T
from baryon_painter_amd.utils import synthetic as syn
from baryon_painter_amd.utils.datasets import BAHAMASDataset
from oracle.philox import tile_normals

import gpu_util as G

pytestmark = pytest.mark.gpu


def test_philox_normals_equal_the_oracle():
    lib = L.load()
    ids = np.array([0, 1, 7, 2 ** 33 + 5, 123456789012], dtype=np.int64)
    for seed, per_tile, Ls in ((0, 256, 1), (20190101, 64, 2), (2 ** 40 + 17, 10, 1)):
        eps = torch.zeros((Ls, len(ids), per_tile), device="cuda")
        L.check(lib.bp_philox_normal(seed, L.ptr(torch.from_numpy(ids).cuda()), len(ids), Ls, per_tile, L.ptr(eps),
                                     G.stream()))
        ref = tile_normals(seed, ids, per_tile, Ls)
        got = eps.cpu().numpy()
        assert np.abs(got - ref).max() <= 2.0 ** -22 and (got == ref).mean() > 0.999      # libm: a last bit at most
    big = tile_normals(5, np.arange(64), 256)[0]
    assert abs(big.mean()) < 0.02 and abs(big.std() - 1) < 0.02


@pytest.fixture(scope="module")
def painter(tmp_path_factory):
    """A 64x64 painter loaded from checkpoint files (the path that has transforms), non-trivial running statistics."""
    from baryon_painter_amd.painter import CVAEPainter
    size = 64
    arch = A.fiducial_architecture(size)
    fwd, inv = T.create_range_compress_transforms(HC.K_VALUES, HC.MODES)
    tr = T.chain_transformations([fwd, T.atleast_3d, T.as_float32])
    itr = T.chain_transformations([T.squeeze, inv])
    ds = BAHAMASDataset(data=HC.data_dict("random"), redshifts=list(HC.REDSHIFTS), label_fields=["pressure"], n_tile=1,
                        n_stack=3, transform=tr, inverse_transform=itr, scale_to_SLICS=True)
    torch.manual_seed(3)
    p = CVAEPainter(training_data_set=ds, test_data_set=ds, architecture=arch, compute_device="cuda:0")
    x, y, aux = syn.synthetic_batch(4, size, size, seed=77)
    with torch.no_grad():
        p.model(torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(aux))
    d = tmp_path_factory.mktemp("ckpt")
    files = (str(d / "state"), str(d / "meta"))
    p.save_state_to_file(files)
    q = CVAEPainter(filename=files, compute_device="cuda:0")
    tiles = np.stack([np.asarray(ds.get_input_sample(i % len(ds), transform=False), np.float32) for i in range(11)])
    tiles *= (1.0 + 0.1 * np.arange(11, dtype=np.float32))[:, None, None]          # 11 distinct tiles
    zs = np.array([0.0, 0.3, 2.0, 0.5, 1.1, 0.0, 2.0, 0.125, 1.9, 0.7, 0.3])
    q.checkpoint_files = files
    return q, arch, tiles, zs


def test_paint_stream_equals_per_tile_paint(painter):
    """Device transforms + captured graph + pipelined copies == paint(tile, z) of the drop-in API tile by tile, given
    the same prior noise (the oracle's Philox normals for (seed, tile id))."""
    q, arch, tiles, zs = painter
    seed, ids = 99, np.arange(11, dtype=np.int64) + 1000
    out = q.paint_stream(tiles, zs, batch_size=4, tile_ids=ids, seed=seed)
    assert out.shape == tiles.shape and out.dtype == np.float32 and np.isfinite(out).all()
    per_tile = int(np.prod(arch["dim_z"]))
    for i in range(len(tiles)):
        q.model._eps_override = tile_normals(seed, [ids[i]], per_tile).reshape(1, 1, *arch["dim_z"])
        ref = np.asarray(q.paint(tiles[i], z=float(zs[i])), np.float64)
        # host: float64 product with sigma; device: the same value rounded to float32; exp may differ by one ulp
        tol = 3e-7 * np.abs(ref).max()
        assert np.abs(out[i] - ref).max() <= tol, (i, np.abs(out[i] - ref).max(), tol)
    q.model._eps_override = None
    with pytest.raises(ValueError):
        q.paint_stream(tiles[:, :32], zs)


def test_paint_stream_is_independent_of_batching_and_sharding(painter):
    q, arch, tiles, zs = painter
    ref = q.paint_stream(tiles, zs, batch_size=4, seed=7)
    assert np.array_equal(q.paint_stream(tiles, zs, batch_size=11, seed=7), ref)
    assert np.array_equal(q.paint_stream(tiles, zs, batch_size=3, seed=7), ref)
    for world in (2, 3):
        parts = [q.paint_stream(tiles, zs, batch_size=4, seed=7, rank=r, world_size=world) for r in range(world)]
        assert [p[1] for p in parts][0][0] == 0 and parts[-1][1][1] == len(tiles)
        assert np.array_equal(np.concatenate([p[0] for p in parts]), ref)
    assert not np.array_equal(q.paint_stream(tiles, zs, batch_size=4, seed=8), ref)          # another realisation
    # pinned torch tensors in and out: no staging copies on the host
    tin = torch.from_numpy(tiles).pin_memory()
    tout = torch.empty(tiles.shape, dtype=torch.float32).pin_memory()
    q.paint_stream(tin, zs, batch_size=4, seed=7, out=tout)
    assert np.array_equal(tout.numpy(), ref)


def test_paint_plane_equals_the_per_tile_loop(painter):
    """lightcone.paint_plane (batched, pipelined) against the reference's serial loop (process_SLICS.py:198-220):
    get_tile -> paint -> weight -> accumulate, tile by tile."""
    from baryon_painter_amd import lightcone as LC
    q, arch, tiles, zs = painter
    n_tile = 64
    rng = np.random.Generator(np.random.PCG64(31))
    delta = (np.exp(rng.standard_normal((150, 150)) * 0.5) * 0.05).astype(np.float32)       # periodic plane
    rel = n_tile / 150
    z, seed = 0.42, 5
    plane = LC.paint_plane(q, delta, rel, n_tile, z, seed=seed)
    origins, slices = LC.generate_tiling(150, n_tile, 0.5)
    assert plane.shape == (150, 150) and len(origins) >= 4
    acc, wsum = np.zeros((150, 150)), np.zeros((150, 150))
    per_tile = int(np.prod(arch["dim_z"]))
    tid = 0
    for j, xs in enumerate(origins):
        for k, ys in enumerate(origins):
            tile = np.asarray(LC.get_tile(delta, (xs, ys), rel), np.float32)
            q.model._eps_override = tile_normals(seed, [tid], per_tile).reshape(1, 1, *arch["dim_z"])
            painted = np.asarray(q.paint(tile, z=z), np.float64)
            w = LC.make_weight_map(tile.shape, falloff=0.05, sigma=0.5)
            acc[slices[j][k]] += w * painted
            wsum[slices[j][k]] += w
            tid += 1
    q.model._eps_override = None
    with np.errstate(invalid="ignore"):
        ref = acc / wsum
    # (pixels no tile covers -- the int() truncations of generate_tiling can leave the last row / column out -- are
    #  0 / 0 in the reference's blend as well)
    ok = np.isfinite(ref)
    assert np.array_equal(np.isfinite(plane), ok) and ok.mean() > 0.95
    assert np.abs(plane[ok] - ref[ok]).max() <= 1e-6 * np.abs(ref[ok]).max()


def test_paint_stream_bf16_mode(painter):
    """CVAEPainter(dtype="bf16") on the same checkpoint files: painting with the bf16 trunk stays within the stated bf16
    tolerance of the fp32 painting of the same tiles with the same noise (compared in the network's log domain: the
    inverse transform exponentiates 4x the network output, which turns 1e-2 of the output into 4e-2 of the field)."""
    from baryon_painter_amd.painter import CVAEPainter
    q, arch, tiles, zs = painter
    b = CVAEPainter(filename=q.checkpoint_files, compute_device="cuda:0", dtype="bf16")
    assert b.model.dtype == "bf16"
    ref = q.paint_stream(tiles, zs, batch_size=4, seed=3).astype(np.float64)
    got = b.paint_stream(tiles, zs, batch_size=4, seed=3).astype(np.float64)
    plan = next(iter(b.model._graphs.values()))["plans"][0]
    assert plan.h.buf.dtype == torch.bfloat16, "the generator trunk must be stored as bf16"
    sigma = ref.std()
    lr, lg = np.log1p(np.maximum(ref, 0) / sigma), np.log1p(np.maximum(got, 0) / sigma)
    assert np.linalg.norm(lg - lr) / np.linalg.norm(lr) <= 2e-2


def test_one_graph_serves_every_seed(painter):
    """The Philox key travels in the per-batch parameter block (bp_philox_normal_dev): painting with many seeds
    neither re-captures graphs nor grows the plan cache."""
    q, arch, tiles, zs = painter
    q.paint_stream(tiles[:4], zs[:4], batch_size=4, seed=1)
    n_graphs, n_plans = len(q.model._graphs), len(q.model._plans)
    outs = [q.paint_stream(tiles[:4], zs[:4], batch_size=4, seed=s) for s in (2, 3, 2 ** 63 + 5, 2)]
    assert (len(q.model._graphs), len(q.model._plans)) == (n_graphs, n_plans)
    assert np.array_equal(outs[0], outs[3]) and not np.array_equal(outs[0], outs[1])
    assert not np.array_equal(outs[1], outs[2])
    # the key is all 64 bits of the seed: the oracle's normals for (seed, tile id)
    per_tile = int(np.prod(arch["dim_z"]))
    q.model._eps_override = tile_normals(2 ** 63 + 5, [0], per_tile).reshape(1, 1, *arch["dim_z"])
    ref = np.asarray(q.paint(tiles[0], z=float(zs[0])), np.float64)
    q.model._eps_override = None
    assert np.abs(outs[2][0] - ref).max() <= 3e-7 * np.abs(ref).max()


def test_paint_plane_noise_is_fresh_per_call_and_reproducible(painter):
    """Default arguments must not correlate the planes of a light cone (the reference draws fresh torch.randn per
    tile, cvae.py:64): two calls differ; torch.manual_seed makes the sequence reproducible; an explicit seed pins
    one plane."""
    from baryon_painter_amd import lightcone as LC
    q, arch, tiles, zs = painter
    rng = np.random.Generator(np.random.PCG64(32))
    delta = (np.exp(rng.standard_normal((100, 100)) * 0.5) * 0.05).astype(np.float32)
    args = (q, delta, 64 / 100, 64, 0.3)
    torch.manual_seed(1234)
    a, b = LC.paint_plane(*args), LC.paint_plane(*args)
    ok = np.isfinite(a)
    assert not np.array_equal(a[ok], b[ok])
    torch.manual_seed(1234)
    a2, b2 = LC.paint_plane(*args), LC.paint_plane(*args)
    assert np.array_equal(a[ok], a2[ok]) and np.array_equal(b[ok], b2[ok])
    assert np.array_equal(LC.paint_plane(*args, seed=5)[ok], LC.paint_plane(*args, seed=5)[ok])


def test_paint_plane_falls_back_where_the_device_pipeline_has_no_form(painter, monkeypatch):
    """Transforms other than a plain shift-log chain (custom steps, no transform) are refused by the device pipeline
    with NotImplementedError BEFORE anything is captured, and paint_plane then paints through paint_batch as it
    did before the pipeline existed."""
    from baryon_painter_amd import lightcone as LC
    q, arch, tiles, zs = painter
    rng = np.random.Generator(np.random.PCG64(33))
    delta = (np.exp(rng.standard_normal((100, 100)) * 0.5) * 0.05).astype(np.float32)

    def doubled(x, field, z, stats):
        return 2.0 * x
    good = q.transform
    calls = []
    orig = q.paint_batch
    monkeypatch.setattr(q, "paint_batch", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    try:
        q.transform = type(good)(T.chain_transformations([doubled] + list(good.func.steps)), good.stats)
        n_graphs = len(q.model._graphs)
        with pytest.raises(NotImplementedError):
            q.paint_stream(tiles[:2], zs[:2], batch_size=2)
        assert len(q.model._graphs) == n_graphs
        plane = LC.paint_plane(q, delta, 64 / 100, 64, 0.3)
        assert len(calls) == 1 and np.isfinite(plane).mean() > 0.9
        q.transform = None
        with pytest.raises(NotImplementedError):
            q.paint_stream(tiles[:2], zs[:2], batch_size=2)
    finally:
        q.transform = good
    # the plain chain still takes the device pipeline
    LC.paint_plane(q, delta, 64 / 100, 64, 0.3)
    assert len(calls) == 1


@pytest.fixture(scope="module")
def painter512(tmp_path_factory):
    """The fiducial network at its real tile size with the 64-pixel fixture's transforms and statistics."""
    from baryon_painter_amd.painter import CVAEPainter
    arch = A.fiducial_architecture(512)
    fwd, inv = T.create_range_compress_transforms(HC.K_VALUES, HC.MODES)
    tr = T.chain_transformations([fwd, T.atleast_3d, T.as_float32])
    itr = T.chain_transformations([T.squeeze, inv])
    ds = BAHAMASDataset(data=HC.data_dict("random"), redshifts=list(HC.REDSHIFTS), label_fields=["pressure"], n_tile=1,
                        n_stack=3, transform=tr, inverse_transform=itr, scale_to_SLICS=True)
    torch.manual_seed(3)
    p = CVAEPainter(training_data_set=ds, test_data_set=ds, architecture=arch, compute_device="cuda:0")
    x, y, aux = syn.synthetic_batch(2, 512, 512, seed=77)
    with torch.no_grad():
        p.model(torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(aux))
    d = tmp_path_factory.mktemp("ckpt512")
    files = (str(d / "state"), str(d / "meta"))
    p.save_state_to_file(files)
    del p
    q = CVAEPainter(filename=files, compute_device="cuda:0")
    rng = np.random.Generator(np.random.PCG64(5))
    tiles = (np.exp(rng.standard_normal((8, 512, 512)) * 0.7) * 0.8).astype(np.float32)
    zs = np.array([0.0, 0.3, 2.0, 0.5, 1.1, 0.0, 2.0, 0.125])
    q.checkpoint_files = files
    return q, arch, tiles, zs


def test_paint_stream_at_512(painter512):
    """BASELINE.json configs[4] geometry: 8 tiles of 512^2 in batches of 4 through the pipeline == paint(tile, z)
    tile by tile with the same Philox noise, fp32 to 3e-7 of the tile's maximum."""
    q, arch, tiles, zs = painter512
    seed, ids = 11, np.arange(8, dtype=np.int64) + 50
    out = q.paint_stream(tiles, zs, batch_size=4, tile_ids=ids, seed=seed)
    assert out.shape == tiles.shape and np.isfinite(out).all()
    per_tile = int(np.prod(arch["dim_z"]))
    for i in range(len(tiles)):
        q.model._eps_override = tile_normals(seed, [ids[i]], per_tile).reshape(1, 1, *arch["dim_z"])
        ref = np.asarray(q.paint(tiles[i], z=float(zs[i])), np.float64)
        assert np.abs(out[i] - ref).max() <= 3e-7 * np.abs(ref).max(), i
    q.model._eps_override = None


def test_paint_stream_at_512_bf16(painter512):
    """The bf16 trunk on the same tiles and noise: <= 1e-2 relative L2 in the NETWORK's domain (x_mu: the inverse
    transform exponentiates 4x the network output) against the fp32 painting."""
    from baryon_painter_amd.painter import CVAEPainter
    q, arch, tiles, zs = painter512
    b = CVAEPainter(filename=q.checkpoint_files, compute_device="cuda:0", dtype="bf16")
    ref = q.paint_stream(tiles, zs, batch_size=4, seed=3).astype(np.float64)
    got = b.paint_stream(tiles, zs, batch_size=4, seed=3).astype(np.float64)
    # back to the network domain with the painter's own forward transform of the label field
    net = lambda a, z: np.asarray(q.transform(a.astype(np.float32), field=q.label_fields[0], z=float(z)), np.float64)
    num = den = 0.0
    for i in range(len(tiles)):
        r, g = net(ref[i], zs[i]), net(got[i], zs[i])
        num += ((g - r) ** 2).sum()
        den += (r ** 2).sum()
    assert np.sqrt(num / den) <= 1e-2, np.sqrt(num / den)
