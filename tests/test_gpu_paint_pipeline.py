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
