"""SURVEY.md 8f rows pinned to the reference (CPU): tests/golden/host.npz holds what the REAL reference returned
for the seeded cases of tests/host_cases.py (generator: tests/golden/make_goldens_host.py, build container only).

  * tile indexing is integer work -> bit-exact, including the reference's quirks;
  * transforms are float64 NumPy formulas -> bit-exact against the reference on the same NumPy;
    (tolerance 1 ulp in case a different libm ends up behind np.log / np.exp on another host);
  * tiling / weights / wrap-around cuts -> bit-exact.
"""
import os

import numpy as np
import pytest

import host_cases as HC

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "host.npz"))


def _transforms():
    from baryon_painter_amd.utils import data_transforms as T
    fwd, inv = T.create_range_compress_transforms(HC.K_VALUES, HC.MODES)
    return T.chain_transformations([fwd, T.atleast_3d]), inv


def _make(kind, n_stack, off, perm, **kw):
    from baryon_painter_amd.utils.datasets import BAHAMASDataset
    return BAHAMASDataset(data=HC.data_dict(kind), redshifts=list(HC.REDSHIFTS), label_fields=["pressure"],
                          n_tile=HC.N_TILE, n_stack=n_stack, stack_offset=off, tile_permutations=perm,
                          scale_to_SLICS=True, **kw)


@pytest.mark.parametrize("tag,n_stack,off,perm", HC.DATASET_CASES)
def test_tile_indexing_is_the_references_bit_for_bit(tag, n_stack, off, perm, gold):
    """Which pixel of which stack lands where, for ~1000 indices per setting: the address-coded stacks went through
    the reference's get_stack(); ours must return the same addresses (stack, tile and dihedral permutation)."""
    tr, inv = _transforms()
    ds = _make("random", n_stack, off, perm, transform=tr, inverse_transform=inv)
    c100, c150 = _make("coded100", n_stack, off, perm), _make("coded150", n_stack, off, perm)
    assert len(ds) == int(gold[f"ds/{tag}/len"]) and ds.n_sample == int(gold[f"ds/{tag}/n_sample"])
    idx = HC.sample_indices(len(ds), tag)
    assert np.array_equal(idx, gold[f"ds/{tag}/idx"])
    code, zs = gold[f"ds/{tag}/code"], gold[f"ds/{tag}/z"]
    sums, corners = gold[f"ds/{tag}/sum"], gold[f"ds/{tag}/corner"]
    G, t = HC.N_GRID, ds.tile_size
    bad = 0
    for n, i in enumerate(idx):
        i = int(i)
        z = ds.sample_idx_to_redshift(i)
        assert z == zs[n]
        got = np.stack([HC.corner_code(c100.get_stack("dm", z, i)), HC.corner_code(c150.get_stack("dm", z, i))])
        if not np.array_equal(got, code[n]):
            bad += 1
            continue
        # the same answer from the integer helpers the device-side assembler is built on
        s100, y100, x100, s150, y150, x150 = ds.sample_idx_to_tile(i)
        p100, p150 = ds.sample_idx_to_tile_permutation(i)
        for (s, ty, tx, p), c in (((s100, y100, x100, p100), code[n, 0]), ((s150, y150, x150, p150), code[n, 1])):
            rows, cols = np.meshgrid(np.arange(t), np.arange(t), indexing="ij")
            pr, pc = ds.apply_tile_permutation(rows, p), ds.apply_tile_permutation(cols, p)
            addr = lambda a, b: ((s * G + ty * t + pr[a, b]) * G + tx * t + pc[a, b]) + 1.0
            assert (addr(0, 0), addr(1, 0), addr(0, 1)) == tuple(c)
        sample, ri, rz = ds[i]
        assert ri == i and rz == z and len(sample) == 2
        for k, s in enumerate(sample):
            assert s.shape == (1, t, t)
            assert np.asarray(s, np.float64).sum() == pytest.approx(sums[n, k], rel=1e-12)
            assert np.array_equal(np.array([s[0, 0, 0], s[0, 0, -1], s[0, -1, 0], s[0, 5, 3]], np.float64), corners[n, k])
    assert bad == 0, f"{bad} of {len(idx)} indices map to other tiles than the reference's"
    st = np.array([[ds.stats[f][z]["mean"], ds.stats[f][z]["var"]] for f in HC.FIELDS for z in HC.REDSHIFTS])
    assert np.array_equal(st, gold[f"ds/{tag}/stats"])
    for i in (int(idx[3]), int(idx[len(idx) // 2]), int(idx[-1])):
        assert np.array_equal(np.asarray(ds.get_input_sample(i, transform=False), np.float64), gold[f"ds/{tag}/raw_input/{i}"])
        assert np.array_equal(np.asarray(ds.get_label_sample(i, transform=False)[0], np.float64),
                              gold[f"ds/{tag}/raw_label/{i}"])
    b, bi, bz = ds.get_batch(idx=idx[:5])
    assert tuple(b.shape) == tuple(gold[f"ds/{tag}/batch_shape"])
    assert np.allclose(np.asarray(b, np.float64).sum(axis=(2, 3, 4)), gold[f"ds/{tag}/batch_sum"], rtol=1e-12)


def test_transforms_equal_the_references(gold):
    from baryon_painter_amd.utils import data_transforms as T
    stats = HC.fiducial_like_stats()
    worst = 0.0
    for mi, (mode, k) in enumerate(HC.TRANSFORM_MODES):
        fwd, inv = T.create_range_compress_transforms({"dm": k, "pressure": k}, {"dm": mode, "pressure": mode}, eps=1e-3)
        for dt in (np.float32, np.float64):
            x = HC.transform_input(dt)
            for f in HC.FIELDS:
                for z in HC.TRANSFORM_Z:
                    with np.errstate(all="ignore"):
                        y = np.asarray(fwd(x, f, z, stats))
                        back = np.asarray(inv(y, f, z, stats))
                    key = f"tf/{mi}/{np.dtype(dt).name}/{f}/{z}"
                    for got, ref in ((y, gold[key + "/y"]), (back, gold[key + "/back"])):
                        assert got.dtype == ref.dtype and got.shape == ref.shape, key
                        fin = np.isfinite(ref)
                        assert np.array_equal(np.isfinite(got), fin), key
                        if fin.any():
                            err = np.abs(got[fin] - ref[fin]) / np.maximum(np.abs(ref[fin]), 1e-300)
                            worst = max(worst, float(err.max()))
    assert worst <= 2.3e-16, worst          # one float64 ulp (bit-equal on the build container's NumPy)
    # round trip within the reference's own test tolerance 2e-5*sigma (tests/test_dataset.py:80-83)
    fwd, inv = T.create_range_compress_transforms(HC.K_VALUES, HC.MODES)
    x = HC.transform_input(np.float32)
    for f in HC.FIELDS:
        for z in HC.TRANSFORM_Z:
            sigma = np.sqrt(T.interpolate_z(stats[f], z)["var"])
            assert np.abs(inv(fwd(x, f, z, stats), f, z, stats) - x).max() <= 2e-5 * sigma * max(1.0, x.max())


def test_lightcone_tiling_equals_the_references(gold):
    from baryon_painter_amd import lightcone as LC
    for ci, (n_plane, n_tile, ov) in enumerate(HC.TILING_CASES):
        origins, slices = LC.generate_tiling(n_plane, n_tile, min_tile_overlap=ov)
        assert np.array_equal(np.asarray(origins, np.float64), gold[f"tiling/{ci}/origins"])
        starts = np.array([[(s[0].start, s[0].stop, s[1].start, s[1].stop) for s in row] for row in slices], np.int64)
        assert np.array_equal(starts, gold[f"tiling/{ci}/starts"])
    for ci, (shape, falloff, sigma) in enumerate(HC.WEIGHT_CASES):
        assert np.array_equal(LC.make_weight_map(shape, falloff=falloff, sigma=sigma), gold[f"weight/{ci}"])
    for ci, (n, shift, rel, exp) in enumerate(HC.TILE_CASES):
        assert np.array_equal(LC.get_tile(HC.plane(n), shift, rel, expansion_factor=exp), gold[f"tile/{ci}"])
