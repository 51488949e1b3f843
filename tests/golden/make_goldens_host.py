#!/usr/bin/env python3
"""Generate tests/golden/host.npz by IMPORTING THE REAL REFERENCE (build container only): the rows of SURVEY.md 8f.

    python tests/golden/make_goldens_host.py

What is pinned (inputs come from tests/host_cases.py seeds; only the reference's outputs are stored):
  * BAHAMASDataset (utils/datasets.py:15-508): for six (n_stack, stack_offset, tile_permutations) settings and ~1000
    indices each, which pixels of which stack land where -- decoded from address-coded stacks that went through the
    reference's own get_stack() -- plus dataset[idx] on random stacks (transformed tiles: full sums, corner values);
  * the range-compression transforms and their inverses (utils/data_transforms.py:51-110) at tabulated,
    interpolated and out-of-range redshifts;
  * generate_tiling / get_tile / make_weight_map (process_SLICS.py:68-126);
  * CVAEPainter.save_state_to_file -> load -> paint (painter.py:371-445) on a 64x64 model: transform, sample_P,
    inverse transform end to end.

Harness shims (none of them touches the arithmetic): empty stub modules for the absent plotting /
cosmology packages the reference imports at module level (cosmotools, pyccl, astropy.io.fits), and
``np.unravel_index`` given back the ``dims=`` keyword NumPy removed (datasets.py:329,372).
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")

for name in ("cosmotools", "cosmotools.utils", "cosmotools.power_spectrum_tools", "cosmotools.plotting", "pyccl",
             "astropy", "astropy.io", "astropy.io.fits"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["cosmotools.utils"].rebin_2d = None
sys.modules["astropy.io"].fits = sys.modules["astropy.io.fits"]
_unravel = np.unravel_index
np.unravel_index = lambda indices, shape=None, order="C", dims=None: _unravel(indices, shape if dims is None else dims, order)

from baryon_painter.utils import datasets as ref_ds                  # noqa: E402  (the reference)
from baryon_painter.utils import data_transforms as ref_T            # noqa: E402
from baryon_painter import process_SLICS as ref_S                    # noqa: E402
from baryon_painter import painter as ref_painter                    # noqa: E402
import host_cases as HC                                              # noqa: E402
from baryon_painter_amd.utils import synthetic as syn                # noqa: E402
from make_goldens import inject_eps, load_params                     # noqa: E402


def dataset_goldens(out):
    fwd, inv = ref_T.create_range_compress_transforms(HC.K_VALUES, HC.MODES)
    tr = ref_T.chain_transformations([fwd, ref_T.atleast_3d])
    with tempfile.TemporaryDirectory() as tmp:
        # the file path of the constructor (np.load + mmap) once, on the random stacks
        for f in HC.FIELDS:
            for zi, z in enumerate(HC.REDSHIFTS):
                for slab in ("100", "150"):
                    np.save(os.path.join(tmp, f"{f}_z{zi}_{slab}.npy"), HC.random_stack(f, zi, slab))
        for tag, n_stack, off, perm in HC.DATASET_CASES:
            kw = dict(redshifts=list(HC.REDSHIFTS), label_fields=["pressure"], n_tile=HC.N_TILE, n_stack=n_stack,
                      stack_offset=off, tile_permutations=perm, scale_to_SLICS=True)
            ds = ref_ds.BAHAMASDataset(files=HC.stack_files_info(), root_path=tmp, transform=tr, inverse_transform=inv,
                                       **kw)
            c100 = ref_ds.BAHAMASDataset(data=HC.data_dict("coded100"), **kw)
            c150 = ref_ds.BAHAMASDataset(data=HC.data_dict("coded150"), **kw)
            idx = HC.sample_indices(len(ds), tag)
            out[f"ds/{tag}/len"] = np.array(len(ds))
            out[f"ds/{tag}/n_sample"] = np.array(ds.n_sample)
            out[f"ds/{tag}/idx"] = idx
            codes = np.zeros((len(idx), 2, 3))
            zs = np.zeros(len(idx))
            sums = np.zeros((len(idx), 2))
            corners = np.zeros((len(idx), 2, 4), np.float64)
            for n, i in enumerate(idx):
                i = int(i)
                z = ds.sample_idx_to_redshift(i)
                zs[n] = z
                codes[n, 0] = HC.corner_code(c100.get_stack("dm", z, i))
                codes[n, 1] = HC.corner_code(c150.get_stack("dm", z, i))
                sample, ri, rz = ds[i]
                assert ri == i and rz == z and len(sample) == 2
                for k, s in enumerate(sample):
                    assert s.shape == (1, ds.tile_size, ds.tile_size), s.shape
                    sums[n, k] = np.asarray(s, np.float64).sum()
                    corners[n, k] = [s[0, 0, 0], s[0, 0, -1], s[0, -1, 0], s[0, 5, 3]]
            out[f"ds/{tag}/code"] = codes
            out[f"ds/{tag}/z"] = zs
            out[f"ds/{tag}/sum"] = sums
            out[f"ds/{tag}/corner"] = corners
            st = ds.stats
            out[f"ds/{tag}/stats"] = np.array([[st[f][z]["mean"], st[f][z]["var"]] for f in HC.FIELDS for z in HC.REDSHIFTS])
            # raw (untransformed) input / label tiles of three indices, whole
            for i in (int(idx[3]), int(idx[len(idx) // 2]), int(idx[-1])):
                out[f"ds/{tag}/raw_input/{i}"] = np.asarray(ds.get_input_sample(i, transform=False), np.float64)
                out[f"ds/{tag}/raw_label/{i}"] = np.asarray(ds.get_label_sample(i, transform=False)[0], np.float64)
            # get_batch with explicit indices
            b, bi, bz = ds.get_batch(idx=idx[:5])
            out[f"ds/{tag}/batch_shape"] = np.array(b.shape)
            out[f"ds/{tag}/batch_sum"] = np.asarray(b, np.float64).sum(axis=(2, 3, 4))
            print(tag, "len", len(ds), "probed", len(idx))


def transform_goldens(out):
    stats = HC.fiducial_like_stats()
    for mi, (mode, k) in enumerate(HC.TRANSFORM_MODES):
        fwd, inv = ref_T.create_range_compress_transforms({"dm": k, "pressure": k}, {"dm": mode, "pressure": mode},
                                                          eps=1e-3)
        for dt in (np.float32, np.float64):
            x = HC.transform_input(dt)
            for f in HC.FIELDS:
                for z in HC.TRANSFORM_Z:
                    with np.errstate(all="ignore"):
                        y = fwd(x, f, z, stats)
                        back = inv(y, f, z, stats)
                    key = f"tf/{mi}/{np.dtype(dt).name}/{f}/{z}"
                    out[key + "/y"] = np.asarray(y)
                    out[key + "/back"] = np.asarray(back)


def tiling_goldens(out):
    for ci, (n_plane, n_tile, ov) in enumerate(HC.TILING_CASES):
        origins, slices = ref_S.generate_tiling(n_plane, n_tile, min_tile_overlap=ov)
        out[f"tiling/{ci}/origins"] = np.asarray(origins, np.float64)
        out[f"tiling/{ci}/starts"] = np.array([[(s[0].start, s[0].stop, s[1].start, s[1].stop) for s in row]
                                               for row in slices], np.int64)
    for ci, (shape, falloff, sigma) in enumerate(HC.WEIGHT_CASES):
        out[f"weight/{ci}"] = ref_S.make_weight_map(shape, falloff=falloff, sigma=sigma)
    for ci, (n, shift, rel, exp) in enumerate(HC.TILE_CASES):
        out[f"tile/{ci}"] = ref_S.get_tile(HC.plane(n), shift, rel, expansion_factor=exp)


def paint_goldens(out):
    """The reference painter end to end: state + meta files written by save_state_to_file, loaded by a fresh
    CVAEPainter, paint() with an injected eps (transform -> sample_P in eval mode -> inverse transform)."""
    from baryon_painter_amd.models import arch as our_arch
    size = 64
    arch = our_arch.fiducial_architecture(size)
    fwd, inv = ref_T.create_range_compress_transforms(HC.K_VALUES, HC.MODES)
    tr = ref_T.chain_transformations([fwd, ref_T.atleast_3d, lambda x, field, z, stats: x.astype(np.float32)])
    itr = ref_T.chain_transformations([ref_T.squeeze, inv])
    ds = ref_ds.BAHAMASDataset(data=HC.data_dict("random"), redshifts=list(HC.REDSHIFTS), label_fields=["pressure"],
                               n_tile=1, n_stack=3, transform=tr, inverse_transform=itr, scale_to_SLICS=True)
    p = ref_painter.CVAEPainter(training_data_set=ds, test_data_set=ds, architecture=arch, compute_device="cpu")
    load_params(p.model, 7)
    # one train-mode forward so that the running statistics are not the initial ones
    x, y, aux = syn.synthetic_batch(3, size, size, seed=77)
    with inject_eps(syn.synthetic_eps((1, 3, *arch["dim_z"]), seed=78)), torch.no_grad():
        p.model(torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(aux))
    with tempfile.TemporaryDirectory() as tmp:
        files = (os.path.join(tmp, "state"), os.path.join(tmp, "meta"))
        p.save_state_to_file(files)
        q = ref_painter.CVAEPainter(filename=files)
    for k, b in q.model.named_buffers():
        out[f"paint/buf/{k}"] = b.numpy()
    for ti, (idx, z) in enumerate(((0, 0.0), (1, 0.3), (2, 2.0))):
        tile = np.asarray(ds.get_input_sample(idx, transform=False), np.float32)      # raw dm tile (64, 64)
        eps = syn.synthetic_eps((1, 1, *arch["dim_z"]), seed=80 + ti)
        with inject_eps(eps):
            painted = q.paint(tile, z=z)
        with inject_eps(eps):
            raw = q.paint(tile, z=z, inverse_transform=False)
        assert painted.shape == (size, size) and raw.shape == (1, 1, size, size)
        out[f"paint/{ti}/painted"] = np.asarray(painted, np.float64)
        out[f"paint/{ti}/net_output"] = np.asarray(raw, np.float64)
    print("paint goldens:", painted.dtype, float(np.abs(painted).max()))


def main():
    out = {}
    dataset_goldens(out)
    transform_goldens(out)
    tiling_goldens(out)
    paint_goldens(out)
    path = os.path.join(HERE, "host.npz")
    np.savez_compressed(path, **out)
    print("host.npz", os.path.getsize(path) // 1024, "KiB,", len(out), "arrays")


if __name__ == "__main__":
    main()
