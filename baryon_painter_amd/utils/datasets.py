"""Tile datasets that feed ``CVAEPainter.train`` / ``validate``.

``BAHAMASDataset`` keeps the reference's public surface and its sample-index -> tile mapping
(/root/reference/baryon_painter/utils/datasets.py:15-508) -- including the mapping's quirks,
which define "bit-exact tile indexing":
  * the (stack, tile) part of the index is taken ``% n_tile_permutation**2`` (datasets.py:327),
    so with permutations on only 64 (stack, tile) combinations are ever visited and with
    permutations off only the first;
  * flip code 3 is a no-op (duplicated ``elif flip_idx == 2``, datasets.py:353-358);
  * ``get_batch`` multiplies the drawn indices by ``len(redshifts)`` (datasets.py:459).
``fixed_indexing=True`` opts out of the first two.

``SyntheticTileDataset`` implements the same protocol on seeded synthetic tiles (no files).
"""
import collections
import copy
import os

import numpy as np

from . import synthetic as syn

# dark-matter mass -> SLICS delta-plane units (datasets.py:301-302, 399)
def slics_scale(n_grid):
    return 1 / (n_grid / 8 * 5) * 0.2793 / (0.2793 - 0.0463)


def _identity(x, field, z, stats):
    return x


class _Compiled:
    """``f(x, field=..., z=...)`` with the transform and statistics bound (datasets.py:8-13);
    a class instead of a lambda so that checkpoints pickle without dill's closure support."""

    def __init__(self, transform, stats, field=None, z=None):
        self.func = copy.deepcopy(transform)
        self.stats = copy.deepcopy(stats)
        self.field = copy.deepcopy(field)
        self.z = copy.deepcopy(z)

    def __call__(self, x, field=None, z=None):
        return self.func(x, self.field if field is None else field, self.z if z is None else z, self.stats)


def compile_transform(transform, stats={}, field=None, z=None):
    return _Compiled(transform, stats, field, z)


class _TileDatasetBase:
    """What the painter reads from a dataset (painter.py:114-116,307,321,399-412)."""

    def create_transform(self, field, z):
        return compile_transform(self.transform_func, self.stats, field, z)

    def create_inverse_transform(self, field, z):
        return compile_transform(self.inverse_transform_func, self.stats, field, z)

    def _field_list(self):
        return [self.input_field] + self.label_fields

    def get_transforms(self, idx=None, z=None):
        if idx is None and z is None:
            raise ValueError("Either idx or z have to be specified.")
        z = self.sample_idx_to_redshift(idx) if z is None else z
        return [self.create_transform(f, z) for f in self._field_list()]

    def get_inverse_transforms(self, idx=None, z=None):
        if idx is None and z is None:
            raise ValueError("Either idx or z have to be specified.")
        z = self.sample_idx_to_redshift(idx) if z is None else z
        return [self.create_inverse_transform(f, z) for f in self._field_list()]

    def sample_idx_to_redshift(self, idx):
        return self.redshifts[idx // self.n_sample]

    def __len__(self):
        return self.n_sample * len(self.redshifts)

    def get_batch(self, size=1, z=None, idx=None):
        """Random batch: (fields (1+F, N, C, H, W), indices, redshifts) -- datasets.py:434-473."""
        if idx is None:
            idx = np.random.choice(self.n_sample, size=size, replace=False)
            if z is None:
                idx *= len(self.redshifts)            # sic (datasets.py:459)
                z = [self.sample_idx_to_redshift(i) for i in idx]
            else:
                idx += self.redshifts.index(z) * self.n_sample
                z = [z] * size
        else:
            z = [self.sample_idx_to_redshift(i) for i in idx]
        samples = [self[i][0] for i in idx]
        return np.array(samples).swapaxes(0, 1), idx, np.array(z)


class BAHAMASDataset(_TileDatasetBase):
    def __init__(self, data=None, files=None, root_path=None, redshifts=[], input_field="dm", label_fields=[],
                 n_tile=4, L=400, n_stack=None, stack_offset=0, transform=_identity, inverse_transform=_identity,
                 n_feature_per_field=1, tile_permutations=False, scale_to_SLICS=True, subtract_minimum=False,
                 mmap_mode="r", verbose=False, fixed_indexing=False):
        fields, zs = [], []
        if data is not None:
            self.data = data
            fields = list(data.keys())
            zs = list(data[fields[0]].keys())
        elif files is not None:
            self.data = {}
            for f in files:
                if not isinstance(f, dict):
                    raise ValueError("files entry is not a dict.")
                fields.append(f["field"])
                zs.append(f["z"])
        else:
            raise ValueError("Either data or files need to be provided.")
        self.fields = list(collections.OrderedDict.fromkeys(fields))
        self.redshifts = list(collections.OrderedDict.fromkeys(zs))
        self.input_field = input_field
        if label_fields != []:
            self.label_fields = label_fields
            missing = set([input_field] + label_fields) - set(self.fields)
            if missing:
                raise ValueError(f"The requested fields are not in the file list: field(s) {missing} is missing.")
            self.fields = [input_field] + label_fields
        else:
            self.label_fields = [f for f in self.fields if f != input_field]
        if redshifts != []:
            missing = set(redshifts) - set(self.redshifts)
            if missing:
                raise ValueError(f"The requested redshifts are not in the file list: redshift(s) {missing} is missing.")
            self.redshifts = redshifts
        else:
            self.redshifts = sorted(self.redshifts)

        if files is not None:
            for f in files:
                field, z = f["field"], f["z"]
                if field not in self.fields or z not in self.redshifts:
                    continue
                entry = self.data.setdefault(field, {}).setdefault(z, {})
                for slab in ("100", "150"):
                    fn = f["file_" + slab]
                    if root_path is not None:
                        fn = os.path.join(root_path, fn)
                    entry[slab] = np.load(fn, mmap_mode=mmap_mode)
                    entry["mean_" + slab] = f["mean_" + slab]
                    entry["var_" + slab] = f["var_" + slab]

        first = self.data[self.fields[0]][self.redshifts[0]]
        self.n_stack_100, self.n_grid, _ = first["100"].shape
        self.n_stack_150 = first["150"].shape[0]
        self.n_stack = min(self.n_stack_100, self.n_stack_150) if n_stack is None else n_stack
        self.stack_offset = stack_offset
        if min(self.n_stack_100, self.n_stack_150) < self.stack_offset + self.n_stack:
            raise ValueError("Highest stack exceeds number of available stacks.")
        self.n_tile_permutation = 8 if tile_permutations else 1
        self.n_tile = n_tile
        self.tile_size = self.n_grid // n_tile
        self.n_total_sample = ((self.n_stack_100 * n_tile ** 2 * self.n_tile_permutation)
                               * (self.n_stack_150 * n_tile ** 2 * self.n_tile_permutation))
        self.n_sample = self.n_stack ** 2 * n_tile ** 4 * self.n_tile_permutation ** 2
        self.L = L
        self.tile_L = L / n_tile
        self.transform_func, self.inverse_transform_func = transform, inverse_transform
        self.n_feature_per_field = n_feature_per_field
        self.scale_to_SLICS, self.subtract_minimum = scale_to_SLICS, subtract_minimum
        self.fixed_indexing = fixed_indexing
        self.stats = collections.OrderedDict()
        for field in self.fields:
            self.stats[field] = collections.OrderedDict((z, self.get_stack_stats(field, z)) for z in self.redshifts)
        self.transform = compile_transform(transform, self.stats)
        self.inverse_transform = compile_transform(inverse_transform, self.stats)

    def get_stack_stats(self, field, z):
        d = self.data[field][z]
        stats = {"mean": d["mean_100"] + d["mean_150"], "var": d["var_100"] + d["var_150"]}
        if field == self.input_field and self.scale_to_SLICS:
            s = slics_scale(self.n_grid)
            stats["mean"] *= s
            stats["var"] *= s ** 2
        return stats

    # ---- the index arithmetic (pure integers)
    def sample_idx_to_tile_permutation(self, idx):
        sample_idx = idx % self.n_sample
        p = sample_idx // (self.n_sample // self.n_tile_permutation ** 2)
        return tuple(int(v) for v in np.unravel_index(p, (self.n_tile_permutation,) * 2))

    def sample_idx_to_tile(self, flat_idx):
        """-> (stack_100, ty_100, tx_100, stack_150, ty_150, tx_150) of a flat sample index."""
        no_z = flat_idx % self.n_sample
        if self.fixed_indexing:
            base = no_z % (self.n_sample // self.n_tile_permutation ** 2)
        else:
            base = no_z % self.n_tile_permutation ** 2          # sic (datasets.py:327)
        i = np.unravel_index(base, (self.n_stack, self.n_tile, self.n_tile) * 2)
        i = [int(v) for v in i]
        return (i[0] + self.stack_offset, i[1], i[2], i[3] + self.stack_offset, i[4], i[5])

    def apply_tile_permutation(self, tile, permutation_idx):
        rot, flip = permutation_idx // 4, permutation_idx % 4
        if rot > 0:
            tile = np.rot90(tile, k=rot)
        if flip == 1:
            tile = tile[:, ::-1]
        elif flip == 2:
            tile = tile[::-1]
        elif flip == 3 and self.fixed_indexing:
            tile = tile[::-1, ::-1]                               # unreachable in the reference
        return tile

    def get_stack(self, field, z, flat_idx):
        s100, y100, x100, s150, y150, x150 = self.sample_idx_to_tile(flat_idx)
        t = self.tile_size
        d100 = self.data[field][z]["100"][s100][y100 * t:(y100 + 1) * t, x100 * t:(x100 + 1) * t]
        d150 = self.data[field][z]["150"][s150][y150 * t:(y150 + 1) * t, x150 * t:(x150 + 1) * t]
        p100, p150 = self.sample_idx_to_tile_permutation(flat_idx)
        return self.apply_tile_permutation(d100, p100) + self.apply_tile_permutation(d150, p150)

    def get_input_sample(self, idx, transform=True):
        z = self.sample_idx_to_redshift(idx)
        d = self.get_stack(self.input_field, z, idx)
        if self.scale_to_SLICS:
            d = slics_scale(self.n_grid) * d
        if self.subtract_minimum:
            d = d - d.min()
        return self.transform(d, self.input_field, z) if transform else d

    def get_label_sample(self, idx, transform=True):
        z = self.sample_idx_to_redshift(idx)
        out = []
        for f in self.label_fields:
            d = self.get_stack(f, z, idx)
            out.append(self.transform(d, f, z) if transform else d)
        return out

    def __getitem__(self, idx):
        if isinstance(idx, collections.abc.Iterable):
            raise NotImplementedError("Only int indicies are supported for now.")
        return [self.get_input_sample(idx)] + self.get_label_sample(idx), idx, self.sample_idx_to_redshift(idx)


class SyntheticTileDataset(_TileDatasetBase):
    """Seeded synthetic (dm, pressure) tiles in the transformed domain, same protocol as
    ``BAHAMASDataset``; sample ``i`` is a pure function of ``(seed, i)``."""

    def __init__(self, n_sample=256, tile_size=512, redshifts=syn.REDSHIFTS, seed=0, L=400, n_tile=4):
        self.n_sample = n_sample
        self.redshifts = list(redshifts)
        self.tile_size = tile_size
        self.seed = seed
        self.input_field, self.label_fields = "dm", ["pressure"]
        self.fields = ["dm", "pressure"]
        self.n_feature_per_field = 1
        self.L, self.n_tile = L, n_tile
        self.n_grid = tile_size * n_tile
        self.tile_L = L / n_tile
        self.scale_to_SLICS = False
        self.stats = collections.OrderedDict()
        for f in self.fields:
            self.stats[f] = collections.OrderedDict(
                (z, {"mean": 1.0, "var": syn.field_sigma(f, z) ** 2}) for z in self.redshifts)
        from . import data_transforms as T
        fwd, inv = T.create_range_compress_transforms({"dm": 4.0, "pressure": 4}, {"dm": "shift-log",
                                                                                  "pressure": "shift-log"})
        self.transform_func = T.chain_transformations([fwd, T.atleast_3d, T.as_float32])
        self.inverse_transform_func = T.chain_transformations([T.squeeze, inv])
        self.transform = compile_transform(self.transform_func, self.stats)
        self.inverse_transform = compile_transform(self.inverse_transform_func, self.stats)

    def raw_fields(self, idx):
        z = self.sample_idx_to_redshift(idx)
        rng = np.random.Generator(np.random.PCG64([self.seed, idx]))
        t = self.tile_size
        g = np.sqrt(-2 * np.log(1 - rng.random((t, t)))) * np.cos(2 * np.pi * rng.random((t, t)))
        g2 = np.sqrt(-2 * np.log(1 - rng.random((t, t)))) * np.cos(2 * np.pi * rng.random((t, t)))
        dm = (np.exp(0.9 * g - 0.4) * syn.field_sigma("dm", z)).astype(np.float32)
        pr = (0.05 * np.exp(0.8 * g + 0.6 * g2 - 0.5) * syn.field_sigma("pressure", z) / 0.05).astype(np.float32)
        return dm, pr, z

    def __getitem__(self, idx):
        if isinstance(idx, collections.abc.Iterable):
            raise NotImplementedError("Only int indicies are supported for now.")
        dm, pr, z = self.raw_fields(int(idx))
        return [self.transform(dm, "dm", z), self.transform(pr, "pressure", z)], idx, z


class DeviceTileAssembler:
    """Batch assembly on the GPU for a ``BAHAMASDataset``: the stacks are uploaded to HBM once
    (fiducial training set: 2 fields x 11 redshifts x 2 slabs x 14 x 2048^2 floats = 10 GB of the
    288 GB) and every batch is one gather launch per field that reads the two tiles of each sample,
    applies the tile permutation, adds them, scales and transforms them -- bit-for-bit the index
    arithmetic of ``BAHAMASDataset.get_stack`` (computed on the host, integers only), so 8 GPUs are not
    starved by a serial ``DataLoader(num_workers=0)`` (painter.py:88).  Supported transform: the
    reference's "shift-log" range compression (plus identity)."""

    REC100 = np.dtype([("base", np.int64), ("pitch", np.int32), ("r0", np.int32), ("rr", np.int32),
                       ("rc", np.int32), ("c0", np.int32), ("cr", np.int32), ("cc", np.int32), ("pad", np.int32)])
    XF = np.dtype([("scale", np.float64), ("inv_sigma", np.float64), ("inv_k", np.float64),
                   ("mode", np.int32), ("pad", np.int32)])

    def __init__(self, dataset, device="cuda:0", k_values=None, mode="shift-log"):
        import torch
        from .. import _lib as L
        from . import data_transforms as T
        self.ds, self.device, self.torch, self.L = dataset, torch.device(device), torch, L
        self.lib = L.load()
        if mode not in ("shift-log", None):
            raise NotImplementedError("DeviceTileAssembler implements the 'shift-log' transform only")
        self.mode = mode
        self.k_values = k_values or {}
        self._interp = T.interpolate_z
        self.stacks = {}
        for f in dataset.fields:
            for z in dataset.redshifts:
                for slab in ("100", "150"):
                    a = np.ascontiguousarray(dataset.data[f][z][slab], dtype=np.float32)
                    self.stacks[(f, z, slab)] = torch.from_numpy(a).to(self.device)

    def _perm_affine(self, p):
        """(r0, rr, rc, c0, cr, cc) of ``apply_tile_permutation`` for code p, from the images of three
        index points under the very same NumPy operations."""
        t = self.ds.tile_size
        rows, cols = np.meshgrid(np.arange(t), np.arange(t), indexing="ij")
        pr = self.ds.apply_tile_permutation(rows, p)
        pc = self.ds.apply_tile_permutation(cols, p)
        r0, c0 = int(pr[0, 0]), int(pc[0, 0])
        return (r0, int(pr[1, 0]) - r0, int(pr[0, 1]) - r0, c0, int(pc[1, 0]) - c0, int(pc[0, 1]) - c0)

    def _descriptors(self, field, indices):
        ds, t = self.ds, self.ds.tile_size
        d100 = np.zeros(len(indices), self.REC100)
        d150 = np.zeros(len(indices), self.REC100)
        xf = np.zeros(len(indices), self.XF)
        for n, idx in enumerate(indices):
            idx = int(idx)
            z = ds.sample_idx_to_redshift(idx)
            s100, y100, x100, s150, y150, x150 = ds.sample_idx_to_tile(idx)
            p100, p150 = ds.sample_idx_to_tile_permutation(idx)
            for rec, slab, s, ty, tx, p in ((d100, "100", s100, y100, x100, p100), (d150, "150", s150, y150, x150, p150)):
                st = self.stacks[(field, z, slab)]
                g = st.shape[-1]
                rec[n]["base"] = st.data_ptr() + 4 * ((s * g + ty * t) * g + tx * t)
                rec[n]["pitch"] = g
                (rec[n]["r0"], rec[n]["rr"], rec[n]["rc"], rec[n]["c0"], rec[n]["cr"], rec[n]["cc"]) = self._perm_affine(p)
            scale = slics_scale(ds.n_grid) if (field == ds.input_field and ds.scale_to_SLICS) else 1.0
            xf[n]["scale"] = scale
            if self.mode == "shift-log":
                sig = np.sqrt(self._interp(ds.stats[field], z)["var"])
                xf[n]["inv_sigma"] = 1.0 / sig
                xf[n]["inv_k"] = 1.0 / float(self.k_values[field])
                xf[n]["mode"] = 1
        return d100, d150, xf

    def get_batch(self, indices):
        """-> (x = label field(s) (N,F,t,t), y = input field (N,1,t,t), aux = redshift (N,)) on the device."""
        import ctypes as C
        torch, L, ds, t = self.torch, self.L, self.ds, self.ds.tile_size
        if ds.subtract_minimum:
            raise NotImplementedError("subtract_minimum on the device path")
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        outs, keep = [], []
        for field in [ds.input_field] + ds.label_fields:
            d100, d150, xf = self._descriptors(field, indices)
            dev = [torch.from_numpy(a.view(np.uint8)).to(self.device) for a in (d100, d150, xf)]
            keep.append(dev)
            out = torch.empty((len(indices), 1, t, t), device=self.device)
            L.check(self.lib.bp_gather_tiles(L.ptr(dev[0]), L.ptr(dev[1]), L.ptr(dev[2]), len(indices), t,
                                             L.ptr(out), st), "gather tiles")
            outs.append(out)
        torch.cuda.current_stream().synchronize()        # descriptors may be freed after this
        z = torch.tensor([ds.sample_idx_to_redshift(int(i)) for i in indices], device=self.device,
                         dtype=torch.float32)
        return torch.cat(outs[1:], dim=1), outs[0], z
