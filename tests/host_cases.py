"""Case definitions shared by tests/golden/make_goldens_host.py (which feeds them to the real reference in the
build container) and tests/test_host_golden.py / test_gpu_painter.py (which feed them to this repo's code).

Everything is regenerated from seeds, so only the reference's OUTPUTS are stored in tests/golden/host.npz.
"""
import numpy as np

REDSHIFTS = (0.0, 0.5, 2.0)
FIELDS = ("dm", "pressure")
N_GRID, N_TILE, N_STACK_FILE = 64, 4, 14           # files hold 14 stacks like the real ones (tile 16x16)

# (tag, n_stack, stack_offset, tile_permutations): the training set (11, 3, on), the validation set (3, 0, off) of
# scripts/CVAE_single_scale.py:67-90 and the small / mixed settings VERDICT r01 asked for
DATASET_CASES = [("train11", 11, 3, True), ("val3", 3, 0, False), ("perm3", 3, 0, True), ("perm2", 2, 1, True),
                 ("plain2", 2, 0, False), ("plain11", 11, 0, False)]

K_VALUES = {"dm": 4.0, "pressure": 4}              # scripts/CVAE_single_scale.py:35-38
MODES = {"dm": "shift-log", "pressure": "shift-log"}


def stack_files_info():
    """The ``files`` list of dicts BAHAMASDataset takes (means / variances are arbitrary but field- and
    redshift-dependent, so that a mixed-up lookup shows)."""
    out = []
    for fi, f in enumerate(FIELDS):
        for zi, z in enumerate(REDSHIFTS):
            out.append({"field": f, "z": z, "file_100": f"{f}_z{zi}_100.npy", "file_150": f"{f}_z{zi}_150.npy",
                        "mean_100": 0.4 + 0.1 * fi + 0.01 * zi, "mean_150": 0.6 + 0.05 * fi + 0.02 * zi,
                        "var_100": 0.07 / (1 + zi) * (1 + fi), "var_150": 0.09 / (1 + zi) * (1 + 2 * fi)})
    return out


def random_stack(field, zi, slab):
    """Seeded positive float32 stack (14, 64, 64)."""
    rng = np.random.Generator(np.random.PCG64([11, FIELDS.index(field), zi, int(slab)]))
    return (rng.random((N_STACK_FILE, N_GRID, N_GRID)) * 3.0 + 0.01).astype(np.float32)


def coded_stack(slab, which):
    """Stack whose pixel values spell their own address: value = ((s*G + r)*G + c) + 1 for the slab ``which``
    ("100" or "150"), zeros for the other -- exact in float32 (< 2^24), so that the sum the dataset returns
    decodes to (stack, row, column) of the pixel that landed at each output position."""
    if slab != which:
        return np.zeros((N_STACK_FILE, N_GRID, N_GRID), np.float32)
    return (np.arange(N_STACK_FILE * N_GRID * N_GRID, dtype=np.float32) + 1.0).reshape(N_STACK_FILE, N_GRID, N_GRID)


def data_dict(kind):
    """``data=`` argument of BAHAMASDataset: kind "random", "coded100" or "coded150"."""
    info = {(d["field"], d["z"]): d for d in stack_files_info()}
    data = {}
    for f in FIELDS:
        data[f] = {}
        for zi, z in enumerate(REDSHIFTS):
            d = info[(f, z)]
            e = {k: d[k] for k in ("mean_100", "mean_150", "var_100", "var_150")}
            for slab in ("100", "150"):
                e[slab] = random_stack(f, zi, slab) if kind == "random" else coded_stack(slab, kind[-3:])
            data[f][z] = e
    return data


def sample_indices(n_total, tag):
    """Indices probed per dataset case: the first 200, the last 8, the redshift boundaries and 800 seeded ones."""
    rng = np.random.Generator(np.random.PCG64([17, sum(map(ord, tag))]))
    n_z = len(REDSHIFTS)
    per = n_total // n_z
    fixed = list(range(min(200, n_total))) + list(range(max(0, n_total - 8), n_total))
    fixed += [per - 1, per, per + 1, 2 * per - 1, 2 * per, 2 * per + 63, 2 * per + 64]
    rnd = (rng.random(800) * n_total).astype(np.int64).tolist()
    return np.array(sorted(set(i for i in fixed + rnd if 0 <= i < n_total)), dtype=np.int64)


def corner_code(tile):
    """(t[0,0], t[1,0], t[0,1]) of an address-coded tile: origin + the images of the two unit steps fix the stack,
    the tile and the dihedral permutation."""
    return np.array([tile[0, 0], tile[1, 0], tile[0, 1]], dtype=np.float64)


# ---- transforms
TRANSFORM_Z = (0.0, 0.125, 0.3, 0.5, 1.7, 2.0, 2.5, -0.2)
TRANSFORM_MODES = [("shift-log", 4.0), ("shift-log", 4), ("log", 2.0), ("shift-log-2p", (0.5, 3.0)), ("log-tanh", 6.0),
                   ("x/(1+x)", (2.0, 1.0)), ("1/x", 2.0)]


def fiducial_like_stats():
    """stats[field][z] for the 11 training redshifts with the fiducial end values of SURVEY.md 8c-vii."""
    from baryon_painter_amd.utils import synthetic as syn
    import collections
    stats = collections.OrderedDict()
    for f, (m0, m2) in (("dm", (1.0018, 1.0008)), ("pressure", (0.0442, 0.0075))):
        stats[f] = collections.OrderedDict()
        for z in syn.REDSHIFTS:
            w = z / 2.0
            stats[f][z] = {"mean": (1 - w) * m0 + w * m2, "var": syn.field_sigma(f, z) ** 2}
    return stats


def transform_input(dtype):
    rng = np.random.Generator(np.random.PCG64(23))
    x = np.exp(rng.random((16, 16)) * 6.0 - 4.0)
    x[0, 0] = 0.0
    return x.astype(dtype)


# ---- light-cone tiling
TILING_CASES = [(512, 256, 0.0), (512, 250, 0.0), (512, 256, 0.5), (512, 128, 0.0),      # the reference's 4 asserts
                (7745, 512, 0.5), (1536, 512, 0.5), (1000, 512, 0.33), (600, 512, 0.5), (512, 512, 0.5)]
WEIGHT_CASES = [((100, 100), 0.05, 0.5), ((512, 512), 0.05, 0.5), ((64, 64), 0.1, 1), ((40, 40), 0.05, 1)]
TILE_CASES = [(24, (0.75, 0.5), 0.5, 1), (24, (0.9, 0.95), 0.25, 1), (24, (0.0, 0.3), 1 / 3, 1.5), (10, (0.55, 0.0), 0.7, 1)]


def plane(n):
    rng = np.random.Generator(np.random.PCG64([29, n]))
    return rng.random((n, n))
