"""Tiling / blending of large mass planes around ``painter.paint`` -- the production caller of
the hot path (/root/reference/baryon_painter/process_SLICS.py:68-126, 198-220).

Only the integer tiling, the wrap-around tile cut, the feathering weights and the blend are here;
the cosmology of ``create_y_map`` (pyccl / astropy) and the SLICS file handling are out of scope.
Unlike the reference's serial per-tile loop, ``paint_plane`` sends all tiles of a plane through
``CVAEPainter.paint_batch`` (hipGraph-captured batches).
"""
import numpy as np


def generate_tiling(n_pixel_plane, n_pixel_tile, min_tile_overlap=0.5):
    """Origins (as fractions of the plane) and slices of a regular grid of square tiles that covers
    the plane with at least ``min_tile_overlap`` relative overlap between neighbours
    (process_SLICS.py:102-126; known answers in the reference's tests/test_SLICS_tiling.py:72-81)."""
    rel = n_pixel_tile / n_pixel_plane
    n_inner = 0
    if rel < 1 - rel + rel * min_tile_overlap:                 # two tiles do not overlap enough
        step = rel * (1 - min_tile_overlap)
        gap = 1 - 2 * rel + rel * min_tile_overlap
        n_inner = 1 if gap <= step else int(np.ceil((gap - step) / step)) + 1
    origins = np.linspace(0, 1 - rel, n_inner + 2, endpoint=True)
    px = [int(o * n_pixel_plane) for o in origins]
    slices = [[np.s_[x:x + n_pixel_tile, y:y + n_pixel_tile] for y in px] for x in px]
    return origins, slices


def get_tile(m, shift, tile_relative_size, expansion_factor=1):
    """Square cut-out of a periodic plane starting at ``shift`` (fractions), wrapping around the
    edges (process_SLICS.py:68-83)."""
    if expansion_factor < 1:
        raise ValueError("Expension factors < 1 not supported.")
    n = m.shape[0]
    size = int(n * tile_relative_size * expansion_factor)
    off = int(n * tile_relative_size * (expansion_factor - 1) / 2)
    x0, y0 = int(n * shift[0]) - off, int(n * shift[1]) - off
    return m.take(range(x0, x0 + size), axis=0, mode="wrap").take(range(y0, y0 + size), axis=1, mode="wrap")


def make_weight_map(tile_shape, falloff=0.05, sigma=1):
    """Feathering weights: 1 inside, Gaussian roll-off over ``falloff`` of the tile size at every
    edge (process_SLICS.py:85-99)."""
    w = np.ones(tile_shape)
    n_edge = int(tile_shape[0] * falloff)
    s = n_edge * sigma
    for i in range(n_edge):
        f = np.exp(-0.5 * (n_edge - i) ** 2 / s ** 2)
        w[i] *= f
        w[-i - 1] *= f
        w[:, i] *= f
        w[:, -i - 1] *= f
    return w


def paint_plane(painter, delta, tile_relative_size, n_pixel_tile, z, min_tile_overlap=0.5, falloff=0.05,
                sigma=0.5, regularise_std=None, batch_size=64, seed=None, first_tile_id=0):
    """Paint a periodic mass plane tile by tile and blend (the inner loop of ``process_SLICS``,
    process_SLICS.py:198-220): tiles are cut with wrap-around, resampled to the network's tile size
    if necessary, painted in batches, weighted by ``make_weight_map`` and accumulated.

    Painters with a ``paint_stream`` (CVAEPainter) paint all tiles of the plane through the pipelined device path;
    tile (j, k) of the plane draws its prior noise from Philox under the key ``seed`` and the counter
    ``first_tile_id + j * n_side + k``.  ``seed=None`` (default) draws a FRESH key from torch's global generator for
    every call: like the reference (fresh ``torch.randn`` per tile, cvae.py:64) two planes never share their latent
    noise unless asked to, and ``torch.manual_seed`` makes a whole light cone reproducible.  Pass an explicit ``seed``
    (+ distinct ``first_tile_id`` ranges) to reproduce one plane.  Painters / transforms the device pipeline has no
    form for (modes other than 'shift-log', no transform, several label fields, L != 1, no prior network) go through
    ``paint_batch`` as before."""
    n_plane = int(n_pixel_tile / tile_relative_size)
    origins, slices = generate_tiling(n_plane, n_pixel_tile, min_tile_overlap)
    tiles = []
    for xs in origins:
        for ys in origins:
            t = get_tile(delta, (xs, ys), tile_relative_size)
            if t.shape[0] != n_pixel_tile:
                import scipy.ndimage
                t = scipy.ndimage.zoom(t, zoom=n_pixel_tile / t.shape[0], mode="reflect")
            tiles.append(np.asarray(t, dtype=np.float32))
    painted = None
    # eligibility is decided UP FRONT (no capture attempted, no random number consumed): an error raised later, deep
    # inside the device pipeline, then propagates instead of silently selecting the slow host path
    if hasattr(painter, "paint_stream") and getattr(painter, "can_paint_stream", lambda z=0.0: True)(z):
        if seed is None:
            import torch
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        ids = first_tile_id + np.arange(len(tiles), dtype=np.int64)
        painted = painter.paint_stream(np.stack(tiles), z, batch_size=min(batch_size, len(tiles)), tile_ids=ids,
                                       seed=seed)
    if painted is None:
        painted = painter.paint_batch(np.stack(tiles), z, batch_size=batch_size)
    plane = np.zeros((n_plane, n_plane))
    weight = np.zeros((n_plane, n_plane))
    it = iter(painted)
    for j in range(len(origins)):
        for k in range(len(origins)):
            p = next(it)
            w = make_weight_map(p.shape, falloff=falloff, sigma=sigma)
            if regularise_std is not None:
                w[np.abs(p - p.mean()) > p.std() * regularise_std] = 0
            plane[slices[j][k]] += w * p
            weight[slices[j][k]] += w
    with np.errstate(invalid="ignore"):          # 0 / 0 where no tile reaches, as in the reference
        return plane / weight
