"""Philox4x32-10 + Box-Muller in NumPy (TEST INFRASTRUCTURE, see oracle/__init__.py): the checker of
``bp_philox_normal`` (csrc/paint.hip), the per-tile prior noise of the paint pipeline.

Algorithm: Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3" (SC'11), Philox-4x32 with 10
rounds, multipliers 0xD2511F53 / 0xCD9E8D57, Weyl key increments 0x9E3779B9 / 0xBB67AE85.  Known-answer vectors of
the Random123 distribution (kat_vectors) are checked in tests/test_host_golden.py.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(counter, key):
    """counter: (..., 4) uint32, key: (k0, k1) -> (..., 4) uint32."""
    c = [np.asarray(counter[..., i], dtype=np.uint64) for i in range(4)]
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        h0, l0, h1, l1 = p0 >> np.uint64(32), p0 & MASK, p1 >> np.uint64(32), p1 & MASK
        c = [h1 ^ c[1] ^ np.uint64(k0), l1, h0 ^ c[3] ^ np.uint64(k1), l0]
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return np.stack(c, axis=-1).astype(np.uint32)


def tile_normals(seed, tile_ids, per_tile, L=1):
    """eps of shape (L, n, per_tile): element i = 4 g + r is the r-th Box-Muller normal of Philox block
    counter = (g, l, tile_id low, tile_id high), key = (seed low, seed high)."""
    tile_ids = np.asarray(tile_ids, dtype=np.uint64)
    n, groups = len(tile_ids), (per_tile + 3) // 4
    ctr = np.zeros((L, n, groups, 4), dtype=np.uint32)
    ctr[..., 0] = np.arange(groups, dtype=np.uint32)[None, None, :]
    ctr[..., 1] = np.arange(L, dtype=np.uint32)[:, None, None]
    ctr[..., 2] = (tile_ids & np.uint64(0xFFFFFFFF)).astype(np.uint32)[None, :, None]
    ctr[..., 3] = (tile_ids >> np.uint64(32)).astype(np.uint32)[None, :, None]
    x = philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)).astype(np.float64)
    out = np.empty((L, n, groups, 4))
    for h in range(2):
        u1, u2 = (x[..., 2 * h] + 1.0) / 4294967296.0, x[..., 2 * h + 1] / 4294967296.0
        rad = np.sqrt(-2.0 * np.log(u1))
        out[..., 2 * h] = rad * np.cos(2 * np.pi * u2)
        out[..., 2 * h + 1] = rad * np.sin(2 * np.pi * u2)
    return out.reshape(L, n, groups * 4)[..., :per_tile].astype(np.float32)
