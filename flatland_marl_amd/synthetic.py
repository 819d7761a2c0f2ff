"""Synthetic batches of environments built from stored rail grids.

Batches are made from a few real grids (uint16 transition maps) by applying the
eight symmetries of the square to the grid and by re-drawing which (start,
heading) and which target each agent gets from the grid's own station cells, with
numpy.random.default_rng(seed + env).  (railgen.py draws whole grids instead.)
"""
import numpy as np

# heading permutation of each primitive symmetry (N,E,S,W = 0..3)
_FLIP_LR = np.array([0, 3, 2, 1])      # mirror columns: E <-> W
_TRANSPOSE = np.array([3, 2, 1, 0])    # (r,c) -> (c,r): N <-> W, E <-> S


def _permute_codes(grid, perm):
    """Re-label headings in every transition code: bit (h -> n) becomes (perm[h] -> perm[n])."""
    g = grid.astype(np.uint32)
    out = np.zeros_like(g)
    for h in range(4):
        for n in range(4):
            bit = (g >> np.uint32((3 - h) * 4 + (3 - n))) & np.uint32(1)
            out |= bit << np.uint32((3 - int(perm[h])) * 4 + (3 - int(perm[n])))
    return out.astype(np.uint16)


def transform(grid, pos_rc, heading, sym):
    """Apply symmetry sym in 0..7 (bit0 = mirror columns, bit1 = transpose, bit2 = mirror rows)."""
    grid = np.asarray(grid, np.uint16)
    pos = [np.array(p, dtype=np.int64).copy() for p in pos_rc]
    heading = [np.array(h, dtype=np.int64).copy() for h in heading]
    if sym & 1:
        W = grid.shape[1]
        grid = _permute_codes(grid[:, ::-1], _FLIP_LR)
        for p in pos:
            p[..., 1] = W - 1 - p[..., 1]
        heading = [_FLIP_LR[h] for h in heading]
    if sym & 2:
        grid = _permute_codes(grid.T, _TRANSPOSE)
        pos = [p[..., ::-1].copy() for p in pos]
        heading = [_TRANSPOSE[h] for h in heading]
    if sym & 4:
        H = grid.shape[0]
        flip_ud = np.array([2, 1, 0, 3])
        grid = _permute_codes(grid[::-1], flip_ud)
        for p in pos:
            p[..., 0] = H - 1 - p[..., 0]
        heading = [flip_ud[h] for h in heading]
    return np.ascontiguousarray(grid), pos, heading


def make_batch(fixtures, E, A, seed=0, square_only=True, first=0):
    """E environments with A agents each, cycling over fixtures x symmetries.

    fixtures: list of dicts with grid, initial_position, direction, target, speed (tests/golden/envs/*.npz).
    first: global index of the batch's first environment; environment first + e of a larger job is the same
    whichever process builds it (a rank's shard equals the slice of the single-process batch).
    Returns dict(grid [E,H,W] uint16, start [E,A,2], direction [E,A], target [E,A,2], speed [E,A]).
    """
    variants = []
    for fx in fixtures:
        H, W = fx["grid"].shape
        syms = range(8) if (H == W or not square_only) else [0, 1, 4, 5]
        for s in syms:
            g, (st, tg), (hd,) = transform(fx["grid"], [fx["initial_position"], fx["target"]], [fx["direction"]], s)
            variants.append((g, st, hd, np.unique(tg, axis=0)))
    shape = variants[0][0].shape
    variants = [v for v in variants if v[0].shape == shape]
    speeds = np.array([1.0, 0.5, 1.0 / 3.0, 0.25])
    grid = np.empty((E,) + shape, np.uint16)
    start = np.empty((E, A, 2), np.int32)
    direction = np.empty((E, A), np.int32)
    target = np.empty((E, A, 2), np.int32)
    speed = np.empty((E, A), np.float64)
    for e in range(E):
        g, st, hd, tg = variants[(first + e) % len(variants)]
        rng = np.random.default_rng(seed + first + e)
        grid[e] = g
        pick = rng.choice(len(st), size=A, replace=len(st) < A)
        start[e] = st[pick]
        direction[e] = hd[pick]
        for a in range(A):
            while True:
                t = tg[rng.integers(len(tg))]
                if not (t[0] == start[e, a, 0] and t[1] == start[e, a, 1]):
                    break
            target[e, a] = t
        speed[e] = speeds[rng.integers(4, size=A)]
    return dict(grid=grid, start=start, direction=direction, target=target, speed=speed)
