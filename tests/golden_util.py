"""Helpers to read the committed fixtures (tests/golden)."""
import glob
import hashlib
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def env_names():
    return sorted(os.path.splitext(os.path.basename(f))[0] for f in glob.glob(os.path.join(GOLDEN, "envs", "*.npz")))


def load_env(name):
    return dict(np.load(os.path.join(GOLDEN, "envs", name + ".npz")))


def load_golden(name, depth=150):
    return dict(np.load(os.path.join(GOLDEN, "obs", "%s_d%d.npz" % (name, depth))))


def golden_cases():
    out = []
    for n in env_names():
        out.append((n, 150))
        if n.startswith("Test_9"):
            out.append((n, 300))
    return out


def dense_from_runs(runs, shape, dtype=np.int8):
    out = np.zeros(shape, dtype)
    for a, r, t0, t1, v in runs:
        out[a, r, t0:t1] = v
    return out


def sha(arr):
    return hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest()


def path_rows_of(g, a):
    return g["path_rows"][g["path_off"][a]:g["path_off"][a + 1]].astype(np.int32)


def docs_kat():
    """The reference's own text dumps (docs/map.txt, docs/cell_seq.txt, docs/z-bitmaps-0.txt) as parsed by
    oracle/gen_golden_docs.py, plus the per-rail end table the dumps do not print, rebuilt from the map:

    each rail's cells form a chain whose two end cells each touch exactly one switch cell; the cardinal point
    stored for an end is the side of that switch facing the rail, and the end reached first by the graph
    builder's loop "switch ids ascending, cardinal points 0..3" is the source (src/rail_observations.py:463-491).
    Rails without cells (adjacent switches) never show in a bitmap; their rows stay [-1,-1,-1,-1,0].
    -> dict(node_id [H*W], cell_rail [H*W], edge_info [R,5], cell_seq [4,T1,2], bitmaps [4,R,T1], clean_agents, H, W, R, T1)
    """
    k = dict(np.load(os.path.join(GOLDEN, "docs_kat.npz")))
    node_id, cell_rail = k["node_id"], k["cell_rail"]
    H, W = node_id.shape
    R = k["bitmaps"].shape[1]
    dr, dc = (-1, 0, 1, 0), (0, 1, 0, -1)
    edge_info = np.tile(np.array([-1, -1, -1, -1, 0], np.int32), (R, 1))
    for rail in np.unique(cell_rail[cell_rail >= 0]):
        cells = {(int(r), int(c)) for r, c in np.argwhere(cell_rail == rail)}
        ends = []
        for (r, c) in sorted(cells):
            inside = sum(((r + dr[d], c + dc[d]) in cells) for d in range(4))
            if inside > 1:
                continue
            for d in range(4):
                rr, cc = r + dr[d], c + dc[d]
                if 0 <= rr < H and 0 <= cc < W and node_id[rr, cc] >= 0:
                    ends.append((int(node_id[rr, cc]), (d + 2) % 4))
        assert len(ends) == 2, (rail, ends)
        src, dst = sorted(ends)
        edge_info[rail] = (src[0], src[1], dst[0], dst[1], len(cells))
    # the builder numbers rails in the order it meets their source ends
    with_cells = [r for r in range(R) if edge_info[r, 0] >= 0]
    keys = [(edge_info[r, 0], edge_info[r, 1]) for r in with_cells]
    assert keys == sorted(keys)
    k.update(node_id=node_id.reshape(-1).astype(np.int32), cell_rail=cell_rail.reshape(-1).astype(np.int32),
             edge_info=edge_info, H=H, W=W, R=R, T1=k["cell_seq"].shape[1])
    return k
