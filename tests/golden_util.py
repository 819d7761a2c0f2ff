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
