"""Convert the reference's bundled evaluation maps into small .npz fixtures.

Run in the build container only (reads /root/reference/test-envs/Test_N/Level_M.pkl,
which are msgpack dictionaries, not pickles: {grid, agents, malfunction}).  The
output holds data only: the uint16 transition grid and, per agent, the integer
fields and the speed.  Nothing is executed from the input files.

    python tests/golden/make_env_fixtures.py
"""
import glob
import os
import sys

import msgpack
import numpy as np

SRC = "/root/reference/test-envs"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "envs")


def _dec(o):
    """msgpack-numpy scalars/arrays -> python ints / ndarrays."""
    if isinstance(o, dict) and b"nd" in o:
        arr = np.frombuffer(o[b"data"], dtype=o[b"type"])
        if not o[b"nd"]:
            return int(arr[0])
        return arr.reshape(o[b"shape"])
    if isinstance(o, list):
        return [_dec(x) for x in o]
    if isinstance(o, dict):
        return {k: _dec(v) for k, v in o.items()}
    return o


def convert(path):
    d = _dec(msgpack.unpackb(open(path, "rb").read(), raw=False, strict_map_key=False))
    grid = np.array(d["grid"], dtype=np.uint16)
    ag = d["agents"]
    n = len(ag)
    # tuple order: initial_position, initial_direction, direction, target, moving,
    # speed_data, malfunction_data, handle, status, position, old_direction, old_position
    out = dict(
        grid=grid,
        initial_position=np.array([a[0] for a in ag], dtype=np.int32).reshape(n, 2),
        initial_direction=np.array([a[1] for a in ag], dtype=np.int32),
        direction=np.array([a[2] for a in ag], dtype=np.int32),
        target=np.array([a[3] for a in ag], dtype=np.int32).reshape(n, 2),
        speed=np.array([a[5]["speed"] for a in ag], dtype=np.float64),
        handle=np.array([a[7] for a in ag], dtype=np.int32),
        status=np.array([a[8] for a in ag], dtype=np.int32),
        malfunction=np.array(d["malfunction"], dtype=np.int64),
    )
    for a in ag:
        assert a[9] is None and a[11] is None and a[4] is False
    return out


def main():
    os.makedirs(DST, exist_ok=True)
    for f in sorted(glob.glob(os.path.join(SRC, "Test_*", "Level_*.pkl"))):
        test = os.path.basename(os.path.dirname(f))
        level = os.path.splitext(os.path.basename(f))[0]
        name = "%s_%s.npz" % (test, level)
        np.savez_compressed(os.path.join(DST, name), **convert(f))
        print("wrote", name)


if __name__ == "__main__":
    sys.exit(main())
