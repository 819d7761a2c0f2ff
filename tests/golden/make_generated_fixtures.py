"""Environment fixtures from the package's own rail generator (same layout as the converted test-envs).

    python tests/golden/make_generated_fixtures.py

The grids are committed so that the goldens recorded from the reference (oracle/gen_golden.py) stay tied to
the exact inputs even if the generator changes later.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from flatland_marl_amd.railgen import generate_env  # noqa: E402

# name -> (H, W, agents, seed, generator arguments)
CASES = {
    "Gen_25x25_0": (25, 25, 10, 1, dict(spacing=4, max_rails=70)),
    # targets on the start's own track allowed: the reference's bitmap builder raises IndexError on these
    "Gen_25x25_1": (25, 25, 16, 2, dict(spacing=3, keep=0.45, allow_same_track=True)),
    "Gen_30x45_2": (30, 45, 24, 3, dict(spacing=4, keep=0.6)),
    "Gen_64x64_3": (64, 64, 40, 4, dict(spacing=5, keep=0.5)),
}


def main():
    for name, (H, W, A, seed, kw) in CASES.items():
        env = generate_env(H, W, A, seed=seed, **kw)
        env["malfunction"] = np.array([0, 0, 0], np.int64)
        np.savez_compressed(os.path.join(HERE, "envs", name + ".npz"), **env)
        print(name, env["grid"].shape, "cells", int((env["grid"] != 0).sum()), "codes", len(np.unique(env["grid"])) - 1)


if __name__ == "__main__":
    main()
