"""Seeded real-shaped inputs for the DQN tests and golden generation (observations of a bundled map,
computed with the CPU checker)."""
import numpy as np

from oracle import oracle as orc
from tests.golden_util import load_env

MAX_RAILS, DEPTH = 100, 150


def observations(n, name="Test_0_Level_0"):
    """-> float32 [n, 3*MAX_RAILS, DEPTH+1] observations of the first n agents."""
    env = orc.OracleEnv(load_env(name), DEPTH)
    maps, _, errs = env.get_many()
    assert not any(errs)
    return np.stack([orc.get_obs(maps, a, maps[a], MAX_RAILS) for a in range(n)]).astype(np.float32)


def learn_batch(B):
    obs = observations(2 * B)
    s, sn = obs[:B], obs[B:]
    a = np.arange(B) % 2
    r = -(np.arange(B) % 4) * 0.25
    d = (np.arange(B) % 5 == 0)
    return s, a.astype(np.int64), r.astype(np.float32), sn, d
