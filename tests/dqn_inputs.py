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


TRAINED_RAILS = 7      # the small network of the trained-weights golden (its weights are stored in the repository)


def _crop(obs, rails):
    n = obs.shape[0]
    return np.ascontiguousarray(obs.reshape(n, 3, MAX_RAILS, DEPTH + 1)[:, :, :rails]).reshape(n, 3 * rails, DEPTH + 1)


def trained_observations(n, name="Test_2_Level_0"):
    """-> float32 [n, 3*TRAINED_RAILS, DEPTH+1]: observations of the first n agents, first TRAINED_RAILS rails of each plane."""
    return _crop(observations(n, name), TRAINED_RAILS)


def trained_learn_batch(B):
    obs = _crop(observations(2 * B, "Test_0_Level_1"), TRAINED_RAILS)
    a = (np.arange(B) * 7 % 3 % 2).astype(np.int64)
    r = (-(np.arange(B) % 4) * 0.25).astype(np.float32)
    d = (np.arange(B) % 5 == 0)
    return obs[:B], a, r, obs[B:], d
