"""The device environment step against the stand-in stepper the golden traces were recorded with
(oracle/flatland_shim, test infrastructure), under random actions: every action value, invalid turns,
stops and restarts, blocked cells, arrivals.  flatland's own stepper is not available: parity unpinned
beyond this agreement."""
import os
import sys

import numpy as np
import pytest

from tests.golden_util import load_env

pytestmark = pytest.mark.gpu
SHIM = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "flatland_shim")


class _NoObs:
    def set_env(self, env):
        self.env = env

    def reset(self):
        pass

    def get_many(self, handles=None):
        return None


@pytest.mark.parametrize("name,A,seed", [("Test_0_Level_0", 50, 1), ("Test_2_Level_1", 40, 2), ("Test_4_Level_0", 60, 3),
                                          ("Test_7_Level_1", 30, 4)])
def test_env_step_matches_stand_in(name, A, seed):
    import torch
    import flatland_marl_amd as fm
    from flatland_marl_amd.rollout import RolloutBatch
    if SHIM not in sys.path:
        sys.path.insert(0, SHIM)
    from flatland.envs.rail_env import RailEnv
    fx = load_env(name)
    A = min(A, len(fx["handle"]))
    env = RailEnv.from_fixture(fx, _NoObs(), num_agents=A)
    env.reset()
    E = 2
    rep = lambda x: np.repeat(np.asarray(x)[None, :A], E, axis=0)
    b = fm.ObservationBatch(np.repeat(fx["grid"][None], E, axis=0), rep(fx["initial_position"]), rep(fx["initial_direction"]),
                            rep(fx["target"]), rep(fx["speed"]), 20)
    ro = RolloutBatch(b, rep(fx["speed"]), rep(fx["initial_direction"]))
    rng = np.random.default_rng(seed)
    W = env.width
    for t in range(250):
        # mostly "go", with every other action mixed in
        act = np.where(rng.random(A) < 0.6, 2, rng.integers(0, 5, A)).astype(np.int32)
        ro.action.copy_(torch.as_tensor(np.repeat(act[None], E, axis=0), device=ro.device))
        ro.env_step()
        _, reward, done, info = env.step({i: int(act[i]) for i in range(A)})
        torch.cuda.synchronize()
        for e in range(E):
            assert np.array_equal(ro.status[e].cpu().numpy(), [int(a.status) for a in env.agents]), t
            assert np.array_equal(ro.pos[e].cpu().numpy(),
                                  [-1 if a.position is None else a.position[0] * W + a.position[1] for a in env.agents]), t
            live = np.array([a.position is not None for a in env.agents])
            assert np.array_equal(ro.heading[e].cpu().numpy()[live], np.array([a.direction for a in env.agents])[live]), t
            assert np.array_equal(ro.moving[e].cpu().numpy().astype(bool), [bool(a.moving) for a in env.agents]), t
            assert np.array_equal(ro.fraction[e].cpu().numpy(), [a.speed_data["position_fraction"] for a in env.agents]), t
            assert np.array_equal(ro.reward[e].cpu().numpy(), [float(reward[i]) for i in range(A)]), t
            assert np.array_equal(ro.done[e].cpu().numpy().astype(bool), [bool(done[i]) for i in range(A)]), t
            assert np.array_equal(ro.action_required[e].cpu().numpy().astype(bool),
                                  [bool(info["action_required"][i]) for i in range(A)]), t
            assert bool(ro.all_done[e].item()) == bool(done["__all__"]), t
        if done["__all__"]:
            break
