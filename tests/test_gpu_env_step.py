"""The device environment step against the stand-in stepper the golden traces were recorded with
(oracle/flatland_shim, test infrastructure), under random actions: every action value, invalid turns,
stops and restarts, blocked cells, arrivals, with and without the malfunction process.  flatland's own stepper is not available: parity unpinned
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


@pytest.mark.parametrize("name,A,seed,malf", [
    ("Test_0_Level_0", 50, 1, None), ("Test_2_Level_1", 40, 2, None), ("Test_4_Level_0", 60, 3, None),
    ("Test_7_Level_1", 30, 4, None),
    ("Test_0_Level_0", 50, 5, dict(rate=10, min_duration=0, max_duration=0, seed=77)),
    ("Test_2_Level_1", 40, 6, dict(rate=30, min_duration=2, max_duration=9, seed=78)),
    ("Test_4_Level_0", 60, 7, dict(rate=2000, min_duration=0, max_duration=0, seed=79)),
    ("Test_7_Level_1", 30, 8, dict(rate=4, min_duration=0, max_duration=3, seed=80))])
def test_env_step_matches_stand_in(name, A, seed, malf):
    import torch
    import flatland_marl_amd as fm
    from flatland_marl_amd.rollout import RolloutBatch
    if SHIM not in sys.path:
        sys.path.insert(0, SHIM)
    from flatland.envs.rail_env import RailEnv
    fx = load_env(name)
    A = min(A, len(fx["handle"]))
    env = RailEnv.from_fixture(fx, _NoObs(), num_agents=A, malfunction=malf)
    env.reset()
    E = 2
    rep = lambda x: np.repeat(np.asarray(x)[None, :A], E, axis=0)
    b = fm.ObservationBatch(np.repeat(fx["grid"][None], E, axis=0), rep(fx["initial_position"]), rep(fx["initial_direction"]),
                            rep(fx["target"]), rep(fx["speed"]), 20)
    ro = RolloutBatch(b, rep(fx["speed"]), rep(fx["initial_direction"]),
                      malfunction=dict(malf, env_seeds=[malf["seed"]] * E) if malf else None)
    n_broken = 0
    rng = np.random.default_rng(seed)
    W = env.width
    for t in range(250):
        # mostly "go", with every other action mixed in
        act = np.where(rng.random(A) < 0.6, 2, rng.integers(0, 5, A)).astype(np.int32)
        ro.action.copy_(torch.as_tensor(np.repeat(act[None], E, axis=0), device=ro.device))
        ro.env_step()
        ro.t += 1
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
            if malf:
                assert np.array_equal(ro.malfunction[e].cpu().numpy(), [int(info["malfunction"][i]) for i in range(A)]), t
        n_broken += sum(a.malfunction_data["nr_malfunctions"] for a in env.agents) if t == 249 or done["__all__"] else 0
        if done["__all__"]:
            break
    if malf and malf["rate"] < 100:
        assert n_broken > 0


def test_global_reward_is_paid_once():
    """Two trains on a straight track: the step in which the last one arrives pays +1 to both, later steps 0."""
    import torch
    import flatland_marl_amd as fm
    from flatland_marl_amd.rollout import RolloutBatch
    if SHIM not in sys.path:
        sys.path.insert(0, SHIM)
    from flatland._impl import RailEnv, EnvAgent
    grid = np.full((1, 8), 0x0401, np.uint16)
    start, heading, target, speed = [(0, 1), (0, 0)], [1, 1], [(0, 4), (0, 3)], [1.0, 0.5]
    env = RailEnv(grid, [EnvAgent(start[i], heading[i], heading[i], target[i], speed[i], i) for i in range(2)], _NoObs())
    env.reset()
    b = fm.ObservationBatch(grid[None], np.array([start]), np.array([heading]), np.array([target]), np.array([speed]), 10)
    ro = RolloutBatch(b, np.array([speed]), np.array([heading]))
    ro.action.fill_(2)
    paid = []
    for t in range(14):
        ro.env_step()
        ro.t += 1
        _, reward, done, _ = env.step({0: 2, 1: 2})
        torch.cuda.synchronize()
        assert np.array_equal(ro.reward[0].cpu().numpy(), [float(reward[0]), float(reward[1])]), t
        assert np.array_equal(ro.status[0].cpu().numpy(), [int(a.status) for a in env.agents]), t
        assert bool(ro.all_done[0].item()) == bool(done["__all__"]), t
        paid.append(float(reward[0]))
    assert paid.count(1.0) == 1 and paid[-1] == 0.0
