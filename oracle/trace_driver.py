"""Deterministic rollout driver used to record and to replay golden traces.

TEST INFRASTRUCTURE ONLY.  It plays the role the reference's training script plays
for the observation builder (the five per-agent cases of /root/reference/src/main.py:146-249
and the environment step at :252), but the stop/go network is replaced by a
hash of (step, agent) so that every implementation of the builder can be driven
through exactly the same decisions.  The builder, predictor and preprocessor are
passed in: the reference's classes when goldens are generated
(oracle/gen_golden_rollout.py), the package's drop-in classes when they are tested.
The environment is the stand-in RailEnv of oracle/flatland_shim (flatland's own
stepper is not available offline: PARITY UNPINNED for the stepping rules).
"""
import hashlib

import numpy as np

READY, ACTIVE, DONE, DONE_REMOVED = 0, 1, 2, 3
DO_NOTHING, MOVE_LEFT, MOVE_FORWARD, MOVE_RIGHT, STOP_MOVING = 0, 1, 2, 3, 4


def policy(step, a):
    """-> (network_action, selector): go 7 times out of 8."""
    x = (step * 7919 + a * 104729 + 12345) & 0x7FFFFFFF
    x = (x * 1103515245 + 12345) & 0x7FFFFFFF
    x = (x * 1103515245 + 12345) & 0x7FFFFFFF
    return (1 if ((x >> 16) % 8) != 0 else 0), (x >> 8) & 0xFFFF


def _sha(arr, dtype):
    return hashlib.sha256(np.ascontiguousarray(np.asarray(arr).astype(dtype)).tobytes()).hexdigest()


def run_trace(env, obs_builder, preprocessor, n_steps, hash_obs=True):
    """Returns a dict of arrays describing every decision and the occupancy maps after every step."""
    A = env.get_num_agents()
    maps, info = env.reset()
    maps = np.array(maps)
    rec = dict(branch=[], net=[], crash=[], action=[], best=[], n_alt=[], maps_sha=[], status=[], pos=[], malf=[])
    obs_h = hashlib.sha256()
    altmaps = [None] * A
    altpaths = [[] for _ in range(A)]
    snapshots = {}
    for step in range(n_steps):
        buffer_maps = maps.copy()
        actions = {}
        row = dict(branch=[], net=[], crash=[], action=[], best=[], n_alt=[])
        for a in range(A):
            agent = env.agents[a]
            crash, best, n_alt = False, -1, 0
            net, sel = policy(step, a)
            if agent.status in (DONE, DONE_REMOVED):
                branch = 0
                maps[a, :, :] = 0
                net, action = 0, DO_NOTHING
            elif agent.status == READY:
                branch = 1
                if hash_obs:
                    obs_h.update(np.asarray(preprocessor.get_obs(a, maps[a], buffer_maps)).astype(np.int16).tobytes())
                if net == 0:
                    action = DO_NOTHING
                else:
                    crash = bool(obs_builder.check_crash(a, maps))
                    if crash:
                        net, action = 0, STOP_MOVING
                    else:
                        maps = obs_builder.update_bitmaps(a, maps)
                        action = int(obs_builder.get_agent_action(a))
            elif obs_builder.is_before_switch(a) and info["action_required"][a]:
                branch = 2
                if len(altpaths[a]) == 0 or tuple(agent.position) != tuple(altpaths[a][0][0].position):
                    altmaps[a], altpaths[a] = obs_builder.get_altmaps(a)
                n_alt = len(altmaps[a])
                if n_alt > 0:
                    if hash_obs:
                        for i in range(n_alt):
                            o = preprocessor.get_obs(a, altmaps[a][i], buffer_maps)
                            obs_h.update(np.asarray(o).astype(np.int16).tobytes())
                    best = sel % n_alt
                    maps[a, :, :] = altmaps[a][best]
                    obs_builder.set_agent_path(a, altpaths[a][best])
                else:
                    net = 0
                if net == 0:
                    action = STOP_MOVING
                else:
                    crash = bool(obs_builder.check_crash(a, maps, is_before_switch=True))
                    if crash:
                        net, action = 0, STOP_MOVING
                    else:
                        action = int(obs_builder.get_agent_action(a))
                        maps = obs_builder.update_bitmaps(a, maps, is_before_switch=True)
            elif info["action_required"][a]:
                branch = 3
                crash = bool(obs_builder.check_crash(a, maps))
                if crash:
                    net, action = 0, STOP_MOVING
                else:
                    net = 1
                    action = int(obs_builder.get_agent_action(a))
                    maps = obs_builder.update_bitmaps(a, maps)
            else:
                branch = 4
                net, action = 1, DO_NOTHING
                maps = obs_builder.update_bitmaps(a, maps)
            actions[a] = action
            for k, v in (("branch", branch), ("net", net), ("crash", int(crash)), ("action", action), ("best", best),
                         ("n_alt", n_alt)):
                row[k].append(v)
        _, reward, done, info = env.step(actions)
        for k in row:
            rec[k].append(row[k])
        rec["maps_sha"].append(_sha(maps, np.int8))
        rec["status"].append([int(ag.status) for ag in env.agents])
        rec["pos"].append([(-1 if ag.position is None else ag.position[0] * env.width + ag.position[1])
                           for ag in env.agents])
        rec["malf"].append([int(info["malfunction"][a]) for a in range(A)])
        if step % 10 == 9 or step == n_steps - 1:
            snapshots[step] = maps.astype(np.int8).copy()
        if done["__all__"]:
            break
    out = {k: np.array(v, dtype=np.int32) for k, v in rec.items() if k != "maps_sha"}
    out["maps_sha"] = np.array(rec["maps_sha"])
    out["obs_sha"] = np.array(obs_h.hexdigest())
    out["n_steps_run"] = np.array(len(rec["maps_sha"]), dtype=np.int32)
    return out, snapshots
