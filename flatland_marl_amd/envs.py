"""Reader / writer for the environment files the reference evaluates on (test-envs/Test_N/Level_M.pkl,
loaded by flatland's rail_from_file / schedule_from_file in run.py and src/main.py).

Despite the extension the files are msgpack dictionaries {grid, agents, malfunction}: the uint16 transition
grid as nested lists, one 12-field tuple per agent (initial_position, initial_direction, direction, target,
moving, speed_data, malfunction_data, handle, status, position, old_direction, old_position) whose integers
may be msgpack-numpy scalars, and [malfunction_rate, min_duration, max_duration].  Nothing in a file is
executed.  `load_env_file` returns the arrays ObservationBatch / RolloutBatch take.
"""
import numpy as np


def _plain(o):
    """msgpack-numpy scalars / arrays -> python ints / ndarrays, recursively."""
    if isinstance(o, dict) and (b"nd" in o or "nd" in o):
        get = lambda k: o.get(k.encode(), o.get(k))
        dt = get("type")
        arr = np.frombuffer(get("data"), dtype=dt.decode() if isinstance(dt, bytes) else dt)
        if not get("nd"):
            return arr[0].item()
        return arr.reshape(get("shape"))
    if isinstance(o, (list, tuple)):
        return [_plain(x) for x in o]
    if isinstance(o, dict):
        return {(k.decode() if isinstance(k, bytes) else k): _plain(v) for k, v in o.items()}
    return o


def load_env_file(path):
    """-> dict(grid [H,W] uint16, initial_position [A,2], initial_direction [A], direction [A], target [A,2],
    speed [A] float64, handle [A], status [A], malfunction [3])."""
    import msgpack
    with open(path, "rb") as f:
        d = _plain(msgpack.unpackb(f.read(), raw=False, strict_map_key=False))
    ag = d["agents"]
    n = len(ag)
    return dict(
        grid=np.array(d["grid"], dtype=np.uint16),
        initial_position=np.array([a[0] for a in ag], dtype=np.int32).reshape(n, 2),
        initial_direction=np.array([a[1] for a in ag], dtype=np.int32),
        direction=np.array([a[2] for a in ag], dtype=np.int32),
        target=np.array([a[3] for a in ag], dtype=np.int32).reshape(n, 2),
        speed=np.array([a[5]["speed"] for a in ag], dtype=np.float64),
        handle=np.array([a[7] for a in ag], dtype=np.int32),
        status=np.array([a[8] for a in ag], dtype=np.int32),
        malfunction=np.array(d.get("malfunction", [0, 0, 0]), dtype=np.int64),
    )


def save_env_file(path, env):
    """Write the arrays of load_env_file back in the same layout (plain msgpack integers)."""
    import msgpack
    n = len(env["handle"])
    agents = []
    for i in range(n):
        agents.append([
            [int(x) for x in env["initial_position"][i]], int(env["initial_direction"][i]), int(env["direction"][i]),
            [int(x) for x in env["target"][i]], False,
            {"position_fraction": 0.0, "speed": float(env["speed"][i]), "transition_action_on_cellexit": 0},
            {"malfunction": 0, "malfunction_rate": 0.0, "next_malfunction": 0, "nr_malfunctions": 0},
            int(env["handle"][i]), int(env["status"][i]), None, None, None])
    d = {"grid": np.asarray(env["grid"]).astype(int).tolist(), "agents": agents,
         "malfunction": [int(x) for x in env.get("malfunction", [0, 0, 0])]}
    with open(path, "wb") as f:
        f.write(msgpack.packb(d, use_bin_type=True))
