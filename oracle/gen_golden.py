"""Generate golden vectors by running the reference's own modules (imported
unmodified from /root/reference/src) on the bundled evaluation maps.

TEST INFRASTRUCTURE ONLY; runs in the build container only (the reference does
not travel).  flatland-rl itself is absent, so the reference runs against the
stand-in in oracle/flatland_shim: results are pinned for the reference's own
code, PARITY UNPINNED for flatland semantics (distance map, valid moves).

    python oracle/gen_golden.py            # all maps, depth 150 (+ Test_9 at depth 300)

Outputs tests/golden/obs/<env>_d<depth>.npz and tests/golden/ref_timing.json.
"""
import glob
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "flatland_shim"))
sys.path.insert(0, "/root/reference")

from flatland.envs.rail_env import RailEnv  # noqa: E402
from flatland.envs.agent_utils import RailAgentStatus  # noqa: E402
from src.rail_observations import RailObsForRailEnv  # noqa: E402
from src.predictions import ShortestPathPredictorForRailEnv  # noqa: E402
from src.preprocessing import ObsPreprocessor  # noqa: E402

ENV_DIR = os.path.join(ROOT, "tests", "golden", "envs")
OUT_DIR = os.path.join(ROOT, "tests", "golden", "obs")


def runs_of(arr2d_stack):
    """[A,R,T] small ints -> int32 [n,5] rows (a, r, t0, t1_exclusive, value)."""
    out = []
    A = arr2d_stack.shape[0]
    for a in range(A):
        m = arr2d_stack[a]
        rows = np.nonzero(np.any(m != 0, axis=1))[0]
        for r in rows:
            row = m[r]
            t = 0
            T = row.shape[0]
            while t < T:
                if row[t] != 0:
                    t0 = t
                    v = row[t]
                    while t < T and row[t] == v:
                        t += 1
                    out.append((a, r, t0, t, int(v)))
                else:
                    t += 1
    return np.array(out, dtype=np.int32).reshape(-1, 5)


def path_rows(path):
    return [(e.position[0], e.position[1], int(e.direction), int(e.next_action_element.action),
             e.next_action_element.next_position[0], e.next_action_element.next_position[1],
             int(e.next_action_element.next_direction)) for e in path]


def sha(arr):
    return hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest()


def golden_for(name, depth, timing):
    fx = np.load(os.path.join(ENV_DIR, name + ".npz"))
    ob = RailObsForRailEnv(ShortestPathPredictorForRailEnv(depth))
    env = RailEnv.from_fixture(fx, ob)
    A = env.get_num_agents()
    out = {}
    t0 = time.perf_counter()
    ob.reset()
    t_graph = time.perf_counter() - t0
    t0 = time.perf_counter()
    env.distance_map.get()
    t_dist = time.perf_counter() - t0

    # graph tables
    N = len(ob.id_node_to_cell)
    out["node_cells"] = np.array([ob.id_node_to_cell[i] for i in range(N)], dtype=np.int32).reshape(N, 2)
    R = ob.num_rails
    out["info"] = np.array([(ob.info[e][0][0], ob.info[e][0][1], ob.info[e][1][0], ob.info[e][1][1],
                             ob.info[e][2]) for e in range(R)], dtype=np.int32).reshape(R, 5)
    off = [0]
    cells = []
    for e in range(R):
        for (pos, d) in ob.id_edge_to_cells[e]:
            cells.append((pos[0], pos[1], d))
        off.append(len(cells))
    out["edge_cell_off"] = np.array(off, dtype=np.int32)
    out["edge_cells"] = np.array(cells, dtype=np.int32).reshape(-1, 3)

    # distance map (stand-in semantics): -1 = unreachable
    dm = env.distance_map.get()
    dmi = np.where(np.isinf(dm), -1, dm).astype(np.int32)
    out["dist_sha"] = np.array(sha(dmi))
    tg = fx["target"]
    uniq, first = np.unique(tg, axis=0, return_index=True)
    if dmi[first].size <= 600000:
        out["dist_targets"] = uniq.astype(np.int32)
        out["dist_unique"] = dmi[first]

    # get_many
    ob.recompute_bitmap = True
    err = ""
    t0 = time.perf_counter()
    try:
        maps = ob.get_many(list(range(A)))
    except (KeyError, IndexError, AssertionError) as ex:
        maps = None
        err = type(ex).__name__
    t_many = time.perf_counter() - t0
    out["error"] = np.array(err)
    timing[name + "_d%d" % depth] = dict(agents=A, rails=R, graph_s=t_graph, dist_shim_s=t_dist,
                                         get_many_s=t_many)
    # shortest paths (available even when the bitmap step raised)
    sp = ob.predictor.shortest_paths
    p_off = [0]
    p_rows = []
    p_none = np.zeros(A, dtype=np.int8)
    for a in range(A):
        if sp[a] is None:
            p_none[a] = 1
        else:
            p_rows.extend(path_rows(sp[a]))
        p_off.append(len(p_rows))
    out["path_off"] = np.array(p_off, dtype=np.int32)
    out["path_rows"] = np.array(p_rows, dtype=np.int16).reshape(-1, 7)
    out["path_none"] = p_none
    if maps is None:
        return out

    out["maps_shape"] = np.array(maps.shape, dtype=np.int32)
    out["maps_runs"] = runs_of(maps)
    out["maps_sha"] = np.array(sha(maps.astype(np.int8)))

    # get_obs for every agent (hash) + heat planes of three agents
    Rp = max(100, R)
    pp = ObsPreprocessor(Rp, False)
    h = hashlib.sha256()
    t0 = time.perf_counter()
    sample_agents = sorted(set([0, A // 2, A - 1]))
    heat_runs = []
    for a in range(A):
        o = pp.get_obs(a, maps[a], maps)
        assert o.shape == (3 * Rp, depth + 1) and o.dtype == np.float64
        oi = o.astype(np.int16)
        assert np.array_equal(oi.astype(np.float64), o)
        h.update(oi.tobytes())
        if a in sample_agents:
            heat_runs.append(runs_of(oi.reshape(3, Rp, depth + 1)))
    t_obs = time.perf_counter() - t0
    timing[name + "_d%d" % depth]["get_obs_all_s"] = t_obs
    out["obs_rp"] = np.array(Rp, dtype=np.int32)
    out["obs_sha"] = np.array(h.hexdigest())
    out["obs_sample_agents"] = np.array(sample_agents, dtype=np.int32)
    for i, a in enumerate(sample_agents):
        out["obs_sample_runs_%d" % i] = heat_runs[i]   # rows: (plane, r, t0, t1, value)

    # alternative paths/bitmaps at each agent's first "before switch" cell
    cell_rail = {}
    for e in range(R - 1, -1, -1):
        for (pos, d) in ob.id_edge_to_cells[e]:
            cell_rail[pos] = e
    alt_agent, alt_k, alt_count = [], [], []
    alt_path_off, alt_path_rows = [0], []
    alt_maps = []
    n_alt_agents = min(A, 60)
    for a in range(n_alt_agents):
        path = sp[a]
        if path is None:
            continue
        agent = env.agents[a]
        k_found = -1
        for k, el in enumerate(path):
            nxt = el.next_action_element.next_position
            if el.position in cell_rail and nxt not in cell_rail and el.position != agent.target \
                    and int(el.next_action_element.action) != 4:
                k_found = k
                break
        if k_found < 0:
            continue
        el = path[k_found]
        agent.status = RailAgentStatus.ACTIVE
        agent.position = el.position
        save_dir = agent.direction
        agent.direction = el.direction
        try:
            bm, ap = ob.get_altmaps(a)
        except (KeyError, IndexError, AssertionError):
            bm, ap = None, None
        agent.status = RailAgentStatus.READY_TO_DEPART
        agent.position = None
        agent.direction = save_dir
        if bm is None:
            alt_agent.append(a); alt_k.append(k_found); alt_count.append(-1)
            continue
        alt_agent.append(a); alt_k.append(k_found); alt_count.append(len(bm))
        for i in range(len(bm)):
            alt_path_rows.extend(path_rows(ap[i]))
            alt_path_off.append(len(alt_path_rows))
            alt_maps.append(bm[i])
    out["alt_agent"] = np.array(alt_agent, dtype=np.int32)
    out["alt_k"] = np.array(alt_k, dtype=np.int32)
    out["alt_count"] = np.array(alt_count, dtype=np.int32)
    out["alt_path_off"] = np.array(alt_path_off, dtype=np.int32)
    out["alt_path_rows"] = np.array(alt_path_rows, dtype=np.int16).reshape(-1, 7)
    if alt_maps:
        out["alt_maps_runs"] = runs_of(np.stack(alt_maps))   # first column = flat alternative index
    else:
        out["alt_maps_runs"] = np.zeros((0, 5), dtype=np.int32)
    return out


def main():
    os.makedirs(OUT_DIR, exist_ok=True)
    timing = {}
    names = sorted(os.path.splitext(os.path.basename(f))[0] for f in glob.glob(os.path.join(ENV_DIR, "*.npz")))
    only = sys.argv[1:]
    for name in names:
        if only and name not in only:
            continue
        for depth in ([150, 300] if name.startswith("Test_9") else [150]):
            t0 = time.time()
            out = golden_for(name, depth, timing)
            np.savez_compressed(os.path.join(OUT_DIR, "%s_d%d.npz" % (name, depth)), **out)
            print(name, depth, "err=%r" % str(out["error"]), "%.1fs" % (time.time() - t0), flush=True)
    timing["_host"] = dict(cpus=os.cpu_count(), note="single Python thread, reference src/ unmodified, flatland stand-in")
    if not only:
        with open(os.path.join(ROOT, "tests", "golden", "ref_timing.json"), "w") as f:
            json.dump(timing, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
