"""Drop-in for the reference's ShortestPathPredictorForRailEnv (src/predictions.py:19-195).

Same constructor, methods and return values; the distance maps, the greedy shortest paths and the
alternative paths are computed on the GPU (flm_distance_bfs, flm_shortest_paths, flm_altpaths) from
the environment's transition grid.  The environment object only has to provide what the reference
reads: env.rail.grid (uint16 transition map), env.agents[*] with initial_position, position,
direction, target, status, speed_data['speed'], handle.
"""
from collections import defaultdict

import numpy as np

from .batch import ObservationBatch
from .types import RailAgentStatus, RailEnvActions, path_from_device

try:  # pragma: no cover
    from flatland.core.env_prediction_builder import PredictionBuilder as _Base
except Exception:
    class _Base:
        def __init__(self, max_depth=20):
            self.max_depth = max_depth

        def set_env(self, env):
            self.env = env

        def reset(self):
            pass


def _grid_of(env):
    rail = env.rail
    return np.ascontiguousarray(getattr(rail, "grid"), dtype=np.uint16)


def agent_arrays(env):
    """status, virtual cell (src/predictions.py:89-94 / shortest_path.py:38-47), current cell, heading."""
    W = env.width
    status, vcell, pcell, heading = [], [], [], []
    for ag in env.agents:
        st = int(ag.status)
        status.append(st)
        if st == RailAgentStatus.READY_TO_DEPART:
            v = ag.initial_position
        elif st == RailAgentStatus.ACTIVE:
            v = ag.position
        else:
            v = ag.target
        vcell.append(v[0] * W + v[1])
        pcell.append(-1 if ag.position is None else ag.position[0] * W + ag.position[1])
        heading.append(int(ag.direction))
    return (np.array(status, np.int32), np.array(vcell, np.int32), np.array(pcell, np.int32),
            np.array(heading, np.int32))


class ShortestPathPredictorForRailEnv(_Base):
    def __init__(self, max_depth: int = 20):
        self.shortest_paths = None
        self.empty_prediction = np.full((max_depth + 1, 5), np.nan)
        self.empty_prediction[:, 0] = np.arange(max_depth + 1)
        self._state = None
        super().__init__(max_depth)

    # ---- device state shared with the observation builder
    def reset(self):
        self._state = None

    def device_state(self, max_rails=None):
        """The ObservationBatch (E = 1) of the current environment, built on first use after reset()."""
        env = self.env
        if self._state is None:
            ag = env.agents
            st = ObservationBatch(_grid_of(env), [a.initial_position for a in ag], [a.direction for a in ag],
                                  [a.target for a in ag], [a.speed_data["speed"] for a in ag], self.max_depth,
                                  max_rails=max_rails)
            st.build_graph()
            st.compute_distances()
            self._state = st
        return self._state

    def _sync_agents(self):
        st = self.device_state()
        status, vcell, _, heading = agent_arrays(self.env)
        st.set_agent_state(status=status, pos_cell=vcell, direction=heading)
        return st

    # ---- reference surface
    def get(self, handle: int = None):
        """Shortest path of every agent and, per requested agent, the [(max_depth+1), 5] prediction table
        (time, row, col, heading, action)."""
        agents = self.env.agents
        if handle:  # the reference's truthiness test: handle 0 means "all agents"
            agents = [self.env.agents[handle]]
        st = self._sync_agents()
        st.shortest_paths(max_depth=self.max_depth + 1)
        pc = st.path_cell[0].cpu().numpy()
        pm = st.path_meta[0].cpu().numpy().view(np.uint32)
        pl = st.path_len[0].cpu().numpy()
        self.shortest_paths = {a.handle: path_from_device(pc[i], pm[i], int(pl[i]), st.W)
                               for i, a in enumerate(self.env.agents)}
        return {a.handle: self.prediction_from_path(a.handle, self.shortest_paths[a.handle]) for a in agents}

    def get_altpaths(self, handle, cell_to_id_node=None):
        """-> (alternative paths, their cell sequences); AssertionError as in the reference."""
        st = self._sync_agents()
        res = st.altpaths([handle], max_depth=500)
        err = int(res["err"][0].item())
        if err == 4:
            raise AssertionError("agent %d: no unique move on a plain cell, or already on its target" % handle)
        if err:
            raise OverflowError("agent %d: more alternatives than the device buffers hold" % handle)
        n = int(res["count"][0].item())
        lens = res["len"][0].cpu().numpy()
        cells = res["cell"][0].cpu().numpy()
        meta = res["meta"][0].cpu().numpy().view(np.uint32)
        altpaths = [path_from_device(cells[i], meta[i], int(lens[i]), st.W) for i in range(n)]
        self._last_alt = (res, n)
        cells_seqs = [self.cells_seq_from_prediction(handle, self.prediction_from_path(handle, p)) for p in altpaths]
        return altpaths, cells_seqs

    def prediction_from_path(self, handle, path):
        agent = self.env.agents[handle]
        st = int(agent.status)
        if st == RailAgentStatus.READY_TO_DEPART:
            pos = agent.initial_position
        elif st == RailAgentStatus.ACTIVE:
            pos = agent.position
        elif st == RailAgentStatus.DONE:
            pos = agent.target
        else:
            return self.empty_prediction
        T1 = self.max_depth + 1
        tpc = int(np.reciprocal(agent.speed_data["speed"]))
        rows = [(pos[0], pos[1], int(agent.direction), int(RailEnvActions.MOVE_FORWARD))]
        if path:
            for el in path[1:]:
                rows.append((el.position[0], el.position[1], int(el.direction), int(el.next_action_element.action)))
        tgt = tuple(agent.target)
        kstop = len(rows) - 1
        for k, rw in enumerate(rows):
            if (rw[0], rw[1]) == tgt:
                kstop = k
                break
        pred = np.zeros((T1, 5), dtype=int)
        idx = np.arange(T1)
        k = np.minimum(idx // tpc, kstop)
        tab = np.array(rows, dtype=int)
        pred[:, 0] = idx
        pred[:, 1:4] = tab[k, 0:3]
        # the action column: MOVE_FORWARD while waiting inside a cell, the path's action on the step the
        # agent enters a new cell, STOP_MOVING once it no longer advances
        moved = (idx % tpc == 0) & (idx // tpc <= kstop) & (idx > 0)
        act = np.full(T1, int(RailEnvActions.MOVE_FORWARD))
        act[moved] = tab[k[moved], 3]
        stopped = (idx > 0) & ((idx - 1) // tpc >= kstop)   # position at idx-1 was already the last one
        act[stopped] = int(RailEnvActions.STOP_MOVING)
        act[0] = int(RailEnvActions.MOVE_FORWARD)
        pred[:, 4] = act
        return pred

    def cells_seq_from_prediction(self, handle, prediction):
        return [(step[1], step[2]) for step in prediction]

    def compute_cells_sequence(self, prediction_dict):
        cells_sequence = defaultdict(list)
        for a in self.env.agents:
            cells_sequence[a.handle] = self.cells_seq_from_prediction(a.handle, prediction_dict[a.handle])
        return cells_sequence

    def get_prediction_depth(self):
        return self.max_depth

    def get_shortest_path_action(self, handle):
        agent = self.env.agents[handle]
        if agent.status == RailAgentStatus.READY_TO_DEPART:
            return RailEnvActions.MOVE_FORWARD
        if agent.status == RailAgentStatus.ACTIVE:
            if self.shortest_paths[handle] is None:
                return RailEnvActions.STOP_MOVING
            return self.shortest_paths[handle][0][2][0]
        return RailEnvActions.DO_NOTHING
