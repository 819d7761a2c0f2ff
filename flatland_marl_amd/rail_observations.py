"""Drop-in for the reference's RailObsForRailEnv (src/rail_observations.py:18-520).

Same constructor, methods, attributes and error behaviour; the switch graph, the occupancy planes, the
alternative planes and the rollout bookkeeping run on the GPU through the C ABI (include/flmarl.h).
`maps` stays a host ndarray owned by the caller, exactly as in the reference: the bookkeeping methods
copy it to the device, update it there and write the result back into the caller's array.
"""
import numpy as np
import torch

from ._lib import AGENT_ERRORS
from .predictions import agent_arrays
from .types import CardinalNode, RailAgentStatus, RailEnvActions

try:  # pragma: no cover
    from flatland.core.env_observation_builder import ObservationBuilder as _Base
except Exception:
    class _Base:
        def __init__(self):
            self.env = None

        def set_env(self, env):
            self.env = env


class RailObsForRailEnv(_Base):
    def __init__(self, predictor):
        super(RailObsForRailEnv, self).__init__()
        self.predictor = predictor
        self.num_agents = None
        self.num_rails = None
        self.max_time_steps = self.predictor.max_depth
        self.cell_to_id_node = {}
        self.id_node_to_cell = {}
        self.info = {}
        self.id_edge_to_cells = {}
        self.nodes = set()
        self.edges = set()
        self.recompute_bitmap = True
        self.max_rails = None  # optional padding of the device planes (set before reset())

    def set_env(self, env):
        super().set_env(env)
        if self.predictor:
            self.predictor.set_env(self.env)

    # ---- graph ---------------------------------------------------------------------------------------
    def reset(self):
        self.predictor.reset()
        self._map_to_graph()
        self.recompute_bitmap = True
        self.num_agents = len(self.env.agents)
        self.tpc = dict()
        for a in range(self.num_agents):
            self.tpc[a] = int(np.reciprocal(self.env.agents[a].speed_data['speed']))

    def _map_to_graph(self):
        st = self.predictor.device_state(max_rails=self.max_rails)
        W = st.W
        n, r, nc = int(st.num_nodes[0]), int(st.R[0]), None
        node_cell = st.node_cell[0, :n].cpu().numpy()
        info = st.edge_info[0, :r].cpu().numpy()
        off = st.edge_cell_off[0, :r + 1].cpu().numpy()
        cells = st.edge_cells[0, :int(off[-1]) if r else 0].cpu().numpy()
        self.id_node_to_cell = {i: (int(c) // W, int(c) % W) for i, c in enumerate(node_cell)}
        self.cell_to_id_node = {v: k for k, v in self.id_node_to_cell.items()}
        self.info = {e: (CardinalNode(int(info[e, 0]), int(info[e, 1])), CardinalNode(int(info[e, 2]), int(info[e, 3])),
                         int(info[e, 4])) for e in range(r)}
        self.id_edge_to_cells = {
            e: [(((int(x) >> 2) // W, (int(x) >> 2) % W), int(x) & 3) for x in cells[off[e]:off[e + 1]]]
            for e in range(r)}
        self.nodes = self.id_node_to_cell.keys()
        self.edges = self.info.keys()
        self.num_rails = r
        self._cell_rail = st.cell_rail[0].cpu().numpy()

    def _get_edge_from_cell(self, cell):
        H, W = self.env.height, self.env.width
        if not (0 <= cell[0] < H and 0 <= cell[1] < W):
            return -1, -1
        rail = int(self._cell_rail[cell[0] * W + cell[1]])
        if rail < 0:
            return -1, -1
        return rail, [c[0] for c in self.id_edge_to_cells[rail]].index((int(cell[0]), int(cell[1])))

    # ---- observations --------------------------------------------------------------------------------
    def _raise_agent_error(self, err):
        bad = np.nonzero(err)[0]
        if bad.size:
            raise AGENT_ERRORS[int(err[bad[0]])]("agent %d: the predicted path cannot be mapped on the rail graph"
                                                 % int(bad[0]))

    def get_many(self, handles=None):
        maps = None
        if self.recompute_bitmap:
            self.recompute_bitmap = False
            self.predictor.get()
            self.paths = self.predictor.shortest_paths
            st = self.predictor.device_state()
            st.bitmaps(shift=1)
            self._raise_agent_error(st.agent_err[0].cpu().numpy())
            maps = st.maps_numpy(0).astype(int)
        return maps

    def get_altmaps(self, handle):
        agent = self.env.agents[handle]
        altpaths, _ = self.predictor.get_altpaths(handle, self.cell_to_id_node)
        st = self.predictor.device_state()
        res, n = self.predictor._last_alt
        if n == 0:
            return [], altpaths
        Lcap = res["Lcap"]
        path_cell = res["cell"][0, :n].contiguous()
        path_len = res["len"][0, :n].contiguous()
        agent_flat = torch.full((n,), int(handle), dtype=torch.int32, device=st.device)
        shift = 1 if agent.status == RailAgentStatus.READY_TO_DEPART else 0
        planes, err = st.planes_for_paths(path_cell, path_len, agent_flat, Lcap, shift)
        self._raise_agent_error(err.cpu().numpy())
        maps = [st.plane_numpy(planes[i], self.num_rails).astype(int) for i in range(n)]
        return maps, altpaths

    # ---- rollout bookkeeping -------------------------------------------------------------------------
    def get_agent_action(self, handle):
        agent = self.env.agents[handle]
        action = RailEnvActions.DO_NOTHING
        if agent.status == RailAgentStatus.READY_TO_DEPART:
            action = RailEnvActions.MOVE_FORWARD
        elif agent.status == RailAgentStatus.ACTIVE:
            if self.paths[handle] is None or len(self.paths[handle]) == 0:
                print('[WARN] AGENT {} RAIL DISRUPTED'.format(handle))
                action = RailEnvActions.STOP_MOVING
            else:
                nxt = self.paths[handle][0].next_action_element.action
                if nxt == 1:
                    action = RailEnvActions.MOVE_LEFT
                elif nxt == 2:
                    action = RailEnvActions.MOVE_FORWARD
                elif nxt == 3:
                    action = RailEnvActions.MOVE_RIGHT
                self.paths[handle] = self.paths[handle][1:]
        return action

    def is_before_switch(self, a):
        agent = self.env.agents[a]
        before_switch = False
        if agent.status == RailAgentStatus.ACTIVE:
            if len(self.paths[a]) > 0:
                curr_rail, _ = self._get_edge_from_cell(agent.position)
                next_rail, _ = self._get_edge_from_cell(self.paths[a][0].next_action_element.next_position)
                before_switch = curr_rail != -1 and next_rail == -1
            else:
                print('[WARN] agent\'s {} path run out'.format(a))
                before_switch = True
        return before_switch

    def _device_maps(self, maps):
        st = self.predictor.device_state()
        status, _, pcell, heading = agent_arrays(self.env)
        st.set_agent_state(status=status, pos_cell=pcell, direction=heading)
        return st, st.upload_maps(maps)

    def check_crash(self, a, maps, is_before_switch=False):
        st, dmaps = self._device_maps(maps)
        nxt = -1
        if self.paths[a] is not None and len(self.paths[a]) > 0:
            p = self.paths[a][0].next_action_element.next_position
            nxt = p[0] * st.W + p[1]
        crash = st.check_crash(dmaps, [a], [1 if is_before_switch else 0], [nxt])
        return bool(crash[0].item())

    def update_bitmaps(self, a, maps, is_before_switch=False):
        st, dmaps = self._device_maps(maps)
        st.update_bitmaps(dmaps, [a], [1 if is_before_switch else 0])
        maps[a] = st.plane_numpy(dmaps[0, a], maps.shape[1])
        return maps

    def set_agent_path(self, a, path):
        self.paths[a] = path

    @staticmethod
    def _reverse_dir(direction):
        return int((direction + 2) % 4)
