"""Observation builder with the reference's method surface, computed by the CPU checker.

TEST INFRASTRUCTURE ONLY: lets tests drive oracle/flmarl_oracle.c through the same rollout
driver as the reference (golden generation) and the product's drop-in classes.
Method-by-method citations are in flmarl_oracle.c.
"""
import ctypes as C
from collections import namedtuple

import numpy as np

from . import oracle as orc

NextAction = namedtuple("NextAction", ["action", "next_position", "next_direction"])
WalkingElement = namedtuple("WalkingElement", ["position", "direction", "next_action_element"])


def rows_to_path(rows):
    if rows is None:
        return None
    return [WalkingElement((int(r[0]), int(r[1])), int(r[2]), NextAction(int(r[3]), (int(r[4]), int(r[5])), int(r[6])))
            for r in rows]


class OracleRailObs:
    def __init__(self, depth):
        self.max_time_steps = depth
        self.env = None

    def set_env(self, env):
        self.env = env

    def reset(self):
        env = self.env
        self.grid = np.ascontiguousarray(env.rail.grid, np.uint16)
        self.g = orc.Graph(self.grid)
        self.num_rails = self.g.R
        self.num_agents = len(env.agents)
        self.tpc = np.array([orc.tpc_of(a.speed_data["speed"]) for a in env.agents], np.int32)
        self.recompute_bitmap = True
        self._dist = {}
        self.paths = None

    # ---- helpers
    def _dist_for(self, a):
        t = tuple(self.env.agents[a].target)
        if t not in self._dist:
            self._dist[t] = orc.distance_map(self.grid, *t)
        return self._dist[t]

    def _virtual(self, agent):
        if agent.status == 0:
            return agent.initial_position
        if agent.status == 1:
            return agent.position
        return agent.target

    def _bitmap(self, a, rows, pos):
        agent = self.env.agents[a]
        cells = orc.prediction_cells(rows, pos[0], pos[1], agent.target[0], agent.target[1], self.tpc[a],
                                     self.max_time_steps)
        bm, err = orc.bitmap_from_cells(self.g, cells, agent.target[0], agent.target[1], self.max_time_steps + 1)
        if err:
            raise {"IndexError": IndexError, "KeyError": KeyError, "AssertionError": AssertionError}[err](err)
        return bm

    def _state(self):
        ag = self.env.agents
        W = self.env.width
        status = np.array([int(x.status) for x in ag], np.int32)
        pos = np.array([(-1 if x.position is None else x.position[0] * W + x.position[1]) for x in ag], np.int32)
        init = np.array([x.initial_position[0] * W + x.initial_position[1] for x in ag], np.int32)
        return status, pos, init

    # ---- reference surface
    def get_many(self, handles=None):
        if not self.recompute_bitmap:
            return None
        self.recompute_bitmap = False
        T1 = self.max_time_steps + 1
        maps = np.zeros((self.num_agents, self.num_rails, T1), np.int8)
        self.paths = []
        rows_all = []
        for a, agent in enumerate(self.env.agents):
            pos = self._virtual(agent)
            rows = None if agent.status == 3 else orc.shortest_path(
                self.grid, self._dist_for(a), pos[0], pos[1], agent.direction, agent.target[0], agent.target[1], T1)
            rows_all.append(rows)
            self.paths.append(rows_to_path(rows))
        for a, agent in enumerate(self.env.agents):
            maps[a] = orc.shift_right(self._bitmap(a, rows_all[a], self._virtual(agent)))
        return maps

    def get_altmaps(self, handle):
        agent = self.env.agents[handle]
        pos = self._virtual(agent)
        paths, err = orc.altpaths(self.grid, self._dist_for(handle), self.g.node_id, pos[0], pos[1], agent.direction,
                                  agent.target[0], agent.target[1])
        if err:
            raise AssertionError(err)
        out = []
        for rows in paths:
            bm = self._bitmap(handle, rows, pos)
            if agent.status == 0:
                bm[:, -1] = 0
                orc.shift_right(bm)
            out.append(bm)
        return out, [rows_to_path(r) for r in paths]

    def get_agent_action(self, handle):
        agent = self.env.agents[handle]
        action = 0
        if agent.status == 0:
            action = 2
        elif agent.status == 1:
            if self.paths[handle] is None or len(self.paths[handle]) == 0:
                action = 4
            else:
                nxt = self.paths[handle][0].next_action_element.action
                if nxt in (1, 2, 3):
                    action = nxt
                self.paths[handle] = self.paths[handle][1:]
        return action

    def _rail_of(self, cell):
        H, W = self.grid.shape
        if cell is None or not (0 <= cell[0] < H and 0 <= cell[1] < W):
            return -1
        return int(self.g.cell_rail[cell[0] * W + cell[1]])

    def is_before_switch(self, a):
        agent = self.env.agents[a]
        if agent.status != 1:
            return False
        if len(self.paths[a]) > 0:
            nxt = self.paths[a][0].next_action_element.next_position
            return self._rail_of(agent.position) != -1 and self._rail_of(nxt) == -1
        return True

    def check_crash(self, a, maps, is_before_switch=False):
        status, pos, init = self._state()
        m = np.ascontiguousarray(maps, np.int8)
        A, R, T1 = m.shape
        nxt = -1
        if self.paths[a] is not None and len(self.paths[a]) > 0:
            p = self.paths[a][0].next_action_element.next_position
            nxt = p[0] * self.env.width + p[1]
        vp = C.c_void_p
        rc = orc.lib().orc_check_crash(m.ctypes.data_as(vp), A, R, T1, status.ctypes.data_as(vp),
                                       self.tpc.ctypes.data_as(vp), pos.ctypes.data_as(vp), init.ctypes.data_as(vp),
                                       int(a), int(bool(is_before_switch)), int(nxt))
        return bool(rc)

    def update_bitmaps(self, a, maps, is_before_switch=False):
        status, _, _ = self._state()
        m = np.ascontiguousarray(maps, np.int8)
        A, R, T1 = m.shape
        vp = C.c_void_p
        orc.lib().orc_update_bitmaps(m.ctypes.data_as(vp), A, R, T1, status.ctypes.data_as(vp),
                                     self.tpc.ctypes.data_as(vp), int(a), int(bool(is_before_switch)))
        maps[...] = m
        return maps

    def set_agent_path(self, a, path):
        self.paths[a] = path


class OraclePreprocessor:
    def __init__(self, max_rails, reorder_rails=False):
        self.max_rails, self.reorder_rails = max_rails, reorder_rails

    def get_obs(self, handle, bitmap, maps):
        return orc.get_obs(maps, handle, bitmap, self.max_rails, self.reorder_rails)
