"""Minimal stand-in for the parts of flatland-rl 2.1.10 the reference's hot path touches.

TEST INFRASTRUCTURE ONLY.  flatland-rl is a third-party dependency of the
reference (pinned 2.1.10 in /root/reference/environment.yml:84) whose source is
not available offline.  This module restates, from its published behaviour, the
few functions the reference calls (call sites: rail_observations.py:355,367,384,
429,441,474,497,501; shortest_path.py:56-62,111,148; predictions.py:62) so that
the reference's own src/ modules can be imported unmodified to generate golden
vectors.  PARITY UNPINNED at this boundary: nothing in the reference's files
pins these semantics.

Grid encoding: uint16 per cell, four nibbles MSB first, one per heading of the
agent inside the cell (N,E,S,W); inside a nibble MSB..LSB = exit headings N,E,S,W.
"""
import collections
from collections import deque
from enum import IntEnum
from typing import NamedTuple, Tuple

import numpy as np


class Grid4TransitionsEnum(IntEnum):
    NORTH = 0
    EAST = 1
    SOUTH = 2
    WEST = 3


class RailEnvActions(IntEnum):
    DO_NOTHING = 0
    MOVE_LEFT = 1
    MOVE_FORWARD = 2
    MOVE_RIGHT = 3
    STOP_MOVING = 4


class RailAgentStatus(IntEnum):
    READY_TO_DEPART = 0
    ACTIVE = 1
    DONE = 2
    DONE_REMOVED = 3


RailEnvNextAction = NamedTuple(
    "RailEnvNextAction",
    [("action", RailEnvActions), ("next_position", Tuple[int, int]), ("next_direction", int)],
)

_MOVES = ((-1, 0), (0, 1), (1, 0), (0, -1))


def get_new_position(position, movement):
    return (position[0] + _MOVES[movement][0], position[1] + _MOVES[movement][1])


def direction_to_point(pos1, pos2):
    diff_vec = np.array((pos1[0] - pos2[0], pos1[1] - pos2[1]))
    axis = np.argmax(np.power(diff_vec, 2))
    direction = np.sign(diff_vec[axis])
    if axis == 0:
        return Grid4TransitionsEnum.NORTH if direction > 0 else Grid4TransitionsEnum.SOUTH
    return Grid4TransitionsEnum.WEST if direction > 0 else Grid4TransitionsEnum.EAST


class OrderedSet(collections.OrderedDict):
    def add(self, item):
        self[item] = None


class GridTransitionMap:
    def __init__(self, grid):
        self.grid = np.asarray(grid, dtype=np.uint16)
        self.height, self.width = self.grid.shape

    def get_full_transitions(self, row, col):
        return int(self.grid[row, col])

    def get_transitions(self, row, col, orientation):
        nib = (int(self.grid[row, col]) >> ((3 - orientation) * 4)) & 0xF
        return ((nib >> 3) & 1, (nib >> 2) & 1, (nib >> 1) & 1, nib & 1)

    def get_transition(self, cell_id, transition_index):
        return self.get_transitions(cell_id[0], cell_id[1], cell_id[2])[transition_index]

    def is_dead_end(self, pos):
        return bin(int(self.grid[pos[0], pos[1]])).count("1") == 1


def get_valid_move_actions_(agent_direction, agent_position, rail):
    valid_actions = OrderedSet()
    possible_transitions = rail.get_transitions(*agent_position, agent_direction)
    num_transitions = int(np.count_nonzero(possible_transitions))
    if rail.is_dead_end(agent_position):
        exit_direction = (agent_direction + 2) % 4
        if possible_transitions[exit_direction]:
            valid_actions.add(RailEnvNextAction(RailEnvActions.MOVE_FORWARD,
                                                get_new_position(agent_position, exit_direction),
                                                exit_direction))
    elif num_transitions == 1:
        for new_direction in [(agent_direction + i) % 4 for i in range(-1, 2)]:
            if possible_transitions[new_direction]:
                valid_actions.add(RailEnvNextAction(RailEnvActions.MOVE_FORWARD,
                                                    get_new_position(agent_position, new_direction),
                                                    new_direction))
    else:
        for new_direction in [(agent_direction + i) % 4 for i in range(-1, 2)]:
            if possible_transitions[new_direction]:
                if new_direction == agent_direction:
                    action = RailEnvActions.MOVE_FORWARD
                elif new_direction == (agent_direction + 1) % 4:
                    action = RailEnvActions.MOVE_RIGHT
                else:
                    action = RailEnvActions.MOVE_LEFT
                valid_actions.add(RailEnvNextAction(action,
                                                    get_new_position(agent_position, new_direction),
                                                    new_direction))
    return valid_actions


class DistanceMap:
    """float64[A,H,W,4], inf where unreachable; lazy reverse BFS per agent target."""

    def __init__(self, agents, env_height, env_width):
        self.env_height = env_height
        self.env_width = env_width
        self.distance_map = None
        self.agents = agents
        self.rail = None

    def reset(self, agents, rail):
        self.rail = rail
        self.agents = agents
        self.env_height = rail.height
        self.env_width = rail.width
        self.distance_map = None

    def get(self):
        if self.distance_map is None:
            self._compute()
        return self.distance_map

    def _compute(self):
        H, W = self.env_height, self.env_width
        self.distance_map = np.full((len(self.agents), H, W, 4), np.inf)
        cache = {}
        for i, agent in enumerate(self.agents):
            tgt = (int(agent.target[0]), int(agent.target[1]))
            if tgt not in cache:           # identical targets give identical maps
                cache[tgt] = self._walk(tgt)
            self.distance_map[i] = cache[tgt]

    def _neighbors(self, dm, position, current_distance, enforce=-1):
        out = []
        dirs = [0, 1, 2, 3] if enforce < 0 else [(enforce + 2) % 4]
        for nd in dirs:
            nc = get_new_position(position, nd)
            if 0 <= nc[0] < self.env_height and 0 <= nc[1] < self.env_width:
                want = (nd + 2) % 4
                for o in range(4):
                    if self.rail.get_transition((nc[0], nc[1], o), want):
                        nd_ = min(dm[nc[0], nc[1], o], current_distance + 1)
                        out.append((nc[0], nc[1], o, nd_))
                        dm[nc[0], nc[1], o] = nd_
        return out

    def _walk(self, position):
        dm = np.full((self.env_height, self.env_width, 4), np.inf)
        dm[position[0], position[1], :] = 0
        queue = deque(self._neighbors(dm, position, 0))
        visited = {(position[0], position[1], d) for d in range(4)}
        while queue:
            node = queue.popleft()
            nid = (node[0], node[1], node[2])
            if nid not in visited:
                visited.add(nid)
                for n in self._neighbors(dm, (node[0], node[1]), node[3], node[2]):
                    queue.append(n)
        return dm


class Environment:
    pass


class ObservationBuilder:
    def __init__(self):
        self.env = None

    def set_env(self, env):
        self.env = env

    def reset(self):
        raise NotImplementedError()

    def get_many(self, handles=None):
        return {h: self.get(h) for h in (handles or [])}

    def get(self, handle=0):
        raise NotImplementedError()


class PredictionBuilder:
    def __init__(self, max_depth=20):
        self.max_depth = max_depth

    def set_env(self, env):
        self.env = env

    def reset(self):
        pass

    def get(self, handle=0):
        raise NotImplementedError()


class EnvAgent:
    """Attribute bag with the 12 fields the bundled test-envs serialise."""

    def __init__(self, initial_position, initial_direction, direction, target, speed, handle,
                 status=RailAgentStatus.READY_TO_DEPART):
        self.initial_position = (int(initial_position[0]), int(initial_position[1]))
        self.initial_direction = int(initial_direction)
        self.direction = int(direction)
        self.target = (int(target[0]), int(target[1]))
        self.moving = False
        self.speed_data = {"position_fraction": 0.0, "speed": float(speed),
                           "transition_action_on_cellexit": 0}
        self.malfunction_data = {"malfunction": 0, "malfunction_rate": 0.0,
                                 "next_malfunction": 0, "nr_malfunctions": 0}
        self.handle = int(handle)
        self.status = RailAgentStatus(int(status))
        self.position = None
        self.old_direction = None
        self.old_position = None


def _fmix32(h):
    h &= 0xFFFFFFFF
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & 0xFFFFFFFF
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & 0xFFFFFFFF
    h ^= h >> 16
    return h


def malfunction_draws(seed, step, agent):
    """Two 32-bit draws for (environment seed, step, agent): the first decides whether the agent breaks
    (draw < threshold), the second how long.  flatland draws from the environment's MT19937 stream
    (np_random.rand(), np_random.randint()); that stream depends on every earlier draw of the
    environment and cannot be reproduced without flatland, so a counter hash stands in for it."""
    h = _fmix32(seed ^ _fmix32(step * 0x9E3779B1 + agent * 0x85EBCA77 + 1))
    return h, _fmix32(h ^ 0x27D4EB2F)


def malfunction_threshold(rate):
    """flatland's per-step break probability 1 - exp(-1/rate) as a 32-bit threshold (0 = never)."""
    if rate <= 0:
        return 0
    return min(0xFFFFFFFF, int((1.0 - np.exp(-1.0 / rate)) * 4294967296.0))


class RailEnv(Environment):
    """Holds a fixed map + agents; reset() drives the observation builder the way
    flatland's RailEnv.reset does (obs_builder.reset() then get_many(all handles)).
    step() implements a deterministic stepper used only to create rollout scenarios for golden
    traces (flatland's own stepper is unavailable).  `malfunction` = dict(rate, min_duration,
    max_duration, seed) switches on the malfunction process of flatland's malfunction_from_params
    (an agent that is not broken breaks with probability 1 - exp(-1/rate) per step for
    randint(min, max + 1) + 1 steps; a broken agent on the grid ignores its action and does not
    move; the counter goes down by one after every step and `moving` is restored when it reaches 0)."""

    def __init__(self, grid, agents, obs_builder_object=None, malfunction=None):
        self.malfunction = malfunction
        self._elapsed = 0
        self.rail = GridTransitionMap(grid)
        self.height, self.width = self.rail.height, self.rail.width
        self.agents = agents
        self.distance_map = DistanceMap(self.agents, self.height, self.width)
        self.distance_map.reset(self.agents, self.rail)
        self.obs_builder = obs_builder_object
        self.dones = {}
        if self.obs_builder is not None:
            self.obs_builder.set_env(self)

    @classmethod
    def from_fixture(cls, fx, obs_builder_object=None, num_agents=None, malfunction=None):
        n = len(fx["handle"]) if num_agents is None else num_agents
        agents = [EnvAgent(fx["initial_position"][i], fx["initial_direction"][i], fx["direction"][i],
                           fx["target"][i], fx["speed"][i], i, fx["status"][i]) for i in range(n)]
        return cls(fx["grid"], agents, obs_builder_object, malfunction)

    def get_num_agents(self):
        return len(self.agents)

    def _info(self):
        return {
            "action_required": {i: (a.status == RailAgentStatus.READY_TO_DEPART or
                                    (a.status == RailAgentStatus.ACTIVE and
                                     abs(a.speed_data["position_fraction"]) < 1e-9))
                                for i, a in enumerate(self.agents)},
            "malfunction": {i: a.malfunction_data["malfunction"] for i, a in enumerate(self.agents)},
            "speed": {i: a.speed_data["speed"] for i, a in enumerate(self.agents)},
            "status": {i: a.status for i, a in enumerate(self.agents)},
        }

    def reset(self):
        for a in self.agents:
            a.status = RailAgentStatus.READY_TO_DEPART
            a.position = None
            a.direction = a.initial_direction
            a.moving = False
            a.speed_data["position_fraction"] = 0.0
            a.speed_data["transition_action_on_cellexit"] = 0
            a.malfunction_data["malfunction"] = 0
            a.malfunction_data["nr_malfunctions"] = 0
            a.malfunction_data.pop("moving_before_malfunction", None)
        self._elapsed = 0
        self.dones = {i: False for i in range(len(self.agents))}
        self.dones["__all__"] = False
        self.distance_map.reset(self.agents, self.rail)
        self.obs_builder.reset()
        obs = self.obs_builder.get_many(list(range(len(self.agents))))
        return obs, self._info()

    def cell_free(self, pos):
        return all(a.position != pos for a in self.agents)

    def _check_action(self, agent, action):
        """(new_cell_valid, new_direction, new_position, transition_valid) for a MOVE_* action."""
        transition_valid = None
        possible = self.rail.get_transitions(*agent.position, agent.direction)
        num = int(np.count_nonzero(possible))
        new_direction = agent.direction
        if action == RailEnvActions.MOVE_LEFT:
            new_direction = agent.direction - 1
            if num <= 1:
                transition_valid = False
        elif action == RailEnvActions.MOVE_RIGHT:
            new_direction = agent.direction + 1
            if num <= 1:
                transition_valid = False
        new_direction %= 4
        if action == RailEnvActions.MOVE_FORWARD and num == 1:
            new_direction = int(np.argmax(possible))
            transition_valid = True
        new_position = get_new_position(agent.position, new_direction)
        inside = 0 <= new_position[0] < self.height and 0 <= new_position[1] < self.width
        new_cell_valid = inside and self.rail.get_full_transitions(*new_position) > 0
        if transition_valid is None:
            transition_valid = bool(possible[new_direction])
        return new_cell_valid, new_direction, new_position, transition_valid

    def _break_agent(self, i, agent):
        m = self.malfunction
        md = agent.malfunction_data
        if not m or md["malfunction"] >= 1:
            return
        h1, h2 = malfunction_draws(int(m["seed"]), self._elapsed, i)
        if h1 < malfunction_threshold(m["rate"]):
            md["malfunction"] = int(m["min_duration"]) + h2 % (int(m["max_duration"]) - int(m["min_duration"]) + 1) + 1
            md["moving_before_malfunction"] = agent.moving
            md["nr_malfunctions"] += 1

    def _fix_agent_after_malfunction(self, agent):
        md = agent.malfunction_data
        if md["malfunction"] < 1:
            return
        md["malfunction"] -= 1
        if md["malfunction"] == 0 and "moving_before_malfunction" in md:
            agent.moving = md["moving_before_malfunction"]

    def step(self, action_dict):
        rewards = {}
        already_over = self.dones.get("__all__", False)
        for i, agent in enumerate(self.agents):
            self._break_agent(i, agent)
        for i, agent in enumerate(self.agents):
            self._step_agent(i, agent, action_dict, rewards)
            self._fix_agent_after_malfunction(agent)
        self._elapsed += 1
        self.dones["__all__"] = all(self.dones[i] for i in range(len(self.agents)))
        if self.dones["__all__"] and not already_over:
            rewards = {i: 1.0 for i in range(len(self.agents))}       # flatland's global reward, paid once
        obs = self.obs_builder.get_many(list(range(len(self.agents))))
        return obs, rewards, dict(self.dones), self._info()

    def _step_agent(self, i, agent, action_dict, rewards):
        rewards[i] = 0
        if agent.status in (RailAgentStatus.DONE, RailAgentStatus.DONE_REMOVED):
            return
        action = RailEnvActions(int(action_dict.get(i, 0)))
        sd = agent.speed_data
        if agent.status == RailAgentStatus.READY_TO_DEPART:
            if action in (RailEnvActions.MOVE_LEFT, RailEnvActions.MOVE_RIGHT,
                          RailEnvActions.MOVE_FORWARD) and self.cell_free(agent.initial_position):
                agent.status = RailAgentStatus.ACTIVE
                agent.position = agent.initial_position
                agent.direction = agent.initial_direction
                agent.moving = True
            rewards[i] = -sd["speed"]
            return
        agent.old_direction, agent.old_position = agent.direction, agent.position
        if agent.malfunction_data["malfunction"] > 0:
            rewards[i] = -sd["speed"]
            return
        if abs(sd["position_fraction"]) < 1e-9:
            if action == RailEnvActions.DO_NOTHING and agent.moving:
                action = RailEnvActions.MOVE_FORWARD
            if action == RailEnvActions.STOP_MOVING and agent.moving:
                agent.moving = False
            if not agent.moving and action not in (RailEnvActions.DO_NOTHING,
                                                   RailEnvActions.STOP_MOVING):
                agent.moving = True
            if agent.moving:
                valid = False
                if action not in (RailEnvActions.DO_NOTHING, RailEnvActions.STOP_MOVING):
                    cv, nd, npos, tv = self._check_action(agent, action)
                    if cv and tv:
                        valid = True
                    elif action in (RailEnvActions.MOVE_LEFT, RailEnvActions.MOVE_RIGHT):
                        action = RailEnvActions.MOVE_FORWARD
                        cv, nd, npos, tv = self._check_action(agent, action)
                        valid = cv and tv
                if valid:
                    sd["transition_action_on_cellexit"] = int(action)
                else:
                    agent.moving = False
        if agent.moving:
            sd["position_fraction"] += sd["speed"]
            if sd["position_fraction"] > 1.0 - 1e-9:
                cv, nd, npos, tv = self._check_action(
                    agent, RailEnvActions(sd["transition_action_on_cellexit"]))
                if cv and tv and self.cell_free(npos):
                    agent.position = npos
                    agent.direction = nd
                    sd["position_fraction"] = 0.0
            if agent.position == agent.target:
                agent.status = RailAgentStatus.DONE_REMOVED
                agent.position = None
                agent.moving = False
                self.dones[i] = True
        rewards[i] = -sd["speed"] if not self.dones[i] else 0
