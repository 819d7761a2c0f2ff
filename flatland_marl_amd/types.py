"""Value types of the path interface (same field names as the reference's src/utils/types.py:6-22 and
the flatland types it builds on).  flatland's own definitions are used when flatland is importable."""
from enum import IntEnum
from typing import NamedTuple, Tuple

try:  # pragma: no cover - depends on the host environment
    from flatland.envs.rail_env import RailEnvActions, RailEnvNextAction
    from flatland.envs.agent_utils import RailAgentStatus
except Exception:  # flatland absent: identical enumerations, defined locally
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

    RailEnvNextAction = NamedTuple("RailEnvNextAction", [("action", RailEnvActions),
                                                         ("next_position", Tuple[int, int]),
                                                         ("next_direction", int)])

WalkingElement = NamedTuple("WalkingElement", [("position", Tuple[int, int]), ("direction", int),
                                               ("next_action_element", RailEnvNextAction)])
Waypoint = NamedTuple("Waypoint", [("position", Tuple[int, int]), ("direction", int)])
CardinalNode = NamedTuple("CardinalNode", [("id_node", int), ("cardinal_point", int)])

_DR = (-1, 0, 1, 0)
_DC = (0, 1, 0, -1)


def path_from_device(cells, meta, n, W):
    """Decode one device path (cell ids + packed heading/action/next heading) into WalkingElements."""
    if n < 0:
        return None
    out = []
    for k in range(n):
        cell, m = int(cells[k]), int(meta[k])
        h, act, nh = m & 0xFF, (m >> 8) & 0xFF, (m >> 16) & 0xFF
        r, c = divmod(cell, W)
        nxt = (r, c) if act == 4 else (r + _DR[nh], c + _DC[nh])
        out.append(WalkingElement((r, c), h, RailEnvNextAction(RailEnvActions(act), nxt, nh)))
    return out


def path_to_device(path, W, cells_out, meta_out):
    """Encode a list of WalkingElements into preallocated host arrays; returns the length."""
    n = len(path)
    for k, el in enumerate(path):
        cells_out[k] = el.position[0] * W + el.position[1]
        na = el.next_action_element
        meta_out[k] = int(el.direction) | (int(na.action) << 8) | (int(na.next_direction) << 16)
    return n
