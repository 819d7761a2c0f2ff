"""Rail-grid generator for synthetic batches.

flatland's sparse_rail_generator / sparse_schedule_generator (src/main.py:44-60) are third-party and not
available, so this is a generator of its own, not a restatement: it only promises grids in flatland's
transition encoding (uint16 per cell, one nibble per heading N,E,S,W, bits = allowed new headings) built
from flatland's element types - straight, curve, simple switch, diamond crossing, single and double slip -
without dead ends, and agents whose target is reachable from their start state.

A grid is a coarse lattice of junction points joined by straight track.  Lattice edges are thinned out at
random while the network stays connected and every junction keeps at least two arms; what remains at a
junction decides its element: two arms = straight or curve, three = simple switch (the through track plus
one curve), four = diamond crossing with zero, one or two slip curves.
"""
import numpy as np

from .synthetic import transform

_OPP = (2, 3, 0, 1)
_DR = (-1, 0, 1, 0)
_DC = (0, 1, 0, -1)
SPEEDS = np.array([1.0, 0.5, 1.0 / 3.0, 0.25])       # src/main.py:55-58


def _bit(heading, new_heading):
    return 1 << ((3 - heading) * 4 + (3 - new_heading))


def _connect(p, q):
    """Track between the sides p and q of a cell, usable both ways: a train that came in through side p
    heads away from it."""
    return _bit(_OPP[p], q) | _bit(_OPP[q], p)


def _connected(n_nodes, edges):
    adj = [[] for _ in range(n_nodes)]
    for u, v in edges:
        adj[u].append(v)
        adj[v].append(u)
    seen = {0}
    todo = [0]
    while todo:
        u = todo.pop()
        for v in adj[u]:
            if v not in seen:
                seen.add(v)
                todo.append(v)
    return len(seen) == n_nodes


def generate_rail(H, W, seed=0, spacing=4, keep=0.55, max_rails=None):
    """-> uint16 [H,W] transition grid.  spacing = cells between lattice lines (>= 2), keep = fraction of the
    lattice edges one would like to keep, max_rails = upper bound on the number of junction-to-junction
    tracks (the generator thins further, then gives up with ValueError)."""
    if spacing < 2 or H < spacing + 1 or W < spacing + 1:
        raise ValueError("grid too small for the lattice spacing")
    rng = np.random.default_rng(seed)
    nr, nc = (H - 1) // spacing + 1, (W - 1) // spacing + 1
    r0, c0 = (H - 1 - (nr - 1) * spacing) // 2, (W - 1 - (nc - 1) * spacing) // 2
    node = lambda i, j: i * nc + j
    edges = [(node(i, j), node(i, j + 1)) for i in range(nr) for j in range(nc - 1)] + \
            [(node(i, j), node(i + 1, j)) for i in range(nr - 1) for j in range(nc)]
    deg = np.zeros(nr * nc, np.int64)
    for u, v in edges:
        deg[u] += 1
        deg[v] += 1

    def n_rails(es, dg):
        # every junction with two arms joins two tracks into one
        return len(es) - int((dg == 2).sum())

    order = rng.permutation(len(edges))
    alive = [True] * len(edges)
    want = int(round(keep * len(edges)))
    for pass_ in range(2):
        for k in order:
            live = [e for e, a in zip(edges, alive) if a]
            enough = len(live) <= want and (max_rails is None or n_rails(live, deg) <= max_rails)
            if enough:
                break
            u, v = edges[k]
            if not alive[k] or deg[u] < 3 or deg[v] < 3:
                continue
            alive[k] = False
            if _connected(nr * nc, [e for e, a in zip(edges, alive) if a]):
                deg[u] -= 1
                deg[v] -= 1
            else:
                alive[k] = True
        if max_rails is None:
            break
    live = [e for e, a in zip(edges, alive) if a]
    if max_rails is not None and n_rails(live, deg) > max_rails:
        raise ValueError("cannot thin the lattice to %d tracks" % max_rails)
    if int((deg >= 3).sum()) < 2:
        raise ValueError("the network has fewer than two switches")

    grid = np.zeros((H, W), np.uint16)
    arms = [set() for _ in range(nr * nc)]
    for u, v in live:
        (iu, ju), (iv, jv) = divmod(u, nc), divmod(v, nc)
        if iu == iv:                                          # horizontal track
            r = r0 + iu * spacing
            for c in range(c0 + ju * spacing + 1, c0 + jv * spacing):
                grid[r, c] = _connect(1, 3)
            arms[u].add(1)
            arms[v].add(3)
        else:
            c = c0 + ju * spacing
            for r in range(r0 + iu * spacing + 1, r0 + iv * spacing):
                grid[r, c] = _connect(0, 2)
            arms[u].add(2)
            arms[v].add(0)
    for n, s in enumerate(arms):
        i, j = divmod(n, nc)
        s = sorted(s)
        if len(s) == 2:
            code = _connect(s[0], s[1])
        elif len(s) == 3:
            through = [p for p in s if _OPP[p] in s]
            side = [p for p in s if _OPP[p] not in s][0]
            code = _connect(through[0], through[1]) | _connect(side, through[int(rng.integers(2))])
        else:
            code = _connect(0, 2) | _connect(1, 3)
            kind = int(rng.integers(3))                       # diamond, single slip, double slip
            if kind:
                p = int(rng.integers(4))
                q = (p + 1) % 4
                code |= _connect(p, q)
                if kind == 2:
                    code |= _connect(_OPP[p], _OPP[q])
        grid[r0 + i * spacing, c0 + j * spacing] = code
    return grid


class _StateGraph:
    """Directed graph over the (cell, heading) states of a grid."""

    def __init__(self, grid):
        from scipy.sparse import csr_matrix
        H, W = grid.shape
        self.shape = (H, W)
        src, dst = [], []
        rr, cc = np.nonzero(grid)
        for h in range(4):
            nib = (grid[rr, cc].astype(np.int64) >> ((3 - h) * 4)) & 15
            for n in range(4):
                ok = (nib & (8 >> n)) != 0
                r2, c2 = rr[ok] + _DR[n], cc[ok] + _DC[n]
                inside = (r2 >= 0) & (r2 < H) & (c2 >= 0) & (c2 < W)
                r1, c1, r2, c2 = rr[ok][inside], cc[ok][inside], r2[inside], c2[inside]
                on_rail = grid[r2, c2] != 0
                src.append(((r1 * W + c1) * 4 + h)[on_rail])
                dst.append(((r2 * W + c2) * 4 + n)[on_rail])
        src, dst = np.concatenate(src), np.concatenate(dst)
        self.csr = csr_matrix((np.ones(len(src), np.int8), (src, dst)), shape=(H * W * 4, H * W * 4))
        # tracks between switches: connected runs of cells that are neither a diamond crossing nor offer a choice
        # (the node rule of src/rail_observations.py:428-450)
        from scipy.sparse.csgraph import connected_components
        g = grid.astype(np.int64)
        choice = np.zeros((H, W), bool)
        for h in range(4):
            nib = (g >> ((3 - h) * 4)) & 15
            choice |= (nib & (nib - 1)) != 0
        plain = (g != 0) & ~choice & (g != 0x8421)
        a, b = src // 4, dst // 4
        both = plain.ravel()[a] & plain.ravel()[b]
        cells = csr_matrix((np.ones(int(both.sum()), np.int8), (a[both], b[both])), shape=(H * W, H * W))
        label = connected_components(cells, directed=False)[1].reshape(H, W)
        self.track = np.where(plain, label, -1)
        self.grid = g

    def cells_to_first_switch(self, cell, heading, limit):
        """Cells a train moves from (cell, heading) before it stands on a switch or crossing (limit if more)."""
        H, W = self.shape
        r, c, h = int(cell[0]), int(cell[1]), int(heading)
        for n in range(limit):
            if self.track[r, c] < 0:
                return n
            nib = (int(self.grid[r, c]) >> ((3 - h) * 4)) & 15
            if nib == 0:
                return limit
            h = (8, 4, 2, 1).index(nib)
            r, c = r + _DR[h], c + _DC[h]
            if not (0 <= r < H and 0 <= c < W):
                return limit
        return limit

    def reach(self, cell, heading):
        from scipy.sparse.csgraph import breadth_first_order
        H, W = self.shape
        order = breadth_first_order(self.csr, (int(cell[0]) * W + int(cell[1])) * 4 + int(heading), directed=True,
                                    return_predecessors=False)
        out = np.zeros(H * W, bool)
        out[order // 4] = True
        return out.reshape(H, W)


def reachable_cells(grid, cell, heading):
    """Boolean [H,W]: cells a train at `cell` with `heading` can get to (the start cell included)."""
    return _StateGraph(np.asarray(grid)).reach(cell, heading)


def stations(grid):
    """(cell, heading) of every plain straight cell, both ways along the track."""
    out = []
    ns, ew = _connect(0, 2), _connect(1, 3)
    for r, c in zip(*np.nonzero((grid == ns) | (grid == ew))):
        for h in ((0, 2) if grid[r, c] == ns else (1, 3)):
            out.append(((int(r), int(c)), h))
    return out


def draw_agents(grid, A, rng, graph=None, allow_same_track=False, depth=150):
    """A agents (start cell, heading, target cell, speed): starts on distinct straight cells while they last,
    targets among the straight cells the start state reaches, speeds uniform over 1, 1/2, 1/3, 1/4.  A target
    on the start's own track (no switch in between) is not drawn unless allow_same_track: the reference's
    bitmap builder walks off the end of such a path (src/rail_observations.py:348-350, IndexError).  For the
    same reason a start must see its first switch within the prediction depth at the slowest speed."""
    st = stations(grid)
    if not st:
        raise ValueError("no straight track to start from")
    graph = graph or _StateGraph(grid)
    is_station = np.zeros(grid.shape, bool)
    for (r, c), _ in st:
        is_station[r, c] = True
    start = np.empty((A, 2), np.int32)
    heading = np.empty((A,), np.int32)
    target = np.empty((A, 2), np.int32)
    order = rng.permutation(len(st))
    k = 0
    for a in range(A):
        for _ in range(4 * len(st)):
            (cell, h) = st[order[k % len(st)]]
            k += 1
            if not allow_same_track and graph.cells_to_first_switch(cell, h, depth) > depth // 4 - 2:
                continue
            reach = graph.reach(cell, h) & is_station
            reach[cell[0], cell[1]] = False
            if not allow_same_track:
                reach &= graph.track != graph.track[cell[0], cell[1]]
            cand = np.argwhere(reach)
            if len(cand):
                break
        else:
            raise ValueError("no start with a reachable target")
        start[a], heading[a] = cell, h
        target[a] = cand[rng.integers(len(cand))]
    return start, heading, target, SPEEDS[rng.integers(4, size=A)]


def draw_agents_scc(grid, A, rng, graph, depth=150):
    """draw_agents for large batches: instead of one breadth-first search per agent, the strongly connected
    components of the state graph are computed once per grid, and a target is drawn among the straight cells that
    have a state in the start state's component (every state of a component reaches every other one, so the
    target is reachable; targets outside the component are never drawn).  Same exclusions as draw_agents
    without allow_same_track."""
    from scipy.sparse.csgraph import connected_components
    H, W = grid.shape
    cache = getattr(graph, "_scc", None)
    if cache is None:
        st = stations(grid)
        comp = connected_components(graph.csr, directed=True, connection="strong")[1]
        is_station = np.zeros(grid.shape, bool)
        for (r, c), _ in st:
            is_station[r, c] = True
        ok_start = [i for i, (cell, h) in enumerate(st) if graph.cells_to_first_switch(cell, h, depth) <= depth // 4 - 2]
        cache = graph._scc = dict(st=st, comp=comp.reshape(H, W, 4), is_station=is_station, ok_start=ok_start, cand={})
    st, comp, cand = cache["st"], cache["comp"], cache["cand"]
    if not cache["ok_start"]:
        raise ValueError("no start that sees a switch within the prediction depth")
    start = np.empty((A, 2), np.int32)
    heading = np.empty((A,), np.int32)
    target = np.empty((A, 2), np.int32)
    order = rng.permutation(len(cache["ok_start"]))
    k = 0
    for a in range(A):
        for _ in range(4 * len(order)):
            cell, h = st[cache["ok_start"][order[k % len(order)]]]
            k += 1
            label = int(comp[cell[0], cell[1], h])
            if label not in cand:
                cand[label] = (comp == label).any(axis=2) & cache["is_station"]
            ok = cand[label] & (graph.track != graph.track[cell[0], cell[1]])
            ok[cell[0], cell[1]] = False
            cells = np.argwhere(ok)
            if len(cells):
                break
        else:
            raise ValueError("no start with a reachable target")
        start[a], heading[a] = cell, h
        target[a] = cells[rng.integers(len(cells))]
    return start, heading, target, SPEEDS[rng.integers(4, size=A)]


def generate_env(H, W, A, seed=0, allow_same_track=False, depth=150, **rail_kw):
    """One environment in the layout of the stored fixtures: dict(grid, initial_position, initial_direction,
    direction, target, speed, handle, status)."""
    rng = np.random.default_rng(seed)
    for attempt in range(20):
        try:
            grid = generate_rail(H, W, seed=int(rng.integers(1 << 31)), **rail_kw)
            break
        except ValueError:
            if attempt == 19:
                raise
    start, heading, target, speed = draw_agents(grid, A, rng, allow_same_track=allow_same_track, depth=depth)
    return dict(grid=grid, initial_position=start, initial_direction=heading.copy(), direction=heading, target=target,
                speed=speed, handle=np.arange(A, dtype=np.int32), status=np.zeros(A, np.int32))


def make_generated_batch(E, A, H, W, seed=0, n_maps=8, depth=150, symmetries=True, scc_targets=False, **rail_kw):
    """E environments of A agents on generated H x W grids: n_maps distinct grids, each used under the
    symmetries of the grid shape (symmetries=False: as generated only), agents re-drawn per environment.
    scc_targets draws the agents with draw_agents_scc (large batches).  Same layout as synthetic.make_batch."""
    maps = [generate_env(H, W, 1, seed=seed * 1000 + m, **rail_kw)["grid"] for m in range(n_maps)]
    syms = [0] if not symmetries else (range(8) if H == W else [0, 1, 4, 5])
    variants = [transform(g, [], [], s)[0] for g in maps for s in syms]
    graphs = [_StateGraph(g) for g in variants]
    out = dict(grid=np.empty((E, H, W), np.uint16), start=np.empty((E, A, 2), np.int32),
               direction=np.empty((E, A), np.int32), target=np.empty((E, A, 2), np.int32),
               speed=np.empty((E, A), np.float64))
    for e in range(E):
        g = variants[e % len(variants)]
        rng = np.random.default_rng(seed + 7919 * (e + 1))
        out["grid"][e] = g
        if scc_targets:
            out["start"][e], out["direction"][e], out["target"][e], out["speed"][e] = \
                draw_agents_scc(g, A, rng, graphs[e % len(variants)], depth=depth)
        else:
            out["start"][e], out["direction"][e], out["target"][e], out["speed"][e] = \
                draw_agents(g, A, rng, graphs[e % len(variants)], depth=depth)
    return out
