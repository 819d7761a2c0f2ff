/*
 * CPU restatement of the flatland-marl observation hot path, one environment at a time.
 *
 * TEST INFRASTRUCTURE ONLY: this file is the checker for the HIP kernels and the
 * CPU leg of the benchmark.  Nothing in the product package may call it.
 *
 * Every function cites the reference lines it restates (paths are relative to
 * /root/reference).  The few flatland-rl 2.1.10 functions the path relies on
 * (third-party, source not available) are restated from their published
 * behaviour and marked [flatland]; PARITY UNPINNED for those, pinned for the
 * rest by golden vectors produced by the reference's own modules
 * (oracle/gen_golden.py -> tests/golden/obs).
 *
 * Conventions: headings N,E,S,W = 0..3; a cell is r*W+c; transition code is
 * four nibbles, MSB nibble = heading N, and inside a nibble MSB = exit N.
 * Occupancy values are int8 in {-1,0,+1}; distances are int32 with -1 = unreachable.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_ERR_CAPACITY -1
#define ORC_ERR_INDEX -2     /* reference raises IndexError  (rail_observations.py:350) */
#define ORC_ERR_KEY -3       /* reference raises KeyError    (rail_observations.py:353,365,382) */
#define ORC_ERR_ASSERT -4    /* reference raises AssertionError (rail_observations.py:395, shortest_path.py:108,125) */
#define ORC_ERR_RECURSION -5 /* reference would recurse without bound */

static const int DR[4] = {-1, 0, 1, 0};
static const int DC[4] = {0, 1, 0, -1};

static inline int nibble(uint16_t code, int h) { return (code >> ((3 - h) * 4)) & 0xF; }
static inline int has_exit(int nib, int n) { return (nib >> (3 - n)) & 1; }
static inline int popc4(int nib) { return (nib & 1) + ((nib >> 1) & 1) + ((nib >> 2) & 1) + ((nib >> 3) & 1); }
static inline int popc16(uint16_t v) { int n = 0; while (v) { n += v & 1; v >>= 1; } return n; }

/* [flatland] grid4_utils.direction_to_point: heading from p1 towards p2; row axis wins ties, equal cells give South. */
static int direction_to_point(int r1, int c1, int r2, int c2) {
  int dr = r1 - r2, dc = c1 - c2;
  if (dr * dr >= dc * dc) return dr > 0 ? 0 : 2;
  return dc > 0 ? 3 : 1;
}

/* [flatland] rail_env_shortest_paths.get_valid_move_actions_: candidates in the order left, forward, right.
 * Returns the count; act[i] in {1,2,3}, ndir[i] the new heading (next cell = cell + move(ndir)). */
static int valid_moves(const uint16_t* grid, int W, int r, int c, int h, int* act, int* ndir) {
  uint16_t code = grid[r * W + c];
  int nib = nibble(code, h);
  int n = 0;
  if (popc16(code) == 1) { /* dead end: only the U-turn, reported as MOVE_FORWARD */
    int ex = (h + 2) & 3;
    if (has_exit(nib, ex)) { act[n] = 2; ndir[n] = ex; n++; }
  } else if (popc4(nib) == 1) {
    for (int i = -1; i <= 1; i++) {
      int nd = (h + i + 4) & 3;
      if (has_exit(nib, nd)) { act[n] = 2; ndir[n] = nd; n++; }
    }
  } else {
    for (int i = -1; i <= 1; i++) {
      int nd = (h + i + 4) & 3;
      if (has_exit(nib, nd)) { act[n] = 2 + i; ndir[n] = nd; n++; }
    }
  }
  return n;
}

/* ------------------------------------------------------------------------------------------
 * rail_observations.py:410-510  _map_to_graph
 * node_id[H*W] (-1 = not a node), node_cell[capN], edge_info[capR*5] = src node, src cp, dst node,
 * dst cp, length; edge_cell_off[capR+1]; edge_cells[capC] = cell*4 + heading, per edge in walk order;
 * cell_rail[H*W] = lowest edge id whose cell list holds the cell (what _get_edge_from_cell,
 * rail_observations.py:311-321, returns), -1 if none.  counts = {nodes, rails, edge cells}.
 * Deviations where the reference does not terminate or indexes outside the grid: a walk whose first
 * step leaves the grid, or that exceeds 4*H*W steps, yields no edge.
 */
int orc_graph_build(const uint16_t* grid, int H, int W, int32_t* node_id, int32_t* node_cell, int capN,
                    int32_t* edge_info, int capR, int32_t* edge_cell_off, int32_t* edge_cells, int capC,
                    int32_t* cell_rail, int32_t* counts) {
  int nn = 0;
  uint8_t* conn = (uint8_t*)malloc((size_t)H * W);
  for (int i = 0; i < H * W; i++) { node_id[i] = -1; cell_rail[i] = -1; }
  for (int i = 0; i < H; i++)
    for (int j = 0; j < W; j++) { /* :420-456 row-major numbering */
      uint16_t code = grid[i * W + j];
      int is_node = 0, mask = 0;
      if (code == 0x8421) { is_node = 1; mask = 0xF; } /* :429-433 diamond crossing */
      else
        for (int d = 0; d < 4; d++) { /* :440-448 */
          int nib = nibble(code, d);
          if (nib) mask |= 1 << ((d + 2) & 3);
          for (int t = 0; t < 4; t++) if (has_exit(nib, t)) mask |= 1 << t;
          if (popc4(nib) > 1) is_node = 1;
        }
      if (is_node) {
        if (nn >= capN) { free(conn); return ORC_ERR_CAPACITY; }
        node_id[i * W + j] = nn; node_cell[nn] = i * W + j; conn[nn] = (uint8_t)mask; nn++;
      }
    }
  uint8_t* visited = (uint8_t*)calloc((size_t)nn * 4 + 4, 1);
  int ne = 0, nc = 0;
  long cap_steps = 4L * H * W;
  edge_cell_off[0] = 0;
  for (int n = 0; n < nn; n++)
    for (int cp = 0; cp < 4; cp++) { /* :463-464 */
      if (!((conn[n] >> cp) & 1)) continue; /* :465 */
      visited[n * 4 + cp] = 1; /* :466 */
      int dir = cp, pr = node_cell[n] / W, pc = node_cell[n] % W;
      int start = nc, len = 0;
      long steps = 0;
      for (;;) {
        int nr = pr + DR[dir], ncc = pc + DC[dir]; /* :474 */
        if (nr < 0 || nr >= H || ncc < 0 || ncc >= W) break; /* leaves the grid: no edge */
        if (++steps > cap_steps) break;
        int id2 = node_id[nr * W + ncc];
        if (id2 >= 0) { /* :476-492 */
          int cp2 = (dir + 2) & 3;
          if (!visited[id2 * 4 + cp2]) {
            if (ne >= capR) { free(conn); free(visited); return ORC_ERR_CAPACITY; }
            int32_t* e = edge_info + ne * 5;
            e[0] = n; e[1] = cp; e[2] = id2; e[3] = cp2; e[4] = len;
            nc = start + len; /* keep the cells of this edge */
            edge_cell_off[ne + 1] = nc;
            ne++;
            visited[id2 * 4 + cp2] = 1;
            start = nc;
          }
          break;
        }
        if (nc - start != len) abort();
        if (nc >= capC) { free(conn); free(visited); return ORC_ERR_CAPACITY; }
        edge_cells[nc++] = (nr * W + ncc) * 4 + dir; /* :475 */
        len++; /* :493 */
        pr = nr; pc = ncc;
        int nib = nibble(grid[pr * W + pc], dir) & ~(1 << (3 - ((dir + 2) & 3))); /* :496-498 */
        int t = 0; /* np.argmax: first set exit, 0 when none (:500) */
        for (int k = 0; k < 4; k++) if (has_exit(nib, k)) { t = k; break; }
        int tr = pr + DR[t], tc = pc + DC[t];
        if (tr >= 0 && tr < H && tc >= 0 && tc < W) dir = t; /* :502-504 */
        else break; /* :505-506 */
      }
      nc = start; /* drop cells of a walk that made no edge */
    }
  for (int e = ne - 1; e >= 0; e--)
    for (int k = edge_cell_off[e]; k < edge_cell_off[e + 1]; k++) cell_rail[edge_cells[k] >> 2] = e;
  counts[0] = nn; counts[1] = ne; counts[2] = edge_cell_off[ne];
  free(conn); free(visited);
  return ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * [flatland] DistanceMap.get() for one target: reverse unit-cost BFS over (cell, heading) states.
 * Called by the reference at shortest_path.py:60-62.  dist[H*W*4], -1 = unreachable. */
void orc_distance_map(const uint16_t* grid, int H, int W, int tr, int tc, int32_t* dist) {
  int S = H * W * 4;
  int32_t* queue = (int32_t*)malloc(sizeof(int32_t) * (size_t)S);
  int head = 0, tail = 0;
  for (int i = 0; i < S; i++) dist[i] = -1;
  for (int h = 0; h < 4; h++) { dist[(tr * W + tc) * 4 + h] = 0; queue[tail++] = (tr * W + tc) * 4 + h; }
  while (head < tail) {
    int s = queue[head++];
    int cell = s >> 2, h = s & 3, r = cell / W, c = cell % W;
    int pr = r - DR[h], pc = c - DC[h]; /* the cell an agent heading h came from */
    if (pr < 0 || pr >= H || pc < 0 || pc >= W) continue;
    uint16_t code = grid[pr * W + pc];
    for (int o = 0; o < 4; o++)
      if (has_exit(nibble(code, o), h)) {
        int p = (pr * W + pc) * 4 + o;
        if (dist[p] < 0) { dist[p] = dist[s] + 1; queue[tail++] = p; }
      }
  }
  free(queue);
}

/* ------------------------------------------------------------------------------------------
 * shortest_path.py:37-82  _shortest_path_for_agent (start cell/heading chosen by the caller, :38-49).
 * rows[cap*7] = r, c, heading, action, next r, next c, next heading.  Returns the element count,
 * -1 when the reference yields None (:73-75), ORC_ERR_CAPACITY - 10 on overflow. */
int orc_shortest_path(const uint16_t* grid, int H, int W, const int32_t* dist, int r, int c, int h,
                      int tr, int tc, int max_depth, int32_t* rows, int cap) {
  int n = 0, depth = 0;
  long long distance = -1; /* -1 = inf, carried across steps (:52) */
  while (!(r == tr && c == tc) && depth < max_depth) { /* :55 */
    int act[3], nd[3];
    int k = valid_moves(grid, W, r, c, h, act, nd);
    int best = -1;
    for (int i = 0; i < k; i++) { /* :59-65 strict improvement, first wins ties */
      int nr = r + DR[nd[i]], nc = c + DC[nd[i]];
      if (nr < 0 || nr >= H || nc < 0 || nc >= W) continue; /* malformed map: treated as unreachable */
      int32_t d = dist[(nr * W + nc) * 4 + nd[i]];
      if (d >= 0 && (distance < 0 || d < distance)) { best = i; distance = d; }
    }
    if (best < 0) return -1; /* :73-75 */
    if (n >= cap) return ORC_ERR_CAPACITY - 10;
    int32_t* e = rows + n * 7;
    e[0] = r; e[1] = c; e[2] = h; e[3] = act[best];
    e[4] = r + DR[nd[best]]; e[5] = c + DC[nd[best]]; e[6] = nd[best];
    n++; depth++;
    r = e[4]; c = e[5]; h = e[6];
  }
  if (depth < max_depth) { /* :79-82 final STOP_MOVING element */
    if (n >= cap) return ORC_ERR_CAPACITY - 10;
    int32_t* e = rows + n * 7;
    e[0] = r; e[1] = c; e[2] = h; e[3] = 4; e[4] = r; e[5] = c; e[6] = h;
    n++;
  }
  return n;
}

/* ------------------------------------------------------------------------------------------
 * shortest_path.py:92-163  get_altpaths (agent not DONE_REMOVED; start chosen by the caller, :95-104).
 * paths[cap_alt][cap_len*7], lens[cap_alt].  Returns the number of alternatives or a negative error. */
typedef struct {
  const uint16_t* grid; int H, W; const int32_t* dist; const int32_t* node_id;
  int tr, tc, max_depth;
  int32_t* paths; int32_t* lens; int cap_alt, cap_len, n_alt;
  int32_t prefix[64 * 7]; int n_prefix;
} alt_ctx;

static int alt_from(alt_ctx* x, int r, int c, int h) { /* :110-146 */
  int act[3], nd[3];
  int k = valid_moves(x->grid, x->W, r, c, h, act, nd);
  if (x->n_prefix >= 63) return ORC_ERR_RECURSION;
  if (x->node_id[r * x->W + c] >= 0) { /* :114-123 */
    for (int i = 0; i < k; i++) {
      int32_t* e = x->prefix + x->n_prefix * 7;
      e[0] = r; e[1] = c; e[2] = h; e[3] = act[i]; e[4] = r + DR[nd[i]]; e[5] = c + DC[nd[i]]; e[6] = nd[i];
      x->n_prefix++;
      int rc = alt_from(x, e[4], e[5], e[6]);
      x->n_prefix--;
      if (rc < 0) return rc;
    }
    return 0;
  }
  if (k != 1) return ORC_ERR_ASSERT; /* :125 */
  int32_t* e = x->prefix + x->n_prefix * 7;
  e[0] = r; e[1] = c; e[2] = h; e[3] = act[0]; e[4] = r + DR[nd[0]]; e[5] = c + DC[nd[0]]; e[6] = nd[0];
  int np = x->n_prefix + 1;
  if (x->n_alt >= x->cap_alt) return ORC_ERR_CAPACITY;
  int32_t* out = x->paths + (size_t)x->n_alt * x->cap_len * 7;
  if (np > x->cap_len) return ORC_ERR_CAPACITY;
  memcpy(out, x->prefix, sizeof(int32_t) * 7 * (size_t)np);
  if (r == x->tr && c == x->tc) { /* :133-134 */
    x->lens[x->n_alt++] = np;
    return 0;
  }
  int n = orc_shortest_path(x->grid, x->H, x->W, x->dist, e[4], e[5], e[6], x->tr, x->tc, x->max_depth,
                            out + np * 7, x->cap_len - np); /* :136-141 */
  if (n < -1) return ORC_ERR_CAPACITY;
  if (n >= 0) x->lens[x->n_alt++] = np + n; /* :143-144; None is dropped */
  return 0;
}

int orc_altpaths(const uint16_t* grid, int H, int W, const int32_t* dist, const int32_t* node_id, int r, int c,
                 int h, int tr, int tc, int max_depth, int32_t* paths, int32_t* lens, int cap_alt, int cap_len) {
  if (r == tr && c == tc) return ORC_ERR_ASSERT; /* :108 */
  alt_ctx x;
  x.grid = grid; x.H = H; x.W = W; x.dist = dist; x.node_id = node_id; x.tr = tr; x.tc = tc;
  x.max_depth = max_depth; x.paths = paths; x.lens = lens; x.cap_alt = cap_alt; x.cap_len = cap_len; x.n_alt = 0;
  int act[3], nd[3];
  int k = valid_moves(grid, W, r, c, h, act, nd); /* :148-149 */
  for (int i = 0; i < k; i++) { /* :152-161; the first action is always recorded as MOVE_FORWARD */
    int32_t* e = x.prefix;
    e[0] = r; e[1] = c; e[2] = h; e[3] = 2; e[4] = r + DR[nd[i]]; e[5] = c + DC[nd[i]]; e[6] = nd[i];
    x.n_prefix = 1;
    int rc = alt_from(&x, e[4], e[5], e[6]);
    if (rc < 0) return rc;
  }
  return x.n_alt;
}

/* ------------------------------------------------------------------------------------------
 * predictions.py:85-143  prediction_from_path + cells_seq_from_prediction, for an agent that is not
 * DONE_REMOVED.  rows/n = the path (n < 0 for None); (sr,sc) = the agent's virtual position (:89-94).
 * cells[(depth+1)*2] = (row, col) per time step. */
void orc_prediction_cells(const int32_t* rows, int n, int sr, int sc, int tr, int tc, int tpc, int depth,
                          int32_t* cells) {
  int pr = sr, pc = sc;
  int k = 1; /* :106-107 the first element is dropped */
  cells[0] = pr; cells[1] = pc; /* :102 */
  for (int idx = 1; idx <= depth; idx++) {
    int exhausted = (n < 0) || (k >= n);
    if (!((pr == tr && pc == tc) || exhausted)) { /* :117-121 */
      if (idx % tpc == 0) { pr = rows[k * 7 + 0]; pc = rows[k * 7 + 1]; k++; } /* :123-129 */
    }
    cells[idx * 2] = pr; cells[idx * 2 + 1] = pc;
  }
}

/* ------------------------------------------------------------------------------------------
 * rail_observations.py:323-407  _bitmap_from_cells_seq.  bitmap[R*T1] int8 is overwritten.
 * Returns ORC_OK or the error standing for the exception the reference raises. */
int orc_bitmap_from_cells(const int32_t* cells, int ncells, int tr, int tc, const int32_t* node_id,
                          const int32_t* cell_rail, const int32_t* edge_info, int W, int R, int T1, int8_t* bitmap) {
  memset(bitmap, 0, (size_t)R * T1);
  int n = ncells;
  for (int i = 0; i < ncells; i++) /* :332-335 cut after the first visit of the target */
    if (cells[i * 2] == tr && cells[i * 2 + 1] == tc) { n = i + 1; break; }
#define RAIL(i) cell_rail[cells[(i)*2] * W + cells[(i)*2 + 1]]
#define NODE(i) node_id[cells[(i)*2] * W + cells[(i)*2 + 1]]
  int entry_node = -1, entry_cp = -1; /* agent_entry_node; -1 = None */
  int i = 0;
  int rail = RAIL(0);
  if (rail != -1) { /* :345-362 */
    int initial = rail;
    while (rail != -1) {
      i++;
      if (i >= n) return ORC_ERR_INDEX; /* :350 */
      rail = RAIL(i);
    }
    int nid = NODE(i);
    if (nid < 0) return ORC_ERR_KEY; /* :353 */
    int cp = (direction_to_point(cells[(i - 1) * 2], cells[(i - 1) * 2 + 1], cells[i * 2], cells[i * 2 + 1]) + 2) & 3;
    const int32_t* e = edge_info + initial * 5;
    if (nid == e[2] && cp == e[3]) { entry_node = e[0]; entry_cp = e[1]; }      /* reached dst: entered at src */
    else if (nid == e[0] && cp == e[1]) { entry_node = e[2]; entry_cp = e[3]; } /* reached src: entered at dst */
  } else { /* :363-369 */
    int nid = NODE(0);
    if (nid < 0) return ORC_ERR_KEY; /* :365 */
    if (n < 2) return ORC_ERR_INDEX;  /* :367 path[1] */
    entry_node = nid;
    entry_cp = direction_to_point(cells[0], cells[1], cells[2], cells[3]);
  }
  int holes = 0;
  for (int ts = 0; ts < n; ts++) { /* :373-401 */
    rail = RAIL(ts);
    if (rail == -1) {
      holes++;
      if (ts + 1 < n && (cells[ts * 2] != cells[ts * 2 + 2] || cells[ts * 2 + 1] != cells[ts * 2 + 3])) {
        int nid = NODE(ts);
        if (nid < 0) return ORC_ERR_KEY; /* :382 */
        entry_node = nid;
        entry_cp = direction_to_point(cells[ts * 2], cells[ts * 2 + 1], cells[ts * 2 + 2], cells[ts * 2 + 3]);
      }
    } else {
      const int32_t* e = edge_info + rail * 5;
      int dir = 0;
      if (entry_node == e[2] && entry_cp == e[3]) dir = 1;        /* :390-391 */
      else if (entry_node == e[0] && entry_cp == e[1]) dir = -1;  /* :392-393 */
      if (dir == 0) return ORC_ERR_ASSERT; /* :395 */
      bitmap[rail * T1 + ts] = (int8_t)dir;
      for (int b = ts - holes; b < ts; b++) /* :399-401; numpy slice semantics for a negative start */
        if (b >= 0) bitmap[rail * T1 + b] = (int8_t)dir;
      holes = 0;
    }
  }
#undef RAIL
#undef NODE
  return ORC_OK;
}

/* rail_observations.py:92-93 (get_many) and :105-107 (get_altmaps): shift one step to the right. */
void orc_shift_right(int8_t* bitmap, int R, int T1) {
  for (int r = 0; r < R; r++) {
    memmove(bitmap + r * T1 + 1, bitmap + r * T1, (size_t)T1 - 1);
    bitmap[r * T1] = 0;
  }
}

/* ------------------------------------------------------------------------------------------
 * preprocessing.py:24-30 + 61-74  get_obs without rail reordering.
 * maps[A*R*T1] int8, bitmap[R*T1] int8 (may be an alternative), obs[3*Rp*T1] int16 overwritten. */
int orc_get_obs(const int8_t* maps, int A, int R, int T1, int handle, const int8_t* bitmap, int Rp, int16_t* obs) {
  if (R > Rp) return ORC_ERR_CAPACITY; /* numpy broadcast error at preprocessing.py:19 */
  size_t plane = (size_t)Rp * T1;
  memset(obs, 0, sizeof(int16_t) * 3 * plane);
  for (int i = 0; i < R * T1; i++) obs[i] = bitmap[i];
  for (int a = 0; a < A; a++) {
    if (a == handle) continue; /* :26 */
    const int8_t* m = maps + (size_t)a * R * T1;
    for (int i = 0; i < R * T1; i++) {
      int8_t v = m[i];
      obs[plane + i] += (v > 0) ? v : 0;      /* :27 */
      obs[2 * plane + i] += (v < 0) ? -v : 0; /* :28 */
    }
  }
  return ORC_OK;
}

/* preprocessing.py:32-59  _reorder_rails applied to the three planes of one observation
 * (planes are [Rp,T1] int16, R real rails).  Mirrors the in-place semantics for the returned state.
 * Returns ORC_ERR_INDEX where the reference loops forever (rail occupied through the last column). */
static int argmax_abs_col(const int16_t* bm, int R, int T1, int ts) {
  int best = 0, bv = -1;
  for (int r = 0; r < R; r++) { int v = bm[r * T1 + ts]; v = v < 0 ? -v : v; if (v > bv) { bv = v; best = r; } }
  return best;
}
int orc_reorder_rails(int16_t* obs, int R, int Rp, int T1) {
  int16_t* bm = obs; size_t plane = (size_t)Rp * T1;
  int* swap = (int*)malloc(sizeof(int) * (size_t)(T1 + 2));
  int ns = 0, ts = 0;
  int rail = argmax_abs_col(bm, R, T1, ts);
  if (bm[rail * T1 + ts] == 0) { ts = 1; rail = argmax_abs_col(bm, R, T1, ts); } /* :44-46 */
  while (bm[rail * T1 + ts] != 0) { /* :49-52 */
    if (ns > T1) { free(swap); return ORC_ERR_INDEX; }
    swap[ns++] = rail;
    int k = 0;
    while (ts + k < T1 && bm[rail * T1 + ts + k] != 0) k++;
    if (ts + k >= T1) k = 0; /* np.argmax of an all-False slice is 0: the reference never advances */
    if (k == 0) { free(swap); return ORC_ERR_INDEX; }
    ts += k;
    rail = argmax_abs_col(bm, R, T1, ts);
  }
  if (ns > 0) { /* :54-57, fancy-index gather then zero the rest (over the R real rows) */
    int16_t* tmp = (int16_t*)malloc(sizeof(int16_t) * (size_t)ns * T1);
    for (int p = 0; p < 3; p++) {
      int16_t* pl = obs + p * plane;
      for (int i = 0; i < ns; i++) memcpy(tmp + (size_t)i * T1, pl + (size_t)swap[i] * T1, sizeof(int16_t) * T1);
      memcpy(pl, tmp, sizeof(int16_t) * (size_t)ns * T1);
      memset(pl + (size_t)ns * T1, 0, sizeof(int16_t) * (size_t)(R - ns) * T1);
    }
    free(tmp);
  }
  free(swap);
  return ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * Rollout bookkeeping, rail_observations.py:161-299.  maps[A*R*T1] int8 is mutated in place.
 * status[A]: 0 READY_TO_DEPART, 1 ACTIVE, 2 DONE, 3 DONE_REMOVED.  tpc[A].
 */
static int first_zero(const int8_t* row, int from, int T1) { /* np.argmax(row[from:] == 0) */
  for (int t = from; t < T1; t++) if (row[t] == 0) return t - from;
  return 0;
}
/* :161-164 _get_rail_dir */
void orc_get_rail_dir(const int8_t* maps, int R, int T1, int a, int ts, int* rail, int* dir) {
  const int8_t* m = maps + (size_t)a * R * T1;
  int best = 0;
  for (int r = 0; r < R; r++) if (m[r * T1 + ts] != 0) { best = r; break; }
  *rail = best; *dir = m[best * T1 + ts];
}
/* :257-299 _last_train_on_rail */
void orc_last_train_on_rail(const int8_t* maps, int A, int R, int T1, const int32_t* status, const int32_t* tpc,
                            int a, int rail, int* last, int* last_exit) {
  int l = 0, le = 0;
  for (int o = 0; o < A; o++) {
    if (o == a || status[o] == 0) continue; /* :268 */
    const int8_t* m = maps + (size_t)o * R * T1;
    const int8_t* row = m + (size_t)rail * T1;
    int t = tpc[o];
    if (row[0] != 0) { /* :274-279 */
      int ex = first_zero(row, 0, T1);
      if (ex > le) { l = o; le = ex; }
    } else if (row[t - 1] != 0) { /* :284-297 */
      int orail, odir;
      orc_get_rail_dir(maps, R, T1, o, 0, &orail, &odir);
      int ex = 0;
      if (orail != rail) ex = first_zero(m + (size_t)orail * T1, 0, T1);
      ex += first_zero(row, ex, T1);
      if (ex > le) { l = o; le = ex; }
    }
  }
  *last = l; *last_exit = le;
}
/* :192-204 _check_headon_crash */
int orc_check_headon_crash(const int8_t* maps, int A, int R, int T1, const int32_t* status, const int32_t* tpc,
                           int a, int rail, int dir) {
  int last, le;
  orc_last_train_on_rail(maps, A, R, T1, status, tpc, a, rail, &last, &le);
  if (le > 0) return maps[((size_t)last * R + rail) * T1 + le - 1] != dir;
  return 0;
}
/* :206-230 check_crash.  pos[A] = current cell (r*W+c) or -1 when off the grid; init_pos[A];
 * next_cell = next path cell for the action_required case, or -1 when the path is empty. */
int orc_check_crash(const int8_t* maps, int A, int R, int T1, const int32_t* status, const int32_t* tpc,
                    const int32_t* pos, const int32_t* init_pos, int a, int is_before_switch, int next_cell) {
  if (status[a] == 0) { /* :210-218 */
    for (int o = 0; o < A; o++) if (o != a && pos[o] == init_pos[a]) return 1;
    int rail, dir;
    orc_get_rail_dir(maps, R, T1, a, 1, &rail, &dir);
    return orc_check_headon_crash(maps, A, R, T1, status, tpc, a, rail, dir);
  }
  if (is_before_switch) { /* :220-223 */
    int rail, dir;
    orc_get_rail_dir(maps, R, T1, a, tpc[a], &rail, &dir);
    return orc_check_headon_crash(maps, A, R, T1, status, tpc, a, rail, dir);
  }
  if (next_cell >= 0) /* :225-228 */
    for (int o = 0; o < A; o++) if (o != a && pos[o] == next_cell) return 1;
  return 0;
}
/* :167-180 _delay */
void orc_delay(int8_t* maps, int R, int T1, const int32_t* tpc, int a, int rail, int dir, int delay) {
  int8_t* m = maps + (size_t)a * R * T1;
  int t = tpc[a];
  int orail, odir;
  orc_get_rail_dir(maps, R, T1, a, 0, &orail, &odir);
  int z = delay + t; if (z > T1) z = T1;
  for (int r = 0; r < R; r++) {
    int8_t* row = m + (size_t)r * T1;
    if (delay < T1) memmove(row + delay, row, (size_t)(T1 - delay));
    memset(row, 0, (size_t)z);
  }
  for (int k = 0; k < t && k < T1; k++) m[(size_t)orail * T1 + k] = (int8_t)odir;
  for (int k = t; k < t + delay && k < T1; k++) m[(size_t)rail * T1 + k] = (int8_t)dir;
}
/* :232-252 update_bitmaps */
void orc_update_bitmaps(int8_t* maps, int A, int R, int T1, const int32_t* status, const int32_t* tpc, int a,
                        int is_before_switch) {
  int8_t* m = maps + (size_t)a * R * T1;
  if (is_before_switch) {
    int t = tpc[a], rail, dir, last, le;
    orc_get_rail_dir(maps, R, T1, a, t, &rail, &dir);
    orc_last_train_on_rail(maps, A, R, T1, status, tpc, a, rail, &last, &le);
    if (le > 0) {
      int cur = first_zero(m + (size_t)rail * T1, t, T1) + t; /* :242-244 */
      if (cur <= le) orc_delay(maps, R, T1, tpc, a, rail, dir, le + t - cur); /* :246-248 */
    }
  }
  for (int r = 0; r < R; r++) { /* :250-251 shift one step to the left, zero fill */
    int8_t* row = m + (size_t)r * T1;
    memmove(row, row + 1, (size_t)T1 - 1);
    row[T1 - 1] = 0;
  }
}
