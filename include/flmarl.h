/*
 * flmarl.h — C ABI of the MI355X (gfx950) observation hot path of flatland-marl.
 *
 * Every entry point takes plain device pointers, sizes and a HIP stream
 * (hipStream_t passed as void*), launches asynchronously on that stream and
 * returns 0 on success or a negative FLM_E_* code.  No entry point allocates
 * device memory, and none keeps per-call state: what the library remembers
 * between calls is two one-time caches per device (that a kernel's dynamic
 * LDS limit has been raised; the device address of flm_bitmaps_static's
 * workspace array), both set once under std::call_once and never changed, and
 * the diagnostic stamp buffer of flm_debug_fused_stamps when a caller sets it.
 * One "batch" is E independent environments of identical grid size H x W with
 * A agents each.
 *
 * Encodings shared by all calls
 *   heading        0..3 = N,E,S,W; a cell is r*W + c
 *   grid           uint16 transition code per cell: four nibbles, the most
 *                  significant one for an agent heading N, then E,S,W; inside a
 *                  nibble the most significant bit allows leaving towards N
 *   status         0 READY_TO_DEPART, 1 ACTIVE, 2 DONE, 3 DONE_REMOVED
 *   occupancy      int8 in {-1,0,+1}; -1 = travelling the rail from its source
 *                  node to its destination node, +1 the other way
 *   plane          one agent's [Rp, T1] int8 rows (rail-major), stored with a
 *                  stride of plane_stride bytes, plane_stride % 16 == 0 and
 *                  plane_stride >= Rp*T1; bytes past Rp*T1 are zero
 *   per-agent err  0 ok, FLM_AGENT_* when the reference would raise
 *
 * Reference = misterdev/flatland-marl; file:line below are relative to its root.
 */
#ifndef FLMARL_H
#define FLMARL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FLM_OK 0
#define FLM_E_ARG (-1)      /* bad size, null pointer, unsupported capacity */
#define FLM_E_LAUNCH (-2)   /* the HIP runtime refused a launch or a memset */
#define FLM_E_CAPACITY (-3) /* the call needs more than a fixed-size internal buffer holds */

#define FLM_AGENT_OK 0
#define FLM_AGENT_INDEX_ERROR 2   /* src/rail_observations.py:350, :367 */
#define FLM_AGENT_KEY_ERROR 3     /* src/rail_observations.py:353, :365, :382 */
#define FLM_AGENT_ASSERT_ERROR 4  /* src/rail_observations.py:395; src/utils/shortest_path.py:108, :125 */
#define FLM_AGENT_OVERFLOW 5      /* more alternatives / longer path than the caller's capacity */
#define FLM_AGENT_NONTERM 6       /* src/preprocessing.py:49-52 would loop forever on this plane */

#define FLM_GRAPH_ERR_NODES 1     /* counts[3] bits: capacity exceeded */
#define FLM_GRAPH_ERR_RAILS 2
#define FLM_GRAPH_ERR_CELLS 4
#define FLM_GRAPH_ERR_STATES 8

#define FLM_DIST_INF 0xFFFFu
#define FLM_MAX_NODES 2048        /* per environment (LDS-resident port tables) */

int flm_abi_version(void);

/* Switch graph of each grid.
 * Replaces RailObsForRailEnv._map_to_graph, src/rail_observations.py:410-510, and the cell->rail
 * lookup _get_edge_from_cell, src/rail_observations.py:311-321.
 *   grid          [E,H,W] uint16
 *   node_id       [E,H*W] int32   id of the switch/diamond in the cell, -1 otherwise (row-major numbering)
 *   node_cell     [E,capN] int32
 *   edge_info     [E,capR,5] int32  source node, source cardinal point, destination node, destination
 *                                   cardinal point, number of cells between them
 *   edge_cell_off [E,capR+1] int32, edge_cells [E,capC] int32 = cell*4 + heading, in walk order
 *   cell_rail     [E,H*W] int32   lowest rail id whose cell list holds the cell, -1 if none
 *   counts        [E,4] int32     nodes, rails, edge cells, FLM_GRAPH_ERR_* bits
 * capN <= FLM_MAX_NODES. */
int flm_graph_build(const uint16_t* grid, int E, int H, int W, int capN, int capR, int capC,
                    int32_t* node_id, int32_t* node_cell, int32_t* edge_info, int32_t* edge_cell_off,
                    int32_t* edge_cells, int32_t* cell_rail, int32_t* counts, void* stream);

/* Compact numbering of the rail cells of each grid and the reverse-move table the distance BFS walks.
 * No reference counterpart of its own: it prepares flatland's DistanceMap computation used at
 * src/utils/shortest_path.py:60-62.
 *   cell_idx [E,H*W] int32 (-1 = empty cell), n_cells [E] int32,
 *   pred     [E,capCells*4,4] uint16: for state cell_idx*4+heading, the states that reach it in one move
 *            (0xFFFF = none); NULL = numbering only (the chain-compressed distances do not need it).
 *            capCells*4 < 65535.  counts[e*4+3] gets FLM_GRAPH_ERR_STATES on overflow.
 *   idx_cell [E,capCells] int32 (may be NULL) out: the cell of each compact number (the inverse of cell_idx). */
int flm_state_index(const uint16_t* grid, int E, int H, int W, int capCells, int32_t* cell_idx,
                    int32_t* n_cells, uint16_t* pred, int32_t* counts, int32_t* idx_cell, void* stream);

/* Distance of every (cell, heading) state to each agent's target, one BFS per distinct target.
 * Replaces flatland DistanceMap.get() as called at src/utils/shortest_path.py:60-62.
 *   target_cell [E,A] int32
 *   tslot       [E,A] int32 out: row of dist that holds the agent's target
 *   n_slots     [E] int32 out, slot_target [E,A] int32 out (target cell of each used row)
 *   dist        [E,A,capCells*4] uint16 out, FLM_DIST_INF = unreachable */
int flm_distance_bfs(const uint16_t* pred, const int32_t* cell_idx, const int32_t* n_cells,
                     const int32_t* target_cell, int E, int A, int H, int W, int capCells, int32_t* tslot,
                     int32_t* n_slots, int32_t* slot_target, uint16_t* dist, void* stream);

/* The same distance maps computed by chain compression (the default; flm_distance_bfs is the fallback for
 * grids whose state tables do not fit the 160 KB LDS).  Three calls:
 *   flm_target_slots   numbers the distinct targets of every environment: tslot [E,A], n_slots [E],
 *                      slot_target [E,A] out;
 *   flm_chain_index    per environment: root [E,capCells*4] uint16 = number of the anchor state (a state
 *                      with a choice of moves, or on a target cell) each state runs into, 0xFFFF = none;
 *                      steps [E,capCells*4] uint16 = moves to get there; edges [E,capAnch,4] uint32 =
 *                      anchor | weight<<16 per move of an anchor; n_anchor [E]; sets FLM_GRAPH_ERR_STATES
 *                      in counts[e*4+3] when capAnch is too small.  Needs cell_idx/n_cells of
 *                      flm_state_index and the agents' target cells; succ [E,capCells*4] uint16 (may be NULL)
 *                      out: the state a forced move leads to (one exit in the heading's transitions, the cell
 *                      neither a dead end nor a target), 0xFFFF otherwise: the path walks below follow it
 *                      without distance look-ups; choice [E,capCells*4,4] uint16 (may be NULL) out: per state
 *                      the states its moves left, forward, right lead to (0xFFFF = no such move; the U-turn of
 *                      a dead end sits in the forward slot) and 1 when the move is reported as forward
 *                      (src/utils/shortest_path.py:56-57 through get_valid_move_actions_), so that a walk reads
 *                      one entry where it has a choice; written only for states whose succ is 0xFFFF;
 *                      walk [E,capCells*4] uint16, move_edge [E,capAnch,4] uint32, move_succ [E,capAnch,4] uint16
 *                      (all three or none; they need succ and capCells*4 <= 32768) out: the tables of
 *                      flm_paths_columns: walk = the successor of a forced state, 0x8000 | anchor number for an
 *                      anchor, 0xFFFF for a state without moves; move_edge = an anchor's edges (anchor |
 *                      weight<<16, 0xFFFFFFFF = none) in the order a walk tries its moves (left, forward, right;
 *                      the U-turn of a dead end in the forward slot), fourth entry 1 when the move is reported
 *                      as forward; move_succ = the states those moves lead to, fourth entry the same flag.
 *                      Anchors are the states with two or more moves, the states on target cells, and the
 *                      states with one move that a walk does not treat as forced;
 *                      idx_cell [E,capCells] int32 (may be NULL): the inverse of cell_idx as written by
 *                      flm_state_index; with it the kernel visits the rail cells directly instead of scanning
 *                      the H*W grid (same tables);
 *   flm_distance_chains  dist [E,A,capCells*4] as flm_distance_bfs; ldsAnch <= capAnch is the number of
 *                      anchors the workgroup's LDS tables are sized for (80 bytes each); an environment
 *                      with more anchors is flagged in counts[e*4+3] and its rows are left unwritten. */
int flm_target_slots(const int32_t* target_cell, int E, int A, int32_t* tslot, int32_t* n_slots, int32_t* slot_target,
                     void* stream);
int flm_chain_index(const uint16_t* grid, const int32_t* cell_idx, const int32_t* n_cells, const int32_t* target_cell,
                    int E, int A, int H, int W, int capCells, int capAnch, uint16_t* root, uint16_t* steps,
                    uint32_t* edges, int32_t* n_anchor, int32_t* counts, uint16_t* succ, uint16_t* choice,
                    uint16_t* walk, uint32_t* move_edge, uint16_t* move_succ, const int32_t* idx_cell, void* stream);
int flm_distance_chains(const uint16_t* root, const uint16_t* steps, const uint32_t* edges, const int32_t* n_anchor,
                        const int32_t* n_cells, const int32_t* cell_idx, const int32_t* slot_target,
                        const int32_t* n_slots, int E, int A, int H, int W, int capCells, int capAnch, int ldsAnch,
                        uint16_t* dist, int32_t* counts, void* stream);

/* Greedy descent on the distance map, ties broken left, forward, right.
 * Replaces get_shortest_paths, src/utils/shortest_path.py:11-90.
 *   start_cell/start_dir [E,A] int32: the position chosen at :38-49; status [E,A] int32
 *   path_cell [E,A,Lcap] int32, path_meta [E,A,Lcap] uint32 = heading | action<<8 | next heading<<16
 *   (next cell = cell + move(next heading), or the same cell when action == 4)
 *   path_len  [E,A] int32, -1 = None.  Lcap >= max_depth + 1.
 *   succ, choice: the tables written by flm_chain_index for the same targets, or NULL (same paths, more
 *   look-ups; choice is only used together with succ). */
int flm_shortest_paths(const uint16_t* grid, const int32_t* cell_idx, const uint16_t* dist, const uint16_t* succ,
                       const uint16_t* choice, const int32_t* tslot, const int32_t* start_cell, const int32_t* start_dir,
                       const int32_t* target_cell, const int32_t* status, int E, int A, int H, int W,
                       int capCells, int max_depth, int Lcap, int32_t* path_cell, uint32_t* path_meta,
                       int32_t* path_len, void* stream);

/* flm_shortest_paths + flm_columns of every agent from the chain index alone: the dense distance map of
 * flm_distance_chains is not needed (anchor distances, the walks and the column scans stay in LDS; two
 * launches hand each other two bits per (anchor, target): the move a walk takes there).  Replaces what env.reset() makes get_many compute:
 * ShortestPathPredictorForRailEnv.get -> get_shortest_paths (src/predictions.py:36-72,
 * src/utils/shortest_path.py:11-90, with flatland's DistanceMap.get() as used at :60-62), prediction_from_path
 * (src/predictions.py:85-143), _bitmap_from_cells_seq and the shift (src/rail_observations.py:78-95, 323-407).
 *   walk, root, steps, edges, move_edge, move_succ, n_anchor: tables of flm_chain_index; tslot / n_slots / slot_target of
 *   flm_target_slots; cell_idx / idx_cell / n_cells of flm_state_index; cell_rail / node_id / edge_info of
 *   flm_graph_build;
 *   start_cell / start_dir / status as flm_shortest_paths, pos_cell / tpc as flm_columns (all [E,A]);
 *   ldsAnch / ldsSlots (a multiple of 8): anchors and distinct targets per environment the LDS tables are sized
 *   for; an environment with more is flagged in counts[e*4+3] (FLM_GRAPH_ERR_STATES) and its outputs are left
 *   unwritten; flm_paths_columns_lds_bytes gives the LDS bytes a launch needs (FLM_E_CAPACITY above 159.5 KiB);
 *   workspace: flm_paths_columns_workspace(E, A, ldsAnch, ldsSlots) int32 words;
 *   path_cell / path_meta [E,A,Lcap] (both or neither; NULL = paths are not stored), path_len [E,A], cols
 *   [E,A,T1], err [E,A]: exactly what flm_shortest_paths followed by flm_columns write.
 * H, W <= 256, capCells*4 <= 32768, T1 <= 512, Lcap >= max_depth + 1. */
long flm_paths_columns_lds_bytes(int A, int capCells, int ldsAnch, int ldsSlots, int capR, int Lcap, int T1);
long flm_paths_columns_workspace(int E, int A, int ldsAnch, int ldsSlots);   /* int32 words */
int flm_paths_columns(const uint16_t* walk, const uint16_t* root, const uint16_t* steps, const uint32_t* edges,
                      const uint32_t* move_edge,
                      const uint16_t* move_succ, const int32_t* n_anchor, const int32_t* n_cells,
                      const int32_t* cell_idx, const int32_t* idx_cell, const int32_t* cell_rail, const int32_t* node_id,
                      const int32_t* edge_info, const int32_t* tslot, const int32_t* n_slots,
                      const int32_t* slot_target, const int32_t* start_cell, const int32_t* start_dir,
                      const int32_t* pos_cell, const int32_t* target_cell, const int32_t* status, const int32_t* tpc,
                      int E, int A, int H, int W, int capCells, int capAnch, int ldsAnch, int ldsSlots, int capR, int Rp,
                      int T1, int max_depth, int Lcap, int shift, int32_t* workspace, int32_t* path_cell,
                      uint32_t* path_meta, int32_t* path_len, int16_t* cols, int32_t* err, int32_t* counts, void* stream);

/* Path -> predicted cell per time step -> rail-occupancy plane, optionally moved one step to the right.
 * Replaces prediction_from_path + cells_seq_from_prediction, src/predictions.py:85-143,
 * _bitmap_from_cells_seq, src/rail_observations.py:323-407, and the shift of get_many
 * (src/rail_observations.py:92-93) / get_altmaps (:105-107).
 *   n_paths paths in all: path_cell [n_paths,Lcap], path_len [n_paths]; agent_flat [n_paths] int32 gives
 *   the agent (e*A + a) each path belongs to (NULL = identity, n_paths == E*A); pos_cell [E,A] the agent's
 *   virtual position (src/predictions.py:89-94); tpc [E,A] time steps per cell;
 *   planes [n_paths,plane_stride] int8 out; err [n_paths] int32 out; cols_ws [n_paths,T1] int16 is a
 *   workspace that receives the column form (see flm_columns). */
int flm_bitmaps(const int32_t* path_cell, const int32_t* path_len, const int32_t* agent_flat,
                const int32_t* pos_cell, const int32_t* target_cell, const int32_t* tpc,
                const int32_t* node_id, const int32_t* cell_rail, const int32_t* edge_info, int E, int A,
                int n_paths, int H, int W, int capR, int Rp, int T1, int plane_stride, int Lcap, int shift,
                int16_t* cols_ws, int8_t* planes, int32_t* err, void* stream);

/* flm_bitmaps for callers that cannot allocate device memory (code under HIP graph capture, serving loops,
 * harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments and results as flm_bitmaps
 * without cols_ws: the column-form workspace is a fixed 4 MiB array inside the library (one copy per GPU, held
 * until the process ends), so n_paths * T1 <= FLM_BITMAPS_STATIC_MAX_COLS (2,097,152); larger calls return
 * FLM_E_CAPACITY and launch nothing.  The call allocates and frees nothing.  Every call on a GPU uses that GPU's one
 * array: calls that overlap on the same GPU (different streams or host threads) race; calls on one stream do not. */
#define FLM_BITMAPS_STATIC_MAX_COLS (2 * 1024 * 1024)
int flm_bitmaps_static(const int32_t* path_cell, const int32_t* path_len, const int32_t* agent_flat,
                       const int32_t* pos_cell, const int32_t* target_cell, const int32_t* tpc,
                       const int32_t* node_id, const int32_t* cell_rail, const int32_t* edge_info, int E, int A,
                       int n_paths, int H, int W, int capR, int Rp, int T1, int plane_stride, int Lcap, int shift,
                       int8_t* planes, int32_t* err, void* stream);

/* Observation of every agent: its own plane and the two heat planes "sum over the other agents" of the
 * positive and of the negative entries.
 * Replaces ObsPreprocessor._get_heatmap/_fill_padding/get_obs, src/preprocessing.py:8-30, 61-74
 * (without rail reordering) for all agents of all environments in one pass.
 *   maps [E,A,plane_stride] int8; own (may be NULL = maps) the plane shown as the agent's own
 *   obs  [E,A,3,plane_stride] uint8 out: own plane (int8 bit pattern), positive heat, negative heat.
 * A <= 255. */
int flm_observations(const int8_t* maps, const int8_t* own, int E, int A, int plane_stride, uint8_t* obs,
                     void* stream);

/* The fused form of flm_bitmaps + flm_observations, which never materialises the dense input planes:
 *   flm_columns              same inputs as flm_bitmaps; cols [n_paths,T1] int16 out: per time step
 *                            rail*2 + (sign > 0), -1 where the column is empty (a plane has at most one
 *                            entry per column, src/rail_observations.py:397-401); err as flm_bitmaps;
 *   flm_observations_columns cols [E,A,T1] -> obs [E,A,3,plane_stride] uint8, identical to what
 *                            flm_observations writes for the planes those columns describe.  obs[e,a,0]
 *                            is the agent's dense plane (stride 3*plane_stride between agents).
 * T1 <= 512, A <= 255, Rp <= 16383. */
int flm_columns(const int32_t* path_cell, const int32_t* path_len, const int32_t* agent_flat,
                const int32_t* pos_cell, const int32_t* target_cell, const int32_t* tpc, const int32_t* node_id,
                const int32_t* cell_rail, const int32_t* edge_info, int E, int A, int n_paths, int H, int W, int capR,
                int Rp, int T1, int Lcap, int shift, int16_t* cols, int32_t* err, void* stream);
int flm_observations_columns(const int16_t* cols, int E, int A, int Rp, int T1, int plane_stride, uint8_t* obs,
                             void* stream);

/* The same observation for selected agents only (rollout: the agents that consult the network this step).
 * Replaces one ObsPreprocessor.get_obs call per request, src/preprocessing.py:61-74.
 *   req_agent [n_req] int32 = e*A + a; own_planes [n_req,plane_stride] int8 or NULL (an alternative plane
 *   shown instead of the agent's own; the agent's plane in maps is what is left out of the heat sums,
 *   src/preprocessing.py:25-26); out [n_req,3,plane_stride] uint8.  n_req <= 65535, A <= 255. */
int flm_observations_req(const int8_t* maps, const int32_t* req_agent, const int8_t* own_planes, int n_req, int E,
                         int A, int plane_stride, uint8_t* out, void* stream);

/* flm_observations_req on the column form (see flm_columns): cols [E,A,T1] int16 are the agents' planes,
 * own_cols [.,T1] int16 or NULL the planes shown as their own: row own_row[g] of own_cols for request g
 * (own_row NULL = row g; a negative own_row[g] = the agent's own plane).  Same output as
 * flm_observations_req for the planes those columns describe. */
int flm_observations_req_columns(const int16_t* cols, const int32_t* req_agent, const int16_t* own_cols,
                                 const int32_t* own_row, int n_req, int E, int A, int Rp, int T1, int plane_stride,
                                 uint8_t* out, void* stream);
/* The same launch sized before the host knows the number of requests: n_max is an upper bound, *n_live (memory the
 * device can read, written by the kernel that counted the requests - flm_rollout_alt_finish's counts[2] - earlier on
 * the same stream) their number; requests from *n_live on are skipped.  out holds n_max rows.  A caller uses it to
 * enqueue the observations first and wait for the count afterwards, instead of leaving the device idle meanwhile.
 * plane_limit (0 = plane_stride; a multiple of 16): only the first plane_limit bytes of every plane are written.  With
 * observations padded to max_rails the rails beyond the environments' own are zero: a caller whose buffer holds zeros
 * there (and who raises the limit, or clears the buffer, when the rail count changes) saves writing them every step. */
int flm_observations_req_columns_counted(const int16_t* cols, const int32_t* req_agent, const int16_t* own_cols,
                                         const int32_t* own_row, int n_max, const int32_t* n_live, int plane_limit, int E,
                                         int A, int Rp, int T1, int plane_stride, uint8_t* out, void* stream);

/* flm_observations_req_columns of the agents' own planes, written to chosen rows: request g goes to row
 * out_row[g] of out [.,3,plane_stride] (a negative out_row[g] skips the request).  The training loop's
 * next-state observations (get_obs on the planes after the step, src/main.py:277) go straight into the replay
 * memory's ring this way.  Any n_req. */
int flm_observations_req_columns_rows(const int16_t* cols, const int32_t* req_agent, const int32_t* out_row, int n_req,
                                      int E, int A, int Rp, int T1, int plane_stride, uint8_t* out, void* stream);

/* Rail reordering of finished observations.
 * Replaces ObsPreprocessor._reorder_rails / _swap_rails, src/preprocessing.py:32-59 (enabled by
 * --reorder-rails, src/main.py:395): the rails the own plane (plane 0) visits, in order of first visit,
 * become rows 0..k-1 of all three planes; every other row is cleared.  Out of place: obs_in != obs_out,
 * both [n,3,plane_stride] uint8 with planes of Rp rows x T1 bytes.
 *   err [n] int32: 0, or FLM_AGENT_NONTERM when the visited rail stays occupied up to the last time step
 *   (the reference's loop never ends there); such an observation is copied unchanged. */
int flm_reorder_rails(const uint8_t* obs_in, uint8_t* obs_out, int n, int Rp, int T1, int plane_stride,
                      int32_t* err, void* stream);

/* Alternative paths of agents that stand in front of a switch.
 * Replaces get_altpaths, src/utils/shortest_path.py:92-163 (called through
 * ShortestPathPredictorForRailEnv.get_altpaths, src/predictions.py:74-83, with max_depth = 500).
 *   req_agent [n_req] int32: e*A + a of each request; pos_cell/direction/target_cell [E,A] as above
 *   succ, choice [E,capCells*4], [E,capCells*4,4] uint16: the forced-successor and choice tables of flm_chain_index
 *   (or NULL: every step is then worked out from the grid); idx_cell [E,capCells] int32: the grid cell of every
 *   rail cell (flm_state_index), or NULL; with all three and tables that fit the LDS a request's descents run on
 *   staged copies
 *   alt_count [n_req] out; alt_len [n_req,cap_alt] out (0 for the alternatives beyond alt_count);
 *   alt_cell/alt_meta [n_req,cap_alt,Lcap] out in the
 *   path encoding of flm_shortest_paths; err [n_req] out (FLM_AGENT_ASSERT_ERROR where the reference's
 *   assertions at :108/:125 fail, FLM_AGENT_OVERFLOW when the branches before the first plain cells number
 *   more than cap_alt, or Lcap or 32 consecutive switch cells are exceeded).  Lcap >= max_depth + 4,
 *   cap_alt <= 64.
 * Alternatives come in the reference's order: moves left, forward, right, depth first. */
int flm_altpaths(const uint16_t* grid, const int32_t* cell_idx, const uint16_t* dist, const uint16_t* succ,
                 const uint16_t* choice, const int32_t* idx_cell, const int32_t* tslot, const int32_t* node_id,
                 const int32_t* req_agent, int n_req, const int32_t* pos_cell,
                 const int32_t* direction, const int32_t* target_cell, int E, int A, int H, int W, int capCells,
                 int max_depth, int cap_alt, int Lcap, int32_t* alt_count, int32_t* alt_len, int32_t* alt_cell,
                 uint32_t* alt_meta, int32_t* err, void* stream);
/* flm_altpaths sized before the host knows the number of requests (see flm_observations_req_columns_counted): the
 * launch covers requests live_base .. live_base + n_max - 1 of the *n_live that flm_rollout_requests counted
 * (counts[1]); req_agent and the outputs point at request live_base.  Requests beyond *n_live are not touched.
 * agent_rep (optional) [n_max,cap_alt] int32: the request's agent once per alternative, the agent list flm_columns takes
 * for the paths in alt_cell. */
int flm_altpaths_counted(const uint16_t* grid, const int32_t* cell_idx, const uint16_t* dist, const uint16_t* succ,
                         const uint16_t* choice, const int32_t* idx_cell, const int32_t* tslot, const int32_t* node_id,
                         const int32_t* req_agent, int n_max, const int32_t* n_live, int live_base, int32_t* agent_rep,
                         const int32_t* pos_cell, const int32_t* direction, const int32_t* target_cell, int E, int A, int H,
                         int W, int capCells, int max_depth, int cap_alt, int Lcap, int32_t* alt_count, int32_t* alt_len,
                         int32_t* alt_cell, uint32_t* alt_meta, int32_t* err, void* stream);

/* Would the selected agent of each environment crash if it moved now?
 * Replaces RailObsForRailEnv.check_crash with _is_cell_occupied, _check_headon_crash, _last_train_on_rail
 * and _get_rail_dir, src/rail_observations.py:161-164, 182-230, 257-299.
 *   maps [E,A,plane_stride] int8; status/tpc/pos_cell/init_cell [E,A] int32 (pos_cell -1 = not on the grid)
 *   agent_sel [E] int32 (-1 = skip the environment), before_switch [E] int32, next_cell [E] int32 (the
 *   next cell of the agent's path, -1 when the path is empty; used when the agent is neither
 *   READY_TO_DEPART nor before a switch); crash [E] int32 out (untouched where agent_sel < 0). */
int flm_check_crash(const int8_t* maps, const int32_t* status, const int32_t* tpc, const int32_t* pos_cell,
                    const int32_t* init_cell, const int32_t* agent_sel, const int32_t* before_switch,
                    const int32_t* next_cell, int E, int A, int Rp, int T1, int plane_stride, int32_t* crash,
                    void* stream);

/* One time step passes for the selected agent of each environment: its plane moves one column to the
 * left; before a switch the agent is first delayed behind the last train on the rail it is about to enter.
 * Replaces RailObsForRailEnv.update_bitmaps and _delay, src/rail_observations.py:167-180, 232-252.
 * maps is updated in place. */
int flm_update_bitmaps(int8_t* maps, const int32_t* status, const int32_t* tpc, const int32_t* agent_sel,
                       const int32_t* before_switch, int E, int A, int Rp, int T1, int plane_stride, void* stream);

/* ---- Batched rollout step on the column form of the planes (cols [E,A,T1] int16, see flm_columns) ----
 * Paths are [E,A,Lp] arrays in the encoding of flm_shortest_paths with a read position path_head [E,A]
 * and a length path_len [E,A] (-1 = None): the list paths[a] of the reference is elements head..len-1.
 *
 * flm_rollout_classify: which case of the training loop (src/main.py:146-249) every agent is in:
 *   branch [E,A] out: 0 done (:147), 1 ready to depart (:160), 2 before a switch with an action required
 *   (:187), 3 action required (:232), 4 otherwise (:243); before_switch [E,A] out follows is_before_switch,
 *   src/rail_observations.py:142-159.  status/pos_cell/action_required [E,A]; cell_rail of flm_graph_build. */
int flm_rollout_classify(const int32_t* status, const int32_t* pos_cell, const int32_t* action_required,
                         const int32_t* path_cell, const uint32_t* path_meta, const int32_t* path_head,
                         const int32_t* path_len, const int32_t* cell_rail, int E, int A, int H, int W, int Lp,
                         int32_t* branch, int32_t* before_switch, void* stream);

/* flm_rollout_apply: the per-agent part of one step of the training loop for every environment, agents
 * in handle order (src/main.py:146-249 with check_crash, update_bitmaps, _delay, get_agent_action and
 * set_agent_path, src/rail_observations.py:113-140, 161-299), given the stop/go decisions:
 *   net_in [E,A]: 1 = go, for the agents of cases 1 and 2 (the network's choice, src/main.py:168-172, 202-208);
 *   alt_slot [E,A]: row of the alt_* tables that holds the agent's alternatives (-1 = none), best [E,A] the
 *   alternative it commits to; alt_count [n], alt_len [n,cap_alt], alt_cell/alt_meta [n,cap_alt,Lp] as
 *   written by flm_altpaths, alt_cols [n,cap_alt,T1] their planes in column form (flm_columns);
 *   cols, path_* are updated in place; action [E,A] out is the RailEnvActions value handed to the
 *   environment, crash [E,A] out, net_out [E,A] the decision after the crash checks.
 * T1 <= 512, A <= 512 (FLM_E_ARG beyond). */
int flm_rollout_apply(int16_t* cols, const int32_t* status, const int32_t* tpc, const int32_t* pos_cell,
                      const int32_t* init_cell, const int32_t* branch, const int32_t* net_in, const int32_t* alt_slot,
                      const int32_t* best, const int32_t* alt_count, const int32_t* alt_len, const int32_t* alt_cell,
                      const uint32_t* alt_meta, const int16_t* alt_cols, int32_t* path_cell, uint32_t* path_meta,
                      int32_t* path_head, int32_t* path_len, int E, int A, int W, int T1, int Lp, int cap_alt,
                      int32_t* action, int32_t* crash, int32_t* net_out, void* stream);

/* Request lists and the stop/go policy on the device (the glue of src/main.py:146-213 between the cases, the
 * alternatives, the network and the bookkeeping; no counterpart of their own in the reference).
 *   flm_rollout_requests   ready [E*A] / sw [E*A] out: the agents (e*A + a, ascending) in case 1 / case 2 of
 *                          branch; counts[0..1] their numbers (counts has 3 ints); alt_slot [E,A] out: position
 *                          in sw or -1; net/best [E,A] cleared; batch_pos_cell/direction/status [E,A] out: the
 *                          agents' cell (start cell while off the grid), heading, status for the path kernels;
 *                          workspace: 2*ceil(E*A/1024) int32.
 *   flm_rollout_alt_finish after flm_altpaths + flm_columns for the n_switch agents of sw: an agent whose
 *                          alternatives raised is flagged in agent_err [E,A] (first error kept) and its
 *                          alt_count set to 0; row_offset [n_switch+1] out = alternatives before agent i;
 *                          row_req / row_own [n_ready + n_switch*cap_alt] out list what the network evaluates:
 *                          the n_ready departing agents (row_own -1), then alternative i*cap_alt + k of
 *                          every agent; counts[2] = number of rows.
 *   flm_policy_select      epsilon-greedy choice from q [rows,2] (src/main.py:168-172, 201-208): net/best
 *                          [E,A] written for the consulting agents; sel_row / sel_act [n_ready + n_switch]
 *                          out: the row each consulting agent acted on (-1 = no alternative) and its action.
 *                          Random numbers: a counter hash of (seed, step, agent). */
int flm_rollout_requests(const int32_t* branch, const int32_t* status, const int32_t* pos_cell,
                         const int32_t* direction, const int32_t* init_cell, int E, int A, int32_t* ready, int32_t* sw,
                         int32_t* counts, int32_t* alt_slot, int32_t* net, int32_t* best, int32_t* batch_pos_cell,
                         int32_t* batch_direction, int32_t* batch_status, int32_t* workspace, void* stream);
int flm_rollout_alt_finish(int32_t* alt_count, const int32_t* alt_err, const int32_t* col_err, const int32_t* sw,
                           int n_switch, int cap_alt, const int32_t* ready, int n_ready, int32_t* agent_err,
                           int32_t* row_offset, int32_t* row_req, int32_t* row_own, int32_t* counts, void* stream);
int flm_policy_select(const float* q, const int32_t* ready, int n_ready, const int32_t* sw, int n_switch,
                      const int32_t* row_offset, const int32_t* alt_count, double eps, uint32_t seed, int step,
                      int32_t* net, int32_t* best, int32_t* sel_row, int32_t* sel_act, void* stream);

/* flm_env_step: one environment step for every environment, agents in handle order.  Stands in for
 * flatland RailEnv.step as called at src/main.py:252 (third-party, absent: restated from its published
 * rules; the same rules as the stand-in the golden traces were recorded with).
 *   status/pos_cell/direction/moving/fraction/exit_action/done [E,A] are updated in place (fraction,
 *   speed, reward are double); reward [E,A], action_required [E,A], all_done [E] out.
 * Malfunction process (the malfunction_from_params generator handed to RailEnv at src/main.py:72-77):
 *   malfunction [E,A] = steps an agent stays broken (in/out; NULL switches the process off and the
 *   remaining malfunction arguments are ignored), malfunction_moving [E,A] = `moving` at the time of the
 *   break, env_seed [E] one 32-bit seed per environment, malfunction_threshold = floor((1 - exp(-1/rate))
 *   * 2^32), durations min_duration..max_duration (+1, as flatland draws them), step_index = number of
 *   steps since reset.  The draws are a counter hash of (seed, step, agent), not flatland's MT19937
 *   stream. */
int flm_env_step(const uint16_t* grid, const int32_t* action, const int32_t* init_cell, const int32_t* init_dir,
                 const int32_t* target_cell, const double* speed, int32_t* status, int32_t* pos_cell,
                 int32_t* direction, int32_t* moving, double* fraction, int32_t* exit_action, int32_t* done, int E,
                 int A, int H, int W, double* reward, int32_t* action_required, int32_t* all_done,
                 int32_t* malfunction, int32_t* malfunction_moving, const uint32_t* env_seed,
                 uint32_t malfunction_threshold, int min_duration, int max_duration, int step_index, void* stream);

/* ---- Dueling Double-DQN (fp32 values and accumulation; matrix products on the matrix cores, the large ones with
 * every fp32 operand split into three bf16 terms: accuracy of an fp32 multiply-add, Inf/NaN operands give NaN) -------------------------------------
 * Replaces Dueling_DQN.forward, src/model.py:31-43, and the numeric part of DQNAgent.act / learn /
 * soft_update, src/agent.py:69-131.  H3 = 3*max_rails input rows, T1 = depth+1 columns, NA = number of
 * actions (the reference's action_space; src/main.py:92 uses 2), 1 <= NA <= FLM_DQN_MAX_ACTIONS.
 * Parameters are one flat float buffer in the order of the module's parameters():
 *   conv1.weight[64,T1] conv1.bias[64] fc1_adv.weight[512,64*H3] fc1_adv.bias[512] fc1_val.weight[512,64*H3]
 *   fc1_val.bias[512] fc2_adv.weight[NA,512] fc2_adv.bias[NA] fc2_val.weight[1,512] fc2_val.bias[1]. */
#define FLM_DQN_MAX_ACTIONS 16
long flm_dqn_param_count(int H3, int T1, int NA);
long flm_dqn_forward_workspace(int B, int H3);              /* floats */
long flm_dqn_learn_workspace(int B, int H3, int T1, int NA); /* floats */

/* q[B,NA] = network(x[B,H3,T1]).  workspace: flm_dqn_forward_workspace(B,H3) floats. */
int flm_dqn_forward(const float* params, const float* x, int B, int H3, int T1, int NA, float* workspace, float* q,
                    void* stream);

/* The same forward pass straight from uint8 observations [n,3,plane_stride] (plane 0 int8, heat planes
 * uint8, rows of T1 bytes: what flm_observations* write): the convolution reads the bytes, no float copy
 * of the input is made.  workspace: flm_dqn_forward_workspace(n, 3*Rp) floats.
 * Limits (FLM_E_ARG beyond them): n*3*Rp < 2^31 rows; T1 <= 313 (a row tile and the filter bank share the LDS). */
int flm_dqn_forward_u8(const float* params, const uint8_t* obs, int n, int Rp, int T1, int plane_stride, int NA,
                       float* workspace, float* q, void* stream);

/* flm_dqn_forward_u8 for observations whose rails [rails_used, Rp) are empty in every plane (the padding of
 * src/preprocessing.py:36-50 up to max_rails: an environment with R rails fills rows 0..R-1 of each plane only).
 * The empty rows leave the convolution as relu(bias) for every sample, so their share of the first dense layer
 * (src/model.py:33-41) is one vector per set of weights: it is added to the layer's bias, and the convolution and the
 * layer's product run over the used rails only.  Same Q-values as flm_dqn_forward_u8 up to fp32 rounding (the sums are
 * taken in another order).  rails_used = 0, rails_used > Rp - 8 or n <= 8: the plain form runs.  The caller vouches
 * for the empty rows: what they hold is not read.  Workspace as flm_dqn_forward_u8. */
int flm_dqn_forward_u8_rails(const float* params, const uint8_t* obs, int n, int Rp, int T1, int plane_stride, int NA,
                             int rails_used, float* workspace, float* q, void* stream);

/* uint8 observations [n,3,plane_stride] (own plane int8) -> float rows [n,3*Rp,T1], the network's input. */
int flm_obs_to_f32(const uint8_t* obs, int n, int Rp, int T1, int plane_stride, float* out, void* stream);

/* One learn step up to the gradient (src/agent.py:94-114): the forward passes, the mean-squared TD error
 * (loss_out[1]) and its gradient w.r.t. the local network (grads, flat layout).  double_dqn != 0: the action of
 * the next state is the local network's, its value the target network's (:98-101); double_dqn == 0: the
 * reference's plain-DQN branch (:103-105) as it really computes - its next-state maxima keep a trailing axis, so
 * every sample is compared with the targets built from every sample's maximum ([B,B] broadcast inside mse_loss).
 * action int64 [B] in 0..NA-1 (values outside are clamped into the range), reward/done float [B].  workspace: flm_dqn_learn_workspace floats.
 * The greedy action of a tie is the first maximum, as torch.max returns it.
 * The reference also back-propagates into the target network (its Q_targets are not detached); those
 * gradients are never used by the optimizer and are not computed here. */
int flm_dqn_learn_grads(const float* local, const float* target, const float* s, const float* s_next,
                        const int64_t* action, const float* reward, const float* done, float gamma, int double_dqn,
                        int B, int H3, int T1, int NA, float* workspace, float* grads, float* loss_out, void* stream);

/* The same learn step on uint8 observations [B,3,plane_stride] as stored by a device replay memory: the
 * forward passes read the bytes; only the states s are expanded to float (inside the workspace) for the
 * convolution's weight gradient.  workspace: flm_dqn_learn_workspace_u8(B,Rp,T1,NA) floats. */
long flm_dqn_learn_workspace_u8(int B, int Rp, int T1, int NA);
int flm_dqn_learn_grads_u8(const float* local, const float* target, const uint8_t* s, const uint8_t* s_next,
                           int plane_stride, const int64_t* action, const float* reward, const float* done, float gamma,
                           int double_dqn, int B, int Rp, int T1, int NA, float* workspace, float* grads,
                           float* loss_out, void* stream);

/* The same learn step in two calls, for a data-parallel caller that overlaps the gradient exchange with the
 * backward pass (no counterpart in the reference, which is a single process, src/agent.py:14).  phase 1: forward
 * passes, loss, and every gradient from float flm_dqn_grad_split(H3, T1, NA) of grads on (output layers and first
 * dense layer: final values); phase 2, with the same arguments and the workspace untouched in between: the floats
 * before the split (the convolution's weight and bias gradients).  Both phases together launch exactly the kernels of
 * the one-call form in the same order: the same bits. */
long flm_dqn_grad_split(int H3, int T1, int NA);
int flm_dqn_learn_grads_phase(const float* local, const float* target, const float* s, const float* s_next,
                              const int64_t* action, const float* reward, const float* done, float gamma, int double_dqn,
                              int B, int H3, int T1, int NA, float* workspace, float* grads, float* loss_out, int phase,
                              void* stream);
int flm_dqn_learn_grads_u8_phase(const float* local, const float* target, const uint8_t* s, const uint8_t* s_next,
                                 int plane_stride, const int64_t* action, const float* reward, const float* done,
                                 float gamma, int double_dqn, int B, int Rp, int T1, int NA, float* workspace,
                                 float* grads, float* loss_out, int phase, void* stream);

/* flm_dqn_learn_grads_u8 (phase 0) or one half of it (phase 1 / 2, as flm_dqn_learn_grads_u8_phase; fill_padding = 1:
 * see below) for observations
 * whose rails [rails_used, Rp) are empty in every plane, states and next states alike (see flm_dqn_forward_u8_rails).
 * Forward passes and the products of the backward pass run over the used rails only; what the empty rails contribute
 * is added in closed form: to the first dense layer's pre-activations a vector per network, to its weight gradient
 * relu(conv bias[c]) * (the layer's bias gradient) under every empty row, to the convolution's bias gradient the
 * summed activation gradient of those rows; the convolution's weight gradient gets nothing from rows of zeros.  Same
 * loss and gradients as flm_dqn_learn_grads_u8 up to fp32 rounding.  Falls back to the plain form when too few rails
 * are empty, for B <= 8 and for horizons beyond the byte form of the convolution's weight gradient (T1 > 159).
 * Workspace: flm_dqn_learn_workspace_u8. */
long flm_dqn_learn_rails(int B, int Rp, int T1, int rails_used);   /* rails per plane it computes over (0: the plain form runs) */
int flm_dqn_learn_grads_u8_rails(const float* local, const float* target, const uint8_t* s, const uint8_t* s_next,
                                 int plane_stride, const int64_t* action, const float* reward, const float* done,
                                 float gamma, int double_dqn, int B, int Rp, int T1, int NA, int rails_used,
                                 float* workspace, float* grads, float* loss_out, int fill_padding, int phase,
                                 void* stream);

/* fill_padding = 0 or 2 (only when flm_dqn_learn_rails(...) > 0; 2: see the compact form below): the first dense
 * layer's weight gradient under the empty rows is left unwritten - it is relu(conv bias[c]) times the layer's bias gradient, and flm_dqn_adam_soft_update_rails
 * forms it from the bias gradient in `grads` (after any averaging over ranks) and from the 64 values relu(conv bias)
 * that the learn step left in its workspace at float offset flm_dqn_learn_rails_rcb_offset(...): the same update as
 * flm_dqn_adam_soft_update after a step with fill_padding = 1, without writing and reading those products.
 * rails4 = flm_dqn_learn_rails(...); n = flm_dqn_param_count(3*Rp, T1, NA); 512 * 64 * 3 * Rp < 2^24.
 * The compact form (moments_in, moments_out, m_packed, v_packed: all or none) keeps the optimizer's own data of the two
 * weight blocks without the empty rows.  moments_in / moments_out: two different buffers of 2*512*64 float pairs (m, v),
 * 8-byte aligned - the moments of the weights under the empty rows once per (stream, output unit, filter); they are
 * equal along the rows as long as every step so far gave those weights the same gradient, i.e. every step since
 * m = v = 0 ran over used rails <= rails4.  m_packed / v_packed [2][512][64*3*rails4]: the moments under the used rails
 * (column = (filter*3 + plane)*rails4 + rail).  The gradient of the two weight blocks is then read in that packed form
 * from the start of each block of `grads`, where flm_dqn_learn_grads_u8_rails leaves it with fill_padding = 2.  The pass
 * reads moments_in, writes moments_out, updates m_packed / v_packed and does not touch m and v of the two weight blocks
 * (they go stale); parameters and target get the same bits as without.
 * flm_dqn_adam_moments moves the moments between the two forms: fill = 0 takes pairs and packed moments from m and v
 * (the pairs at plane 0, rail rails4), fill = 1 writes m and v of the two weight blocks from them.  The caller keeps
 * track of which form is current. */
int flm_dqn_adam_soft_update_rails(float* params, const float* grads, float* m, float* v, float* target, long n, double lr,
                                   double beta1, double beta2, double eps, int step, double tau, int Rp, int T1, int NA,
                                   int rails4, const float* rcb, const float* moments_in, float* moments_out,
                                   float* m_packed, float* v_packed, void* stream);
int flm_dqn_adam_moments(float* m, float* v, long n, int Rp, int T1, int NA, int rails4, float* moments, float* m_packed,
                         float* v_packed, int fill, void* stream);
long flm_dqn_learn_rails_rcb_offset(int B, int Rp, int T1, int NA);

/* Uniform sampling without replacement from a replay memory that lives in device memory, and the gather of the
 * sampled transitions.  Replaces ReplayBuffer.sample, src/agent.py:159-174 (random.sample over the deque, then
 * one stacked tensor per field).
 *   flm_replay_sample  idx [batch] int32 out: `batch` distinct entries of 0..size-1 in random order: the images of
 *                      0..batch-1 under a pseudo-random permutation of the entries keyed by (seed, draw) - give
 *                      every call its own `draw`.  batch <= size.
 *   flm_replay_gather  rows idx[b] of state / next_state (row_bytes bytes each, a multiple of 16, 16-byte aligned
 *                      bases) and of action int64 / reward float / done float -> the out_* arrays, row b. */
int flm_replay_sample(int size, int batch, uint32_t seed, uint32_t draw, int32_t* idx, void* stream);
int flm_replay_gather(const void* state, const void* next_state, const int64_t* action, const float* reward,
                      const float* done, const int32_t* idx, int batch, long row_bytes, void* out_state, void* out_next,
                      int64_t* out_action, float* out_reward, float* out_done, void* stream);
/* flm_replay_gather for rows of n_planes planes, plane_bytes apart, of which only the first used_bytes matter
 * (observations whose rails beyond the environments' own are empty, see flm_dqn_learn_grads_u8_rails): those bytes of
 * every plane are gathered, the rest of the output rows is left as it is.  plane_bytes and used_bytes multiples of 16. */
int flm_replay_gather_planes(const void* state, const void* next_state, const int64_t* action, const float* reward,
                             const float* done, const int32_t* idx, int batch, long row_bytes, int n_planes,
                             long plane_bytes, long used_bytes, void* out_state, void* out_next, int64_t* out_action,
                             float* out_reward, float* out_done, void* stream);

/* flm_replay_sample(size, batch, seed, draw, idx) followed by flm_replay_gather_planes(..., idx, ...) in one launch:
 * the same idx [batch] (still written) and the same gathered rows and scalars.  Whole rows: n_planes = 1,
 * plane_bytes = used_bytes = row_bytes.  row_bytes < 2^31.  Replaces ReplayBuffer.sample, src/agent.py:159-174. */
int flm_replay_draw_gather(const void* state, const void* next_state, const int64_t* action, const float* reward,
                           const float* done, int size, uint32_t seed, uint32_t draw, int batch, long row_bytes,
                           int n_planes, long plane_bytes, long used_bytes, void* out_state, void* out_next,
                           int64_t* out_action, float* out_reward, float* out_done, int32_t* idx, void* stream);

/* Storing one rollout step's transitions in the replay memory's ring.  Replaces the per-agent dqn.step(...) calls of
 * the training script's default feeding mode, src/main.py:275-278 (-> ReplayBuffer.add, src/agent.py:153-157), for
 * every environment of the batch at once.
 *   ready [n_ready], sw [n_switch] int32   the agents (e*A + a) that consulted the network this step: departing
 *                                          agents, then agents in front of a switch (flm_rollout_requests)
 *   sel_row [n_ready+n_switch] int32       row of obs the agent acted on, -1 = it had nothing to choose from and
 *                                          is not stored (flm_policy_select)
 *   obs [.,row_bytes] bytes                the step's network rows (row_bytes % 16 == 0, 16-byte aligned)
 *   net [E*A] int32, reward [E*A] double, done [E*A] int32   per agent after the step: the decision left by the
 *                                          crash checks (src/main.py:180,225), the step's reward and done flag
 *   head, capacity                         next ring slot to write and the ring's size
 *   ring_state [capacity,row_bytes], ring_action int64 / ring_reward float / ring_done float [capacity]   the ring
 * Stored requests take consecutive slots from head on (modulo capacity) in request order; when one step stores more
 * than capacity, the last capacity are kept (deque(maxlen), src/agent.py:147).  Out: slot [n] int32 = ring slot of
 * request i or -1; req_out [n] int32 = the concatenated agent list; count_out [1] int32 = transitions of the step
 * (kept or not: what the learn cadence counts).  The next-state rows are then written with
 * flm_observations_req_columns_rows(cols, req_out, slot, ...) into the ring's next-state array. */
int flm_replay_store(const int32_t* ready, int n_ready, const int32_t* sw, int n_switch, const int32_t* sel_row,
                     const void* obs, long row_bytes, const int32_t* net, const double* reward, const int32_t* done,
                     int head, int capacity, void* ring_state, int64_t* ring_action, float* ring_reward,
                     float* ring_done, int32_t* slot, int32_t* req_out, int32_t* count_out, void* stream);

/* flm_replay_store_slots: the replay feeding of one rollout step in the modes of the training script that keep an
 * observation per agent between steps - the crash penalty (src/main.py:271-273) and switch-to-switch
 * (src/main.py:280-296) - and the default transitions beside them (src/main.py:275-278).  Requests, sel_row, obs, net,
 * reward, done, head, capacity and the ring arrays as for flm_replay_store; in addition
 *   crash [n_agents] int32                 the step's crash flags (src/main.py:180, 225, 245)
 *   crash_penalty, switch2switch           the script's two flags (src/main.py:394, 396)
 *   curr_obs [n_agents,row_bytes], has_curr [n_agents] bytes   curr_obs of src/main.py:165,213 and whether it is set
 *   buf_obs, has_buf, buf_rew [n_agents] double   buffer_obs / buffer_rew of src/main.py:123,131 (switch2switch only;
 *                                          NULL otherwise); all five are read and updated
 *   ring_next [capacity,row_bytes]         the ring's next-state rows
 *   workspace int32 [flm_replay_store_slots_workspace(n_agents, n_ready + n_switch)]
 * In stream order: every request with a row sets curr_obs / has_curr of its agent; the step's transitions are listed
 * in the order they enter the ring - penalties (agents ascending with a crash flag and a curr_obs: state = next =
 * curr_obs, action 1, reward -100, done), then either the default transitions of flm_replay_store or, switch2switch,
 * first the matured ones (requests whose agent holds a buffer_obs: state buffer_obs, action 1, reward buffer_rew,
 * next curr_obs, the step's done flag; buffer released) and then the stops (requests whose decision is 0: state =
 * next = curr_obs, action 1, the step's reward, not done); a request that goes keeps curr_obs in buffer_obs, and
 * buffer_rew collects the step's reward of every agent holding a buffer_obs.  Slots as in flm_replay_store (the last
 * capacity transitions of an oversized step are kept).  Out: count_out [1]; without switch2switch also slot / req_out
 * [n_ready + n_switch] for flm_observations_req_columns_rows, which writes the default transitions' next states.
 * n_ready + n_switch may be 0 (penalties of agents that crashed without consulting the network; the reward
 * collection). */
long flm_replay_store_slots_workspace(int n_agents, int n_requests);
int flm_replay_store_slots(const int32_t* ready, int n_ready, const int32_t* sw, int n_switch, const int32_t* sel_row,
                           const void* obs, long row_bytes, const int32_t* net, const double* reward, const int32_t* done,
                           const int32_t* crash, int n_agents, int crash_penalty, int switch2switch, void* curr_obs,
                           uint8_t* has_curr, void* buf_obs, uint8_t* has_buf, double* buf_rew, int head, int capacity,
                           void* ring_state, void* ring_next, int64_t* ring_action, float* ring_reward, float* ring_done,
                           int32_t* workspace, int32_t* slot, int32_t* req_out, int32_t* count_out, void* stream);

/* torch.optim.Adam step with default options over flat buffers (src/agent.py:38,113-115); step counts from 1. */
int flm_dqn_adam(float* params, const float* grads, float* m, float* v, long n, double lr, double beta1, double beta2,
                 double eps, int step, void* stream);

/* flm_dqn_adam followed by flm_dqn_soft_update of target with the new parameters, in one pass over the
 * buffers (src/agent.py:113-118): the same bits as the two calls. */
int flm_dqn_adam_soft_update(float* params, const float* grads, float* m, float* v, float* target, long n, double lr,
                             double beta1, double beta2, double eps, int step, double tau, void* stream);

/* target <- tau*local + (1-tau)*target, src/agent.py:120-131. */
int flm_dqn_soft_update(float* target, const float* local, long n, double tau, void* stream);

#ifdef __cplusplus
}
#endif
#endif
