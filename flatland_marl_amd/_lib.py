"""ctypes binding of the C ABI declared in include/flmarl.h (libflmarl_hip.so).

There is no CPU fallback: if the shared library is missing or a call fails,
the caller gets an exception.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FLMARL_LIB names another build of the same library (experiments); the default is the in-tree one
LIB_PATH = os.environ.get("FLMARL_LIB") or os.path.join(_HERE, "libflmarl_hip.so")

FLM_OK = 0
AGENT_ERRORS = {2: IndexError, 3: KeyError, 4: AssertionError, 5: OverflowError}
GRAPH_ERR_NODES, GRAPH_ERR_RAILS, GRAPH_ERR_CELLS, GRAPH_ERR_STATES = 1, 2, 4, 8
MAX_NODES = 2048
ABI_VERSION = 4
DQN_MAX_ACTIONS = 16                                       # FLM_DQN_MAX_ACTIONS of include/flmarl.h

_p = C.c_void_p
_i = C.c_int
_u32 = C.c_uint32

_SIGNATURES = {
    "flm_abi_version": [],
    "flm_graph_build": [_p, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p],
    "flm_state_index": [_p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p],
    "flm_distance_bfs": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p],
    "flm_target_slots": [_p, _i, _i, _p, _p, _p, _p],
    "flm_chain_index": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "flm_paths_columns_lds_bytes": [_i, _i, _i, _i, _i, _i, _i],
    "flm_paths_columns_workspace": [_i, _i, _i, _i],
    "flm_paths_columns": [_p] * 22 + [_i] * 14 + [_p] * 8,
    "flm_distance_chains": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p],
    "flm_shortest_paths": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p],
    "flm_bitmaps": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p],
    "flm_bitmaps_static": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p],
    "flm_observations": [_p, _p, _i, _i, _i, _p, _p],
    "flm_columns": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p],
    "flm_observations_columns": [_p, _i, _i, _i, _i, _i, _p, _p],
    "flm_observations_req": [_p, _p, _p, _i, _i, _i, _i, _p, _p],
    "flm_reorder_rails": [_p, _p, _i, _i, _i, _i, _p, _p],
    "flm_observations_req_columns": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p],
    "flm_observations_req_columns_counted": [_p, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p, _p],
    "flm_rollout_requests": [_p] * 5 + [_i, _i] + [_p] * 11,
    "flm_rollout_alt_finish": [_p] * 4 + [_i, _i, _p, _i] + [_p] * 6,
    "flm_policy_select": [_p, _p, _i, _p, _i, _p, _p, C.c_double, _u32, _i, _p, _p, _p, _p, _p],
    "flm_rollout_classify": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p],
    "flm_rollout_apply": [_p] * 18 + [_i] * 6 + [_p, _p, _p, _p],
    "flm_env_step": [_p] * 13 + [_i] * 4 + [_p] * 6 + [_u32, _i, _i, _i, _p],
    "flm_altpaths": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p],
    "flm_altpaths_counted": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i,
                             _p, _p, _p, _p, _p, _p],
    "flm_check_crash": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p],
    "flm_update_bitmaps": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "flm_dqn_param_count": [_i, _i, _i],
    "flm_dqn_forward_workspace": [_i, _i],
    "flm_dqn_learn_workspace": [_i, _i, _i, _i],
    "flm_dqn_learn_workspace_u8": [_i, _i, _i, _i],
    "flm_dqn_learn_grads_u8": [_p, _p, _p, _p, _i, _p, _p, _p, C.c_float, _i, _i, _i, _i, _i, _p, _p, _p, _p],
    "flm_dqn_grad_split": [_i, _i, _i],
    "flm_dqn_learn_grads_phase": [_p, _p, _p, _p, _p, _p, _p, C.c_float, _i, _i, _i, _i, _i, _p, _p, _p, _i, _p],
    "flm_dqn_learn_grads_u8_phase": [_p, _p, _p, _p, _i, _p, _p, _p, C.c_float, _i, _i, _i, _i, _i, _p, _p, _p, _i, _p],
    "flm_dqn_learn_rails": [_i, _i, _i, _i],
    "flm_dqn_learn_grads_u8_rails": [_p, _p, _p, _p, _i, _p, _p, _p, C.c_float, _i, _i, _i, _i, _i, _i, _p, _p, _p, _i, _i, _p],
    "flm_dqn_learn_rails_rcb_offset": [_i, _i, _i, _i],
    "flm_dqn_adam_soft_update_rails": [_p, _p, _p, _p, _p, C.c_long, C.c_double, C.c_double, C.c_double, C.c_double, _i,
                                       C.c_double, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p],
    "flm_dqn_adam_moments": [_p, _p, C.c_long, _i, _i, _i, _i, _p, _p, _p, _i, _p],
    "flm_dqn_forward": [_p, _p, _i, _i, _i, _i, _p, _p, _p],
    "flm_obs_to_f32": [_p, _i, _i, _i, _i, _p, _p],
    "flm_dqn_forward_u8": [_p, _p, _i, _i, _i, _i, _i, _p, _p, _p],
    "flm_dqn_forward_u8_rails": [_p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _p],
    "flm_dqn_learn_grads": [_p, _p, _p, _p, _p, _p, _p, C.c_float, _i, _i, _i, _i, _i, _p, _p, _p, _p],
    "flm_observations_req_columns_rows": [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p],
    "flm_replay_store": [_p, _i, _p, _i, _p, _p, C.c_long, _p, _p, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p],
    "flm_replay_store_slots_workspace": [_i, _i],
    "flm_replay_store_slots": [_p, _i, _p, _i, _p, _p, C.c_long, _p, _p, _p, _p, _i, _i, _i, _p, _p, _p, _p, _p, _i, _i,
                               _p, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "flm_replay_sample": [_i, _i, _u32, _u32, _p, _p],
    "flm_replay_gather": [_p, _p, _p, _p, _p, _p, _i, C.c_long, _p, _p, _p, _p, _p, _p],
    "flm_replay_draw_gather": [_p, _p, _p, _p, _p, _i, _u32, _u32, _i, C.c_long, _i, C.c_long, C.c_long,
                               _p, _p, _p, _p, _p, _p, _p],
    "flm_replay_gather_planes": [_p, _p, _p, _p, _p, _p, _i, C.c_long, _i, C.c_long, C.c_long, _p, _p, _p, _p, _p, _p],
    "flm_dqn_adam": [_p, _p, _p, _p, C.c_long, C.c_double, C.c_double, C.c_double, C.c_double, _i, _p],
    "flm_dqn_soft_update": [_p, _p, C.c_long, C.c_double, _p],
    "flm_dqn_adam_soft_update": [_p, _p, _p, _p, _p, C.c_long, C.c_double, C.c_double, C.c_double, C.c_double, _i,
                                 C.c_double, _p],
}

_LONG_RESULT = {"flm_paths_columns_lds_bytes", "flm_paths_columns_workspace", "flm_dqn_param_count", "flm_dqn_grad_split", "flm_dqn_learn_rails", "flm_dqn_learn_rails_rcb_offset", "flm_replay_store_slots_workspace", "flm_dqn_forward_workspace", "flm_dqn_learn_workspace",
                "flm_dqn_learn_workspace_u8"}
_lib = None


class FlmError(RuntimeError):
    pass


def exported_symbols():
    return sorted(_SIGNATURES)


def load():
    """Load libflmarl_hip.so (built by __graft_entry__.build()); raises if it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FlmError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`"
                           % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, args in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = C.c_long if name in _LONG_RESULT else C.c_int
        if lib.flm_abi_version() != ABI_VERSION:
            raise FlmError("libflmarl_hip.so ABI %d, expected %d" % (lib.flm_abi_version(), ABI_VERSION))
        _lib = lib
    return _lib


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous()
    return t.data_ptr()


def query(name, *args):
    """Size queries (return a count, not a status)."""
    return int(getattr(load(), name)(*args))


def call(name, *args):
    rc = getattr(load(), name)(*args)
    if rc != FLM_OK:
        raise FlmError("%s failed with code %d" % (name, rc))
