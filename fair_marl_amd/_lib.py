"""ctypes binding of ``libfmarl.so`` (declarations: ``include/fmarl.h``).  Fails loudly if the
HIP library has not been built -- there is deliberately no fallback implementation."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# FMARL_LIB: measurement aid (tools/mkvariant.sh, tools/ab_lib.sh) -- another build of the SAME sources / C-ABI, e.g. a
# -DFMARL_MEASURE build or an experiment's variant, selected per process instead of overwriting the shipped library
LIB_PATH = os.environ.get('FMARL_LIB') or os.path.join(HERE, 'csrc', 'libfmarl.so')

INFO_WIDTH = 14
(F_AGENT_POS, F_AGENT_VEL, F_P_DIST, F_LANDMARK_POS, F_OBSTACLE_POS, F_WALL_AXIS, F_WALL_E0, F_WALL_E1,
 F_WALL_ORIENT, F_WALL_LENGTH, F_GOAL_MATCH, F_DISTS_TO_GOAL, F_TIMES_REQUIRED, F_DIST_LEFT,
 F_NUM_OBST_COLL, F_NUM_AGENT_COLL, F_MIN_TIME, F_CUR_STEP, F_EPISODE, F_SLOT_POS, F_SLOT_OCC,
 F_SLOT_DELTA, F_FORMATION_DONE, F_GOAL_OCC, F_GOAL_HISTORY, F_GOAL_REACHED, F_STATUS, F_RESET_FLAG, F_STAGE_AGENT_POS, F_STAGE_LANDMARK_POS, F_STAGE_OBSTACLE_POS,
 F_STAGE_WALL_AXIS, F_STAGE_WALL_ORIENT, F_STAGE_GOAL_MATCH, F_STAGE_VALID, F_STAGE_NEED, F_PLACE_FAILS, F_STAGE_PLACE_FAILS, F_MATCH_DUAL, F_ROT_TABLE, NUM_FIELDS) = range(41)
FLAG_ASYNC_RESET = 1
FLAG_GLOBAL_FEATURES = 2
FIELD_NAMES = ('agent_pos', 'agent_vel', 'p_dist', 'landmark_pos', 'obstacle_pos', 'wall_axis', 'wall_e0',
               'wall_e1', 'wall_orient', 'wall_length', 'goal_match', 'dists_to_goal', 'times_required',
               'dist_left', 'num_obst_coll', 'num_agent_coll', 'min_time', 'cur_step', 'episode',
               'slot_pos', 'slot_occ', 'slot_delta', 'formation_done', 'goal_occ', 'goal_history', 'goal_reached', 'status',
               'reset_flag', 'stage_agent_pos',
               'stage_landmark_pos', 'stage_obstacle_pos', 'stage_wall_axis', 'stage_wall_orient', 'stage_goal_match',
               'stage_valid', 'stage_need', 'place_fails', 'stage_place_fails', 'internal_match_dual', 'internal_rot_table')
DTYPE_F64, DTYPE_I32, DTYPE_I8 = 0, 1, 2
DTYPE_BYTES = {DTYPE_F64: 8, DTYPE_I32: 4, DTYPE_I8: 1}
SCENARIOS = {'navigation_graph': 0, 'fair_graph_formation': 1, 'nav_fairassign_fairrew_formation_graph': 2}


class FmarlConfig(C.Structure):
    _fields_ = [('scenario', C.c_int32), ('n_envs', C.c_int32), ('num_agents', C.c_int32),
                ('num_landmarks', C.c_int32), ('num_obstacles', C.c_int32), ('num_walls', C.c_int32),
                ('episode_length', C.c_int32), ('has_max_speed', C.c_int32), ('env_offset', C.c_int32),
                ('flags', C.c_int32), ('world_size', C.c_double), ('max_speed', C.c_double),
                ('collision_rew', C.c_double), ('goal_rew', C.c_double), ('min_dist_thresh', C.c_double),
                ('fair_rew', C.c_double), ('zeroshift', C.c_double), ('max_edge_dist', C.c_double), ('min_obs_dist', C.c_double),
                ('seed', C.c_uint64), ('envs_per_workgroup', C.c_int32), ('reserved0', C.c_int32)]


class FmarlSpan(C.Structure):
    _fields_ = [(k, C.c_int64) for k in ('obs', 'node_obs', 'adj', 'reward', 'done', 'info', 'edge_nnz', 'graph_record', 'actions')]


class FmarlOutputs(C.Structure):
    _fields_ = [('obs', C.c_void_p), ('node_obs', C.c_void_p), ('adj', C.c_void_p), ('reward', C.c_void_p),
                ('done', C.c_void_p), ('info', C.c_void_p), ('edge_nnz', C.c_void_p), ('graph_record', C.c_void_p)]


class FmarlReturns(C.Structure):
    _fields_ = [('gamma', C.c_double), ('gae_lambda', C.c_double), ('mean', C.c_float), ('stddev', C.c_float),
                ('denormalize', C.c_int32), ('use_gae', C.c_int32), ('use_proper_time_limits', C.c_int32), ('T', C.c_int32),
                ('columns', C.c_int64)]


BATCH_SRC_ARRAYS = ('obs', 'node_obs', 'adj_env', 'rnn_states', 'rnn_states_critic', 'actions', 'action_log_probs', 'value_preds',
                    'returns', 'masks', 'active_masks', 'advantages', 'available_actions')
# the 16 arrays of a generator's tuple, in the reference's order (graph_buffer.py:448-453), + env_slot
BATCH_DST_ARRAYS = ('share_obs', 'obs', 'node_obs', 'adj', 'agent_id', 'share_agent_id', 'rnn_states', 'rnn_states_critic', 'actions',
                    'value_preds', 'returns', 'masks', 'active_masks', 'old_action_log_probs', 'adv_targ', 'available_actions',
                    'env_slot')


class FmarlBatchSrc(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in BATCH_SRC_ARRAYS] + \
               [(k, C.c_int32) for k in ('T', 'n', 'N', 'D', 'E', 'F', 'rnn_elems', 'act_dim', 'avail_dim', 'reserved0')]


class FmarlBatchDst(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ('share_obs', 'obs', 'node_obs', 'adj', 'agent_id', 'share_agent_id', 'rnn_states',
                                          'rnn_states_critic', 'actions', 'value_preds', 'returns', 'masks', 'active_masks',
                                          'old_action_log_probs', 'adv_targ', 'available_actions', 'env_slot')]


_SIGS = {
    'fmarl_create': (C.c_int, [C.POINTER(FmarlConfig), C.POINTER(C.c_void_p)]),
    'fmarl_destroy': (C.c_int, [C.c_void_p]),
    'fmarl_last_error': (C.c_char_p, []),
    'fmarl_envs_per_workgroup': (C.c_int, [C.c_void_p]),
    'fmarl_state_bytes': (C.c_size_t, [C.POINTER(FmarlConfig)]),
    'fmarl_state_field': (C.c_int, [C.POINTER(FmarlConfig), C.c_int, C.POINTER(C.c_size_t),
                                    C.POINTER(C.c_size_t), C.POINTER(C.c_int)]),
    'fmarl_init_state': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    'fmarl_reset': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(FmarlOutputs), C.c_void_p]),
    'fmarl_step': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(FmarlOutputs),
                             C.c_int, C.c_void_p]),
    'fmarl_step_span': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(FmarlOutputs), C.POINTER(FmarlSpan), C.c_void_p]),
    'fmarl_get_phase': (C.c_int, [C.c_void_p]),
    'fmarl_set_phase': (C.c_int, [C.c_void_p, C.c_int]),
    'fmarl_state_changed': (C.c_int, [C.c_void_p]),
    'fmarl_get_state': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'fmarl_set_state': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'fmarl_profile_enable': (C.c_int, [C.c_void_p, C.c_int]),
    'fmarl_launch_counts': (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    'fmarl_launch_geometry': (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    'fmarl_poison_lds': (C.c_int, [C.c_void_p, C.c_void_p]),
    'fmarl_insert_masks': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    'fmarl_store_pattern': (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_void_p]),
    'fmarl_ring_alloc': (C.c_int, [C.c_size_t, C.c_int, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    'fmarl_ring_free': (C.c_int, [C.c_void_p]),
    'fmarl_ring_stats': (C.c_int, [C.POINTER(C.c_uint64)]),
    'fmarl_store_stream': (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_void_p]),
    'fmarl_profile_read': (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]),
    'fmarl_cost_matrix': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'fmarl_lexifair': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    'fmarl_update_graph': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                     C.c_double, C.c_void_p]),
    'fmarl_update_graph_state': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'fmarl_info_means': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p]),
    'fmarl_edge_count': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p]),
    'fmarl_edge_fill': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int,
                                  C.c_double, C.c_int, C.c_void_p]),
    'fmarl_edge_offsets': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'fmarl_edge_fill_state': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    'fmarl_episode_record_words': (C.c_size_t, [C.POINTER(FmarlConfig)]),
    'fmarl_step_record_words': (C.c_size_t, [C.POINTER(FmarlConfig)]),
    'fmarl_rebuild_graph_rec': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'fmarl_episode_started': (C.c_int, [C.c_void_p]),
    'fmarl_pack_episode': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'fmarl_rebuild_graph': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'fmarl_compute_returns': (C.c_int, [C.POINTER(FmarlReturns), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'fmarl_advantages_workspace': (C.c_size_t, []),
    'fmarl_advantages': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    'fmarl_advantages_sums': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    'fmarl_advantages_apply': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    'fmarl_minibatch_gather': (C.c_int, [C.POINTER(FmarlBatchSrc), C.POINTER(FmarlBatchDst), C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
}
EXPORTS = tuple(_SIGS)
_lib = None


def load():
    """Load libfmarl.so and attach the prototypes of include/fmarl.h."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                'fair_marl_amd: %s is missing -- build it first (python -c "import __graft_entry__ as g; '
                'g.build()").  There is no CPU fallback for the rollout path.' % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)  # AttributeError if the library does not export it
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError('%s failed (%d): %s' % (what, rc, load().fmarl_last_error().decode()))
