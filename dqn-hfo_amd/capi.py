"""ctypes declarations of include/dqnhip.h (one entry per exported symbol)."""
import ctypes as C
import os

from ._build import LIB, build

MAX_HIDDEN = 8
DP_ID_BYTES = 128        # DQNHIP_DP_ID_BYTES
DP_PER_LAYER, DP_HALF_GRADS, DP_SHARD_OPT = 1, 2, 4    # dqnhip_dp_init flags
DP_UNVERIFIED_OK = 256   # lets DP_PER_LAYER / DP_SHARD_OPT through for dp_world > 1 (never run on real links)
TUNE_FP16_WGRAD_PER_LAYER = 1                       # dqnhip_config.tuning_flags bits
TUNE_SEPARATE_HEAD_SEED = 2
TUNE_BWD_UNSHIFTED = 4
TUNE_SEPARATE_ACTOR_HEAD_BWD = 8
TUNE_SEPARATE_Q_TRAIN = 16
TUNE_SEPARATE_FIRST_LAYER = 32
TUNE_SEPARATE_CRITIC_FIRST_LAYERS = 64
TUNE_LATE_GATHER = 128
# dqnhip_update_plan.forms bits (DQNHIP_PLAN_*), in bit order
PLAN_FORMS = ("fp16", "data_parallel", "bwd_shifted_critic", "bwd_shifted_actor", "head_wgrad_rides_critic", "head_wgrad_rides_actor",
              "q_train_in_dgrad", "head_seed_fused", "dqda_head_bwd", "critic_l0_rides", "first_layers_merged", "early_gather_l0", "dp_tails_ride")
ACTOR, CRITIC, ACTOR_TARGET, CRITIC_TARGET = 0, 1, 2, 3
KIND_W, KIND_M, KIND_V, KIND_G = 0, 1, 2, 3


class Config(C.Structure):
    """struct dqnhip_config (include/dqnhip.h)."""
    _fields_ = [
        ("struct_size", C.c_int32), ("minibatch", C.c_int32), ("state_size", C.c_int32),
        ("num_hidden", C.c_int32), ("hidden", C.c_int32 * MAX_HIDDEN),
        ("replay_capacity", C.c_int32), ("soft_update_freq", C.c_int32),
        ("gamma", C.c_double), ("beta", C.c_double), ("tau", C.c_double),
        ("actor_lr", C.c_float), ("critic_lr", C.c_float), ("momentum", C.c_float),
        ("momentum2", C.c_float), ("delta", C.c_float), ("clip_gradients", C.c_float),
        ("device", C.c_int32), ("dp_world", C.c_int32), ("dp_rank", C.c_int32), ("use_graph", C.c_int32),
        ("seed", C.c_uint64), ("stream", C.c_void_p), ("grad_arena", C.c_void_p),
        ("grad_arena_bytes", C.c_size_t), ("precision", C.c_int32), ("loss_scale", C.c_float),
        ("tuning_flags", C.c_int32),
    ]


class UpdatePlan(C.Structure):
    """struct dqnhip_update_plan (include/dqnhip.h)."""
    _fields_ = [("struct_size", C.c_int32), ("forms", C.c_int32), ("launches_single", C.c_int32), ("launches_graph_first", C.c_int32),
                ("launches_in_graph", C.c_int32), ("updates_per_graph", C.c_int32), ("collectives", C.c_int32), ("reserved", C.c_int32)]


fp = C.POINTER(C.c_float)
ip = C.POINTER(C.c_int32)
up = C.POINTER(C.c_uint8)
H = C.c_void_p

# name -> (restype, argtypes): every function include/dqnhip.h declares
SIGNATURES = {
    "dqnhip_default_config": (None, [C.POINTER(Config), C.c_int32]),
    "dqnhip_grad_arena_bytes": (C.c_size_t, [C.POINTER(Config)]),
    "dqnhip_last_error": (C.c_char_p, []),
    "dqnhip_create": (C.c_int, [C.POINTER(Config), C.POINTER(H)]),
    "dqnhip_destroy": (C.c_int, [H]),
    "dqnhip_update": (C.c_int, [H, ip, fp, fp]),
    "dqnhip_update_chained": (C.c_int, [H, ip, ip, fp, fp]),
    "dqnhip_update_async": (C.c_int, [H, ip]),
    "dqnhip_update_async_n": (C.c_int, [H, C.c_int32]),
    "dqnhip_update_pipelined": (C.c_int, [H, ip, fp, fp]),
    "dqnhip_benchmark_blocking": (C.c_int, [H, C.c_int32, C.c_int32, C.c_uint64, C.c_int32, fp]),
    "dqnhip_update_phase": (C.c_int, [H, C.c_int32, ip]),
    "dqnhip_update_abort": (C.c_int, [H]),
    "dqnhip_apply_update": (C.c_int, [H, C.c_int32]),
    "dqnhip_apply_update_sharded": (C.c_int, [H, C.c_int32, C.c_int32]),
    "dqnhip_get_update_plan": (C.c_int, [H, C.POINTER(UpdatePlan)]),
    "dqnhip_grad_buffer": (C.c_int, [H, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "dqnhip_read_stats": (C.c_int, [H, fp, fp]),
    "dqnhip_dp_unique_id": (C.c_int, [C.c_void_p, C.c_size_t]),
    "dqnhip_dp_init": (C.c_int, [H, C.c_void_p, C.c_size_t, C.c_int32]),
    "dqnhip_dp_init_file": (C.c_int, [H, C.c_char_p, C.c_int32, C.c_int32]),
    "dqnhip_dp_rendezvous_file": (C.c_int, [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t]),
    "dqnhip_dp_rendezvous_cleanup": (C.c_int, [C.c_char_p, C.c_int32]),
    "dqnhip_dp_graph_active": (C.c_int, [H, ip]),
    "dqnhip_dp_broadcast_params": (C.c_int, [H, C.c_int32]),
    "dqnhip_dp_update": (C.c_int, [H, ip]),
    "dqnhip_dp_update_n": (C.c_int, [H, C.c_int32]),
    "dqnhip_dp_gather_state": (C.c_int, [H]),
    "dqnhip_dp_destroy": (C.c_int, [H]),
    "dqnhip_dp_info": (C.c_int, [ip, C.c_char_p, C.c_size_t]),
    "dqnhip_skipped_steps": (C.c_int, [H, C.POINTER(C.c_int64)]),
    "dqnhip_reduce_gradients_local": (C.c_int, [C.POINTER(H), C.c_int32, C.c_int32]),
    "dqnhip_sample_states": (C.c_int, [H, ip, C.c_int32, fp]),
    "dqnhip_get_actor_output": (C.c_int, [H, C.c_int32, C.c_int32, fp]),
    "dqnhip_files_matching_regexp": (C.c_int, [C.c_char_p, C.c_char_p, C.c_size_t, ip]),
    "dqnhip_remove_snapshots": (C.c_int, [C.c_char_p, C.c_int32]),
    "dqnhip_benchmark": (C.c_int, [H, C.c_int32, C.c_int32, fp]),
    "dqnhip_select_actions": (C.c_int, [H, fp, C.c_int32, fp]),
    "dqnhip_select_actions_device": (C.c_int, [H, C.c_void_p, C.c_int32, C.c_void_p]),
    "dqnhip_select_actions_net": (C.c_int, [H, C.c_int32, fp, C.c_int32, fp]),
    "dqnhip_critic_forward": (C.c_int, [H, C.c_int32, fp, fp, C.c_int32, fp]),
    "dqnhip_add_transitions": (C.c_int, [H, fp, fp, fp, fp, fp, up, C.c_int32]),
    "dqnhip_add_transition": (C.c_int, [H, fp, fp, C.c_float, C.c_float, fp, C.c_uint8]),
    "dqnhip_add_transitions_device": (C.c_int, [H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_void_p, C.c_int32]),
    "dqnhip_label_transitions": (C.c_int, [C.c_double, fp, C.c_int32, fp]),
    "dqnhip_memory_size": (C.c_int, [H, ip]),
    "dqnhip_clear_memory": (C.c_int, [H]),
    "dqnhip_read_memory": (C.c_int, [H, C.c_int32, C.c_int32, fp, fp, fp, fp, fp, up]),
    "dqnhip_snapshot_replay_memory": (C.c_int, [H, C.c_char_p]),
    "dqnhip_load_replay_memory": (C.c_int, [H, C.c_char_p]),
    "dqnhip_get_config": (C.c_int, [H, C.POINTER(Config)]),
    "dqnhip_save_caffemodel": (C.c_int, [H, C.c_int32, C.c_char_p]),
    "dqnhip_load_caffemodel": (C.c_int, [H, C.c_int32, C.c_char_p]),
    "dqnhip_solver_snapshot": (C.c_int, [H, C.c_int32, C.c_char_p, ip]),
    "dqnhip_solver_restore": (C.c_int, [H, C.c_int32, C.c_char_p]),
    "dqnhip_snapshot": (C.c_int, [H, C.c_char_p, C.c_char_p, C.c_int32, C.c_int32]),
    "dqnhip_find_latest_snapshot": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]),
    "dqnhip_find_hiscore": (C.c_int, [C.c_char_p, ip]),
    "dqnhip_remove_files_matching_regexp": (C.c_int, [C.c_char_p]),
    "dqnhip_param_count": (C.c_int, [H, C.c_int32, C.POINTER(C.c_size_t)]),
    "dqnhip_get_params": (C.c_int, [H, C.c_int32, C.c_int32, fp, C.c_size_t]),
    "dqnhip_set_params": (C.c_int, [H, C.c_int32, C.c_int32, fp, C.c_size_t]),
    "dqnhip_clone_to_target": (C.c_int, [H, C.c_int32]),
    "dqnhip_share_parameters": (C.c_int, [H, H, C.c_int32, C.c_int32]),
    "dqnhip_share_replay_memory": (C.c_int, [H, H]),
    "dqnhip_get_iters": (C.c_int, [H, ip, ip]),
    "dqnhip_set_iters": (C.c_int, [H, C.c_int32, C.c_int32]),
    "dqnhip_debug_read": (C.c_int, [H, C.c_char_p, fp, C.c_size_t]),
    "dqnhip_get_stream": (C.c_int, [H, C.POINTER(C.c_void_p)]),
    "dqnhip_set_kernel_timing": (C.c_int, [H, C.c_int32]),
    "dqnhip_get_kernel_timing": (C.c_int, [H, C.c_char_p, fp, C.POINTER(C.c_int64), C.c_int32]),
}

class EnvConfig(C.Structure):
    """struct dqnhip_env_config (include/dqnhip_env.h)."""
    _fields_ = [("struct_size", C.c_int32), ("workers", C.c_int32), ("max_steps", C.c_int32), ("unum", C.c_int32),
                ("p_end", C.c_float), ("p_goal", C.c_float), ("seed", C.c_uint64)]


SIGNATURES.update({
    "dqnhip_env_create": (C.c_int, [H, C.POINTER(EnvConfig), C.POINTER(C.c_void_p)]),
    "dqnhip_env_destroy": (C.c_int, [C.c_void_p]),
    "dqnhip_env_step": (C.c_int, [C.c_void_p, C.c_float, C.c_int32]),
    "dqnhip_env_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                   C.POINTER(C.c_int64)]),
    "dqnhip_env_debug_read": (C.c_int, [C.c_void_p, C.c_char_p, fp, C.c_size_t]),
})

_lib = None


def load(rebuild=False):
    """dlopen the in-tree libdqnhip.so (building it first if needed).  There is no
    fallback: a missing or unloadable HIP library is a hard error."""
    global _lib
    if _lib is None or rebuild:
        build()
        if not os.path.exists(LIB):
            raise RuntimeError("libdqnhip.so is missing: the HIP extension is mandatory")
        lib = C.CDLL(LIB)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib
