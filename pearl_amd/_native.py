"""ctypes binding of libpearl_amd.so (the C ABI declared in include/pearl_amd.h).

The library is the product: there is NO CPU or PyTorch fallback.  Loading fails
loudly if the shared object is missing, and every compute entry point fails
loudly (``NativeError``) when no HIP device is visible.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpearl_amd.so")

# pa_status
PA_OK = 0
PA_ERR_INVALID = -1
PA_ERR_VALUE = -2
PA_ERR_HIP = -3
PA_ERR_UNSUPPORTED = -4
PA_ERR_NOMEM = -5

# pa_dtype
PA_F32, PA_I64, PA_I32, PA_U8, PA_F64 = 0, 1, 2, 3, 4

_TORCH_TO_PA = {
    torch.float32: PA_F32,
    torch.int64: PA_I64,
    torch.int32: PA_I32,
    torch.uint8: PA_U8,
    torch.bool: PA_U8,
    torch.float64: PA_F64,
}
_PA_TO_TORCH = {PA_F32: torch.float32, PA_I64: torch.int64, PA_I32: torch.int32,
                PA_U8: torch.uint8, PA_F64: torch.float64}


class NativeError(RuntimeError):
    """libpearl_amd reported a failure that has no reference-side exception type."""


def pa_dtype_of(dt: torch.dtype) -> int:
    try:
        return _TORCH_TO_PA[dt]
    except KeyError:
        raise TypeError(f"pearl_amd: unsupported dtype {dt}") from None


def torch_dtype_of(pa: int) -> torch.dtype:
    return _PA_TO_TORCH[pa]


class ArenaDesc(C.Structure):
    _fields_ = [
        ("capacity", C.c_int64),
        ("device", C.c_int32),
        ("state_dim", C.c_int32),
        ("action_elems", C.c_int32),
        ("action_dtype", C.c_int32),
        ("reward_dtype", C.c_int32),
        ("max_actions", C.c_int32),
        ("avail_dim", C.c_int32),
        ("has_next_state", C.c_int32),
        ("has_cost", C.c_int32),
        ("staging_rows", C.c_int64),
    ]


class Transition(C.Structure):
    _fields_ = [
        ("state", C.c_void_p),
        ("action", C.c_void_p),
        ("reward", C.c_void_p),
        ("next_state", C.c_void_p),
        ("curr_avail", C.c_void_p),
        ("curr_mask", C.c_void_p),
        ("next_avail", C.c_void_p),
        ("next_mask", C.c_void_p),
        ("cost", C.c_void_p),
        ("terminated", C.c_uint8),
        ("truncated", C.c_uint8),
    ]


class Columns(C.Structure):
    _fields_ = [
        ("state", C.c_void_p),
        ("action", C.c_void_p),
        ("reward", C.c_void_p),
        ("terminated", C.c_void_p),
        ("truncated", C.c_void_p),
        ("next_state", C.c_void_p),
        ("curr_avail", C.c_void_p),
        ("curr_mask", C.c_void_p),
        ("next_avail", C.c_void_p),
        ("next_mask", C.c_void_p),
        ("cost", C.c_void_p),
        ("avail_bcast", C.c_int32),
    ]


class BatchOut(C.Structure):
    _fields_ = [
        ("state", C.c_void_p),
        ("action", C.c_void_p),
        ("reward", C.c_void_p),
        ("terminated", C.c_void_p),
        ("truncated", C.c_void_p),
        ("next_state", C.c_void_p),
        ("curr_avail", C.c_void_p),
        ("curr_mask", C.c_void_p),
        ("next_avail", C.c_void_p),
        ("next_mask", C.c_void_p),
        ("cost", C.c_void_p),
        ("x", C.c_void_p),
        ("next_avail_rep", C.c_void_p),
        ("reward_f32", C.c_void_p),
        ("rep_dim", C.c_int32),
        ("rep_onehot", C.c_int32),
        ("curr_avail_rep", C.c_void_p),
    ]


class DqnDesc(C.Structure):
    _fields_ = [
        ("device", C.c_int32),
        ("state_dim", C.c_int32),
        ("action_dim", C.c_int32),
        ("hidden1", C.c_int32),
        ("hidden2", C.c_int32),
        ("max_batch", C.c_int32),
        ("max_actions", C.c_int32),
        ("discount", C.c_float),
        ("tau", C.c_float),
        ("lr", C.c_double),
        ("beta1", C.c_double),
        ("beta2", C.c_double),
        ("eps", C.c_double),
        ("weight_decay", C.c_double),
        ("amsgrad", C.c_int32),
        ("double_q", C.c_int32),
    ]


class DqnBuffers(C.Structure):
    _fields_ = [
        ("q", C.c_void_p),
        ("q_target", C.c_void_p),
        ("grad", C.c_void_p),
        ("exp_avg", C.c_void_p),
        ("exp_avg_sq", C.c_void_p),
        ("max_exp_avg_sq", C.c_void_p),
    ]


class DqnBatch(C.Structure):
    _fields_ = [
        ("B", C.c_int32),
        ("A", C.c_int32),
        ("x", C.c_void_p),
        ("state", C.c_void_p),
        ("action_rep", C.c_void_p),
        ("reward", C.c_void_p),
        ("terminated", C.c_void_p),
        ("next_state", C.c_void_p),
        ("next_avail_rep", C.c_void_p),
        ("next_mask", C.c_void_p),
        ("next_avail_bcast", C.c_int32),
        ("next_action_rep", C.c_void_p),
    ]


MLP_MAX_LAYERS = 8


class MlpDesc(C.Structure):
    _fields_ = [
        ("device", C.c_int32),
        ("n_layers", C.c_int32),
        ("dims", C.c_int32 * (MLP_MAX_LAYERS + 1)),
        ("max_batch", C.c_int32),
        ("lr", C.c_double),
        ("beta1", C.c_double),
        ("beta2", C.c_double),
        ("eps", C.c_double),
        ("weight_decay", C.c_double),
        ("amsgrad", C.c_int32),
        ("no_last_bias", C.c_int32),
        ("identity_layers", C.c_int32),
        ("hidden_act", C.c_int32),
        ("layer_norm", C.c_int32),
        ("batch_norm", C.c_int32),
        ("dropout", C.c_int32),
        ("residual", C.c_int32),
    ]


class DsacStepArgs(C.Structure):
    _fields_ = [
        ("actor", C.c_void_p), ("critic1", C.c_void_p), ("critic2", C.c_void_p),
        ("B", C.c_int32), ("S", C.c_int32), ("A", C.c_int32), ("AD", C.c_int32),
        ("state", C.c_void_p), ("ld_state", C.c_int32),
        ("next_state", C.c_void_p), ("ld_next_state", C.c_int32),
        ("xq", C.c_void_p), ("ld_xq", C.c_int32),
        ("reward", C.c_void_p), ("terminated", C.c_void_p),
        ("curr_rep", C.c_void_p), ("curr_rep_bstride", C.c_int64),
        ("next_rep", C.c_void_p), ("next_rep_bstride", C.c_int64),
        ("curr_mask", C.c_void_p), ("next_mask", C.c_void_p),
        ("gamma", C.c_float), ("tau", C.c_float),
        ("alpha", C.c_void_p),
        ("log_alpha", C.c_void_p), ("alpha_m", C.c_void_p), ("alpha_v", C.c_void_p),
        ("target_entropy", C.c_float),
        ("alpha_lr", C.c_double), ("alpha_beta1", C.c_double), ("alpha_beta2", C.c_double),
        ("alpha_eps", C.c_double), ("alpha_weight_decay", C.c_double),
        ("alpha_step", C.c_int64), ("actor_step", C.c_int64), ("critic_step", C.c_int64),
        ("scratch", C.c_void_p), ("losses", C.c_void_p),
        ("h_out", C.c_void_p),
    ]


class IqlStepArgs(C.Structure):
    _fields_ = [
        ("actor", C.c_void_p), ("value", C.c_void_p), ("critic1", C.c_void_p), ("critic2", C.c_void_p),
        ("B", C.c_int32), ("S", C.c_int32), ("A", C.c_int32), ("actor_kind", C.c_int32),
        ("state", C.c_void_p), ("ld_state", C.c_int32),
        ("next_state", C.c_void_p), ("ld_next_state", C.c_int32),
        ("action", C.c_void_p), ("ld_action", C.c_int32),
        ("xq", C.c_void_p), ("ld_xq", C.c_int32),
        ("reward", C.c_void_p), ("terminated", C.c_void_p),
        ("low", C.c_void_p), ("high", C.c_void_p),
        ("pick_value", C.c_int32), ("pick_actor", C.c_int32),
        ("expectile", C.c_float), ("temperature", C.c_float), ("adv_clamp", C.c_float),
        ("gamma", C.c_float), ("tau", C.c_float),
        ("actor_step", C.c_int64), ("value_step", C.c_int64), ("critic_step", C.c_int64),
        ("zeros", C.c_void_p), ("scratch", C.c_void_p), ("losses", C.c_void_p),
    ]


class PpoLearnArgs(C.Structure):
    _fields_ = [
        ("actor", C.c_void_p), ("critic", C.c_void_p),
        ("B", C.c_int32), ("S", C.c_int32), ("A", C.c_int32),
        ("rounds", C.c_int32), ("gather_rounds", C.c_int32),
        ("idx_lists", C.c_void_p),
        ("planes", C.c_void_p), ("plane_stride", C.c_int64),
        ("x", C.c_void_p), ("planes_ws", C.c_void_p),
        ("epsilon", C.c_float), ("entropy_scale", C.c_float), ("value_grad_scale", C.c_float),
        ("d_logits", C.c_void_p), ("d_value", C.c_void_p),
        ("losses", C.c_void_p), ("losses_stride", C.c_int64),
        ("actor_step", C.c_int64), ("critic_step", C.c_int64),
    ]


class BanditStepArgs(C.Structure):
    _fields_ = [
        ("net", C.c_void_p),
        ("x", C.c_void_p), ("ldx", C.c_int32), ("B", C.c_int32),
        ("y", C.c_void_p),
        ("loss_kind", C.c_int32), ("out_act", C.c_int32),
        ("adam_step", C.c_int64),
        ("pred", C.c_void_p), ("d_pred", C.c_void_p), ("scalars", C.c_void_p),
        ("d", C.c_int32),
        ("x_scratch", C.c_void_p), ("r_scratch", C.c_void_p), ("delta", C.c_void_p),
        ("A", C.c_void_p), ("b", C.c_void_p), ("sum_weight", C.c_void_p),
        ("A_snap", C.c_void_p), ("b_snap", C.c_void_p),
        ("side_stream", C.c_void_p), ("ev_slot_free", C.c_void_p), ("ev_ready", C.c_void_p),
        ("ev_done", C.c_void_p),
        ("l2_reg_lambda", C.c_float), ("work", C.c_void_p), ("inv_A", C.c_void_p),
        ("coefs", C.c_void_p), ("singular", C.c_void_p),
    ]


class MlpBuffers(C.Structure):
    _fields_ = [
        ("p", C.c_void_p),
        ("p_target", C.c_void_p),
        ("grad", C.c_void_p),
        ("exp_avg", C.c_void_p),
        ("exp_avg_sq", C.c_void_p),
        ("max_exp_avg_sq", C.c_void_p),
    ]


class SacStepArgs(C.Structure):
    """pa_sac_step_args (include/pearl_amd.h)."""
    _fields_ = [
        ("actor", C.c_void_p), ("critic1", C.c_void_p), ("critic2", C.c_void_p),
        ("state", C.c_void_p), ("ld_state", C.c_int32),
        ("action", C.c_void_p), ("ld_action", C.c_int32),
        ("reward", C.c_void_p),
        ("terminated", C.c_void_p),
        ("next_state", C.c_void_p), ("ld_next_state", C.c_int32),
        ("noise_actor", C.c_void_p),
        ("noise_critic", C.c_void_p),
        ("low", C.c_void_p), ("high", C.c_void_p),
        ("alpha", C.c_void_p),
        ("log_alpha", C.c_void_p),
        ("alpha_m", C.c_void_p), ("alpha_v", C.c_void_p), ("alpha_vmax", C.c_void_p),
        ("target_entropy", C.c_float),
        ("alpha_lr", C.c_double), ("alpha_beta1", C.c_double), ("alpha_beta2", C.c_double),
        ("alpha_eps", C.c_double), ("alpha_weight_decay", C.c_double),
        ("alpha_amsgrad", C.c_int32),
        ("alpha_step", C.c_int64),
        ("B", C.c_int32), ("S", C.c_int32), ("A", C.c_int32),
        ("gamma", C.c_float), ("tau", C.c_float),
        ("actor_step", C.c_int64), ("critic_step", C.c_int64),
        ("scratch", C.c_void_p),
        ("losses", C.c_void_p),
        ("log_prob_out", C.c_void_p),
    ]


class DdpgStepArgs(C.Structure):
    """pa_ddpg_step_args (include/pearl_amd.h)."""
    _fields_ = [
        ("actor", C.c_void_p), ("critic1", C.c_void_p), ("critic2", C.c_void_p),
        ("state", C.c_void_p), ("ld_state", C.c_int32),
        ("action", C.c_void_p), ("ld_action", C.c_int32),
        ("reward", C.c_void_p),
        ("terminated", C.c_void_p),
        ("next_state", C.c_void_p), ("ld_next_state", C.c_int32),
        ("target_noise", C.c_void_p),
        ("noise_clip", C.c_float),
        ("low", C.c_void_p), ("high", C.c_void_p),
        ("zeros", C.c_void_p),
        ("B", C.c_int32), ("S", C.c_int32), ("A", C.c_int32),
        ("gamma", C.c_float),
        ("do_actor", C.c_int32), ("do_targets", C.c_int32),
        ("critic_tau", C.c_float), ("actor_tau", C.c_float),
        ("actor_step", C.c_int64), ("critic_step", C.c_int64),
        ("scratch", C.c_void_p),
        ("losses", C.c_void_p),
    ]


class AcLoopArgs(C.Structure):
    """pa_ac_loop_args (include/pearl_amd.h)."""
    _fields_ = [
        ("rounds", C.c_int32),
        ("idx_lists", C.c_void_p),
        ("batch", BatchOut),
        ("noise", C.c_void_p),
        ("noise_stride", C.c_int64),
        ("losses", C.c_void_p),
        ("losses_stride", C.c_int32),
        ("actor_update_freq", C.c_int32),
        ("training_step0", C.c_int64),
        ("gather_rounds", C.c_int32),
    ]


class LearnArgs(C.Structure):
    _fields_ = [
        ("rounds", C.c_int32),
        ("batch_size", C.c_int32),
        ("rep_onehot", C.c_int32),
        ("target_update_freq", C.c_int32),
        ("training_steps0", C.c_int64),
        ("adam_step0", C.c_int64),
        ("seed", C.c_uint64),
        ("offset0", C.c_uint64),
        ("losses_out", C.c_void_p),
        ("idx_host", C.c_void_p),
        ("grad_world", C.c_int32),
        ("allreduce_start", C.c_void_p),
        ("allreduce_wait", C.c_void_p),
        ("allreduce_ctx", C.c_void_p),
    ]


ALLREDUCE_START_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
ALLREDUCE_WAIT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)


# name -> (restype, argtypes); also the export list the CPU test suite checks
# against include/pearl_amd.h.
_P = C.c_void_p
SIGNATURES = {
    "pa_last_error": (C.c_char_p, []),
    "pa_abi_version": (C.c_int, []),
    "pa_device_count": (C.c_int, []),
    "pa_arena_create": (C.c_int, [C.POINTER(_P), C.POINTER(ArenaDesc)]),
    "pa_arena_destroy": (C.c_int, [_P]),
    "pa_arena_push": (C.c_int, [_P, C.POINTER(Transition)]),
    "pa_arena_push_many": (C.c_int, [_P, C.c_int64, C.POINTER(Columns)]),
    "pa_arena_push_many_device": (C.c_int, [_P, C.c_int64, C.POINTER(Columns), _P]),
    "pa_arena_flush": (C.c_int, [_P, _P]),
    "pa_arena_len": (C.c_int64, [_P]),
    "pa_arena_capacity": (C.c_int64, [_P]),
    "pa_arena_head": (C.c_int64, [_P]),
    "pa_arena_clear": (C.c_int, [_P]),
    "pa_arena_shared_next_table": (C.c_int32, [_P]),
    "pa_arena_gather": (C.c_int, [_P, _P, C.c_int32, C.POINTER(BatchOut), _P, _P]),
    "pa_arena_gather_device": (C.c_int, [_P, _P, C.c_int32, C.POINTER(BatchOut), _P]),
    "pa_arena_sample": (C.c_int, [_P, C.c_uint64, C.c_uint64, C.c_int32, C.POINTER(BatchOut), _P, _P]),
    "pa_sample_indices": (C.c_int, [C.c_int64, C.c_uint64, C.c_uint64, C.c_int32, _P, C.c_int32, _P]),
    "pa_sample_indices_rounds": (C.c_int, [C.c_int64, C.c_uint64, C.c_uint64, C.c_int32, C.c_int32,
                                           _P, C.c_int32, _P]),
    "pa_gather_planes": (C.c_int, [_P, C.c_int64, C.c_int32, _P, C.c_int32, _P, _P]),
    "pa_one_hot": (C.c_int, [_P, C.c_int32, C.c_int64, C.c_int32, _P, _P]),
    "pa_dqn_param_count": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "pa_dqn_param_offsets": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int64)]),
    "pa_dqn_create": (C.c_int, [C.POINTER(_P), C.POINTER(DqnDesc)]),
    "pa_dqn_destroy": (C.c_int, [_P]),
    "pa_dqn_bind": (C.c_int, [_P, C.POINTER(DqnBuffers)]),
    "pa_dqn_invalidate": (C.c_int, [_P]),
    "pa_dqn_qvalues": (C.c_int, [_P, C.POINTER(DqnBatch), _P, _P, _P, _P]),
    "pa_dqn_update_target": (C.c_int, [_P, _P]),
    "pa_dqn_step": (C.c_int, [_P, C.POINTER(DqnBatch), C.c_int32, C.c_int64, C.c_int32, _P, _P]),
    "pa_dqn_apply": (C.c_int, [_P, C.c_int64, _P]),
    "pa_dqn_learn": (C.c_int, [_P, _P, C.POINTER(LearnArgs), _P]),
    "pa_dqn_check": (C.c_int, [_P]),
    "pa_dqn_set_overlap": (C.c_int, [_P, C.c_int32]),
    "pa_comm_available": (C.c_int, []),
    "pa_comm_unique_id": (C.c_int, [_P]),
    "pa_comm_create": (C.c_int, [C.POINTER(_P), C.c_int32, C.c_int32, C.c_int32, _P]),
    "pa_comm_destroy": (C.c_int, [_P]),
    "pa_comm_info": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "pa_comm_create_p2p": (C.c_int, [C.POINTER(_P), C.c_int32, C.c_int32, C.c_int32, C.c_int64]),
    "pa_comm_p2p_handle": (C.c_int, [_P, _P]),
    "pa_comm_p2p_open": (C.c_int, [_P, C.c_int32, _P]),
    "pa_comm_p2p_check": (C.c_int, [_P]),
    "pa_comm_check": (C.c_int, [_P]),
    "pa_comm_max_floats": (C.c_int64, [_P]),
    "pa_comm_allreduce_start": (C.c_int, [_P, _P, C.c_int64, _P]),
    "pa_comm_allreduce_wait": (C.c_int, [_P, _P]),
    "pa_dqn_enable_timing": (C.c_int, [_P, C.c_int32]),
    "pa_dqn_get_timing": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "pa_dqn_get_timing_units": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_int64)]),
    "pa_gather_rows": (C.c_int, [_P, C.c_int32, _P, C.c_int32, _P, _P]),
    "pa_mlp_param_count": (C.c_int64, [C.POINTER(MlpDesc)]),
    "pa_mlp_param_offsets": (C.c_int, [C.POINTER(MlpDesc), C.POINTER(C.c_int64)]),
    "pa_mlp_norm_offsets": (C.c_int, [C.POINTER(MlpDesc), C.POINTER(C.c_int64)]),
    "pa_mlp_bn_offsets": (C.c_int, [C.POINTER(MlpDesc), C.POINTER(C.c_int64)]),
    "pa_mlp_bind_batch_norm": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, _P]),
    "pa_mlp_set_dropout": (C.c_int, [_P, C.c_int32, _P, C.c_int32]),
    "pa_mlp_create": (C.c_int, [C.POINTER(_P), C.POINTER(MlpDesc)]),
    "pa_mlp_destroy": (C.c_int, [_P]),
    "pa_mlp_bind": (C.c_int, [_P, C.POINTER(MlpBuffers)]),
    "pa_mlp_forward": (C.c_int, [_P, C.c_int32, _P, C.c_int32, C.c_int32, _P, C.c_int32, C.c_int32, _P]),
    "pa_mlp_copy_activation": (C.c_int, [_P, C.c_int32, C.c_int32, _P, C.c_int32, _P]),
    "pa_mlp_backward": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, C.c_int32, C.c_int32, _P,
                                  C.c_int32, _P]),
    "pa_mlp_q_all": (C.c_int, [_P, C.c_int32, _P, C.c_int32, _P, C.c_int64, C.c_int32, C.c_int32,
                               C.c_int32, _P, _P]),
    "pa_rowstep_supported": (C.c_int, [_P, _P, C.c_int32]),
    "pa_debug_rowstep_prof": (C.c_int, [_P]),
    "pa_debug_mlp_dw_prof": (C.c_int, [_P]),
    "pa_ppo_rowstep": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, _P, C.c_int32, _P, _P, C.c_float,
                                 C.c_float, _P, C.c_float, _P, C.c_int32, _P, C.c_int32, _P,
                                 C.c_int32, _P, _P, _P]),
    "pa_dsac_actor_rowstep": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, C.c_int32,
                                        _P, _P, _P]),
    "pa_dsac_target_rowstep": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P,
                                         C.c_float, _P, _P]),
    "pa_wmse_rowstep": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, _P, _P, _P, _P]),
    "pa_wloss_rowstep": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, C.c_int32, C.c_int32, _P, _P, _P,
                                   _P, _P]),
    "pa_mse_rowstep2": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, _P, C.c_float, C.c_float, _P, _P,
                                  _P, _P, _P, _P]),
    "pa_mlp_q_all2": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, _P, C.c_int64, C.c_int32, C.c_int32,
                                C.c_int32, _P, _P, _P]),
    "pa_mlp_forward2": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, C.c_int32, _P, C.c_int32, _P,
                                  C.c_int32, C.c_int32, _P]),
    "pa_mlp_backward2": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, _P, C.c_int32, _P, C.c_int32,
                                   C.c_int32, _P, _P, C.c_int32, _P]),
    "pa_expand_state_actions": (C.c_int, [_P, C.c_int32, _P, C.c_int64, C.c_int32, C.c_int32,
                                          C.c_int32, C.c_int32, _P, _P]),
    "pa_dsac_actor_head": (C.c_int, [_P, C.c_int32, _P, _P, _P, _P, C.c_int32, C.c_int32, _P,
                                     C.c_int32, _P, _P, _P]),
    "pa_dsac_target": (C.c_int, [_P, C.c_int32, _P, _P, _P, _P, _P, _P, C.c_float, C.c_int32,
                                 C.c_int32, _P, _P]),
    "pa_cql_head": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, _P,
                              _P, _P]),
    "pa_iql_value_head": (C.c_int, [_P, _P, _P, C.c_int32, C.c_float, C.c_float, C.c_float,
                                    C.c_int32, _P, _P, _P, _P]),
    "pa_awr_head": (C.c_int, [C.c_int32, _P, C.c_int32, _P, C.c_int32, _P, C.c_int32, C.c_int32, _P,
                              C.c_int32, _P, _P]),
    "pa_td_target": (C.c_int, [_P, C.c_int32, _P, C.c_int32, _P, C.c_int32, _P, _P, C.c_float,
                               C.c_int32, C.c_int32, _P, _P, _P]),
    "pa_argmax_rows": (C.c_int, [_P, C.c_int32, _P, C.c_int32, _P, C.c_int64, C.c_int32, C.c_int32,
                                 C.c_int32, _P, _P, _P]),
    "pa_td_head": (C.c_int, [_P, C.c_int32, _P, C.c_int32, C.c_float, _P, _P, _P]),
    "pa_rows_dot": (C.c_int, [_P, C.c_int32, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    "pa_rows_scale": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, _P]),
    "pa_rows_bmm": (C.c_int, [_P, C.c_int64, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    "pa_dueling_q": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, C.c_int32, _P, _P]),
    "pa_dueling_grad": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P]),
    "pa_dueling_feat_grad": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P,
                                       C.c_int32, _P]),
    "pa_dueling_cql_grad": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, _P, _P]),
    "pa_rows_bmm_t": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P,
                                C.c_int32, _P]),
    "pa_squarecb_probs": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32,
                                    C.c_float, C.c_float, _P, _P, _P]),
    "pa_gauss_awr_head": (C.c_int, [_P, C.c_int32, _P, C.c_int32, _P, _P, _P, C.c_int32, C.c_int32,
                                    _P, C.c_int32, _P, _P, _P]),
    "pa_tanh_action": (C.c_int, [_P, C.c_int32, _P, C.c_int32, _P, _P, C.c_float, C.c_int32,
                                 C.c_int32, _P, C.c_int32, _P]),
    "pa_tanh_action_grad": (C.c_int, [_P, C.c_int32, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32,
                                      _P, C.c_int32, _P]),
    "pa_neg_mean_head": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, _P]),
    "pa_mlp_invalidate": (C.c_int, [_P]),
    "pa_mlp_flush_grads": (C.c_int, [_P, _P]),
    "pa_mlp_adam": (C.c_int, [_P, C.c_int64, _P]),
    "pa_mlp_adam2": (C.c_int, [_P, _P, C.c_int64, C.c_float, _P]),
    "pa_mlp_flush_grads2": (C.c_int, [_P, _P, _P]),
    "pa_mlp_adamw2": (C.c_int, [_P, _P, C.c_int64, C.c_int64, _P]),
    "pa_mlp_soft_update": (C.c_int, [_P, C.c_float, _P]),
    "pa_softmax_action_prob": (C.c_int, [_P, C.c_int32, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P]),
    "pa_ppo_actor_loss": (C.c_int, [_P, C.c_int32, _P, C.c_int32, _P, _P, C.c_int32, C.c_int32,
                                    C.c_float, C.c_float, _P, C.c_int32, _P, _P]),
    "pa_ppo_heads": (C.c_int, [_P, C.c_int32, _P, C.c_int32, _P, _P, C.c_int32, C.c_int32,
                               C.c_float, C.c_float, _P, C.c_int32, _P, C.c_int32, _P, _P, _P, _P]),
    "pa_mse_head": (C.c_int, [_P, C.c_int32, _P, C.c_int32, C.c_float, C.c_float, C.c_int32, _P, _P, _P]),
    "pa_ppo_gae": (C.c_int, [_P, _P, _P, _P, _P, C.c_float, C.c_float, C.c_int64, _P, _P, _P]),
    "pa_gauss_sample": (C.c_int, [_P, C.c_int32, _P, C.c_int32, _P, _P, C.c_int32, C.c_int32, _P,
                                  C.c_int32, _P, _P]),
    "pa_gauss_actor_grad": (C.c_int, [_P, C.c_int32, _P, C.c_int32, _P, _P, _P, _P, C.c_int32, _P,
                                      C.c_int32, C.c_int32, _P, C.c_int32, _P]),
    "pa_sac_twin": (C.c_int, [C.c_int32, _P, _P, _P, _P, _P, _P, C.c_float, C.c_int32, _P, _P, _P, _P]),
    "pa_sac_scratch_floats": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "pa_sac_step": (C.c_int, [C.POINTER(SacStepArgs), _P]),
    "pa_ac_check": (C.c_int, [_P]),
    "pa_ddpg_scratch_floats": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "pa_ddpg_step": (C.c_int, [C.POINTER(DdpgStepArgs), _P]),
    "pa_sac_learn": (C.c_int, [C.POINTER(SacStepArgs), _P, C.POINTER(AcLoopArgs), _P]),
    "pa_ddpg_learn": (C.c_int, [C.POINTER(DdpgStepArgs), _P, C.POINTER(AcLoopArgs), _P]),
    "pa_debug_sac_prof": (C.c_int, [_P, _P]),
    "pa_sac_timing": (C.c_int, [C.c_int32]),
    "pa_sac_timing_read": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "pa_mlp_timing": (C.c_int, [C.c_int32]),
    "pa_debug_set_dw_split": (C.c_int, [C.c_int32]),
    "pa_debug_set_target_rows": (C.c_int, [C.c_int32]),
    "pa_debug_workspace": (C.c_int, [_P, C.c_char_p, _P, C.c_int64, C.POINTER(C.c_int32)]),
    "pa_debug_target_workers": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "pa_debug_set_rowstep_split": (C.c_int, [C.c_int32]),
    "pa_rowstep_last_split": (C.c_int, []),
    "pa_mlp_timing_read": (C.c_int, [C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "pa_sac_alpha_step": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int32, C.c_float, C.c_double,
                                    C.c_double, C.c_double, C.c_double, C.c_double, C.c_int32,
                                    C.c_int64, _P, _P]),
    "pa_weighted_mse_head": (C.c_int, [_P, C.c_int32, _P, _P, C.c_int32, _P, _P, _P, _P]),
    "pa_linreg_delta2": (C.c_int, [_P, C.c_int32, _P, _P, C.c_int32, C.c_int32, _P, _P, _P, _P]),
    "pa_linreg_apply2": (C.c_int, [_P, C.c_int32, _P, _P, _P, _P, _P, _P]),
    "pa_bandit_step": (C.c_int, [_P, _P]),
    "pa_ppo_learn": (C.c_int, [_P, _P, _P]),
    "pa_iql_scratch_floats": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "pa_iql_step": (C.c_int, [_P, _P]),
    "pa_iql_learn": (C.c_int, [_P, _P, _P, _P, _P]),
    "pa_dsac_scratch_floats": (C.c_int64, [C.c_int32, C.c_int32]),
    "pa_dsac_step": (C.c_int, [_P, _P]),
    "pa_dsac_learn": (C.c_int, [_P, _P, _P, _P]),
    "pa_mlp_activation": (C.c_int, [_P, C.c_int32, C.POINTER(_P), C.POINTER(C.c_int32)]),
    "pa_weighted_loss_head": (C.c_int, [_P, C.c_int32, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P,
                                        _P, _P, _P]),
    "pa_linreg_delta": (C.c_int, [_P, C.c_int32, _P, _P, C.c_int32, C.c_int32, _P, _P, _P, _P]),
    "pa_linreg_apply": (C.c_int, [_P, C.c_int32, _P, _P, _P, _P]),
    "pa_linreg_solve": (C.c_int, [_P, _P, C.c_float, C.c_int32, _P, _P, _P, _P, _P]),
    "pa_linreg_pinv": (C.c_int, [_P, _P, C.c_float, C.c_int32, _P, _P, _P, _P]),
    "pa_linreg_sigma": (C.c_int, [_P, C.c_int32, _P, C.c_int32, C.c_int32, _P, _P]),
    "pa_concat_cols": (C.c_int, [_P, C.c_int32, _P, C.c_int32, _P, C.c_int32, C.c_int32, C.c_int32, _P]),
    "pa_debug_set_prof": (C.c_int, [_P, _P, _P, C.c_int32]),
    "pa_debug_set_prof_target": (C.c_int, [_P, _P, C.c_int32]),
    "pa_debug_linear": (C.c_int, [_P, C.c_int32, _P, C.c_int32, _P, C.c_int32, _P, _P, C.c_int32,
                                  C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "pa_debug_weight_grad": (C.c_int, [_P, C.c_int32, _P, C.c_int32, _P, C.c_int32, _P, C.c_int32,
                                       C.c_int32, C.c_int32, _P]),
}

_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Load libpearl_amd.so once.  Raises ImportError with build instructions if absent."""
    global _lib
    if _lib is not None:
        return _lib
    # PEARL_AMD_LIB: a diagnostic build of the SAME library (the AddressSanitizer one of
    # `make -C pearl_amd/csrc asan`); never a fallback — the file must exist and export every symbol
    path = os.environ.get("PEARL_AMD_LIB") or LIB_PATH
    if path != LIB_PATH:
        if not os.path.exists(path):
            raise ImportError(f"pearl_amd: PEARL_AMD_LIB={path} does not exist")
        return _load(path)
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"pearl_amd: {LIB_PATH} is missing. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C pearl_amd/csrc` "
            "(hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback for the replay/"
            "learner hot path."
        )
    return _load(LIB_PATH)


def _load(path: str) -> C.CDLL:
    global _lib
    handle = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(handle, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    if handle.pa_abi_version() != 1:
        raise ImportError("pearl_amd: ABI version mismatch between _native.py and libpearl_amd.so")
    _lib = handle
    return handle


def last_error() -> str:
    msg = lib().pa_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc: int) -> None:
    """Translate a pa_status into the exception the reference would raise."""
    if rc == PA_OK:
        return
    msg = last_error()
    if rc == PA_ERR_VALUE:
        raise ValueError(msg)  # tensor_based_replay_buffer.py:271-275
    if rc == PA_ERR_INVALID:
        raise AssertionError(msg)
    if rc == PA_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == PA_ERR_NOMEM:
        raise MemoryError(msg)
    raise NativeError(msg)


def device_count() -> int:
    return int(lib().pa_device_count())


def require_gpu() -> None:
    if device_count() <= 0 or not torch.cuda.is_available():
        raise NativeError(
            "pearl_amd: no HIP device visible. The replay arena and the learner step are HIP "
            "kernels for gfx950; there is no CPU fallback."
        )


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Raw address of a tensor's first element (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def _raw_stream(index: int) -> int:
    return torch.cuda.current_stream(index).cuda_stream


# the same value without building a torch.cuda.Stream object per launch (several us each)
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", _raw_stream)


def stream_ptr(device: torch.device) -> Optional[int]:
    """hipStream_t of torch's current stream on `device`, as an integer."""
    idx = device.index
    s = _raw_stream(torch.cuda.current_device() if idx is None else idx)
    return s if s else None
