"""ctypes binding of ``liborl_hip.so`` - the C ABI declared in ``include/orl_hip.h``.

This is the ONLY compute backend of the package: there is no eager / CPU fallback.  If the
shared object is missing and cannot be built, or a call returns non-zero, an exception is raised.
Tensors are passed as raw device pointers (``tensor.data_ptr()``) plus the current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "liborl_hip.so")

ORL_HEAD_VALUE, ORL_HEAD_CATEGORICAL, ORL_HEAD_GAUSSIAN = 0, 1, 2
ORL_ENV_SYNTH, ORL_ENV_CARTPOLE, ORL_ENV_TTT, ORL_ENV_TTT_POOL, ORL_ENV_MPE_SPREAD = 0, 1, 2, 3, 4
ORL_GATHER_MAX = 12
ORL_IPC_HANDLE_BYTES = 64
ORL_VERSION = 306  # must equal include/orl_hip.h; checked against the loaded library
ORL_N_STATS = 16

c_f32p = C.c_void_p  # device pointers travel as void*


class NativeError(RuntimeError):
    pass


class NetDesc(C.Structure):
    _fields_ = [("obs_dim", C.c_int32), ("hidden", C.c_int32), ("n_out", C.c_int32), ("head_kind", C.c_int32)]


class PackSrc(C.Structure):
    _fields_ = [("policy_obs", C.c_void_p), ("critic_obs", C.c_void_p), ("actions", C.c_void_p),
                ("action_log_probs", C.c_void_p), ("value_preds", C.c_void_p), ("returns", C.c_void_p),
                ("active_masks", C.c_void_p), ("action_masks", C.c_void_p),
                ("Dp", C.c_int32), ("Dc", C.c_int32), ("a", C.c_int32), ("K", C.c_int32)]


class BufferPtrs(C.Structure):
    _fields_ = [("policy_obs", C.c_void_p), ("critic_obs", C.c_void_p), ("rewards", C.c_void_p),
                ("masks", C.c_void_p), ("bad_masks", C.c_void_p), ("active_masks", C.c_void_p),
                ("action_masks", C.c_void_p),
                ("T", C.c_int32), ("N", C.c_int32), ("A", C.c_int32), ("Dp", C.c_int32), ("Dc", C.c_int32),
                ("K", C.c_int32)]


class CopyDesc(C.Structure):
    _fields_ = [("src", C.c_void_p * 8), ("dst", C.c_void_p * 8), ("n", C.c_int64 * 8), ("count", C.c_int32)]


class GatherDesc(C.Structure):
    _fields_ = [("src", C.c_void_p * ORL_GATHER_MAX), ("dst", C.c_void_p * ORL_GATHER_MAX),
                ("width", C.c_int32 * ORL_GATHER_MAX), ("count", C.c_int32)]


class PPOHParams(C.Structure):
    _fields_ = [("clip_param", C.c_float), ("entropy_coef", C.c_float), ("value_loss_coef", C.c_float),
                ("huber_delta", C.c_float), ("dual_clip_coeff", C.c_float), ("max_grad_norm", C.c_float),
                ("use_clipped_value_loss", C.c_int32), ("use_huber_loss", C.c_int32),
                ("use_value_active_masks", C.c_int32), ("use_policy_active_masks", C.c_int32),
                ("use_valuenorm", C.c_int32), ("dual_clip_ppo", C.c_int32), ("use_max_grad_norm", C.c_int32),
                ("reserved", C.c_int32)]


class AdamState(C.Structure):
    _fields_ = [("theta", C.c_void_p), ("grad", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p),
                ("lr", C.c_float), ("eps", C.c_float), ("weight_decay", C.c_float), ("step", C.c_int32)]


class RolloutArgs(C.Structure):
    _fields_ = [("buf", BufferPtrs), ("value_preds", C.c_void_p), ("actions", C.c_void_p),
                ("action_log_probs", C.c_void_p), ("env_state", C.c_void_p), ("ep_stats", C.c_void_p),
                ("env_kind", C.c_int32), ("episode_limit", C.c_int32), ("env_seed", C.c_uint64),
                ("act_seed", C.c_uint64), ("rng_step0", C.c_uint64), ("opp_thetas", C.c_void_p),
                ("opp_theta_stride", C.c_int64), ("opp_group_rows", C.c_int32), ("opp_reserved", C.c_int32),
                ("opp_seed", C.c_uint64), ("opp_rng_step0", C.c_uint64), ("opp_index", C.c_void_p),
                ("opp_per_reset", C.c_int32), ("opp_n_policies", C.c_int32), ("opp_n_filled", C.c_int32),
                ("opp_last_slot", C.c_int32), ("opp_strategy", C.c_int32), ("opp_pad", C.c_int32),
                ("opp_sample_seed", C.c_uint64), ("opp_draw_id0", C.c_uint64)]


class RnnBatch(C.Structure):
    _fields_ = [("records", C.c_void_p), ("rows", C.c_void_p), ("masks", C.c_void_p), ("h_policy", C.c_void_p),
                ("h_critic", C.c_void_p), ("rec_width", C.c_int32), ("n_chunks", C.c_int32), ("L", C.c_int32),
                ("n_chunks_critic", C.c_int32), ("rows_critic", C.c_void_p)]


class RnnRolloutArgs(C.Structure):
    _fields_ = [("buf", BufferPtrs), ("value_preds", C.c_void_p), ("actions", C.c_void_p),
                ("action_log_probs", C.c_void_p), ("rnn_states", C.c_void_p), ("rnn_states_critic", C.c_void_p),
                ("env_state", C.c_void_p), ("ep_stats", C.c_void_p), ("obs_policy_out", C.c_void_p),
                ("obs_critic_out", C.c_void_p), ("next_value", C.c_void_p), ("env_kind", C.c_int32),
                ("world_length", C.c_int32), ("deterministic", C.c_int32), ("reserved", C.c_int32),
                ("env_seed", C.c_uint64), ("act_seed", C.c_uint64), ("rng_step0", C.c_uint64),
                ("rng_step_dev", C.c_void_p), ("sync_flags", C.c_void_p), ("env_step0", C.c_uint64)]


class HeadDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("n_out", C.c_int32), ("n_heads", C.c_int32), ("nvec", C.c_int32 * 8)]


ORL_GEN_MLP_MAX_LAYERS = 14


class GenMlpLayer(C.Structure):
    _fields_ = [("W", C.c_void_p), ("bias", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("n_in", C.c_int32), ("n_out", C.c_int32), ("act", C.c_int32), ("reserved", C.c_int32)]


class GenMlpDesc(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("n_heads", C.c_int32), ("fn_gamma", C.c_void_p), ("fn_beta", C.c_void_p),
                ("layer", GenMlpLayer * ORL_GEN_MLP_MAX_LAYERS)]


ORL_GT_MAX_LAYERS = 4


class GtDesc(C.Structure):
    """``orl_gt_desc``: a feed-forward general tower for the cross-layer fused kernels (offsets into ``theta``)."""
    _fields_ = [("theta", C.c_void_p), ("D", C.c_int32), ("H", C.c_int32), ("n_layers", C.c_int32), ("n_heads", C.c_int32),
                ("o_fn_g", C.c_int32), ("o_fn_be", C.c_int32),
                ("oW", C.c_int32 * ORL_GT_MAX_LAYERS), ("ob", C.c_int32 * ORL_GT_MAX_LAYERS),
                ("og", C.c_int32 * ORL_GT_MAX_LAYERS), ("obe", C.c_int32 * ORL_GT_MAX_LAYERS),
                ("act", C.c_int32 * ORL_GT_MAX_LAYERS),
                ("head_oW", C.c_int32 * 2), ("head_ob", C.c_int32 * 2), ("head_n", C.c_int32 * 2)]


class GtLoss(C.Structure):
    """``orl_gt_loss``: what ``orl_gt_train`` needs to evaluate the losses of its rows in the backward kernel."""
    _fields_ = [("head", HeadDesc), ("logstd", C.c_void_p), ("den", C.c_void_p), ("vn_state", C.c_void_p),
                ("hp", PPOHParams), ("Dp", C.c_int32), ("Dc", C.c_int32), ("a_w", C.c_int32), ("K", C.c_int32),
                ("policy_head", C.c_int32), ("value_head", C.c_int32), ("policy_grad", C.c_int32), ("reserved", C.c_int32)]


ORL_ACT_NONE, ORL_ACT_TANH, ORL_ACT_RELU, ORL_ACT_LEAKY_RELU, ORL_ACT_ELU = -1, 0, 1, 2, 3
ORL_HEAD_MULTI_DISCRETE = 3
ORL_HEAD_MIXED = 4

# order of orl_abi_struct_size(which)
_ABI_STRUCTS = (NetDesc, PackSrc, BufferPtrs, CopyDesc, GatherDesc, PPOHParams, AdamState, RolloutArgs, RnnBatch,
                RnnRolloutArgs, GenMlpDesc, GtDesc, GtLoss)

# name -> (restype, argtypes); must list EVERY symbol of include/orl_hip.h (tests check this)
_P = C.c_void_p
_SIGNATURES = {
    "orl_version": (C.c_int, []),
    "orl_build_experiments": (C.c_int, []),
    "orl_tower_split_terms": (C.c_int, []),
    "orl_abi_struct_size": (C.c_int, [C.c_int]),
    "orl_last_error_string": (C.c_char_p, []),
    "orl_param_count": (C.c_int, [C.POINTER(NetDesc)]),
    "orl_raw_grad_count": (C.c_int, [C.POINTER(NetDesc)]),
    "orl_gae_scan": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int,
                               _P, _P, _P, C.POINTER(C.c_int), _P]),
    "orl_gae_max_partials": (C.c_int, [C.c_int, C.c_int]),
    "orl_adv_stats": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, _P, _P, C.POINTER(C.c_int), _P]),
    "orl_record_width": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "orl_adv_normalize_pack": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.POINTER(PackSrc), _P, _P]),
    "orl_buffer_insert": (C.c_int, [C.POINTER(BufferPtrs), C.c_int, _P, _P, _P, _P, _P, _P, _P]),
    "orl_buffer_insert_rnn": (C.c_int, [C.POINTER(BufferPtrs), C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int, _P]),
    "orl_multi_copy": (C.c_int, [C.POINTER(CopyDesc), _P]),
    "orl_gather_minibatch": (C.c_int, [C.POINTER(GatherDesc), _P, C.c_int, _P]),
    "orl_perm_feistel": (C.c_int, [_P, C.c_int64, C.c_uint64, C.c_uint64, _P]),
    "orl_perm_feistel_vn": (C.c_int, [_P, C.c_int64, C.c_uint64, C.c_uint64, _P, _P, C.c_double, _P]),
    "orl_act_step": (C.c_int, [C.POINTER(NetDesc), _P, C.POINTER(NetDesc), _P, _P, _P, _P, C.c_int, C.c_int,
                               C.c_uint64, C.c_uint64, C.c_uint64, _P, _P, _P, _P, _P, _P]),
    "orl_ttt_state_width": (C.c_int, []),
    "orl_ttt_reset": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_uint64, _P]),
    "orl_ttt_step": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_uint64, _P]),
    "orl_ttt_agent_move": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, _P]),
    "orl_ttt_opponent_move": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_uint64, _P]),
    "orl_critic_values": (C.c_int, [C.POINTER(NetDesc), _P, _P, C.c_int64, _P, _P]),
    "orl_act_step_grouped": (C.c_int, [C.POINTER(NetDesc), _P, C.c_int64, C.c_int, _P, _P, C.c_int, C.c_int, C.c_uint64,
                                       C.c_uint64, C.c_uint64, _P, _P, _P, _P]),
    "orl_act_step_pool": (C.c_int, [C.POINTER(NetDesc), _P, C.c_int64, C.c_int, _P, _P, _P, C.c_int, C.c_int, C.c_uint64,
                                    C.c_uint64, C.c_uint64, _P, _P, _P, _P]),
    "orl_opponent_sample": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, _P,
                                      _P]),
    "orl_evaluate_actions": (C.c_int, [C.POINTER(NetDesc), _P, C.POINTER(NetDesc), _P, _P, _P, _P, _P, _P, C.c_int, _P,
                                       _P, _P, _P, _P]),
    "orl_ppo_max_blocks": (C.c_int, []),
    "orl_ppo_fwd_bwd": (C.c_int, [C.POINTER(NetDesc), _P, C.POINTER(NetDesc), _P, _P, C.c_int, _P, C.c_int, _P,
                                  C.POINTER(PPOHParams), _P, C.POINTER(C.c_int), _P]),
    "orl_ppo_reduce": (C.c_int, [_P, C.c_int, C.c_int, _P, _P]),
    "orl_ppo_reduce_pair": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "orl_ppo_apply": (C.c_int, [C.POINTER(NetDesc), C.POINTER(NetDesc), _P, C.POINTER(PPOHParams),
                                C.POINTER(AdamState), C.POINTER(AdamState), _P, _P]),
    "orl_ppo_apply_perm": (C.c_int, [C.POINTER(NetDesc), C.POINTER(NetDesc), _P, C.POINTER(PPOHParams),
                                     C.POINTER(AdamState), C.POINTER(AdamState), _P, _P, C.c_int64, C.c_uint64,
                                     C.c_uint64, _P, _P, C.c_double, _P]),
    "orl_comm_create": (C.c_int, [C.c_int, C.c_int, C.c_int64, C.POINTER(_P), _P]),
    "orl_comm_connect": (C.c_int, [_P, _P]),
    "orl_comm_destroy": (C.c_int, [_P]),
    "orl_comm_error": (C.c_int, [_P, _P]),
    "orl_comm_error_copy": (C.c_int, [_P, _P, _P]),
    "orl_allreduce_small": (C.c_int, [_P, _P, C.c_int, _P]),
    "orl_ppo_reduce_pair_comm": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "orl_ppo_apply_comm": (C.c_int, [_P, C.POINTER(NetDesc), C.POINTER(NetDesc), _P, C.POINTER(PPOHParams),
                                     C.POINTER(AdamState), C.POINTER(AdamState), _P, _P, C.c_int64, C.c_uint64,
                                     C.c_uint64, _P, _P, C.c_double, _P]),
    "orl_ppo_reduce_apply": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.POINTER(NetDesc),
                                       C.POINTER(NetDesc), C.POINTER(PPOHParams), C.POINTER(AdamState),
                                       C.POINTER(AdamState), _P, _P, C.c_int64, C.c_uint64, C.c_uint64, _P, _P, C.c_double,
                                       _P, _P]),
    "orl_ppo_step": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.POINTER(NetDesc),
                               C.POINTER(NetDesc), C.POINTER(PPOHParams), C.POINTER(AdamState),
                               C.POINTER(AdamState), _P, _P, C.c_int64, C.c_uint64, C.c_uint64, _P, _P, C.c_double,
                               _P, _P]),
    "orl_gen_rollout_fused": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_uint64, C.c_uint64,
                                        C.c_uint64, C.c_uint64, C.c_int, _P]),
    "orl_gemm": (C.c_int, [_P, C.c_int64, C.c_int64, _P, C.c_int64, C.c_int64, _P, C.c_int64, C.c_int, C.c_int, C.c_int,
                           C.c_int, _P, _P]),
    "orl_row_fwd": (C.c_int, [_P, _P, C.c_int, _P, _P, C.c_int, C.c_int, _P, _P, _P, _P, _P]),
    "orl_row_bwd": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, C.POINTER(C.c_int), _P]),
    "orl_gen_layer_fwd": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, C.c_int, _P, _P, C.c_int, _P, _P, _P, _P]),
    "orl_gt_supported": (C.c_int, [C.POINTER(GtDesc)]),
    "orl_gt_image_floats": (C.c_int64, [C.POINTER(GtDesc)]),
    "orl_gt_raw_floats": (C.c_int64, [C.POINTER(GtDesc)]),
    "orl_gt_prep": (C.c_int, [C.POINTER(GtDesc), _P, _P]),
    "orl_gt_fwd": (C.c_int, [C.POINTER(GtDesc), _P, _P, C.c_int, C.c_int, _P, C.c_int, _P, _P, _P]),
    "orl_gt_bwd": (C.c_int, [C.POINTER(GtDesc), _P, _P, C.c_int, C.c_int, _P, C.c_int, _P, _P, _P, C.c_int64, _P, _P, _P]),
    "orl_gt_train": (C.c_int, [C.POINTER(GtDesc), _P, _P, C.c_int, C.c_int, _P, C.c_int, C.POINTER(GtLoss), _P, C.c_int64, _P, _P,
                               _P, _P]),
    "orl_gen_mlp_fwd": (C.c_int, [C.POINTER(GenMlpDesc), _P, C.c_int, _P, _P, _P, _P]),
    "orl_gen_act": (C.c_int, [C.POINTER(GenMlpDesc), _P, C.POINTER(GenMlpDesc), _P, C.c_int, _P, _P, C.POINTER(HeadDesc), _P,
                              _P, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, _P, _P, C.c_int, _P, _P, _P]),
    "orl_gen_layer_bwd": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P, _P, _P, C.c_int,
                                    C.POINTER(C.c_int), _P]),
    "orl_gen_wgrad": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int64, _P]),
    "orl_gen_colsum": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int, _P, C.c_int, _P, C.c_int, _P]),
    "orl_gather_cols": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P, _P]),
    "orl_gen_denoms": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P, _P, _P]),
    "orl_gen_policy_loss": (C.c_int, [C.POINTER(HeadDesc), _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P,
                                      C.c_int, _P, C.POINTER(PPOHParams), _P, _P, C.c_int, C.POINTER(C.c_int), _P, _P,
                                      _P]),
    "orl_gen_value_loss": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P, _P,
                                     C.POINTER(PPOHParams), _P, _P, C.c_int, C.POINTER(C.c_int), _P]),
    "orl_gen_sample": (C.c_int, [C.POINTER(HeadDesc), _P, _P, _P, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64,
                                 _P, _P, C.c_int, _P, _P, _P]),
    "orl_gen_adam": (C.c_int, [C.POINTER(AdamState), C.c_int64, C.c_float, C.c_int, C.c_int, _P, _P, C.c_int, C.c_int,
                               _P]),
    "orl_gen_matmul": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int, _P, _P]),
    "orl_gen_colsum_rows": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, C.c_int64, _P]),
    "orl_gen_gru_gate_fwd": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, _P, _P, _P, _P]),
    "orl_gen_gru_gate_bwd": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, _P, _P, _P]),
    "orl_gen_lstm_gate_fwd": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P]),
    "orl_gen_lstm_gate_bwd": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, _P, _P, _P]),
    "orl_gen_row_affine": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, _P, _P]),
    "orl_vec_add": (C.c_int, [_P, _P, C.c_int64, _P]),
    "orl_gen_info": (C.c_int, [_P, _P, _P, C.POINTER(PPOHParams), C.c_float, C.c_float, _P, _P]),
    "orl_valuenorm_update": (C.c_int, [_P, _P, C.c_double, _P]),
    "orl_minibatch_moments": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int, _P, _P, _P]),
    "orl_env_state_width": (C.c_int, [C.c_int]),
    "orl_env_reset": (C.c_int, [C.c_int, _P, _P, _P, C.c_int, C.c_int, C.c_uint64, C.c_int, _P]),
    "orl_env_step": (C.c_int, [C.c_int, _P, _P, _P, C.c_int, _P, _P, _P, C.c_int, C.c_int, C.c_uint64, C.c_int,
                               C.c_uint64, _P]),
    "orl_env_step_dev": (C.c_int, [C.c_int, _P, _P, _P, C.c_int, _P, _P, _P, C.c_int, C.c_int, C.c_uint64, C.c_int,
                                   C.c_uint64, _P, _P]),
    "orl_rollout_fused": (C.c_int, [C.POINTER(NetDesc), _P, C.POINTER(NetDesc), _P, C.POINTER(RolloutArgs), _P, _P]),
    "orl_mpe_state_width": (C.c_int, []),
    "orl_mpe_reset": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_uint64, _P]),
    "orl_mpe_step": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_uint64, C.c_int, _P]),
    "orl_rnn_param_count": (C.c_int, [C.POINTER(NetDesc)]),
    "orl_rnn_raw_grad_count": (C.c_int, [C.POINTER(NetDesc)]),
    "orl_rnn_act_step": (C.c_int, [C.POINTER(NetDesc), _P, C.POINTER(NetDesc), _P, _P, _P, _P, _P, _P, _P, C.c_int,
                                   C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, _P, _P, _P, _P, _P, _P, _P, _P]),
    "orl_rnn_eval_step": (C.c_int, [C.POINTER(NetDesc), _P, C.POINTER(NetDesc), _P, _P, _P, _P, _P, _P, _P, _P, C.c_int, _P,
                                    _P, _P, _P, _P, _P]),
    "orl_rnn_rollout_fused": (C.c_int, [C.POINTER(NetDesc), _P, C.POINTER(NetDesc), _P, C.POINTER(RnnRolloutArgs), _P]),
    "orl_rnn_chunk_rows": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "orl_rnn_chunk_rows_v3": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "orl_rnn_jrpo_records": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "orl_rnn_workspace_floats": (C.c_int64, [C.POINTER(NetDesc), C.POINTER(NetDesc), C.c_int, C.c_int]),
    "orl_rnn_ppo_fwd_bwd": (C.c_int, [C.POINTER(NetDesc), _P, C.POINTER(NetDesc), _P, C.POINTER(RnnBatch), _P,
                                      C.POINTER(PPOHParams), _P, _P, _P]),
    "orl_rnn_ppo_apply": (C.c_int, [C.POINTER(NetDesc), C.POINTER(NetDesc), _P, C.POINTER(PPOHParams),
                                    C.POINTER(AdamState), C.POINTER(AdamState), _P, _P, _P]),
}

_lib: Optional[C.CDLL] = None


def exported_symbols():
    return sorted(_SIGNATURES)


def load(build_if_missing: bool = True) -> C.CDLL:
    """Load (building first if necessary) the gfx950 extension.  Raises if unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    from .csrc import build as _b

    if not os.path.exists(LIB_PATH):
        if not build_if_missing:
            raise NativeError("%s is missing - run `python -m openrl_amd.csrc.build`" % LIB_PATH)
        _b.build()
    elif build_if_missing and _b.have_hipcc() and not os.environ.get("ORL_KEEP_BUILD"):
        # (ORL_KEEP_BUILD=1: measurement tools that load a deliberately different build, e.g. --prof)
        _b.build()  # no-op when the source digest matches the stamp; rebuilds a stale shared object
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale -> loud failure
        fn.restype = res
        fn.argtypes = args
    # a stale build can export the same names with another struct layout: refuse it instead of corrupting memory
    if lib.orl_version() != ORL_VERSION:
        raise NativeError("%s is version %d, the bindings are %d - rebuild with `python -m openrl_amd.csrc.build "
                          "--force`" % (LIB_PATH, lib.orl_version(), ORL_VERSION))
    for which, st in enumerate(_ABI_STRUCTS):
        if lib.orl_abi_struct_size(which) != C.sizeof(st):
            raise NativeError("ABI mismatch: %s is %d bytes in %s, %d in the bindings" %
                              (st.__name__, lib.orl_abi_struct_size(which), LIB_PATH, C.sizeof(st)))
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().orl_last_error_string()
        raise NativeError("%s failed (rc=%d): %s" % (what or "native call", rc, msg.decode() if msg else ""))


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Device pointer of a contiguous tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise NativeError("non-contiguous tensor passed to the native engine")
    return t.data_ptr()


def fptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is not None and t.dtype != torch.float32:
        raise NativeError("float32 tensor expected, got %s" % t.dtype)
    return ptr(t)


def stream_ptr(device=None) -> int:
    return torch.cuda.current_stream(device).cuda_stream


_GPU_OK = None  # torch.cuda.is_available() costs ~18 us per call (device-count query); the answer cannot change


def require_gpu(device) -> torch.device:
    global _GPU_OK
    if not isinstance(device, torch.device):
        device = torch.device(device)
    if _GPU_OK is None:
        _GPU_OK = bool(torch.cuda.is_available())
    if device.type != "cuda" or not _GPU_OK:
        raise NativeError(
            "openrl_amd is a MI355X (ROCm) engine: device %r has no HIP runtime behind it and there is no CPU "
            "fallback. Use the reference on CPU, or run on a gfx950 device." % (str(device),))
    return device


class DeviceErrorWatch:
    """Host-side view of a device error word that kernels may set while the host runs ahead (the 10 s peer timeout of
    ``orl_comm``, the bounded critic-chases-policy wait of ``orl_rnn_rollout_fused``).  ``post(flag)`` queues an
    asynchronous copy of the int32 device word into pinned host memory behind the work already on the stream;
    ``poll()`` raises if a completed copy carried a non-zero word - without synchronising (``wait=True``: after waiting
    for the copy).  Callers post once per update / rollout and poll at their next natural host touch point and never
    pay a sync for it.  The contract that makes ONE pinned word enough: the device word is STICKY (kernels only ever set
    it; neither ``orl_comm`` nor ``orl_rnn_rollout_fused`` clears it), so when the host has run ahead and a posted
    copy is overwritten by a later one before any poll saw it, the later copy carries the error too - a failure in
    iteration k surfaces at the first poll that finds ANY completed copy posted at or after k, at the latest the
    ``wait=True`` poll at the end of training."""

    def __init__(self, what: str) -> None:
        self.what = what
        self._host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._event = None

    def post(self, flag: torch.Tensor) -> None:
        self.poll()  # the previous copy, if it has landed
        self._host.copy_(flag.reshape(-1)[:1], non_blocking=True)
        if self._event is None:
            self._event = torch.cuda.Event()
        self._event.record(torch.cuda.current_stream(flag.device))

    def poll(self, wait: bool = False) -> None:
        if self._event is None:
            return
        if wait:
            self._event.synchronize()
        if self._event.query() and int(self._host[0]) != 0:
            raise NativeError("%s (device error word = %d)" % (self.what, int(self._host[0])))
