"""ctypes binding of `libimitation_hip.so` (the C-ABI boundary, `include/imitation_hip.h`).

PyTorch is used for device memory and streams only: tensors are passed as raw device
pointers + sizes, the current HIP stream as `void*`. There is NO CPU fallback: if the shared
library is missing or a call returns non-zero, a loud exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch as th

_HERE = os.path.dirname(os.path.abspath(__file__))
# (IA_LIB: another build of the same library -- same-box A / B runs of a kernel change, tools only)
LIB_PATH = os.environ.get("IA_LIB") or os.path.join(_HERE, "libimitation_hip.so")

IA_MAX_LAYERS = 8
ACT_NONE, ACT_RELU, ACT_TANH, ACT_SOFTPLUS = 0, 1, 2, 3
GEMM_NT, GEMM_NN, GEMM_TN = 0, 1, 2
ERR_ARG, ERR_UNSUPPORTED = -1, -2


class MlpDesc(C.Structure):
    _fields_ = [("n_layers", C.c_int), ("dims", C.c_int * (IA_MAX_LAYERS + 1)), ("hidden_act", C.c_int)]


class AdamArgs(C.Structure):
    """Mirror of `ia_adam_args` (include/imitation_hip.h)."""
    _fields_ = ([(n, C.c_void_p) for n in ("grads", "exp_avg", "exp_avg_sq")] +
                [(n, C.c_float) for n in ("beta1", "beta2", "eps", "weight_decay", "step_size", "bc2_sqrt")])


class PolicyDesc(C.Structure):
    _fields_ = [("obs_dim", C.c_int), ("act_dim", C.c_int), ("hidden", C.c_int), ("discrete", C.c_int),
                ("has_norm", C.c_int), ("norm_eps", C.c_float)]


class AirlUpdateArgs(C.Structure):
    """Mirror of `ia_airl_update_args` (include/imitation_hip.h): one AIRL update of the fused shaped-net path."""
    _fields_ = ([(n, C.c_void_p) for n in ("obs0", "act0_f32", "act0_i64", "next0", "done0", "idx0")] + [("n0", C.c_int)] +
                [(n, C.c_void_p) for n in ("obs1", "act1_f32", "act1_i64", "next1", "done1", "idx1")] + [("n1", C.c_int)] +
                [(n, C.c_int) for n in ("obs_dim", "act_dim", "use_state", "use_action", "use_next_state", "use_done")] +
                [("Xb", C.c_void_p), ("ldb", C.c_int), ("Sn", C.c_void_p), ("Sc", C.c_void_p), ("ldp", C.c_int),
                 ("dones", C.c_void_p)] +
                [(n, C.c_void_p) for n in ("ws_b", "ws_n", "ws_c", "pol_obs", "pol_act")] +
                [("Db", C.c_int), ("Dp", C.c_int)] +
                [(n, C.c_void_p) for n in ("bmean", "bvar", "bcount", "pmean", "pvar", "pcount", "snapA", "merge_ticket")] +
                [("pol", C.POINTER(PolicyDesc))] +
                [(n, C.c_void_p) for n in ("pol_params", "pol_params_t", "pol_norm_mean", "pol_norm_var", "logp")] +
                [("f_bmean", C.c_void_p), ("f_bvar", C.c_void_p), ("beps", C.c_float)] +
                [(n, C.c_void_p) for n in ("pmeanA", "pvarA", "pmeanB", "pvarB")] + [("peps", C.c_float)] +
                [("params_base", C.c_void_p), ("params_pot", C.c_void_p), ("gamma", C.c_float), ("scale", C.c_float),
                 ("n_expert", C.c_int)] +
                [("Ab", C.c_void_p), ("ldab", C.c_int), ("Db1", C.c_void_p), ("Ap", C.c_void_p), ("ldap", C.c_int)] +
                [(n, C.c_void_p) for n in ("H1", "Dp1", "Dp2", "partials", "logits", "stats", "bce_part", "ticket")] +
                [("adam", AdamArgs)] +
                [("gp_e", C.c_void_p), ("gp_coef", C.c_float), ("gp_target", C.c_float), ("n_slabs", C.c_int),
                 ("n_params", C.c_int64)] +
                [(n, C.c_void_p) for n in ("U1b", "Cb", "U1p", "Cp", "U2p", "V1p", "gp_partials", "pen_part", "pen_out",
                                           "gp_ticket")])


class RolloutTailArgs(C.Structure):
    """Mirror of `ia_rollout_tail_args` (include/imitation_hip.h)."""
    _fields_ = ([(n, C.c_void_p) for n in ("obs", "act_f32", "act_i64", "next_obs", "dones")] +
                [(n, C.c_int) for n in ("obs_dim", "act_dim", "use_state", "use_action", "use_next_state", "use_done")] +
                [("X", C.c_void_p), ("ldx", C.c_int), ("desc", C.POINTER(MlpDesc))] +
                [(n, C.c_void_p) for n in ("params", "norm_mean", "norm_var")] + [("norm_eps", C.c_float), ("out_act", C.c_int)] +
                [(n, C.c_void_p) for n in ("predict_ws", "rewards", "rewards_host", "values", "episode_starts", "last_values",
                                           "last_dones")] +
                [("T", C.c_int), ("n", C.c_int), ("gamma", C.c_float), ("gae_lambda", C.c_float)] +
                [("advantages", C.c_void_p), ("returns", C.c_void_p)])


class DiscStepArgs(C.Structure):
    """Mirror of `ia_disc_step_args` (include/imitation_hip.h)."""
    _fields_ = ([("desc", C.POINTER(MlpDesc))] +
                [(n, C.c_void_p) for n in ("params", "grads", "exp_avg", "exp_avg_sq", "norm_mean", "norm_var",
                                           "norm_count")] +
                [("norm_eps", C.c_float), ("update_norm", C.c_int)] +
                [(n, C.c_void_p) for n in ("obs0", "act0_f32", "act0_i64", "next0", "done0", "idx0")] + [("n0", C.c_int)] +
                [(n, C.c_void_p) for n in ("obs1", "act1_f32", "act1_i64", "next1", "done1", "idx1")] + [("n1", C.c_int)] +
                [(n, C.c_int) for n in ("obs_dim", "act_dim", "use_state", "use_action", "use_next_state", "use_done",
                                        "n_expert")] + [("loss_scale", C.c_float)] +
                [("X", C.c_void_p), ("Xn", C.c_void_p), ("ldx", C.c_int)] +
                [(n, C.c_void_p) for n in ("hidden", "dhidden", "logits", "dlogits", "partials")] + [("splits", C.c_int)] +
                [(n, C.c_void_p) for n in ("rn_ws", "bce_ws", "stats")] +
                [("accumulate", C.c_int), ("adam", C.c_int)] +
                [(n, C.c_float) for n in ("beta1", "beta2", "adam_eps", "weight_decay", "step_size", "bc2_sqrt")] +
                [(n, C.c_void_p) for n in ("pnorm_mean", "pnorm_var", "pnorm_count")] + [("pnorm_dim", C.c_int)] +
                [("fused_ws", C.c_void_p), ("pre_assembled", C.c_int)] +
                [("gp_e", C.c_void_p), ("gp_coef", C.c_float), ("gp_target", C.c_float), ("gp_ws", C.c_void_p),
                 ("gp_out", C.c_void_p)])


class HipExtensionMissing(RuntimeError):
    pass


_P, _I, _L, _F, _D = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double
_SIGS = {
    "ia_version": ([], C.c_int),
    "ia_host_mt19937_permutations": ([_P, C.POINTER(C.c_int), _L, _I, _P], C.c_int),
    "ia_host_mt19937_permutations_then_randint": ([_P, C.POINTER(C.c_int), _L, _I, _P, _L, _L, _L, _P, _P,
                                                   C.POINTER(C.c_int)], C.c_int),
    "ia_host_mt19937_seeded_permutations": ([_P, _I, _L, _I, _P], C.c_int),
    "ia_im2col_u8_nchw": ([_P, _I, _I, _I, _I, _I, _I, _I, _F, _P, _P], C.c_int),
    "ia_im2col_f32_nhwc": ([_P, _I, _I, _I, _I, _I, _I, _I, _P, _P], C.c_int),
    "ia_col2im_nhwc": ([_P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P], C.c_int),
    "ia_im2col_f32_nhwc_pad": ([_P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P], C.c_int),
    "ia_col2im_nhwc_pad": ([_P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P], C.c_int),
    "ia_relu_backward": ([_P, _P, _L, _P, _P], C.c_int),
    "ia_avgpool_nhwc": ([_P, _I, _I, _I, _P, _P], C.c_int),
    "ia_avgpool_nhwc_backward": ([_P, _I, _I, _I, _P, _P], C.c_int),
    "ia_conv3x3_c32_wgrad_slabs": ([_I], C.c_int),
    "ia_conv3x3_c32_wgrad": ([_P, _P, _I, _I, _I, _P, _P, _P], C.c_int),
    "ia_avgpool_relu_backward": ([_P, _P, _I, _I, _I, _P, _P], C.c_int),
    "ia_conv3x3_c4_forward": ([_P, _P, _P, _I, _I, _I, _I, _P, _P], C.c_int),
    "ia_conv3x3_c32_conv": ([_P, _P, _P, _P, _I, _I, _I, _I, _P, _P], C.c_int),
    "ia_conv3x3_c4_wgrad": ([_P, _P, _I, _I, _I, _P, _P, _P], C.c_int),
    "ia_categorical_loss": ([_P, _I, _P, _I, _I, _F, _F, _P, _P, _P, _P], C.c_int),
    "ia_gemm_f32_im2col": ([_I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _P], C.c_int),
    "ia_gemm_f32_im2col_pad": ([_I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _I,
                               C.POINTER(C.c_int), _P, _P], C.c_int),
    "ia_airl_fused_ok": ([_I, _I, _I, _I, _I], C.c_int),
    "ia_airl_fused_slabs": ([_I], C.c_int),
    "ia_airl_debug_timing": ([_P], C.c_int),
    "ia_airl_prepare": ([_P] * 6 + [_I] + [_P] * 6 + [_I] + [_I] * 6 + [_P, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P],
                        C.c_int),
    "ia_airl_stats_merge": ([_P, _P, _P, _I, _L, _I, _I, _I] + [_P] * 9, C.c_int),
    "ia_airl_round": ([C.POINTER(AirlUpdateArgs), _I, _P], C.c_int),
    "ia_obs_moments_round": ([_P, _P, _I, _P, _P, _I, _I, _I, _L, _P, _I, _P, _L, _P], C.c_int),
    "ia_airl_gp_shaped": ([_P, _I, _I, _P, _P, _I, _I, _P, _P, _P, _P, _F, _P, _P, _F, _P, _P, _I, _I, _I, _I, _I, _I, _F, _F,
                           _F, _I] + [_P] * 12, C.c_int),
    "ia_airl_step_shaped": ([_P, _I, _I, _P, _P, _I, _I, _P, _P, _P, _P, _F, _P, _P, _P, _P, _F, _P, _P, _F, _F, _I, _I,
                            _P, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P], C.c_int),
    "ia_conv1_u8_implicit_ok": ([_I, _I, _I, _I, _I, _I, _I], C.c_int),
    "ia_conv1_u8_forward": ([_P, _I, _I, _I, _P, _P, _F, _P, _P, _P], C.c_int),
    "ia_conv1_u8_wgrad_ws_floats": ([_I], C.c_longlong),
    "ia_conv1_u8_wgrad": ([_P, _I, _I, _I, _P, _F, _P, _I, _P, _P, _P], C.c_int),
    "ia_gauss_act": ([_P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P], C.c_int),
    "ia_gauss_eval": ([_P, _P, _P, _I, _I, _P, _P, _P], C.c_int),
    "ia_adv_moments": ([_P, _I, _P, _P], C.c_int),
    "ia_clip_grad_norm_ws_floats": ([], C.c_longlong),
    "ia_clip_grad_norm": ([_P, C.c_longlong, _F, _P, _P, _P], C.c_int),
    "ia_ppo_head_loss_ws_floats": ([_I], C.c_longlong),
    "ia_ppo_head_loss": ([_I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _F, _F, _P, _P, _P, _P, _P, _P], C.c_int),
    "ia_mlp_param_count": ([C.POINTER(MlpDesc)], C.c_int64),
    "ia_mlp_hidden_floats_per_row": ([C.POINTER(MlpDesc)], C.c_int64),
    "ia_gemm_f32": ([_I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _I, _P, _I, _I, _P, _P], C.c_int),
    "ia_gemm_f32_nt_splitk": ([_P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _P, _I, _I, _P], C.c_int),
    "ia_gemm_set_config": ([_I], C.c_int),
    "ia_prof_enable": ([_I], C.c_int),
    "ia_prof_collect": ([_P, _P, _P], C.c_int),
    "ia_mlp_forward": ([C.POINTER(MlpDesc), _P, _P, _I, _I, _P, _P, _I, _P], C.c_int),
    "ia_mlp_backward": ([C.POINTER(MlpDesc), _P, _P, _I, _I, _P, _P, _P, _P, _I, _P, _P], C.c_int),
    "ia_reduce_partials": ([_P, _I, _L, _F, _I, _P, _P], C.c_int),
    "ia_reduce_partials_strided": ([_P, _I, _L, _L, _F, _I, _P, _P], C.c_int),
    "ia_copy_pieces": ([_I, _P, _P, _P, _P], C.c_int),
    "ia_reduce_partials_multi": ([_I, _P, _P, _P, _P, _P], C.c_int),
    "ia_adam_step": ([_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _F, _P], C.c_int),
    "ia_adam_step_scalars": ([_P, C.c_double, _P, C.c_double, C.c_double, _P, _P], C.c_int),
    "ia_adam_step_dev": ([_P, _P, _P, _P, _L, _F, _F, _F, _F, _P, _P], C.c_int),
    "ia_running_norm_ws_floats": ([_I, _I], C.c_int64),
    "ia_running_norm_update": ([_P, _I, _I, _I, _P, _P, _P, _P, _P], C.c_int),
    "ia_running_norm_partial": ([_P, _I, _I, _I, _P, _P], C.c_int),
    "ia_running_norm_merge": ([_P, _I, _I, _I, _I, _P, _P, _P, _P], C.c_int),
    "ia_ema_norm_merge": ([_P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _F, _P], C.c_int),
    "ia_running_norm_merge_seq": ([_P, _I, _L, _I, _I, _I, _I, _P, _P, _P, _P, _P], C.c_int),
    "ia_running_norm_apply": ([_P, _I, _I, _I, _P, _P, _F, _P, _I, _P], C.c_int),
    "ia_gather_concat": ([_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _I, _P], C.c_int),
    "ia_bce_ws_floats": ([_I], C.c_int64),
    "ia_bce_logits": ([_P, _I, _I, _F, _P, _P, _P, _P], C.c_int),
    "ia_reduce_partials_adam": ([_P, _I, _L, _F, _P, _P, _P, _P, _F, _F, _F, _F, _F, _F, _P], C.c_int),
    "ia_disc_step_basic": ([C.POINTER(DiscStepArgs), _P], C.c_int),
    "ia_disc_round_basic": ([C.POINTER(DiscStepArgs), _I, _P], C.c_int),
    "ia_disc_fused_ws_floats": ([C.POINTER(MlpDesc), _I, _I], C.c_int64),
    "ia_disc_fused_gp_ws_floats": ([C.POINTER(MlpDesc), _I, _I], C.c_int64),
    "ia_disc_fused_debug_timing": ([_P], C.c_int),
    "ia_disc_fused_tile_rows": ([_I], C.c_int),
    "ia_disc_fused_split_tiles": ([_I], C.c_int),
    "ia_disc_fused_side_reduce": ([_I], C.c_int),
    "ia_disc_fused_gp_groups": ([_I], C.c_int),
    "ia_rollout_tail": ([C.POINTER(RolloutTailArgs), _P], C.c_int),
    "ia_disc_fused_predict_ws_floats": ([C.POINTER(MlpDesc), _I], C.c_int64),
    "ia_disc_fused_predict": ([C.POINTER(MlpDesc), _P, _P, _I, _I, _P, _P, _F, _I, _P, _P, _P], C.c_int),
    "ia_disc_assemble_round": ([C.POINTER(DiscStepArgs), _I, _L, _L, _L, _P], C.c_int),
    "ia_disc_fused_prepare": ([C.POINTER(MlpDesc), _P, _I, _I, _P, _P], C.c_int),
    "ia_disc_fused_adam": ([C.POINTER(MlpDesc), _P, _F, _I, _I, _P, _P, _P], C.c_int),
    "ia_gp_interpolate": ([_P, _I, _I, _I, _P, _P, _P, _F, _P, _I, _P], C.c_int),
    "ia_gp_row_coeffs": ([_P, _I, _I, _I, _P, _F, _F, _F, _P, _P, _P], C.c_int),
    "ia_gp_shaped_coeffs": ([_P, _I, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P, _F, _P, _F, _F, _F, _F, _P, _P, _P, _P,
                             _P], C.c_int),
    "ia_airl_logits": ([_P, _P, _P, _P, _P, _F, _I, _P, _P], C.c_int),
    "ia_airl_route_grad": ([_P, _P, _F, _I, _P, _P, _P, _P], C.c_int),
    "ia_gather_rows": ([_P, _P, _I, _I, _P, _P], C.c_int),
    "ia_reward_norm_sequential": ([_P, _I, _I, _F, _I, _P, _P, _P, _P, _P], C.c_int),
    "ia_reward_step_moments": ([_P, _I, _I, _P, _P], C.c_int),
    "ia_reward_norm_sequential_groups": ([_P, _I, _I, _F, _I, _P, _I, _P, _P, _P, _P, _P], C.c_int),
    "ia_policy_param_count": ([C.POINTER(PolicyDesc)], C.c_int64),
    "ia_policy_transpose": ([C.POINTER(PolicyDesc), _P, _P, _P], C.c_int),
    "ia_policy_act": ([C.POINTER(PolicyDesc), _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P], C.c_int),
    "ia_policy_rollout_mailbox": ([C.POINTER(PolicyDesc), _P, _P, _P, _P, _I, _P, _P, _P, _L, _P, _L, _P, _L, _P, _L, _P, _L,
                                   _P, _L, _P, _I, _P, _P, _D, _P], C.c_int),
    "ia_policy_logits_mailbox": ([C.POINTER(PolicyDesc), _P, _P, _P, _P, _I, _P, _L, _P, _P, _L, _I, _P, _P, _D, _P],
                                 C.c_int),
    "ia_host_wait_i32": ([_P, _I, _I, _D], C.c_int),
    "ia_policy_evaluate": ([C.POINTER(PolicyDesc), _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P], C.c_int),
    "ia_policy_logits": ([C.POINTER(PolicyDesc), _P, _P, _P, _P, _P, _I, _P, _P, _P], C.c_int),
    "ia_gae": ([_P, _P, _P, _P, _P, _I, _I, _F, _F, _P, _P, _P], C.c_int),
    "ia_timeout_bootstrap": ([_P, _P, _P, _F, _L, _P], C.c_int),
    "ia_ppo_ws_floats": ([C.POINTER(PolicyDesc), _I, _L], C.c_int64),
    "ia_ppo_minibatch": ([C.POINTER(PolicyDesc), _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F,
                          _F, _F, _F, _P, _P, _F, _F, _F, _F, _F, _P, _P, _P], C.c_int),
    "ia_ppo_minibatch_grad": ([C.POINTER(PolicyDesc), _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I,
                               _F, _F, _F, _P, _P], C.c_int),
    "ia_ppo_debug_timing": ([_P], C.c_int),
    "ia_ppo_epoch_split": ([_I], C.c_int),
    "ia_ppo_epoch_debug_timing": ([_P], C.c_int),
    "ia_ppo_grad_offset": ([C.POINTER(PolicyDesc), _I], C.c_int64),
    "ia_ppo_minibatch_apply": ([C.POINTER(PolicyDesc), _P, _P, _I, _F, _F, _F, _P, _P, _F, _F, _F, _F, _F, _P, _P, _P],
                               C.c_int),
    "ia_ppo_epoch": ([C.POINTER(PolicyDesc), _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _F,
                      _F, _F, _P, _P, _D, _D, _D, _F, _L, _P, _P, _P], C.c_int),
    "ia_ppo_epochs": ([C.POINTER(PolicyDesc), _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F,
                       _F, _F, _P, _P, _D, _D, _D, _F, _L, _P, _P, _P], C.c_int),
    "ia_ppo_update_ws_floats": ([C.POINTER(PolicyDesc), _I], C.c_int64),
    "ia_ppo_update_xcd_pack": ([_I], C.c_int),
    "ia_ppo_update_assume_cus": ([_I], C.c_int),
    "ia_ppo_update": ([C.POINTER(PolicyDesc), _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F,
                       _F, _F, _F, _P, _P, _D, _D, _D, _F, _L, _P, _P, _P], C.c_int),
    "ia_ppo_update_sharded_ws_floats": ([C.POINTER(PolicyDesc), _I, _I], C.c_int64),
    "ia_ppo_shard_recv_bytes": ([C.POINTER(PolicyDesc), _I], C.c_int64),
    "ia_ppo_update_sharded": ([C.POINTER(PolicyDesc), _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F,
                               _F, _F, _F, _P, _P, _D, _D, _D, _F, _L, _P, _P,
                               _I, _I, C.c_uint32, _P, _P, _I, _D, _P], C.c_int),
    "ia_peer_alloc": ([C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_int)], C.c_int),
    "ia_peer_free": ([_P], C.c_int),
    "ia_peer_ipc_export": ([_P, _P], C.c_int),
    "ia_peer_ipc_open": ([_P, C.POINTER(C.c_void_p)], C.c_int),
    "ia_peer_ipc_close": ([_P], C.c_int),
    "ia_peer_handshake": ([_I, _I, C.c_uint32, _P, _P, _D, _P, _P], C.c_int),
}

EXPORTED_SYMBOLS = tuple(_SIGS)
_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Loads the shared library (idempotent). Raises `HipExtensionMissing` when it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipExtensionMissing(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C imitation_amd/csrc). The HIP path has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (argtypes, restype) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = restype
    _lib = lib
    return lib


def ptr(t: Optional[th.Tensor]):
    """Raw device pointer of a tensor (None -> NULL). The tensor must be contiguous."""
    if t is None:
        return None
    assert t.is_contiguous(), "C-ABI expects contiguous buffers"
    return t.data_ptr()


_raw_stream = getattr(th._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(th._C, "_cuda_getDevice", None)


def stream():
    """Raw handle of torch's current stream on the current device (the C-level getters when this torch has them:
    `current_stream().cuda_stream` costs ~5 us of Python per call, and every launch asks)."""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return th.cuda.current_stream().cuda_stream


_side_streams: dict = {}


def side_stream(device, name: str, priority: int = 0) -> "th.cuda.Stream":
    """The process-wide side stream `name` of `device` (created on first use). Trainers share them instead
    of creating their own: HIP multiplexes streams onto a handful of hardware queues in creation order, so
    every additional stream risks landing on the queue of one it is supposed to overlap with (a second
    trainer built in the same process ran 30 % slower that way)."""
    dev = th.device(device)
    key = (dev.index if dev.index is not None else th.cuda.current_device(), name)
    if key not in _side_streams:
        _side_streams[key] = th.cuda.Stream(device=dev, priority=priority)
    return _side_streams[key]


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"libimitation_hip: {what} failed with code {rc}")


def call(name: str, *args) -> None:
    lib = load()
    check(getattr(lib, name)(*args), name)


def mlp_desc(dims, hidden_act: int) -> MlpDesc:
    d = MlpDesc()
    assert 2 <= len(dims) <= IA_MAX_LAYERS + 1
    d.n_layers = len(dims) - 1
    for i, v in enumerate(dims):
        d.dims[i] = int(v)
    d.hidden_act = hidden_act
    return d
