"""
ctypes binding of libsurreal_amd.so (include/surreal_amd.h).  The product path has NO fallback:
if the HIP library is missing or a symbol does not resolve, importing/using it raises.

PyTorch is plumbing here: device memory (tensor.data_ptr()) and the current HIP stream.
"""
import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_double, c_float, c_int32, c_int64, c_size_t,
                    c_uint64, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SMX_LIB_PATH') or os.path.join(_HERE, 'libsurreal_amd.so')

SMX_ACT_NONE, SMX_ACT_RELU, SMX_ACT_TANH = 0, 1, 2
SMX_PPO_CLIP, SMX_PPO_ADAPT = 0, 1
SMX_E_UNSUPPORTED = -3   # include/surreal_amd.h: shape outside what the kernel is built for
# stats slots (include/surreal_amd.h)
PS_SURR, PS_LOSS, PS_ENTROPY, PS_KL, PS_GRADNORM, PS_LB, PS_ISW, PS_REFBEH, PS_STRIDE = range(9)
VS_LOSS, VS_EXPVAR, VS_GRADNORM = 0, 1, 2
VS_STRIDE = 4


class Mlp3(Structure):
    """smx_mlp3_t"""
    _fields_ = [('W1', c_void_p), ('b1', c_void_p), ('W2', c_void_p), ('b2', c_void_p),
                ('W3', c_void_p), ('b3', c_void_p),
                ('D', c_int32), ('H1', c_int32), ('H2', c_int32), ('OUT', c_int32)]


class Mlp3Job(Structure):
    """smx_mlp3_job_t"""
    _fields_ = [('net', POINTER(Mlp3)), ('x', c_void_p), ('rows', c_int64), ('h1', c_void_p),
                ('h2', c_void_p), ('out', c_void_p), ('out_act', c_int32), ('out_ld', c_int32),
                ('dz3', c_void_p), ('dz2', c_void_p), ('dz1', c_void_p), ('grads', c_void_p),
                ('sumsq_partials', c_void_p), ('stop_flag', c_void_p),
                ('h1T', c_void_p), ('h2T', c_void_p), ('xT', c_void_p), ('dz3T', c_void_p),
                ('dz2T', c_void_p), ('dz1T', c_void_p), ('ldT', c_int64)]


class EpochJob(Structure):
    """smx_epoch_job_t"""
    _fields_ = [('net', POINTER(Mlp3)), ('x', c_void_p), ('rows', c_int64), ('h1T', c_void_p),
                ('h2T', c_void_p), ('ldT', c_int64), ('out', c_void_p), ('out_ld', c_int32),
                ('out_act', c_int32), ('loss', c_int32), ('reserved', c_int32), ('stop_flag', c_void_p),
                ('dz3', c_void_p), ('dz3T', c_void_p), ('dz2T', c_void_p), ('dz1T', c_void_p),
                ('packed', c_void_p)]


class EpochPack(Structure):
    """smx_epoch_pack_t"""
    _fields_ = [('net', POINTER(Mlp3)), ('packed', c_void_p)]


class LearnEpilogue(Structure):
    """smx_learn_epilogue_t"""
    _fields_ = [('x', c_void_p), ('ldx', c_int64), ('rows', c_int64), ('D', c_int32), ('A', c_int32),
                ('running_sum', c_void_p), ('running_sumsq', c_void_p), ('count', c_void_p), ('count_rows', c_float),
                ('n_epochs', c_int32), ('ret', c_void_p), ('n_ret', c_int64), ('ret_moments', c_void_p),
                ('v_partials', c_void_p), ('nblk', c_int32), ('stats_stride', c_int32), ('v_stats', c_void_p),
                ('log_var', c_void_p), ('out4', c_void_p), ('ticket', c_void_p)]


class EpochPrep(Structure):
    """smx_epoch_prep_t"""
    _fields_ = [('obs0', c_void_p), ('ld_obs0', c_int64), ('rows', c_int64), ('D', c_int32), ('A', c_int32),
                ('zmean', c_void_p), ('zstd', c_void_p), ('xn', c_void_p), ('xnT', c_void_p), ('ldT', c_int64),
                ('ref_sum', c_void_p), ('ref_sumsq', c_void_p), ('ref_count', c_void_p), ('ref_eps', c_float),
                ('ref_filter', c_int32), ('xr', c_void_p), ('obs_next', c_void_p), ('ld_next', c_int64),
                ('xnext', c_void_p), ('ref_log_var', c_void_p), ('ref_std', c_void_p), ('ld_ref', c_int64),
                ('zero_words', c_void_p), ('n_zero', c_int32), ('n_pack', c_int32), ('pack', EpochPack * 4)]


EPOCH_LOSS_NONE, EPOCH_LOSS_POLICY, EPOCH_LOSS_VALUE, EPOCH_RHS_SURR, EPOCH_RHS_KL = 0, 1, 2, 3, 4


class Lstm(Structure):
    """smx_lstm_t"""
    _fields_ = [('W_ih', c_void_p), ('W_hh', c_void_p), ('b_ih', c_void_p), ('b_hh', c_void_p),
                ('D', c_int32), ('H', c_int32)]


class AdamGroup(Structure):
    """smx_adam_group_t"""
    _fields_ = [('theta', c_void_p), ('grads', c_void_p), ('exp_avg', c_void_p),
                ('exp_avg_sq', c_void_p), ('n', c_int64), ('sumsq_partials', c_void_p),
                ('npart', c_int32), ('honour_stop', c_int32), ('grad_norm_out', c_void_p),
                ('pack_net', POINTER(Mlp3)), ('packed', c_void_p)]


class PpoLosses(Structure):
    """smx_ppo_losses_t"""
    _fields_ = [('mode', c_int32), ('A', c_int32), ('mean', c_void_p), ('log_var', c_void_p),
                ('actions', c_void_p), ('behave', c_void_p), ('ref', c_void_p), ('adv', c_void_p),
                ('ld_act', c_int32), ('ld_beh', c_int32), ('ld_ref', c_int32), ('check_stop', c_int32),
                ('rows', c_int64), ('g_surr', c_void_p), ('g_kl', c_void_p),
                ('row_partials', c_void_p), ('dz3', c_void_p), ('dz3_t', c_void_p), ('ld_t', c_int64),
                ('dlogvar', c_void_p), ('dlogvar_sumsq', c_void_p), ('stats', c_void_p),
                ('values', c_void_p), ('returns', c_void_p), ('v_dz3', c_void_p),
                ('v_partials', c_void_p), ('will_update', c_int32), ('v_will_update', c_int32)]


class PpoCombine(Structure):
    """smx_ppo_combine_t"""
    _fields_ = [('mode', c_int32), ('A', c_int32), ('row_partials', c_void_p), ('nblk', c_int32),
                ('check_stop', c_int32), ('n_total', c_int64), ('log_var', c_void_p),
                ('stats', c_void_p), ('grads_a', c_void_p), ('grads_kl', c_void_p),
                ('n_mlp', c_int64), ('n_a', c_int64), ('sumsq_a', c_void_p), ('grads_c', c_void_p),
                ('n_c', c_int64), ('sumsq_c', c_void_p), ('will_update', c_int32),
                ('reserved', c_int32)]


class LinearJob(Structure):
    """smx_linear_job_t"""
    _fields_ = [('kind', c_int32), ('act', c_int32), ('A', c_void_p), ('B', c_void_p), ('bias', c_void_p),
                ('relu_mask', c_void_p), ('C', c_void_p), ('dbias', c_void_p), ('lda', c_int32), ('ldb', c_int32),
                ('ldc', c_int32), ('a_kcontig', c_int32), ('b_kcontig', c_int32), ('M', c_int32), ('N', c_int32),
                ('K', c_int32), ('stop_flag', c_void_p)]


class DdpgNet(Structure):
    """smx_ddpg_net_t"""
    _fields_ = [(n, c_void_p) for n in ('W1', 'b1', 'W2', 'b2', 'W3', 'b3')]


class DdpgUpdate(Structure):
    """smx_ddpg_update_t"""
    _fields_ = ([(n, c_void_p) for n in ('theta', 'grads', 'exp_avg', 'exp_avg_sq', 'target')] + [('n', c_int64)] +
                [(n, c_void_p) for n in ('lr', 'step')] +
                [(n, c_float) for n in ('weight_decay', 'clip_value', 'tau')] + [('interval', c_int32), ('stats', c_void_p), ('stats_host', c_void_p)])


class DdpgRows(Structure):
    """smx_ddpg_rows_t"""
    _fields_ = ([('rows', c_int64)] + [(n, c_int32) for n in ('D', 'A', 'H1', 'H2', 'c1', 'c2')] +
                [(n, DdpgNet) for n in ('actor', 'critic', 'target_actor', 'target_critic')] +
                [(n, c_void_p) for n in ('packed', 'x', 'x_next', 'actions', 'rewards', 'dones')] +
                [('gamma_n', c_float)] +
                [(n, c_void_p) for n in ('xcat', 'h2c', 'q', 'q_next', 'y', 'dz3', 'dz2', 'dxcat', 'h1a', 'h2a', 'act',
                                         'q_actor', 'dz3a', 'dz2a', 'dz1a', 'step')])


class GatherJob(Structure):
    """smx_gather_job_t"""
    _fields_ = [('table', c_void_p), ('dst', c_void_p), ('row_bytes', c_int64)]


class SynthRollout(Structure):
    """smx_synth_rollout_t"""
    _fields_ = [('net', POINTER(Mlp3)), ('packed', c_void_p), ('out_act', c_int32), ('n', c_int32),
                ('log_var', c_void_p), ('noise_scale', c_void_p), ('eps', c_void_p), ('zsum', c_void_p),
                ('zsumsq', c_void_p), ('zcount', c_void_p), ('zeps', c_float), ('t', c_int32),
                ('episode_len', c_int32), ('steps', c_int32), ('rows_per_actor', c_int32), ('slot', c_int32),
                ('state', c_void_p), ('init_state', c_void_p), ('obs_roll', c_void_p), ('act_roll', c_void_p),
                ('rew_roll', c_void_p), ('done_roll', c_void_p), ('pd_roll', c_void_p), ('obs_last', c_void_p)]


class Xchg(Structure):
    """smx_xchg_t"""
    _fields_ = [('world', c_int32), ('rank', c_int32), ('capacity', c_int64), ('peer', c_void_p * 8)]


class SynthActStep(Structure):
    """smx_synth_act_step_t"""
    _fields_ = [('state', c_void_p), ('init_state', c_void_p), ('mean', c_void_p), ('log_var', c_void_p),
                ('noise_scale', c_void_p), ('eps', c_void_p), ('ld_mean', c_int64), ('ld_eps', c_int64),
                ('n', c_int32), ('D', c_int32), ('A', c_int32), ('t', c_int32), ('episode_len', c_int32),
                ('slot', c_int32), ('T', c_int32), ('reserved', c_int32),
                ('obs_roll', c_void_p), ('act_roll', c_void_p), ('rew_roll', c_void_p),
                ('done_roll', c_void_p), ('pd_roll', c_void_p), ('zsum', c_void_p), ('zsumsq', c_void_p),
                ('zcount', c_void_p), ('zeps', c_float), ('reserved_f', c_float), ('xn_out', c_void_p)]


# smx_ppo_ctrl_t as 16 x 4-byte words: index of each field (floats 0-9, int32 10-15)
CTRL_WORDS = 16
(C_LR_ACTOR, C_LR_CRITIC, C_BETA, C_ETA, C_CLIP_EPS, C_KL_TARGET, C_ACTOR_MAX_NORM,
 C_CRITIC_MAX_NORM, C_ACTOR_WD, C_CRITIC_WD, C_STEP_ACTOR, C_STEP_CRITIC, C_STOP,
 C_EPOCHS_DONE) = range(14)
C_XCHG_ERR = 14          # reserved[0]: raised by a peer exchange that timed out (zeroed with the per-learn words)
C_SYNC_ERR = 15          # reserved[1]: raised by smx_epoch_fwdbwd_f32 when its in-launch wait timed out

_P = c_void_p
_SIGS = {
    'smx_abi_version': (c_int32, []),
    'smx_error_string': (c_char_p, [c_int32]),
    'smx_zfilter_stats_f32': (c_int32, [_P, _P, _P, c_int32, c_float, _P, _P, _P]),
    'smx_zfilter_forward_f32': (c_int32, [_P, c_int64, c_int64, c_int32, _P, _P, _P, _P]),
    'smx_zfilter_forward_sums_f32': (c_int32, [_P, c_int64, c_int64, c_int32, _P, _P, _P, c_float, _P, _P]),
    'smx_diaggauss_sample_f32': (c_int32, [_P, c_int64, _P, _P, _P, c_int64, c_int64, c_int32, _P,
                                           c_int64, _P, c_int64, _P]),
    'smx_zfilter_update_f32': (c_int32, [_P, c_int64, c_int64, c_int32, _P, _P, _P, c_float, _P]),
    'smx_mlp3_packed_bytes': (c_size_t, [c_int32, c_int32, c_int32, c_int32]),
    'smx_mlp3_pack_f32': (c_int32, [POINTER(Mlp3), _P, c_size_t, _P]),
    'smx_mlp3_pack_zstats_f32': (c_int32, [POINTER(Mlp3), _P, c_size_t, _P, _P, _P, c_int32, c_float, _P, _P, _P]),
    'smx_mlp3_fused_exact_zfilter': (c_int32, [c_int32]),
    'smx_mlp3_forward_fused_f32': (c_int32, [_P, c_int32, c_int32, c_int32, c_int32, _P, _P,
                                             c_int64, c_int32, c_int32, _P, _P, _P, c_int32, _P]),
    'smx_linear_f32': (c_int32, [_P, c_int32, c_int32, _P, c_int32, c_int32, _P, _P, c_int32,
                                 c_int32, c_int32, c_int32, c_int32, _P, _P, _P]),
    'smx_linear_wgrad_f32': (c_int32, [_P, c_int32, _P, c_int32, _P, c_int32, _P, c_int32, c_int32,
                                       c_int32, _P]),
    'smx_mlp3_forward_f32': (c_int32, [POINTER(Mlp3), _P, c_int64, _P, _P, _P, c_int32, _P, _P]),
    'smx_mlp3_forward_multi_f32': (c_int32, [POINTER(Mlp3Job), c_int32, _P]),
    'smx_mlp3_backward_multi_f32': (c_int32, [POINTER(Mlp3Job), c_int32, _P]),
    'smx_mlp3_wgrad_multi_f32': (c_int32, [POINTER(Mlp3Job), c_int32, _P]),
    'smx_epoch_blocks': (c_int32, [c_int64]),
    'smx_epoch_supported': (c_int32, [c_int32, c_int32, c_int32, c_int32]),
    'smx_epoch_packed_floats': (c_int64, [c_int32, c_int32, c_int32, c_int32]),
    'smx_epoch_pack_f32': (c_int32, [POINTER(EpochPack), c_int32, _P]),
    'smx_epoch_prepare_f32': (c_int32, [POINTER(EpochPrep), _P]),
    'smx_epoch_forward_f32': (c_int32, [POINTER(EpochJob), c_int32, POINTER(PpoLosses), _P, c_int64, _P]),
    'smx_epoch_backward_f32': (c_int32, [POINTER(EpochJob), c_int32, POINTER(PpoLosses), _P, c_int64, _P]),
    'smx_epoch_fwdbwd_supported': (c_int32, [c_int32, c_int32, c_int32, c_int32]),
    'smx_epoch_fwdbwd_f32': (c_int32, [POINTER(EpochJob), c_int32, POINTER(PpoLosses), _P, c_int64, _P, _P, _P]),
    'smx_device_occupy': (c_int32, [c_int32, c_int64, _P]),
    'smx_layernorm_forward_f32': (c_int32, [_P, c_int64, c_int64, c_int32, _P, _P, c_float, _P, c_int64, _P, _P, _P]),
    'smx_layernorm_backward_ws_floats': (c_int64, [c_int64, c_int32]),
    'smx_layernorm_backward_f32': (c_int32, [_P, c_int64, _P, c_int64, _P, _P, _P, c_int64, c_int32, c_int32, _P, c_int64, _P, _P,
                                             _P, c_int64, _P]),
    'smx_zfilter_update_ws_floats': (c_int64, [c_int64, c_int32]),
    'smx_zfilter_update_ws_f32': (c_int32, [_P, c_int64, c_int64, c_int32, _P, _P, _P, c_float, _P, c_int64, _P]),
    'smx_ppo_partials_fold_f32': (c_int32, [_P, c_int32, c_int32, _P, c_int32, _P, _P]),
    'smx_mlp3_forward_rows_supported': (c_int32, [c_int32, c_int32, c_int32, c_int32]),
    'smx_mlp3_forward_rows_f32': (c_int32, [POINTER(Mlp3), _P, c_int64, _P, _P, _P, c_int32, c_int32, _P, c_size_t, _P, _P]),
    'smx_mlp3_backward_partials': (c_int32, [c_int32, c_int32, c_int32, c_int32]),
    'smx_mlp3_backward_f32': (c_int32, [POINTER(Mlp3), _P, _P, _P, _P, c_int64, _P, _P, _P, _P,
                                        _P, _P]),
    'smx_mlp3_backward_ws_floats': (c_int64, [c_int32, c_int32, c_int32, c_int32, c_int64]),
    'smx_mlp3_backward_splitk_f32': (c_int32, [POINTER(Mlp3), _P, _P, _P, _P, c_int64, _P, _P, _P, _P, c_int64, _P, _P]),
    'smx_mlp3_dgrad_rows_supported': (c_int32, [c_int32, c_int32, c_int32, c_int32]),
    'smx_mlp3_dgrad_rows_ws_floats': (c_int64, [c_int32, c_int32, c_int32, c_int32]),
    'smx_mlp3_backward_rows_f32': (c_int32, [POINTER(Mlp3), _P, _P, _P, _P, c_int64, _P, _P, _P, _P, _P, c_int64, _P, c_int64,
                                             _P, _P]),
    'smx_windowed_gae_returns_f32': (c_int32, [_P, _P, _P, _P, _P, _P, c_float, c_float, c_int32,
                                               c_int32, c_int32, _P, _P, _P]),
    'smx_windowed_gae_norm_f32': (c_int32, [_P, _P, _P, _P, _P, _P, c_float, c_float, c_int32, c_int32, c_int32, _P, _P, _P,
                                            c_float, _P, _P]),
    'smx_ppo_learn_epilogue_f32': (c_int32, [POINTER(LearnEpilogue), _P]),
    'smx_reward_filter_f32': (c_int32, [_P, c_int64, c_float, c_int32, _P, c_float, c_int32, _P, _P, _P, _P, _P]),
    'smx_reward_filter_partials': (c_int32, []),
    'smx_moments_f32': (c_int32, [_P, c_int64, _P, _P]),
    'smx_moments_merge_f32': (c_int32, [_P, c_int32, _P, _P]),
    'smx_adv_normalize_f32': (c_int32, [_P, c_int64, _P, c_float, _P]),
    'smx_ppo_loss_blocks': (c_int32, [c_int64]),
    'smx_ppo_loss_partial_stride': (c_int32, [c_int32]),
    'smx_ppo_policy_loss_f32': (c_int32, [c_int32, _P, _P, _P, c_int32, _P, c_int32, _P, c_int32,
                                          _P, c_int64, c_int32, _P, _P, _P, _P, _P]),
    'smx_ppo_loss_finalize_f32': (c_int32, [c_int32, _P, c_int32, _P, _P, _P, c_int64, c_int64,
                                            c_int32, _P, c_int32, c_int32, _P, _P, c_int64, _P, _P,
                                            _P, _P]),
    'smx_ppo_epoch_losses_f32': (c_int32, [POINTER(PpoLosses), _P, _P]),
    'smx_ppo_epoch_losses_dp_f32': (c_int32, [POINTER(PpoLosses), c_int64, _P, _P, _P, _P]),
    'smx_ppo_epoch_combine_f32': (c_int32, [POINTER(PpoCombine), _P, _P]),
    'smx_ppo_final_stats_f32': (c_int32, [_P, c_int32, _P, _P, _P, c_int32, _P, _P]),
    'smx_value_loss_blocks': (c_int32, [c_int64]),
    'smx_value_loss_f32': (c_int32, [_P, _P, c_int64, c_int64, _P, _P, _P, c_int32, _P]),
    'smx_value_loss_finalize_f32': (c_int32, [_P, c_int32, c_int32, _P, c_int32, _P]),
    'smx_clip_adam_step_f32': (c_int32, [_P, _P, _P, _P, c_int64, _P, c_int32, _P, c_int32,
                                         c_int32, _P, _P]),
    'smx_clip_adam_step_group_f32': (c_int32, [POINTER(AdamGroup), c_int32, _P, _P]),
    'smx_clip_adam_step_pair_f32': (c_int32, [POINTER(AdamGroup), POINTER(AdamGroup), _P, _P]),
    'smx_sumsq_blocks': (c_int32, [c_int64]),
    'smx_sumsq_partials_f32': (c_int32, [_P, c_int64, _P, _P]),
    'smx_synth_act_env_step_head_f32': (c_int32, [POINTER(SynthActStep), _P, _P, _P, c_int64, c_int32, c_int32, _P]),
    'smx_conv_u8_forward_f32': (c_int32, [_P, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32, _P, _P, c_int32,
                                          _P, _P, _P]),
    'smx_conv_u8_wgrad_ws_floats': (c_int64, [c_int32, c_int32]),
    'smx_conv_u8_wgrad_f32': (c_int32, [_P, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32, _P, c_int32, _P, _P, _P,
                                        c_int64, _P, _P]),
    'smx_conv_cl_forward_f32': (c_int32, [_P, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32, _P, _P, c_int32,
                                          _P, _P, _P]),
    'smx_conv_cl_wgrad_ws_floats': (c_int64, [c_int32, c_int32]),
    'smx_conv_cl_wgrad_f32': (c_int32, [_P, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32, _P, c_int32, _P, _P, _P,
                                        c_int64, _P, _P]),
    'smx_conv_cl_dgrad_f32': (c_int32, [_P, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32, _P, c_int32, _P, _P,
                                        _P, _P]),
    'smx_ring_insert_bytes': (c_int32, [_P, c_int64, c_int64, c_int64, _P, c_int64, _P]),
    'smx_gather_rows_bytes': (c_int32, [_P, c_int64, c_int64, _P, c_int64, _P, _P]),
    'smx_window_emit_bytes': (c_int32, [_P, c_int32, c_int32, c_int64, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    'smx_ring_insert_f32': (c_int32, [_P, c_int64, c_int32, c_int64, _P, c_int64, _P]),
    'smx_gather_rows_f32': (c_int32, [_P, c_int64, c_int32, _P, c_int64, _P, _P]),
    'smx_philox4x32_10': (c_int32, [_P, c_int64, _P, _P]),
    'smx_uniform_indices': (c_int32, [_P, c_int64, c_int64, c_uint64, c_uint64, _P]),
    'smx_uniform_gather_multi': (c_int32, [POINTER(GatherJob), c_int32, c_int64, c_int64, _P, c_int64, c_uint64, c_uint64,
                                           _P, _P]),
    'smx_window_emit_f32': (c_int32, [_P, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                      c_int32, _P, _P]),
    'smx_synth_act_env_step_f32': (c_int32, [POINTER(SynthActStep), _P]),
    'smx_synth_env_step_f32': (c_int32, [_P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_int32,
                                         c_int32, c_int32, _P, _P, _P, _P, _P]),
    'smx_ddpg_critic_loss_f32': (c_int32, [_P, _P, _P, _P, c_float, c_int64, _P, _P, _P]),
    'smx_tanh_backward_f32': (c_int32, [_P, _P, c_int64, _P, _P]),
    'smx_fill_f32': (c_int32, [_P, c_int64, c_float, _P]),
    'smx_adam_step_f32': (c_int32, [_P, _P, _P, _P, c_int64, c_double, c_int32, c_double, c_double,
                                    _P]),
    'smx_soft_update_f32': (c_int32, [_P, _P, c_float, c_int64, _P]),
    'smx_ddpg_critic_loss_step_f32': (c_int32, [_P, _P, _P, _P, c_float, c_int64, _P, _P, _P, _P]),
    'smx_adam_step_dev_f32': (c_int32, [_P, _P, _P, _P, c_int64, _P, _P, c_double, c_double, _P]),
    'smx_hard_update_every_f32': (c_int32, [_P, _P, c_int64, _P, c_int32, _P]),
    'smx_ddpg_rows_supported': (c_int32, [c_int32] * 6),
    'smx_ddpg_rows_supported_at': (c_int32, [c_int32] * 6 + [c_int64]),
    'smx_ddpg_rows_packed_floats': (c_int64, [c_int32] * 6),
    'smx_ddpg_rows_pack_f32': (c_int32, [_P, c_int32, _P]),
    'smx_ddpg_rows_critic_f32': (c_int32, [_P, _P]),
    'smx_ddpg_rows_actor_f32': (c_int32, [_P, _P]),
    'smx_ddpg_rows_update_f32': (c_int32, [_P, c_int32, _P, _P]),
    'smx_ddpg_rows_wgrad_update_f32': (c_int32, [_P, c_int32, _P, _P]),
    'smx_ddpg_stats_f32': (c_int32, [_P, _P, _P, _P, c_int32, c_int32, _P, c_int64, _P, _P]),
    'smx_lstm_param_count': (c_int64, [c_int32, c_int32]),
    'smx_lstm_forward_f32': (c_int32, [POINTER(Lstm), _P, c_int64, c_int32, _P, _P, _P, _P, _P, _P,
                                       _P, _P, _P, _P]),
    'smx_lstm_backward_f32': (c_int32, [POINTER(Lstm), _P, c_int64, c_int32, _P, _P, _P, _P, _P, _P,
                                        _P, _P, _P, c_int64, _P]),
    'smx_lstm_backward_ws_floats': (c_int64, [c_int32, c_int32, c_int64, c_int32]),
    'smx_linear_wgrad_ws_floats': (c_int64, [c_int32, c_int32, c_int32]),
    'smx_linear_wgrad_splitk_f32': (c_int32, [_P, c_int32, _P, c_int32, _P, c_int32, _P, c_int32,
                                              c_int32, c_int32, _P, c_int64, _P]),
    'smx_linear_wgrad_splitk_pair_f32': (c_int32, [_P, c_int32, c_int32, c_int32, _P, c_int32, _P, _P, c_int32, _P,
                                                    c_int32, _P, _P, c_int32, _P, c_int64, _P]),
    'smx_im2col_f32': (c_int32, [_P, c_int32, c_int32, c_int64, c_int32, c_int32, c_int32, c_int32,
                                 c_int32, c_int32, c_float, _P, _P]),
    'smx_col2im_f32': (c_int32, [_P, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                 _P, _P, _P]),
    'smx_flatten_order_f32': (c_int32, [_P, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    'smx_linear_multi_f32': (c_int32, [POINTER(LinearJob), c_int32, _P]),
    'smx_frame_stack_u8': (c_int32, [_P, c_int32, c_int32, c_int64, c_int32, _P, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    'smx_synth_frame_u8': (c_int32, [_P, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32, _P, c_int64, _P]),
    'smx_synth_rollout_supported': (c_int32, [c_int32, c_int32, c_int32, c_int32]),
    'smx_synth_rollout_f32': (c_int32, [POINTER(SynthRollout), _P]),
    'smx_xchg_bytes': (c_int64, [c_int64, c_int32]),
    'smx_xchg_alloc': (c_int32, [c_int64, c_double, POINTER(c_void_p), POINTER(c_int32), _P]),
    'smx_xchg_free': (c_int32, [_P]),
    'smx_xchg_export': (c_int32, [_P, _P]),
    'smx_xchg_open': (c_int32, [_P, POINTER(c_void_p)]),
    'smx_xchg_close': (c_int32, [_P]),
    'smx_xchg_allreduce_f32': (c_int32, [POINTER(Xchg), _P, _P, c_int64, _P, _P]),
    'smx_xchg_allgather_f32': (c_int32, [POINTER(Xchg), _P, c_int64, _P, _P, _P]),
    'smx_xchg_status': (c_int32, [POINTER(Xchg), _P, _P]),
}

EXPORTED_SYMBOLS = tuple(sorted(_SIGS))


class SmxError(RuntimeError):
    pass


_lib = None


def load():
    """dlopen the library and bind every symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SmxError(
            'libsurreal_amd.so not found at %s -- build it with `python -m surreal_amd.build` '
            '(hipcc, gfx950).  There is no CPU fallback.' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=''):
    if rc != 0:
        msg = load().smx_error_string(int(rc))
        raise SmxError('%s failed: rc=%d (%s)' % (what, rc, msg.decode() if msg else '?'))


def call(name, *args):
    rc = getattr(load(), name)(*args)
    check(rc, name)


def ptr(t):
    """device pointer of a torch tensor (None -> NULL)"""
    if t is None:
        return None
    return c_void_p(t.data_ptr())


_raw_stream = None


def current_stream():
    """the HIP stream torch's launches would go to now (torch.cuda.stream() contexts included).  Through the raw-handle
    call: torch.cuda.current_stream() builds a Stream object through four layers of device-index look-ups, 9 us per launch
    -- a tenth of a DDPG iteration"""
    global _raw_stream
    import torch
    if _raw_stream is None:
        _raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', False)
    if _raw_stream:
        return c_void_p(_raw_stream(torch.cuda.current_device()))
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise SmxError('surreal_amd needs a ROCm GPU (gfx950): torch.cuda.is_available() is False '
                       'and there is no CPU fallback')
    load()
