"""ctypes binding of the C-ABI HIP library (``include/cookietts_hip.h``).

The product path has no CPU fallback: if ``libcookietts_hip.so`` is missing or fails to
load, :func:`lib` raises, and every op built on it raises with it.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

from . import build as _build

_LOCK = threading.Lock()
_LIB = None


class WaveGlowConfig(C.Structure):
    """``ctts_waveglow_config`` (include/cookietts_hip.h)."""
    _fields_ = [(n, C.c_int32) for n in (
        "n_mel_channels", "n_group", "n_flows", "n_early_every", "n_early_size",
        "win_length", "hop_length", "n_layers", "n_channels", "kernel_size", "cond_hidden", "speaker_embed_dim",
        "f32_gemm_mode")]


class StftConfig(C.Structure):
    """``ctts_stft_config``."""
    _fields_ = [("filter_length", C.c_int32), ("hop_length", C.c_int32), ("win_length", C.c_int32),
                ("n_mel_channels", C.c_int32), ("clamp_val", C.c_float)]


class WaveGlowGeometry(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("steps", "ld", "pad", "n_remaining")]


_FP = C.c_void_p  # device pointers travel as integers


class WaveGlowFlowWeights(C.Structure):
    _fields_ = [
        ("start_w", _FP), ("start_b", _FP),
        ("cond_w", _FP * 3), ("cond_b", _FP * 3),
        ("in_w", C.POINTER(_FP)), ("in_b", C.POINTER(_FP)),
        ("rs_w", C.POINTER(_FP)), ("rs_b", C.POINTER(_FP)),
        ("end_w", _FP), ("end_b", _FP), ("w_inverse", _FP), ("speaker_embed", _FP),
    ]


class WaveFlowConfig(C.Structure):
    """``ctts_waveflow_config``."""
    _fields_ = [(n, C.c_int32) for n in ("n_mel_channels", "n_flows", "n_group", "n_layers", "n_channels",
                                         "kernel_size_w", "kernel_size_h", "dilation_h", "seperable_conv",
                                         "cond_precomputed", "gated_unit", "merge_res_skip", "n_early_every", "n_early_size",
                                         "mixing", "mix_first")] + [("dilation_w", C.c_int32 * 12),
                                                                     ("dilation_h_l", C.c_int32 * 12),
                                                                     ("f32_gemm_mode", C.c_int32)]


class WaveFlowFlowWeights(C.Structure):
    _fields_ = [("start_w", _FP), ("start_b", _FP), ("cond_w", _FP), ("cond_b", _FP),
                ("in_w", C.POINTER(_FP)), ("in_b", C.POINTER(_FP)), ("rs_w", C.POINTER(_FP)), ("rs_b", C.POINTER(_FP)),
                ("end_w", _FP), ("end_b", _FP), ("dw_w", C.POINTER(_FP)), ("dw_b", C.POINTER(_FP)), ("w_inverse", _FP)]


class WgaxConfig(C.Structure):
    """``ctts_wgax_config`` (ax core, waveflow=False)."""
    _fields_ = [(n, C.c_int32) for n in ("n_flows", "n_group", "n_early_every", "n_early_size", "n_layers",
                                         "n_channels", "kernel_size", "mixing", "mix_first", "ignore_nan", "gated_unit",
                                         "merge_res_skip")] + [("dilation_w", C.c_int32 * 12), ("f32_gemm_mode", C.c_int32)]


def dilation_array(spec, n_layers):
    """WN_config['n_layers_dilations_w'] (None | int | list, glow_ax.py:328-333) -> the config structs' int32[12]
    (0 = the default 2^i)."""
    arr = (C.c_int32 * 12)()
    if spec is not None:
        vals = [spec] * n_layers if isinstance(spec, int) else list(spec)
        for i in range(n_layers):
            arr[i] = int(vals[i])
    return arr


# WN_config['gated_unit'] -> CTTS_GATE_* (get_gate_func, glow_ax.py:168-198; the reference upper-cases the name)
GATED_UNITS = {n: i for i, n in enumerate(("GTU", "GTRU", "GTLRU", "GLU", "TTU", "STU", "GTSU", "SPTU", "GSIU", "GSIRU",
                                           "GTSRU", "GSIRRU", "GSIRLRU", "GSIRRLRU"))}


class WgaxFlowWeights(C.Structure):
    _fields_ = [("start_w", _FP), ("start_b", _FP), ("in_w", C.POINTER(_FP)), ("in_b", C.POINTER(_FP)),
                ("rs_w", C.POINTER(_FP)), ("rs_b", C.POINTER(_FP)), ("end_w", _FP), ("end_b", _FP), ("w_inverse", _FP)]


MIX_PERMUTE, MIX_CONV1X1 = 0, 1
N_SPEAKERS = 512       # CTTS_N_SPEAKERS


class TacoDecoderConfig(C.Structure):
    """``ctts_taco_decoder_config``."""
    _fields_ = [(n, C.c_int32) for n in (
        "n_mel_channels", "memory_in_dim", "memory_dim", "attention_dim", "attention_rnn_dim", "decoder_rnn_dim",
        "second_decoder_rnn_dim", "prenet_dim", "location_n_filters", "location_kernel_size", "window_range")]


class LstmWeights(C.Structure):
    _fields_ = [("w_ih", _FP), ("w_hh", _FP), ("b_ih", _FP), ("b_hh", _FP)]


class TacoDecoderWeights(C.Structure):
    _fields_ = [("bottleneck_w", _FP), ("memory_layer_w", _FP), ("query_w", _FP), ("v_w", _FP), ("loc_conv_w", _FP),
                ("loc_dense_w", _FP), ("prenet_w1", _FP), ("prenet_w2", _FP),
                ("att_rnn", LstmWeights), ("dec_rnn", LstmWeights), ("dec2_rnn", LstmWeights),
                ("proj_w", _FP), ("proj_b", _FP), ("gate_w", _FP), ("gate_b", _FP),
                ("windowed_att_pos_offset", C.c_float), ("exp_smoothing_factor", C.c_float)]


class Conv1dDesc(C.Structure):
    """``ctts_conv1d_desc``."""
    _fields_ = [("c_in", C.c_int32), ("c_out", C.c_int32), ("kernel_size", C.c_int32), ("act", C.c_int32),
                ("slope", C.c_float), ("f32_gemm_mode", C.c_int32)]


class TacoMemoryWeights(C.Structure):
    _fields_ = [(n, _FP) for n in ("sylps_w", "sylps_b", "speaker_embedding", "syl_w0", "syl_b0", "syl_w2", "syl_b2",
                                   "syl_res_weight", "tm_gamma", "tm_beta", "tm_mean", "tm_var", "tm_w", "tm_b")]


# name -> (restype, argtypes); kept in one table so tests can check every symbol the
# header declares is exported.
_CFG = C.POINTER(WaveGlowConfig)
SIGNATURES = {
    "ctts_abi_version": (C.c_int, []),
    "ctts_last_error": (C.c_char_p, []),
    "ctts_waveglow_geometry_for": (C.c_int, [_CFG, C.c_int32, C.POINTER(WaveGlowGeometry)]),
    "ctts_fold_weightnorm_f32": (C.c_int, [_FP, _FP, _FP, C.c_int32, C.c_int32, _FP]),
    "ctts_waveglow_packed_bytes": (C.c_size_t, [_CFG]),
    "ctts_waveglow_pack_upsample": (C.c_int, [_CFG, _FP, _FP, _FP, _FP]),
    "ctts_waveglow_pack_flow": (C.c_int, [_CFG, C.c_int32, C.POINTER(WaveGlowFlowWeights), _FP, _FP]),
    "ctts_waveglow_workspace_bytes": (C.c_size_t, [_CFG, C.c_int32, C.c_int32]),
    "ctts_waveglow_infer_f32": (C.c_int, [_CFG, _FP, _FP, _FP, _FP, C.c_int32, C.c_int32, _FP, C.c_size_t, _FP]),
    "ctts_waveglow_infer_spk_f32": (C.c_int, [_CFG, _FP, _FP, _FP, _FP, _FP, C.c_int32, C.c_int32, _FP, C.c_size_t, _FP]),
    "ctts_waveglow_infer_spk_bf16": (C.c_int, [_CFG, _FP, _FP, _FP, _FP, _FP, _FP, C.c_int32, C.c_int32, _FP, C.c_size_t,
                                              _FP]),
    "ctts_waveglow_infer_spk_bf16x3": (C.c_int, [_CFG, _FP, _FP, _FP, _FP, _FP, _FP, C.c_int32, C.c_int32, _FP, C.c_size_t,
                                               _FP]),
    "ctts_waveglow_infer_spk_f16": (C.c_int, [_CFG, _FP, _FP, _FP, _FP, _FP, _FP, C.c_int32, C.c_int32, _FP, C.c_size_t,
                                             _FP]),
    "ctts_waveglow_pack_flow_f16": (C.c_int, [_CFG, C.c_int32, C.POINTER(WaveGlowFlowWeights), _FP, _FP]),
    "ctts_waveglow_packed_bf16x3_bytes": (C.c_size_t, [_CFG]),
    "ctts_waveglow_pack_flow_bf16x3": (C.c_int, [_CFG, C.c_int32, C.POINTER(WaveGlowFlowWeights), _FP, _FP]),
    "ctts_waveglow_workspace_bf16x3_bytes": (C.c_size_t, [_CFG, C.c_int32, C.c_int32]),
    "ctts_waveglow_packed_bf16_bytes": (C.c_size_t, [_CFG]),
    "ctts_waveglow_pack_flow_bf16": (C.c_int, [_CFG, C.c_int32, C.POINTER(WaveGlowFlowWeights), _FP, _FP]),
    "ctts_waveglow_workspace_bf16_bytes": (C.c_size_t, [_CFG, C.c_int32, C.c_int32]),
    "ctts_waveglow_infer_bf16": (C.c_int, [_CFG, _FP, _FP, _FP, _FP, _FP, C.c_int32, C.c_int32, _FP, C.c_size_t, _FP]),
    "ctts_upsample_squeeze_f32": (C.c_int, [_CFG, _FP, _FP, _FP, C.c_int32, C.c_int32, _FP]),
    "ctts_wn_cond_f32": (C.c_int, [_CFG, _FP, _FP, _FP, _FP, _FP, C.c_int32, C.c_int32, _FP]),
    "ctts_wn_stack_f32": (C.c_int, [_CFG, _FP, C.c_int32, _FP, _FP, _FP, _FP, _FP, C.c_int32, C.c_int32, _FP]),
    "ctts_flow_tail_f32": (C.c_int, [_CFG, _FP, C.c_int32, _FP, _FP, _FP, C.c_int32, C.c_int32, _FP]),
    "ctts_waveflow_packed_bytes": (C.c_size_t, [C.POINTER(WaveFlowConfig)]),
    "ctts_waveflow_pack_flow": (C.c_int, [C.POINTER(WaveFlowConfig), C.c_int32, C.POINTER(WaveFlowFlowWeights), _FP, _FP]),
    "ctts_waveflow_workspace_bytes": (C.c_size_t, [C.POINTER(WaveFlowConfig), C.c_int32, C.c_int32]),
    "ctts_waveflow_inverse_f32": (C.c_int, [C.POINTER(WaveFlowConfig), _FP, _FP, _FP, _FP, C.c_int32, C.c_int32,
                                            C.c_int32, _FP, C.c_size_t, _FP]),
    "ctts_taco_decoder_packed_bytes": (C.c_size_t, [C.POINTER(TacoDecoderConfig)]),
    "ctts_taco_decoder_pack": (C.c_int, [C.POINTER(TacoDecoderConfig), C.POINTER(TacoDecoderWeights), _FP, _FP]),
    "ctts_taco_decoder_max_batch": (C.c_int32, [C.POINTER(TacoDecoderConfig)]),
    "ctts_taco_decoder_workspace_bytes": (C.c_size_t, [C.POINTER(TacoDecoderConfig), C.c_int32, C.c_int32]),
    "ctts_taco_decoder_init_f32": (C.c_int, [C.POINTER(TacoDecoderConfig), _FP, _FP, _FP, C.c_int32, C.c_int32, _FP,
                                             C.c_size_t, _FP]),
    "ctts_taco_decoder_steps_f32": (C.c_int, [C.POINTER(TacoDecoderConfig), _FP, _FP, _FP, _FP, _FP, C.c_int32,
                                              C.c_int32, C.c_int32, C.c_int32, C.c_int32, _FP, _FP]),
    "ctts_taco_decoder_steps_hidden_f32": (C.c_int, [C.POINTER(TacoDecoderConfig), _FP, _FP, _FP, _FP, _FP, _FP, C.c_int32,
                                                     C.c_int32, C.c_int32, C.c_int32, C.c_int32, _FP, _FP]),
    "ctts_taco_decoder_persistent_bytes": (C.c_size_t, [C.POINTER(TacoDecoderConfig), C.c_int32, C.c_int32]),
    "ctts_taco_decoder_steps_persistent_f32": (C.c_int, [C.POINTER(TacoDecoderConfig), _FP, _FP, _FP, _FP, _FP, C.c_int32,
                                                         C.c_int32, C.c_int32, C.c_int32, C.c_int32, _FP, _FP, C.c_size_t,
                                                         _FP]),
    "ctts_taco_decoder_persistent_debug": (C.c_int, [_FP]),
    "ctts_conv1d_packed_bytes": (C.c_size_t, [C.POINTER(Conv1dDesc)]),
    "ctts_conv1d_pack_f32": (C.c_int, [C.POINTER(Conv1dDesc), _FP, _FP, _FP, _FP, _FP, _FP, C.c_float, _FP, _FP]),
    "ctts_conv1d_f32": (C.c_int, [C.POINTER(Conv1dDesc), _FP, _FP, _FP, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                  C.c_int32, _FP]),
    "ctts_lstm_seq_packed_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "ctts_lstm_seq_pack_f32": (C.c_int, [C.POINTER(LstmWeights), C.c_int32, C.c_int32, _FP, _FP]),
    "ctts_lstm_seq_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "ctts_lstm_seq_f32": (C.c_int, [_FP, _FP, _FP, C.c_int32, _FP, C.c_int64, C.c_int32, C.c_int32, _FP, C.c_int32,
                                    C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _FP,
                                    C.c_size_t, _FP]),
    "ctts_lstm_biseq_f32": (C.c_int, [_FP, _FP, _FP, _FP, _FP, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _FP, C.c_int32,
                                      C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                      _FP, _FP, C.c_size_t, _FP]),
    "ctts_taco_embed_f32": (C.c_int, [_FP, _FP, _FP, _FP, _FP, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_int32, _FP]),
    "ctts_taco_memory_f32": (C.c_int, [C.POINTER(TacoMemoryWeights), _FP, _FP, _FP, _FP, _FP, C.c_int32, C.c_int32,
                                       C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _FP]),
    "ctts_taco_memory_sylps_f32": (C.c_int, [C.POINTER(TacoMemoryWeights), _FP, _FP, _FP, _FP, _FP, _FP, C.c_int32, C.c_int32,
                                             C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _FP]),
    "ctts_pad_rows_f32": (C.c_int, [_FP, C.c_int64, C.c_int32, _FP, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                    C.c_int32, _FP]),
    "ctts_unpad_rows_f32": (C.c_int, [_FP, _FP, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_int32, _FP]),
    "ctts_stft_packed_bytes": (C.c_size_t, [C.POINTER(StftConfig)]),
    "ctts_stft_pack": (C.c_int, [C.POINTER(StftConfig), _FP, _FP, _FP, _FP]),
    "ctts_stft_workspace_bytes": (C.c_size_t, [C.POINTER(StftConfig), C.c_int32, C.c_int32]),
    "ctts_stft_mel_f32": (C.c_int, [C.POINTER(StftConfig), _FP, _FP, _FP, _FP, C.c_int32, C.c_int32, _FP,
                                    C.c_size_t, _FP]),
    "ctts_stft_pack_inverse": (C.c_int, [C.POINTER(StftConfig), _FP, _FP, _FP, _FP]),
    "ctts_stft_transform_f32": (C.c_int, [C.POINTER(StftConfig), _FP, _FP, _FP, _FP, C.c_int32, C.c_int32, _FP,
                                          C.c_size_t, _FP]),
    "ctts_stft_inverse_f32": (C.c_int, [C.POINTER(StftConfig), _FP, _FP, _FP, _FP, C.c_float, _FP, C.c_int32,
                                        C.c_int32, _FP, C.c_size_t, _FP]),
    "ctts_stft_inverse_bias_f32": (C.c_int, [C.POINTER(StftConfig), _FP, _FP, _FP, _FP, C.c_int32, C.c_float, _FP,
                                             C.c_int32, C.c_int32, _FP, C.c_size_t, _FP]),
    "ctts_waveflow_inverse_cond_f32": (C.c_int, [C.POINTER(WaveFlowConfig), _FP, _FP, _FP, C.c_int32, C.c_int32, _FP,
                                                 C.c_int32, C.c_int32, C.c_int32, _FP, C.c_size_t, _FP]),
    "ctts_waveflow_abort_status": (C.c_int, [C.POINTER(WaveFlowConfig), C.c_int32, C.c_int32, _FP, C.c_size_t, _FP]),
    "ctts_wgax_packed_bytes": (C.c_size_t, [C.POINTER(WgaxConfig)]),
    "ctts_wgax_pack_flow": (C.c_int, [C.POINTER(WgaxConfig), C.c_int32, C.POINTER(WgaxFlowWeights), _FP, _FP]),
    "ctts_wgax_workspace_bytes": (C.c_size_t, [C.POINTER(WgaxConfig), C.c_int32, C.c_int64]),
    "ctts_wgax_inverse_f32": (C.c_int, [C.POINTER(WgaxConfig), _FP, _FP, _FP, C.c_int32, C.c_int32, C.c_int32, _FP,
                                        C.c_int32, C.c_int64, _FP, C.c_size_t, _FP]),
    "ctts_replicate_halo_f32": (C.c_int, [_FP, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _FP]),
    "ctts_embed_rows_f32": (C.c_int, [_FP, _FP, _FP, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_int32, _FP]),
    "ctts_scale_add_rows_f32": (C.c_int, [_FP, _FP, _FP, _FP, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                          _FP]),
    "ctts_deemphasis_f32": (C.c_int, [_FP, _FP, C.c_int32, C.c_int32, C.c_double, _FP]),
    "ctts_alignment_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "ctts_alignment_metric_f32": (C.c_int, [_FP, _FP, _FP, C.c_int32, C.c_int32, C.c_int32, C.c_float, _FP, _FP,
                                            C.c_size_t, _FP]),
    "ctts_first_over_thresh_f32": (C.c_int, [_FP, C.c_int32, C.c_int32, C.c_float, _FP, _FP]),
    "ctts_taco_stop_state_bytes": (C.c_size_t, [C.c_int32]),
    "ctts_taco_stop_reset": (C.c_int, [_FP, C.c_int32, C.c_int32, _FP]),
    "ctts_taco_stop_rule_f32": (C.c_int, [_FP, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, _FP, _FP]),
    "ctts_affine_rows_f32": (C.c_int, [_FP, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                       C.c_float, _FP]),
    "ctts_vol_unscale_f32": (C.c_int, [_FP, C.c_int64, _FP]),
    "ctts_resample_rows_f32": (C.c_int, [_FP, _FP] + [C.c_int32] * 9 + [C.c_float, _FP]),
    "ctts_interleave_phases_f32": (C.c_int, [_FP, _FP] + [C.c_int32] * 10 + [_FP]),
    "ctts_last_gemm_loop": (C.c_int, []),
    "ctts_tuning_reload": (C.c_int, []),
    "ctts_tuning_flags": (C.c_int, []),
    "ctts_profile_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "ctts_profile_bind": (C.c_int, [C.c_void_p]),
    "ctts_profile_collect": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "ctts_profile_destroy": (C.c_int, [C.c_void_p]),
}

# CTTS_GEMM_*: the f32_gemm_mode field of the config structs (the mode a MODEL asks for).  There is no process-wide default
# any more (ABI 6): "default" / None = fp32 MFMA.
MODEL_GEMM_MODES = {None: 0, "default": 0, "f32": 1, "bf16x3": 2, "bf16x6": 3}
GEMM_MODES = {"f32": 1, "bf16x3": 2, "bf16x6": 3}


def model_gemm_mode(mode):
    """``None`` / ``"default"`` (fp32 MFMA), ``"f32"``, ``"bf16x3"`` or ``"bf16x6"`` -> CTTS_GEMM_*."""
    try:
        return MODEL_GEMM_MODES[mode]
    except KeyError:
        raise ValueError(f"f32 GEMM mode {mode!r}: expected one of 'default', 'f32', 'bf16x3', 'bf16x6'") from None


def set_f32_gemm_mode(mode):
    """Removed with ABI 6: the process-wide default main loop was hidden state shared by every model and thread.  Choose
    per model: ``model.set_f32_gemm_mode("bf16x6")`` (it travels in the model's config struct)."""
    if mode in (None, "default", "f32"):
        return "f32"
    raise RuntimeError(f"cookietts_amd.set_f32_gemm_mode({mode!r}): the process-wide default was removed; "
                       f"call model.set_f32_gemm_mode({mode!r}) on each model instead")


class Profile:
    """A caller-owned set of kernel-timing slots (``ctts_profile_create``).  Inside ``with profile:`` the calling THREAD's
    WaveGlow launches are bracketed with HIP events on their stream and filed here; ``collect(which)`` -> (launches, total
    ms) and empties the slot.  Two threads with two profiles never see each other's launches."""

    def __init__(self):
        h = C.c_void_p()
        check(lib().ctts_profile_create(C.byref(h)), "ctts_profile_create")
        self._h = h

    def __enter__(self):
        check(lib().ctts_profile_bind(self._h), "ctts_profile_bind")
        return self

    def __exit__(self, *exc):
        check(lib().ctts_profile_bind(None), "ctts_profile_bind")
        return False

    bind = __enter__

    def unbind(self):
        self.__exit__()

    def collect(self, which):
        n, ms = C.c_int64(), C.c_double()
        check(lib().ctts_profile_collect(self._h, which, C.byref(n), C.byref(ms)), "ctts_profile_collect")
        return int(n.value), float(ms.value)

    def close(self):
        if self._h is not None and _LIB is not None:
            _LIB.ctts_profile_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


TUNING_BITS = {"CTTS_F32_NO_GLDS": 0, "CTTS_GEMM_NO_XCD_PAIR": 1, "CTTS_BF16_NO_GLDS": 2, "CTTS_BF16_NO_WIDE": 3,
               "CTTS_BF16_NO_PP": 4, "CTTS_BF16_W4": 5, "CTTS_BF16_PP_STAGES": 6, "CTTS_WF_NO_FUSE": 7, "CTTS_TACO_NO_FUSE": 8,
               "CTTS_F32_NO_SMALL": 9, "CTTS_F32_FORCE_SMALL": 10, "CTTS_F32_NO_SPLITK": 11, "CTTS_WF_NO_VEC_INTERP": 12, "CTTS_F32_NO_DEFER_SKIP": 13, "CTTS_WF_NO_REGION_SPLIT": 14,
               "CTTS_WF_NO_ROW_QUEUE": 15, "CTTS_WF_ROW_QUEUE_MIN": 16, "CTTS_WF_INJECT_ABORT": 17, "CTTS_WF_QUEUE_DEBUG": 18, "CTTS_F32_NO_ROUND_SPLIT": 19, "CTTS_BF16_PS": 20, "CTTS_BF16_NO_PS": 21, "CTTS_F32_SPLITK_W4": 22, "CTTS_TACO_POLL_DELAY": 23, "CTTS_TACO_VALU": 24, "CTTS_UP_NO_MFMA": 25}


def tuning_reload():
    """Re-read the CTTS_* launch-shape knobs from the environment (the library reads them once, at the first launch)."""
    check(lib().ctts_tuning_reload(), "ctts_tuning_reload")


def tuning_active(name):
    """True when the library currently runs with knob ``name`` set (``ctts_tuning_flags``)."""
    return bool(lib().ctts_tuning_flags() >> TUNING_BITS[name] & 1)


PROF_WN_IN = 0
PROF_WN_RS = 1
PROF_WN_SKIP = 2


class HipLibraryError(RuntimeError):
    pass


def lib_path():
    return _build.LIB_PATH


def lib():
    """Load (once) and return the ctypes handle; raises HipLibraryError if unavailable."""
    global _LIB
    if _LIB is not None:
        return _LIB
    with _LOCK:
        if _LIB is not None:
            return _LIB
        path = lib_path()
        if not os.path.exists(path):
            raise HipLibraryError(
                f"{path} is missing: build it with `python -m cookietts_amd.build` "
                "(or __graft_entry__.build()); there is no CPU fallback for the hot path")
        try:
            handle = C.CDLL(path)
        except OSError as e:  # pragma: no cover - depends on the box
            raise HipLibraryError(f"cannot load {path}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:
                raise HipLibraryError(f"{path} does not export {name}") from e
            fn.restype = res
            fn.argtypes = args
        if handle.ctts_abi_version() != 7:
            raise HipLibraryError(f"ABI version mismatch: library reports {handle.ctts_abi_version()}")
        _LIB = handle
    return _LIB


def check(rc, what):
    if rc != 0:
        msg = lib().ctts_last_error()
        raise HipLibraryError(f"{what} failed (rc={rc}): {msg.decode(errors='replace') if msg else ''}")


def ptr(t):
    """Device (or host) address of a torch tensor / None as a ctypes void pointer value."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())
