"""CPU-side checks: the C-ABI library builds, loads and exports what the header declares;
the Python host mirrors the reference's module surface."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import REPO
from cookietts_amd import WaveGlow, _lib, synthetic


def _declared_symbols():
    text = open(os.path.join(REPO, "include", "cookietts_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ctts_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(hip_lib_path):
    declared = _declared_symbols()
    assert len(declared) >= 12
    handle = ctypes.CDLL(hip_lib_path)
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in include/cookietts_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == declared, "ctypes signature table out of sync with the header"
    assert handle.ctts_abi_version() == 7


def test_host_side_queries_run_without_gpu(hip_lib_path):
    """Geometry / size queries are pure host code: callable on the build box."""
    lib = _lib.lib()
    m = WaveGlow(**synthetic.WAVEGLOW_CONFIGS["toy"])
    c = m.c_config()
    geo = _lib.WaveGlowGeometry()
    assert lib.ctts_waveglow_geometry_for(ctypes.byref(c), 900, ctypes.byref(geo)) == 0
    assert geo.steps == 28800 and geo.pad == 128 and geo.ld == 28928 + 256 and geo.n_remaining == 8
    assert lib.ctts_waveglow_packed_bytes(ctypes.byref(c)) > 0
    assert lib.ctts_waveglow_workspace_bytes(ctypes.byref(c), 2, 10) > 0
    bad = m.c_config()
    bad.n_channels = 100
    assert lib.ctts_waveglow_packed_bytes(ctypes.byref(bad)) == 0
    assert b"n_channels" in lib.ctts_last_error()
    bad = m.c_config()
    bad.kernel_size = 5
    assert lib.ctts_waveglow_geometry_for(ctypes.byref(bad), 10, ctypes.byref(geo)) == -1


@pytest.mark.parametrize("key", ["toy", "toy_early", "small", "toy_spk_rezero", "toy_simple"])
def test_state_dict_keys_and_shapes_match_reference_format(key):
    cfg = synthetic.WAVEGLOW_CONFIGS[key]
    sd = synthetic.waveglow_state_dict(cfg, seed=1)
    m = WaveGlow(**cfg)
    own = m.state_dict()
    assert sorted(own) == sorted(sd)
    for k, v in sd.items():
        assert tuple(own[k].shape) == v.shape, k
    m.load_state_dict(synthetic.to_torch(sd))
    assert torch.equal(m.WN[0].start.weight_v, torch.from_numpy(sd["WN.0.start.weight_v"]))
    assert m.n_remaining_channels == synthetic.waveglow_flow_channels(cfg)[-1][0]
    assert m.multispeaker is (cfg["WN_config"]["speaker_embed_dim"] > 0)


def test_default_init_matches_reference_conventions():
    m = WaveGlow(**synthetic.WAVEGLOW_CONFIGS["toy"])
    assert float(m.WN[0].end.weight.abs().max()) == 0.0           # glow.py:141-144
    W = m.convinv[0].conv.weight.squeeze(-1)
    assert torch.allclose(W @ W.t(), torch.eye(W.shape[0]), atol=1e-5) and torch.det(W) > 0   # glow.py:76-83
    g = m.WN[0].start.weight_g.flatten()
    assert torch.allclose(g, m.WN[0].start.weight_v.flatten(1).norm(dim=1))


def test_unsupported_options_fail_loudly():
    cfg = dict(synthetic.WAVEGLOW_CONFIGS["toy"])
    with pytest.raises(ValueError):
        WaveGlow(**dict(cfg, upsample_mode="bilinear"))                   # glow.py:241 prints "invalid" and crashes
    with pytest.raises(NotImplementedError):
        WaveGlow(**dict(cfg, memory_efficient=True))                      # the reference builds no layers (glow.py:263)
    m = WaveGlow(**dict(cfg, spect_scaling=True))
    with pytest.raises(NotImplementedError):
        m.infer_from_noise(torch.zeros(1, 80, 4), torch.zeros(1, 8, 128))  # parameters never created (glow.py:233-235)
    # shapes glow.py:226-265 accepts and the HIP path does not build are refused at CONSTRUCTION, naming the option
    wn = cfg["WN_config"]
    for bad, word in ((dict(n_group=6, hop_length=300, win_length=1200), "n_group=6"), (dict(n_group=32), "n_group=32"),
                      (dict(n_group=12, hop_length=300, win_length=1200), "hop_length / n_group = 25"),
                      (dict(WN_config=dict(wn, kernel_size=5)), "kernel_size=5"), (dict(WN_config=dict(wn, n_channels=192)), "n_channels=192"),
                      (dict(WN_config=dict(wn, n_layers=13)), "n_layers=13"), (dict(n_mel_channels=81), "n_mel_channels * n_group")):
        with pytest.raises(NotImplementedError, match=re.escape(word)):
            WaveGlow(**dict(cfg, **bad))
    for ok in (dict(n_group=16, hop_length=512), dict(n_group=12, hop_length=384, win_length=1152), dict(n_group=4)):
        WaveGlow(**dict(cfg, **ok))                                        # other hop / n_group combinations build
    half = WaveGlow(**dict(cfg, n_group=16, hop_length=512))
    with pytest.raises(NotImplementedError, match="n_group"):
        half.set_compute_dtype(torch.float16)                             # reduced precision: flow boundaries of <= 8 channels


def test_unknown_constructor_options_raise_type_error():
    """efficient_model_ax.py:19 has a closed keyword list: a misspelt option is a TypeError there and here (it used to be
    swallowed by **unsupported and the model loaded with different arithmetic)."""
    from cookietts_amd.waveglow_ax import WaveGlow as AxWaveGlow
    cfg = dict(synthetic.WAVEFLOW_CONFIGS["toy"])
    AxWaveGlow(**cfg)
    with pytest.raises(TypeError, match="chanel_mixing"):
        AxWaveGlow(**dict(cfg, chanel_mixing="permuteheight"))
    with pytest.raises(NotImplementedError):
        AxWaveGlow(**dict(cfg, iso226_empthasis=True))                    # out of scope, refused by name
    with pytest.raises(TypeError):
        WaveGlow(**dict(synthetic.WAVEGLOW_CONFIGS["toy"], n_flow=4))


def test_option_parameter_trees():
    """glow.py:127-133, 181-183, 238-241: ReZero scalars, per-flow speaker tables, grouped upsampling weights."""
    m = WaveGlow(**synthetic.WAVEGLOW_CONFIGS["toy_spk_rezero"])
    assert len(m.WN[0].alpha_i) == 3 and 0.09 <= float(m.WN[0].alpha_i[0]) <= 0.11
    assert tuple(m.WN[1].speaker_embed.weight.shape) == (512, 20)
    assert tuple(m.WN[0].cond_layers[0].weight_v.shape) == (256, 640 + 20, 1)
    assert m.c_config().speaker_embed_dim == 20
    assert tuple(WaveGlow(**synthetic.WAVEGLOW_CONFIGS["toy_simple"]).upsample.weight.shape) == (80, 1, 1024)
    assert tuple(WaveGlow(**synthetic.WAVEGLOW_CONFIGS["toy_simple_half"]).upsample.weight.shape) == (80, 2, 1024)


def test_synthetic_recipe_is_deterministic():
    cfg = synthetic.WAVEGLOW_CONFIGS["toy"]
    a = synthetic.waveglow_state_dict(cfg, seed=7)
    b = synthetic.waveglow_state_dict(cfg, seed=7)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert np.abs(a["WN.0.end.weight"]).max() > 0
    z = synthetic.synthetic_noise(2, 8, 64, seed=3)
    assert z.shape == (2, 8, 64) and abs(float(z.std()) - 1.0) < 0.1


def test_host_side_queries_of_the_other_families(hip_lib_path):
    """Size queries and argument validation of the WaveFlow / Tacotron / STFT / conv1d / alignment entry points are host
    code too: they answer (or refuse with a message) without a GPU."""
    from cookietts_amd.waveglow_ax import WaveGlow as WaveFlow
    lib = _lib.lib()
    for key, folded in (("full", True), ("author", False)):
        m = WaveFlow(**synthetic.WAVEFLOW_CONFIGS[key])
        c = m.c_config()
        assert bool(c.cond_precomputed) != folded and bool(c.seperable_conv) == (key == "author")
        assert lib.ctts_waveflow_packed_bytes(ctypes.byref(c)) > 0
        hop = synthetic.WAVEFLOW_CONFIGS[key]["hop_length"]
        assert lib.ctts_waveflow_workspace_bytes(ctypes.byref(c), 2, 7 * hop) > 0      # L not a multiple of 4 is fine
    bad = m.c_config()
    bad.seperable_conv = 0                                  # dense 7x7 = 49 taps: more segments than the GEMM takes
    assert lib.ctts_waveflow_packed_bytes(ctypes.byref(bad)) == 0 and b"seperable_conv" in lib.ctts_last_error()
    bad = m.c_config()
    bad.n_channels = 96
    assert lib.ctts_waveflow_packed_bytes(ctypes.byref(bad)) == 0 and b"n_channels" in lib.ctts_last_error()
    # conv1d primitive: c_in must be a multiple of 16, odd kernel <= 11
    ok = _lib.Conv1dDesc(c_in=32, c_out=24, kernel_size=9, act=1, slope=0.0)
    assert lib.ctts_conv1d_packed_bytes(ctypes.byref(ok)) > 0
    for kw in (dict(c_in=40), dict(kernel_size=4), dict(kernel_size=13)):
        d = _lib.Conv1dDesc(**{**dict(c_in=32, c_out=24, kernel_size=9, act=1, slope=0.0), **kw})
        assert lib.ctts_conv1d_packed_bytes(ctypes.byref(d)) == 0
    # alignment scoring and packed-sequence LSTM workspaces
    assert lib.ctts_alignment_workspace_bytes(4, 900, 200) == (4 * 900 * 2 + 4 * 29 * 200) * 4
    assert lib.ctts_alignment_workspace_bytes(0, 900, 200) == 0
    assert lib.ctts_lstm_seq_workspace_bytes(512, 4, 256) > 0 and lib.ctts_lstm_seq_workspace_bytes(512, 256, 256) > 0      # batched MFMA steps: H % 64 == 0
    assert lib.ctts_lstm_seq_workspace_bytes(512, 257, 256) == 0 and lib.ctts_lstm_seq_workspace_bytes(100, 5, 256) == 0 and lib.ctts_lstm_seq_workspace_bytes(100, 4, 256) > 0
    # Tacotron decoder: up to ctts_taco_decoder_max_batch rows per workspace - 256 where the batched MFMA form is built
    # (every K a multiple of 64, every width of 16, the window kernel's limits), 4 otherwise; persistent form: <= 4 rows
    from cookietts_amd.tacotron2 import Tacotron2
    dc = Tacotron2(synthetic.tacotron_hparams()).decoder.c_config()
    assert lib.ctts_taco_decoder_packed_bytes(ctypes.byref(dc)) > 0
    assert lib.ctts_taco_decoder_max_batch(ctypes.byref(dc)) == 256
    assert lib.ctts_taco_decoder_workspace_bytes(ctypes.byref(dc), 4, 200) > 0
    assert lib.ctts_taco_decoder_workspace_bytes(ctypes.byref(dc), 256, 200) > lib.ctts_taco_decoder_workspace_bytes(ctypes.byref(dc), 5, 200) > 0
    assert lib.ctts_taco_decoder_workspace_bytes(ctypes.byref(dc), 257, 200) == 0
    assert lib.ctts_taco_decoder_persistent_bytes(ctypes.byref(dc), 5, 200) == 0
    small = Tacotron2(synthetic.tacotron_hparams(**synthetic.TACOTRON_SMALL_OVERRIDES)).decoder.c_config()
    assert lib.ctts_taco_decoder_max_batch(ctypes.byref(small)) == 256      # the non-default golden checkpoint shape too
    odd = Tacotron2(synthetic.tacotron_hparams()).decoder.c_config()
    odd.prenet_dim = 200                                                    # K of the second prenet layer not a multiple of 64
    assert lib.ctts_taco_decoder_max_batch(ctypes.byref(odd)) == 4
    assert lib.ctts_taco_decoder_workspace_bytes(ctypes.byref(odd), 5, 200) == 0


def test_gemm_mode_names_and_tuning_bits(hip_lib_path):
    """The arithmetic mode names map onto the header's CTTS_GEMM_* values (ABI 4 + the six-product loop), unknown names are
    refused on the host, the library refuses unknown defaults, and every launch-shape knob has its bit in ctts_tuning_flags."""
    import re
    from cookietts_amd import WaveGlow, _lib
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "cookietts_hip.h")).read()
    consts = {k: int(v) for k, v in re.findall(r"#define (CTTS_GEMM_\w+) (\d+)", hdr)}
    assert consts == {"CTTS_GEMM_DEFAULT": 0, "CTTS_GEMM_F32": 1, "CTTS_GEMM_BF16X3": 2, "CTTS_GEMM_BF16X6": 3}
    assert [_lib.model_gemm_mode(m) for m in (None, "default", "f32", "bf16x3", "bf16x6")] == [0, 0, 1, 2, 3]
    with pytest.raises(ValueError):
        _lib.model_gemm_mode("bf16x9")
    m = WaveGlow(**synthetic.WAVEGLOW_CONFIGS["toy"])
    assert m.set_f32_gemm_mode("bf16x6") is m and m.c_config().f32_gemm_mode == 3
    # one encoding (CTTS_GEMM_*); no process-wide default since ABI 6: the old setter is read-only
    assert _lib.GEMM_MODES == {"f32": 1, "bf16x3": 2, "bf16x6": 3}
    assert all(_lib.MODEL_GEMM_MODES[k] == v for k, v in _lib.GEMM_MODES.items())
    lib = _lib.lib()
    assert lib.ctts_abi_version() == 7
    # the process-wide switch of ABI <= 5 is not even a symbol any more (ABI 7)
    assert not hasattr(lib, "ctts_set_f32_gemm_mode") and not hasattr(lib, "ctts_get_f32_gemm_mode")
    assert "ctts_set_f32_gemm_mode(" not in hdr and lib.ctts_last_gemm_loop() == 0
    # profiles are caller-owned handles (no GPU needed to create, bind, collect an empty slot and destroy one)
    import ctypes as C
    h = C.c_void_p()
    assert lib.ctts_profile_create(C.byref(h)) == 0 and h.value
    n, ms = C.c_int64(-1), C.c_double(-1.0)
    assert lib.ctts_profile_bind(h) == 0 and lib.ctts_profile_collect(h, 0, C.byref(n), C.byref(ms)) == 0
    assert n.value == 0 and ms.value == 0.0
    assert lib.ctts_profile_collect(h, 9, C.byref(n), C.byref(ms)) != 0 and lib.ctts_profile_collect(None, 0, C.byref(n), C.byref(ms)) != 0
    assert lib.ctts_profile_bind(None) == 0 and lib.ctts_profile_destroy(h) == 0
    assert "int ctts_profile_enable" not in hdr
    for name, bit in _lib.TUNING_BITS.items():
        assert f"{bit} {name}" in hdr or name in hdr, name
    assert _lib.TUNING_BITS["CTTS_F32_NO_SPLITK"] == 11


def test_checkpoint_config_selects_the_model_class():
    """_4_mtw/waveglow/train.py:385-394: the trainer builds efficient_model_ax.WaveGlow (``ax = True``), so the checkpoints it
    writes (train.py:128-145) carry that class's kwargs; a glow.py config has only glow.py:225-226's thirteen keys."""
    from cookietts_amd.vocoder import is_ax_config, waveglow_from_checkpoint, WaveGlowVocoder
    from cookietts_amd.waveglow_ax import WaveGlow as WaveGlowAx
    for table, make, keys in ((synthetic.WAVEFLOW_CONFIGS, synthetic.waveflow_state_dict, ("toy", "author_toy")),
                              (synthetic.WAVEGLOW_AX_CONFIGS, synthetic.waveglow_ax_state_dict, ("notebook_toy",))):
        for key in keys:
            cfg = table[key]
            assert is_ax_config(cfg)
            # legacy key spellings are renamed on load like train.py:121 does
            sd = {k.replace("convinv", "invconv1x1", 1): v for k, v in synthetic.to_torch(make(cfg, seed=1)).items()}
            m = waveglow_from_checkpoint({"model": sd, "waveglow_config": cfg})
            assert isinstance(m, WaveGlowAx) and m.waveflow == bool(cfg.get("waveflow", True))
            v = WaveGlowVocoder(m, speaker_lookup={11: 0, 12: 3})
            assert v.is_ax and v.speaker_ids_for([12, 11]).tolist() == [3, 0]
    cfg = synthetic.WAVEGLOW_CONFIGS["toy"]
    assert not is_ax_config(cfg)
    m = waveglow_from_checkpoint({"model": synthetic.to_torch(synthetic.waveglow_state_dict(cfg, seed=1)), "waveglow_config": cfg})
    assert isinstance(m, WaveGlow) and not WaveGlowVocoder(m).is_ax
