"""The split-bf16 main loop of the fp32 conv-GEMM (``model.set_f32_gemm_mode("bf16x3")``): every vocoder path under it stays
inside the north-star waveform bound against the REFERENCE goldens; the STFT keeps exact fp32 products."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rms_rel_err

pytestmark = pytest.mark.gpu


def test_there_is_no_process_wide_mode(hip_lib_path):
    """The mode travels in each model's config struct only (ABI 6); the old process-wide switch is gone from the C ABI (ABI 7) and
    the Python function of the same name raises for the split modes."""
    import cookietts_amd
    from cookietts_amd import _lib
    lib = _lib.lib()
    assert not hasattr(lib, "ctts_set_f32_gemm_mode") and _lib.MODEL_GEMM_MODES["f32"] == 1
    assert cookietts_amd.set_f32_gemm_mode("f32") == "f32"
    with pytest.raises(RuntimeError, match="model.set_f32_gemm_mode"):
        cookietts_amd.set_f32_gemm_mode("bf16x3")


@pytest.mark.parametrize("name", ["toy_early", "full_short"])
def test_waveglow_fp32_layout_split_mode(hip_lib_path, name):
    from cookietts_amd import WaveGlow, synthetic
    g = np.load(os.path.join(GOLDEN, f"waveglow_{name}.npz"))
    cfg = synthetic.WAVEGLOW_CONFIGS[str(g["config_key"])]
    m = WaveGlow(**cfg)
    m.load_state_dict(synthetic.to_torch(synthetic.waveglow_state_dict(cfg, seed=int(g["seed"]))))
    m = m.cuda().eval().set_f32_gemm_mode("bf16x3")
    wave = m.infer_from_noise(torch.from_numpy(g["mel"]).cuda(), torch.from_numpy(g["z_scaled"]).cuda()).cpu().numpy()
    err = rms_rel_err(wave, g["wave"])
    print(f"split-bf16 GEMM loop, waveglow {name}: rms rel err vs reference = {err:.3e}")
    assert err < 1e-4                                                       # bound 1e-3; an order inside it


def test_waveflow_split_mode(hip_lib_path):
    from cookietts_amd import WaveFlow, synthetic
    g = np.load(os.path.join(GOLDEN, "waveflow_full_short.npz"))
    cfg = synthetic.WAVEFLOW_CONFIGS[str(g["config_key"])]
    m = WaveFlow(**cfg)
    m.load_state_dict(synthetic.to_torch(synthetic.waveflow_state_dict(cfg, seed=int(g["seed"]))))
    m = m.cuda().eval().set_f32_gemm_mode("bf16x3")
    melp = torch.from_numpy(np.pad(g["mel"], ((0, 0), (0, 0), (0, 1)))).cuda()
    audio, _ = m.inverse(torch.from_numpy(g["z"]).cuda(), melp)
    err = rms_rel_err(audio.numpy(), g["inverse_full"])
    print(f"split-bf16 GEMM loop, waveflow full_short: rms rel err vs reference = {err:.3e}")
    assert err < 1e-4


def test_stft_has_no_mode_and_keeps_exact_fp32_products(hip_lib_path):
    """The STFT GEMMs always run exact fp32 products (their sums cancel; the log-mel bound is 1e-4 absolute): there is no
    mode to set on them, and a split-bf16 model living in the same process does not change their bits."""
    from cookietts_amd import TacotronSTFT, WaveGlow, synthetic
    stft = TacotronSTFT(1024, 256, 1024, 80, 22050, 0.0, 8000.0).cuda()
    assert not hasattr(stft, "set_f32_gemm_mode")
    audio = torch.from_numpy(np.random.default_rng(3).uniform(-0.7, 0.7, (2, 20000)).astype(np.float32)).cuda()
    mel_before = stft.mel_spectrogram(audio)
    cfg = synthetic.WAVEGLOW_CONFIGS["toy"]
    m = WaveGlow(**cfg)
    m.load_state_dict(synthetic.to_torch(synthetic.waveglow_state_dict(cfg, seed=1)))
    m = m.cuda().eval().set_f32_gemm_mode("bf16x3")
    m.infer_from_noise(torch.from_numpy(synthetic.synthetic_mel(1, 8, seed=2)).cuda(),
                       torch.from_numpy(synthetic.synthetic_noise(1, 8, 256, seed=2)).cuda())
    assert torch.equal(stft.mel_spectrogram(audio), mel_before)


def test_two_models_two_modes_interleaved_on_two_streams(hip_lib_path):
    """The arithmetic mode travels in each model's config struct (ABI 4): an fp32-MFMA model and a split-bf16 model of one
    process, called alternately on two streams, each reproduce bit for bit what they give when run alone, and the
    library default (left at fp32) is untouched."""
    from cookietts_amd import WaveGlow, _lib, synthetic
    g = np.load(os.path.join(GOLDEN, "waveglow_toy_early.npz"))
    cfg = synthetic.WAVEGLOW_CONFIGS[str(g["config_key"])]
    sd = synthetic.to_torch(synthetic.waveglow_state_dict(cfg, seed=int(g["seed"])))
    models = {}
    for mode in ("f32", "bf16x3"):
        m = WaveGlow(**cfg)
        m.load_state_dict(sd)
        models[mode] = m.cuda().eval().set_f32_gemm_mode(mode)
    assert models["f32"].c_config().f32_gemm_mode == 1 and models["bf16x3"].c_config().f32_gemm_mode == 2
    mel, z = torch.from_numpy(g["mel"]).cuda(), torch.from_numpy(g["z_scaled"]).cuda()
    alone = {k: m.infer_from_noise(mel, z).clone() for k, m in models.items()}
    torch.cuda.synchronize()
    assert not torch.equal(alone["f32"], alone["bf16x3"])                   # the two loops really differ in the last bits
    streams = {k: torch.cuda.Stream() for k in models}
    outs = {k: [] for k in models}
    for _ in range(3):
        for k, m in models.items():
            with torch.cuda.stream(streams[k]):
                outs[k].append(m.infer_from_noise(mel, z).clone())
    torch.cuda.synchronize()
    for k in models:
        for o in outs[k]:
            assert torch.equal(o, alone[k]), k
        err = rms_rel_err(alone[k].cpu().numpy(), g["wave"])
        print(f"{k}: rms rel err vs reference = {err:.3e}")
        assert err < (1e-5 if k == "f32" else 1e-4)
    with pytest.raises(ValueError):
        models["f32"].set_f32_gemm_mode("tf32")


# ---- "bf16x6": three-way split, the six products >= 2^-16: an fp32-GRADE fast path ----------------------------------
# (honest if its error against the goldens is within 2x of the fp32 MFMA path's)
X6_VS_F32_ERROR_RATIO = 2.0


@pytest.mark.parametrize("name", ["toy_early", "small", "full_short"])
def test_waveglow_bf16x6_is_fp32_grade(hip_lib_path, name):
    """glow.py path under the six-product loop (large shape at full_short, small shape at toy size): the error against
    the reference golden stays within 2x of what exact fp32 products give, and far below the three-product loop's."""
    from cookietts_amd import WaveGlow, synthetic
    g = np.load(os.path.join(GOLDEN, f"waveglow_{name}.npz"))
    cfg = synthetic.WAVEGLOW_CONFIGS[str(g["config_key"])]
    m = WaveGlow(**cfg)
    m.load_state_dict(synthetic.to_torch(synthetic.waveglow_state_dict(cfg, seed=int(g["seed"]))))
    m = m.cuda().eval()
    mel, z = torch.from_numpy(g["mel"]).cuda(), torch.from_numpy(g["z_scaled"]).cuda()
    err = {}
    for mode in ("f32", "bf16x3", "bf16x6"):
        m.set_f32_gemm_mode(mode)
        err[mode] = rms_rel_err(m.infer_from_noise(mel, z).cpu().numpy(), g["wave"])
    print(f"waveglow {name}: rms rel err vs reference: fp32 MFMA {err['f32']:.3e}, bf16x6 {err['bf16x6']:.3e}, bf16x3 {err['bf16x3']:.3e}")
    assert m.c_config().f32_gemm_mode == 3
    assert err["bf16x6"] <= X6_VS_F32_ERROR_RATIO * err["f32"]
    assert err["bf16x6"] < err["bf16x3"]


def test_ax_core_and_waveflow_bf16x6_are_fp32_grade(hip_lib_path):
    """The ax 1-D core (small shape, interpolated conditioning addend) and WaveFlow config 4 (fused layer: its main loop
    takes the six products, the in-register second GEMM stays fp32) under the six-product loop."""
    from cookietts_amd import WaveFlow, synthetic
    from cookietts_amd.waveglow_ax import WaveGlow as AxWaveGlow
    g = np.load(os.path.join(GOLDEN, "waveglow_ax_notebook_toy.npz"))
    cfg = synthetic.WAVEGLOW_AX_CONFIGS[str(g["config_key"])]
    m = AxWaveGlow(**cfg)
    m.load_state_dict(synthetic.to_torch(synthetic.waveglow_ax_state_dict(cfg, seed=int(g["seed"]))))
    m = m.cuda().eval()
    z, mel = torch.from_numpy(g["z"]).cuda(), torch.from_numpy(np.pad(g["mel"], ((0, 0), (0, 0), (0, 1)))).cuda()
    ids = torch.from_numpy(g["speaker_ids"]).cuda()
    err = {}
    for mode in ("f32", "bf16x6"):
        m.set_f32_gemm_mode(mode)
        err[mode] = rms_rel_err(m.inverse(z, mel, speaker_ids=ids, return_CPU=False)[0].cpu().numpy(), g["inverse_full"])
    print(f"waveglow_ax notebook_toy: fp32 MFMA {err['f32']:.3e}, bf16x6 {err['bf16x6']:.3e}")
    assert err["bf16x6"] <= X6_VS_F32_ERROR_RATIO * err["f32"]

    g = np.load(os.path.join(GOLDEN, "waveflow_full_short.npz"))
    cfg = synthetic.WAVEFLOW_CONFIGS[str(g["config_key"])]
    w = WaveFlow(**cfg)
    w.load_state_dict(synthetic.to_torch(synthetic.waveflow_state_dict(cfg, seed=int(g["seed"]))))
    w = w.cuda().eval()
    melp = torch.from_numpy(np.pad(g["mel"], ((0, 0), (0, 0), (0, 1)))).cuda()
    zz = torch.from_numpy(g["z"]).cuda()
    for mode in ("f32", "bf16x6"):
        w.set_f32_gemm_mode(mode)
        err[mode] = rms_rel_err(w.inverse(zz, melp)[0].numpy(), g["inverse_full"])
    print(f"waveflow full_short: fp32 MFMA {err['f32']:.3e}, bf16x6 {err['bf16x6']:.3e}")
    assert err["bf16x6"] <= X6_VS_F32_ERROR_RATIO * err["f32"]
