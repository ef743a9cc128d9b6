"""Parity of the HIP path (through the C ABI) with the reference goldens and the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rms_rel_err
from cookietts_amd import WaveGlow, synthetic

pytestmark = pytest.mark.gpu

# BASELINE.json: waveform RMS relative error <= 1e-3 (fp32, identical weights / mel / noise)
WAVE_TOL = 1e-3


def _model(key, seed):
    cfg = synthetic.WAVEGLOW_CONFIGS[key]
    sd = synthetic.waveglow_state_dict(cfg, seed=seed)
    m = WaveGlow(**cfg)
    m.load_state_dict(synthetic.to_torch(sd))
    return m.cuda().eval(), cfg, sd


@pytest.mark.parametrize("name", ["toy", "toy_early", "small", "full_short"])
def test_waveglow_matches_reference_golden(hip_lib_path, name):
    g = np.load(os.path.join(GOLDEN, f"waveglow_{name}.npz"))
    m, cfg, _ = _model(str(g["config_key"]), int(g["seed"]))
    wave = m.infer_from_noise(torch.from_numpy(g["mel"]).cuda(), torch.from_numpy(g["z_scaled"]).cuda())
    torch.cuda.synchronize()
    wave = wave.cpu().numpy()
    assert wave.shape == g["wave"].shape and np.isfinite(wave).all()
    err = rms_rel_err(wave, g["wave"])
    print(f"{name}: rms rel err vs reference = {err:.3e}")
    assert err < WAVE_TOL


@pytest.mark.parametrize("B,F", [(1, 1), (3, 5), (2, 37), (1, 130)])
def test_waveglow_matches_oracle_ragged_shapes(hip_lib_path, B, F):
    """Ragged tile edges: L = 32*F is not a multiple of the 128-step GEMM tile for most F."""
    from oracle import waveglow_oracle as wo
    m, cfg, sd = _model("toy_early", 21)
    mel = synthetic.synthetic_mel(B, F, seed=F)
    z = synthetic.synthetic_noise(B, cfg["n_group"], F * 32, seed=F) * np.float32(0.7)
    ref = wo.waveglow_infer(sd, cfg, mel, z)
    wave = m.infer_from_noise(torch.from_numpy(mel).cuda(), torch.from_numpy(z).cuda()).cpu().numpy()
    assert rms_rel_err(wave, ref) < WAVE_TOL
    # second call on the same workspace (halo must still be zero) and after a different geometry
    m.infer_from_noise(torch.from_numpy(synthetic.synthetic_mel(1, 3)).cuda(),
                       torch.from_numpy(synthetic.synthetic_noise(1, 8, 96)).cuda())
    wave2 = m.infer_from_noise(torch.from_numpy(mel).cuda(), torch.from_numpy(z).cuda()).cpu().numpy()
    assert np.array_equal(wave, wave2)


def test_batch_items_are_independent(hip_lib_path):
    m, cfg, _ = _model("toy", 5)
    mel = torch.from_numpy(synthetic.synthetic_mel(4, 20, seed=9)).cuda()
    z = torch.from_numpy(synthetic.synthetic_noise(4, 8, 640, seed=9)).cuda()
    full = m.infer_from_noise(mel, z)
    for b in range(4):
        one = m.infer_from_noise(mel[b:b + 1], z[b:b + 1])
        assert torch.equal(one[0], full[b])


def test_infer_api_contract(hip_lib_path):
    """glow.WaveGlow.infer contract: [B, n_mel, F] -> [B, F*hop] on the input device, sigma scales noise."""
    m, cfg, _ = _model("toy", 5)
    mel = torch.from_numpy(synthetic.synthetic_mel(2, 10)).cuda()
    torch.manual_seed(0)
    a = m.infer(mel, sigma=0.6)
    assert a.shape == (2, 10 * 256) and a.is_cuda and a.dtype == mel.dtype and torch.isfinite(a).all()
    torch.manual_seed(0)
    b = m.infer(mel, sigma=0.6)
    assert torch.equal(a, b)
    assert m.convinv[0].W_inverse.shape == (8, 8, 1)         # cached like glow.py:97
    # weight-norm removal must not change the function
    z = torch.from_numpy(synthetic.synthetic_noise(2, 8, 320)).cuda()
    before = m.infer_from_noise(mel, z)
    WaveGlow.remove_weightnorm(m)
    assert "WN.0.start.weight" in m.state_dict() and "WN.0.start.weight_g" not in m.state_dict()
    after = m.infer_from_noise(mel, z)
    assert rms_rel_err(after.cpu().numpy(), before.cpu().numpy()) < 1e-5


def test_stage_upsample_squeeze_against_oracle(hip_lib_path):
    import ctypes as C
    from cookietts_amd import _lib
    from oracle import waveglow_oracle as wo
    m, cfg, sd = _model("toy", 5)
    B, F = 2, 19
    mel = synthetic.synthetic_mel(B, F, seed=4)
    ref = wo.upsample_squeeze(mel, sd["upsample.weight"], sd["upsample.bias"], 256, 8)
    blob, _ = m._ensure_packed(torch.device("cuda", 0))
    lib = _lib.lib()
    c = m.c_config()
    geo = _lib.WaveGlowGeometry()
    _lib.check(lib.ctts_waveglow_geometry_for(C.byref(c), F, C.byref(geo)), "geometry")
    spect = torch.zeros(B, 640, geo.ld, device="cuda")
    melt = torch.from_numpy(mel).cuda()
    _lib.check(lib.ctts_upsample_squeeze_f32(C.byref(c), _lib.ptr(blob), _lib.ptr(melt), _lib.ptr(spect), B, F,
                                            None), "upsample_squeeze")
    torch.cuda.synchronize()
    got = spect[:, :, geo.pad:geo.pad + geo.steps].cpu().numpy()
    assert np.abs(got - ref).max() < 1e-4                     # mel-domain L_inf bound of BASELINE.json
    halo = spect.clone()
    halo[:, :, geo.pad:geo.pad + geo.steps] = 0
    assert float(halo.abs().max()) == 0.0                     # halo columns stay zero


@pytest.mark.parametrize("B,F", [(1, 19), (3, 333), (2, 900)])
def test_mfma_upsampling_is_bit_identical_to_the_valu_kernel(hip_lib_path, tuning, B, F):
    """The benchmark's upsampling shape (n_mel 80, hop 256, win 1024, n_group 8) runs as a W-stationary fp32 MFMA GEMM
    (waveglow_kernels.hip upsample_squeeze_mfma_kernel); ``v_mfma_f32_16x16x4_f32`` adds its four k in order, and the kernel walks
    k = (input channel, tap) in the VALU kernel's order - so the two agree bit for bit, on whole and ragged chunks (333 frames =
    2 chunks of 176 / 157; 900 = 304 / 304 / 292), and neither touches the halo columns."""
    import ctypes as C
    from cookietts_amd import _lib
    from oracle import waveglow_oracle as wo
    m, cfg, sd = _model("toy", 5)
    mel = synthetic.synthetic_mel(B, F, seed=11)
    blob, _ = m._ensure_packed(torch.device("cuda", 0))
    lib = _lib.lib()
    c = m.c_config()
    geo = _lib.WaveGlowGeometry()
    _lib.check(lib.ctts_waveglow_geometry_for(C.byref(c), F, C.byref(geo)), "geometry")
    melt = torch.from_numpy(mel).cuda()

    def run():
        spect = torch.zeros(B, 640, geo.ld, device="cuda")
        _lib.check(lib.ctts_upsample_squeeze_f32(C.byref(c), _lib.ptr(blob), _lib.ptr(melt), _lib.ptr(spect), B, F, None), "upsample_squeeze")
        torch.cuda.synchronize()
        return spect
    mfma = run()
    tuning.set("CTTS_UP_NO_MFMA")
    valu = run()
    assert torch.equal(mfma, valu)
    assert float(mfma[:, :, :geo.pad].abs().max()) == 0.0 and float(mfma[:, :, geo.pad + geo.steps:].abs().max()) == 0.0
    if F <= 333:
        ref = wo.upsample_squeeze(mel, sd["upsample.weight"], sd["upsample.bias"], 256, 8)
        assert np.abs(mfma[:, :, geo.pad:geo.pad + geo.steps].cpu().numpy() - ref).max() < 1e-4


def test_no_cpu_fallback(hip_lib_path):
    from cookietts_amd import _lib
    cfg = synthetic.WAVEGLOW_CONFIGS["toy"]
    m = WaveGlow(**cfg)
    with pytest.raises(_lib.HipLibraryError):
        m.infer(torch.zeros(1, 80, 4))


# ---- bf16 variant (BASELINE config 3) -----------------------------------------------------------
# Gate: against the bf16-ROUNDED CPU restatement (same rounding points, fp32 sums).  The two differ only by
# fp32 summation order and the hardware exp/rcp in the gate, which flips an occasional bf16 rounding
# (1 ulp = 2^-8 relative) that then propagates through 8 layers x n flows; bound found empirically.
BF16_VS_BF16_ORACLE_TOL = 5e-3
# Against the fp32 REFERENCE goldens the bf16 path is gated at what a single bf16 product can deliver (measured on the
# MI355X: toy_early 5e-4, small 1.5e-3, full_short 2.1e-3, full_len (80 x 900) 2.4e-3; tests/test_bf16_error_budget.py pins on the CPU why: operand
# rounding alone gives 1.4e-3 / 1.9e-3 on small / full_short even with every tensor stored in fp32).  Limits = measured
# x ~1.5.  The north-star bound of 1e-3 is met by the fp32 path and by bf16x3, not by config 3's arithmetic.
BF16_VS_REFERENCE_LIMIT = {"toy_early": 1.0e-3, "small": 2.5e-3, "full_short": 3.2e-3, "full_len": 3.6e-3}


@pytest.mark.parametrize("name", ["toy_early", "small", "full_short"])
def test_waveglow_bf16_matches_bf16_rounded_oracle(hip_lib_path, name):
    from oracle import waveglow_oracle as wo
    g = np.load(os.path.join(GOLDEN, f"waveglow_{name}.npz"))
    m, cfg, sd = _model(str(g["config_key"]), int(g["seed"]))
    m.set_compute_dtype(torch.bfloat16)
    wave = m.infer_from_noise(torch.from_numpy(g["mel"]).cuda(), torch.from_numpy(g["z_scaled"]).cuda()).cpu().numpy()
    assert np.isfinite(wave).all()
    ref16 = wo.waveglow_infer(sd, cfg, g["mel"], g["z_scaled"], bf16=True)
    e16 = rms_rel_err(wave, ref16)
    e32 = rms_rel_err(wave, g["wave"])
    print(f"bf16 {name}: rms rel err vs bf16-rounded oracle = {e16:.3e}; vs fp32 reference (reported) = {e32:.3e}")
    assert e16 < BF16_VS_BF16_ORACLE_TOL
    assert e32 < BF16_VS_REFERENCE_LIMIT[name]
    # switching back restores the exact fp32 path
    m.set_compute_dtype(torch.float32)
    w32 = m.infer_from_noise(torch.from_numpy(g["mel"]).cuda(), torch.from_numpy(g["z_scaled"]).cuda()).cpu().numpy()
    assert rms_rel_err(w32, g["wave"]) < WAVE_TOL


# ---- IEEE half ("f16"): the bf16 path's layouts and kernels on half storage / v_mfma_f32_32x32x16_f16 -----------------
# The reference's own reduced-precision mode (glow.py:343).  11-bit significands: against the fp32 REFERENCE goldens this
# path is held to the north-star bound itself (RMS rel <= 1e-3), which one bf16 product per MAC cannot meet.
F16_VS_F16_ORACLE_TOL = 1e-3


@pytest.mark.parametrize("name", ["toy_early", "small", "full_short"])
def test_waveglow_f16_matches_f16_rounded_oracle_and_the_reference_golden(hip_lib_path, name):
    from oracle import waveglow_oracle as wo
    g = np.load(os.path.join(GOLDEN, f"waveglow_{name}.npz"))
    m, cfg, sd = _model(str(g["config_key"]), int(g["seed"]))
    m.set_compute_dtype(torch.float16)
    mel, z = torch.from_numpy(g["mel"]).cuda(), torch.from_numpy(g["z_scaled"]).cuda()
    wave = m.infer_from_noise(mel, z).cpu().numpy()
    assert np.isfinite(wave).all()
    ref16 = wo.waveglow_infer(sd, cfg, g["mel"], g["z_scaled"], f16=True)
    e16, e32 = rms_rel_err(wave, ref16), rms_rel_err(wave, g["wave"])
    print(f"f16 {name}: rms rel err vs f16-rounded oracle = {e16:.3e}; vs the fp32 reference golden = {e32:.3e}")
    assert e16 < F16_VS_F16_ORACLE_TOL
    assert e32 < WAVE_TOL                                     # the north-star bound, against the reference's own output
    # not the bf16 arithmetic under another name: bf16 on the same input is several times further from the reference
    m.set_compute_dtype(torch.bfloat16)
    eb = rms_rel_err(m.infer_from_noise(mel, z).cpu().numpy(), g["wave"])
    assert eb > 2.5 * e32, (eb, e32)
    m.set_compute_dtype(torch.float32)
    assert rms_rel_err(m.infer_from_noise(mel, z).cpu().numpy(), g["wave"]) < 1e-5


@pytest.mark.parametrize("knob", ["CTTS_BF16_NO_WIDE", "CTTS_BF16_PS", "CTTS_BF16_NO_PS", "CTTS_GEMM_NO_XCD_PAIR"])
def test_waveglow_f16_block_shapes_agree(hip_lib_path, tuning, knob):
    """The f16 instantiations of the skewed / persistent / narrow kernels accumulate K in the same order: identical bits
    (same size as the bf16 block-shape test: the wide kernels with a ragged last tile, two or three tiles per persistent
    workgroup)."""
    m, cfg, sd = _model("full", 5)
    m.set_compute_dtype(torch.float16)
    B, F = 8, 131
    mel = torch.from_numpy(synthetic.synthetic_mel(B, F, seed=6)).cuda()
    z = torch.from_numpy(synthetic.synthetic_noise(B, 8, F * 32, seed=6) * np.float32(0.6)).cuda()
    default = m.infer_from_noise(mel, z)
    assert torch.isfinite(default).all()
    for b in (0, 5):                                          # (narrow kernels: a single utterance is below the wide threshold)
        assert torch.equal(m.infer_from_noise(mel[b:b + 1].contiguous(), z[b:b + 1].contiguous())[0], default[b])
    tuning.set(knob)
    assert torch.equal(default, m.infer_from_noise(mel, z))


# ---- split bf16 ("bf16x3"): hi + lo bf16 operands, three bf16 MFMA products per contraction ---------------------
# Held to the fp32 bar: the reference fp32 goldens at the north-star tolerance (RMS rel <= 1e-3); the measured error is
# printed (expected ~1e-5: operands carry 16 mantissa bits).
@pytest.mark.parametrize("name", ["toy", "toy_early", "small", "full_short"])
def test_waveglow_bf16x3_matches_reference_golden(hip_lib_path, name):
    g = np.load(os.path.join(GOLDEN, f"waveglow_{name}.npz"))
    m, cfg, _ = _model(str(g["config_key"]), int(g["seed"]))
    ref32 = m.infer_from_noise(torch.from_numpy(g["mel"]).cuda(), torch.from_numpy(g["z_scaled"]).cuda()).cpu().numpy()
    m.set_compute_dtype("bf16x3")
    wave = m.infer_from_noise(torch.from_numpy(g["mel"]).cuda(), torch.from_numpy(g["z_scaled"]).cuda()).cpu().numpy()
    err = rms_rel_err(wave, g["wave"])
    print(f"bf16x3 {name}: rms rel err vs reference golden = {err:.3e}; vs the fp32 MFMA path = {rms_rel_err(wave, ref32):.3e}")
    assert np.isfinite(wave).all()
    assert err < WAVE_TOL
    assert err < 1e-4            # and an order of magnitude inside it: this is not the bf16 path


def test_waveglow_bf16x3_speaker_options_and_block_shapes(hip_lib_path, tuning):
    """Speaker rows + ReZero through the split path (ragged 20-wide embedding), and the narrow / wide block shapes of
    the split GEMMs agree bit for bit (same K order)."""
    g = np.load(os.path.join(GOLDEN, "waveglow_toy_spk_rezero.npz"))
    m, cfg, sd = _model("toy_spk_rezero", int(g["seed"]))
    m.set_compute_dtype("bf16x3")
    ids = torch.from_numpy(g["speaker_ids"]).cuda()
    w = m.infer_from_noise(torch.from_numpy(g["mel"]).cuda(), torch.from_numpy(g["z_scaled"]).cuda(), speaker_id=ids)
    err = rms_rel_err(w.cpu().numpy(), g["wave"])
    print(f"bf16x3 toy_spk_rezero: {err:.3e}")
    assert err < 1e-4
    m, cfg, sd = _model("full", 5)
    m.set_compute_dtype("bf16x3")
    B, F = 8, 131
    mel = torch.from_numpy(synthetic.synthetic_mel(B, F, seed=6)).cuda()
    z = torch.from_numpy(synthetic.synthetic_noise(B, 8, F * 32, seed=6) * np.float32(0.6)).cuda()
    default = m.infer_from_noise(mel, z)
    tuning.set("CTTS_BF16_NO_WIDE")
    assert torch.equal(default, m.infer_from_noise(mel, z))
    tuning.clear("CTTS_BF16_NO_WIDE")
    tuning.set("CTTS_BF16_NO_GLDS")
    assert torch.equal(default, m.infer_from_noise(mel, z))


def test_waveglow_bf16_ragged_and_batch_independent(hip_lib_path):
    m, cfg, sd = _model("toy_early", 21)
    m.set_compute_dtype(torch.bfloat16)
    mel = torch.from_numpy(synthetic.synthetic_mel(3, 37, seed=2)).cuda()
    z = torch.from_numpy(synthetic.synthetic_noise(3, 8, 37 * 32, seed=2) * np.float32(0.7)).cuda()
    full = m.infer_from_noise(mel, z)
    assert torch.isfinite(full).all() and full.shape == (3, 37 * 256)
    for b in range(3):
        assert torch.equal(m.infer_from_noise(mel[b:b + 1], z[b:b + 1])[0], full[b])


@pytest.mark.parametrize("knob", ["CTTS_BF16_W4", "CTTS_BF16_NO_WIDE", "CTTS_BF16_NO_PP", "CTTS_BF16_NO_GLDS",
                                  "CTTS_GEMM_NO_XCD_PAIR", "CTTS_BF16_PS", "CTTS_BF16_NO_PS"])
def test_waveglow_bf16_block_shapes_agree(hip_lib_path, tuning, knob):
    """Full model at a size that selects the 256x256 skewed 8-wave kernel (>= 512 workgroups, ragged last tile; W4
    selects the four-wave 128x128-wave-tile kernel of the same block; PS the persistent form, in which a workgroup walks
    two or three tiles here and the DMA runs across the tile boundaries):
    every block shape / staging variant accumulates K in the same order, so the waveforms must be identical."""
    m, cfg, sd = _model("full", 5)
    m.set_compute_dtype(torch.bfloat16)
    B, F = 8, 131
    mel = torch.from_numpy(synthetic.synthetic_mel(B, F, seed=6)).cuda()
    z = torch.from_numpy(synthetic.synthetic_noise(B, 8, F * 32, seed=6) * np.float32(0.6)).cuda()
    default = m.infer_from_noise(mel, z)
    assert torch.isfinite(default).all()
    tuning.set(knob)
    other = m.infer_from_noise(mel, z)
    assert torch.equal(default, other)


@pytest.mark.parametrize("knob", ["CTTS_F32_NO_GLDS", "CTTS_GEMM_NO_XCD_PAIR"])
def test_waveglow_fp32_staging_variants_agree(hip_lib_path, tuning, knob):
    """fp32 conv-GEMM: DMA-staged 3-stage kernel (default) vs the register-staged one, and both block mappings:
    same K order, so the waveforms are identical (full model, ragged width, enough tiles for every path)."""
    m, cfg, sd = _model("full", 5)
    B, F = 2, 37
    mel = torch.from_numpy(synthetic.synthetic_mel(B, F, seed=6)).cuda()
    z = torch.from_numpy(synthetic.synthetic_noise(B, 8, F * 32, seed=6) * np.float32(0.6)).cuda()
    default = m.infer_from_noise(mel, z)
    assert torch.isfinite(default).all()
    tuning.set(knob)
    assert torch.equal(default, m.infer_from_noise(mel, z))


@pytest.mark.parametrize("name", ["toy_early", "full_short"])
def test_waveglow_per_layer_res_skip_form_matches_golden_and_the_default(hip_lib_path, tuning, name):
    """The default WN stack keeps every layer's gated activation and sums the skip rows of four layers in one K = 4C GEMM
    (glow.py:211-220 restated); CTTS_F32_NO_DEFER_SKIP is the one-res/skip-GEMM-per-layer form of rounds 1-3.  Same products,
    another order of the skip sum: both match the reference golden, and each other far inside the bound."""
    g = np.load(os.path.join(GOLDEN, f"waveglow_{name}.npz"))
    m, cfg, _ = _model(str(g["config_key"]), int(g["seed"]))
    args = (torch.from_numpy(g["mel"]).cuda(), torch.from_numpy(g["z_scaled"]).cuda())
    deferred = m.infer_from_noise(*args).cpu().numpy()
    tuning.set("CTTS_F32_NO_DEFER_SKIP")
    per_layer = m.infer_from_noise(*args).cpu().numpy()
    e = (rms_rel_err(deferred, g["wave"]), rms_rel_err(per_layer, g["wave"]), rms_rel_err(deferred, per_layer))
    print(f"{name}: deferred vs reference {e[0]:.3e}, per-layer vs reference {e[1]:.3e}, deferred vs per-layer {e[2]:.3e}")
    assert e[0] < WAVE_TOL and e[1] < WAVE_TOL and e[2] < 1e-5


def test_5_infer_vocoder_slot(hip_lib_path, tmp_path):
    """The two call sites of _5_infer/t2s_server/text2speech.py (:175-179, :658-665) against a reference-format
    checkpoint (train.py:128-145)."""
    from cookietts_amd import load_waveglow
    cfg = synthetic.WAVEGLOW_CONFIGS["toy"]
    sd = synthetic.waveglow_state_dict(cfg, seed=8)
    ckpt = {"model": synthetic.to_torch(sd), "waveglow_config": cfg, "iteration": 1, "speaker_lookup": {}, "learning_rate": 1e-4}
    path = str(tmp_path / "waveglow_ckpt.pt")
    torch.save(ckpt, path)
    vocoder, vcfg = load_waveglow(path)
    assert vcfg["n_group"] == 8
    vocoder_dtype = next(vocoder.parameters()).dtype
    mel_batch = torch.from_numpy(synthetic.synthetic_mel(3, 21)).cuda()
    audio = vocoder(mel_batch.to(vocoder_dtype)).squeeze(1).cpu().split(1, dim=0)       # text2speech.py:664
    assert len(audio) == 3 and audio[0].shape == (1, 21 * 256) and torch.isfinite(audio[0]).all()
    torch.manual_seed(3)
    audio32 = vocoder(mel_batch).squeeze(1)
    vocoder.half()                                                                       # text2speech.py:261
    assert next(vocoder.parameters()).dtype == torch.float32
    torch.manual_seed(3)
    audio16 = vocoder(mel_batch).squeeze(1)
    assert audio16.shape == (3, 21 * 256) and torch.isfinite(audio16).all()
    # .half() = IEEE-half storage + fp16 MFMA from fp32 masters: the same noise gives the fp32 audio within the 1e-3 bound
    assert rms_rel_err(audio16.cpu().numpy(), audio32.cpu().numpy()) < WAVE_TOL


def _reference_format_checkpoint(path, cfg, sd, lookup):
    """What _4_mtw/waveglow/train.py:128-145 writes (optimizer / scheduler states left out: the loader never reads them)."""
    torch.save({"model": synthetic.to_torch(sd), "waveglow_config": cfg, "iteration": 7, "learning_rate": 1e-4,
                "speaker_lookup": lookup}, path)


@pytest.mark.parametrize("golden,table", [("waveflow_author_toy", "WAVEFLOW_CONFIGS"), ("waveflow_author_short", "WAVEFLOW_CONFIGS"),
                                          ("waveglow_ax_notebook_toy", "WAVEGLOW_AX_CONFIGS"), ("waveglow_ax_notebook", "WAVEGLOW_AX_CONFIGS")])
def test_5_infer_vocoder_slot_loads_the_checkpoints_cookietts_trains(hip_lib_path, tmp_path, golden, table):
    """train.py:385-388 hard-codes the ax core, so a cookietts-trained vocoder checkpoint carries efficient_model_ax kwargs.
    ``load_waveglow`` must build the ax class from it, and ``vocoder(mel)`` must be the reference's own ``infer`` of that
    checkpoint: the author's WaveFlow option set and the inference notebook's 1-D ax WaveGlow against their reference goldens
    (``audio`` = efficient_model_ax.WaveGlow.infer with artifact_trimming=1, [b, (F-1)*hop])."""
    from cookietts_amd import load_waveglow
    from cookietts_amd.waveglow_ax import WaveGlow as WaveGlowAx
    g = np.load(os.path.join(GOLDEN, f"{golden}.npz"))
    cfg = getattr(synthetic, table)[str(g["config_key"])]
    make_sd = synthetic.waveflow_state_dict if table == "WAVEFLOW_CONFIGS" else synthetic.waveglow_ax_state_dict
    sd = make_sd(cfg, seed=int(g["seed"]))
    ext = [1000 + 7 * int(i) for i in g["speaker_ids"]]                       # dataset ids -> the model's internal ids
    path = str(tmp_path / "ax_ckpt.pt")
    _reference_format_checkpoint(path, cfg, sd, {e: int(i) for e, i in zip(ext, g["speaker_ids"])})
    vocoder, vcfg = load_waveglow(path)
    assert isinstance(vocoder.waveglow, WaveGlowAx) and vcfg["hop_length"] == cfg["hop_length"]
    ids = vocoder.speaker_ids_for(ext)
    assert ids.tolist() == [int(i) for i in g["speaker_ids"]]
    mel = torch.from_numpy(g["mel"]).cuda()
    vocoder_dtype = next(vocoder.parameters()).dtype                          # text2speech.py:661
    audio = vocoder(mel.to(vocoder_dtype), speaker_ids=ids, noise=torch.from_numpy(g["z"]).cuda())
    assert audio.is_cuda and tuple(audio.shape) == (mel.shape[0], 1, (mel.shape[2] - 1) * cfg["hop_length"])
    err = rms_rel_err(audio.squeeze(1).cpu().numpy(), g["audio"])
    print(f"{golden} through load_waveglow + vocoder(mel): rms rel err vs reference infer = {err:.3e}")
    assert err < WAVE_TOL
    # the server's call (no noise argument: the latent is drawn inside, efficient_model_ax.py:376-378) and its .half()
    out = vocoder(mel, speaker_ids=ids).squeeze(1).cpu().split(1, dim=0)      # text2speech.py:664-665
    assert len(out) == mel.shape[0] and torch.isfinite(out[0]).all() and out[0].shape == (1, g["audio"].shape[1])
    vocoder.half()
    assert next(vocoder.parameters()).dtype == torch.float32
    audio_h = vocoder(mel, speaker_ids=ids, noise=torch.from_numpy(g["z"]).cuda())
    assert rms_rel_err(audio_h.squeeze(1).cpu().numpy(), g["audio"]) < WAVE_TOL
    with pytest.raises(KeyError):
        vocoder.speaker_ids_for([5])


def test_packed_weights_follow_parent_load_state_dict_and_in_place_updates(hip_lib_path):
    """A parent's load_state_dict never calls the child's override (it recurses through _load_from_state_dict):
    the packed blob must still be rebuilt, or the second checkpoint would silently play the first one's weights."""
    from cookietts_amd import WaveGlowVocoder
    from oracle import waveglow_oracle as wo
    cfg = synthetic.WAVEGLOW_CONFIGS["toy"]
    sd_a, sd_b = synthetic.waveglow_state_dict(cfg, seed=31), synthetic.waveglow_state_dict(cfg, seed=32)
    inner = WaveGlow(**cfg)
    inner.load_state_dict(synthetic.to_torch(sd_a))
    voc = WaveGlowVocoder(inner.cuda().eval())
    mel = synthetic.synthetic_mel(2, 9, seed=1)
    z = synthetic.synthetic_noise(2, 8, 9 * 32, seed=1) * np.float32(0.7)
    tm, tz = torch.from_numpy(mel).cuda(), torch.from_numpy(z).cuda()
    assert rms_rel_err(inner.infer_from_noise(tm, tz).cpu().numpy(), wo.waveglow_infer(sd_a, cfg, mel, z)) < WAVE_TOL
    voc.load_state_dict({"waveglow." + k: v for k, v in synthetic.to_torch(sd_b).items()})      # through the PARENT
    ref_b = wo.waveglow_infer(sd_b, cfg, mel, z)
    assert rms_rel_err(inner.infer_from_noise(tm, tz).cpu().numpy(), ref_b) < WAVE_TOL
    # in-place update of one tensor (what an optimizer step does): picked up through the version counter
    with torch.no_grad():
        inner.WN[1].end.bias.add_(0.05)
    sd_c = dict(sd_b)
    sd_c["WN.1.end.bias"] = sd_b["WN.1.end.bias"] + np.float32(0.05)
    out_c = inner.infer_from_noise(tm, tz).cpu().numpy()
    assert rms_rel_err(out_c, wo.waveglow_infer(sd_c, cfg, mel, z)) < WAVE_TOL
    assert rms_rel_err(out_c, ref_b) > 1e-2                                   # and it really changed the output


@pytest.mark.parametrize("name", ["toy_spk_rezero", "toy_simple", "toy_hop512_g16", "toy_hop384_g12"])
def test_waveglow_options_match_reference_golden(hip_lib_path, name):
    """Multispeaker + ReZero WaveGlow, grouped ('simple') upsampling, and hop_length / n_group other than the benchmark's
    256 / 8 (512 / 16: flows of 16 and 12 channels; 384 / 12) against the reference's own outputs."""
    g = np.load(os.path.join(GOLDEN, f"waveglow_{name}.npz"))
    m, cfg, sd = _model(str(g["config_key"]), int(g["seed"]))
    ids = torch.from_numpy(g["speaker_ids"]).cuda() if "speaker_ids" in g.files else None
    mel, z = torch.from_numpy(g["mel"]).cuda(), torch.from_numpy(g["z_scaled"]).cuda()
    wave = m.infer_from_noise(mel, z, speaker_id=ids).cpu().numpy()
    err = rms_rel_err(wave, g["wave"])
    print(f"{name}: rms rel err vs reference = {err:.3e}")
    assert err < WAVE_TOL
    if ids is not None:
        assert m.multispeaker
        with pytest.raises(RuntimeError):
            m.infer_from_noise(mel, z)                                      # ids are required (glow.py:193-198)
        with pytest.raises(IndexError):                                     # host ids: checked like nn.Embedding, no sync
            m.infer_from_noise(mel, z, speaker_id=torch.tensor([0, 1, 512]))
        # device ids are not read back (no host sync per call): the kernel poisons THAT utterance instead of reading
        # outside the table; the other utterances are untouched
        bad_ids = ids.clone()
        bad_ids[2] = 512
        bad = m.infer_from_noise(mel, z, speaker_id=bad_ids).cpu().numpy()
        assert np.isnan(bad[2]).all() and np.array_equal(bad[:2], wave[:2])
        other = m.infer_from_noise(mel, z, speaker_id=ids.flip(0)).cpu().numpy()
        assert rms_rel_err(other, g["wave"]) > 5e-3                          # the embedding really conditions the flows
        assert rms_rel_err(other[1], g["wave"][1]) < WAVE_TOL                # ... per utterance: the middle id is unchanged
        out = m.infer(mel, speaker_id=ids, sigma=0.7)
        assert out.shape == (3, 10 * 256) and torch.isfinite(out).all()
        # bf16 MFMA path with the speaker rows as extra K of cond layer 0, against the bf16-rounded oracle
        from oracle import waveglow_oracle as wo
        m.set_compute_dtype(torch.bfloat16)
        w16 = m.infer_from_noise(mel, z, speaker_id=ids).cpu().numpy()
        ref16 = wo.waveglow_infer(sd, cfg, g["mel"], g["z_scaled"], bf16=True, speaker_ids=g["speaker_ids"])
        print(f"{name} bf16: vs bf16-rounded oracle {rms_rel_err(w16, ref16):.3e}")
        assert rms_rel_err(w16, ref16) < BF16_VS_BF16_ORACLE_TOL


def test_waveglow_simple_half_upsampling_matches_oracle(hip_lib_path):
    from oracle import waveglow_oracle as wo
    m, cfg, sd = _model("toy_simple_half", 5)
    assert tuple(m.upsample.weight.shape) == (80, 2, 1024)
    mel = synthetic.synthetic_mel(2, 9, seed=5)
    z = synthetic.synthetic_noise(2, 8, 9 * 32, seed=5) * np.float32(0.9)
    wave = m.infer_from_noise(torch.from_numpy(mel).cuda(), torch.from_numpy(z).cuda()).cpu().numpy()
    assert rms_rel_err(wave, wo.waveglow_infer(sd, cfg, mel, z)) < WAVE_TOL


def test_round_aligned_launch_is_bit_identical_to_the_single_launch(hip_lib_path, tuning):
    """The fp32 conv-GEMM peels the column tiles beyond the last whole round of 2 x CUs workgroups into a small-shape launch
    (gemm_f32.hip launch_gemm_f32; default on the headline path): same packed operands, same K order - the audio must be bit for
    bit that of the single launch (CTTS_F32_NO_ROUND_SPLIT).  512 channels (4 m-blocks) x 270 column tiles = 2.1 rounds."""
    cfg = synthetic.waveglow_config(n_flows=2, n_channels=512, n_layers=2, n_early_every=4)
    m = WaveGlow(**cfg)
    m.load_state_dict(synthetic.to_torch(synthetic.waveglow_state_dict(cfg, seed=77)))
    m = m.cuda().eval()
    F = 1080                                                    # 34560 columns = 270 tiles of 128
    mel = torch.from_numpy(synthetic.synthetic_mel(1, F, seed=7)).cuda()
    z = torch.from_numpy(synthetic.synthetic_noise(1, 8, F * 32, seed=7)).cuda() * 0.6
    split = m.infer_from_noise(mel, z)
    tuning.set("CTTS_F32_NO_ROUND_SPLIT")
    one = m.infer_from_noise(mel, z)
    assert torch.isfinite(split).all() and torch.equal(split, one)


def test_vocoder_slot_with_lengths_runs_buckets_of_similar_length(hip_lib_path):
    """``vocoder(mel, lengths=output_lengths)``: the server pads the mels of a vocoder call to the longest with -11.52 and trims
    the audio afterwards (text2speech.py:651, 677).  With lengths the padding frames are not computed: buckets of similar
    length, each at its own longest; an utterance's audio is the reference's infer of its bucket-trimmed mel (checked against
    the single-utterance call, sigma = 0 so that no noise is drawn), silence beyond its own frames."""
    from cookietts_amd import WaveGlowVocoder
    m, cfg, sd = _model("toy", 8)
    voc = WaveGlowVocoder(m, sigma=0.0)
    lengths = [40, 37, 21, 20, 9]
    T = max(lengths)
    mel = torch.full((5, 80, T), -11.52)
    raw = synthetic.synthetic_mel(5, T, seed=3)
    for i, n in enumerate(lengths):
        mel[i, :, :n] = torch.from_numpy(raw[i, :, :n])
    mel = mel.cuda()
    out = voc(mel, lengths=torch.tensor(lengths))
    assert out.shape == (5, 1, T * 256) and torch.isfinite(out).all()
    buckets = {0: 40, 1: 40, 2: 21, 3: 21, 4: 9}                            # 37 >= 0.85 * 40, 20 >= 0.85 * 21
    for i, n in enumerate(lengths):
        alone = voc(mel[i:i + 1, :, :buckets[i]].contiguous())
        assert float((out[i, :, :n * 256] - alone[0, :, :n * 256]).abs().max()) < 1e-5, i
        assert (out[i, :, n * 256:] == 0).all()
    padded = voc(mel)                                                        # the reference's call: padding frames computed
    assert float((padded[2, :, :21 * 256] - out[2, :, :21 * 256]).abs().max()) > 1e-4     # ... and felt near the end of item 2
    with pytest.raises(ValueError):
        voc(mel, lengths=[41, 1, 1, 1, 1])
