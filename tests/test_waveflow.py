"""WaveFlow (ax core, config 4): oracle vs reference goldens (CPU), HIP vs goldens / oracle (GPU)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rms_rel_err
from cookietts_amd import synthetic
from oracle import waveflow_oracle as wf

WAVE_TOL = 1e-3           # BASELINE.json waveform RMS rel-err bound
ORACLE_TOL = 5e-6


def _load(name):
    g = np.load(os.path.join(GOLDEN, f"waveflow_{name}.npz"))
    cfg = synthetic.WAVEFLOW_CONFIGS[str(g["config_key"])]
    return g, cfg, synthetic.waveflow_state_dict(cfg, seed=int(g["seed"]))


ALL = ["toy", "toy_odd", "full_short", "author_toy", "author_short", "untts_toy", "toy_merge", "author_toy_gate",
       "toy_groupconv", "toy_wn_tconv", "toy_wn_tconv_crop", "toy_conv_early", "toy_permute_mixfirst_early",
       "toy_conv_mixlast", "toy_upsample_first", "toy_no_res_skip",
       "toy_dilations", "toy_dilations_h", "author_toy_dilations_h",
       "table_g50_c128", "table_g50_c256_sep", "table_g20_c512", "table_g12_c256_sep"]   # author_*: SURVEY 8f.4 option set;
# table_*: corners of the reference's published sweep (n_group 50 / 20 / 12, 128 ... 512 channels, dense / separable);
# untts_toy: the same family with shift_spect / scale_spect (scripts/"UnTTS Inference.ipynb"); toy_merge / author_toy_gate:
# merge_res_skip with the GLU / GSIRRU gated units on the dense and the separable 2-D core


def _ids(g):
    return g["speaker_ids"] if "speaker_ids" in g.files else None


@pytest.mark.parametrize("name", ALL)
def test_oracle_matches_reference(name):
    g, cfg, sd = _load(name)
    out = wf.waveflow_infer(sd, cfg, g["mel"], g["z"], speaker_ids=_ids(g))
    assert out.shape == g["audio"].shape                       # (F-1)*hop samples
    assert rms_rel_err(out, g["audio"]) < ORACLE_TOL
    melp = np.pad(g["mel"], ((0, 0), (0, 0), (0, 1)))
    assert rms_rel_err(wf.waveflow_inverse(sd, cfg, g["z"], melp, _ids(g)), g["inverse_full"]) < ORACLE_TOL


def test_activation_table_restates_the_reference_mapping():
    """ax:100-111: 'lrelu' is F.relu, 'relu' is LeakyReLU(negative_slope) - as written in the reference."""
    x = np.array([-2.0, 3.0], np.float32)
    assert list(wf.activation('lrelu', 0.25)(x)) == [0.0, 3.0]
    assert list(wf.activation('relu', 0.25)(x)) == [-0.5, 3.0]
    from cookietts_amd.waveglow_ax import _act_code
    assert _act_code('lrelu', 0.25) == (1, 0.0) and _act_code('relu', 0.25) == (1, 0.25) and _act_code('none', None) == (0, 0.0)


def test_permutation_is_involution_and_matches_reference_pattern():
    assert wf.permutation(0, 8) == [7, 6, 5, 4, 3, 2, 1, 0]
    assert wf.permutation(2, 8) == [3, 2, 1, 0, 7, 6, 5, 4]    # efficient_modules.py:366-367 example
    for k in range(8):
        p = wf.permutation(k, 16)
        assert [p[i] for i in p] == list(range(16))


def test_nan_is_zeroed_per_flow():
    cfg = synthetic.WAVEFLOW_CONFIGS["toy"]
    sd = synthetic.waveflow_state_dict(cfg, seed=2)
    z = np.random.default_rng(0).standard_normal((1, 512)).astype(np.float32)
    z[0, 17] = np.nan
    out = wf.waveflow_inverse(sd, cfg, z, synthetic.synthetic_mel(1, 3))
    assert np.isfinite(out).all()


def test_host_state_dict_keys_match_reference_format():
    from cookietts_amd.waveglow_ax import WaveGlow
    for key in ("toy", "full", "author_toy", "author"):
        cfg = synthetic.WAVEFLOW_CONFIGS[key]
        sd = synthetic.waveflow_state_dict(cfg, seed=1)
        m = WaveGlow(**cfg)
        own = m.state_dict()
        assert sorted(own) == sorted(sd)
        assert all(tuple(own[k].shape) == sd[k].shape for k in sd)
    mc = WaveGlow(**synthetic.WAVEFLOW_CONFIGS["toy_conv_early"])
    assert mc.z_split_sizes == [2, 2, 4] and [tuple(c.weight.shape) for c in mc.convinv] == [(8, 8, 1)] * 2 + [(6, 6, 1)] * 2 + [(4, 4, 1)]
    import copy
    odd = copy.deepcopy(synthetic.WAVEFLOW_CONFIGS["toy"])
    odd.update(waveflow=False)
    odd["WN_config"]["n_channels"] = 48                    # the 1-D core takes multiples of 32 (round 3: no longer only of 128)
    with pytest.raises(NotImplementedError):
        WaveGlow(**odd)


def _model(key, seed):
    from cookietts_amd.waveglow_ax import WaveGlow
    cfg = synthetic.WAVEFLOW_CONFIGS[key]
    sd = synthetic.waveflow_state_dict(cfg, seed=seed)
    m = WaveGlow(**cfg)
    m.load_state_dict(synthetic.to_torch(sd))
    return m.cuda().eval(), cfg, sd


@pytest.mark.gpu
def test_hip_shift_scale_disable_the_folded_cond_layer(hip_lib_path):
    """config-4-style model (one linear WN cond layer, normally folded into the in-layer GEMM on the raw mel) with
    shift_spect / scale_spect: the mel must be shifted and scaled first (ax:281-284), so the fold is off."""
    from cookietts_amd.waveglow_ax import WaveGlow
    cfg = dict(synthetic.WAVEFLOW_CONFIGS["toy"], shift_spect=2.0, scale_spect=0.5)
    sd = synthetic.waveflow_state_dict(cfg, seed=21)
    m = WaveGlow(**cfg)
    m.load_state_dict(synthetic.to_torch(sd))
    m = m.cuda().eval()
    assert not m._folded and WaveGlow(**synthetic.WAVEFLOW_CONFIGS["toy"])._folded
    mel = synthetic.synthetic_mel(2, 5, seed=21)
    melp = np.pad(mel, ((0, 0), (0, 0), (0, 1)))
    z = np.random.default_rng(21).standard_normal((2, 5 * cfg["hop_length"])).astype(np.float32) * np.float32(0.7)
    ref = wf.waveflow_inverse(sd, cfg, z, melp)
    got, _ = m.inverse(torch.from_numpy(z).cuda(), torch.from_numpy(melp).cuda())
    assert rms_rel_err(got.numpy(), ref) < WAVE_TOL
    assert rms_rel_err(got.numpy(), wf.waveflow_inverse(sd, synthetic.WAVEFLOW_CONFIGS["toy"], z, melp)) > 1e-2   # it matters


@pytest.mark.gpu
@pytest.mark.parametrize("name", ALL)
def test_hip_matches_reference_golden(hip_lib_path, name):
    g, cfg, _ = _load(name)
    m, _, _ = _model(str(g["config_key"]), int(g["seed"]))
    melp = torch.from_numpy(np.pad(g["mel"], ((0, 0), (0, 0), (0, 1)))).cuda()
    ids = None if _ids(g) is None else torch.from_numpy(_ids(g)).cuda()
    audio, _ = m.inverse(torch.from_numpy(g["z"]).cuda(), melp, speaker_ids=ids)
    assert not audio.is_cuda                                   # return_CPU=True default (ax:348-349)
    err = rms_rel_err(audio.numpy(), g["inverse_full"])
    print(f"waveflow {name}: rms rel err vs reference = {err:.3e}")
    assert err < WAVE_TOL


@pytest.mark.gpu
def test_hip_infer_contract_and_ragged_vs_oracle(hip_lib_path):
    m, cfg, sd = _model("toy", 9)
    B, Fr = 2, 37                                              # L = 36*256/8 ... ragged vs the 256-step tile
    mel = synthetic.synthetic_mel(B, Fr, seed=3)
    torch.manual_seed(4)
    out = m.infer(torch.from_numpy(mel).cuda(), sigma=0.8, return_CPU=False)
    assert out.is_cuda and out.shape == (B, (Fr - 1) * 256) and torch.isfinite(out).all()
    z = np.random.default_rng(5).standard_normal((B, Fr * 256)).astype(np.float32) * np.float32(0.8)
    melp = np.pad(mel, ((0, 0), (0, 0), (0, 1)))
    ref = wf.waveflow_inverse(sd, cfg, z, melp)
    got, _ = m.inverse(torch.from_numpy(z).cuda(), torch.from_numpy(melp).cuda())
    assert rms_rel_err(got.numpy(), ref) < WAVE_TOL
    # NaN in the latent is zeroed per flow, like ax:333-334
    z2 = z.copy()
    z2[0, 100] = np.nan
    got2, _ = m.inverse(torch.from_numpy(z2).cuda(), torch.from_numpy(melp).cuda())
    assert torch.isfinite(got2).all()


@pytest.mark.gpu
def test_fused_res_skip_epilogue_matches_two_kernel_path(hip_lib_path, tuning):
    """C = 64 runs the res/skip GEMM inside the in-layer kernel; CTTS_WF_NO_FUSE keeps the two-launch path."""
    g, cfg, _ = _load("toy")
    m, _, _ = _model(str(g["config_key"]), int(g["seed"]))
    melp = torch.from_numpy(np.pad(g["mel"], ((0, 0), (0, 0), (0, 1)))).cuda()
    z = torch.from_numpy(g["z"]).cuda()
    fused, _ = m.inverse(z, melp)
    tuning.set("CTTS_WF_NO_FUSE")
    plain, _ = m.inverse(z, melp)
    assert rms_rel_err(plain.numpy(), g["inverse_full"]) < WAVE_TOL
    assert rms_rel_err(fused.numpy(), plain.numpy()) < 1e-5


@pytest.mark.gpu
def test_author_options_ragged_vs_oracle_and_speaker_requirement(hip_lib_path, tuning):
    """8f.4 option set at a width that is not a multiple of 4 or of the 256-column tile, batch 3, both fused
    (C = 64) and two-kernel res/skip paths; a multispeaker model refuses to run without speaker ids (ax:288)."""
    m, cfg, sd = _model("author_toy", 12)
    B, Fr = 3, 71
    n_in = cfg["n_mel_channels"] * 2
    mel = synthetic.synthetic_mel(B, Fr, n_in, seed=8)
    ids = np.array([0, 511, 42], np.int64)
    z = (np.random.default_rng(9).standard_normal((B, (Fr - 1) * cfg["hop_length"])) * 0.8).astype(np.float32)
    ref = wf.waveflow_inverse(sd, cfg, z, mel, ids)
    got, _ = m.inverse(torch.from_numpy(z).cuda(), torch.from_numpy(mel).cuda(), speaker_ids=torch.from_numpy(ids).cuda())
    assert rms_rel_err(got.numpy(), ref) < WAVE_TOL
    tuning.set("CTTS_WF_NO_FUSE")
    plain, _ = m.inverse(torch.from_numpy(z).cuda(), torch.from_numpy(mel).cuda(), speaker_ids=torch.from_numpy(ids).cuda())
    tuning.clear("CTTS_WF_NO_FUSE")
    assert rms_rel_err(plain.numpy(), ref) < WAVE_TOL
    with pytest.raises(Exception, match="requires speaker ids"):
        m.inverse(torch.from_numpy(z).cuda(), torch.from_numpy(mel).cuda())
    out = m.infer(torch.from_numpy(mel).cuda(), speaker_ids=torch.from_numpy(ids).cuda(), sigma=0.7, return_CPU=False)
    assert out.is_cuda and out.shape == (B, (Fr - 1) * cfg["hop_length"]) and torch.isfinite(out).all()


@pytest.mark.gpu
def test_conditioning_interpolation_kernels_are_bit_identical(hip_lib_path, tuning):
    """The 16-byte-store and the scalar form of the WaveFlow conditioning interpolation (glow_ax.py:545-554) share one
    explicitly contracted lerp: the waveform must not depend on which one the geometry selects."""
    m, cfg, sd = _model("author_toy", 12)
    B, Fr = 2, 23
    mel = torch.from_numpy(synthetic.synthetic_mel(B, Fr, cfg["n_mel_channels"] * 2, seed=3)).cuda()
    ids = torch.from_numpy(np.array([5, 77], np.int64)).cuda()
    z = torch.from_numpy((np.random.default_rng(4).standard_normal((B, (Fr - 1) * cfg["hop_length"])) * 0.8).astype(np.float32)).cuda()
    vec, _ = m.inverse(z, mel, speaker_ids=ids)
    tuning.set("CTTS_WF_NO_VEC_INTERP")
    scalar, _ = m.inverse(z, mel, speaker_ids=ids)
    tuning.clear("CTTS_WF_NO_VEC_INTERP")
    assert torch.equal(vec, scalar)


@pytest.mark.gpu
def test_author_full_width_fused_1x1_stages_vs_oracle(hip_lib_path, tuning):
    """The author's full option set (C = 128: depthwise launch + fused pointwise/gate/res-skip kernel), two utterances,
    L = 150 (three 64-column tiles, ragged, not a multiple of 4), against the oracle and against the unfused launches."""
    m, cfg, sd = _model("author", 31)
    B, Fr = 2, 6
    mel = synthetic.synthetic_mel(B, Fr, cfg["n_mel_channels"] * 2, seed=5)
    ids = np.array([7, 300], np.int64)
    z = (np.random.default_rng(6).standard_normal((B, (Fr - 1) * cfg["hop_length"])) * 0.7).astype(np.float32)
    ref = wf.waveflow_inverse(sd, cfg, z, mel, ids)
    args = (torch.from_numpy(z).cuda(), torch.from_numpy(mel).cuda())
    got, _ = m.inverse(*args, speaker_ids=torch.from_numpy(ids).cuda())
    assert rms_rel_err(got.numpy(), ref) < WAVE_TOL
    tuning.set("CTTS_WF_NO_FUSE")
    plain, _ = m.inverse(*args, speaker_ids=torch.from_numpy(ids).cuda())
    tuning.clear("CTTS_WF_NO_FUSE")
    assert rms_rel_err(plain.numpy(), ref) < WAVE_TOL
    assert rms_rel_err(got.numpy(), plain.numpy()) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["author_toy", "author_toy_gate"])
def test_author_options_under_the_split_gemm_modes(hip_lib_path, name):
    """The author's option set (separable in-layers: the fused pointwise + res/skip kernel of waveflow_sep.hip, conditioning
    stacks, speaker embeddings) under the split-bf16 GEMM loops: three products stay far inside the waveform bound, six
    products (the fused separable layer keeps fp32 MFMA there, the conv-GEMMs around it take the six-product loop) stay
    within 2x of the fp32 MFMA path's error."""
    g, cfg, _ = _load(name)
    m, _, _ = _model(str(g["config_key"]), int(g["seed"]))
    melp = torch.from_numpy(np.pad(g["mel"], ((0, 0), (0, 0), (0, 1)))).cuda()
    ids = None if _ids(g) is None else torch.from_numpy(_ids(g)).cuda()
    z = torch.from_numpy(g["z"]).cuda()
    err = {}
    for mode in ("f32", "bf16x3", "bf16x6"):
        m.set_f32_gemm_mode(mode)
        err[mode] = rms_rel_err(m.inverse(z, melp, speaker_ids=ids)[0].numpy(), g["inverse_full"])
    print(f"waveflow {name}: rms rel err vs reference: fp32 MFMA {err['f32']:.3e}, bf16x3 {err['bf16x3']:.3e}, bf16x6 {err['bf16x6']:.3e}")
    assert err["f32"] < WAVE_TOL and err["bf16x3"] < 1e-4 and err["bf16x6"] <= 2.0 * err["f32"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["author_toy", "untts_toy"])
def test_batched_conditioning_convs_equal_the_per_utterance_launches(hip_lib_path, name):
    """The conditioning stacks run each conv as ONE launch for the batch when the buffers have the operator's exact row
    counts; same kernel, same K order: bit-identical to one launch per utterance."""
    from cookietts_amd.waveglow_ax import _CondConv
    g, cfg, _ = _load(name)
    m, _, _ = _model(str(g["config_key"]), int(g["seed"]))
    melp = torch.from_numpy(np.pad(g["mel"], ((0, 0), (0, 0), (0, 1)))).cuda()
    ids = None if _ids(g) is None else torch.from_numpy(_ids(g)).cuda()
    z = torch.from_numpy(g["z"]).cuda()
    batched, _ = m.inverse(z, melp, speaker_ids=ids)
    try:
        _CondConv.batched = False
        looped, _ = m.inverse(z, melp, speaker_ids=ids)
    finally:
        _CondConv.batched = True
    assert torch.equal(batched, looped)
