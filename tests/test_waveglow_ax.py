"""ax-core WaveGlow with waveflow=False: AffineCouplingBlock + 1-D WN, InvertibleConv1x1 /
PermuteHeight mixing in both mix_first orders, early outputs.  Oracle vs the reference's own outputs (CPU), HIP path
through the C ABI vs the same goldens and the oracle (GPU)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rms_rel_err
from cookietts_amd import synthetic
from oracle import waveglow_ax_oracle as ao

WAVE_TOL = 1e-3           # BASELINE.json: waveform RMS relative error
ORACLE_TOL = 5e-6
SMALL = ["toy_conv", "toy_conv_mixlast", "toy_permute", "toy_permute_mixfirst", "notebook_toy", "untts_toy", "toy_merge",
         "toy_groupconv", "toy_groupconv_dense", "toy_wn_tconv", "toy_wn_tconv_crop",
         "toy_sigmoid_vol", "toy_no_res_skip", "toy_no_res_skip_1layer",
         "toy_dilations", "toy_dilations_const", "toy_c96", "toy_c160", "toy_g32", "toy_g32_permute"]
GATES = sorted(k for k in synthetic.WAVEGLOW_AX_CONFIGS if k.startswith("toy_gate_"))       # the 13 non-GTU units


def _load(key):
    g = np.load(os.path.join(GOLDEN, f"waveglow_ax_{key}.npz"))
    cfg = synthetic.WAVEGLOW_AX_CONFIGS[str(g["config_key"])]
    return g, cfg, synthetic.waveglow_ax_state_dict(cfg, seed=int(g["seed"]))


def _ids(g):
    return g["speaker_ids"] if "speaker_ids" in g.files else None


@pytest.mark.parametrize("key", SMALL + GATES)
def test_oracle_matches_reference(key):
    g, cfg, sd = _load(key)
    melp = np.pad(g["mel"], ((0, 0), (0, 0), (0, 1)))
    # the SIREN units take sin(16 x): the fp32 summation-order noise of x comes out 16 x larger (measured 5e-6 .. 8e-6)
    tol = 10 * ORACLE_TOL if "gsir" in key else ORACLE_TOL
    assert rms_rel_err(ao.waveglow_ax_inverse(sd, cfg, g["z"], melp, _ids(g)), g["inverse_full"]) < tol
    audio = ao.waveglow_ax_infer(sd, cfg, g["mel"], g["z"], speaker_ids=_ids(g))
    assert audio.shape == g["audio"].shape and rms_rel_err(audio, g["audio"]) < tol


def test_oracle_pieces():
    # early-output bookkeeping (ax:170-189): every n_early_every flows two channels leave the latent
    cfg = synthetic.WAVEGLOW_AX_CONFIGS["notebook"]
    ch = ao.flow_channels(cfg)
    assert ch[0] == 24 and ch[15] == 24 and ch[16] == 22 and ch[32] == 20 and ch[47] == 20
    # the substring test of ax:24-25: 'permute' selects PermuteHeight, the default '1x1conv' the invertible conv
    assert ao.mixing_kind({"channel_mixing": "permute"}) == 'permuteheight'
    assert ao.mixing_kind({"channel_mixing": "InvertibleConv1x1"}) == '1x1conv' and ao.mixing_kind({}) == '1x1conv'
    # replicate padding differs from zero padding only in the k//2 edge columns
    from oracle.waveflow_oracle import conv1d_same
    rng = np.random.default_rng(0)
    x = rng.standard_normal((1, 3, 9)).astype(np.float32)
    w = rng.standard_normal((2, 3, 3)).astype(np.float32)
    b = np.zeros(2, np.float32)
    yz, yr = conv1d_same(x, w, b), conv1d_same(x, w, b, 'replicate')
    assert np.array_equal(yz[:, :, 1:-1], yr[:, :, 1:-1]) and not np.allclose(yz[:, :, 0], yr[:, :, 0])
    assert np.allclose(yr[:, :, 0], w[:, :, 0] @ x[0, :, 0] + w[:, :, 1] @ x[0, :, 0] + w[:, :, 2] @ x[0, :, 1], atol=1e-6)


def test_host_state_dict_keys_match_reference_format():
    """The goldens were produced by loading these recipes into the reference with strict=True, so equality with the
    recipe IS equality with the reference's own state_dict keys and shapes."""
    from cookietts_amd.waveglow_ax import WaveGlow
    for key, cfg in synthetic.WAVEGLOW_AX_CONFIGS.items():
        if key in ("notebook", "untts") or key.startswith("toy_gate_"):
            continue
        sd = synthetic.waveglow_ax_state_dict(cfg, seed=1)
        own = WaveGlow(**cfg).state_dict()
        assert sorted(own) == sorted(sd)
        assert all(tuple(own[k].shape) == sd[k].shape for k in sd)
    m = WaveGlow(**synthetic.WAVEGLOW_AX_CONFIGS["notebook_toy"])
    assert m.z_split_sizes == [2, 2, 8] and m.channel_mixing == 'permuteheight' and m.mix_first is False
    assert float(m.WN[0].WN.end.weight.abs().max()) == 0.0                  # zero-init end (glow_ax.py:278-281)
    with pytest.raises(NotImplementedError):
        WaveGlow(**dict(synthetic.WAVEGLOW_AX_CONFIGS["toy_conv"], upsample_first=True))       # no TransposedUpsampleNet
    WaveGlow(**dict(synthetic.WAVEGLOW_AX_CONFIGS["untts_toy"], hop_length=48))   # 2*3 == 48 // 8: cropped, not interpolated
    with pytest.raises(Exception, match="gated_unit is invalid"):                                 # glow_ax.py:198
        WaveGlow(**synthetic.waveglow_ax_config(WN=dict(gated_unit="GXU")))
    mg = WaveGlow(**synthetic.WAVEGLOW_AX_CONFIGS["toy_merge"])
    assert all(tuple(l.weight_v.shape) == (128, 128, 1) for l in mg.WN[0].WN.res_skip_layers)      # merge_res_skip: C rows
    assert mg.c_config_1d().gated_unit == 1 and mg.c_config_1d().merge_res_skip == 1
    u = WaveGlow(**synthetic.WAVEGLOW_AX_CONFIGS["untts_toy"])
    assert [type(t).__name__ for t in u.upsample_net.t_convs] == ["ConvTranspose1d", "LeakyReLU"] * 2
    assert tuple(u.res_conv.weight.shape) == (48, 24, 1) and u.upsample_net.res_weight is not None
    with pytest.raises(AssertionError):
        WaveGlow(**dict(synthetic.WAVEGLOW_AX_CONFIGS["toy_conv"], channel_mixing='shuffle'))


def test_c_abi_size_queries_and_argument_validation(hip_lib_path):
    import ctypes as C
    from cookietts_amd import _lib
    lib = _lib.lib()
    cfg = _lib.WgaxConfig(n_flows=48, n_group=24, n_early_every=16, n_early_size=2, n_layers=8, n_channels=256,
                          kernel_size=3, mixing=_lib.MIX_PERMUTE, mix_first=0, ignore_nan=1)
    nbytes = lib.ctts_wgax_packed_bytes(C.byref(cfg))
    # 48 flows x 8 layers x (in 512x768 + res/skip 512x256 (last 256x256)) fp32 + small stuff
    dense = 48 * (8 * 512 * 768 + 7 * 512 * 256 + 256 * 256) * 4
    assert dense < nbytes < 1.2 * dense
    assert lib.ctts_wgax_workspace_bytes(C.byref(cfg), 1, 24 * 11675) > 0
    assert lib.ctts_wgax_workspace_bytes(C.byref(cfg), 1, 24 * 11675 + 1) == 0          # not a multiple of n_group
    assert b"multiple of n_group" in lib.ctts_last_error()
    for field, bad in (("n_group", 34), ("n_channels", 144), ("kernel_size", 4), ("mixing", 2), ("n_early_size", 3)):
        c2 = _lib.WgaxConfig.from_buffer_copy(cfg)
        setattr(c2, field, bad)
        assert lib.ctts_wgax_packed_bytes(C.byref(c2)) == 0, field
    c3 = _lib.WgaxConfig.from_buffer_copy(cfg)
    c3.n_flows = 47                                                                     # PermuteHeight: even n_flows
    assert lib.ctts_wgax_packed_bytes(C.byref(c3)) == 0
    assert lib.ctts_replicate_halo_f32(None, 1, 1, 4, 32, 8, 1, None) != 0
    # the row operators of the model-level upsampling path validate before they launch (no GPU needed for the refusals)
    assert lib.ctts_affine_rows_f32(None, 1, 16, 16, 8, 64, 8, 1.0, 1.0, None) != 0
    assert lib.ctts_resample_rows_f32(None, None, 1, 16, 8, 64, 8, 40, 128, 8, 0, 0.0, None) != 0
    assert lib.ctts_interleave_phases_f32(None, None, 1, 16, 2, 1, 8, 64, 8, 16, 64, 8, None) != 0
    assert b"interleave_phases" in lib.ctts_last_error()


def _model(key, seed):
    from cookietts_amd.waveglow_ax import WaveGlow
    cfg = synthetic.WAVEGLOW_AX_CONFIGS[key]
    sd = synthetic.waveglow_ax_state_dict(cfg, seed=seed)
    m = WaveGlow(**cfg)
    m.load_state_dict(synthetic.to_torch(sd))
    return m.cuda().eval(), cfg, sd


@pytest.mark.gpu
@pytest.mark.parametrize("key", SMALL + GATES)
def test_hip_matches_reference_golden(hip_lib_path, key):
    g, cfg, _ = _load(key)
    m, _, _ = _model(str(g["config_key"]), int(g["seed"]))
    melp = torch.from_numpy(np.pad(g["mel"], ((0, 0), (0, 0), (0, 1)))).cuda()
    ids = None if _ids(g) is None else torch.from_numpy(_ids(g)).cuda()
    audio, _ = m.inverse(torch.from_numpy(g["z"]).cuda(), melp, speaker_ids=ids)
    assert not audio.is_cuda                                   # return_CPU=True default (ax:348-349)
    err = rms_rel_err(audio.numpy(), g["inverse_full"])
    print(f"waveglow_ax {key}: rms rel err vs reference = {err:.3e}")
    assert err < WAVE_TOL
    if key.endswith("conv"):
        assert m.convinv[0].W_inverse.shape == (cfg["n_group"], cfg["n_group"], 1)     # cached like em:271-276


def test_oracle_transposed_upsampling_pieces():
    """conv_transpose1d / interp_scale against torch's own operators (the reference calls exactly these)."""
    import torch.nn.functional as F
    from oracle.waveflow_oracle import conv_transpose1d, interp_scale
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 5, 7)).astype(np.float32)
    for k, s in ((4, 2), (9, 3), (5, 5), (3, 1)):
        w = rng.standard_normal((5, 6, k)).astype(np.float32)
        b = rng.standard_normal(6).astype(np.float32)
        ref = F.conv_transpose1d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), stride=s, padding=(k - s) // 2)
        got = conv_transpose1d(x, w, b, s, (k - s) // 2)
        assert got.shape == tuple(ref.shape) == (2, 6, 7 * s) and np.abs(got - ref.numpy()).max() < 1e-5
    for scale in (6, 30):
        for linear in (True, False):
            ref = F.interpolate(torch.from_numpy(x), scale_factor=scale, mode='linear' if linear else 'nearest',
                                **({"align_corners": False} if linear else {}))
            assert np.abs(interp_scale(x, scale, linear) - ref.numpy()).max() < 1e-6


def test_oracle_untts_width_matches_reference():
    """24 flows x 8 x 384, 256 mel channels, model-level transposed-conv upsampling: the untts notebook's vocoder."""
    g, cfg, sd = _load("untts")
    melp = np.pad(g["mel"], ((0, 0), (0, 0), (0, 1)))
    assert rms_rel_err(ao.waveglow_ax_inverse(sd, cfg, g["z"], melp, _ids(g)), g["inverse_full"]) < ORACLE_TOL


@pytest.mark.gpu
def test_hip_untts_width_matches_reference_golden(hip_lib_path):
    g, cfg, _ = _load("untts")
    m, _, _ = _model("untts", int(g["seed"]))
    melp = torch.from_numpy(np.pad(g["mel"], ((0, 0), (0, 0), (0, 1)))).cuda()
    audio, _ = m.inverse(torch.from_numpy(g["z"]).cuda(), melp, speaker_ids=torch.from_numpy(g["speaker_ids"]).cuda())
    err = rms_rel_err(audio.numpy(), g["inverse_full"])
    print(f"waveglow_ax untts: rms rel err vs reference = {err:.3e}")
    assert err < WAVE_TOL


@pytest.mark.gpu
def test_hip_notebook_width_matches_reference_golden(hip_lib_path):
    """48 flows x 8 x 256, n_group 24, 160 mel channels: the reference's timed config at full width, short mel."""
    path = os.path.join(GOLDEN, "waveglow_ax_notebook.npz")
    g = np.load(path)
    m, cfg, _ = _model("notebook", int(g["seed"]))
    melp = torch.from_numpy(np.pad(g["mel"], ((0, 0), (0, 0), (0, 1)))).cuda()
    audio, _ = m.inverse(torch.from_numpy(g["z"]).cuda(), melp, speaker_ids=torch.from_numpy(g["speaker_ids"]).cuda())
    err = rms_rel_err(audio.numpy(), g["inverse_full"])
    print(f"waveglow_ax notebook: rms rel err vs reference = {err:.3e}")
    assert err < WAVE_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("B,Fr", [(3, 5), (1, 1), (2, 37)])
def test_hip_untts_upsampling_ragged_lengths_vs_oracle(hip_lib_path, B, Fr):
    """Model-level transposed-conv upsampling at lengths the goldens do not have (a single frame; 37 frames = 1480
    latent columns, ragged against every tile width), different speakers per utterance, batch independence."""
    m, cfg, sd = _model("untts_toy", 11)
    mel = synthetic.synthetic_mel(B, Fr, cfg["n_mel_channels"], seed=Fr)
    melp = np.pad(mel, ((0, 0), (0, 0), (0, 1)))
    ids = np.array([5, 400, 77][:B], np.int64)
    z = np.random.default_rng(Fr).standard_normal((B, Fr * cfg["hop_length"])).astype(np.float32) * np.float32(0.7)
    ref = ao.waveglow_ax_inverse(sd, cfg, z, melp, ids)
    tz, tm, ti = torch.from_numpy(z).cuda(), torch.from_numpy(melp).cuda(), torch.from_numpy(ids).cuda()
    got, _ = m.inverse(tz, tm, speaker_ids=ti, return_CPU=False)
    err = rms_rel_err(got.cpu().numpy(), ref)
    print(f"untts_toy B={B} F={Fr}: rms rel err vs oracle = {err:.3e}")
    assert err < WAVE_TOL
    one, _ = m.inverse(tz[-1:], tm[-1:], speaker_ids=ti[-1:], return_CPU=False)
    assert torch.equal(one[0], got[-1])


@pytest.mark.gpu
def test_hip_infer_contract_ragged_nan_and_batch_independence(hip_lib_path):
    m, cfg, sd = _model("toy_conv_mixlast", 9)
    B, Fr = 3, 23                                              # L = 22*240/12 = 440: ragged vs the 128-step tile
    mel = synthetic.synthetic_mel(B, Fr, seed=3)
    torch.manual_seed(4)
    out = m.infer(torch.from_numpy(mel).cuda(), sigma=0.8, return_CPU=False)
    assert out.is_cuda and out.shape == (B, (Fr - 1) * 240) and torch.isfinite(out).all()
    z = np.random.default_rng(5).standard_normal((B, Fr * 240)).astype(np.float32) * np.float32(0.8)
    melp = np.pad(mel, ((0, 0), (0, 0), (0, 1)))
    ref = ao.waveglow_ax_inverse(sd, cfg, z, melp)
    tz, tm = torch.from_numpy(z).cuda(), torch.from_numpy(melp).cuda()
    got, _ = m.inverse(tz, tm, return_CPU=False)
    assert rms_rel_err(got.cpu().numpy(), ref) < WAVE_TOL
    for b in range(B):                                         # utterances do not interact; workspace reuse
        one, _ = m.inverse(tz[b:b + 1], tm[b:b + 1], return_CPU=False)
        assert torch.equal(one[0], got[b])
    again, _ = m.inverse(tz, tm, return_CPU=False)
    assert torch.equal(again, got)
    # NaN in the latent: zeroed after the coupling of every flow like ax:333-334 - same result as the oracle's
    z2 = z.copy()
    z2[0, 100] = np.nan
    got2, _ = m.inverse(torch.from_numpy(z2).cuda(), tm)
    ref2 = ao.waveglow_ax_inverse(sd, cfg, z2, melp)
    assert torch.isfinite(got2).all() and rms_rel_err(got2.numpy(), ref2) < WAVE_TOL


@pytest.mark.gpu
def test_hip_permute_moves_nan_like_the_reference(hip_lib_path):
    """PermuteHeight is an index shuffle, not a 0/1 matrix product: a NaN row stays ONE row until ignore_nan zeroes it."""
    m, cfg, sd = _model("toy_permute", 11)
    mel = synthetic.synthetic_mel(1, 6, seed=1)
    melp = np.pad(mel, ((0, 0), (0, 0), (0, 1)))
    z = np.random.default_rng(2).standard_normal((1, 6 * 256)).astype(np.float32)
    z[0, 8 * 50 + 7] = np.nan                                   # last latent row at step 50
    got, _ = m.inverse(torch.from_numpy(z).cuda(), torch.from_numpy(melp).cuda())
    ref = ao.waveglow_ax_inverse(sd, cfg, z, melp)
    assert np.isfinite(got.numpy()).all() and rms_rel_err(got.numpy(), ref) < WAVE_TOL
