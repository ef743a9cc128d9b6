"""BASELINE.json's full sizes, through size-independent properties (the oracle cannot run them in test time):
utterance independence, locality (finite receptive field), attention-weight invariants, run-to-run determinism."""
import numpy as np
import pytest
import torch

from conftest import rms_rel_err
from cookietts_amd import synthetic

pytestmark = pytest.mark.gpu


def test_config2_waveglow_full_size_properties(hip_lib_path):
    """Config 2: 12 flows x 512 channels, B = 8 x (80 x 900) mel, fp32."""
    from cookietts_amd import WaveGlow
    cfg = synthetic.WAVEGLOW_CONFIGS["full"]
    m = WaveGlow(**cfg)
    m.load_state_dict(synthetic.to_torch(synthetic.waveglow_state_dict(cfg, seed=77)))
    m = m.cuda().eval()
    B, F = 8, 900
    mel = torch.from_numpy(synthetic.synthetic_mel(B, F, seed=1)).cuda()
    z = torch.from_numpy(synthetic.synthetic_noise(B, cfg["n_group"], F * 32, seed=1) * np.float32(0.6)).cuda()
    full = m.infer_from_noise(mel, z)
    assert full.shape == (B, F * 256) and torch.isfinite(full).all()
    assert torch.equal(full, m.infer_from_noise(mel, z))                      # deterministic, workspace reused
    for b in (0, 7):                                                           # utterances do not interact
        assert torch.equal(m.infer_from_noise(mel[b:b + 1], z[b:b + 1])[0], full[b])
    # locality: a WN sees +-255 steps, 12 flows -> 3060 steps = 96 frames; frames 0..299 of a 600-frame cut must
    # reproduce the full run (different tile count / ragged edge, same arithmetic per column)
    cut = m.infer_from_noise(mel[:2, :, :600].contiguous(), z[:2, :, :600 * 32].contiguous())
    n = 300 * 256
    assert rms_rel_err(cut[:, :n].cpu().numpy(), full[:2, :n].cpu().numpy()) < 1e-5


def test_config4_waveflow_full_size_properties(hip_lib_path):
    """Config 4: 8 flows x 64 channels, n_group 16, B = 8 x (80 x 900) mel."""
    from cookietts_amd import WaveFlow
    cfg = synthetic.WAVEFLOW_CONFIGS["full"]
    m = WaveFlow(**cfg)
    m.load_state_dict(synthetic.to_torch(synthetic.waveflow_state_dict(cfg, seed=78)))
    m = m.cuda().eval()
    B, F = 8, 901                                                              # inverse() sees the padded mel
    mel = torch.from_numpy(synthetic.synthetic_mel(B, F, seed=2)).cuda()
    g = torch.Generator().manual_seed(5)
    z = (torch.randn(B, (F - 1) * 256, generator=g) * 0.6).cuda()
    full, _ = m.inverse(z, mel, return_CPU=False)
    assert full.shape == (B, (F - 1) * 256) and torch.isfinite(full).all()
    again, _ = m.inverse(z, mel, return_CPU=False)
    assert torch.equal(full, again)
    for b in (0, 7):
        one, _ = m.inverse(z[b:b + 1], mel[b:b + 1], return_CPU=False)
        assert torch.equal(one[0], full[b])


def test_config5_tacotron_decoder_full_size_properties(hip_lib_path):
    """Config 5: B = 4, 200 symbols, 900 forced decoder steps."""
    from cookietts_amd.tacotron2 import Tacotron2
    hp = synthetic.tacotron_hparams()
    m = Tacotron2(hp)
    m.load_state_dict(synthetic.to_torch(synthetic.tacotron_state_dict(hp, seed=79)))
    m = m.cuda().eval()
    B, T, steps = 4, 200, 900
    rng = np.random.default_rng(3)
    mem = torch.from_numpy((rng.standard_normal((B, T, synthetic.tacotron_memory_in_dim(hp))) * 0.5).astype(np.float32)).cuda()
    lens = torch.tensor([200, 195, 150, 100]).cuda()
    masks = synthetic.prenet_dropout_masks(steps, B, seed=4)
    mel, gate, align, _ = m.decoder.inference(mem, lens, keep_masks=masks, fixed_steps=steps)
    assert mel.shape == (B, hp.n_mel_channels, steps) and gate.shape == (B, steps) and align.shape == (B, steps, T)
    assert torch.isfinite(mel).all() and torch.isfinite(gate).all()
    a = align.cpu().numpy()
    assert np.allclose(a.sum(axis=2), 1.0, atol=1e-5)                          # softmax rows
    assert ((a > 0).sum(axis=2) <= 33).all()                                   # window +-16 (model.py:114-123)
    for b, n in enumerate([200, 195, 150, 100]):
        assert n == T or a[b, :, n:].max() == 0.0                              # padding never attended
    # one utterance alone == its row in the batch, and the run is deterministic
    one = m.decoder.inference(mem[2:3].contiguous(), lens[2:3], keep_masks=np.ascontiguousarray(masks[:, :, 2:3]),
                              fixed_steps=steps)
    assert (one[0][0] - mel[2]).abs().max() < 1e-4 and (one[2][0] - align[2]).abs().max() < 1e-4
    rep = m.decoder.inference(mem, lens, keep_masks=masks, fixed_steps=steps)
    assert torch.equal(rep[0], mel)
