"""``_5_infer`` vocoder slot for the MI355X WaveGlow path.

The reference server wires Tacotron2 -> a vocoder object through two call sites only
(``/root/reference/CookieTTS/_5_infer/t2s_server/text2speech.py``):

  * ``self.vocoder, self.vocoder_config = self.load_hifigan(path)``            (:175-179, 258-279)
  * ``dtype = next(self.vocoder.parameters()).dtype``;
    ``self.vocoder(mel[b<=16, n_mel, T].to(dtype)).squeeze(1).cpu().split(1, 0)``   (:658-665)

``WaveGlowVocoder`` satisfies that contract on top of ``cookietts_amd.WaveGlow`` (``forward(mel) -> [b, 1, T*hop]``,
``parameters()`` report the compute dtype's natural input dtype), and ``load_waveglow`` mirrors ``load_hifigan``:
it reads a reference-format WaveGlow checkpoint (``{'model', 'waveglow_config', 'speaker_lookup', ...}``,
``_4_mtw/waveglow/train.py:128-145``, including the legacy key renames of :121) and returns ``(vocoder, config)``.
"""
from __future__ import annotations

import torch

from .waveglow import WaveGlow

__all__ = ["WaveGlowVocoder", "load_waveglow", "waveglow_from_checkpoint"]


class WaveGlowVocoder(torch.nn.Module):
    def __init__(self, waveglow: WaveGlow, sigma: float = 0.8):
        super().__init__()
        self.waveglow = waveglow
        self.sigma = sigma

    @torch.no_grad()
    def forward(self, mel):
        """mel [b, n_mel, T] -> audio [b, 1, T*hop] on the model's device, in mel's dtype."""
        audio = self.waveglow.infer(mel.to(next(self.waveglow.parameters()).device), sigma=self.sigma)
        return audio.unsqueeze(1)

    def half(self):
        """``load_hifigan`` calls ``vocoder.half()`` (text2speech.py:261): the WN stacks then run on IEEE-half storage and
        fp16 MFMA with fp32 accumulation (``set_compute_dtype(torch.float16)``, the reference's own half mode, inside the
        1e-3 waveform bound) from fp32 master weights - the parameters, and hence the dtype the server casts its mels to,
        stay fp32.  (Until round 5 this selected the bf16 path: same speed, 8x the error.)"""
        self.waveglow.set_compute_dtype(torch.float16)
        return self


def waveglow_from_checkpoint(checkpoint: dict) -> WaveGlow:
    cfg = checkpoint['waveglow_config']
    model = WaveGlow(**cfg)
    sd = checkpoint['model']
    sd = {k.replace("invconv1x1", "convinv").replace(".F.", ".WN.").replace("WNs.", "WN."): v for k, v in sd.items()}
    model.load_state_dict(sd)
    return model


def load_waveglow(vocoder_path, device='cuda', sigma=0.8, trust_checkpoint=False):
    """Counterpart of ``T2S.load_hifigan``: returns ``(vocoder, vocoder_config)``.

    A reference checkpoint is a dict of tensors, a config dict and a speaker lookup: it loads with
    ``weights_only=True`` (no pickle code execution).  Checkpoints that pickle other objects (some training runs
    store the optimizer / hparams objects) need ``trust_checkpoint=True`` - only for files you produced yourself."""
    try:
        checkpoint = torch.load(vocoder_path, map_location='cpu', weights_only=True)
    except Exception:
        if not trust_checkpoint:
            raise
        checkpoint = torch.load(vocoder_path, map_location='cpu', weights_only=False)
    model = waveglow_from_checkpoint(checkpoint).to(device).eval()
    return WaveGlowVocoder(model, sigma=sigma), checkpoint['waveglow_config']
