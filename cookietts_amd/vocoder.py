"""``_5_infer`` vocoder slot for the MI355X WaveGlow / WaveFlow paths.

The reference server wires Tacotron2 -> a vocoder object through two call sites only
(``/root/reference/CookieTTS/_5_infer/t2s_server/text2speech.py``):

  * ``self.vocoder, self.vocoder_config = self.load_hifigan(path)``            (:175-179, 258-279)
  * ``dtype = next(self.vocoder.parameters()).dtype``;
    ``self.vocoder(mel[b<=16, n_mel, T].to(dtype)).squeeze(1).cpu().split(1, 0)``   (:658-665)

``WaveGlowVocoder`` satisfies that contract (``forward(mel) -> [b, 1, samples]``, ``parameters()`` report the dtype the
server casts its mels to) and ``load_waveglow`` mirrors ``load_hifigan``: it reads a reference-format checkpoint
(``{'model', 'waveglow_config', 'speaker_lookup', ...}``, ``_4_mtw/waveglow/train.py:128-145``, including the legacy key
renames of :121) and returns ``(vocoder, config)``.

Which class a checkpoint builds follows its ``waveglow_config``, as the reference's own trainer does
(``train.py:385-394``): that file hard-codes ``ax = True``, so every checkpoint cookietts trains carries the option set
of ``efficient_model_ax.WaveGlow`` (``upsample_first``, ``speaker_embed``, ``cond_layers``, ``waveflow`` ...:
efficient_model_ax.py:18-19) and builds ``cookietts_amd.waveglow_ax.WaveGlow`` (WaveFlow or the 1-D ax WaveGlow);
a config with only ``glow.py``'s thirteen keys (glow.py:225-226) builds ``cookietts_amd.WaveGlow``.
"""
from __future__ import annotations

import torch

from .waveglow import WaveGlow
from .waveglow_ax import WaveGlow as WaveGlowAx

__all__ = ["WaveGlowVocoder", "load_waveglow", "waveglow_from_checkpoint", "is_ax_config"]

# keyword arguments only efficient_model_ax.WaveGlow.__init__ has (efficient_model_ax.py:18-19): the first five are
# REQUIRED there, so every config written for the ax core names them; the rest are its optional extras
_AX_ONLY_KEYS = ("upsample_first", "speaker_embed", "cond_layers", "cond_hidden_channels", "cond_output_channels",
                 "cond_kernel_size", "cond_residual", "cond_padding_mode", "waveflow", "channel_mixing", "mix_first",
                 "preceived_vol_scaling", "shift_spect", "scale_spect", "preempthasis", "use_logvar_channels",
                 "transposed_conv_scales", "group_conv_output_dim", "iso226_empthasis", "sampling_rate")


def is_ax_config(cfg: dict) -> bool:
    """True when ``waveglow_config`` is written for efficient_model_ax.WaveGlow (what train.py:385-388 builds)."""
    return any(k in cfg for k in _AX_ONLY_KEYS)


class WaveGlowVocoder(torch.nn.Module):
    """``vocoder(mel)`` as text2speech.py:661-665 calls it, over either model family.

    * glow.py model: ``WaveGlow.infer(mel, sigma)`` -> ``[b, 1, T*hop]`` (glow.py:314-350).
    * ax model: ``WaveGlow.infer(mel, speaker_ids, sigma=sigma, return_CPU=False)`` -> ``[b, 1, (T-1)*hop]``: the ax
      ``infer`` pads one frame and trims ``artifact_trimming * hop`` samples from what the padded mel produces
      (efficient_model_ax.py:359-388); the audio stays on the device (the server's ``.cpu()`` at :665 moves it).

    ``speaker_ids`` (internal ids, as the model's embedding tables index them) is needed by checkpoints trained with a
    speaker embedding; ``speaker_lookup`` is the checkpoint's external-id -> internal-id table (train.py:140, used as
    train.py:248 does) and ``speaker_ids_for`` applies it.  ``noise`` replaces the internally drawn latent (already
    multiplied by sigma; the tests' deterministic entry) - shape ``[b, samples + hop]`` for an ax model,
    ``[b, n_group, T*hop/n_group]`` for a glow.py model.
    """

    def __init__(self, waveglow, sigma: float = 0.8, speaker_lookup=None):
        super().__init__()
        self.waveglow = waveglow
        self.sigma = sigma
        self.is_ax = isinstance(waveglow, WaveGlowAx)
        self.speaker_lookup = dict(speaker_lookup or {})

    def speaker_ids_for(self, external_ids):
        """External speaker ids (dataset ids, train.py:248) -> a LongTensor of the model's internal ids."""
        try:
            return torch.tensor([self.speaker_lookup[int(i)] for i in external_ids], dtype=torch.int64)
        except KeyError as e:
            raise KeyError(f"speaker id {e.args[0]} is not in the checkpoint's speaker_lookup") from None

    # lengths=: utterances whose length is within this factor of a bucket's longest share its call
    RAGGED_BUCKET_RATIO = 0.85
    RAGGED_MAX_BUCKETS = 4

    @torch.no_grad()
    def forward(self, mel, speaker_ids=None, noise=None, lengths=None):
        """mel [b, n_mel, T] -> audio [b, 1, samples] on the model's device, in mel's dtype.

        ``lengths`` [b] (frames; the server has them: ``output_lengths``, text2speech.py:646, 677) makes the call ragged-aware: the
        server pads every mel of a vocoder call to the longest with -11.52 (:651) and trims the audio afterwards (:677), so the
        padding frames are computed and thrown away.  With lengths the batch is cut into <= 4 buckets of similar length, each
        run at its own longest; every utterance's audio equals the reference's ``infer`` of a batch trimmed to that bucket
        length (the receptive field sees zeros past the end instead of padding frames), the tail beyond
        ``lengths[i] * hop`` is zero.  Without lengths: the reference's call, padding included."""
        device = next(self.waveglow.parameters()).device
        mel = mel.to(device)
        if speaker_ids is not None:
            speaker_ids = torch.as_tensor(speaker_ids).to(device)
        if lengths is not None and noise is None and mel.dim() == 3 and mel.shape[0] > 1:
            return self._forward_ragged(mel, speaker_ids, [int(x) for x in torch.as_tensor(lengths).reshape(-1).tolist()])
        if self.is_ax:
            if noise is None:
                audio = self.waveglow.infer(mel, speaker_ids=speaker_ids, sigma=self.sigma, return_CPU=False)
            else:
                audio = self.waveglow.infer_from_noise(mel, noise, speaker_ids=speaker_ids, return_CPU=False)
        elif noise is None:
            audio = self.waveglow.infer(mel, speaker_id=speaker_ids, sigma=self.sigma)
        else:
            audio = self.waveglow.infer_from_noise(mel, noise, speaker_id=speaker_ids).to(mel.dtype)
        return audio.unsqueeze(1)

    def _forward_ragged(self, mel, speaker_ids, lengths):
        b, _, T = mel.shape
        if len(lengths) != b or min(lengths) < 1 or max(lengths) > T:
            raise ValueError(f"lengths {lengths} do not describe a [{b}, n_mel, {T}] batch")
        order = sorted(range(b), key=lambda i: -lengths[i])
        buckets = [[order[0]]]
        for i in order[1:]:
            head = lengths[buckets[-1][0]]
            if lengths[i] >= self.RAGGED_BUCKET_RATIO * head or len(buckets) == self.RAGGED_MAX_BUCKETS:
                buckets[-1].append(i)
            else:
                buckets.append([i])
        out = None
        for idx in buckets:
            Tb = lengths[idx[0]]
            sel = torch.tensor(idx, device=mel.device)
            part = self.forward(mel.index_select(0, sel)[:, :, :Tb].contiguous(),
                                None if speaker_ids is None else speaker_ids.index_select(0, sel))
            if out is None:       # the longest bucket comes first: its sample count (glow.py: T hop; ax: (T - 1) hop) is the batch's
                out = torch.zeros(b, 1, part.shape[2] + (T - Tb) * self.waveglow.hop_length, dtype=part.dtype, device=part.device)
            out[sel, :, :part.shape[2]] = part
        hop = self.waveglow.hop_length
        for i in range(b):        # beyond an utterance's own frames: silence (the server trims there anyway)
            out[i, :, lengths[i] * hop:] = 0
        return out

    def half(self):
        """``load_hifigan`` calls ``vocoder.half()`` (text2speech.py:261): the reference's request for its reduced-precision mode.
        The parameters - and hence the dtype the server casts its mels to - stay fp32 masters in every case.

        * glow.py model: the WN stacks run on IEEE-half storage and fp16 MFMA with fp32 accumulation
          (``set_compute_dtype(torch.float16)``, the reference's own half mode: inside the 1e-3 waveform bound).
        * ax model (what cookietts' own trainer writes: WaveFlow, 1-D ax WaveGlow): every conv-GEMM takes its products on the
          bf16 matrix pipe as hi + lo splits with fp32 accumulation (``set_f32_gemm_mode("bf16x3")``: three bf16 products per
          MAC, tensors stay fp32) - 1.3-2x the fp32 MFMA rate where a launch is arithmetic-bound (the notebook's 1-D WaveGlow:
          124 -> 186x real time at batch 1, 146 -> 291x at 8; WaveFlow config 4 at batch 8: 163 -> 110 ms), <= 7e-6 RMS
          from the fp32 reference.  Half STORAGE is not built for the ax core: its batch-1 launches are latency-bound
          (64-256 channels), where narrower tensors buy nothing.
        """
        if self.is_ax:
            self.waveglow.set_f32_gemm_mode("bf16x3")
        else:
            self.waveglow.set_compute_dtype(torch.float16)
        return self


def waveglow_from_checkpoint(checkpoint: dict):
    """Build the class the checkpoint's ``waveglow_config`` was written for and load its ``model`` state dict."""
    cfg = checkpoint['waveglow_config']
    model = WaveGlowAx(**cfg) if is_ax_config(cfg) else WaveGlow(**cfg)
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
    vocoder = WaveGlowVocoder(model, sigma=sigma, speaker_lookup=checkpoint.get('speaker_lookup'))
    return vocoder, checkpoint['waveglow_config']
