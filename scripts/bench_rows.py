#!/usr/bin/env python3
"""Secondary measurements for SURVEY.md 8 rows B (WaveFlow, config 4), C (Tacotron2 decoder, config 5)
and D (STFT/mel).  bench.py stays the headline (config 2): it imports the row functions of this file and attaches
short runs of configs 3 / 4 / 5 to its JSON line as ``rows`` (after, never inside, the headline's timed region);
run as a script this prints one JSON line per row."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cookietts_amd import synthetic  # noqa: E402


def _batches(args, default):
    return tuple(int(x) for x in args.batches.split(",")) if getattr(args, "batches", "") else default


def timed(fn, warmup, steps):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


FP32_MFMA_PEAK_TFLOPS = 157.3


def gemm_loop_label():
    """What the calling thread's last conv-GEMM launch really ran (ctts_last_gemm_loop): the fused WaveFlow layer runs
    fp32 MFMA in its split-K shape (batch <= 2) whatever mode the model asked for."""
    from cookietts_amd import _lib
    code = _lib.lib().ctts_last_gemm_loop()
    loop = {0: "fp32 MFMA", 3: "split-bf16 x3", 6: "split-bf16 x6"}.get(code & 15, f"level {code & 15}")
    if code & 64:
        shape = ("row queue (one launch per flow), " if code & 128 else "row queue (one launch per row), ") + \
            ("split-K items of 128 x 64" if code & 32 else "items of 128 x 128")
    else:
        shape = "split-K shape" if code & 32 else "small shape" if code & 16 else "large shape"
    return f"{loop}, {shape}"


def _mode(m, args):
    """The rows' GEMM main loop travels in each model's config (``--gemm-mode``; there is no process-wide default)."""
    mode = getattr(args, "gemm_mode", "f32")
    if mode != "f32" and hasattr(m, "set_f32_gemm_mode"):
        m.set_f32_gemm_mode(mode)
    return m


def row_waveflow(args):
    from cookietts_amd.waveglow_ax import WaveGlow
    cfg = synthetic.WAVEFLOW_CONFIGS["full"]
    m = WaveGlow(**cfg)
    m.load_state_dict(synthetic.to_torch(synthetic.waveflow_state_dict(cfg, seed=1234)))
    m = m.cuda().eval()
    _mode(m, args)
    rows = []
    for B in _batches(args, (1, 8)):     # B = 1 too: the reference's own WaveFlow timing table is batch 1
        F = 900
        mel = torch.from_numpy(synthetic.synthetic_mel(B, F)).cuda()
        dt = timed(lambda: m.infer(mel, sigma=0.6, return_CPU=False), args.warmup, args.steps)
        loop = gemm_loop_label()
        samples = B * (F - 1) * 256
        wn = cfg["WN_config"]
        C, G = wn["n_channels"], cfg["n_group"]
        mac = 0.6515e6 * (G - 1) * cfg["n_flows"] / G          # per output sample
        rows.append({"row": "B/config4", "metric": "audio samples/sec (22.05kHz) WaveFlow infer (8 flows, 64 ch, h=16), 80x900 mel",
                     "value": samples / dt, "unit": "samples/s", "rtf": samples / dt / 22050, "ms_per_call": dt * 1e3,
                     "dtype": "f32", "batch": B, "frames": F,
                     # per row of the recurrence: start + tail + (row queue: ONE launch for the n_layers fused layers | one per layer)
                     "kernel_launches_per_utterance_batch": (2 * cfg["n_flows"] if "per flow" in loop else
                                                             ((1 if "row queue" in loop else wn["n_layers"]) + 2) * cfg["n_flows"] * (G - 1)),
                     # common schema (the headline's): the fused WaveFlow layer kernels against the fp32 MFMA peak on SURVEY 8d's
                     # algorithmic MACs per output sample; the survey's 138 KB-per-sample HBM denominator stays below as extra keys
                     "roofline": {"kernel": "fused WaveFlow layer (" + loop + ")", "bound": "mfma",
                                  "achieved": 2 * mac * samples / dt / 1e12, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": 2 * mac * samples / dt / 1e12 / FP32_MFMA_PEAK_TFLOPS, "traffic": None},
                     "last_gemm_loop": loop,
                     # 138 KB of per-layer-kernel traffic per output sample (2304 B per row, step, layer)
                     "achieved_GBps_vs_138KB_per_sample": 138e3 * samples / dt / 1e9,
                     "hbm_frac_vs_138KB_per_sample": 138e3 * samples / dt / 8e12})
    return rows


# "WaveFlow Inference Times.png" (CookieTTS/_4_mtw/): the reference's only published numbers for this path.
# Columns: n_group, n_flows, n_channels, separable, published 22 kHz real-time factor (batch 1, 8 layers, hardware not named).
PUBLISHED_WAVEFLOW_TABLE = [
    (8, 6, 64, 0, 17.002), (8, 6, 64, 1, 16.432), (8, 8, 64, 0, 12.448), (8, 8, 64, 1, 12.251), (8, 8, 128, 1, 9.316),
    (20, 4, 128, 1, 8.807), (20, 6, 64, 0, 6.200), (20, 6, 64, 1, 6.011), (20, 6, 128, 1, 6.003), (8, 8, 128, 0, 5.376),
    (20, 6, 256, 1, 4.956), (20, 8, 64, 0, 4.911), (20, 8, 64, 1, 4.612), (8, 8, 256, 1, 4.488), (20, 8, 128, 1, 4.450),
    (12, 8, 256, 1, 4.218), (20, 8, 128, 0, 4.070), (20, 8, 256, 1, 3.778), (20, 10, 128, 1, 3.674), (20, 12, 128, 1, 3.076),
    (20, 6, 256, 0, 1.953), (50, 8, 128, 0, 1.929), (50, 8, 256, 1, 1.747), (20, 8, 512, 1, 1.630), (20, 8, 256, 0, 1.459),
    (50, 8, 256, 0, 1.131), (20, 8, 512, 0, 0.375), (50, 8, 512, 0, 0.336)]


def row_waveflow_table(args):
    """Every architecture of the reference's published WaveFlow sweep at batch 1 (the only batch it published), ~10 s of audio:
    3x3 kernels, 8 layers, permuteheight mixing (config 4's family), hop 256 for n_group 8 and 300 otherwise
    (hop % n_group == 0, efficient_model_ax.py:23); random-init weights.  The published factor is printed beside ours for
    context only: its hardware is not named."""
    import gc
    from cookietts_amd.waveglow_ax import WaveGlow
    rows = []
    for G, n_flows, C, sep, published in PUBLISHED_WAVEFLOW_TABLE:
        hop = 256 if G == 8 else 300
        cfg = synthetic.waveflow_config(n_flows=n_flows, n_group=G, n_channels=C, hop_length=hop, win_length=4 * hop,
                                        WN=dict(seperable_conv=bool(sep)))
        m = WaveGlow(**cfg)
        m.load_state_dict(synthetic.to_torch(synthetic.waveflow_state_dict(cfg, seed=1234)))
        m = m.cuda().eval()
        _mode(m, args)
        F = 220500 // hop
        mel = torch.from_numpy(synthetic.synthetic_mel(1, F)).cuda()
        dt = timed(lambda: m.infer(mel, sigma=0.6, return_CPU=False), args.warmup, args.steps)
        samples = (F - 1) * hop
        rows.append({"row": "B/waveflow_table", "metric": "real-time factor at 22.05 kHz, WaveFlow infer, batch 1",
                     "n_group": G, "n_flows": n_flows, "n_layers": 8, "n_channels": C, "seperable_conv": sep, "hop_length": hop,
                     "frames": F, "value": samples / dt / 22050, "unit": "x real time", "ms_per_call": dt * 1e3,
                     "published_rtf_22khz": published, "vs_published": samples / dt / 22050 / published,
                     "published_source": "CookieTTS/_4_mtw/WaveFlow Inference Times.png (hardware not named)",
                     "last_gemm_loop": gemm_loop_label(), "dtype": "f32"})
        del m
        gc.collect()
        torch.cuda.empty_cache()
    return rows


def row_waveflow_author(args):
    """the option set / sizes of the author's own WaveFlow checkpoints (48 kHz, hop 600, n_group 20,
    128 channels, separable 7x7 in-layers, speaker embeddings, 5 + 3 layer conditioning stacks, de-emphasis)."""
    from cookietts_amd.waveglow_ax import WaveGlow
    cfg = synthetic.WAVEFLOW_CONFIGS["author"]
    m = WaveGlow(**cfg)
    m.load_state_dict(synthetic.to_torch(synthetic.waveflow_state_dict(cfg, seed=1234)))
    m = m.cuda().eval()
    _mode(m, args)
    B, F = 8, 400
    mel = torch.from_numpy(synthetic.synthetic_mel(B, F, cfg["n_mel_channels"] * 2)).cuda()
    ids = torch.arange(B).cuda()
    dt = timed(lambda: m.infer(mel, speaker_ids=ids, sigma=0.6, return_CPU=False), args.warmup, args.steps)
    samples = B * (F - 1) * cfg["hop_length"]
    return {"row": "B/8f.4", "metric": "audio samples/sec (48kHz) WaveFlow infer, author's option set (8 flows, 128 ch, "
                                       "h=20, separable 7x7, speaker + cond stacks), 320x400 mel",
            "value": samples / dt, "unit": "samples/s", "rtf": samples / dt / cfg["sampling_rate"],
            "ms_per_call": dt * 1e3, "dtype": "f32", "batch": B, "frames": F}


def row_waveglow_ax_notebook(args):
    """the ONLY WaveGlow timing recorded in the reference tree - scripts/"WaveGlowFlow Inference
    Speed Testing.ipynb" cells 2-6: ax core, waveflow=False, 48 flows, n_group 24, 8 x 256 WN, 'permute' mixing,
    speaker embeddings, 3-layer cond stack, batch 1, one 5.8375 s clip at 48 kHz (hop 600 -> 468 mel frames):
    1.27 s = 4.60x real time (eager, fp16), 1.125 s = 5.19x (jit-traced), GPU not stated.  Same model shape, same
    clip length, batch 1 here; fp32 (the reference ran .half())."""
    from cookietts_amd.waveglow_ax import WaveGlow
    cfg = synthetic.WAVEGLOW_AX_CONFIGS["notebook"]
    m = WaveGlow(**cfg)
    m.load_state_dict(synthetic.to_torch(synthetic.waveglow_ax_state_dict(cfg, seed=1234)))
    m = m.cuda().eval()
    _mode(m, args)
    rows = []
    for B in _batches(args, (1, 8)):
        F = 468                                                   # -> (F - 1) * 600 = 280 200 samples = 5.8375 s
        mel = torch.from_numpy(synthetic.synthetic_mel(B, F, cfg["n_mel_channels"])).cuda()
        ids = torch.zeros(B, dtype=torch.int64).cuda()
        dt = timed(lambda: m.infer(mel, speaker_ids=ids, sigma=1.0, return_CPU=False), args.warmup, args.steps)
        samples = B * (F - 1) * cfg["hop_length"]                 # infer() pads one frame and trims one hop
        wn = cfg["WN_config"]
        C, nl = wn["n_channels"], wn["n_layers"]
        flop = 2.0 * cfg["n_flows"] * nl * (3 * C * 2 * C + 2 * C * C) * (samples / cfg["n_group"])   # in + res/skip GEMMs
        rows.append({"row": "W5/notebook", "metric": "real-time factor (48 kHz), ax WaveGlow waveflow=False, 48 flows x 8 x 256, "
                                                     "n_group 24, 160x468 mel (5.84 s clip)",
                     "value": samples / dt / 48000.0, "unit": "x real time (48 kHz)", "batch": B, "ms_per_call": dt * 1e3,
                     "samples_per_s": samples / dt, "rtf_22k_equiv": samples / dt / 48000.0 * 48 / 22, "dtype": "f32",
                     "reference_published": {"eager_fp16_rtf_48k": 4.5977, "jit_fp16_rtf_48k": 5.1905, "batch": 1,
                                             "hardware": "not stated", "source": "BASELINE.md section 1"},
                     "vs_reference_eager": (samples / dt / 48000.0) / 4.5977 if B == 1 else None,
                     "roofline": {"kernel": "ax 1-D WN in-layer + res/skip conv-GEMMs", "bound": "mfma", "achieved": flop / dt / 1e12,
                                  "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": flop / dt / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                                  "traffic": None}})
    return rows


def row_tacotron(args, vocoder=None):
    """``vocoder``: a WaveGlow (config 2 weights) already on the GPU, or None to build one."""
    from cookietts_amd.tacotron2 import Tacotron2
    hp = synthetic.tacotron_hparams()
    m = Tacotron2(hp)
    m.load_state_dict(synthetic.to_torch(synthetic.tacotron_state_dict(hp, seed=1234)))
    m = m.cuda().eval()
    B, T, steps = 4, 200, 900
    rng = np.random.default_rng(1234)
    text = torch.from_numpy(rng.integers(1, 179, size=(B, T))).cuda()
    lens = torch.tensor([200, 195, 150, 100]).cuda()
    spk = torch.arange(B).cuda()
    tm = torch.from_numpy(rng.standard_normal((B, 2304)).astype(np.float32)).cuda()
    dt = timed(lambda: m.inference(text, lens, spk, tm, fixed_steps=steps), args.warmup, args.steps)
    mem = torch.from_numpy((rng.standard_normal((B, T, 1313)) * 0.5).astype(np.float32)).cuda()
    dd = timed(lambda: m.decoder.inference(mem, lens, fixed_steps=steps), 1, args.steps)
    weights_mb = sum(p.numel() for n, p in m.decoder.named_parameters()
                     if "rnn" in n or "projection" in n or "gate" in n or "query" in n or "prenet" in n) * 4 / 1e6
    # chained vocoder: WaveGlow config 2 weights on the B=4 x 900-frame mel the model just produced
    wg = vocoder
    if wg is None:
        from cookietts_amd import WaveGlow
        wcfg = synthetic.WAVEGLOW_CONFIGS["full"]
        wg = WaveGlow(**wcfg)
        wg.load_state_dict(synthetic.to_torch(synthetic.waveglow_state_dict(wcfg, seed=1234)))
        wg = wg.cuda().eval()
    mel = m.inference(text, lens, spk, tm, fixed_steps=steps)["pred_mel_postnet"].clamp(-11.52, 2.0)
    dv = timed(lambda: wg.infer(mel, sigma=0.6), 1, max(1, args.steps - 1))
    # the same vocoder in the reference's half mode (WaveGlowVocoder.half(): IEEE-half storage + fp16 MFMA from the fp32 masters,
    # inside the 1e-3 waveform bound) - what the _5_infer slot runs after load_hifigan-style .half()
    dv16 = None
    try:
        wg.set_compute_dtype(torch.float16)
        dv16 = timed(lambda: wg.infer(mel, sigma=0.6), 1, max(1, args.steps - 1))
    finally:
        wg.set_compute_dtype(torch.float32)
    samples = B * steps * 256
    # the batch sizes the reference's server decodes per call (text2speech.py:418-424, 537, 554): one batched-form workspace
    server = []
    for Bs in (16, 64, 256):
        mem_s = torch.from_numpy((rng.standard_normal((Bs, T, 1313)) * 0.5).astype(np.float32)).cuda()
        lens_s = torch.full((Bs,), T, dtype=torch.int64).cuda()
        ds = timed(lambda: m.decoder.inference(mem_s, lens_s, fixed_steps=256), 1, max(1, args.steps))
        us = ds / 256 * 1e6
        server.append({"batch": Bs, "us_per_step": us, "mel_frames_per_s": Bs * 256 / ds, "workspaces": len(next(iter(m.decoder._ws.values()))),
                       # a step = one pass over the decoder's weights, whatever the batch; the batched form streams them
                       "roofline": {"kernel": "bg_kernel<CELL> x 3 + query / projection / prenet GEMMs + attention (7 launches per step)",
                                    "bound": "hbm", "achieved": weights_mb / 1e3 / (us * 1e-6), "peak": 8000.0, "unit": "GB/s",
                                    "frac": weights_mb / 1e3 / (us * 1e-6) / 8000.0, "traffic": None}})
        del mem_s
    m.decoder._ws, m.decoder._xchg = {}, {}
    return {"row": "C/config5", "metric": "Tacotron2-TM decoder step time, B=4, 200 symbols, 900 forced steps",
            "value": dd / steps * 1e6, "unit": "us/step", "higher_is_better": False,
            "server_batches": server,
            # common schema (the headline's): SURVEY 8d's algorithmic bytes per step = one pass over the weights.  The persistent
            # form keeps them on the CUs (0.78 MB fetched per step), so `achieved` is the rate a streaming step of this length would need
            "roofline": {"kernel": "taco_persistent_kernel", "bound": "hbm", "achieved": weights_mb / 1e3 / (dd / steps), "peak": 8000.0,
                         "unit": "GB/s", "frac": weights_mb / 1e3 / (dd / steps) / 8000.0, "traffic": 0.78e6},
            "mel_frames_per_s_batch": B * steps / dd, "end_to_end_ms_incl_encoder_postnet": dt * 1e3,
            "dtype": "f32", "decoder_form": m.decoder.persistent_state,
            # SURVEY 8d prices a step as one pass over the decoder's weights.  Since round 4 the persistent decoder keeps them in
            # registers / LDS for the whole launch (PMC: 0.78 MB fetched per step, profiles/r4_09_pmc_*): the figures below say what
            # a STREAMING step would cost, not what this kernel moves - the step is bound by its seven all-gathers
            "decoder_weights_MB": weights_mb,
            "streaming_floor_us_per_step_at_8TBps": weights_mb / 1e3 / 8000.0 * 1e6,
            "fetched_per_step_MB_pmc": 0.78 if m.decoder.persistent_state == "ok" else None,
            "fetched_per_step_source": "profiles/r4_09_pmc_config5_tacotron_resident.json (committed PMC pass, not re-measured in this run)",
            "chained_vocoder_samples_per_s": samples / dv, "chained_vocoder_ms": dv * 1e3,
            "text_to_wave_rtf": samples / (dt + dv) / 22050.0,
            "chained_vocoder_f16_ms": None if dv16 is None else dv16 * 1e3,
            "text_to_wave_rtf_f16_vocoder": None if dv16 is None else samples / (dt + dv16) / 22050.0}


def row_stft(args):
    from cookietts_amd import TacotronSTFT
    taco = TacotronSTFT().cuda()
    B, T = 8, 230400
    y = (torch.rand(B, T, device="cuda") * 2 - 1) * 0.5
    dt = timed(lambda: taco.mel_spectrogram(y), args.warmup, args.steps)
    frames = T // 256 + 1
    flop = 2.0 * B * frames * (1026 * 1024 + 80 * 513)
    return {"row": "D/stft", "metric": "TacotronSTFT.mel_spectrogram, 8 x 230400 samples (1024/256/1024, 80 mel)",
            "value": B * T / dt, "unit": "samples/s", "ms_per_call": dt * 1e3, "dtype": "f32",
            "achieved_tflops_algorithmic": flop / dt / 1e12}


def row_waveglow_ax_untts(args):
    """The vocoder config printed by the reference's _2_ttm/untts/inference.ipynb: ax core, waveflow=False, 24 flows x
    8 x 384, n_group 24, 256-channel mel, conditioning upsampled at model level by a TransposedUpsampleNet (2*3*5) and
    handed to the WNs at sample rate.  Same 5.8375 s clip as the notebook row (468 frames, hop 600, 48 kHz)."""
    from cookietts_amd.waveglow_ax import WaveGlow
    cfg = synthetic.WAVEGLOW_AX_CONFIGS["untts"]
    m = WaveGlow(**cfg)
    m.load_state_dict(synthetic.to_torch(synthetic.waveglow_ax_state_dict(cfg, seed=1234)))
    m = m.cuda().eval()
    _mode(m, args)
    rows = []
    for B in _batches(args, (1, 4)):
        F = 468
        mel = torch.from_numpy(synthetic.synthetic_mel(B, F, cfg["n_mel_channels"])).cuda()
        ids = torch.zeros(B, dtype=torch.int64).cuda()
        dt = timed(lambda: m.infer(mel, speaker_ids=ids, sigma=1.0, return_CPU=False), args.warmup, args.steps)
        samples = B * (F - 1) * cfg["hop_length"]
        rows.append({"row": "W5/untts", "metric": "real-time factor (48 kHz), ax WaveGlow waveflow=False, 24 flows x 8 x 384, "
                                                  "upsample_first + TransposedUpsampleNet, 256x468 mel (5.84 s clip)",
                     "value": samples / dt / 48000.0, "unit": "x real time (48 kHz)", "batch": B, "ms_per_call": dt * 1e3,
                     "samples_per_s": samples / dt, "dtype": "f32"})
    return rows


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", default="waveflow,tacotron,stft")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--gemm-mode", default="f32", choices=["f32", "bf16x3", "bf16x6"],
                    help="main loop of the rows' fp32 conv-GEMMs, set on each model (model.set_f32_gemm_mode)")
    ap.add_argument("--batches", default="", help="comma list: restrict the multi-batch rows (waveglow_ax, waveglow_ax_untts) to these batch sizes (PMC passes)")
    args = ap.parse_args()
    fns = {"waveflow": row_waveflow, "waveflow_table": row_waveflow_table, "waveflow_author": row_waveflow_author, "tacotron": row_tacotron, "stft": row_stft,
           "waveglow_ax": row_waveglow_ax_notebook, "waveglow_ax_untts": row_waveglow_ax_untts}
    for r in args.rows.split(","):
        out = fns[r](args)
        for line in (out if isinstance(out, list) else [out]):
            line["f32_gemm_mode"] = args.gemm_mode    # set on every model of the row (model.set_f32_gemm_mode)
            print(json.dumps(line), flush=True)
