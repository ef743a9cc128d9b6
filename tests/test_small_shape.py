"""The small-problem shape of the fp32 conv-GEMM (csrc/gemm_f32_small.hip: 128 x 64 blocks, 64 x 32 wave tiles on the SAME
packed operands) against the 256 x 128 shape: every output element sums the same chunks and k-steps in the same order, so
the two must agree BIT FOR BIT on every epilogue; and the goldens still hold when the large shape is forced at toy size
(the toy goldens otherwise all run through the small shape now).  Knobs: CTTS_F32_NO_SMALL / CTTS_F32_FORCE_SMALL."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rms_rel_err
from cookietts_amd import synthetic

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("c_in,c_out,k,act,slope,T,acc", [
    (32, 24, 1, 0, 0.0, 7, False),        # SPLIT, ragged rows and columns, one chunk
    (48, 80, 5, 0, 0.0, 200, True),       # SPLIT with accumulate into a ragged M (the postnet's residual convs)
    (512, 512, 5, 2, 0.0, 131, False),    # TANH, 160 chunks, five interleaved segments
    (416, 512, 9, 1, 0.25, 90, False),    # LRELU, nine segments (segment table path), 234 chunks
])
def test_conv1d_small_shape_is_bit_identical(hip_lib_path, tuning, c_in, c_out, k, act, slope, T, acc):
    from cookietts_amd.waveglow_ax import PAD, _CondConv
    from cookietts_amd import _lib
    rng = np.random.default_rng(c_in * 7 + k)
    w = (rng.standard_normal((c_out, c_in, k)) / np.sqrt(c_in * k)).astype(np.float32)
    b = rng.standard_normal(c_out).astype(np.float32)
    dev = torch.device("cuda", 0)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    op = _CondConv(torch.from_numpy(w).to(dev), torch.from_numpy(b).to(dev), act, slope, dev, stream)
    B = 2
    ld = -(-T // 128) * 128 + 2 * PAD
    xin = torch.zeros(B, op.c_in, ld, device=dev)
    xin[:, :c_in, PAD:PAD + T] = torch.from_numpy(rng.standard_normal((B, c_in, T)).astype(np.float32)).to(dev)
    y0 = torch.zeros(B, -(-c_out // 16) * 16, ld, device=dev)
    if acc:
        y0[:, :c_out, PAD:PAD + T] = torch.from_numpy(rng.standard_normal((B, c_out, T)).astype(np.float32)).to(dev)

    def run():
        y = y0.clone()
        if acc:
            _lib.check(_lib.lib().ctts_conv1d_f32(C.byref(op.desc), _lib.ptr(op.blob), _lib.ptr(xin), _lib.ptr(y), 1, B, T, ld, PAD,
                                                  stream), "ctts_conv1d_f32")
        else:
            op(xin, y, B, T, ld, stream)
        torch.cuda.synchronize()
        return y
    small = run()                           # a handful of 256 x 128 blocks: the small shape is chosen
    tuning.set("CTTS_F32_NO_SMALL")
    big = run()
    assert torch.equal(small, big)
    assert float(small[:, :c_out, PAD:PAD + T].abs().max()) > 0


@pytest.mark.parametrize("name", ["waveglow_ax_notebook_toy", "waveglow_ax_toy_gate_glu", "waveglow_ax_toy_merge",
                                  "waveglow_ax_toy_c96", "waveglow_ax_toy_c160"])
def test_ax_core_small_shape_is_bit_identical_and_large_shape_still_meets_the_golden(hip_lib_path, tuning, name):
    """GATE with the interpolated conditioning addend, GATEX (GLU), merged res/skip: small shape (default at this size) ==
    large shape (forced), both against the reference golden."""
    from cookietts_amd.waveglow_ax import WaveGlow
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    cfg = synthetic.WAVEGLOW_AX_CONFIGS[str(g["config_key"])]
    m = WaveGlow(**cfg)
    m.load_state_dict(synthetic.to_torch(synthetic.waveglow_ax_state_dict(cfg, seed=int(g["seed"]))))
    m = m.cuda().eval()
    z, mel = torch.from_numpy(g["z"]).cuda(), torch.from_numpy(np.pad(g["mel"], ((0, 0), (0, 0), (0, 1)))).cuda()
    kw = {"speaker_ids": torch.from_numpy(g["speaker_ids"]).cuda()} if "speaker_ids" in g.files else {}
    small, _ = m.inverse(z, mel, return_CPU=False, **kw)
    tuning.set("CTTS_F32_NO_SMALL")
    big, _ = m.inverse(z, mel, return_CPU=False, **kw)
    assert torch.equal(small, big)
    err = rms_rel_err(big.cpu().numpy(), g["inverse_full"])
    print(f"{name}: large shape forced, rms rel err vs reference = {err:.3e}")
    assert err < 1e-3


def test_waveglow_and_stft_small_shape_is_bit_identical(hip_lib_path, tuning):
    """glow.py core (GATE with a plain addend-free K, SPLIT read-modify-write) on a 12 x 512 model at a width where the
    large shape is the default, against the small shape forced; and the STFT's MAG / LOG epilogues."""
    from cookietts_amd import TacotronSTFT, WaveGlow
    cfg = synthetic.WAVEGLOW_CONFIGS["full"]
    m = WaveGlow(**cfg)
    m.load_state_dict(synthetic.to_torch(synthetic.waveglow_state_dict(cfg, seed=5)))
    m = m.cuda().eval()
    B, F = 3, 101                                           # 4 x 26 x 3 = 312 blocks: the default IS the small shape here
    mel = torch.from_numpy(synthetic.synthetic_mel(B, F, seed=6)).cuda()
    zz = torch.from_numpy(synthetic.synthetic_noise(B, 8, F * 32, seed=6) * np.float32(0.6)).cuda()
    small = m.infer_from_noise(mel, zz)
    tuning.set("CTTS_F32_NO_SMALL")
    big = m.infer_from_noise(mel, zz)
    assert torch.isfinite(big).all() and torch.equal(small, big)
    tuning.clear("CTTS_F32_NO_SMALL")
    B, F = 8, 300                                           # 4 x 75 x 8 = 2400 blocks: the large shape by default
    mel = torch.from_numpy(synthetic.synthetic_mel(B, F, seed=7)).cuda()
    zz = torch.from_numpy(synthetic.synthetic_noise(B, 8, F * 32, seed=7) * np.float32(0.6)).cuda()
    big = m.infer_from_noise(mel, zz)
    tuning.set("CTTS_F32_FORCE_SMALL")
    assert torch.equal(big, m.infer_from_noise(mel, zz))
    tuning.clear("CTTS_F32_FORCE_SMALL")
    stft = TacotronSTFT(1024, 256, 1024, 80, 22050, 0.0, 8000.0).cuda()
    audio = torch.from_numpy(np.random.default_rng(3).uniform(-0.7, 0.7, (2, 20000)).astype(np.float32)).cuda()
    mel_small = stft.mel_spectrogram(audio)
    tuning.set("CTTS_F32_NO_SMALL")
    assert torch.equal(mel_small, stft.mel_spectrogram(audio))


@pytest.mark.parametrize("name", ["toy", "full_short", "toy_dilations_h"])
def test_waveflow_fused_layer_small_shapes(hip_lib_path, tuning, name):
    """The fused WaveFlow layer (GATE_RS: dilated 2-D conv GEMM + gate + res/skip GEMM in one launch) in its three shapes:
    the 128 x 128 shape (128 x 32 wave tiles) is BIT-IDENTICAL to the 128 x 256 one; the split-K shape (128 x 64 blocks,
    the K halves on wave pairs: the default at this size) sums (even chunks) + (odd chunks) and agrees to fp32 summation
    noise - its eight-wave form (the default of the per-layer launches) and its four-wave form (the row queue's items) are
    bit-identical; every shape meets the reference golden."""
    from cookietts_amd import WaveFlow
    g = np.load(os.path.join(GOLDEN, f"waveflow_{name}.npz"))
    cfg = synthetic.WAVEFLOW_CONFIGS[str(g["config_key"])]
    m = WaveFlow(**cfg)
    m.load_state_dict(synthetic.to_torch(synthetic.waveflow_state_dict(cfg, seed=int(g["seed"]))))
    m = m.cuda().eval()
    z, mel = torch.from_numpy(g["z"]).cuda(), torch.from_numpy(np.pad(g["mel"], ((0, 0), (0, 0), (0, 1)))).cuda()
    splitk, _ = m.inverse(z, mel, return_CPU=False)
    tuning.set("CTTS_F32_SPLITK_W4")                        # four waves per split-K tile (until round 5) instead of eight: same sums
    assert torch.equal(splitk, m.inverse(z, mel, return_CPU=False)[0])
    tuning.clear("CTTS_F32_SPLITK_W4")
    tuning.set("CTTS_F32_NO_SPLITK")
    small, _ = m.inverse(z, mel, return_CPU=False)
    tuning.set("CTTS_F32_NO_SMALL")
    big, _ = m.inverse(z, mel, return_CPU=False)
    assert torch.equal(small, big)
    assert not torch.equal(splitk, big)                      # the split-K shape really ran
    d = rms_rel_err(splitk.cpu().numpy(), big.cpu().numpy())
    err, err_sk = rms_rel_err(big.cpu().numpy(), g["inverse_full"]), rms_rel_err(splitk.cpu().numpy(), g["inverse_full"])
    print(f"waveflow {name}: rms rel err vs reference: large shape {err:.3e}, split-K shape {err_sk:.3e}; split-K vs large {d:.3e}")
    assert err < 1e-3 and err_sk < 1e-3 and d < 5e-6


@pytest.mark.parametrize("name", ["table_g50_c128", "table_g20_c512", "table_g12_c256_sep"])
def test_small_shape_on_the_128_row_packing_is_bit_identical(hip_lib_path, tuning, name):
    """WaveFlow above 64 channels (the reference's published sweep: 128-512 channels at batch 1) runs its in-layer and res/skip
    GEMMs unfused on the 128-row packing.  Since round 5 they take the 128 x 64 small shape too (their 128 x 256 launches were
    44-176 blocks on 256 CUs): same chunk order, same MFMAs - bit-identical to the large shape, and inside the golden's bound."""
    from cookietts_amd import WaveFlow, _lib
    g = np.load(os.path.join(GOLDEN, f"waveflow_{name}.npz"))
    cfg = synthetic.WAVEFLOW_CONFIGS[str(g["config_key"])]
    m = WaveFlow(**cfg)
    m.load_state_dict(synthetic.to_torch(synthetic.waveflow_state_dict(cfg, seed=int(g["seed"]))))
    m = m.cuda().eval()
    z, mel = torch.from_numpy(g["z"]).cuda(), torch.from_numpy(np.pad(g["mel"], ((0, 0), (0, 0), (0, 1)))).cuda()
    small, _ = m.inverse(z, mel, return_CPU=False)
    assert _lib.lib().ctts_last_gemm_loop() & 16, "the small shape did not run"
    tuning.set("CTTS_F32_NO_SMALL")
    big, _ = m.inverse(z, mel, return_CPU=False)
    assert not (_lib.lib().ctts_last_gemm_loop() & 16)
    assert torch.equal(small, big)
    err = rms_rel_err(small.cpu().numpy(), g["inverse_full"])
    print(f"waveflow {name}: small shape on the 128-row packing, rms rel err vs reference = {err:.3e}")
    assert err < 1e-3


def test_small_shape_in_split_bf16_mode_is_bit_identical(hip_lib_path, tuning):
    """The split-bf16 main loop (model.set_f32_gemm_mode('bf16x3')) in the small shape: same three products in the same
    order per chunk as the large shape, so bit-identical there too; and inside 1e-4 of the reference golden."""
    from cookietts_amd.waveglow_ax import WaveGlow
    g = np.load(os.path.join(GOLDEN, "waveglow_ax_notebook_toy.npz"))
    cfg = synthetic.WAVEGLOW_AX_CONFIGS[str(g["config_key"])]
    m = WaveGlow(**cfg)
    m.load_state_dict(synthetic.to_torch(synthetic.waveglow_ax_state_dict(cfg, seed=int(g["seed"]))))
    m = m.cuda().eval().set_f32_gemm_mode("bf16x3")
    z, mel = torch.from_numpy(g["z"]).cuda(), torch.from_numpy(np.pad(g["mel"], ((0, 0), (0, 0), (0, 1)))).cuda()
    ids = torch.from_numpy(g["speaker_ids"]).cuda()
    small, _ = m.inverse(z, mel, speaker_ids=ids, return_CPU=False)
    tuning.set("CTTS_F32_NO_SMALL")
    big, _ = m.inverse(z, mel, speaker_ids=ids, return_CPU=False)
    assert torch.equal(small, big)
    err = rms_rel_err(small.cpu().numpy(), g["inverse_full"])
    print(f"waveglow_ax notebook_toy, split-bf16 loop, small shape: rms rel err vs reference = {err:.3e}")
    assert err < 1e-4
    m.set_f32_gemm_mode("f32")
    tuning.clear("CTTS_F32_NO_SMALL")
    exact, _ = m.inverse(z, mel, speaker_ids=ids, return_CPU=False)
    assert not torch.equal(exact, small)                    # the split loop really ran
