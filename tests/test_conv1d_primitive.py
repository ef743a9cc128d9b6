"""`ctts_conv1d_f32` (the operator the Tacotron encoder / postnet and the WaveFlow conditioning stacks are composed from)
against the numpy restatement, over the shapes that select its different code paths."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import waveflow_oracle as wf

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("c_in,c_out,k,act,slope,T", [
    (32, 24, 1, 0, 0.0, 7),          # 1x1, ragged everything
    (40, 32, 3, 1, 0.0, 37),         # input channels padded to 16 by the host wrapper; ReLU ('lrelu' of the reference)
    (416, 512, 9, 1, 0.25, 131),     # author's model-level stack: 234 K-chunks (DMA-staged kernel)
    (512, 416, 9, 2, 0.0, 130),      # 288 K-chunks: beyond the chunk table -> register-staged kernel; tanh
])
def test_conv1d_matches_numpy(hip_lib_path, c_in, c_out, k, act, slope, T):
    from cookietts_amd import _lib
    from cookietts_amd.waveglow_ax import PAD, _CondConv
    rng = np.random.default_rng(c_in + k)
    w = (rng.standard_normal((c_out, c_in, k)) / np.sqrt(c_in * k)).astype(np.float32)
    b = rng.standard_normal(c_out).astype(np.float32)
    x = rng.standard_normal((2, c_in, T)).astype(np.float32)
    dev = torch.device("cuda", 0)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    op = _CondConv(torch.from_numpy(w).to(dev), torch.from_numpy(b).to(dev), act, slope, dev, stream)
    ld = -(-T // 128) * 128 + 2 * PAD
    xin = torch.zeros(2, op.c_in, ld, device=dev)
    xin[:, :c_in, PAD:PAD + T] = torch.from_numpy(x).to(dev)
    y = torch.zeros(2, -(-c_out // 16) * 16, ld, device=dev)
    op(xin, y, 2, T, ld, stream)
    torch.cuda.synchronize()
    ref = wf.conv1d_same(x, w, b)
    if act == 1:
        ref = np.where(ref >= 0, ref, ref * np.float32(slope))
    elif act == 2:
        ref = np.tanh(ref)
    got = y[:, :c_out, PAD:PAD + T].cpu().numpy()
    assert np.abs(got - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())
    halo = y.clone()
    halo[:, :, PAD:PAD + T] = 0
    assert float(halo.abs().max()) == 0.0                     # halo columns and padding rows stay zero


def test_accumulate_epilogue_stays_inside_a_ragged_destination(hip_lib_path):
    """The postnet's residual convs accumulate into an 80-row tensor (M = 80: the last 32-row tile is ragged).  The
    epilogue used to read the 16 non-existent rows 80..95 of the destination too - 17 KB past the end of the last batch
    item, a GPU memory fault whenever the tensor ends at a mapping boundary (r3: Tacotron2.inference at B=1 after a
    vocoder call).  Here the destination is the very END of its own 32 MiB device allocation, so a read past row 79
    leaves the allocation; the result is also checked."""
    from cookietts_amd import _lib
    from cookietts_amd.tacotron2 import PAD, _HipConv1d, _ld_for
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(80)
    c_in, c_out, k, T, B = 512, 80, 5, 256, 1
    conv = torch.nn.Conv1d(c_in, c_out, k, padding=2)
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy((rng.standard_normal((c_out, c_in, k)) / np.sqrt(c_in * k)).astype(np.float32)))
        conv.bias.copy_(torch.from_numpy(rng.standard_normal(c_out).astype(np.float32)))
    op = _HipConv1d(conv.to(dev), None, 0, 0.0, dev)
    ld = _ld_for(T)
    x = rng.standard_normal((B, c_in, T)).astype(np.float32)
    y0 = rng.standard_normal((B, c_out, T)).astype(np.float32)
    xin = torch.zeros(B, c_in, ld, device=dev)
    xin[:, :, PAD:PAD + T] = torch.from_numpy(x).to(dev)
    torch.cuda.empty_cache()
    big = torch.zeros(32 << 20 >> 2, device=dev)                  # >= 20 MiB: its own hipMalloc of exactly this size
    n = B * c_out * ld
    y = big[-n:].view(B, c_out, ld)                               # ends exactly where the allocation ends
    assert y.data_ptr() + n * 4 == big.data_ptr() + big.numel() * 4
    y[:, :, PAD:PAD + T] = torch.from_numpy(y0).to(dev)
    op(xin, y, True, B, T, ld)
    torch.cuda.synchronize()
    ref = y0 + wf.conv1d_same(x, conv.weight.detach().cpu().numpy(), conv.bias.detach().cpu().numpy())
    assert np.abs(y[:, :, PAD:PAD + T].cpu().numpy() - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())
