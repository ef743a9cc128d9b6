"""Why BASELINE config 3 (bf16) sits outside the north-star's 1e-3 waveform bound, pinned on the CPU against the
REFERENCE's fp32 goldens ("bring config 3 inside 1e-3 or say why it cannot be").

The numpy restatement of the WN stack is run with the bf16 roundings switched on one class at a time:
  * storage rounding   - x, act, cond hidden, skip sum kept as bf16 tensors in HBM (what the bf16 kernels do);
  * operand rounding   - a tensor / weight is rounded only where it enters a GEMM (fp32 in HBM: "bf16x1", the variant
                         the verdict proposed), optionally on one side only (the other side exact = two bf16 products).
Result (12 x 512 model, `waveglow_full_short`): fp32 storage + bf16 operands is 1.9e-3, the kernels' own rounding
points 2.2e-3 - keeping the residual stream in fp32 buys 14 %, not a factor of two; and even with ONE operand side
exact (two MFMA products per contraction) it is 1.3e-3.  The limiter is the 8-bit mantissa of the single product
(each GEMM output carries ~2^-9 relative error whatever K is, and the flows compound it), not where x is stored: no
single-product bf16 scheme reaches 1e-3 on this model; three products (bf16x3, shipped: 4e-6) do.  So the bf16 path is
gated against the reference at what it measures (tests/test_waveglow_gpu.py BF16_VS_REFERENCE_LIMIT) and reported.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN, rms_rel_err
from cookietts_amd import synthetic
from oracle import waveglow_oracle as wo

F32 = np.float32


def _wn_forward_variant(round_w, round_act, round_store):
    rnd = wo.bf16_round
    rw = rnd if round_w else (lambda a: a)
    ra = rnd if round_act else (lambda a: a)
    rs = rnd if round_store else (lambda a: a)

    def fwd(sd, prefix, audio0, spect, n_layers, n_channels, speaker_ids=None, rnd=None):   # (rnd: the oracle's storage rounding, unused here)
        C = n_channels
        x = rs(wo._conv1x1(wo._conv_weight(sd, prefix + ".start"), sd[prefix + ".start.bias"], audio0))
        h = rs(spect)
        for j in range(2):
            h = rs(wo._conv1x1(rw(wo._conv_weight(sd, f"{prefix}.cond_layers.{j}")), sd[f"{prefix}.cond_layers.{j}.bias"], ra(h)))
        wc2 = rw(wo._conv_weight(sd, f"{prefix}.cond_layers.2")[:, :, 0])
        bc2 = sd[f"{prefix}.cond_layers.2.bias"]
        out = None
        for i in range(n_layers):
            d = 2 ** i
            w = rw(wo._conv_weight(sd, f"{prefix}.in_layers.{i}"))
            bias = (sd[f"{prefix}.in_layers.{i}.bias"] + bc2[2 * C * i:2 * C * (i + 1)]).astype(F32)
            u = np.matmul(np.ascontiguousarray(wc2[2 * C * i:2 * C * (i + 1)]), ra(h))
            xa = ra(x)
            for t in range(3):
                u = u + np.matmul(np.ascontiguousarray(w[:, :, t]), wo._shift(xa, (t - 1) * d))
            u = (u + bias[None, :, None]).astype(F32)
            act = rs(np.tanh(u[:, :C]) * (F32(1.0) / (F32(1.0) + np.exp(-u[:, C:]))))
            r = wo._conv1x1(rw(wo._conv_weight(sd, f"{prefix}.res_skip_layers.{i}")), sd[f"{prefix}.res_skip_layers.{i}.bias"],
                            ra(act)).astype(F32)
            if i < n_layers - 1:
                x, sk = rs(x + r[:, :C]), r[:, C:]
            else:
                sk = r
            out = sk if out is None else (out + sk).astype(F32)
        e = wo._conv1x1(np.asarray(sd[prefix + ".end.weight"], dtype=F32), sd[prefix + ".end.bias"], rs(out))
        hh = e.shape[1] // 2
        return e[:, :hh], e[:, hh:]
    return fwd


def _errors(name, monkeypatch):
    g = np.load(os.path.join(GOLDEN, f"waveglow_{name}.npz"))
    cfg = synthetic.WAVEGLOW_CONFIGS[str(g["config_key"])]
    sd = synthetic.waveglow_state_dict(cfg, seed=int(g["seed"]))
    out = {"kernel": rms_rel_err(wo.waveglow_infer(sd, cfg, g["mel"], g["z_scaled"], bf16=True), g["wave"])}
    for key, flags in (("fp32", (0, 0, 0)), ("x1_fp32_storage", (1, 1, 0)), ("act_only", (0, 1, 0)), ("w_only", (1, 0, 0))):
        monkeypatch.setattr(wo, "wn_forward_bf16", _wn_forward_variant(*flags))
        out[key] = rms_rel_err(wo.waveglow_infer(sd, cfg, g["mel"], g["z_scaled"], bf16=True), g["wave"])
        monkeypatch.undo()
    return out


def test_single_product_bf16_cannot_reach_1e3_on_the_full_model(monkeypatch):
    e = _errors("full_short", monkeypatch)
    print("waveglow_full_short, rms rel err vs the fp32 reference:", {k: f"{v:.3e}" for k, v in e.items()})
    assert e["fp32"] < 1e-5                                   # the variant machinery itself is exact when nothing is rounded
    assert e["x1_fp32_storage"] > 1.5e-3                      # fp32 residual stream, bf16 operands: still outside 1e-3
    assert e["kernel"] < 1.3 * e["x1_fp32_storage"]           # bf16 storage adds < 30 % on top of operand rounding
    assert e["act_only"] > 1e-3 and e["w_only"] > 1e-3        # even with one side exact (two products)
    assert e["kernel"] < 3e-3


def test_error_budget_on_the_small_model(monkeypatch):
    """Config 1's 4 x 256 model: fewer flows, same picture at a lower level (the single product is ~1.4e-3)."""
    e = _errors("small", monkeypatch)
    print("waveglow_small, rms rel err vs the fp32 reference:", {k: f"{v:.3e}" for k, v in e.items()})
    assert e["x1_fp32_storage"] > 1e-3 and e["kernel"] < 1.3 * e["x1_fp32_storage"]
    assert e["act_only"] < e["x1_fp32_storage"] and e["w_only"] < e["x1_fp32_storage"]
