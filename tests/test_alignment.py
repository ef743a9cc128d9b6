"""Alignment scoring of the T2S retry loop: oracle vs reference goldens (CPU), HIP vs both (GPU)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import alignment_oracle as ao

KEYS = ("diagonalitys", "avg_prob", "encoder_max_focus", "encoder_min_focus", "encoder_avg_focus", "p_missing_enc")
TOL = 2e-6      # fp32 sums over <= 900 steps in a different (fixed) order than torch's


def _g():
    return np.load(os.path.join(GOLDEN, "alignment.npz"))


@pytest.mark.parametrize("tag", ["small", "wide"])
@pytest.mark.parametrize("lens", [True, False])
def test_oracle_matches_reference(tag, lens):
    g = _g()
    r = ao.alignment_metric(g[f"{tag}_alignments"], g[f"{tag}_in_len"] if lens else None,
                            g[f"{tag}_out_len"] if lens else None)
    for k in KEYS:
        ref = g[f"{tag}_{'lens' if lens else 'nolens'}_{k}"]
        assert r[k].dtype == ref.dtype and r[k].shape == ref.shape
        np.testing.assert_allclose(r[k], ref, rtol=TOL, atol=TOL)


def test_oracle_first_over_thresh_matches_reference():
    g = _g()
    got = ao.get_first_over_thresh(g["gate"], float(g["gate_threshold"]))
    assert got.dtype == np.int32 and (got == g["gate_first"]).all()
    assert list(got) == [17, 40, 0, 9, 40]      # first crossing, never, immediately, == threshold, never


def test_product_path_has_no_cpu_fallback():
    from cookietts_amd import alignment_metric, get_first_over_thresh
    from cookietts_amd._lib import HipLibraryError
    with pytest.raises(HipLibraryError):
        alignment_metric(torch.rand(2, 5, 4))
    with pytest.raises(HipLibraryError):
        get_first_over_thresh(torch.rand(2, 5), 0.5)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["small", "wide"])
@pytest.mark.parametrize("lens", [True, False])
def test_hip_matches_reference_golden(hip_lib_path, tag, lens):
    from cookietts_amd import alignment_metric
    g = _g()
    al = torch.from_numpy(g[f"{tag}_alignments"]).cuda()
    keep = al.clone()
    r = alignment_metric(al, torch.from_numpy(g[f"{tag}_in_len"]).cuda() if lens else None,
                         torch.from_numpy(g[f"{tag}_out_len"]).cuda() if lens else None)
    assert torch.equal(al, keep)                               # input untouched
    for k in KEYS:
        ref = g[f"{tag}_{'lens' if lens else 'nolens'}_{k}"]
        got = r[k].cpu().numpy()
        assert got.dtype == ref.dtype and got.shape == ref.shape
        np.testing.assert_allclose(got, ref, rtol=TOL, atol=TOL)
    avg = alignment_metric(al, average_across_batch=True)
    assert all(v.dim() == 0 for v in avg.values())


@pytest.mark.gpu
def test_hip_first_over_thresh(hip_lib_path):
    from cookietts_amd import get_first_over_thresh
    g = _g()
    got = get_first_over_thresh(torch.from_numpy(g["gate"]).cuda(), float(g["gate_threshold"]))
    assert got.dtype == torch.int32 and got.is_cuda
    assert (got.cpu().numpy() == g["gate_first"]).all()
    # decoder-sized rows, ragged vs the 256-thread stride, against the oracle
    rng = np.random.default_rng(3)
    x = rng.uniform(0, 0.6, (7, 1801)).astype(np.float32)
    x[3] = 0.1
    assert (get_first_over_thresh(torch.from_numpy(x).cuda(), 0.55).cpu().numpy()
            == ao.get_first_over_thresh(x, 0.55)).all()


@pytest.mark.gpu
def test_hip_config5_size_vs_oracle(hip_lib_path):
    """B=4, 900 decoder steps x 200 tokens (BASELINE config 5), ragged lengths, exact ties in a row."""
    from cookietts_amd import alignment_metric
    rng = np.random.default_rng(11)
    B, dec, enc = 4, 900, 200
    al = rng.random((B, dec, enc)).astype(np.float32) ** 8
    al /= al.sum(-1, keepdims=True)
    al[1, 10, 5] = al[1, 10, 50] = 0.9                         # tie: the first index wins (torch.max)
    il = np.array([200, 150, 77, 1], np.int32)
    ol = np.array([900, 451, 32, 899], np.int32)
    ref = ao.alignment_metric(al, il, ol)
    got = alignment_metric(torch.from_numpy(al).cuda(), torch.from_numpy(il).cuda(), torch.from_numpy(ol).cuda())
    for k in KEYS:
        np.testing.assert_allclose(got[k].cpu().numpy(), ref[k], rtol=1e-5, atol=1e-5)
