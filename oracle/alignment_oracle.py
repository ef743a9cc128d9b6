"""CPU restatement (numpy) of the attention-alignment scoring used by the T2S retry loop.

TEST INFRASTRUCTURE ONLY: imported by tests/, never by the product path (cookietts_amd/alignment.py calls the HIP
library and fails loudly without it).

Follows the reference line by line:
  * ``get_first_over_thresh``  - CookieTTS/utils/model/utils.py:47-56 (same function again in
    _5_infer/t2s_server/text2speech.py:152-161): last column forced to the threshold, values above it clamped,
    first arg-max -> the first step whose gate reaches the threshold, else T-1.
  * ``alignment_metric``       - CookieTTS/utils/model/utils.py:59-120.
Pinned against fixtures produced by running those two functions here (tests/golden/alignment.npz).
"""
import numpy as np


def get_first_over_thresh(x, threshold):
    x = np.array(x, dtype=np.float32, copy=True)              # utils.py:50
    x[:, -1] = threshold                                      # :51
    x[x > threshold] = threshold                              # :52
    return x.argmax(axis=1).astype(np.int32)                  # :56 (first maximum)


def _mask(lengths, max_len):                                  # utils.py:8-13
    return np.arange(max_len)[None, :] < np.asarray(lengths)[:, None]


def alignment_metric(alignments, input_lengths=None, output_lengths=None, enc_min_thresh=0.7):
    """alignments [B, dec, enc] float32 -> dict of [B] arrays (utils.py:59-120, average_across_batch=False)."""
    al = np.array(alignments, dtype=np.float32, copy=True).transpose(0, 2, 1)   # :60  [B, enc, dec]
    B, enc, dec = al.shape
    if input_lengths is None:
        input_lengths = np.ones(B, np.float32) * (enc - 1)    # :65
    if output_lengths is None:
        output_lengths = np.ones(B, np.float32) * (dec - 1)   # :67
    input_lengths = np.asarray(input_lengths)
    output_lengths = np.asarray(output_lengths)
    optimums = np.sqrt(input_lengths.astype(np.float64) ** 2 + output_lengths.astype(np.float64) ** 2)   # :69

    values = al.max(axis=1)                                   # :72  [B, dec]
    cur = al.argmax(axis=1).astype(np.float32)                # :74
    prev = np.concatenate([cur[:, :1], cur[:, :-1]], axis=1)  # :75
    dist = np.sqrt((prev - cur) ** 2 + np.float32(1)).astype(np.float32)        # :76
    omask = _mask(output_lengths, dec)
    dist = np.where(omask, dist, np.float32(0))               # :77
    dist = dist.sum(axis=1, dtype=np.float32)                 # :78
    diagonalitys = (dist + np.float32(1.4142135)).astype(np.float64) / optimums  # :79

    al = np.where(omask[:, None, :], al, np.float32(0))       # :81
    total = al.sum(axis=2, dtype=np.float32)                  # :82  [B, enc]
    imask = _mask(input_lengths, enc)
    total = np.where(imask, total, np.float32(0))             # :85
    encoder_max_focus = total.max(axis=1)                     # :86
    encoder_avg_focus = total.mean(axis=1, dtype=np.float32) * (np.float32(enc) / input_lengths.astype(np.float32))  # :89-90
    total = np.where(imask, total, np.float32(1))             # :93
    encoder_min_focus = total.min(axis=1)                     # :94
    values = np.where(omask, values, np.float32(0))           # :97
    avg_prob = values.mean(axis=1, dtype=np.float32) * (np.float32(dec) / output_lengths.astype(np.float32))   # :98-99
    total = np.where(imask, total, np.float32(1e3))           # :102
    p_missing_enc = (total < np.float32(enc_min_thresh)).sum(axis=1) / input_lengths.astype(np.float32)       # :103
    return {
        "diagonalitys": diagonalitys.astype(np.float64),
        "avg_prob": avg_prob.astype(np.float32),
        "encoder_max_focus": encoder_max_focus.astype(np.float32),
        "encoder_min_focus": encoder_min_focus.astype(np.float32),
        "encoder_avg_focus": encoder_avg_focus.astype(np.float32),
        "p_missing_enc": p_missing_enc.astype(np.float32),
    }
