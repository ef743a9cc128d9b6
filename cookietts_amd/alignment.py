"""Attention-alignment scoring of the T2S retry loop, on the device.

Host mirror of ``CookieTTS/utils/model/utils.py``: ``get_first_over_thresh`` (:47-56; the same function again in
``_5_infer/t2s_server/text2speech.py:152-161``) and ``alignment_metric`` (:59-120), called once per generated
batch at text2speech.py:565-568.  The reference copies the gate row to the host for the first ("using CPU
because ...") and runs a dozen small torch ops for the second; here each is one call into the HIP library
(``csrc/alignment.hip``) and nothing leaves the GPU.  No CPU fallback: a CPU tensor or a missing library raises.
"""
import ctypes as C

import torch

from . import _lib

__all__ = ["alignment_metric", "get_first_over_thresh"]

_KEYS = ("diagonalitys", "avg_prob", "encoder_max_focus", "encoder_min_focus", "encoder_avg_focus", "p_missing_enc")


def _device_f32(x, what):
    if not (torch.is_tensor(x) and x.is_cuda):
        raise _lib.HipLibraryError(f"{what}: expected a CUDA/HIP tensor (this path has no CPU fallback)")
    return x.detach().to(torch.float32).contiguous()


def get_first_over_thresh(x, threshold):
    """[B, T] -> int32 [B]: first step whose value reaches ``threshold``; T-1 if none does (utils.py:47-56)."""
    x = _device_f32(x, "get_first_over_thresh")
    if x.dim() != 2:
        raise ValueError(f"get_first_over_thresh expects [B, T], got {tuple(x.shape)}")
    out = torch.empty(x.shape[0], dtype=torch.int32, device=x.device)
    with torch.cuda.device(x.device):
        stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        _lib.check(_lib.lib().ctts_first_over_thresh_f32(_lib.ptr(x), x.shape[0], x.shape[1], float(threshold),
                                                         _lib.ptr(out), stream), "ctts_first_over_thresh_f32")
    return out


def alignment_metric(alignments, input_lengths=None, output_lengths=None, enc_min_thresh=0.7,
                     average_across_batch=False):
    """alignments [B, dec, enc] -> dict of per-item scores (utils.py:59-120), same keys and dtypes
    (``diagonalitys`` float64, the rest float32).  Unlike the reference (:81) the argument is left untouched."""
    al = _device_f32(alignments, "alignment_metric")
    if al.dim() != 3:
        raise ValueError(f"alignment_metric expects [B, dec, enc], got {tuple(al.shape)}")
    B, dec, enc = al.shape
    dev = al.device
    il = None if input_lengths is None else _device_f32(input_lengths.to(dev), "input_lengths").reshape(B)
    ol = None if output_lengths is None else _device_f32(output_lengths.to(dev), "output_lengths").reshape(B)
    lib = _lib.lib()
    nbytes = lib.ctts_alignment_workspace_bytes(B, dec, enc)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    out = torch.empty(B, 6, dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(lib.ctts_alignment_metric_f32(_lib.ptr(al), _lib.ptr(il), _lib.ptr(ol), B, dec, enc,
                                                 float(enc_min_thresh), _lib.ptr(out), _lib.ptr(ws), nbytes, stream),
                   "ctts_alignment_metric_f32")
    res = {k: (out[:, i] if i == 0 else out[:, i].to(torch.float32)) for i, k in enumerate(_KEYS)}
    if average_across_batch:                                   # utils.py:105-111
        res = {k: v.mean() for k, v in res.items()}
    return res
