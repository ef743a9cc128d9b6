"""Utterance-batch sharding of vocoder inference across the GPUs of one node.

The reference has no multi-GPU inference (its ``start_worker`` is a stub,
``_5_infer/t2s_server/text2speech.py:779-787``).  Utterances are independent through the
whole path, so the shard unit is the utterance: every rank holds a full
weight replica and runs its slice of the batch; there is no collective inside the hot
path.  The only exchange steps are the ones a serving process needs around it:

  * ``broadcast_state_dict``  - once per model, rank 0 -> all (RCCL broadcast over xGMI);
  * ``scatter_mels``          - per request batch, rank 0 -> each rank's slice;
  * ``gather_waves``          - per request batch, every rank -> rank 0, direct
                                point-to-point (a ring would be bound by one xGMI link).

One process per GPU, ``torch.distributed`` (backend "nccl" = RCCL on ROCm; "gloo" on CPU for
the tests).  Nothing here touches the HIP library: it moves tensors only.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import torch
import torch.distributed as dist

__all__ = ["shard_counts", "broadcast_state_dict", "scatter_mels", "gather_waves", "sharded_infer"]


def shard_counts(n_items: int, world: int) -> List[int]:
    """Contiguous split, first ``n_items % world`` ranks take one extra utterance."""
    base, extra = divmod(n_items, world)
    return [base + (1 if r < extra else 0) for r in range(world)]


def broadcast_state_dict(module: torch.nn.Module, src: int = 0, group=None, bucket_bytes: int = 256 << 20) -> int:
    """Make every rank's parameters/buffers equal to rank ``src``'s (in place); returns the bytes broadcast.

    Tensors are coalesced per dtype into flat buckets of up to ``bucket_bytes`` so the 905 MB fp32 WaveGlow
    replica goes out as four large RCCL broadcasts instead of ~500 small ones (xGMI is per-link bound:
    few, large messages).  Packed-weight caches of the module are dropped afterwards.
    """
    rank = dist.get_rank(group)
    tensors = [t.data for t in list(module.parameters()) + list(module.buffers())]
    total = 0
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault((t.dtype, t.device), []).append(t)
    for (dtype, device), ts in by_dtype.items():
        cap = max(bucket_bytes // max(ts[0].element_size(), 1), 1)
        i = 0
        while i < len(ts):
            j, n = i, 0
            while j < len(ts) and (j == i or n + ts[j].numel() <= cap):
                n += ts[j].numel()
                j += 1
            if j - i == 1 and ts[i].is_contiguous():
                dist.broadcast(ts[i], src=src, group=group)
            else:
                flat = torch.empty(n, dtype=dtype, device=device)
                if rank == src:
                    torch.cat([t.reshape(-1) for t in ts[i:j]], out=flat)
                dist.broadcast(flat, src=src, group=group)
                if rank != src:
                    off = 0
                    for t in ts[i:j]:
                        t.copy_(flat[off:off + t.numel()].view_as(t))
                        off += t.numel()
            total += n * ts[i].element_size()
            i = j
    if hasattr(module, "repack"):
        module.repack()
    return total


def scatter_mels(mels: Optional[torch.Tensor], n_mel: int, device, src: int = 0, group=None, dtype=torch.float32):
    """rank ``src`` holds ``mels`` [N, n_mel, F]; returns this rank's slice [n_r, n_mel, F] in ``dtype``.

    Slices are padded to the largest shard so the collective is regular; the pad is cut
    off again before returning.  ``dtype=torch.bfloat16`` is the wire format SURVEY.md 8e names for config 3
    (32 x 80 x 900 bf16 = 4.6 MB per rank, half the fp32 scatter).  The bf16 WN path rounds the mel to bf16 before its first
    conditioning GEMM, which is the only consumer of the mel in ``glow.py`` WaveGlow, so for THAT model the wire rounding is
    the rounding the model applies itself; a model that feeds the mel to an fp32 stage first (the ax core's upsampling
    stack) would see a different input - keep fp32 on the wire there.  Every rank must pass the same ``dtype``.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    meta = torch.zeros(2, dtype=torch.int64, device=device)
    if rank == src:
        assert mels is not None and mels.dim() == 3 and mels.shape[1] == n_mel
        meta[0], meta[1] = mels.shape[0], mels.shape[2]
    dist.broadcast(meta, src=src, group=group)
    n_items, frames = int(meta[0]), int(meta[1])
    counts = shard_counts(n_items, world)
    cmax = max(max(counts), 1)
    recv = torch.empty(cmax, n_mel, frames, dtype=dtype, device=device)
    chunks = None
    if rank == src:
        chunks, start = [], 0
        for c in counts:
            buf = torch.zeros(cmax, n_mel, frames, dtype=dtype, device=device)
            buf[:c] = mels[start:start + c].to(device=device, dtype=dtype)
            chunks.append(buf)
            start += c
    dist.scatter(recv, chunks, src=src, group=group)
    return recv[:counts[rank]], counts


_WAVE_DTYPES = (torch.float32, torch.bfloat16, torch.float16, torch.int16)


def gather_waves(wave: Optional[torch.Tensor], counts: Sequence[int], dst: int = 0, group=None, device=None):
    """Every rank contributes ``wave`` [n_r, T]; rank ``dst`` gets [sum n_r, T], others None.

    A rank whose share is empty (``counts[rank] == 0``) may pass ``None``: the slab width and dtype are agreed with one
    16-byte MAX all-reduce, so such a rank never has to run an inference just to learn the output shape."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    cmax = max(max(counts), 1)
    if wave is not None:
        device = wave.device
    meta = torch.zeros(2, dtype=torch.int64, device=device)
    if wave is not None and wave.shape[0] > 0:
        meta[0], meta[1] = wave.shape[1], _WAVE_DTYPES.index(wave.dtype)
    dist.all_reduce(meta, op=dist.ReduceOp.MAX, group=group)
    T, dtype = int(meta[0]), _WAVE_DTYPES[int(meta[1])]
    send = torch.zeros(cmax, T, dtype=dtype, device=device)
    if wave is not None and wave.shape[0] > 0:
        send[:wave.shape[0]] = wave
    bufs = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)


def sharded_infer(infer_fn: Callable[[torch.Tensor], torch.Tensor], mels: Optional[torch.Tensor],
                  n_mel: int, device, root: int = 0, group=None, wire_dtype=torch.float32):
    """scatter -> local ``infer_fn(mel_slice) -> [n_r, T]`` -> gather.  Returns waves on ``root``.
    ``wire_dtype=torch.bfloat16`` ships the mels as bf16 (config 3); ``infer_fn`` receives that dtype.
    A rank that received no utterance (fewer utterances than ranks) does not call ``infer_fn`` at all."""
    local, counts = scatter_mels(mels, n_mel, device, src=root, group=group, dtype=wire_dtype)
    wave = infer_fn(local) if local.shape[0] > 0 else None
    return gather_waves(wave, counts, dst=root, group=group, device=device)
