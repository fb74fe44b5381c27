"""Data-parallel glue: one process per GPU, gradients summed with ONE collective over the flat fp32 gradient buffer.

Reference semantics: every rank back-propagates ``loss_r * world_size`` and DDP averages the
bucketed gradients, i.e. the applied gradient is the SUM over ranks of the local gradients; BN statistics stay per
rank and DDP re-broadcasts rank 0's buffers before every forward (``broadcast_buffers=True``).  Here: all_reduce(SUM)
of one flat tensor (RCCL over xGMI on GPUs, gloo in the CPU tests) and one broadcast of the flat buffer tensor.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def _via_host(t):
    """gloo rehearsal of the GPU path (several ranks sharing one GPU): gloo moves host memory."""
    return t.is_cuda and dist.get_backend() == "gloo"


def all_reduce_flat(flat_grad: torch.Tensor, world_size: int | None = None):
    """In-place SUM of the flat gradient buffer over all ranks (no-op for a single process)."""
    if dist.is_available() and dist.is_initialized() and (world_size or dist.get_world_size()) > 1:
        if _via_host(flat_grad):
            h = flat_grad.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            flat_grad.copy_(h)
        else:
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    return flat_grad


def broadcast_buffers(flat_buffers: torch.Tensor, src: int = 0):
    """DDP's per-forward buffer broadcast: rank ``src``'s BN running statistics win."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if _via_host(flat_buffers):
            h = flat_buffers.cpu()
            dist.broadcast(h, src=src)
            flat_buffers.copy_(h)
        else:
            dist.broadcast(flat_buffers, src=src)
    return flat_buffers


def shard_batch(batch: dict, rank: int, world_size: int) -> dict:
    """DistributedSampler-style contiguous split of a batch dict (img, batch_idx, cls, bboxes) by image index."""
    B = batch["img"].shape[0]
    assert B % world_size == 0, "global batch must divide evenly"
    b = B // world_size
    lo, hi = rank * b, (rank + 1) * b
    bi = batch["batch_idx"].reshape(-1)
    sel = (bi >= lo) & (bi < hi)
    return dict(img=batch["img"][lo:hi], batch_idx=bi[sel] - lo, cls=batch["cls"].reshape(-1, 1)[sel],
                bboxes=batch["bboxes"].reshape(-1, 4)[sel])
