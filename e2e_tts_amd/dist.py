"""Multi-GPU plumbing: one process per GPU, utterances sharded by the host, RCCL used once.

The hot path has no data-path collective: utterances are independent, so rank r synthesises
its own shard with its own engine.  ``torch.distributed`` (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in CPU tests) is used for (1) one broadcast of the packed weight blob from rank 0 and (2) gathering the
int16 PCM on rank 0 (44 KB per audio-second).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np


def shard_utterances(seq_lens: Sequence[int], world: int) -> List[List[int]]:
    """Deal utterance indices to `world` ranks in snake order over the length-sorted list, so that the summed
    length (~ frames ~ work) per rank is balanced.  Deterministic; every index appears exactly once."""
    order = np.argsort(-np.asarray(seq_lens, dtype=np.int64), kind="stable")
    shards: List[List[int]] = [[] for _ in range(world)]
    for pos, idx in enumerate(order.tolist()):
        rnd, slot = divmod(pos, world)
        rank = slot if rnd % 2 == 0 else world - 1 - slot
        shards[rank].append(idx)
    return shards


def broadcast_blob(blob: Optional[np.ndarray], src: int = 0, device=None):
    """Broadcast the packed weight image from rank `src`; returns a uint8 torch tensor on `device`
    (HBM when device is a cuda device: e2etts_load_weights then copies device-to-device)."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank()
    dev = torch.device(device) if device is not None else torch.device("cpu")
    n = torch.tensor([blob.size if rank == src else 0], dtype=torch.int64, device=dev)
    dist.broadcast(n, src=src)
    if rank == src:
        t = torch.from_numpy(np.ascontiguousarray(blob)).to(dev)
    else:
        t = torch.empty(int(n.item()), dtype=torch.uint8, device=dev)
    dist.broadcast(t, src=src)
    return t


def gather_pcm(local: List[Tuple[int, np.ndarray]], dst: int = 0):
    """Collect (utterance index, int16 PCM) pairs on rank `dst`; returns them sorted by index there, None elsewhere."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    bucket = [None] * world if rank == dst else None
    dist.gather_object(local, bucket, dst=dst)
    if rank != dst:
        return None
    merged = [item for part in bucket for item in part]
    merged.sort(key=lambda kv: kv[0])
    return merged
