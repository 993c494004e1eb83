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


def synthesize_sharded(engine, id_lists: Sequence[Sequence[int]], speaker: int = 0, batch_size: int = 32, hop_length: Optional[int] = None,
                       controls: Tuple[float, float, float] = (1.0, 1.0, 1.0), dst: int = 0):
    """BASELINE config 4 end to end (256 utterances over 8 GPUs): every rank holds the SAME list of phoneme-id lists, takes its shard
    (`shard_utterances`), synthesises it in padded batches of `batch_size` (longest first, as `TTS.input_parse` sorts,
    reference API/utils.py:84) with its own engine, and rank `dst` receives the int16 PCM of every utterance in input order
    (None elsewhere).  No collective on the data path: the only communication is the final gather of the PCM.

    `engine` is an `e2e_tts_amd._lib.Engine` (anything with `.synthesize(ids, lens, speaker, d, p, e) -> (pcm, mel_lens, T)` and
    `.dims.hop_length`)."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    hop = hop_length if hop_length is not None else engine.dims.hop_length
    lens_all = [len(x) for x in id_lists]
    if any(n <= 0 for n in lens_all):
        raise ValueError("empty utterance")
    mine = shard_utterances(lens_all, world)[rank]          # already longest first
    spk = np.array([int(speaker)], np.int64)
    local: List[Tuple[int, np.ndarray]] = []
    for start in range(0, len(mine), batch_size):
        idx = mine[start:start + batch_size]
        lens = np.array([lens_all[i] for i in idx], np.int64)
        ids = np.zeros((len(idx), int(lens.max())), np.int64)
        for b, i in enumerate(idx):
            ids[b, :lens[b]] = np.asarray(id_lists[i], np.int64)
        pcm, mel_lens, _ = engine.synthesize(ids, lens, spk, *controls)
        for b, i in enumerate(idx):
            local.append((i, np.array(pcm[b, :int(mel_lens[b]) * hop], copy=True)))
    merged = gather_pcm(local, dst=dst)
    if merged is None:
        return None
    assert [k for k, _ in merged] == list(range(len(id_lists)))
    return [pcm for _, pcm in merged]
