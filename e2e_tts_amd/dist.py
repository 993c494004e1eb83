"""Multi-GPU plumbing: one process per GPU, utterances sharded by the host, RCCL used once.

The hot path has no data-path collective: utterances are independent, so rank r synthesises
its own shard with its own engine.  ``torch.distributed`` (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in CPU tests) is used for (1) one broadcast of the packed weight blob from rank 0 and (2) gathering the
int16 PCM on rank 0 (44 KB per audio-second).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

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


def _world_rank() -> Tuple[int, int]:
    """(world size, rank) of the default process group; (1, 0) when torch.distributed is not initialised (one GPU, no launcher)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


def _collective_device(device=None):
    """Where the collectives' tensors live.  An explicit `device` wins; otherwise the process group decides: backend "nccl" (RCCL) moves
    device memory only -- a CPU tensor there raises 'No backend type associated with device type cpu' on every rank -- so the default is
    this process's current cuda device; gloo (CPU tests) and no group at all default to the host."""
    import torch
    import torch.distributed as dist
    if device is not None:
        return torch.device(device)
    if dist.is_available() and dist.is_initialized() and "nccl" in str(dist.get_backend()).lower():
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def broadcast_blob(blob: Optional[np.ndarray], src: int = 0, device=None):
    """Broadcast the packed weight image from rank `src`; returns a uint8 torch tensor on `device`
    (HBM when device is a cuda device: e2etts_load_weights then copies device-to-device; default: `_collective_device`)."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank()
    dev = _collective_device(device)
    n = torch.tensor([blob.size if rank == src else 0], dtype=torch.int64, device=dev)
    dist.broadcast(n, src=src)
    if rank == src:
        t = torch.from_numpy(np.ascontiguousarray(blob)).to(dev)
    else:
        t = torch.empty(int(n.item()), dtype=torch.uint8, device=dev)
    dist.broadcast(t, src=src)
    return t


def gather_pcm(local: List[Tuple[int, np.ndarray]], dst: int = 0, device=None, stats: Optional[Dict] = None, _force_collectives: bool = False):
    """Collect (utterance index, int16 PCM) pairs on rank `dst`; returns them sorted by index there, None elsewhere.

    Tensor collectives only (no pickling of the payload): every rank sends ONE flat int16 tensor -- its PCM back to back, padded to the
    longest rank's total -- plus a small (index, sample count) table, and `dst` slices views out of what it received.  `device`: where
    the collectives run -- a cuda device under backend "nccl" (RCCL moves device memory: the PCM takes one H2D copy here, rides xGMI,
    and `dst` brings all of it back with ONE D2H copy into page-locked memory), the host under gloo; None picks by backend
    (`_collective_device`).  `stats` (optional dict) receives `samples_per_rank` on `dst`.
    Every (index, PCM) pair is validated BEFORE the first collective: a rank with a bad entry must not raise alone between two
    collectives and leave the others waiting in the next one."""
    import torch
    import torch.distributed as dist
    world, rank = _world_rank()
    for idx, pcm in local:
        if not isinstance(pcm, np.ndarray) or pcm.dtype != np.int16 or pcm.ndim != 1:
            raise TypeError(f"gather_pcm: PCM of utterance {idx} must be a 1-D int16 array")
    if world == 1 and not (_force_collectives and dist.is_initialized()):   # (_force_collectives: tests run the tensor path on a one-rank group)
        merged = sorted(local, key=lambda kv: kv[0])
        if stats is not None:
            stats["samples_per_rank"] = [int(sum(p.size for _, p in local))]
        return merged
    dev = _collective_device(device)
    n_local = len(local)
    total = int(sum(p.size for _, p in local))
    head = torch.tensor([n_local, total], dtype=torch.int64, device=dev)
    dist.all_reduce(head, op=dist.ReduceOp.MAX)                     # longest table, longest payload
    max_n, max_total = int(head[0].item()), int(head[1].item())
    table = torch.full((max(max_n, 1), 2), -1, dtype=torch.int64)
    for k, (idx, pcm) in enumerate(local):
        table[k, 0], table[k, 1] = int(idx), int(pcm.size)
    tables = [torch.empty_like(table, device=dev) for _ in range(world)]
    dist.all_gather(tables, table.to(dev))
    flat = torch.zeros(max(max_total, 1), dtype=torch.int16)
    if total:
        flat[:total] = torch.from_numpy(np.concatenate([p for _, p in local]))
    flat = flat.view(torch.uint8).to(dev)      # bytes on the wire: gloo has no int16 collectives
    bucket = [torch.empty_like(flat) for _ in range(world)] if rank == dst else None
    dist.gather(flat, bucket, dst=dst)
    if rank != dst:
        return None
    if dev.type == "cuda":   # one device-to-host copy of everything, into page-locked memory
        host = torch.empty((world, flat.numel()), dtype=torch.uint8, pin_memory=True)
        host.copy_(torch.stack(bucket), non_blocking=False)
    else:
        host = torch.stack(bucket)
    host_np = host.view(torch.int16).numpy()
    merged: List[Tuple[int, np.ndarray]] = []
    per_rank = []
    for r in range(world):
        tb = tables[r].cpu().numpy()
        off = 0
        for idx, cnt in tb:
            if idx < 0:
                break
            merged.append((int(idx), host_np[r, off:off + int(cnt)]))
            off += int(cnt)
        per_rank.append(off)
    merged.sort(key=lambda kv: kv[0])
    if stats is not None:
        stats["samples_per_rank"] = per_rank
    return merged


def _agree_or_raise(failure: Optional[BaseException], device=None) -> None:
    """Every rank learns whether ANY rank failed before the first collective of the gather (one MIN all-reduce of a flag), so that a rank
    whose engine raised makes all ranks raise instead of leaving the others waiting in a collective it never enters."""
    import torch
    import torch.distributed as dist
    world, rank = _world_rank()
    if world == 1:
        if failure is not None:
            raise failure
        return
    dev = _collective_device(device)
    ok = torch.tensor([0 if failure is not None else 1], dtype=torch.int32, device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if failure is not None:
        raise RuntimeError(f"synthesize_sharded: rank {rank} failed: {failure!r}") from failure
    if int(ok.item()) == 0:
        raise RuntimeError(f"synthesize_sharded: another rank failed (rank {rank} was fine); nothing gathered")


def synthesize_sharded(engine, id_lists: Sequence[Sequence[int]], speaker: int = 0, batch_size: int = 32, hop_length: Optional[int] = None,
                       controls: Tuple[float, float, float] = (1.0, 1.0, 1.0), dst: int = 0, device=None, stats: Optional[Dict] = None):
    """BASELINE config 4 end to end (256 utterances over 8 GPUs): every rank holds the SAME list of phoneme-id lists, takes its shard
    (`shard_utterances`), synthesises it in padded batches of `batch_size` (longest first, as `TTS.input_parse` sorts,
    reference API/utils.py:84) with its own engine, and rank `dst` receives the int16 PCM of every utterance in input order
    (None elsewhere).  No collective on the data path: the only communication is the final gather of the PCM (`gather_pcm`; `device`
    as there).  Without an initialised process group it is the one-GPU form of the same loop (reference API/utils.py:130-151).
    A rank whose engine raises makes EVERY rank raise (the ranks agree on one flag before the gather), never a hang.

    `engine` is an `e2e_tts_amd._lib.Engine` (anything with `.synthesize(ids, lens, speaker, d, p, e) -> (pcm, mel_lens, T)` and
    `.dims.hop_length`).  `stats` (optional dict) receives on `dst`: `samples_per_rank`, `balance_max_over_mean` (largest rank's valid
    samples over the mean: 1.0 = perfectly even shards) and, on every rank, `batches` (this rank's batch count)."""
    world, rank = _world_rank()
    hop = hop_length if hop_length is not None else engine.dims.hop_length
    lens_all = [len(x) for x in id_lists]
    if any(n <= 0 for n in lens_all):
        raise ValueError("empty utterance")
    mine = shard_utterances(lens_all, world)[rank]          # already longest first
    spk = np.array([int(speaker)], np.int64)
    local: List[Tuple[int, np.ndarray]] = []
    n_batches = 0
    failure: Optional[BaseException] = None
    try:
        for start in range(0, len(mine), batch_size):
            idx = mine[start:start + batch_size]
            lens = np.array([lens_all[i] for i in idx], np.int64)
            ids = np.zeros((len(idx), int(lens.max())), np.int64)
            for b, i in enumerate(idx):
                ids[b, :lens[b]] = np.asarray(id_lists[i], np.int64)
            pcm, mel_lens, _ = engine.synthesize(ids, lens, spk, *controls)
            n_batches += 1
            for b, i in enumerate(idx):
                local.append((i, np.array(pcm[b, :int(mel_lens[b]) * hop], copy=True)))
    except Exception as ex:   # noqa: BLE001 -- re-raised below, on every rank
        failure = ex
    _agree_or_raise(failure, device)
    if stats is not None:
        stats["batches"] = n_batches
    merged = gather_pcm(local, dst=dst, device=device, stats=stats)
    if merged is None:
        return None
    assert [k for k, _ in merged] == list(range(len(id_lists)))
    if stats is not None and stats.get("samples_per_rank"):
        spr = stats["samples_per_rank"]
        stats["balance_max_over_mean"] = max(spr) / (sum(spr) / len(spr)) if sum(spr) else 1.0
    return [pcm for _, pcm in merged]
