"""world_size-2 `gloo` rehearsal of the multi-GPU path on CPU: weight-blob broadcast, utterance sharding, PCM gather.
(The data path itself has no collective; on the GPU box the same code runs with backend "nccl" = RCCL.)"""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

from conftest import ROOT
from e2e_tts_amd import dist as edist


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_utterances_partition_and_balance():
    rng = np.random.Generator(np.random.PCG64(0))
    lens = rng.integers(5, 300, size=101)
    for world in (1, 2, 3, 8):
        shards = edist.shard_utterances(lens, world)
        flat = sorted(i for s in shards for i in s)
        assert flat == list(range(len(lens)))
        loads = [int(lens[s].sum()) for s in shards]
        assert max(loads) - min(loads) <= 300, loads
    assert edist.shard_utterances([], 4) == [[], [], [], []]
    assert edist.shard_utterances([3], 2) == [[0], []]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from e2e_tts_amd import config as cfgmod, packer, synth_weights as sw
    cfg = cfgmod.tiny_config()
    dims = cfgmod.dims_from_config(cfg, cfgmod.DEFAULT_STATS, 4)
    blob = None
    if rank == 0:
        blob = packer.pack(dims, sw.make_acoustic_state(cfg, cfgmod.DEFAULT_STATS, 4, mode="varied"), sw.make_vocoder_state(cfg))
    t = edist.broadcast_blob(blob, src=0)
    np.save(os.path.join(out_dir, f"blob_sum_{rank}.npy"), np.array([t.numel(), int(t.to(dtype=__import__("torch").int64).sum())]))
    # each rank "synthesises" its shard: the stand-in PCM encodes the utterance index so the gather can be checked
    lens = [40, 7, 33, 33, 12, 90, 5]
    mine = edist.shard_utterances(lens, world)[rank]
    local = [(i, np.full(lens[i], i, dtype=np.int16)) for i in mine]
    merged = edist.gather_pcm(local, dst=0)
    if rank == 0:
        assert [k for k, _ in merged] == list(range(len(lens)))
        for k, pcm in merged:
            assert pcm.shape == (lens[k],) and (pcm == k).all()
        np.save(os.path.join(out_dir, "gather_ok.npy"), np.array([1]))
    else:
        assert merged is None
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_and_gather_world2(tmp_path):
    port = free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = np.load(tmp_path / "blob_sum_0.npy")
    b = np.load(tmp_path / "blob_sum_1.npy")
    np.testing.assert_array_equal(a, b)
    assert a[0] > 1_000_000
    assert (tmp_path / "gather_ok.npy").exists()


class _StubEngine:
    """Stands in for the HIP engine (no GPU in the CPU suite): 'synthesises' 3 frames per phoneme whose samples encode the first id
    of the utterance, and records the batches it was given."""
    class dims:
        hop_length = 4

    def __init__(self):
        self.batches = []

    def synthesize(self, ids, lens, speaker, d=1.0, p=1.0, e=1.0):
        self.batches.append((ids.shape, lens.tolist()))
        assert (np.diff(lens) <= 0).all()                    # longest first inside a batch
        mel_lens = lens * 3
        T = int(mel_lens.max())
        pcm = np.zeros((ids.shape[0], T * 4), np.int16)
        for b in range(ids.shape[0]):
            assert (ids[b, lens[b]:] == 0).all() and (ids[b, :lens[b]] > 0).all()
            pcm[b, :mel_lens[b] * 4] = ids[b, 0]
        return pcm, mel_lens, T


def _sharded_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.Generator(np.random.PCG64(11))           # same list on every rank
    lists = [[int(k + 1)] + rng.integers(4, 131, size=int(n) - 1).tolist() for k, n in enumerate(rng.integers(1, 60, size=23))]
    eng = _StubEngine()
    out = edist.synthesize_sharded(eng, lists, speaker=1, batch_size=5)
    assert all(shape[0] <= 5 for shape, _ in eng.batches)
    if rank == 0:
        assert len(out) == len(lists)
        for k, (pcm, ids) in enumerate(zip(out, lists)):
            assert pcm.dtype == np.int16 and pcm.shape == (len(ids) * 3 * 4,) and (pcm == k + 1).all()
        np.save(os.path.join(out_dir, "sharded_ok.npy"), np.array([len(eng.batches)]))
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_synthesize_sharded_world2(tmp_path):
    """BASELINE config 4's flow (utterance-sharded batch) with two ranks: shard, batch, 'synthesise', gather in input order on rank 0."""
    port = free_port()
    mp.spawn(_sharded_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "sharded_ok.npy").exists()


def test_bench_multi_rank_branch_contract_world2():
    """bench.py's N > 1 branch on CPU: `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` with E2ETTS_BENCH_STUB=1
    (a stand-in engine that computes nothing) under gloo -- rendezvous from the env, weight broadcast from rank 0, barrier-bracketed
    timed region, MAX over ranks, exactly ONE JSON line (rank 0) carrying the driver's keys plus the multi-GPU ones."""
    import json
    import subprocess
    env = dict(os.environ, E2ETTS_BENCH_STUB="1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["data"] == "stub"                                    # can never be mistaken for a measurement
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 64 and d["config"]["parallelism"] == "utterance-sharded x2"
    assert d["collective_backend"] == "gloo" and d["rccl_ranks"] == 0   # rccl_ranks counts ranks of an RCCL ("nccl") group only
    assert d["weight_bcast_ms"] > 0 and d["weight_blob_bytes"] == 1 << 20
    samples = 2 * 32 * 768 * 256                                  # whole-job aggregate: both ranks' utterances
    assert abs(d["value"] - samples / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert "cpu_baseline" not in d and "split_precision_mode" not in d   # N = 1 extras
    for k in ("metric", "unit", "higher_is_better", "vs_baseline", "dtype", "roofline", "ms_per_step_median"):
        assert k in d, k
