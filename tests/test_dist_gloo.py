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


def _failing_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lists = [[k + 1] * (k + 2) for k in range(9)]
    eng = _StubEngine()
    if rank == 1:
        def boom(*a, **k):
            raise ValueError("engine error on one rank")
        eng.synthesize = boom
    try:
        edist.synthesize_sharded(eng, lists, speaker=1, batch_size=4)
    except RuntimeError as ex:
        with open(os.path.join(out_dir, f"raised{rank}.txt"), "w") as fh:
            fh.write(str(ex))
    dist.barrier()   # both ranks are still in step with each other
    dist.destroy_process_group()


def test_synthesize_sharded_one_failing_rank_fails_all_ranks(tmp_path):
    """An engine error on one rank must not leave the others waiting in the gather: every rank raises."""
    port = free_port()
    mp.spawn(_failing_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert "rank 1 failed" in (tmp_path / "raised1.txt").read_text()
    assert "another rank failed" in (tmp_path / "raised0.txt").read_text()


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


def _run_bench_plainly(*args):
    import json
    import subprocess
    env = dict(os.environ, E2ETTS_BENCH_STUB="1", OMP_NUM_THREADS="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout      # rank 0's line and nothing else on stdout
    return json.loads(lines[0])


def test_bench_called_plainly_with_gpus_2_launches_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher in the command.  The parent starts two fresh rank processes
    before importing torch, relays rank 0's one JSON line and exits 0; the line carries the headline keys for n_gpus = 2 and the
    config-4 record (256 utterances sharded over both ranks) beside them.  Stand-in engine, gloo: the harness, not a measurement."""
    d = _run_bench_plainly("--gpus", "2", "--steps", "3", "--warmup", "1")
    assert d["data"] == "stub" and d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 64 and d["collective_backend"] == "gloo" and d["rccl_ranks"] == 0
    assert d["weight_bcast_ms"] > 0
    c4 = d["c4_sharded"]
    assert c4["utterances"] == 256 and c4["ranks"] == 2 and c4["scaling"] == "strong"
    assert len(c4["frames_per_rank"]) == 2 and sum(c4["frames_per_rank"]) == 8 * 23040 // 6 * 6   # config 3's 3 840 phonemes x 8 x 6 frames
    assert 1.0 <= c4["balance_max_over_mean"] <= 1.01 and c4["batches_per_rank"] == 4
    assert "cpu_baseline" not in d and "c5_longform" not in d


def test_bench_rank_failure_is_reported_by_the_launcher():
    """A rank that dies takes the launcher's exit code with it, and no JSON line appears."""
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS="1")       # no stub: without a GPU every rank refuses to run
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "E2ETTS_BENCH_STUB"):
        env.pop(k, None)
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is visible")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "no CPU fallback" in r.stderr
    # the launcher names the rank that failed first and repeats its last stderr line
    assert "failed first (exit code" in r.stderr and "its last stderr line: bench.py needs a GPU" in r.stderr


def test_bench_called_plainly_with_gpus_8_under_gloo():
    """The N = 8 shape of the driver's scaling run, rehearsed here with the stand-in engine over gloo (the driver's own N = 8 run needs an
    8-GPU node): eight self-launched ranks, one JSON line, whole-job aggregate over all eight, the config-4 list dealt evenly."""
    d = _run_bench_plainly("--gpus", "8", "--steps", "2", "--warmup", "1")
    assert d["data"] == "stub" and d["n_gpus"] == 8 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 256 and d["config"]["parallelism"] == "utterance-sharded x8"
    samples = 8 * 32 * 768 * 256
    assert abs(d["value"] - samples / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    c4 = d["c4_sharded"]
    assert c4["utterances"] == 256 and c4["ranks"] == 8 and len(c4["frames_per_rank"]) == 8 and c4["batches_per_rank"] == 1
    assert 1.0 <= c4["balance_max_over_mean"] <= 1.01


def test_collectives_default_to_the_groups_device(monkeypatch):
    """`synthesize_sharded(engine, id_lists, ...)` -- the documented call, device=None -- must build its collective
    tensors where the process group can move them: on this process's cuda device under backend "nccl" (RCCL has no CPU tensors), on the
    host under gloo or without a group.  The backend query is patched: no RCCL group exists on a CPU box."""
    import torch
    import torch.distributed as dist
    assert edist._collective_device(None) == torch.device("cpu")            # no group
    assert edist._collective_device("cpu") == torch.device("cpu")
    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    monkeypatch.setattr(dist, "get_backend", lambda *a, **k: "gloo")
    assert edist._collective_device(None) == torch.device("cpu")
    monkeypatch.setattr(dist, "get_backend", lambda *a, **k: "nccl")
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 3)
    assert edist._collective_device(None) == torch.device("cuda", 3)
    assert edist._collective_device("cuda:1") == torch.device("cuda", 1)    # an explicit device wins


def test_gather_pcm_validates_before_any_collective(monkeypatch):
    """a rank holding a bad (index, PCM) entry must raise BEFORE the first collective, not between two of them (the other
    ranks would wait in the next one).  Every collective is patched to fail the test if it is reached."""
    import pytest
    import torch.distributed as dist
    monkeypatch.setattr(edist, "_world_rank", lambda: (2, 0))
    for name in ("all_reduce", "all_gather", "gather"):
        monkeypatch.setattr(dist, name, lambda *a, _n=name, **k: pytest.fail(f"dist.{_n} reached before validation"))
    with pytest.raises(TypeError, match="utterance 1"):
        edist.gather_pcm([(0, np.zeros(4, np.int16)), (1, np.zeros(4, np.float32))])
    with pytest.raises(TypeError, match="utterance 0"):
        edist.gather_pcm([(0, np.zeros((2, 2), np.int16))])


def test_bench_workload_c4_world2():
    """`--workload c4` (BASELINE config 4) under gloo with two self-launched ranks and the stand-in engine: the sharded loop runs on
    both ranks, rank 0 receives every utterance's PCM in input order (bench.py asserts order and exact lengths on the first pass), and
    the line reports strong scaling over the whole job plus the per-rank balance."""
    d = _run_bench_plainly("--gpus", "2", "--workload", "c4", "--utterances", "64", "--steps", "2", "--warmup", "1")
    assert d["data"] == "stub" and d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["global_batch"] == 64
    c4 = d["c4_sharded"]
    assert c4["utterances"] == 64 and c4["ranks"] == 2 and c4["batches_per_rank"] == 1
    frames = 2 * 23040                                   # config 3's lengths twice, 6 frames per phoneme
    assert sum(c4["frames_per_rank"]) == frames and max(c4["frames_per_rank"]) - min(c4["frames_per_rank"]) <= 6 * 200
    assert 1.0 <= c4["balance_max_over_mean"] < 1.03
    assert abs(d["value"] - frames * 256 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6     # whole-job valid samples over the MAX-over-ranks time
    assert "timed_region" in d["config"] and "rank 0" in d["config"]["timed_region"]


def test_synthesize_sharded_without_a_process_group_is_the_one_gpu_loop():
    rng = np.random.Generator(np.random.PCG64(3))
    lists = [[int(k + 1)] + rng.integers(4, 131, size=int(n) - 1).tolist() for k, n in enumerate(rng.integers(1, 50, size=11))]
    eng = _StubEngine()
    st = {}
    out = edist.synthesize_sharded(eng, lists, speaker=0, batch_size=4, stats=st)
    assert len(out) == 11 and all((pcm == k + 1).all() and pcm.size == len(ids) * 12 for k, (pcm, ids) in enumerate(zip(out, lists)))
    assert st["batches"] == 3 and st["balance_max_over_mean"] == 1.0 and st["samples_per_rank"] == [sum(len(x) for x in lists) * 12]
