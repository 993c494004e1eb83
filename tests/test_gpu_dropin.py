"""GPU tests (-m gpu) of the drop-in classes: TTS built from checkpoint files laid out as the reference writes them
(statedict.pt + config.yaml + speakers.json + stats.json), and the stand-alone model mirrors."""
import json
import os

import numpy as np
import pytest

from conftest import load_golden, states_for
from e2e_tts_amd import config as cfgmod, synth_weights as sw

pytestmark = pytest.mark.gpu


def write_checkpoints(tmp_path, cfg, ac, voc):
    import torch
    import yaml
    d = tmp_path / "exps" / "acoustic"
    v = tmp_path / "exps" / "vocoder"
    d.mkdir(parents=True)
    v.mkdir(parents=True)
    torch.save({"state_dict": sw.to_torch(ac), "optimizer": {}}, d / "statedict.pt")
    torch.save({"state_dict": sw.to_torch(voc)}, v / "statedict.pt")
    full = dict(cfg)
    full["train"] = {"seed": 1234}
    yaml.safe_dump(full, open(d / "config.yaml", "w"))
    json.dump(cfgmod.DEFAULT_SPEAKERS, open(d / "speakers.json", "w"))
    json.dump(cfgmod.DEFAULT_STATS, open(d / "stats.json", "w"))
    return str(d / "statedict.pt"), str(v / "statedict.pt")


def test_tts_from_checkpoint_files_matches_oracle(tmp_path):
    from e2e_tts_amd.api import TTS, Synthesizer
    from e2e_tts_amd.packer import variance_position_table
    from oracle import ref_numpy as orc
    cfg = cfgmod.tiny_config()
    stats = cfgmod.DEFAULT_STATS
    ac = sw.make_acoustic_state(cfg, stats, 4, seed=7, mode="varied")
    voc = sw.make_vocoder_state(cfg, seed=8)
    apath, vpath = write_checkpoints(tmp_path, cfg, ac, voc)
    tts = TTS(apath, vpath, max_len=60)
    assert tts.hop_length == 256 and tts.sample_rate == 22050 and tts.max_wav_value == 32768.0
    rng = np.random.Generator(np.random.PCG64(11))
    seqs = [list(rng.integers(4, 131, n)) for n in (25, 9, 31, 31, 14, 3)]
    pcm = tts.inference_ids(seqs, "spk_c", pitch_control=1.1, energy_control=0.9, duration_control=1.0, silence_distance=0.01)
    # oracle: same batches (the product's pack_sequences is pinned against the reference fixture on CPU)
    batches, revert = TTS.pack_sequences(seqs, 60)
    assert len(batches) >= 2
    o_ac = orc.AcousticOracle(ac, cfg, stats, var_pos_table=variance_position_table(4096, 64))
    o_voc = orc.VocoderOracle(voc, cfg)
    ref = orc.synthesize(o_ac, o_voc, batches, cfgmod.DEFAULT_SPEAKERS["spk_c"], revert, int(0.01 * 22050), 256,
                         controls=(1.1, 0.9, 1.0))
    assert pcm.dtype == np.int16 and pcm.shape == ref.shape
    close = np.abs(pcm.astype(np.int32) - ref.astype(np.int32)) <= 1
    assert close.mean() >= 0.999, close.mean()
    with pytest.raises(KeyError):
        tts.inference_ids(seqs, "nobody")
    # text path + wav writer through the service wrapper
    syn = Synthesizer(apath, vpath, output_dir=str(tmp_path / "out"), max_len=60,
                      text_to_sequence=lambda t: [4 + (ord(c) % 127) for c in t])
    path = syn.synthesis("xin chao , viet nam", save_filepath=str(tmp_path / "out" / "a.wav"), speaker_id="spk_b")
    import wave
    with wave.open(path) as f:
        assert f.getframerate() == 22050 and f.getsampwidth() == 2 and f.getnframes() > 22050 // 2
        n1 = f.getnframes()
    # speed != 1: tempo through the model's duration control (default) or WSOLA on the file (as the reference's ffmpeg atempo);
    # both name the file <name>_<speed>.wav like reference API/utils.py:164-166 and change the duration, not the sample rate
    silence = int(0.5 * 22050)
    for mode in ("duration", "wsola"):
        p2 = syn.synthesis("xin chao , viet nam", save_filepath=str(tmp_path / "out" / f"{mode}.wav"), speaker_id="spk_b", speed=2.0, speed_mode=mode)
        assert p2.endswith(f"{mode}_2.0.wav")
        with wave.open(p2) as f:
            assert f.getframerate() == 22050
            n2 = f.getnframes()
        if mode == "wsola":
            assert abs(n2 - n1 / 2) <= 2
        else:  # speech halves (up to per-phoneme rounding), the 0.5 s of trailing silence does not
            assert 0.3 * (n1 - silence) <= n2 - silence <= 0.7 * (n1 - silence)


@pytest.mark.parametrize("name", ["tiny_b3", "tiny_cf_b3", "tiny_hv_b3"])  # FFT blocks / Conformer blocks (config-selected, U/model.py:24-27) / heads + energy predictor of their own
def test_model_mirrors_match_reference_fixture(name):
    import torch
    from e2e_tts_amd.models import HifiGan, UnsupervisedFastSpeech2
    g = load_golden(name)
    cfg, ac, voc = states_for(g, name)
    m = UnsupervisedFastSpeech2(n_symbols=131, n_speakers=4, n_channels=80, config=cfg["models"]["fastspeech2"],
                                stats=cfgmod.DEFAULT_STATS)
    m.load_state_dict(sw.to_torch(ac))
    m.eval().to(torch.device("cuda", 0))
    (mel, mel_post, dur), mel_lens = m.inference(speaker=torch.tensor([int(g["speaker"])]), texts=torch.from_numpy(g["ids"]),
                                                 txt_lens=torch.from_numpy(g["lens"]), max_txt_len=g["ids"].shape[1])
    assert mel_post.is_cuda and tuple(mel_post.shape) == g["mel_post"].shape
    np.testing.assert_array_equal(dur.cpu().numpy(), g["dur"])
    np.testing.assert_array_equal(mel_lens.cpu().numpy(), g["mel_lens"])
    assert np.abs(mel_post.cpu().numpy() - g["mel_post"]).mean() < 1e-5
    v = HifiGan(cfg["models"]["hifigan"])
    v.load_state_dict(sw.to_torch(voc))
    v.eval().to(torch.device("cuda", 0))
    v.remove_weight_norm()
    wav = v(mel_post.transpose(1, 2)).squeeze(1)   # exactly how TTS.inference calls it (API/utils.py:144)
    assert np.abs(wav.cpu().numpy() - g["wav"]).mean() < 1e-5
    with pytest.raises(RuntimeError):
        m.to("cpu")


def test_synthesize_sharded_single_rank_matches_manual_batches():
    """dist.synthesize_sharded with the real engine (one rank, gloo): every utterance's PCM equals what the same padded batch gives
    when assembled by hand, returned in input order."""
    import socket
    import torch.distributed as tdist
    from e2e_tts_amd import dist as edist
    from e2e_tts_amd.runtime import engine_from_states
    g = load_golden("tiny_b3")
    cfg, ac, voc = states_for(g, "tiny_b3")
    eng = engine_from_states(cfg, cfgmod.DEFAULT_STATS, ac, voc, device=0)
    rng = np.random.Generator(np.random.PCG64(91))
    lists = [rng.integers(4, 131, size=int(n)).tolist() for n in rng.integers(3, 40, size=8)]
    with socket.socket() as sck:
        sck.bind(("127.0.0.1", 0))
        port = sck.getsockname()[1]
    tdist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        out = edist.synthesize_sharded(eng, lists, speaker=2, batch_size=3)
    finally:
        tdist.destroy_process_group()
    assert len(out) == len(lists)
    order = edist.shard_utterances([len(x) for x in lists], 1)[0]
    spk = np.array([2], np.int64)
    for start in range(0, len(order), 3):
        idx = order[start:start + 3]
        lens = np.array([len(lists[i]) for i in idx], np.int64)
        ids = np.zeros((len(idx), int(lens.max())), np.int64)
        for b, i in enumerate(idx):
            ids[b, :lens[b]] = lists[i]
        pcm, ml, _ = eng.synthesize(ids, lens, spk)
        for b, i in enumerate(idx):
            n = int(ml[b]) * 256
            assert out[i].shape == (n,)
            np.testing.assert_array_equal(out[i], pcm[b, :n])
    eng.close()


def test_inputs_written_by_pending_torch_kernels_are_ordered():
    """include/e2etts.h "STREAM ORDERING": the engine runs on a stream of its own, so a mel that torch is STILL WRITING on its
    current stream when v(mel) is called (the `.contiguous()` copy of `mel_post.transpose(1, 2)`, or the caller's own producer kernel)
    must be waited for.  A long chain of torch kernels is queued in front of the kernel that fills the mel; the result has to be
    bit-identical to the same call on a host array.  Output side: the wav lands in a block that torch's allocator has just recycled
    from a tensor a queued kernel still reads."""
    import torch
    from e2e_tts_amd.models import HifiGan
    g = load_golden("tiny_b3")
    cfg, _, voc = states_for(g, "tiny_b3")
    v = HifiGan(cfg["models"]["hifigan"])
    v.load_state_dict(sw.to_torch(voc))
    v.eval().to(torch.device("cuda", 0))
    mel_np = np.ascontiguousarray(g["mel_post"].transpose(0, 2, 1))   # [B, 80, T]
    want = v(mel_np).cpu().numpy()                                     # host input: nothing to order
    dev = torch.device("cuda", 0)
    src = torch.from_numpy(mel_np).to(dev)
    busy = torch.randn(4096, 4096, device=dev)
    for attempt in range(3):
        mel = torch.full_like(src, float("nan"))
        torch.cuda.synchronize()
        acc = busy
        for _ in range(40):                 # ~tens of ms of queued GEMMs on torch's current stream ...
            acc = (acc @ busy) * 1e-3
        mel.copy_(src + 0.0 * acc[0, 0])    # ... and only then the kernel that makes the mel valid
        got = v(mel)                        # the binding orders the engine's stream after torch's (e2etts_order_after)
        np.testing.assert_array_equal(got.cpu().numpy(), want)
        # the reference's own call pattern: a transposed view whose .contiguous() copy is a pending torch kernel
        mel_btc = torch.full((src.shape[0], src.shape[2], src.shape[1]), float("nan"), device=dev)
        acc = busy
        for _ in range(40):
            acc = (acc @ busy) * 1e-3
        mel_btc.copy_(src.transpose(1, 2) + 0.0 * acc[0, 0])
        got = v(mel_btc.transpose(1, 2))
        np.testing.assert_array_equal(got.cpu().numpy(), want)
        # output side: free a tensor that a queued kernel still reads, so that the engine's output block is that memory
        n_out = int(np.prod(want.shape))
        scratch = torch.ones(n_out, device=dev)
        acc = busy
        for _ in range(40):
            acc = (acc @ busy) * 1e-3
        check = (scratch * (1.0 + 0.0 * acc[0, 0])).sum()   # reads `scratch` after the GEMMs
        del scratch                                          # the caching allocator may hand this block to torch.empty in v()
        got = v(src)
        assert float(check.item()) == float(n_out)           # the queued reader saw its ones, not our wav
        np.testing.assert_array_equal(got.cpu().numpy(), want)


_BCAST_CHILD = r"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from e2e_tts_amd import config as cfgmod, packer, synth_weights as sw
from e2e_tts_amd._lib import Engine
import torch  # the packer folds weight norm with torch ops; importing it FIRST means its bundled librccl.so (soname librccl.so.1) is the
# one copy in this process: dlopen by that soname below -- and inside the engine -- returns the already-loaded object
try:
    rccl = C.CDLL("librccl.so.1", mode=C.RTLD_GLOBAL)
except OSError:
    rccl = C.CDLL("/opt/rocm/lib/librccl.so.1", mode=C.RTLD_GLOBAL)
uid = (C.c_char * 128)()
assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
comm = C.c_void_p()
class UID(C.Structure):
    _fields_ = [("b", C.c_char * 128)]
rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UID, C.c_int]
u = UID(); C.memmove(C.byref(u), uid, 128)
assert rccl.ncclCommInitRank(C.byref(comm), 1, u, 0) == 0, "ncclCommInitRank"
cfg = cfgmod.tiny_config(); stats = cfgmod.DEFAULT_STATS
dims = cfgmod.dims_from_config(cfg, stats, n_speakers=4)
ac = sw.make_acoustic_state(cfg, stats, 4, seed=1234, mode="varied"); voc = sw.make_vocoder_state(cfg, seed=4321)
blob = packer.pack(dims, ac, voc)
g = np.load(sys.argv[2])
ids, lens, spk = g["ids"], g["lens"], np.array([int(g["speaker"])], np.int64)
a = Engine(dims, 0); a.load_weights(blob)
b = Engine(dims, 0); b.load_weights_bcast(blob, blob.nbytes, comm.value, root=0)   # ncclBroadcast on the engine's stream
pa, la, ta = a.synthesize(ids, lens, spk)
pb, lb, tb = b.synthesize(ids, lens, spk)
assert ta == tb and np.array_equal(la, lb) and np.array_equal(pa, pb)
try:
    b.load_weights_bcast(None, blob.nbytes, comm.value, root=0)
    raise SystemExit("root without a blob was accepted")
except ValueError:
    pass
try:
    b.load_weights_bcast(blob, blob.nbytes, 0, root=0)
    raise SystemExit("NULL communicator was accepted")
except ValueError:
    pass
a.close(); b.close()
rccl.ncclCommDestroy.argtypes = [C.c_void_p]
rccl.ncclCommDestroy(comm)
print("BCAST_OK", pa.shape, ta)
"""


def test_load_weights_bcast_over_rccl_single_rank_communicator(tmp_path):
    """e2etts_load_weights_bcast: the C entry a non-Python host uses for the one RCCL collective of the path.  Run
    in a child process that creates its own one-rank communicator through RCCL's C API (torch.distributed does not expose its
    ncclComm_t) on this box's GPU; an engine loaded through the broadcast must synthesise the same PCM as one loaded
    directly.  The 8-rank form is the same call on every rank."""
    import subprocess
    import sys
    from conftest import GOLD, ROOT
    script = tmp_path / "bcast_child.py"
    script.write_text(_BCAST_CHILD)
    r = subprocess.run([sys.executable, str(script), ROOT, os.path.join(GOLD, "tiny_b3.npz")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "BCAST_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_bench_collective_branch_runs_on_rccl_single_rank():
    """bench.py's multi-GPU branch on the REAL collective backend: launched by torch.distributed.run with one rank on this box's GPU and
    E2ETTS_BENCH_FORCE_DIST=1, so init_process_group("nccl", device_id=...), the size + blob broadcasts, the barriers and the MAX
    all-reduce execute on RCCL (with N ranks the same calls run unchanged; the N = 2 contract is rehearsed on CPU in
    tests/test_dist_gloo.py)."""
    import socket
    import subprocess
    import sys
    from conftest import ROOT
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, E2ETTS_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "4",
           "--no-cpu-baseline", "--no-extras"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["collective_backend"] == "nccl" and d["rccl_ranks"] == 1 and d["n_gpus"] == 1
    assert d["weight_bcast_ms"] is not None and d["weight_bcast_ms"] >= 0 and d["weight_blob_bytes"] > 100_000_000
    assert d["dtype"] == "f32" and d["value"] > 22050 * 100


def test_gpu_tempo_kernel_follows_the_host_wsola():
    """e2etts_tempo (the WSOLA kernel behind speed_mode="wsola") against e2e_tts_amd.api.time_stretch_wsola, the same algorithm in numpy
    float64: same length, the same signal up to the rare frame where a near-tie of the cross-correlation resolves differently (the two
    sum in different orders).  PARITY UNPINNED against the reference, which shells out to ffmpeg's atempo (API/utils.py:163-172)."""
    from e2e_tts_amd.api import time_stretch_wsola
    from e2e_tts_amd.runtime import engine_from_states
    g = load_golden("tiny_b3")
    cfg, ac, voc = states_for(g, "tiny_b3")
    eng = engine_from_states(cfg, cfgmod.DEFAULT_STATS, ac, voc, device=0)
    sr = 22050
    t = np.arange(int(2.3 * sr)) / sr
    rng = np.random.Generator(np.random.PCG64(3))
    x = (6000.0 * np.sin(2 * np.pi * 220.0 * t) * (1 + 0.3 * np.sin(2 * np.pi * 3 * t)) + 1500.0 * np.sin(2 * np.pi * 1730.0 * t)
         + 200.0 * rng.standard_normal(t.size)).astype(np.int16)
    np.testing.assert_array_equal(eng.tempo(x, 1.0, sr), x)
    for speed in (0.5, 0.8, 1.25, 2.0):
        got = eng.tempo(x, speed, sr)
        ref = np.clip(np.rint(time_stretch_wsola(x.astype(np.float64), speed, sr)), -32768, 32767).astype(np.int16)
        assert got.shape == ref.shape == (round(x.size / speed),)
        same = np.abs(got.astype(np.int32) - ref.astype(np.int32)) <= 1
        assert same.mean() >= 0.98, (speed, same.mean())
        spec = np.abs(np.fft.rfft(got.astype(np.float64) * np.hanning(got.size)))
        assert abs(np.fft.rfftfreq(got.size, 1 / sr)[int(np.argmax(spec))] - 220.0) < 2.0   # pitch stays
    # the C entry takes the speed as a double, so an exact .5 tie of n_in / speed rounds as Python's round() does
    # (len % 8 == 6 at speed 0.8: n_in / 0.8 = k + 0.5; the float 0.8f = 0.800000012 would land just under it)
    x6 = x[:x.size - ((x.size - 6) % 8)]
    assert x6.size % 8 == 6 and (x6.size / 0.8) % 1 == 0.5
    got = eng.tempo(x6, 0.8, sr)
    ref = np.clip(np.rint(time_stretch_wsola(x6.astype(np.float64), 0.8, sr)), -32768, 32767).astype(np.int16)
    assert got.shape == ref.shape == (round(x6.size / 0.8),)
    assert (np.abs(got.astype(np.int32) - ref.astype(np.int32)) <= 1).mean() >= 0.98
    with pytest.raises(ValueError):
        eng.tempo(x, 8.0, sr)
    with pytest.raises(ValueError, match="sample_rate"):
        eng.tempo(x, 1.5, 192000)
    with pytest.raises(TypeError):
        eng.tempo(x.astype(np.float32), 1.5, sr)
    eng.close()


def test_gather_pcm_tensor_path_over_rccl_single_rank():
    """dist.gather_pcm's collectives -- all-reduce of the table / payload sizes, all-gather of the (index, length) tables, gather of the
    flat byte payload on the DEVICE, one copy back into page-locked memory -- on a one-rank RCCL group (the only RCCL this one-GPU box
    offers; world-size-2 runs of the same code use gloo in tests/test_dist_gloo.py).  A child process, so that this process's
    torch.distributed state stays untouched."""
    import socket
    import subprocess
    import sys
    from conftest import ROOT
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    code = (
        "import sys, numpy as np, torch, torch.distributed as dist\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from e2e_tts_amd import dist as edist\n"
        "torch.cuda.set_device(0)\n"
        f"dist.init_process_group('nccl', init_method='tcp://127.0.0.1:{port}', rank=0, world_size=1, device_id=torch.device('cuda', 0))\n"
        "rng = np.random.Generator(np.random.PCG64(5))\n"
        "local = [(i, rng.integers(-32768, 32767, size=int(n)).astype(np.int16)) for i, n in zip((3, 0, 2, 1), (7, 12001, 1, 256))]\n"
        "st = {}\n"
        "out = edist.gather_pcm(local, dst=0, device='cuda:0', stats=st, _force_collectives=True)\n"
        "assert [k for k, _ in out] == [0, 1, 2, 3]\n"
        "want = dict(local)\n"
        "for k, pcm in out: assert pcm.dtype == np.int16 and np.array_equal(pcm, want[k])\n"
        "assert st['samples_per_rank'] == [7 + 12001 + 1 + 256]\n"
        "assert edist.gather_pcm([], dst=0, device='cuda:0', _force_collectives=True) == []\n"
        "dist.destroy_process_group(); print('ok')\n")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-3000:]


_PRODUCT_CHILD = r"""
import hashlib, os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
from e2e_tts_amd import _lib, config as cfgmod, synth_weights as sw
from e2e_tts_amd.runtime import engine_from_states
lib = _lib.load_library()
assert os.path.basename(lib._name) == sys.argv[3], lib._name
cfg = cfgmod.tiny_config(); stats = cfgmod.DEFAULT_STATS
ac = sw.make_acoustic_state(cfg, stats, 4, seed=1234, mode="varied"); voc = sw.make_vocoder_state(cfg, seed=4321)
eng = engine_from_states(cfg, stats, ac, voc, device=0)
g = np.load(sys.argv[2])
pcm, mel_lens, T = eng.synthesize(g["ids"], g["lens"], np.array([int(g["speaker"])], np.int64))
try:
    eng.poison_workspace()
    hook = "hook"
except RuntimeError:
    hook = "nohook"
print("PCM_SHA", hashlib.sha256(np.ascontiguousarray(pcm).tobytes()).hexdigest(), T, hook)
"""


@pytest.mark.gpu
def test_product_library_gives_the_test_builds_bits(tmp_path):
    """The GPU tests load libe2etts_hip_test.so (E2ETTS_TEST_HOOKS=1, tests/conftest.py): the product sources plus ONE more export.  This
    runs the same utterances through both libraries in child processes -- the product one must refuse the hook and give the same PCM
    bit for bit, so what the suite proves about the test build holds for the library a host links."""
    import subprocess
    import sys
    from conftest import GOLD, ROOT
    script = tmp_path / "product_child.py"
    script.write_text(_PRODUCT_CHILD)
    outs = []
    for hooks, name, want in (("0", "libe2etts_hip.so", "nohook"), ("1", "libe2etts_hip_test.so", "hook")):
        env = dict(os.environ, E2ETTS_TEST_HOOKS=hooks)
        r = subprocess.run([sys.executable, str(script), ROOT, os.path.join(GOLD, "tiny_b3.npz"), name], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0 and "PCM_SHA" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("PCM_SHA")][0].split()
        assert line[3] == want, line
        outs.append(line[1:3])
    assert outs[0] == outs[1]
