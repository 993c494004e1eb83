#!/usr/bin/env python3
"""Headline benchmark: audio samples / second of the FastSpeech2 + HiFi-GAN hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  Under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the environment) this process IS
one rank; called plainly (`python bench.py --gpus 8`) it starts N fresh rank processes itself -- before torch is imported or the GPU
touched in any way --, relays rank 0's one JSON line and exits with the first failing rank's code (never an exec of a process that
has initialised the GPU).

One step = one pass of the hot path (ids -> encoder -> variance adaptor -> length regulator -> decoder -> postnet -> vocoder ->
int16 PCM) over one batch of synthetic text, i.e. one iteration of the reference's TTS.inference loop (reference
e2e_tts/src/api/utils.py:130-148).  Workload (BASELINE.json metric): batch 32 per GPU, fixed-length synthetic text
L = 128 phonemes, 6 frames / phoneme -> T = 768 frames = 196 608 samples (8.92 s of 22.05 kHz audio) per utterance; default-config
random-init weights.  The ids are those of fixture tests/golden/bench_b32.npz, for which the reference's own CPU run is on record
(tests/test_gpu_parity.py::test_bench_b32_headline_workload_against_reference checks the engine against it in both precisions).

What `value` is:
  * arithmetic: EXACT fp32 everywhere (v_mfma_f32_32x32x2_f32, an fp32 FMA chain) -- the reference's precision; `dtype` = "f32";
  * timed region: SURVEY.md 8(d)'s boundary, ids in (pinned) host memory -> int16 PCM in (pinned) host memory, exactly --steps steps
    between two barrier + synchronize brackets, MAX over ranks; `ms_per_step` = that time / steps, `ms_per_step_median` the median
    of the per-step times;
  * extra keys carry the rest: `hbm_resident` (same steps with ids / PCM resident in HBM), `split_precision_mode` (the bf16x3 fast
    mode: its own ms_per_step, roofline and measured error against the reference fixtures), `latency_b1_ms`, `parity`.

Multi-GPU: utterances shard across ranks with no data-path collective ("scaling": "weak", 32 utterances per GPU); RCCL is used once,
to broadcast the packed weight blob from rank 0 (`rccl_ranks`, `weight_bcast_ms`, `weight_blob_bytes` in the JSON line).

Other BASELINE.json configurations, each under this run's clock as an extra key (and as `--workload` for a line of its own):
  `c3_mixed`     config 3: B = 32 utterances of 40..200 phonemes (padded to 200, T = 1200), ragged compute, VALID samples/s, own roofline;
  `c4_sharded`   config 4: 256 utterances (config 3's lengths x 8) through e2e_tts_amd.dist.synthesize_sharded -- snake-dealt over the
                 ranks, padded batches of 32, PCM gathered on rank 0 in input order -- with the per-rank balance (strong scaling);
  `c5_longform`  config 5: the 48 kHz generator (8 x 8 x 4 x 2, hop 512), one utterance of >= 60 s through the streaming vocoder in
                 plain bf16 / bf16x3 / fp32 (N = 1 only), own roofline against the bf16 MFMA peak;
  `latency_b1_ms` config 2: B = 1, median of 20 calls.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: peak FP32 (vector = matrix, v_mfma_f32_32x32x2_f32)
PEAK_BF16_TFLOPS = 2500.0  # same guide: dense BF16 MFMA peak; the split-precision path issues 3 bf16 MFMAs per product
BATCH = 32
PHONEMES = 128
FRAMES_PER_PHONEME = 6

# (conv_bf16 / pair_bf16 / rb_bf16: the plain-bf16 kernels of conv_bf16.hip, BASELINE config 5; "x3 " / "f32 " / "pair ": per-layer classes
# under E2ETTS_PROFILE_FINE=1)
CONV_CLASSES = ("conv_gemm", "conv_x3", "conv_bf16", "pair_bf16", "rb_bf16", "resblock", "x3 ", "f32 ", "pair ")
X3_CLASSES = ("conv_x3", "conv_bf16", "pair_bf16", "rb_bf16", "resblock", "x3 ", "pair ")


def log(*a):
    print(*a, file=sys.stderr, flush=True)


# ------------------------------------------------------------------------------------------------ self-launch (N > 1, called plainly)

def self_launch(n: int, argv) -> int:
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their
    environment, as torch.distributed.run would set them), relay rank 0's JSON line, return the first non-zero exit code.  This
    parent has not imported torch and never touches the GPU; nothing is exec'ed from a process that has."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL across processes needs it)
    base.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    import threading
    procs, errs, tails = [], [], []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        # every rank's stderr goes through this parent: relayed live, and its last lines kept so that a failure can be named
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=subprocess.PIPE, text=True))
        tails.append([])

        def relay(p=procs[-1], tail=tails[-1]):
            for line in p.stderr:
                sys.stderr.write(line)
                sys.stderr.flush()
                if line.strip():
                    tail.append(line.rstrip())
                    del tail[:-5]
        errs.append(threading.Thread(target=relay, daemon=True))
        errs[-1].start()
    out0 = procs[0].stdout.read()          # rank 0 prints the one JSON line; everything else the ranks write goes to stderr
    first_fail = None                      # (rank, exit code) of the rank that failed FIRST: the others usually die of its absence
    deadline = None
    alive = set(range(n))
    while alive:
        for r in sorted(alive):
            rc_r = procs[r].poll()
            if rc_r is None:
                continue
            alive.discard(r)
            if rc_r != 0 and first_fail is None:
                first_fail = (r, rc_r)
                deadline = time.time() + 30   # the others get half a minute to notice, then are killed by their exact PIDs
        if alive and deadline is not None and time.time() > deadline:
            for r in sorted(alive):
                procs[r].kill()            # exactly the children this parent started
                procs[r].wait()
            alive.clear()
        if alive:
            time.sleep(0.05)
    for t in errs:
        t.join(timeout=5)
    for line in (out0 or "").splitlines():
        print(line, file=sys.stdout if line.startswith("{") and first_fail is None else sys.stderr, flush=True)
    if first_fail is not None:
        r, rc_r = first_fail
        last = tails[r][-1] if tails[r] else "(no output on stderr)"
        log(f"[bench] launcher: rank {r} failed first (exit code {rc_r}); its last stderr line: {last}")
        return rc_r if rc_r > 0 else 1
    return 0


# ------------------------------------------------------------------------------------------------ CPU baseline

CPU_WARMUPS = 3
CPU_RUNS = 5
CPU_BATCH = 8          # utterances of the batch leg (B = 32 at the ~1 s per utterance measured on this pool's host share would run minutes)
CPU_BATCH_RUNS = 3


def usable_cpus():
    """(logical CPUs this process may run on, what limits them).  os.cpu_count() says 256 on the GPU box, but a lease gets a share:
    the scheduler affinity mask and the cgroup CPU quota (cpu.max, v2; cfs_quota_us, v1) are what a thread pool can really use."""
    n = os.cpu_count() or 1
    why = f"{n} logical"
    try:
        a = len(os.sched_getaffinity(0))
        if a < n:
            n, why = a, f"affinity mask {a}"
    except (AttributeError, OSError):
        pass
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: [t.strip(), open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()])):
        try:
            q, per = parse(open(path).read())
            if q != "max" and int(q) > 0:
                c = max(1, int(int(q) / int(per)))
                if c < n:
                    n, why = c, f"cgroup quota {int(q) / int(per):.1f} CPUs"
            break
        except (OSError, ValueError, IndexError):
            continue
    return n, why


def cpu_baseline(cfg, stats, ac_state, voc_state, ids_batch):
    """The oracle (a numpy port of the reference's CPU path; its two convolution primitives run through the C / OpenMP backend or
    through torch's CPU kernels, whichever is faster on this host) on bounded samples of the same workload, as BASELINE.md 4 asks:
      b1     the C2 shape -- ONE utterance of the benchmark batch, L = 128 -> T = 768 frames = 196 608 samples -- >= 3 warm-ups, median
             of 5 runs, thread count = the best of a probe that reaches every CPU this process may use;
      batch  CPU_BATCH utterances of the same batch (fixed L = 128: the headline shape, bounded) in one call on all usable CPUs, and
             the same utterances as independent single-utterance tasks over a thread pool (the better of the two is reported).
    `value` = the better samples/s of the two legs; `cores` = the threads that run used."""
    from concurrent.futures import ThreadPoolExecutor
    from threadpoolctl import threadpool_limits
    import torch
    from oracle import ref_numpy as orc
    L = ids_batch.shape[-1]
    ac = orc.AcousticOracle(ac_state, cfg, stats)
    voc = orc.VocoderOracle(voc_state, cfg)
    hop = cfg["audio"]["stft"]["hop_length"]
    ncpu = os.cpu_count() or 1
    usable, why = usable_cpus()

    def run(ids, backend, threads):
        orc.set_conv_backend(backend)
        torch.set_num_threads(threads)
        lens = np.full((ids.shape[0],), L, np.int64)
        with threadpool_limits(limits=threads):
            t0 = time.perf_counter()
            (mel, mel_post, dur), mel_lens = ac.inference(np.array([1]), ids, lens)
            wav = voc.forward(mel_post.transpose(0, 2, 1))
            dt = time.perf_counter() - t0
        assert wav.shape[-1] == int(mel_lens.max()) * hop
        return dt, int(mel_lens.sum()) * hop

    one = np.ascontiguousarray(ids_batch[:1])
    # ---- leg b1.  probe (these runs also warm caches, thread pools and oneDNN's primitive cache): both backends, thread counts up to
    # every usable CPU (and one step beyond the share, in case the quota reading is wrong)
    tc = sorted({min(ncpu, t) for t in (4, 8, 16, 32, usable, min(ncpu, 2 * usable))})
    cands = [("torch", t) for t in tc]
    if orc._c_conv():
        cands += [("c", t) for t in sorted({min(ncpu, 16), usable})]
    run(one, *cands[0])
    probe = {}
    for c in cands:
        probe[c] = run(one, *c)[0]
        if probe[c] > 4 * min(probe.values()):
            break                                  # far past the optimum: more threads only get slower
    best = min(probe, key=probe.get)
    for _ in range(max(CPU_WARMUPS - 2, 1)):
        run(one, *best)
    runs = []
    samples = 0
    for _ in range(CPU_RUNS):
        dt, samples = run(one, *best)
        runs.append(dt)
    med = statistics.median(runs)
    name = {"torch": "numpy + torch-CPU conv kernels (ATen / oneDNN)", "c": "numpy + C/OpenMP conv1d"}
    b1 = {"value": samples / med, "shape": f"C2: B=1 L={L} -> {samples} samples", "threads": best[1], "backend": name[best[0]],
          "median_s": round(med, 3), "min_max_s": [round(min(runs), 3), round(max(runs), 3)], "runs": CPU_RUNS, "warmups": CPU_WARMUPS,
          "probe_s": {f"{b}@{t}": round(v, 3) for (b, t), v in probe.items()}}
    # ---- legs batch (headline shape) and c3 (BASELINE config 3's mixed lengths), both bounded to CPU_BATCH utterances standing in for the
    # B = 32 batches, both as independent single-utterance tasks over a pool of W workers x t threads (round 3 also timed the padded batch
    # as ONE call on every usable CPU: 11-14 s against 4-5.5 s, always the loser, and a third of this function's time -- dropped)
    nb = min(CPU_BATCH, ids_batch.shape[0])
    per_task = max(1, min(best[1], max(usable // 2, 1)))   # at least two workers when there are two CPUs
    workers = max(1, min(nb, usable // per_task))

    def pool_leg(id_rows, shape):
        """id_rows: list of 1-D id arrays (one utterance each, unpadded)."""
        def once():
            orc.set_conv_backend(best[0])
            torch.set_num_threads(per_task)

            def task(ids1):
                n = int(ids1.shape[0])
                (mel, mel_post, dur), ml = ac.inference(np.array([1]), ids1.reshape(1, n), np.full((1,), n, np.int64))
                voc.forward(mel_post.transpose(0, 2, 1))
                return int(ml[0]) * hop
            with threadpool_limits(limits=per_task):
                t0 = time.perf_counter()
                with ThreadPoolExecutor(workers) as ex:
                    n = sum(ex.map(task, id_rows))
                return time.perf_counter() - t0, n
        once()   # warm-up of these shapes
        ts, n = [], 0
        for _ in range(CPU_BATCH_RUNS):
            dt, n = once()
            ts.append(dt)
        med = statistics.median(ts)
        return {"value": n / med, "shape": shape, "form": "independent_tasks", "workers": workers, "threads_per_task": per_task,
                "threads": workers * per_task, "median_s": round(med, 3), "runs": CPU_BATCH_RUNS, "warmups": 1, "samples": n,
                "backend": name[best[0]]}

    batch = pool_leg([np.ascontiguousarray(ids_batch[i]) for i in range(nb)],
                     f"{nb} utterances of the headline batch (fixed L = {L}), standing in for B = {ids_batch.shape[0]}")
    # config 3: the first CPU_BATCH utterances of the shuffled length list (40 .. 200 phonemes), ids drawn like the GPU leg's
    c3_lens = mixed_lengths(BATCH)[:nb]
    rng = np.random.Generator(np.random.PCG64(3))
    c3 = pool_leg([rng.integers(4, 131, size=int(n)).astype(np.int64) for n in c3_lens],
                  f"{nb} utterances of config 3's batch (lengths {', '.join(str(int(n)) for n in c3_lens)} of the 40 .. 200 list), standing in for B = {BATCH}")
    orc.set_conv_backend(None)
    top = max((b1, batch, c3), key=lambda leg: leg["value"])
    sr = cfg["audio"]["signal"]["sampling_rate"]
    return {"value": top["value"], "unit": "audio samples/s", "cores": top["threads"], "kind": "port",
            "threads": f"{top['threads']} of {usable} usable CPUs ({why}; {ncpu} logical on the host)",
            "sample": (f"oracle ({top['backend']}); best of three bounded legs -- b1 [{b1['shape']}; {CPU_WARMUPS}+ warm-ups, median of {CPU_RUNS} = "
                       f"{b1['median_s']} s on {b1['threads']} threads], batch [{batch['shape']}; median of {CPU_BATCH_RUNS} = {batch['median_s']} s "
                       f"on {batch['threads']} threads] and c3 [{c3['shape']}; {c3['median_s']} s on {c3['threads']} threads]"),
            "b1": b1, "batch": batch, "c3": c3, "usable_cpus": usable, "real_time_factor": top["value"] / sr}


# ------------------------------------------------------------------------------------------------ stub engine (tests only)

class StubEngine:
    """Stands in for e2e_tts_amd._lib.Engine when E2ETTS_BENCH_STUB=1 (tests/test_dist_gloo.py: the N > 1 branches of this file --
    self-launch, rendezvous, weight broadcast, barrier-bracketed timed region, MAX over ranks, one JSON line on rank 0, the sharded
    config-4 workload -- on CPU under gloo).  It computes nothing; every line it produces says data = "stub" and can never be
    mistaken for a measurement."""

    def __init__(self, dims, device=0):
        self.dims = dims
        self.loaded = 0

    def load_weights(self, blob):
        self.loaded = int(blob.numel())

    def set_precision(self, *a):
        pass

    def set_ragged(self, *a):
        pass

    def synthesize(self, ids, lens, spk, d=1.0, p=1.0, e=1.0, out_pcm=None, out_mel_lens=None, **k):
        time.sleep(0.002)
        lens = np.asarray(lens)
        T = int(ids.shape[1]) * FRAMES_PER_PHONEME
        mel_lens = lens * FRAMES_PER_PHONEME
        if out_mel_lens is not None:
            out_mel_lens[:] = mel_lens
            mel_lens = out_mel_lens
        if out_pcm is None:   # the sharded workload: PCM whose samples carry the utterance's first id (tests check the gather order)
            out_pcm = np.zeros((ids.shape[0], T * self.dims.hop_length), np.int16)
            out_pcm[:] = np.asarray(ids)[:, :1].astype(np.int16)
        return out_pcm, mel_lens, T

    def profile_filter(self, *a):
        pass

    def profile_enable(self, *a):
        pass

    def profile_read(self):
        return [dict(name="stub", launches=1, ms=1.0, flops=1e9, bytes=1e6)]


# ------------------------------------------------------------------------------------------------ helpers

def pinned(shape, dtype, torch, stub):
    """numpy view of page-locked host memory (plain memory in stub mode)."""
    t = torch.empty(shape, dtype=dtype, pin_memory=not stub)
    return t.numpy()


def roofline_of(dom, steps, traffic_json, plain_bf16=False):
    """Roofline record of one kernel class from its HIP-event statistics over `steps` steps.  plain_bf16: the launches ran in plain
    bf16 (one bf16 MFMA per product) rather than split precision (three)."""
    achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
    x3 = dom["name"].startswith(X3_CLASSES) and "_f32_" not in dom["name"]   # resblock_pair_f32_*: the fused pair on the fp32 MFMA
    peak = (PEAK_BF16_TFLOPS if plain_bf16 else PEAK_BF16_TFLOPS / 3.0) if x3 else PEAK_FP32_TFLOPS
    traffic = None if plain_bf16 else traffic_json.get(dom["name"], {}).get("hbm_bytes_per_launch")
    return {"bound": "mfma", "kernel": dom["name"], "achieved": round(achieved, 3), "peak": round(peak, 1), "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "traffic": traffic,
            "peak_note": ("bf16 dense MFMA peak 2500 (one v_mfma_f32_32x32x16_bf16 per product)" if (x3 and plain_bf16) else
                          "bf16 dense MFMA peak 2500 / 3 MFMAs per split-precision product (nominal = what the chip does on zeros; a bare MFMA "
                          "loop on random operands measures 1724-1763 TFLOP/s at 1.65 GHz on this pool = 575-588 here: profiles/r1/mfma_peak.txt)"
                          if x3 else "fp32 MFMA peak (v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD at 2.4 GHz)"),
            "algorithmic_bytes_per_launch": round(dom["bytes"] / max(dom["launches"], 1)),
            "algorithmic_flops_per_launch": round(dom["flops"] / max(dom["launches"], 1)),
            "avg_launch_us": round(dom["ms"] / max(dom["launches"], 1) * 1e3, 2),
            "launches_per_step": dom["launches"] / steps}


def class_table(stats, nsteps, tag):
    for st in sorted(stats, key=lambda st: -st["ms"]):
        tf = st["flops"] / (st["ms"] * 1e-3) / 1e12 if st["ms"] > 0 else 0.0
        gbs = st["bytes"] / (st["ms"] * 1e-3) / 1e9 if st["ms"] > 0 else 0.0
        log(f"[bench] {tag}{st['name']:<20} launches/step {st['launches'] / nsteps:7.1f}  ms/step {st['ms'] / nsteps:9.3f}  "
            f"avg {st['ms'] / max(st['launches'], 1) * 1e3:9.1f} us  {tf:7.2f} TFLOP/s  {gbs:8.1f} GB/s (algorithmic)")


def fixture_errors(eng, hop):
    """Measured error of the engine's CURRENT precision against the reference's own outputs (fixtures generated by
    oracle/make_goldens.py from the imported reference modules): c2_latency (B = 1) and bench_b32 (this workload).  Fixtures are data;
    nothing of oracle/ is imported here."""
    out = {}
    gold = os.path.join(ROOT, "tests", "golden")
    for name in ("c2_latency", "bench_b32"):
        path = os.path.join(gold, name + ".npz")
        if not os.path.exists(path):
            continue
        g = np.load(path, allow_pickle=False)
        spk = np.array([int(g["speaker"])], np.int64)
        r = eng.acoustic(g["ids"], g["lens"], spk, want=("dur", "mel_lens", "pitch_idx", "energy_idx"))
        _, mel_post = eng.fetch_mel(r["B"], r["T"], mel=False)
        wav, pcm = eng.vocoder(None, r["B"], r["T"], wav=True, pcm=True)
        rec = {"discrete_exact": bool(np.array_equal(r["dur"], g["dur"]) and np.array_equal(r["mel_lens"], g["mel_lens"]) and
                                      np.array_equal(r["pitch_idx"], g["pitch_idx"]) and np.array_equal(r["energy_idx"], g["energy_idx"]))}
        ws = int(g["wav_stride"])
        if "mel_post" in g.files:
            rec["mel_post_mean_l1"] = float(np.abs(mel_post.astype(np.float64) - g["mel_post"]).mean())
            ref_w = g["wav_strided"]
            got_w, got_p = wav[:, ::ws], pcm[:, ::ws]
        else:
            sel, fs = g["sel"], int(g["mel_frame_stride"])
            rec["mel_post_mean_l1"] = float(np.abs(mel_post[sel][:, ::fs].astype(np.float64) - g["mel_post_sel"]).mean())
            ref_w = g["wav_strided_sel"]
            got_w, got_p = wav[sel][:, ::ws], pcm[sel][:, ::ws]
        rec["wav_mean_l1"] = float(np.abs(got_w.astype(np.float64) - ref_w).mean())
        ref_p = (ref_w * np.float32(32768.0)).astype(np.int16)   # TTS.combine_audio: x 32768, truncation (API/utils.py:111-117)
        rec["pcm_within_1_lsb"] = float((np.abs(got_p.astype(np.int32) - ref_p.astype(np.int32)) <= 1).mean())
        out[name] = rec
    return out


def mixed_lengths(n=BATCH):
    """lengths linspace(40, 200, 32) rounded, shuffled with PCG64(2)."""
    lens = np.round(np.linspace(40, 200, n)).astype(np.int64)
    np.random.Generator(np.random.PCG64(2)).shuffle(lens)
    return lens


def c4_id_lists(n_utt=256, seed=1004):
    """BASELINE config 4: `n_utt` utterances with config 3's length distribution repeated (x 8 for 256), the same list on every rank."""
    rng = np.random.Generator(np.random.PCG64(seed))
    lens = np.concatenate([mixed_lengths(BATCH) for _ in range((n_utt + BATCH - 1) // BATCH)])[:n_utt]
    return [rng.integers(4, 131, size=int(n)).tolist() for n in lens]


# ------------------------------------------------------------------------------------------------ main

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra keys (hbm_resident / split_precision_mode / latency_b1_ms / parity / c3_mixed / c4_sharded / c5_longform)")
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--workload", choices=("fixed", "mixed", "c4"), default="fixed",
                    help="fixed: the headline B=32 x L=128 batch; mixed: BASELINE config 3 (32 utterances of 40..200 phonemes, padded to 200 "
                         "-> T=1200), value = VALID samples/s; c4: BASELINE config 4 (256 utterances, config 3's lengths x 8, sharded over the "
                         "ranks by e2e_tts_amd.dist.synthesize_sharded, PCM gathered on rank 0; strong scaling)")
    ap.add_argument("--utterances", type=int, default=256, help="--workload c4: utterances in the whole job")
    ap.add_argument("--precision", choices=("fp32", "bf16x3"), default="fp32",
                    help="arithmetic of the timed run: exact fp32 MFMA (default, the reference's precision, the headline) or the "
                         "split-precision bf16x3 fast mode (wav error ~1e-6; then dtype says so)")
    ap.add_argument("--no-ragged", action="store_true", help="tuning aid: compute every padded row of a mixed-length batch (set_ragged(False))")
    ap.add_argument("--blocks", choices=("transformer", "conformer"), default="transformer",
                    help="encoder / decoder building block (reference model_config.yaml:8): the headline number is quoted on the default "
                         "'transformer' FFT blocks; 'conformer' times the same workload with Conformer blocks")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # called plainly: this process becomes the launcher (it has not imported torch and never will)
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    import torch
    from e2e_tts_amd import config as cfgmod, packer, synth_weights as sw

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    stub = os.environ.get("E2ETTS_BENCH_STUB") == "1"   # CPU rehearsal of the harness with StubEngine (tests only)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (or call bench.py plainly and let it start them)")
    if not stub and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if not stub and os.environ.get("E2ETTS_BENCH_REHEARSAL") != "1" and torch.cuda.device_count() < world:
        # before init_process_group: otherwise the ranks beyond the last GPU die inside torch with "invalid device ordinal"
        raise SystemExit(f"bench.py --gpus {world}: WORLD_SIZE={world} ranks but only {torch.cuda.device_count()} GPU(s) visible to rank {rank} "
                         f"(one rank per GPU; E2ETTS_BENCH_REHEARSAL=1 rehearses the N > 1 path on one GPU under gloo)")
    # Rehearsal of the N > 1 path on a one-GPU box (E2ETTS_BENCH_REHEARSAL=1): every rank uses GPU 0 and the collectives go
    # through gloo, since RCCL refuses two ranks on one device.  Never used for a reported number.
    rehearsal = os.environ.get("E2ETTS_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    dev = torch.device("cpu") if stub else torch.device("cuda", local_rank)
    if not stub:
        torch.cuda.set_device(local_rank)
    dist = None
    backend = None
    # E2ETTS_BENCH_FORCE_DIST=1: take the collective branch even with one rank (tests/test_gpu_dropin.py runs it under torchrun on the
    # one-GPU box, so the RCCL code path -- process group on the device, broadcasts, barrier, MAX all-reduce -- has executed somewhere)
    force_dist = os.environ.get("E2ETTS_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist
        backend = "gloo" if (rehearsal or stub) else "nccl"   # "nccl" IS RCCL on ROCm
        if backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")   # where collectives move data: HBM under RCCL, host under gloo

    def sync():
        if not stub:
            torch.cuda.synchronize()

    cfg = cfgmod.default_config()
    cfg["models"]["fastspeech2"]["building_block"]["block_type"] = args.blocks
    stats = cfgmod.DEFAULT_STATS
    dims = cfgmod.dims_from_config(cfg, stats, n_speakers=4)
    hop = dims.hop_length
    ac_state = voc_state = None
    # rank 0 packs the weights; the blob travels to the other GPUs as ONE RCCL broadcast over xGMI
    bcast_dev = dev if backend != "gloo" or stub else torch.device("cpu")   # gloo moves host tensors
    if rank == 0:
        if stub:
            blob = torch.arange(1 << 20, dtype=torch.uint8)
        else:
            ac_state = sw.make_acoustic_state(cfg, stats, 4, seed=1234, mode="fixed", frames_per_phoneme=FRAMES_PER_PHONEME)
            voc_state = sw.make_vocoder_state(cfg, seed=4321)
            blob = torch.from_numpy(packer.pack(dims, ac_state, voc_state)).to(bcast_dev)
        nbytes = torch.tensor([blob.numel()], dtype=torch.int64, device=bcast_dev)
    else:
        nbytes = torch.zeros(1, dtype=torch.int64, device=bcast_dev)
    weight_bcast_ms = None
    if dist is not None:
        dist.broadcast(nbytes, src=0)
        if rank != 0:
            blob = torch.empty(int(nbytes.item()), dtype=torch.uint8, device=bcast_dev)
        dist.barrier()
        sync()
        t0 = time.perf_counter()
        dist.broadcast(blob, src=0)
        sync()
        dist.barrier()
        weight_bcast_ms = (time.perf_counter() - t0) * 1e3
        if rank == 0:
            log(f"[bench] weight blob {blob.numel() / 1e6:.1f} MB broadcast to {world} ranks over {backend} in {weight_bcast_ms:.1f} ms")
    blob_bytes = int(blob.numel())
    if stub:
        eng = StubEngine(dims, 0)
    else:
        from e2e_tts_amd._lib import Engine
        eng = Engine(dims, device=local_rank)
    eng.load_weights(blob if stub else blob.to(dev))
    eng.set_precision(args.precision)
    if args.no_ragged:
        eng.set_ragged(False)
    del blob

    def barrier():
        sync()
        if dist is not None:
            dist.barrier()
        sync()

    def timed(fn, nsteps):
        """Exactly nsteps calls between two barrier + synchronize brackets; per-step host stamps (each call is synchronous)."""
        per = []
        barrier()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            s0 = time.perf_counter()
            fn()
            per.append(time.perf_counter() - s0)
        barrier()
        return time.perf_counter() - t0, per

    def max_over_ranks(x):
        if dist is None:
            return float(x)
        el = torch.tensor([x], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return float(el.item())

    def measure(fn, nsteps):
        """Class table from two fully bracketed steps, then the timed region with events on the dominant class only (two
        hipEventRecord per launch cost ~8 us; ~155 launches per step).  Its duration is measured live, with HIP events on the
        engine's stream, over exactly the timed steps."""
        eng.profile_filter(None)
        eng.profile_enable(True)
        for _ in range(2):
            fn()
        sync()
        stats_all = eng.profile_read()
        eng.profile_enable(False)
        dom_name = max(stats_all, key=lambda st: st["ms"])["name"]
        eng.profile_filter(dom_name)
        eng.profile_enable(True)
        elapsed, per = timed(fn, nsteps)
        dom = [st for st in eng.profile_read() if st["name"] == dom_name][0]
        eng.profile_enable(False)
        eng.profile_filter(None)
        return elapsed, per, stats_all, dom

    def batch_workload(kind, B, roll=0):
        """(ids [B, L], lens [B], source note) of a one-batch workload."""
        rng = np.random.Generator(np.random.PCG64(1000 + roll))
        if kind == "mixed":
            lens_np = mixed_lengths(B)
            L = int(lens_np.max())
            ids_np = np.zeros((B, L), np.int64)
            for b, n in enumerate(lens_np):
                ids_np[b, :n] = rng.integers(4, 131, size=n)
            return ids_np, lens_np, "synthetic ids, PCG64"
        L = PHONEMES
        lens_np = np.full((B,), L, np.int64)
        ids_np = rng.integers(4, 131, size=(B, L)).astype(np.int64)
        src = "synthetic ids, PCG64"
        fx = os.path.join(ROOT, "tests", "golden", "bench_b32.npz")
        if os.path.exists(fx):
            g = np.load(fx, allow_pickle=False)
            if g["ids"].shape[1] == L:   # rank r takes the batch rolled by r rows: the same work on every GPU
                rows = np.roll(g["ids"], -roll, axis=0)
                ids_np = np.ascontiguousarray(np.resize(rows, (B, L)) if B > rows.shape[0] else rows[:B])
                src = "ids of tests/golden/bench_b32.npz (reference-pinned)"
        return ids_np, lens_np, src

    class HostBatch:
        """One batch at SURVEY.md 8(d)'s boundary: ids in (page-locked) host memory -> int16 PCM in (page-locked) host memory."""

        def __init__(self, ids_np, lens_np):
            B, L = ids_np.shape
            self.B, self.L, self.T = B, L, L * FRAMES_PER_PHONEME
            self.valid_frames = int(lens_np.sum()) * FRAMES_PER_PHONEME
            self.ids = pinned((B, L), torch.int64, torch, stub)
            self.lens = pinned((B,), torch.int64, torch, stub)
            self.spk = pinned((1,), torch.int64, torch, stub)
            self.pcm = pinned((B, self.T * hop), torch.int16, torch, stub)
            self.mel_lens = pinned((B,), torch.int64, torch, stub)
            self.ids[:], self.lens[:], self.spk[:] = ids_np, lens_np, 1

        def step(self):
            _, _, t = eng.synthesize(self.ids, self.lens, self.spk, out_pcm=self.pcm, out_mel_lens=self.mel_lens)
            return t

        def warm(self, n):
            for _ in range(max(n, 1)):
                t = self.step()
                assert t == self.T, (t, self.T)
            assert int(self.mel_lens.sum()) == self.valid_frames and int(self.mel_lens.max()) == self.T

    def run_c4(n_utt, nsteps, nwarm):
        """BASELINE config 4 through dist.synthesize_sharded; returns (elapsed max over ranks, per-step times, valid samples, stats)."""
        from e2e_tts_amd import dist as edist
        lists = c4_id_lists(n_utt)
        st = {}
        fn = lambda: edist.synthesize_sharded(eng, lists, speaker=1, batch_size=BATCH, hop_length=hop, device=coll_dev if dist is not None else None, stats=st)
        out = None
        for _ in range(max(nwarm, 1)):
            out = fn()
        if rank == 0:   # input order and exact lengths (6 frames per phoneme with the "fixed" weights)
            assert len(out) == n_utt and all(o.dtype == np.int16 and o.size == len(l) * FRAMES_PER_PHONEME * hop for o, l in zip(out, lists))
            if stub:
                assert all(int(o[0]) == l[0] and int(o[-1]) == l[0] for o, l in zip(out, lists))
        el, per = timed(fn, nsteps)
        el = max_over_ranks(el)
        samples = sum(len(l) for l in lists) * FRAMES_PER_PHONEME * hop
        return el, per, samples, st

    def c4_record(el, per, samples, st, nsteps, n_utt):
        spr = st.get("samples_per_rank") or []
        return {"utterances": n_utt, "ranks": world, "steps": nsteps, "ms_per_pass": el / nsteps * 1e3, "ms_per_pass_median": statistics.median(per) * 1e3,
                "samples_per_s": samples * nsteps / el, "real_time_factor": samples * nsteps / el / dims.sample_rate, "scaling": "strong",
                "frames_per_rank": [int(s // hop) for s in spr], "balance_max_over_mean": st.get("balance_max_over_mean"),
                "batches_per_rank": st.get("batches"),
                "what": ("e2e_tts_amd.dist.synthesize_sharded: id lists on every rank -> snake deal over length-sorted utterances -> padded batches "
                         "of 32 (longest first), ragged compute -> int16 PCM of every utterance on rank 0 in input order (one tensor gather)")}

    # ---- the timed workload
    fp32 = args.precision == "fp32"
    c4 = args.workload == "c4"
    if c4:
        elapsed, per_step, job_samples, c4_stats = run_c4(args.utterances, args.steps, args.warmup)
        value = job_samples * args.steps / elapsed
        B, L, T = BATCH, 200, 1200
        valid_frames = job_samples // hop
        ids_source = "synthetic ids, PCG64"
        # class table / dominant kernel of this rank's share (one more pass; not in the timed region)
        from e2e_tts_amd import dist as edist
        eng.profile_filter(None)
        eng.profile_enable(True)
        edist.synthesize_sharded(eng, c4_id_lists(args.utterances), speaker=1, batch_size=BATCH, hop_length=hop, device=coll_dev if dist is not None else None)
        sync()
        stats_all = eng.profile_read()
        eng.profile_enable(False)
        dom = max(stats_all, key=lambda st: st["ms"])
        prof_steps = 1
    else:
        ids_np, lens_np, ids_source = batch_workload(args.workload, args.batch, roll=rank)
        hb = HostBatch(ids_np, lens_np)
        B, L, T, valid_frames = hb.B, hb.L, hb.T, hb.valid_frames
        hb.warm(args.warmup)
        elapsed, per_step, stats_all, dom = measure(hb.step, args.steps)
        elapsed = max_over_ranks(elapsed)
        value = world * valid_frames * hop * args.steps / elapsed   # valid samples only (padding excluded)
        prof_steps = 2

    if rank == 0:
        try:
            traffic_json = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        except (OSError, ValueError):
            traffic_json = {}
        class_table(stats_all, prof_steps, "")
        roofline = roofline_of(dom, args.steps if not c4 else prof_steps, traffic_json)
        conv = [st for st in stats_all if st["name"].startswith(CONV_CLASSES)]
        roofline["all_conv_tflops"] = round(sum(st["flops"] for st in conv) / max(sum(st["ms"] for st in conv) * 1e-3, 1e-9) / 1e12, 3)
        roofline["kernel_ms_per_step"] = round(sum(st["ms"] for st in stats_all) / prof_steps, 3)
        blocks_note = ("; default model_config (6+6 FFT blocks H=384, HiFi-GAN V1), random-init weights" if args.blocks == "transformer"
                       else "; model_config with block_type=conformer (6+6 Conformer blocks H=384, 8 heads, k31; HiFi-GAN V1), random-init weights")
        if c4:
            wl = (f"BASELINE config 4: {args.utterances} utterances of 40..200 phonemes (config 3's lengths x {args.utterances // BATCH}) "
                  f"utterance-sharded over {world} GPU(s), padded batches of {BATCH}, PCM gathered on rank 0")
        elif args.workload == "fixed":
            wl = (f"B={B}/GPU fixed-length L={L} phonemes x {FRAMES_PER_PHONEME} frames = T={T} frames "
                  f"({T * hop} samples, {T * hop / dims.sample_rate:.2f} s) per utterance")
        else:
            wl = f"B={B}/GPU mixed lengths 40..200 phonemes padded to L={L} (T={T}), {valid_frames} valid frames"
        out = {
            "metric": "audio samples/sec (22.05 kHz, batch-32 per GPU, FastSpeech2 + HiFi-GAN inference, ids on host -> int16 PCM on host)",
            "value": value, "unit": "audio samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if c4 else "weak", "vs_baseline": None,
            "dtype": "f32" if fp32 else "f32 (encoder, variance adaptor) + bf16x3 split-precision (decoder, postnet, vocoder)",
            "data": "stub" if stub else "synthetic",
            "config": {"workload": wl + blocks_note + f"; {ids_source}",
                       "sample_rate": dims.sample_rate, "global_batch": args.utterances if c4 else world * B, "parallelism": f"utterance-sharded x{world}",
                       "timed_region": ("phoneme-id lists on every rank -> int16 PCM of all utterances in host memory on rank 0" if c4 else
                                        "ids in pinned host memory -> int16 PCM in pinned host memory (SURVEY.md 8(d))")},
            "real_time_factor": value / dims.sample_rate,
            "ms_per_step_median": statistics.median(per_step) * 1e3,
            "ms_per_step_min_max": [min(per_step) * 1e3, max(per_step) * 1e3],
            "roofline": roofline,
            "rccl_ranks": (dist.get_world_size() if (dist is not None and backend == "nccl") else (0 if dist is not None else 1)),
            "collective_backend": backend, "weight_bcast_ms": weight_bcast_ms, "weight_blob_bytes": blob_bytes,
        }
        if c4:
            out["c4_sharded"] = c4_record(elapsed, per_step, job_samples, c4_stats, args.steps, args.utterances)
    extras = not args.no_extras and not c4 and args.workload == "fixed" and args.batch == BATCH
    # ---- config 4 beside the headline, at every N (every rank takes part; rank 0 reports)
    if extras:
        k4 = 2 if not stub else 1
        try:   # an extra must not take the headline line down with it (an error raised on every rank alike leaves the ranks in step)
            el4, per4, s4, st4 = run_c4(256, k4, 1)
            if rank == 0:
                out["c4_sharded"] = c4_record(el4, per4, s4, st4, k4, 256)
                log(f"[bench] c4_sharded: {out['c4_sharded']['samples_per_s']:,.0f} valid samples/s, {out['c4_sharded']['ms_per_pass']:.1f} ms per pass of 256 "
                    f"utterances on {world} rank(s), balance {out['c4_sharded']['balance_max_over_mean']}")
        except Exception as ex:
            if rank == 0:
                out["c4_sharded"] = {"error": repr(ex)}
            log(f"[bench] rank {rank}: c4_sharded failed: {ex!r}")
    if rank == 0:
        if world == 1 and extras and not stub:
            # (1) the same steps with ids / PCM resident in HBM (no PCIe in the timed region)
            ids_d, lens_d = torch.from_numpy(ids_np).to(dev), torch.from_numpy(lens_np).to(dev)
            spk_d = torch.tensor([1], dtype=torch.int64, device=dev)
            pcm_d, ml_d = torch.empty((B, T * hop), dtype=torch.int16, device=dev), torch.empty((B,), dtype=torch.int64, device=dev)
            dstep = lambda: eng.synthesize(ids_d, lens_d, spk_d, out_pcm=pcm_d, out_mel_lens=ml_d)
            dstep()
            k2 = max(5, args.steps // 2)
            e2, per2 = timed(dstep, k2)
            out["hbm_resident"] = {"samples_per_s": valid_frames * hop * k2 / e2, "ms_per_step": e2 / k2 * 1e3, "steps": k2,
                                   "ms_per_step_median": statistics.median(per2) * 1e3}
            log(f"[bench] HBM-resident ids / PCM: {out['hbm_resident']['samples_per_s']:,.0f} samples/s ({e2 / k2 * 1e3:.2f} ms/step); host -> host: {value:,.0f}")
            # (2) measured error of the timed arithmetic against the reference's fixtures
            out["parity"] = {args.precision: fixture_errors(eng, hop)}
            log(f"[bench] parity ({args.precision}) vs reference fixtures: {out['parity'][args.precision]}")

            def latency(prec):
                eng.set_precision(prec)
                one = lambda: eng.synthesize(hb.ids[:1], hb.lens[:1], hb.spk, out_pcm=hb.pcm[:1], out_mel_lens=hb.mel_lens[:1])
                for _ in range(3):
                    one()
                _, per = timed(one, 20)
                return statistics.median(per) * 1e3

            out["latency_b1_ms"] = {args.precision: latency(args.precision)}
            # (3) the other arithmetic mode over the same steps, with its own roofline and its own measured error
            other = "bf16x3" if fp32 else "fp32"
            eng.set_precision(other)
            for _ in range(2):
                hb.step()
            e3, per3, stats3, dom3 = measure(hb.step, args.steps)
            class_table(stats3, 2, f"[{other}] ")
            rec = {"precision": other, "dtype": ("f32 (encoder, variance adaptor) + bf16x3 split-precision (decoder, postnet, vocoder): every fp32 "
                                                 "operand = bf16 hi + bf16 lo, product = hi*hi + hi*lo + lo*hi on the bf16 MFMA, fp32 accumulate"
                                                 if other == "bf16x3" else "f32"),
                   "ms_per_step": e3 / args.steps * 1e3, "ms_per_step_median": statistics.median(per3) * 1e3, "steps": args.steps,
                   "samples_per_s": valid_frames * hop * args.steps / e3, "real_time_factor": valid_frames * hop * args.steps / e3 / dims.sample_rate,
                   "roofline": roofline_of(dom3, args.steps, traffic_json),
                   "error_vs_reference_fixtures": fixture_errors(eng, hop),
                   "tolerance": "mel_post / wav mean-L1 <= 1e-4, int16 PCM within 1 LSB on >= 99.9 % of samples, discrete outputs exact (SURVEY.md 8(d))"}
            conv3 = [st for st in stats3 if st["name"].startswith(CONV_CLASSES)]
            rec["roofline"]["all_conv_tflops"] = round(sum(st["flops"] for st in conv3) / max(sum(st["ms"] for st in conv3) * 1e-3, 1e-9) / 1e12, 3)
            rec["roofline"]["kernel_ms_per_step"] = round(sum(st["ms"] for st in stats3) / 2, 3)
            out["split_precision_mode" if other == "bf16x3" else "fp32_mode"] = rec
            out["parity"][other] = rec["error_vs_reference_fixtures"]
            out["latency_b1_ms"][other] = latency(other)
            out["latency_b1_ms"]["what"] = "median of 20 synthesize() calls, B=1 L=128 -> 8.92 s of audio, host -> host"
            eng.set_precision(args.precision)
            log(f"[bench] {other}: {rec['ms_per_step']:.2f} ms/step, {rec['samples_per_s']:,.0f} samples/s; latency B=1 {out['latency_b1_ms']}")
            # (4) BASELINE config 3: the mixed-length batch, ragged compute, in the headline arithmetic
            ids3, lens3, _ = batch_workload("mixed", BATCH)
            hb3 = HostBatch(ids3, lens3)
            hb3.warm(2)
            k3 = max(5, args.steps // 2)
            e4, per4, stats4, dom4 = measure(hb3.step, k3)
            class_table(stats4, 2, "[c3_mixed] ")
            rl3 = roofline_of(dom4, k3, {})
            conv4 = [st for st in stats4 if st["name"].startswith(CONV_CLASSES)]
            rl3["all_conv_tflops"] = round(sum(st["flops"] for st in conv4) / max(sum(st["ms"] for st in conv4) * 1e-3, 1e-9) / 1e12, 3)
            rl3["kernel_ms_per_step"] = round(sum(st["ms"] for st in stats4) / 2, 3)
            rl3["flops_counted"] = "the rows the ragged limits leave (valid frames + receptive-field halo), not the padded batch"
            out["c3_mixed"] = {"workload": f"B={BATCH} mixed lengths 40..200 phonemes padded to L={hb3.L} (T={hb3.T}); {hb3.valid_frames} valid of {BATCH * hb3.T} padded frames; ragged compute",
                               "precision": args.precision, "steps": k3, "ms_per_step": e4 / k3 * 1e3, "ms_per_step_median": statistics.median(per4) * 1e3,
                               "valid_samples_per_s": hb3.valid_frames * hop * k3 / e4, "real_time_factor": hb3.valid_frames * hop * k3 / e4 / dims.sample_rate,
                               "us_per_valid_frame": e4 / k3 * 1e6 / hb3.valid_frames, "us_per_frame_headline": elapsed / args.steps * 1e6 / valid_frames,
                               "roofline": rl3}
            log(f"[bench] c3_mixed: {out['c3_mixed']['ms_per_step']:.2f} ms/step, {out['c3_mixed']['us_per_valid_frame']:.3f} us per valid frame "
                f"(headline {out['c3_mixed']['us_per_frame_headline']:.3f}), dominant {rl3['kernel']} {rl3['achieved']} TFLOP/s = {rl3['frac']}")
            del hb3
            # (5) BASELINE config 5: 48 kHz long-form streaming vocoder
            try:
                out["c5_longform"] = c5_longform(torch, traffic_json)
                log(f"[bench] c5_longform: {json.dumps({k: v for k, v in out['c5_longform'].items() if k in ('bf16', 'bf16x3', 'fp32')})}")
            except Exception as ex:   # an extra must not take the headline line down with it
                out["c5_longform"] = {"error": repr(ex)}
                log(f"[bench] c5_longform failed: {ex!r}")
        if world == 1 and not args.no_cpu_baseline and not stub and not c4:
            out["cpu_baseline"] = cpu_baseline(cfg, stats, ac_state, voc_state, ids_np)
            log(f"[bench] cpu_baseline: {out['cpu_baseline']}")
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


C5_FRAMES = 5632      # x 512 = 2 883 584 samples = 60.07 s at 48 kHz
C5_CHUNK = 512


def c5_longform(torch, traffic_json):
    """BASELINE config 5 on an engine of its own: the 48 kHz generator (upsample 8 x 8 x 4 x 2 = hop 512, kernels 16 / 16 / 8 / 4, width
    512; the reference's config-driven HifiGan class, V/generator.py:14-35, pinned by fixture hifigan_48k), ONE utterance of 60.07 s
    pushed through the streaming vocoder in chunks of 512 frames, int16 PCM fetched to the host per chunk.  Each arithmetic mode:
    1 warm-up pass, median of 3; class table and roofline of the dominant kernel from one more, event-bracketed pass."""
    from e2e_tts_amd import config as cfgmod, synth_weights as sw
    from e2e_tts_amd.models import HifiGan
    cfg = cfgmod.default_config()
    cfg["models"]["hifigan"].update(upsample_rates=[8, 8, 4, 2], upsample_kernel_sizes=[16, 16, 8, 4], upsample_initial_channel=512)
    cfg["audio"]["stft"]["hop_length"] = 512
    cfg["audio"]["signal"]["sampling_rate"] = 48000
    v = HifiGan(cfg["models"]["hifigan"])
    v.load_state_dict(sw.to_torch(sw.make_vocoder_state(cfg, seed=33)))
    eng = v.eval().to(0).engine
    mel = np.random.Generator(np.random.PCG64(7)).standard_normal((1, C5_FRAMES, 80)).astype(np.float32)
    rec = {"workload": f"48 kHz HiFi-GAN (8x8x4x2, hop 512, width 512), 1 utterance of {C5_FRAMES} frames = {C5_FRAMES * 512 / 48000:.2f} s, "
                       f"streamed in chunks of {C5_CHUNK} frames (two chunks in flight: push i + 1, then fetch i), PCM fetched per chunk; mel resident on the host",
           "sample_rate": 48000}

    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic_c5.json")) as fh:
            c5_traffic = json.load(fh)
    except (OSError, ValueError):
        c5_traffic = {}

    def one_chunk_size(chunk, table_tag):
        chunks = [np.ascontiguousarray(mel[:, i:i + chunk]) for i in range(0, C5_FRAMES, chunk)]
        run = lambda: sum(p.shape[1] for p in eng.vocoder_stream(chunks, 1, want_pcm=True))
        res = {}
        for prec in ("bf16", "bf16x3", "fp32"):
            eng.set_precision(prec)
            n = run()
            assert n == C5_FRAMES * 512, n
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                run()
                ts.append(time.perf_counter() - t0)
            dt = statistics.median(ts)
            eng.profile_filter(None)
            eng.profile_enable(True)
            run()
            torch.cuda.synchronize()
            st = eng.profile_read()
            eng.profile_enable(False)
            dom = max(st, key=lambda s: s["ms"])
            rl = roofline_of(dom, 1, {}, plain_bf16=(prec == "bf16"))
            conv = [s for s in st if s["name"].startswith(CONV_CLASSES)]
            rl["all_conv_tflops"] = round(sum(s["flops"] for s in conv) / max(sum(s["ms"] for s in conv) * 1e-3, 1e-9) / 1e12, 3)
            rl["kernel_ms_per_pass"] = round(sum(s["ms"] for s in st), 3)
            # HBM bytes per launch of that class from the PMC passes of tools/profile_c5.sh (chunk 512, plain bf16), when they cover it
            rl["traffic"] = c5_traffic.get(dom["name"], {}).get("hbm_bytes_per_launch") if (prec == "bf16" and chunk == C5_CHUNK) else None
            if prec == "bf16" and table_tag:
                class_table(st, 1, table_tag)
            res[prec] = {"ms": dt * 1e3, "real_time_factor": n / 48000 / dt, "samples_per_s": n / dt, "roofline": rl}
        return res

    rec.update(one_chunk_size(C5_CHUNK, "[c5 bf16] "))
    # the same stream in chunks of 2 048 frames (21.8 s of audio per push): fewer, larger launches -- what the chunk size buys
    rec["chunks_of_2048_frames"] = one_chunk_size(2048, None)
    eng.close()
    return rec


if __name__ == "__main__":
    main()
