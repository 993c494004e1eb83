#!/usr/bin/env python3
"""Headline benchmark: audio samples / second of the FastSpeech2 + HiFi-GAN hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

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

CONV_CLASSES = ("conv_gemm", "conv_x3", "resblock", "x3 ", "f32 ", "pair ")  # the last three: per-layer classes under E2ETTS_PROFILE_FINE=1
X3_CLASSES = ("conv_x3", "resblock", "x3 ", "pair ")


def log(*a):
    print(*a, file=sys.stderr, flush=True)


# ------------------------------------------------------------------------------------------------ CPU baseline

CPU_WARMUPS = 3
CPU_RUNS = 5


def cpu_baseline(cfg, stats, ac_state, voc_state, ids_row):
    """The oracle (a numpy port of the reference's CPU path; its two convolution primitives run through the C / OpenMP backend or
    through torch's CPU kernels, whichever is faster on this host) on a bounded sample of the same workload, as BASELINE.md 4 asks:
    the C2 shape -- ONE utterance of the benchmark batch, L = 128 -> T = 768 frames = 196 608 samples -- with >= 3 warm-ups and the
    median of >= 5 timed runs; thread count = the best of a short probe, reported as "N of M"."""
    from threadpoolctl import threadpool_limits
    import torch
    from oracle import ref_numpy as orc
    L = ids_row.shape[-1]
    ids = np.ascontiguousarray(ids_row.reshape(1, L))
    lens = np.full((1,), L, np.int64)
    ac = orc.AcousticOracle(ac_state, cfg, stats)
    voc = orc.VocoderOracle(voc_state, cfg)
    hop = cfg["audio"]["stft"]["hop_length"]
    ncpu = os.cpu_count() or 1

    def once(backend, threads):
        orc.set_conv_backend(backend)
        torch.set_num_threads(threads)
        with threadpool_limits(limits=threads):
            t0 = time.perf_counter()
            (mel, mel_post, dur), mel_lens = ac.inference(np.array([1]), ids, lens)
            wav = voc.forward(mel_post.transpose(0, 2, 1))
            dt = time.perf_counter() - t0
        assert wav.shape[-1] == int(mel_lens[0]) * hop
        return dt, int(mel_lens[0]) * hop

    # probe (these runs also warm caches, thread pools and oneDNN's primitive cache): both backends, a few thread counts
    cands = [("torch", t) for t in sorted({min(ncpu, t) for t in (8, 16, 32, 64)})]
    if orc._c_conv():
        cands.append(("c", min(ncpu, 16)))  # measured optimum of the C / OpenMP backend on the GPU box's 256-thread host
    once(*cands[0])
    probe = {c: once(*c)[0] for c in cands}
    best = min(probe, key=probe.get)
    for _ in range(max(CPU_WARMUPS - 2, 1)):
        once(*best)
    runs = []
    samples = 0
    for _ in range(CPU_RUNS):
        dt, samples = once(*best)
        runs.append(dt)
    orc.set_conv_backend(None)
    med = statistics.median(runs)
    name = {"torch": "numpy + torch-CPU conv kernels (ATen / oneDNN)", "c": "numpy + C/OpenMP conv1d"}[best[0]]
    return {"value": samples / med, "unit": "audio samples/s", "cores": best[1], "kind": "port",
            "threads": f"{best[1]} of {ncpu} logical cores",
            "sample": (f"oracle ({name}), C2 shape: B=1 L={L} -> {samples} samples; {CPU_WARMUPS}+ warm-ups, median of {CPU_RUNS} runs = {med:.2f} s "
                       f"(min {min(runs):.2f}, max {max(runs):.2f}); probe " + ", ".join(f"{b}@{t}: {v:.2f} s" for (b, t), v in probe.items())),
            "real_time_factor": samples / med / cfg["audio"]["signal"]["sampling_rate"]}


# ------------------------------------------------------------------------------------------------ stub engine (tests only)

class StubEngine:
    """Stands in for e2e_tts_amd._lib.Engine when E2ETTS_BENCH_STUB=1 (tests/test_dist_gloo.py: the N > 1 branch of this file --
    rendezvous, weight broadcast, barrier-bracketed timed region, MAX over ranks, one JSON line on rank 0 -- on CPU under gloo).  It
    computes nothing; every line it produces says data = "stub" and can never be mistaken for a measurement."""

    def __init__(self, dims, device=0):
        self.dims = dims
        self.loaded = 0

    def load_weights(self, blob):
        self.loaded = int(blob.numel())

    def set_precision(self, *a):
        pass

    def synthesize(self, ids, lens, spk, out_pcm=None, out_mel_lens=None, **k):
        time.sleep(0.002)
        T = int(ids.shape[1]) * FRAMES_PER_PHONEME
        if out_mel_lens is not None:
            out_mel_lens[:] = np.asarray(lens) * FRAMES_PER_PHONEME if isinstance(out_mel_lens, np.ndarray) else lens * FRAMES_PER_PHONEME
        return out_pcm, out_mel_lens, T

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


def roofline_of(dom, steps, traffic_json):
    """Roofline record of one kernel class from its HIP-event statistics over `steps` steps."""
    achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
    x3 = dom["name"].startswith(X3_CLASSES) and "_f32_" not in dom["name"]   # resblock_pair_f32_*: the fused pair on the fp32 MFMA
    peak = PEAK_BF16_TFLOPS / 3.0 if x3 else PEAK_FP32_TFLOPS
    traffic = traffic_json.get(dom["name"], {}).get("hbm_bytes_per_launch")
    return {"bound": "mfma", "kernel": dom["name"], "achieved": round(achieved, 3), "peak": round(peak, 1), "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "traffic": traffic,
            "peak_note": ("bf16 dense MFMA peak 2500 / 3 MFMAs per split-precision product (nominal = what the chip does on zeros; a bare MFMA "
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip hbm_resident / split_precision_mode / latency_b1_ms / parity (N = 1 extras)")
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--workload", choices=("fixed", "mixed"), default="fixed",
                    help="fixed: the headline B=32 x L=128 batch; mixed: BASELINE config 3 (32 utterances of 40..200 phonemes, "
                         "padded to 200 -> T=1200), value = VALID samples/s")
    ap.add_argument("--precision", choices=("fp32", "bf16x3"), default="fp32",
                    help="arithmetic of the timed run: exact fp32 MFMA (default, the reference's precision, the headline) or the "
                         "split-precision bf16x3 fast mode (wav error ~1e-6; then dtype says so)")
    ap.add_argument("--blocks", choices=("transformer", "conformer"), default="transformer",
                    help="encoder / decoder building block (reference model_config.yaml:8): the headline number is quoted on the default "
                         "'transformer' FFT blocks; 'conformer' times the same workload with Conformer blocks")
    args = ap.parse_args()

    import torch
    from e2e_tts_amd import config as cfgmod, packer, synth_weights as sw

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    stub = os.environ.get("E2ETTS_BENCH_STUB") == "1"   # CPU rehearsal of the harness with StubEngine (tests only)
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    if not stub and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
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
    del blob

    # ---- workload: ids of the reference-pinned fixture (rank r takes the batch rolled by r rows: the same work on every GPU)
    B, L = args.batch, PHONEMES
    rng = np.random.Generator(np.random.PCG64(1000 + rank))
    ids_source = "synthetic ids, PCG64"
    if args.workload == "mixed":   # lengths linspace(40, 200, 32) shuffled, padded to 200
        lens_np = np.round(np.linspace(40, 200, B)).astype(np.int64)
        np.random.Generator(np.random.PCG64(2)).shuffle(lens_np)
        L = int(lens_np.max())
        ids_np = np.zeros((B, L), np.int64)
        for b, n in enumerate(lens_np):
            ids_np[b, :n] = rng.integers(4, 131, size=n)
    else:
        lens_np = np.full((B,), L, np.int64)
        ids_np = rng.integers(4, 131, size=(B, L)).astype(np.int64)
        fx = os.path.join(ROOT, "tests", "golden", "bench_b32.npz")
        if os.path.exists(fx):
            g = np.load(fx, allow_pickle=False)
            if g["ids"].shape[1] == L:
                rows = np.roll(g["ids"], -rank, axis=0)
                ids_np = np.ascontiguousarray(np.resize(rows, (B, L)) if B > rows.shape[0] else rows[:B])
                ids_source = "ids of tests/golden/bench_b32.npz (reference-pinned)"
    T = L * FRAMES_PER_PHONEME
    valid_frames = int(lens_np.sum()) * FRAMES_PER_PHONEME
    # SURVEY.md 8(d) boundary: ids on the host -> int16 PCM on the host (page-locked buffers, as a serving loop would keep them)
    ids_h, lens_h = pinned((B, L), torch.int64, torch, stub), pinned((B,), torch.int64, torch, stub)
    spk_h, pcm_h, mel_lens_h = pinned((1,), torch.int64, torch, stub), pinned((B, T * hop), torch.int16, torch, stub), pinned((B,), torch.int64, torch, stub)
    ids_h[:], lens_h[:], spk_h[:] = ids_np, lens_np, 1

    def step():
        _, _, t = eng.synthesize(ids_h, lens_h, spk_h, out_pcm=pcm_h, out_mel_lens=mel_lens_h)
        return t

    for _ in range(max(args.warmup, 1)):
        t = step()
        assert t == T, (t, T)
    assert int(mel_lens_h.sum()) == valid_frames and int(mel_lens_h.max()) == T

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

    elapsed, per_step, stats_all, dom = measure(step, args.steps)
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend != "gloo" else torch.device("cpu"))
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())

    samples_per_step = world * valid_frames * hop  # valid samples only (padding excluded)
    value = samples_per_step * args.steps / elapsed

    if rank == 0:
        try:
            traffic_json = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        except (OSError, ValueError):
            traffic_json = {}
        class_table(stats_all, 2, "")
        roofline = roofline_of(dom, args.steps, traffic_json)
        conv = [st for st in stats_all if st["name"].startswith(CONV_CLASSES)]
        roofline["all_conv_tflops"] = round(sum(st["flops"] for st in conv) / max(sum(st["ms"] for st in conv) * 1e-3, 1e-9) / 1e12, 3)
        roofline["kernel_ms_per_step"] = round(sum(st["ms"] for st in stats_all) / 2, 3)
        fp32 = args.precision == "fp32"
        out = {
            "metric": "audio samples/sec (22.05 kHz, batch-32 per GPU, FastSpeech2 + HiFi-GAN inference, ids on host -> int16 PCM on host)",
            "value": value, "unit": "audio samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if fp32 else "f32 (encoder, variance adaptor) + bf16x3 split-precision (decoder, postnet, vocoder)",
            "data": "stub" if stub else "synthetic",
            "config": {"workload": (f"B={B}/GPU fixed-length L={L} phonemes x {FRAMES_PER_PHONEME} frames = T={T} frames "
                                    f"({T * hop} samples, {T * hop / dims.sample_rate:.2f} s) per utterance" if args.workload == "fixed" else
                                    f"B={B}/GPU mixed lengths 40..200 phonemes padded to L={L} (T={T}), {valid_frames} valid frames")
                                   + ("; default model_config (6+6 FFT blocks H=384, HiFi-GAN V1), random-init weights" if args.blocks == "transformer"
                                      else "; model_config with block_type=conformer (6+6 Conformer blocks H=384, 8 heads, k31; HiFi-GAN V1), random-init weights")
                                   + f"; {ids_source}",
                       "sample_rate": dims.sample_rate, "global_batch": world * B, "parallelism": f"utterance-sharded x{world}",
                       "timed_region": "ids in pinned host memory -> int16 PCM in pinned host memory (SURVEY.md 8(d))"},
            "real_time_factor": value / dims.sample_rate,
            "ms_per_step_median": statistics.median(per_step) * 1e3,
            "ms_per_step_min_max": [min(per_step) * 1e3, max(per_step) * 1e3],
            "roofline": roofline,
            "rccl_ranks": (dist.get_world_size() if (dist is not None and backend == "nccl") else (0 if dist is not None else 1)),
            "collective_backend": backend, "weight_bcast_ms": weight_bcast_ms, "weight_blob_bytes": blob_bytes,
        }
        if world == 1 and not args.no_extras and not stub:
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
                one = lambda: eng.synthesize(ids_h[:1], lens_h[:1], spk_h, out_pcm=pcm_h[:1], out_mel_lens=mel_lens_h[:1])
                for _ in range(3):
                    one()
                _, per = timed(one, 20)
                return statistics.median(per) * 1e3

            out["latency_b1_ms"] = {args.precision: latency(args.precision)}
            # (3) the other arithmetic mode over the same steps, with its own roofline and its own measured error
            other = "bf16x3" if fp32 else "fp32"
            eng.set_precision(other)
            for _ in range(2):
                step()
            e3, per3, stats3, dom3 = measure(step, args.steps)
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
        if world == 1 and not args.no_cpu_baseline and not stub:
            out["cpu_baseline"] = cpu_baseline(cfg, stats, ac_state, voc_state, ids_np[0])
            log(f"[bench] cpu_baseline: {out['cpu_baseline']}")
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
