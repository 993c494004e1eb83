#!/usr/bin/env python3
"""Headline benchmark: audio samples / second of the FastSpeech2 + HiFi-GAN hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

One step = one pass of the hot path (ids -> encoder -> variance adaptor -> length regulator -> decoder ->
postnet -> vocoder -> int16 PCM) over one batch of synthetic text, i.e. one iteration of the reference's
TTS.inference loop (reference e2e_tts/src/api/utils.py:130-148).  Workload (BASELINE.json metric):
batch 32 per GPU, fixed-length synthetic text L = 128 phonemes, 6 frames / phoneme -> T = 768 frames =
196 608 samples (8.92 s of 22.05 kHz audio) per utterance; default-config random-init weights; fp32.
Inputs (ids, lens, speaker) are resident in HBM before the timed region and the PCM stays in HBM; the
host-inclusive rate is printed to stderr and recorded in DESIGN.md.

Multi-GPU: utterances shard across ranks with no data-path collective ("scaling": "weak", 32 utterances per
GPU); RCCL is used once, to broadcast the packed weight blob from rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
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


CONV_CLASSES = ("conv_gemm", "conv_x3", "resblock_pair", "x3 ", "f32 ", "pair ")  # the last two: per-layer classes under E2ETTS_PROFILE_FINE=1


def log(*a):
    print(*a, file=sys.stderr, flush=True)


CPU_THREADS = 16  # measured on the GPU box's 256-core host: the oracle is fastest at 16 threads (32: -18 %, 64: -50 %, 256: -94 %)
CPU_BATCH = 8


def cpu_baseline(cfg, stats, ac_state, voc_state):
    """The oracle (numpy + its plain-C / OpenMP conv1d: a port of the reference's CPU path) timed on this host's cores on a
    bounded sample of the same workload: B = 8 of the benchmark's 32 utterances (L = 128 phonemes -> 768 frames each,
    1 572 864 samples = 71 s of audio), BLAS and OpenMP pools limited to CPU_THREADS threads."""
    from threadpoolctl import threadpool_limits
    from oracle import ref_numpy as orc
    L = PHONEMES
    rng = np.random.Generator(np.random.PCG64(1))
    ids = rng.integers(4, 131, size=(CPU_BATCH, L)).astype(np.int64)
    lens = np.full((CPU_BATCH,), L, np.int64)
    ac = orc.AcousticOracle(ac_state, cfg, stats)
    voc = orc.VocoderOracle(voc_state, cfg)
    with threadpool_limits(limits=CPU_THREADS):
        t0 = time.perf_counter()
        (mel, mel_post, dur), mel_lens = ac.inference(np.array([1]), ids, lens)
        wav = voc.forward(mel_post.transpose(0, 2, 1))
        dt = time.perf_counter() - t0
    samples = int(mel_lens.sum()) * cfg["audio"]["stft"]["hop_length"]
    assert wav.shape[0] * wav.shape[-1] == samples
    backend = "numpy + C/OpenMP conv1d" if orc._c_conv() else "numpy only"
    return {"value": samples / dt, "unit": "audio samples/s", "cores": CPU_THREADS, "kind": "port",
            "sample": f"oracle ({backend}), B={CPU_BATCH} L={L} -> T={int(mel_lens[0])} frames each ({samples} samples), {dt:.1f} s wall"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--workload", choices=("fixed", "mixed"), default="fixed",
                    help="fixed: the headline B=32 x L=128 batch; mixed: BASELINE config 3 (32 utterances of 40..200 phonemes, "
                         "padded to 200 -> T=1200), value = VALID samples/s")
    ap.add_argument("--precision", choices=("fp32", "bf16x3"), default="bf16x3",
                    help="vocoder arithmetic: exact fp32 MFMA, or split-precision bf16x3 MFMA (default; wav error ~1e-6)")
    ap.add_argument("--blocks", choices=("transformer", "conformer"), default="transformer",
                    help="encoder / decoder building block (reference model_config.yaml:8): the headline number is quoted on the default "
                         "'transformer' FFT blocks; 'conformer' times the same workload with Conformer blocks")
    args = ap.parse_args()

    import torch
    from e2e_tts_amd import config as cfgmod, packer, synth_weights as sw
    from e2e_tts_amd._lib import Engine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # Rehearsal of the N > 1 path on a one-GPU box (E2ETTS_BENCH_REHEARSAL=1): every rank uses GPU 0 and the collectives go
    # through gloo, since RCCL refuses two ranks on one device.  Never used for a reported number.
    rehearsal = os.environ.get("E2ETTS_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    cfg = cfgmod.default_config()
    cfg["models"]["fastspeech2"]["building_block"]["block_type"] = args.blocks
    stats = cfgmod.DEFAULT_STATS
    dims = cfgmod.dims_from_config(cfg, stats, n_speakers=4)
    hop = dims.hop_length
    ac_state = voc_state = None
    # rank 0 packs the weights; the blob travels to the other GPUs as ONE RCCL broadcast over xGMI
    if rank == 0:
        ac_state = sw.make_acoustic_state(cfg, stats, 4, seed=1234, mode="fixed", frames_per_phoneme=FRAMES_PER_PHONEME)
        voc_state = sw.make_vocoder_state(cfg, seed=4321)
        blob = torch.from_numpy(packer.pack(dims, ac_state, voc_state)).cuda()
        nbytes = torch.tensor([blob.numel()], dtype=torch.int64, device="cuda")
    else:
        nbytes = torch.zeros(1, dtype=torch.int64, device="cuda")
    if world > 1:
        dist.broadcast(nbytes, src=0)
        if rank != 0:
            blob = torch.empty(int(nbytes.item()), dtype=torch.uint8, device="cuda")
        t0 = time.perf_counter()
        dist.broadcast(blob, src=0)
        torch.cuda.synchronize()
        if rank == 0:
            log(f"[bench] weight blob {blob.numel() / 1e6:.1f} MB broadcast to {world} ranks in {(time.perf_counter() - t0) * 1e3:.1f} ms")
    eng = Engine(dims, device=local_rank)
    eng.load_weights(blob)
    eng.set_precision(args.precision)
    del blob

    B, L = args.batch, PHONEMES
    rng = np.random.Generator(np.random.PCG64(1000 + rank))
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
    ids = torch.from_numpy(ids_np).cuda()
    lens = torch.from_numpy(lens_np).cuda()
    spk = torch.tensor([1], dtype=torch.int64, device="cuda")
    T = L * FRAMES_PER_PHONEME
    valid_frames = int(lens_np.sum()) * FRAMES_PER_PHONEME
    pcm = torch.empty((B, T * hop), dtype=torch.int16, device="cuda")
    mel_lens = torch.empty((B,), dtype=torch.int64, device="cuda")

    def step():
        _, _, t = eng.synthesize(ids, lens, spk, out_pcm=pcm, out_mel_lens=mel_lens)
        return t

    for _ in range(args.warmup):
        t = step()
        assert t == T, (t, T)
    assert int(mel_lens.sum().item()) == valid_frames and int(mel_lens.max().item()) == T

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Per-kernel-class table from two fully bracketed steps BEFORE the timed region (two hipEventRecord per launch cost ~8 us;
    # ~155 launches per step = 2 % of a step).  The timed region then brackets the dominant class alone: its duration is still
    # measured live, with HIP events on the engine's stream, over exactly the timed steps.
    eng.profile_filter(None)
    eng.profile_enable(True)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    stats_all = eng.profile_read()
    eng.profile_enable(False)
    dom_name = max(stats_all, key=lambda st: st["ms"])["name"]
    eng.profile_filter(dom_name)
    eng.profile_enable(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    stats_k = eng.profile_read()
    eng.profile_enable(False)
    eng.profile_filter(None)
    el = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())

    samples_per_step = world * valid_frames * hop  # valid samples only (padding excluded)
    value = samples_per_step * args.steps / elapsed

    if rank == 0:
        # roofline of the dominant kernel: algorithmic FLOPs of its launches / sum of their HIP-event durations
        for st in sorted(stats_all, key=lambda st: -st["ms"]):
            tf = st["flops"] / (st["ms"] * 1e-3) / 1e12 if st["ms"] > 0 else 0.0
            gbs = st["bytes"] / (st["ms"] * 1e-3) / 1e9 if st["ms"] > 0 else 0.0
            log(f"[bench] {st['name']:<20} launches/step {st['launches'] / 2:7.1f}  ms/step {st['ms'] / 2:9.3f}  "
                f"avg {st['ms'] / max(st['launches'], 1) * 1e3:9.1f} us  {tf:7.2f} TFLOP/s  {gbs:8.1f} GB/s (algorithmic)")
        stats_k = [st for st in stats_k if st["name"] == dom_name]
        dom = stats_k[0]
        achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        kernel_ms = sum(st["ms"] for st in stats_all) / 2
        # HBM bytes per launch of that kernel come from separate rocprofv3 --pmc passes of this same command
        # (FETCH_SIZE x 2 + WRITE_SIZE, profiles/r*/pmc_summary.md); null when no summary covers the kernel.
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            traffic = pmc.get(dom["name"], {}).get("hbm_bytes_per_launch")
        except (OSError, ValueError):
            pass
        # fp32 kernel: algorithmic FLOPs against the fp32 MFMA peak.  Split-precision kernel: every algorithmic FLOP
        # costs three bf16 MFMA FLOPs, so its ceiling in algorithmic TFLOP/s is the bf16 dense peak / 3.
        x3 = dom["name"].startswith(("conv_x3", "resblock_pair", "x3 ", "pair "))
        peak = PEAK_BF16_TFLOPS / 3.0 if x3 else PEAK_FP32_TFLOPS
        roofline = {"bound": "mfma", "kernel": dom["name"], "achieved": round(achieved, 3), "peak": round(peak, 1),
                    "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
                    "peak_note": ("bf16 dense MFMA peak 2500 / 3 MFMAs per split-precision product (nominal = what the chip does on zeros; a bare "
                                  "MFMA loop on random operands measures 1724-1763 TFLOP/s at 1.65 GHz on this pool = 575-588 here: "
                                  "profiles/r1/mfma_peak.txt)" if x3 else "fp32 MFMA peak (v_mfma_f32_32x32x2_f32)"),
                    "algorithmic_bytes_per_launch": round(dom["bytes"] / dom["launches"]),
                    "avg_launch_us": round(dom["ms"] / dom["launches"] * 1e3, 2),
                    "launches_per_step": dom["launches"] / args.steps,
                    "all_conv_tflops": round(sum(st["flops"] for st in stats_all if st["name"].startswith(CONV_CLASSES)) /
                                             max(sum(st["ms"] for st in stats_all if st["name"].startswith(CONV_CLASSES)) * 1e-3, 1e-9) / 1e12, 3),
                    "kernel_ms_per_step": round(kernel_ms, 3)}
        # the same hot path with every convolution on the exact-fp32 MFMA (e2etts_set_precision fp32): reported beside the
        # default split-precision run so that both kernels' roofline fractions are on record
        fp32_mode = None
        if args.precision != "fp32":
            eng.set_precision("fp32")
            step()
            eng.profile_enable(True)
            torch.cuda.synchronize()
            tf0 = time.perf_counter()
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            tf = (time.perf_counter() - tf0) / 3
            st32 = sorted(eng.profile_read(), key=lambda s: -s["ms"])
            eng.profile_enable(False)
            eng.set_precision(args.precision)
            d32 = st32[0]
            a32 = d32["flops"] / (d32["ms"] * 1e-3) / 1e12
            fp32_mode = {"ms_per_step": round(tf * 1e3, 3), "samples_per_s": round(B * T * hop / tf), "kernel": d32["name"],
                         "achieved": round(a32, 3), "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": round(a32 / PEAK_FP32_TFLOPS, 4),
                         "avg_launch_us": round(d32["ms"] / d32["launches"] * 1e3, 2)}
            log(f"[bench] fp32 mode: {fp32_mode}")
        # host-inclusive variant (ids from host memory, PCM back to host memory): reported, never `value`
        ids_h, lens_h, spk_h = ids.cpu().numpy(), lens.cpu().numpy(), spk.cpu().numpy()
        pcm_h = np.empty((B, T * hop), np.int16)
        eng.synthesize(ids_h, lens_h, spk_h, out_pcm=pcm_h)
        th = time.perf_counter()
        for _ in range(3):
            eng.synthesize(ids_h, lens_h, spk_h, out_pcm=pcm_h)
        host_rate = valid_frames * hop * 3 / (time.perf_counter() - th)
        log(f"[bench] host-inclusive (pageable ids in, PCM out over PCIe): {host_rate:,.0f} samples/s per GPU")
        out = {
            "metric": "audio samples/sec (22.05 kHz, batch-32 per GPU, FastSpeech2 + HiFi-GAN inference)",
            "value": value, "unit": "audio samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "f32 (acoustic model) + bf16x3 split-precision (vocoder)",
            "data": "synthetic",
            "config": {"workload": (f"B={B}/GPU fixed-length L={L} phonemes x {FRAMES_PER_PHONEME} frames = T={T} frames "
                                    f"({T * hop} samples, {T * hop / dims.sample_rate:.2f} s) per utterance" if args.workload == "fixed" else
                                    f"B={B}/GPU mixed lengths 40..200 phonemes padded to L={L} (T={T}), {valid_frames} valid frames")
                                   + ("; default model_config (6+6 FFT blocks H=384, HiFi-GAN V1), random-init weights" if args.blocks == "transformer"
                                      else "; model_config with block_type=conformer (6+6 Conformer blocks H=384, 8 heads, k31; HiFi-GAN V1), random-init weights"),
                       "sample_rate": dims.sample_rate, "global_batch": world * B, "parallelism": f"utterance-sharded x{world}"},
            "real_time_factor": value / dims.sample_rate,
            "host_inclusive_samples_per_s_per_gpu": host_rate,
            "roofline": roofline,
            "fp32_mode": fp32_mode,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, stats, ac_state, voc_state)
            log(f"[bench] cpu_baseline: {out['cpu_baseline']}")
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
