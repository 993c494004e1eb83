#!/usr/bin/env python3
"""BASELINE config 5 timing: 48 kHz-style HiFi-GAN (upsample 8x8x4x2, hop 512), one utterance of >= 60 s (5 632 frames), mel pushed
through the streaming vocoder in chunks.  Prints audio-seconds per wall-second for each arithmetic mode.  (Correctness of this
path: tests/test_gpu_longform.py.)   python tools/longform_bench.py [chunk_frames]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from e2e_tts_amd import config as cfgmod, synth_weights as sw  # noqa: E402
from e2e_tts_amd.models import HifiGan  # noqa: E402


def main():
    chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    cfg = cfgmod.default_config()
    cfg["models"]["hifigan"].update(upsample_rates=[8, 8, 4, 2], upsample_kernel_sizes=[16, 16, 8, 4], upsample_initial_channel=512)
    v = HifiGan(cfg["models"]["hifigan"])
    v.load_state_dict(sw.to_torch(sw.make_vocoder_state(cfg, seed=33)))
    eng = v.eval().to(0).engine
    T = 5632
    mel = np.random.Generator(np.random.PCG64(7)).standard_normal((1, T, 80)).astype(np.float32)
    chunks = [np.ascontiguousarray(mel[:, i:i + chunk]) for i in range(0, T, chunk)]
    for prec in ("bf16", "bf16x3", "fp32"):
        eng.set_precision(prec)
        n = sum(p.shape[1] for p in eng.vocoder_stream(chunks, 1, want_pcm=True))  # warm-up
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            n = sum(p.shape[1] for p in eng.vocoder_stream(chunks, 1, want_pcm=True))
        dt = (time.perf_counter() - t0) / reps
        print(f"{prec:7s} chunk {chunk:4d} frames: {n} samples = {n / 48000:.1f} s of 48 kHz audio in {dt * 1e3:.1f} ms -> {n / 48000 / dt:.0f} x real-time, "
              f"{n / dt / 1e6:.1f} M samples/s (PCM fetched to the host per chunk)")


if __name__ == "__main__":
    main()
