#!/usr/bin/env python3
"""BASELINE config 5 timing: 48 kHz-style HiFi-GAN (upsample 8x8x4x2, hop 512), one utterance of >= 60 s (5 632 frames), mel pushed
through the streaming vocoder in chunks.  Prints audio-seconds per wall-second for each arithmetic mode.  (Correctness of this
path: tests/test_gpu_longform.py.)

    python tools/longform_bench.py [chunk_frames] [modes, comma separated: bf16,bf16x3,fp32] [timed passes]

With E2ETTS_PROFILE_FINE=1 in the environment one more pass per mode runs under the engine's event profile and prints the per-layer table."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from e2e_tts_amd import config as cfgmod, synth_weights as sw  # noqa: E402
from e2e_tts_amd.models import HifiGan  # noqa: E402


def main():
    chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["bf16", "bf16x3", "fp32"]
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    cfg = cfgmod.default_config()
    cfg["models"]["hifigan"].update(upsample_rates=[8, 8, 4, 2], upsample_kernel_sizes=[16, 16, 8, 4], upsample_initial_channel=512)
    v = HifiGan(cfg["models"]["hifigan"])
    v.load_state_dict(sw.to_torch(sw.make_vocoder_state(cfg, seed=33)))
    eng = v.eval().to(0).engine
    T = 5632
    mel = np.random.Generator(np.random.PCG64(7)).standard_normal((1, T, 80)).astype(np.float32)
    chunks = [np.ascontiguousarray(mel[:, i:i + chunk]) for i in range(0, T, chunk)]
    for prec in modes:
        eng.set_precision(prec)
        n = sum(p.shape[1] for p in eng.vocoder_stream(chunks, 1, want_pcm=True))  # warm-up
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            n = sum(p.shape[1] for p in eng.vocoder_stream(chunks, 1, want_pcm=True))
            ts.append(time.perf_counter() - t0)
        dt = sorted(ts)[len(ts) // 2]
        print(f"{prec:7s} chunk {chunk:4d} frames: {n} samples = {n / 48000:.1f} s of 48 kHz audio in {dt * 1e3:.2f} ms (median of {reps}; min {min(ts) * 1e3:.2f}) -> "
              f"{n / 48000 / dt:.0f} x real-time, {n / dt / 1e6:.1f} M samples/s (PCM fetched to the host per chunk)", flush=True)
        if os.environ.get("E2ETTS_PROFILE_FINE"):
            eng.profile_filter(None)
            eng.profile_enable(True)
            sum(p.shape[1] for p in eng.vocoder_stream(chunks, 1, want_pcm=True))
            eng.sync()
            st = eng.profile_read()
            eng.profile_enable(False)
            tot = sum(s["ms"] for s in st)
            print(f"# {prec} chunk {chunk}: per-layer classes of one pass (one kernel at a time: {tot:.2f} ms of kernel time)")
            for s in sorted(st, key=lambda s: -s["ms"]):
                print(f"#   {s['name']:40s} {s['launches']:5d} launches {s['ms']:8.3f} ms  {s['ms'] / max(s['launches'], 1) * 1e3:8.1f} us each  "
                      f"{s['flops'] / max(s['ms'], 1e-9) / 1e9:8.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
