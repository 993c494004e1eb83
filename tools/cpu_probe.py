import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from oracle import ref_numpy as orc
from e2e_tts_amd import config as cfgmod, synth_weights as sw
cfg = cfgmod.default_config(); stats = cfgmod.DEFAULT_STATS
ac = sw.make_acoustic_state(cfg, stats, 4, seed=1234, mode="fixed"); voc = sw.make_vocoder_state(cfg, seed=4321)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 48
ids = np.random.default_rng(1).integers(4, 131, size=(1, L)).astype(np.int64); lens = np.array([L], np.int64)
A = orc.AcousticOracle(ac, cfg, stats); V = orc.VocoderOracle(voc, cfg)
print("C backend:", bool(orc._c_conv()), "OMP_NUM_THREADS", os.environ.get("OMP_NUM_THREADS"), "cpus", len(os.sched_getaffinity(0)))
from threadpoolctl import threadpool_limits
NT = int(sys.argv[2]) if len(sys.argv) > 2 else 32
lim = threadpool_limits(limits=NT)
print("threadpoolctl limit", NT)
for rep in range(2):
    t0 = time.perf_counter()
    (mel, mel_post, dur), ml = A.inference(np.array([1]), ids, lens)
    t1 = time.perf_counter()
    wav = V.forward(mel_post.transpose(0, 2, 1))
    t2 = time.perf_counter()
    print(f"acoustic {t1-t0:.2f}s vocoder {t2-t1:.2f}s -> {int(ml.sum())*256/(t2-t0):.0f} samples/s", flush=True)
