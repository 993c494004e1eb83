"""GPU tests (-m gpu) of BASELINE config 5: 48 kHz-style HiFi-GAN (upsample 8x8x4x2 = hop 512), long-form mel stream
(>= 60 s of audio) through the streaming vocoder in bf16.  The reference ships no 48 kHz config: its
HifiGan class is config-driven, and so are the oracle and the engine: fixture hifigan_48k holds that class's own output for this
configuration (widths 64 and 512), and the >= 60 s run is then checked through size-independent properties (stream = one-shot)."""
import numpy as np
import pytest

from e2e_tts_amd import config as cfgmod, synth_weights as sw

pytestmark = pytest.mark.gpu


def cfg48(width):
    cfg = cfgmod.default_config()
    cfg["models"]["hifigan"].update(upsample_rates=[8, 8, 4, 2], upsample_kernel_sizes=[16, 16, 8, 4], upsample_initial_channel=width)
    cfg["audio"]["stft"]["hop_length"] = 512
    cfg["audio"]["signal"]["sampling_rate"] = 48000
    return cfg


def make_engine(cfg, seed):
    from e2e_tts_amd.models import HifiGan
    voc = sw.make_vocoder_state(cfg, seed=seed)
    v = HifiGan(cfg["models"]["hifigan"])
    v.load_state_dict(sw.to_torch(voc))
    return voc, v.eval().to(0).engine


def mean_l1(a, b):
    return float(np.abs(a.astype(np.float64) - b.astype(np.float64)).mean())


def test_48k_vocoder_matches_oracle_in_all_precisions():
    from oracle import ref_numpy as orc
    cfg = cfg48(64)
    voc, eng = make_engine(cfg, 31)
    assert eng.dims.hop_length == 512
    rng = np.random.Generator(np.random.PCG64(5))
    mel = rng.standard_normal((2, 90, 80)).astype(np.float32)
    ref = orc.VocoderOracle(voc, cfg).forward(mel.transpose(0, 2, 1))[:, 0]
    assert ref.shape == (2, 90 * 512)
    errs = {}
    # plain bf16: no farther from fp32 than the REFERENCE's own bf16 run of this width (5.9e-4, fixture hifigan_48k w64:
    # test_48k_vocoder_matches_reference_fixture pins it on the fixture's mel; this random mel has the same statistics)
    for prec, bar in (("fp32", 1e-5), ("bf16x3", 1e-5), ("bf16", 5.9e-4)):
        eng.set_precision(prec)
        wav, _ = eng.vocoder(mel, 2, 90, channels_first=False)
        errs[prec] = mean_l1(wav, ref)
        assert errs[prec] < bar, (prec, errs[prec])
    print("48k vocoder mean-L1 vs oracle:", errs)
    assert errs["fp32"] <= errs["bf16x3"] < errs["bf16"]


@pytest.mark.parametrize("tag", ["w64", "w512"])
def test_48k_vocoder_matches_reference_fixture(tag):
    """The 48 kHz generator against the reference's own HifiGan class (fixture hifigan_48k; V/generator.py:14-53 instantiated with
    upsample_rates [8, 8, 4, 2] / kernels [16, 16, 8, 4]) at widths 64 and 512.  fp32 and split precision meet the fp32 bar.  Plain
    bf16 -- config 5's arithmetic -- rounds every operand to 8 significant bits (relative 2^-9 per product term); it is pinned against
    the reference's own class run in bfloat16 (below)."""
    from conftest import load_golden
    g = load_golden("hifigan_48k")
    cfg = cfg48(int(g[f"{tag}.width"]))
    _, eng = make_engine(cfg, int(g[f"{tag}.weight_seed"]))
    mel, ref = g[f"{tag}.mel"], g[f"{tag}.wav"]
    B, T = mel.shape[0], mel.shape[1]
    errs = {}
    # Plain bf16 (round 3): the yardstick is the REFERENCE ITSELF run in bfloat16 -- its HifiGan class cast with
    # .bfloat16() on a bf16 mel (fixture keys wav_ref_bf16 / ref_bf16_mean_l1: 5.9e-4 at width 64, 8.6e-4 at width 512 against its own
    # fp32 output).  The engine's bf16 mode rounds the OPERANDS of every convolution to bf16 but keeps activations, residual sums and
    # accumulators in fp32, so it must land no farther from the reference's fp32 output than the reference's bf16 run does (factor 1.0,
    # stated; measured 3.3e-4 / 4.0e-4 = 0.56 x / 0.46 x), and within 1.5 x that distance of the reference's bf16 output itself.
    ref_bf16, ref_bf16_err = g[f"{tag}.wav_ref_bf16"], float(g[f"{tag}.ref_bf16_mean_l1"])
    assert abs(mean_l1(ref_bf16, ref) - ref_bf16_err) < 1e-9
    for prec, bar in (("fp32", 1e-5), ("bf16x3", 1e-5), ("bf16", 1.0 * ref_bf16_err)):
        eng.set_precision(prec)
        wav, pcm = eng.vocoder(mel, B, T, channels_first=False, pcm=True)
        errs[prec] = mean_l1(wav, ref)
        assert errs[prec] < bar, (prec, errs[prec])
        if prec != "bf16":
            lsb = np.abs(pcm.astype(np.int32) - (ref * np.float32(32768.0)).astype(np.int16).astype(np.int32)) <= 1
            assert lsb.mean() >= 0.999, (prec, lsb.mean())
        else:
            errs["bf16_vs_ref_bf16"] = mean_l1(wav, ref_bf16)
            assert errs["bf16_vs_ref_bf16"] < 1.5 * ref_bf16_err, errs
    print(f"48k {tag} mean-L1 vs the reference: {errs}; the reference in bf16 vs itself in fp32: {ref_bf16_err:.3e}")
    assert errs["fp32"] <= errs["bf16x3"] < errs["bf16"]


def test_streaming_equals_one_shot_bit_for_bit():
    cfg = cfg48(64)
    _, eng = make_engine(cfg, 32)
    rng = np.random.Generator(np.random.PCG64(6))
    T = 333
    mel = rng.standard_normal((3, T, 80)).astype(np.float32)
    for prec in ("bf16x3", "bf16"):
        eng.set_precision(prec)
        whole, whole_pcm = eng.vocoder(mel, 3, T, channels_first=False, pcm=True)
        for sizes in ([T], [1, 7, 40, 3, 100, 2, 180], [16] * 20 + [13], [200, 133]):
            assert sum(sizes) == T
            pieces, pos = [], 0
            chunks = []
            for n in sizes:
                chunks.append(np.ascontiguousarray(mel[:, pos:pos + n]))
                pos += n
            out = np.concatenate(list(eng.vocoder_stream(chunks, 3)), axis=1)
            assert out.shape == whole.shape
            np.testing.assert_array_equal(out, whole)
        pcm = np.concatenate(list(eng.vocoder_stream([np.ascontiguousarray(mel[:, :150]), np.ascontiguousarray(mel[:, 150:])], 3, want_pcm=True)), axis=1)
        np.testing.assert_array_equal(pcm, whole_pcm)
    assert 8 <= eng.stream_halo <= 24
    # one-shot calls between the steps of an open stream do not disturb it (its carried context lives in buffers of its own)
    other = rng.standard_normal((2, 50, 80)).astype(np.float32)
    pieces = []
    for piece in eng.vocoder_stream([np.ascontiguousarray(mel[:, i:i + 37]) for i in range(0, T, 37)], 3):
        pieces.append(piece)
        eng.vocoder(other, 2, 50, channels_first=False)
    np.testing.assert_array_equal(np.concatenate(pieces, axis=1), whole)


def test_stream_keeps_two_chunks_in_flight():
    """The C ABI's own orders (include/e2etts.h): push / fetch alternating as before round 4, and push(i + 1) before fetch(i); the
    third unfetched push is refused; chunks in device memory; the resident one-shot result stays what it was."""
    import ctypes as C
    import torch
    from e2e_tts_amd._lib import _addr
    cfg = cfg48(64)
    _, eng = make_engine(cfg, 32)
    lib, h = eng.lib, eng._h
    rng = np.random.Generator(np.random.PCG64(16))
    T, hop = 200, 512
    mel = rng.standard_normal((2, T, 80)).astype(np.float32)
    eng.set_precision("bf16")
    other = rng.standard_normal((1, 40, 80)).astype(np.float32)
    other_wav, _ = eng.vocoder(other, 1, 40, channels_first=False)
    whole, whole_pcm = eng.vocoder(mel, 2, T, channels_first=False, pcm=True)
    chunks = [np.ascontiguousarray(mel[:, i:i + 48]) for i in range(0, T, 48)]

    def push(c, last):
        n_emit = C.c_int(0)
        rc = lib.e2etts_vocoder_stream_push(h, _addr(c), int(c.shape[1]), int(last), C.byref(n_emit))
        return rc, n_emit.value

    def fetch(n_emit):
        wav = np.empty((2, n_emit * hop), np.float32)
        pcm = np.empty((2, n_emit * hop), np.int16)
        assert lib.e2etts_vocoder_stream_fetch(h, _addr(wav), _addr(pcm), wav.size) == 0, lib.e2etts_last_error(h).decode()
        return wav, pcm

    # alternating
    assert lib.e2etts_vocoder_stream_begin(h, 2) >= 0
    got = []
    for i, c in enumerate(chunks):
        rc, n = push(c, i == len(chunks) - 1)
        assert rc == 0 and n > 0
        got.append(fetch(n))
    np.testing.assert_array_equal(np.concatenate([g[0] for g in got], axis=1), whole)
    np.testing.assert_array_equal(np.concatenate([g[1] for g in got], axis=1), whole_pcm)
    assert lib.e2etts_vocoder_stream_fetch(h, _addr(np.empty((2, 1 << 20), np.float32)), None, 2 << 20) != 0   # nothing left

    # two in flight, from device memory; the caller's buffers are overwritten as soon as pageable ones may be.  Nothing the slots'
    # workspaces still hold from the first stream may matter
    eng.poison_workspace()
    again, _ = eng.vocoder(mel, 2, T, channels_first=False)
    np.testing.assert_array_equal(again, whole)
    assert lib.e2etts_vocoder_stream_begin(h, 2) >= 0
    got, waiting = [], []
    for i, c in enumerate(chunks):
        src = torch.from_numpy(c).to(0) if i % 2 else c.copy()
        rc, n = push(src, i == len(chunks) - 1)
        assert rc == 0 and n > 0
        if i % 2 == 0:
            src[:] = 7.0   # pageable: consumed when push returned
        waiting.append((n, src))
        if len(waiting) == 2:
            if i == 1:
                rc3, _ = push(chunks[2], False)
                assert rc3 != 0 and "await" in lib.e2etts_last_error(h).decode()
            got.append(fetch(waiting.pop(0)[0]))
    while waiting:
        got.append(fetch(waiting.pop(0)[0]))
    np.testing.assert_array_equal(np.concatenate([g[0] for g in got], axis=1), whole)
    np.testing.assert_array_equal(np.concatenate([g[1] for g in got], axis=1), whole_pcm)
    # the stream has its own output slots: the last one-shot result is still the resident one
    res = np.empty_like(whole)
    assert lib.e2etts_fetch_wav(h, _addr(res), res.size) == 0
    np.testing.assert_array_equal(res, whole)
    assert other_wav.shape == (1, 40 * hop)


def test_weight_reload_closes_an_open_stream():
    """New weights while chunks are in flight: the slots' kernels finish first, the stream is closed, and a fresh stream gives the new
    weights' result."""
    import ctypes as C
    from e2e_tts_amd.models import HifiGan
    from e2e_tts_amd._lib import _addr
    cfg = cfg48(64)
    v = HifiGan(cfg["models"]["hifigan"])
    v.load_state_dict(sw.to_torch(sw.make_vocoder_state(cfg, seed=41)))
    eng = v.eval().to(0).engine
    lib, h = eng.lib, eng._h
    eng.set_precision("bf16")
    rng = np.random.Generator(np.random.PCG64(17))
    mel = rng.standard_normal((1, 96, 80)).astype(np.float32)
    assert lib.e2etts_vocoder_stream_begin(h, 1) >= 0
    n_emit = C.c_int(0)
    assert lib.e2etts_vocoder_stream_push(h, _addr(mel[:, :48].copy()), 48, 0, C.byref(n_emit)) == 0 and n_emit.value > 0
    v.load_state_dict(sw.to_torch(sw.make_vocoder_state(cfg, seed=42)))   # repacks and reloads into the same engine
    assert v.engine is eng
    rc = lib.e2etts_vocoder_stream_push(h, _addr(mel[:, 48:].copy()), 48, 1, C.byref(n_emit))
    assert rc != 0 and "no open vocoder stream" in lib.e2etts_last_error(h).decode()
    eng.set_precision("bf16")
    whole, _ = eng.vocoder(mel, 1, 96, channels_first=False)
    out = np.concatenate(list(eng.vocoder_stream([mel[:, :48].copy(), mel[:, 48:].copy()], 1)), axis=1)
    np.testing.assert_array_equal(out, whole)


def test_long_form_60s_stream_bf16():
    """>= 60 s of 48 kHz audio (5 632 frames x 512) from one utterance, default-width generator, plain bf16, chunks of 512
    frames; checked against the one-shot run (bit-exact) and against the split-precision run (stated bf16 tolerance)."""
    cfg = cfg48(512)
    _, eng = make_engine(cfg, 33)
    rng = np.random.Generator(np.random.PCG64(7))
    T = 5632
    mel = (0.7 * rng.standard_normal((1, T, 80))).astype(np.float32)
    eng.set_precision("bf16")
    chunks = [np.ascontiguousarray(mel[:, i:i + 512]) for i in range(0, T, 512)]
    out = np.concatenate(list(eng.vocoder_stream(chunks, 1)), axis=1)
    assert out.shape == (1, T * 512) and out.shape[1] / 48000 >= 60.0
    assert np.isfinite(out).all() and np.abs(out).max() <= 1.0
    whole, _ = eng.vocoder(mel, 1, T, channels_first=False)
    np.testing.assert_array_equal(out, whole)
    eng.set_precision("bf16x3")
    exact, _ = eng.vocoder(mel, 1, T, channels_first=False)
    err = mean_l1(out, exact)
    print(f"long-form bf16 vs bf16x3: mean-L1 {err:.3e}")
    # the reference's own bf16 run of this generator (width 512) lies 8.6e-4 from its fp32 output (fixture hifigan_48k w512,
    # test_48k_vocoder_matches_reference_fixture); the engine's bf16 mode must stay inside that distance.  Measured 3.2e-4.
    assert err < 8.6e-4


_BF16_CHILD = r"""
import hashlib, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
from e2e_tts_amd import config as cfgmod, synth_weights as sw
from e2e_tts_amd.models import HifiGan
cfg = cfgmod.default_config()
cfg["models"]["hifigan"].update(upsample_rates=[8, 8, 4, 2], upsample_kernel_sizes=[16, 16, 8, 4], upsample_initial_channel=512)
v = HifiGan(cfg["models"]["hifigan"])
v.load_state_dict(sw.to_torch(sw.make_vocoder_state(cfg, seed=35)))
eng = v.eval().to(0).engine
eng.set_precision("bf16")
T = 1100
mel = (0.7 * np.random.Generator(np.random.PCG64(9)).standard_normal((2, T, 80))).astype(np.float32)
_, whole = eng.vocoder(mel, 2, T, channels_first=False, pcm=True)
chunks = [np.ascontiguousarray(mel[:, i:i + 300]) for i in range(0, T, 300)]
stream = np.concatenate(list(eng.vocoder_stream(chunks, 2, want_pcm=True)), axis=1)
print("SHA", hashlib.sha256(whole.tobytes()).hexdigest(), hashlib.sha256(stream.tobytes()).hexdigest(), int(np.abs(whole.astype(np.int32)).max()))
"""


def test_plain_bf16_kernels_of_round_4_give_round_3s_bits(tmp_path):
    """Round 4 moved the plain-bf16 vocoder onto conv_bf16.hip (whole-slab convolutions with bf16 hand-over, fused pairs, whole ResBlocks,
    grouped launches, joins folded into the next layer's staging).  Every one of them claims the arithmetic of the kernels it replaces,
    term for term: the same utterances, one call and streamed, through the default path and through a child process that switches all
    of it off (E2ETTS_BCONV / BPAIR / BRB / VOC_GROUP / VOC_DEFER_JOIN = 0: round 3's launches) must give the same PCM bit for bit."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    script = tmp_path / "bf16_child.py"
    script.write_text(_BF16_CHILD)
    shas = []
    for off in (False, True):
        env = dict(os.environ)
        if off:
            env.update(E2ETTS_BCONV="0", E2ETTS_BPAIR="0", E2ETTS_BRB="0", E2ETTS_VOC_GROUP="0", E2ETTS_VOC_DEFER_JOIN="0")
        r = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0 and "SHA" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("SHA")][0].split()
        assert line[1] == line[2], "stream != one call"
        assert int(line[3]) > 1000
        shas.append(line[1])
    assert shas[0] == shas[1]
