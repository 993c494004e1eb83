"""GPU parity (-m gpu): the HIP path, called through the C ABI, against (a) the fixtures generated from the
reference and (b) the numpy oracle on the same inputs.

Bars: duration_rounded, mel_lens, pitch / energy bucket indices bit-exact; mel_post and wav
mean-L1 <= 1e-4 (fp32); int16 PCM within 1 LSB on >= 99.9 % of samples.
"""
import numpy as np
import pytest

from conftest import load_golden, states_for
from e2e_tts_amd import config as cfgmod

pytestmark = pytest.mark.gpu

MEL_L1 = 1e-4
WAV_L1 = 1e-4
PRECISIONS = ("fp32", "bf16x3")  # vocoder arithmetic: both must meet the same bars

_ENGINES = {}


def engine_for(g, name):
    from e2e_tts_amd.runtime import engine_from_states
    cfg, ac, voc = states_for(g, name)
    key = (name.startswith("tiny"), "_cf_" in name, "_hv_" in name, "_nouv_" in name, "_plog_" in name, "_lpad_" in name, "_frame_" in name, "_pframe_" in name, "_eframe_" in name, str(g["mode"]), tuple(int(x) for x in g["weight_seeds"]))
    if key not in _ENGINES:
        _ENGINES[key] = engine_from_states(cfg, cfgmod.DEFAULT_STATS, ac, voc, device=0)
    return cfg, _ENGINES[key]


def mean_l1(a, b):
    return float(np.abs(a.astype(np.float64) - b.astype(np.float64)).mean())


def run_acoustic(eng, g, precision="bf16x3"):
    eng.set_precision(precision)
    d, p, e = (float(x) for x in g["controls"])
    spk = np.array([int(g["speaker"])], np.int64)
    r = eng.acoustic(g["ids"], g["lens"], spk, d, p, e,
                     want=("dur", "mel_lens", "pitch_idx", "energy_idx", "log_d", "pitch_pred", "energy_pred"))
    mel, mel_post = eng.fetch_mel(r["B"], r["T"])
    return r, mel, mel_post


def check_discrete(r, g):
    np.testing.assert_array_equal(r["dur"], g["dur"])
    np.testing.assert_array_equal(r["mel_lens"], g["mel_lens"])
    np.testing.assert_array_equal(r["pitch_idx"], g["pitch_idx"])
    np.testing.assert_array_equal(r["energy_idx"], g["energy_idx"])


# *_cf_*: the same model with Conformer blocks (building_block.block_type = "conformer", reference U/blocks/conformer.py);
# tiny_cf_long runs past max_seq_len in the encoder and the decoder (regenerated position tables in every attention module)
@pytest.mark.parametrize("name", ["tiny_b3", "tiny_long", "tiny_ctl", "tiny_b1", "tiny_cf_b3", "tiny_cf_long", "tiny_hv_b3", "tiny_cf_hv_b3",
                                  "tiny_nouv_b3", "tiny_plog_b3", "tiny_lpad_b3", "tiny_frame_b3", "tiny_pframe_b3", "tiny_eframe_b3", "tiny_cf_frame_b3"])
def test_tiny_model_full_trace(name):
    g = load_golden(name)
    cfg, eng = engine_for(g, name)
    B, L = g["ids"].shape
    H = cfg["models"]["fastspeech2"]["encoder_hidden"]
    r, mel, mel_post = run_acoustic(eng, g, "bf16x3")   # split-precision decoder: same bars
    check_discrete(r, g)
    assert mean_l1(eng.fetch_tap("dec_out", (B, r["T"], H)), g["dec_out"]) < 1e-5
    assert mean_l1(mel_post, g["mel_post"]) < MEL_L1 / 10
    r, mel, mel_post = run_acoustic(eng, g, "fp32")
    check_discrete(r, g)
    assert mean_l1(r["log_d"], g["log_d"]) < 1e-5
    assert mean_l1(r["pitch_pred"], g["pitch_pred"]) < 1e-5
    assert mean_l1(r["energy_pred"], g["energy_pred"]) < 1e-5
    assert mean_l1(eng.fetch_tap("enc_out", (B, L, H)), g["enc_out"]) < 1e-5
    assert mean_l1(eng.fetch_tap("dec_out", (B, r["T"], H)), g["dec_out"]) < 1e-5
    assert mel.shape == g["mel"].shape
    assert mean_l1(mel, g["mel"]) < MEL_L1 / 10
    assert mean_l1(mel_post, g["mel_post"]) < MEL_L1 / 10
    # vocoder on the engine's own resident mel_post (the production flow) and on the reference's mel,
    # in both arithmetic modes (exact fp32 MFMA, and the default split-precision bf16x3 MFMA)
    for prec in PRECISIONS:
        eng.set_precision(prec)
        wav, pcm = eng.vocoder(None, r["B"], r["T"], wav=True, pcm=True)
        assert wav.shape == g["wav"].shape
        assert mean_l1(wav, g["wav"]) < WAV_L1 / 10, prec
        wav2, _ = eng.vocoder(np.ascontiguousarray(g["mel_post"].transpose(0, 2, 1)), r["B"], r["T"])
        assert mean_l1(wav2, g["wav"]) < WAV_L1 / 10, prec
        ref_pcm = (g["wav"] * 32768.0).astype(np.int16)
        close = np.abs(pcm.astype(np.int32) - ref_pcm.astype(np.int32)) <= 1
        assert close.mean() >= 0.999, prec


@pytest.mark.parametrize("name", ["c1_plumbing", "full_b3", "c2_latency", "full_cf_b2", "full_frame_b2"])
def test_default_model(name):
    g = load_golden(name)
    cfg, eng = engine_for(g, name)
    for prec in PRECISIONS:
        r, mel, mel_post = run_acoustic(eng, g, prec)
        check_discrete(r, g)
        print(f"{name} {prec}: mel_post mean-L1 {mean_l1(mel_post, g['mel_post']):.3e}")
        assert mean_l1(mel, g["mel"]) < MEL_L1, prec
        assert mean_l1(mel_post, g["mel_post"]) < MEL_L1, prec
    s = int(g["wav_stride"])
    hop = cfg["audio"]["stft"]["hop_length"]
    for prec in PRECISIONS:
        eng.set_precision(prec)
        wav, _ = eng.vocoder(None, r["B"], r["T"])
        err = mean_l1(wav[:, ::s], g["wav_strided"])
        print(f"{name} {prec}: wav mean-L1 {err:.3e}")
        assert err < WAV_L1, prec
        assert mean_l1(wav[:, :2048], g["wav_head"]) < WAV_L1, prec
        for b, n in enumerate(g["mel_lens"] * hop):
            assert abs(np.abs(wav[b, :n].astype(np.float64)).sum() - g["wav_abs_sum"][b]) < WAV_L1 * n, prec


@pytest.mark.parametrize("name", ["c3_mixed", "full_long"])
def test_c3_mixed_batch32(name):
    """c3_mixed -- BASELINE config 3: B = 32 mixed lengths 40..200 (T = 1200 > max_seq_len: regenerated position table).
    full_long -- 512 phonemes next to 37: T = 3 072 frames, 12 key segments in the attention kernels, position table for 3 x max_seq_len."""
    g = load_golden(name)
    cfg, eng = engine_for(g, name)
    sel = g["sel"]
    fs = int(g["mel_frame_stride"])
    for prec in ("fp32", "bf16x3"):   # ends on the default, whose resident mel_post feeds the vocoder checks below
        r, mel, mel_post = run_acoustic(eng, g, prec)
        check_discrete(r, g)
        print(f"{name} {prec}: mel_post mean-L1 {mean_l1(mel_post[sel][:, ::fs], g['mel_post_sel']):.3e}")
        assert mean_l1(mel_post[sel][:, ::fs], g["mel_post_sel"]) < MEL_L1, prec
        for b, n in enumerate(g["mel_lens"]):
            assert abs(np.abs(mel_post[b, :n].astype(np.float64)).sum() - g["mel_post_abs_sum"][b]) < MEL_L1 * n * 80, prec
    ws = int(g["wav_stride"])
    hop = cfg["audio"]["stft"]["hop_length"]
    for prec in PRECISIONS:
        eng.set_precision(prec)
        wav, _ = eng.vocoder(None, r["B"], r["T"])
        err = mean_l1(wav[sel][:, ::ws], g["wav_strided_sel"])
        print(f"{name} {prec}: wav mean-L1 {err:.3e}")
        assert err < WAV_L1, prec
        for b, n in enumerate(g["mel_lens"] * hop):
            assert abs(np.abs(wav[b, :n].astype(np.float64)).sum() - g["wav_abs_sum"][b]) < WAV_L1 * n, prec


def test_bench_b32_headline_workload_against_reference():
    """The workload bench.py times (B = 32 x L = 128 phonemes, "fixed" weights 1234 / 4321 -> T = 768), pinned by the reference itself:
    fixture bench_b32 (oracle/make_goldens.py: case_bench_b32, c3_mixed's digest scheme), in fp32 and in split precision, through
    synthesize() -- the call bench.py makes.  Row 0 carries the ids of c2_latency: the same utterance alone (B = 1) must give the
    same PCM bit for bit, and both must agree with what the reference produced for it at B = 32 and at B = 1."""
    g = load_golden("bench_b32")
    c2 = load_golden("c2_latency")
    cfg, eng = engine_for(g, "bench_b32")
    sel, fs, ws = g["sel"], int(g["mel_frame_stride"]), int(g["wav_stride"])
    hop = cfg["audio"]["stft"]["hop_length"]
    spk = np.array([int(g["speaker"])], np.int64)
    for prec in PRECISIONS:
        r, mel, mel_post = run_acoustic(eng, g, prec)
        check_discrete(r, g)
        assert r["T"] == 768
        e_mel = mean_l1(mel_post[sel][:, ::fs], g["mel_post_sel"])
        assert e_mel < MEL_L1, prec
        for b, n in enumerate(g["mel_lens"]):
            assert abs(np.abs(mel_post[b, :n].astype(np.float64)).sum() - g["mel_post_abs_sum"][b]) < MEL_L1 * n * 80, prec
        wav, pcm_v = eng.vocoder(None, r["B"], r["T"], pcm=True)
        e_wav = mean_l1(wav[sel][:, ::ws], g["wav_strided_sel"])
        assert e_wav < WAV_L1, prec
        for b, n in enumerate(g["mel_lens"] * hop):
            assert abs(np.abs(wav[b, :n].astype(np.float64)).sum() - g["wav_abs_sum"][b]) < WAV_L1 * n, prec
        lsb = np.abs(pcm_v[sel][:, ::ws].astype(np.int32) - g["pcm_strided_sel"].astype(np.int32)) <= 1
        print(f"bench_b32 {prec}: mel_post mean-L1 {e_mel:.3e} wav mean-L1 {e_wav:.3e} PCM within 1 LSB {lsb.mean():.5f}")
        assert lsb.mean() >= 0.999, prec
        # the end-to-end call: same PCM as acoustic() + vocoder()
        pcm, mel_lens, T = eng.synthesize(g["ids"], g["lens"], spk)
        np.testing.assert_array_equal(mel_lens, g["mel_lens"])
        np.testing.assert_array_equal(pcm, pcm_v)
        # the c2_latency utterance alone: bit-identical to its row of the batch, and within the bars of the reference's B = 1 run
        one, ml1, T1 = eng.synthesize(c2["ids"], c2["lens"], spk)
        assert T1 == T
        np.testing.assert_array_equal(one[0], pcm[0])
        ref1 = (c2["wav_strided"][0] * np.float32(32768.0)).astype(np.int16)
        assert (np.abs(one[0, ::int(c2["wav_stride"])].astype(np.int32) - ref1.astype(np.int32)) <= 1).mean() >= 0.999, prec


def test_vocoder_stage_fixture():
    g = load_golden("voc_micro_tiny")
    from e2e_tts_amd import synth_weights as sw
    from e2e_tts_amd.runtime import engine_from_states
    cfg = cfgmod.tiny_config()
    eng = engine_from_states(cfg, cfgmod.DEFAULT_STATS, sw.make_acoustic_state(cfg, cfgmod.DEFAULT_STATS, 4, mode="varied"),
                             sw.make_vocoder_state(cfg, seed=4321))
    B, _, T = g["mel"].shape
    for prec, bar in (("fp32", 1e-6), ("bf16x3", 1e-5)):
        eng.set_precision(prec)
        wav, _ = eng.vocoder(g["mel"], B, T)
        assert mean_l1(wav, g["wav"][:, 0]) < bar, prec
        wav2, _ = eng.vocoder(np.ascontiguousarray(g["mel"].transpose(0, 2, 1)), B, T, channels_first=False)
        np.testing.assert_array_equal(wav, wav2)
    with pytest.raises(KeyError):
        eng.set_precision("fp8")


def test_oracle_agrees_on_fresh_inputs():
    """HIP vs the numpy oracle on inputs no fixture holds (new ids, per-utterance speakers)."""
    from e2e_tts_amd import synth_weights as sw
    from e2e_tts_amd.packer import variance_position_table
    from e2e_tts_amd.runtime import engine_from_states
    from oracle import ref_numpy as orc
    cfg = cfgmod.tiny_config()
    stats = cfgmod.DEFAULT_STATS
    ac = sw.make_acoustic_state(cfg, stats, 4, seed=99, mode="varied")
    voc = sw.make_vocoder_state(cfg, seed=98)
    eng = engine_from_states(cfg, stats, ac, voc)
    rng = np.random.Generator(np.random.PCG64(2024))
    lens = np.array([31, 8, 19, 27, 2], np.int64)
    ids = np.zeros((5, 31), np.int64)
    for b, n in enumerate(lens):
        ids[b, :n] = rng.integers(4, 131, n)
    spk = np.array([0, 3, 1, 2, 2], np.int64)
    o = orc.AcousticOracle(ac, cfg, stats, var_pos_table=variance_position_table(4096, 64))
    (omel, omel_post, odur), omel_lens = o.inference(spk, ids, lens)
    r = eng.acoustic(ids, lens, spk, want=("dur", "mel_lens", "pitch_idx", "energy_idx"))
    np.testing.assert_array_equal(r["dur"], odur)
    np.testing.assert_array_equal(r["mel_lens"], omel_lens)
    np.testing.assert_array_equal(r["pitch_idx"], o.trace["pitch_idx"])
    np.testing.assert_array_equal(r["energy_idx"], o.trace["energy_idx"])
    mel, mel_post = eng.fetch_mel(r["B"], r["T"])
    assert mean_l1(mel_post, omel_post) < 1e-5
    wav, _ = eng.vocoder(None, r["B"], r["T"])
    owav = orc.VocoderOracle(voc, cfg).forward(omel_post.transpose(0, 2, 1))[:, 0]
    assert mean_l1(wav, owav) < 1e-5


def test_error_paths():
    from e2e_tts_amd import synth_weights as sw
    from e2e_tts_amd.runtime import engine_from_states
    cfg = cfgmod.tiny_config()
    eng = engine_from_states(cfg, cfgmod.DEFAULT_STATS, sw.make_acoustic_state(cfg, cfgmod.DEFAULT_STATS, 4, mode="varied"),
                             sw.make_vocoder_state(cfg))
    ids = np.full((1, 4), 5, np.int64)
    with pytest.raises(ValueError):
        eng.acoustic(ids, np.array([5], np.int64), np.array([0], np.int64))       # len > L
    with pytest.raises(ValueError):
        eng.acoustic(ids, np.array([4], np.int64), np.array([9], np.int64))       # unknown speaker
    with pytest.raises(ValueError):
        eng.acoustic(np.full((1, 4), 500, np.int64), np.array([4], np.int64), np.array([0], np.int64))  # unknown symbol
    with pytest.raises(ValueError):
        eng.load_weights(np.zeros(64, np.uint8))                                  # not a blob


@pytest.mark.parametrize("name", ["tiny_b3", "tiny_cf_b3"])
def test_steady_state_allocates_nothing_and_repeats_exactly(name):
    """A serving loop: after the first call of a shape the workspace is reused (device_bytes constant) and every repetition
    returns the same PCM bit for bit -- the whole padded array without ragged compute, every valid sample with it (rows past an
    utterance's end are skipped there and hold whatever the workspace held)."""
    g = load_golden(name)
    cfg, eng = engine_for(g, name)
    eng.set_precision("bf16x3")
    hop = cfg["audio"]["stft"]["hop_length"]
    spk = np.array([int(g["speaker"])], np.int64)
    try:
        for ragged in (False, True):
            eng.set_ragged(ragged)
            first, ml, T = eng.synthesize(g["ids"], g["lens"], spk)
            eng.sync()
            bytes0 = eng.device_bytes()
            for _ in range(8):
                again, ml2, T2 = eng.synthesize(g["ids"], g["lens"], spk)
                assert T2 == T and (ml2 == ml).all()
                if ragged:
                    for b, n in enumerate(ml * hop):
                        np.testing.assert_array_equal(again[b, :n], first[b, :n])
                else:
                    np.testing.assert_array_equal(again, first)
            eng.sync()
            assert eng.device_bytes() == bytes0
    finally:
        eng.set_ragged(True)


def test_changing_shapes_on_one_engine_match_fresh_engines():
    """A long-lived engine sees batches of changing size (its workspace grows, buffers are re-allocated, ragged limits change):
    every result must be the one a fresh engine gives for that batch alone.  Also: a per-utterance speaker list equal to the
    broadcast id gives the same bits."""
    from e2e_tts_amd import synth_weights as sw
    from e2e_tts_amd.runtime import engine_from_states
    cfg = cfgmod.tiny_config()
    ac = sw.make_acoustic_state(cfg, cfgmod.DEFAULT_STATS, 4, seed=41, mode="varied")
    voc = sw.make_vocoder_state(cfg, seed=42)
    hop = cfg["audio"]["stft"]["hop_length"]
    rng = np.random.Generator(np.random.PCG64(43))
    shapes = [(2, 10), (5, 40), (1, 5), (3, 64), (2, 10), (7, 33), (1, 70)]   # (B, L); 64 and 70 > max_seq_len = 60
    batches = []
    for B, L in shapes:
        lens = rng.integers(max(1, L // 3), L + 1, size=B).astype(np.int64)
        lens[rng.integers(0, B)] = L
        ids = np.zeros((B, L), np.int64)
        for b, n in enumerate(lens):
            ids[b, :n] = rng.integers(4, 131, size=n)
        batches.append((ids, lens, np.array([int(rng.integers(0, 4))], np.int64)))
    long_lived = engine_from_states(cfg, cfgmod.DEFAULT_STATS, ac, voc, device=0)
    for ids, lens, spk in batches:
        got, ml, T = long_lived.synthesize(ids, lens, spk)
        fresh = engine_from_states(cfg, cfgmod.DEFAULT_STATS, ac, voc, device=0)
        want, ml2, T2 = fresh.synthesize(ids, lens, spk)
        assert T == T2 and (ml == ml2).all()
        for b, n in enumerate(ml * hop):
            np.testing.assert_array_equal(got[b, :n], want[b, :n])
        per_utt, ml3, _ = fresh.synthesize(ids, lens, np.repeat(spk, ids.shape[0]))
        for b, n in enumerate(ml * hop):
            np.testing.assert_array_equal(per_utt[b, :n], want[b, :n])
        fresh.close()
    long_lived.close()


def test_utterance_with_zero_frames_in_a_batch():
    """d_control < 1 can scale every duration of a short utterance below one frame (U/layers.py:218-221 multiplies after rounding,
    the length regulator truncates): that utterance gets mel_len 0 while the batch goes on.  The reference pads it like any other
    row; here its attention tiles, LayerNorm rows and ragged row limits all see length 0."""
    from oracle import ref_numpy as orc
    from e2e_tts_amd import synth_weights as sw
    from e2e_tts_amd.runtime import engine_from_states
    cfg = cfgmod.tiny_config()
    ac = sw.make_acoustic_state(cfg, cfgmod.DEFAULT_STATS, 4, seed=81, mode="varied")
    voc = sw.make_vocoder_state(cfg, seed=82)
    oracle = orc.AcousticOracle(ac, cfg, cfgmod.DEFAULT_STATS)
    d_control = 0.3
    spk = np.array([1], np.int64)
    found = None
    for seed in range(200):   # a 2-phoneme utterance whose durations both fall to 0 frames, next to a normal one
        rng = np.random.Generator(np.random.PCG64(1000 + seed))
        lens = np.array([18, 2], np.int64)
        ids = np.zeros((2, 18), np.int64)
        for b, n in enumerate(lens):
            ids[b, :n] = rng.integers(4, 131, size=n)
        (mel, mel_post, dur), ml = oracle.inference(spk, ids, lens, d_control, 1.0, 1.0)
        d = np.exp(oracle.trace["log_d"].astype(np.float64)) - 1
        margin = np.abs((d - np.floor(d)) - 0.5)[np.arange(18)[None, :] < lens[:, None]].min()
        if ml[1] == 0 and ml[0] > 0 and margin > 1e-3:
            found = (ids, lens, mel_post, dur, ml)
            break
    assert found is not None, "no seed gives a zero-frame utterance"
    ids, lens, mel_post, dur, ml = found
    eng = engine_from_states(cfg, cfgmod.DEFAULT_STATS, ac, voc, device=0)
    for ragged in (True, False):
        eng.set_ragged(ragged)
        r = eng.acoustic(ids, lens, spk, d_control, 1.0, 1.0)
        np.testing.assert_array_equal(r["dur"], dur)
        np.testing.assert_array_equal(r["mel_lens"], ml)
        _, mp = eng.fetch_mel(2, r["T"])
        assert np.isfinite(mp).all() and mean_l1(mp, mel_post) < MEL_L1 / 10
        pcm, ml2, T = eng.synthesize(ids, lens, spk, d_control, 1.0, 1.0)
        assert (ml2 == ml).all() and T == ml[0]
        want = orc.VocoderOracle(voc, cfg).forward(mel_post.transpose(0, 2, 1))[:, 0]
        n = int(ml[0]) * 256
        ref_pcm = (want[0, :n] * 32768.0).astype(np.int16)
        assert (np.abs(pcm[0, :n].astype(np.int32) - ref_pcm.astype(np.int32)) <= 1).mean() >= 0.999
    eng.close()


@pytest.mark.parametrize("T", [1, 2, 3, 7])
def test_vocoder_on_very_short_inputs_matches_oracle(T):
    """Mel inputs shorter than every receptive field (T = 1: 256 samples from one frame; the dilated k = 11 convolutions reach 25
    rows to either side): zero padding on both sides inside one tile, fused and two-launch forms, both arithmetic modes."""
    from oracle import ref_numpy as orc
    from e2e_tts_amd import packer, synth_weights as sw
    from e2e_tts_amd._lib import Engine
    cfg = cfgmod.tiny_config()
    voc = sw.make_vocoder_state(cfg, seed=71)
    dims = cfgmod.dims_from_config(cfg, cfgmod.DEFAULT_STATS, 4)
    eng = Engine(dims, 0)
    eng.load_weights(packer.pack(dims, None, voc))
    rng = np.random.Generator(np.random.PCG64(72 + T))
    mel = rng.standard_normal((2, 80, T)).astype(np.float32)
    want = orc.VocoderOracle(voc, cfg).forward(mel)[:, 0]
    for prec in PRECISIONS:
        eng.set_precision(prec)
        for fused in (True, False):
            eng.set_fused_resblocks(fused)
            wav, pcm = eng.vocoder(mel, 2, T, pcm=True)
            assert wav.shape == want.shape == (2, T * 256)
            assert mean_l1(wav, want) < WAV_L1 / 10, (prec, fused)
            ref_pcm = (want * 32768.0).astype(np.int16)
            assert (np.abs(pcm.astype(np.int32) - ref_pcm.astype(np.int32)) <= 1).mean() >= 0.999, (prec, fused)
    eng.close()


def test_reload_and_mode_switches_leave_no_stale_state():
    """A live engine takes new weights (fragment images, fused-pair images and bindings are rebuilt), and switching the arithmetic
    mode / fusion / ragged compute away and back returns to the same bits.  A blob with only one of the two models refuses the
    other half's calls."""
    from e2e_tts_amd import packer, synth_weights as sw
    from e2e_tts_amd._lib import Engine
    cfg = cfgmod.tiny_config()
    dims = cfgmod.dims_from_config(cfg, cfgmod.DEFAULT_STATS, 4)
    blobs = [packer.pack(dims, sw.make_acoustic_state(cfg, cfgmod.DEFAULT_STATS, 4, seed=51 + k, mode="varied"),
                         sw.make_vocoder_state(cfg, seed=61 + k)) for k in range(2)]
    rng = np.random.Generator(np.random.PCG64(53))
    lens = np.array([21, 9, 30], np.int64)
    ids = np.zeros((3, 30), np.int64)
    for b, n in enumerate(lens):
        ids[b, :n] = rng.integers(4, 131, size=n)
    spk = np.array([2], np.int64)
    hop = cfg["audio"]["stft"]["hop_length"]

    def run(eng):
        pcm, ml, T = eng.synthesize(ids, lens, spk)
        return [pcm[b, :n].copy() for b, n in enumerate(ml * hop)]

    def same(a, b):
        return len(a) == len(b) and all(x.shape == y.shape and (x == y).all() for x, y in zip(a, b))

    fresh = []
    for blob in blobs:
        e = Engine(dims, 0)
        e.load_weights(blob)
        fresh.append(run(e))
        e.close()
    assert not same(fresh[0], fresh[1])                     # the two weight sets really differ
    eng = Engine(dims, 0)
    eng.load_weights(blobs[0])
    assert same(run(eng), fresh[0])
    eng.load_weights(blobs[1])                              # reload on a live engine
    assert same(run(eng), fresh[1])
    eng.load_weights(blobs[0])
    base = run(eng)                                         # the engine's default arithmetic: exact fp32
    assert same(base, fresh[0])
    eng.set_precision("bf16x3")
    fast = run(eng)
    eng.set_precision("fp32")
    assert same(run(eng), base)                             # away and back: the same bits
    assert all(np.abs(a.astype(np.int32) - b.astype(np.int32)).max() <= 2 for a, b in zip(base, fast))
    eng.set_precision("bf16x3")                             # the fused ResBlock kernels exist in the bf16 modes
    assert same(run(eng), fast)
    for level in (0, 1, 2):
        eng.set_fused_resblocks(level)
        assert same(run(eng), fast)                         # fused == two-launch form, bit for bit
    eng.set_ragged(False)
    assert same(run(eng), fast)
    eng.set_ragged(True)
    assert same(run(eng), fast)
    eng.close()
    # half blobs
    only_ac = Engine(dims, 0)
    only_ac.load_weights(packer.pack(dims, sw.make_acoustic_state(cfg, cfgmod.DEFAULT_STATS, 4, seed=51, mode="varied"), None))
    r = only_ac.acoustic(ids, lens, spk)
    assert r["T"] > 0
    with pytest.raises(RuntimeError):
        only_ac.vocoder(None, 3, r["T"])
    only_ac.close()
    only_voc = Engine(dims, 0)
    only_voc.load_weights(packer.pack(dims, None, sw.make_vocoder_state(cfg, seed=61)))
    wav, _ = only_voc.vocoder(rng.standard_normal((1, 80, 12)).astype(np.float32), 1, 12)
    assert wav.shape == (1, 12 * hop)
    with pytest.raises(RuntimeError):
        only_voc.acoustic(ids, lens, spk)
    only_voc.close()


def test_engines_are_thread_safe():
    """calls on one engine are serialised by its mutex, distinct engines are independent (own stream, own workspace).
    Four threads -- two sharing one engine, two on a second engine -- must each get the PCM a serial call gives (ctypes releases the
    GIL, so the calls really overlap)."""
    import threading
    from e2e_tts_amd import synth_weights as sw
    from e2e_tts_amd.runtime import engine_from_states
    cfg = cfgmod.tiny_config()
    ac = sw.make_acoustic_state(cfg, cfgmod.DEFAULT_STATS, 4, seed=31, mode="varied")
    voc = sw.make_vocoder_state(cfg, seed=32)
    engines = [engine_from_states(cfg, cfgmod.DEFAULT_STATS, ac, voc, device=0) for _ in range(2)]
    rng = np.random.Generator(np.random.PCG64(33))
    jobs = []
    for k in range(4):
        lens = rng.integers(5, 40, size=3).astype(np.int64)
        ids = np.zeros((3, int(lens.max())), np.int64)
        for b, n in enumerate(lens):
            ids[b, :n] = rng.integers(4, 131, size=n)
        jobs.append((engines[k // 2], ids, lens, np.array([k % 4], np.int64)))
    hop = cfg["audio"]["stft"]["hop_length"]

    def valid(pcm, ml):
        return [pcm[b, :n].copy() for b, n in enumerate(ml * hop)]

    serial = []
    for eng, ids, lens, spk in jobs:
        pcm, ml, _ = eng.synthesize(ids, lens, spk)
        serial.append(valid(pcm, ml))
    results, errors = [None] * 4, []

    def work(k):
        try:
            eng, ids, lens, spk = jobs[k]
            for _ in range(5):
                pcm, ml, _ = eng.synthesize(ids, lens, spk)
                results[k] = valid(pcm, ml)
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for k in range(4):
        for a, b in zip(results[k], serial[k]):
            np.testing.assert_array_equal(a, b)
    for e in engines:
        e.close()


def test_destroy_releases_all_device_memory():
    """Every workspace buffer -- including the ones only the iSTFT tail and the Conformer blocks allocate -- is returned by
    e2etts_destroy: free device memory does not shrink over repeated create / load / synthesize / destroy cycles (within allocator
    granularity)."""
    import torch
    from e2e_tts_amd import packer, synth_weights as sw
    from e2e_tts_amd._lib import Engine
    rng = np.random.Generator(np.random.PCG64(5))
    ids = rng.integers(4, 131, size=(2, 24)).astype(np.int64)
    lens = np.array([24, 13], np.int64)
    spk = np.array([1], np.int64)

    def one(vocoder, blocks):
        cfg = cfgmod.tiny_config()
        cfg["models"]["fastspeech2"]["building_block"]["block_type"] = blocks
        dims = cfgmod.dims_from_config(cfg, cfgmod.DEFAULT_STATS, 4, vocoder=vocoder)
        eng = Engine(dims, 0)
        eng.load_weights(packer.pack(dims, sw.make_acoustic_state(cfg, cfgmod.DEFAULT_STATS, 4, mode="varied"),
                                     sw.make_vocoder_state(cfg, vocoder=vocoder)))
        pcm, ml, T = eng.synthesize(ids, lens, spk)
        assert eng.device_bytes() > 0 and pcm.shape[1] == T * 256
        eng.close()

    combos = (("hifigan", "transformer"), ("istft", "conformer"), ("hifigan", "conformer"), ("istft", "transformer"))
    # first cycle: the HIP runtime's own first-use allocations (kernel argument pools, a kernel's first launch) happen here and stay;
    # a LEAK of the engine repeats with every create / destroy cycle, so the second and third cycles are the ones measured
    for vocoder, blocks in combos:
        one(vocoder, blocks)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]
    for _ in range(2):
        for vocoder, blocks in combos:
            one(vocoder, blocks)
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info(0)[0]
    assert free0 - free1 < (4 << 20), f"{(free0 - free1) / 2**20:.1f} MiB not returned over two create / destroy cycles of four engines"


def test_conformer_rejects_sequences_beyond_the_position_table():
    """Conformer attention needs pos_proj(table) rows for every position: a sequence longer than the shipped regenerated table
    is an error, not an out-of-bounds read."""
    from e2e_tts_amd import packer, synth_weights as sw
    from e2e_tts_amd._lib import Engine
    cfg = cfgmod.tiny_config()
    cfg["models"]["fastspeech2"]["building_block"]["block_type"] = "conformer"
    dims = cfgmod.dims_from_config(cfg, cfgmod.DEFAULT_STATS, 4, pos_table_rows=96)
    eng = Engine(dims, 0)
    eng.load_weights(packer.pack(dims, sw.make_acoustic_state(cfg, cfgmod.DEFAULT_STATS, 4, mode="varied"), sw.make_vocoder_state(cfg)))
    rng = np.random.Generator(np.random.PCG64(3))
    ok = rng.integers(4, 131, size=(1, 80)).astype(np.int64)        # 80 > max_seq_len = 60: regenerated table, fits in 96 rows
    with pytest.raises(ValueError):                                 # the decoder's T (several frames per phoneme) does not
        eng.acoustic(ok, np.array([80], np.int64), np.array([0], np.int64))
    with pytest.raises(ValueError):
        eng.acoustic(rng.integers(4, 131, size=(1, 100)).astype(np.int64), np.array([100], np.int64), np.array([0], np.int64))


@pytest.mark.parametrize("name", ["tiny_b3", "tiny_ctl", "full_b3", "c3_mixed", "full_long", "tiny_cf_b3", "tiny_hv_b3", "tiny_cf_hv_b3", "tiny_frame_b3", "tiny_pframe_b3"])
def test_ragged_synthesize_is_bit_identical_on_valid_samples(name):
    """synthesize() with ragged compute (skip what no valid sample depends on) vs the full padded batch: same PCM on every
    valid sample, and within 1 LSB of the reference's waveform there."""
    g = load_golden(name)
    cfg, eng = engine_for(g, name)
    d, p, e = (float(x) for x in g["controls"])
    spk = np.array([int(g["speaker"])], np.int64)
    hop = cfg["audio"]["stft"]["hop_length"]
    for prec in PRECISIONS:
        eng.set_precision(prec)
        eng.set_ragged(False)
        full, mel_lens, T = eng.synthesize(g["ids"], g["lens"], spk, d, p, e)
        eng.set_ragged(True)
        eng.poison_workspace()   # the skipped rows must not matter: without this they would still hold the padded run's (correct) values
        rag, mel_lens2, T2 = eng.synthesize(g["ids"], g["lens"], spk, d, p, e)
        assert T == T2
        np.testing.assert_array_equal(mel_lens, g["mel_lens"])
        np.testing.assert_array_equal(mel_lens2, g["mel_lens"])
        for b, n in enumerate(mel_lens * hop):
            np.testing.assert_array_equal(rag[b, :n], full[b, :n])
        if "wav" in g:
            ref_pcm = (g["wav"] * 32768.0).astype(np.int16)
            ok = [np.abs(rag[b, :n].astype(np.int32) - ref_pcm[b, :n].astype(np.int32)) <= 1 for b, n in enumerate(mel_lens * hop)]
            assert np.concatenate(ok).mean() >= 0.999
    eng.set_ragged(True)


@pytest.mark.parametrize("name", ["tiny_b3", "full_b3", "c3_mixed"])
def test_fused_resblock_pairs_are_bit_identical(name):
    """resblock_pair.hip (a pair's intermediate in LDS; 32 .. 256 channels) and resblock_chain.hip (a whole k = 3 ResBlock, residual
    stream in registers; 32 / 64 channels) against the same pairs as two conv_gemm launches each: same arithmetic in the same order,
    so the PCM and the fp32 waveform must be equal bit for bit -- padded (vocoder) and ragged (synthesize).  The tiny config's
    stages are 32 / 16 / 8 / 4 channels wide (only its first takes the fused kernels); the default config's 256 / 128 / 64 / 32
    exercise every fused form."""
    g = load_golden(name)
    cfg, eng = engine_for(g, name)
    d, p, e = (float(x) for x in g["controls"])
    spk = np.array([int(g["speaker"])], np.int64)
    hop = cfg["audio"]["stft"]["hop_length"]
    try:
        for prec in ("fp32", "bf16x3", "bf16"):   # fp32: fused pairs at <= 64 channels only (no fp32 chain: levels 1 and 2 coincide)
            eng.set_precision(prec)
            out = {}
            for level in (0, 1, 2):   # two launches per pair / fused pairs / fused pairs + whole k = 3 ResBlocks
                eng.set_fused_resblocks(level)
                eng.set_ragged(False)
                r = eng.acoustic(g["ids"], g["lens"], spk, d, p, e, want=("mel_lens",))
                wav, pcm = eng.vocoder(None, r["B"], r["T"], pcm=True)
                eng.set_ragged(True)
                eng.poison_workspace()
                rag, mel_lens, _ = eng.synthesize(g["ids"], g["lens"], spk, d, p, e)
                out[level] = (wav, pcm, rag, mel_lens)
            for level in (1, 2):
                np.testing.assert_array_equal(out[level][0], out[0][0])
                np.testing.assert_array_equal(out[level][1], out[0][1])
                for b, n in enumerate(out[level][3] * hop):
                    np.testing.assert_array_equal(out[level][2][b, :n], out[0][2][b, :n])
            if prec == "bf16x3" and "wav" in g:
                assert mean_l1(out[2][0], g["wav"]) < WAV_L1
    finally:
        eng.set_fused_resblocks(True)
        eng.set_precision("fp32")


@pytest.mark.parametrize("tag", ["tiny_rb2", "tiny_rb1", "full_rb2"])
def test_istft_vocoder_matches_reference(tag):
    """iSTFTNet on the engine (ResBlock2 / ResBlock1 trunk, reflection pad, conv_post to n_fft + 2 channels, exp / sin heads,
    inverse STFT with overlap-add) against outputs of the reference's iSTFT module + inverse_stft."""
    from test_oracle_golden import _istft_case
    from e2e_tts_amd.models import iSTFT
    import torch
    g = load_golden("istft")
    cfg, state = _istft_case(g, tag)
    v = iSTFT(cfg["models"]["istft"], device=0)
    v.load_state_dict(state)
    v.eval()
    mel = g[f"{tag}.mel"]
    for prec, bar in (("fp32", 2e-6), ("bf16x3", 2e-5)):
        v._ensure_engine().set_precision(prec)
        spec, phase = v(torch.from_numpy(mel))
        wav = v.inference(torch.from_numpy(mel))
        assert tuple(wav.shape) == g[f"{tag}.wav"].shape
        assert mean_l1(spec.cpu().numpy(), g[f"{tag}.spec"]) < bar * 10, prec
        assert mean_l1(phase.cpu().numpy(), g[f"{tag}.phase"]) < bar * 10, prec
        assert mean_l1(wav.cpu().numpy(), g[f"{tag}.wav"]) < bar, prec
    # streaming: chunked pushes concatenate to the one-shot waveform bit for bit (the reflection pad and the overlap-add edges of a
    # window fall into its discarded context)
    eng = v._ensure_engine()
    eng.set_precision("bf16x3")
    rng = np.random.Generator(np.random.PCG64(17))
    T = 61
    m2 = rng.standard_normal((2, T, 80)).astype(np.float32)
    whole, _ = eng.vocoder(m2, 2, T, channels_first=False)
    for sizes in ([T], [1, 7, 20, 3, 30], [16, 16, 16, 13]):
        pos, chunks = 0, []
        for n in sizes:
            chunks.append(np.ascontiguousarray(m2[:, pos:pos + n]))
            pos += n
        out = np.concatenate(list(eng.vocoder_stream(chunks, 2)), axis=1)
        np.testing.assert_array_equal(out, whole)


def test_istft_engine_end_to_end_ragged():
    """Acoustic model + iSTFTNet vocoder in one engine: synthesize() with ragged compute gives the padded run's PCM on every valid sample."""
    from e2e_tts_amd import packer, synth_weights as sw
    from e2e_tts_amd._lib import Engine
    cfg = cfgmod.tiny_config()
    dims = cfgmod.dims_from_config(cfg, cfgmod.DEFAULT_STATS, 4, vocoder="istft")
    ac = sw.make_acoustic_state(cfg, cfgmod.DEFAULT_STATS, 4, seed=21, mode="varied")
    voc = sw.make_vocoder_state(cfg, seed=22, vocoder="istft")
    eng = Engine(dims, 0)
    eng.load_weights(packer.pack(dims, ac, voc))
    rng = np.random.Generator(np.random.PCG64(23))
    lens = np.array([30, 11, 22, 5], np.int64)
    ids = np.zeros((4, 30), np.int64)
    for b, n in enumerate(lens):
        ids[b, :n] = rng.integers(4, 131, size=n)
    spk = np.array([2], np.int64)
    eng.set_ragged(False)
    full, ml, T = eng.synthesize(ids, lens, spk)
    eng.set_ragged(True)
    eng.poison_workspace()
    rag, ml2, T2 = eng.synthesize(ids, lens, spk)
    assert T == T2 and (ml == ml2).all() and ml.max() == T and ml.min() < T
    for b, n in enumerate(ml * 256):
        np.testing.assert_array_equal(rag[b, :n], full[b, :n])
    assert np.abs(full.astype(np.int32)).max() > 10


@pytest.mark.parametrize("blocks", ["transformer", "conformer"])
def test_full_size_batch_is_deterministic_and_linear_in_batch(blocks):
    """Size-independent properties at the bench size (B = 32, L = 128 -> T = 768, default model, FFT or Conformer blocks): two runs
    give the same bits, and an utterance's PCM does not depend on WHICH other utterances share the (equally long) batch -- rows are
    independent, also across the tile shapes the launches pick for 32 and for 5 utterances."""
    from e2e_tts_amd import synth_weights as sw
    from e2e_tts_amd.runtime import engine_from_states
    cfg = cfgmod.default_config()
    cfg["models"]["fastspeech2"]["building_block"]["block_type"] = blocks
    ac = sw.make_acoustic_state(cfg, cfgmod.DEFAULT_STATS, 4, seed=1234, mode="fixed", frames_per_phoneme=6)
    voc = sw.make_vocoder_state(cfg, seed=4321)
    eng = engine_from_states(cfg, cfgmod.DEFAULT_STATS, ac, voc, device=0)
    rng = np.random.Generator(np.random.PCG64(99))
    B, L = 32, 128
    ids = rng.integers(4, 131, size=(B, L)).astype(np.int64)
    lens = np.full((B,), L, np.int64)
    spk = np.array([1], np.int64)
    a, ml, T = eng.synthesize(ids, lens, spk)
    b, ml2, T2 = eng.synthesize(ids, lens, spk)
    assert T == T2 == 768 and (ml == 768).all()
    np.testing.assert_array_equal(a, b)
    perm = rng.permutation(B)
    c, _, _ = eng.synthesize(np.ascontiguousarray(ids[perm]), lens, spk)
    np.testing.assert_array_equal(c, a[perm])
    half, _, _ = eng.synthesize(np.ascontiguousarray(ids[:5]), lens[:5], spk)
    np.testing.assert_array_equal(half, a[:5])
    assert np.abs(a.astype(np.int32)).max() > 100  # not silence


def test_fp32_fragment_path_is_bit_identical_to_the_lds_weight_tile(tmp_path):
    """conv_gemm's exact-fp32 kernel reads its weights either as MFMA fragments from L2 (default since round 2) or through the LDS
    weight tile (E2ETTS_NO_FRAG32=1, read once per process): same MFMA order, so the same bits.  The child process runs the default-config
    B = 3 fixture through the LDS-tile kernels with the fused fp32 pairs switched off as well -- the round-1 path -- and hands back its
    mel, waveform and PCM; this process runs the default path (fragments + fused fp32 pairs)."""
    import os
    import subprocess
    import sys
    from conftest import GOLD, ROOT
    g = load_golden("full_b3")
    cfg, eng = engine_for(g, "full_b3")
    eng.set_precision("fp32")
    eng.set_ragged(False)
    spk = np.array([int(g["speaker"])], np.int64)
    d, p, e = (float(x) for x in g["controls"])
    r = eng.acoustic(g["ids"], g["lens"], spk, d, p, e, want=("mel_lens",))
    _, mel_post = eng.fetch_mel(r["B"], r["T"], mel=False)
    wav, pcm = eng.vocoder(None, r["B"], r["T"], pcm=True)
    eng.set_ragged(True)
    script = tmp_path / "nofrag_child.py"
    script.write_text(
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})\n"
        "from conftest import load_golden, states_for\n"
        "from e2e_tts_amd import config as cfgmod\n"
        "from e2e_tts_amd.runtime import engine_from_states\n"
        "g = load_golden('full_b3'); cfg, ac, voc = states_for(g, 'full_b3')\n"
        "eng = engine_from_states(cfg, cfgmod.DEFAULT_STATS, ac, voc, device=0)\n"
        "eng.set_precision('fp32'); eng.set_ragged(False); eng.set_fused_resblocks(0)\n"
        "spk = np.array([int(g['speaker'])], np.int64); d, p, e = (float(x) for x in g['controls'])\n"
        "r = eng.acoustic(g['ids'], g['lens'], spk, d, p, e, want=('mel_lens',))\n"
        "_, mel_post = eng.fetch_mel(r['B'], r['T'], mel=False)\n"
        "wav, pcm = eng.vocoder(None, r['B'], r['T'], pcm=True)\n"
        "np.savez(sys.argv[1], mel_post=mel_post, wav=wav, pcm=pcm)\n")
    out = tmp_path / "nofrag.npz"
    env = dict(os.environ, E2ETTS_NO_FRAG32="1")
    rr = subprocess.run([sys.executable, str(script), str(out)], env=env, capture_output=True, text=True, timeout=900)
    assert rr.returncode == 0, rr.stderr[-3000:]
    o = np.load(out)
    np.testing.assert_array_equal(o["mel_post"], mel_post)
    np.testing.assert_array_equal(o["wav"], wav)
    np.testing.assert_array_equal(o["pcm"], pcm)


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_attention_split_form_is_bit_identical_to_one_wavefront_per_query_tile(tmp_path, precision):
    """Small grids (this B = 3 fixture, the B = 1 latency path) run attention with two wavefronts per 32-query tile, half the head
    dimension each (attention_split_kernel / attention_x3_split_kernel); large ones with one (attention_kernel / attention_x3_kernel),
    which adds its two half sums in the same order.  The child process is held to the one-wavefront kernels (E2ETTS_ATT_SPLIT_MAX=0, read
    once per process): durations, buckets and mel must come out bit for bit the same as from this process's split kernels.  The child
    also runs its few-rows convolutions on conv_gemm's 64 x 64 tile (E2ETTS_ROWS=0) where this process takes conv_rows -- one wavefront
    per 32-row tile, no workgroup barrier, the same MFMA sequence -- and its vocoder's ResBlocks one after the other on one stream
    (E2ETTS_VOC_CONC_FRAMES=0) where this process runs the three of a stage side by side on HIP side streams and joins their sums:
    waveform and PCM bit for bit as well."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    g = load_golden("full_b3")
    cfg, eng = engine_for(g, "full_b3")
    eng.set_precision(precision)
    spk = np.array([int(g["speaker"])], np.int64)
    d, p, e = (float(x) for x in g["controls"])
    r = eng.acoustic(g["ids"], g["lens"], spk, d, p, e, want=("dur", "mel_lens", "pitch_idx", "energy_idx"))
    mel, mel_post = eng.fetch_mel(r["B"], r["T"])
    wav, pcm = eng.vocoder(None, r["B"], r["T"], pcm=True)
    eng.set_precision("fp32")
    script = tmp_path / "att_child.py"
    script.write_text(
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})\n"
        "from conftest import load_golden, states_for\n"
        "from e2e_tts_amd import config as cfgmod\n"
        "from e2e_tts_amd.runtime import engine_from_states\n"
        "g = load_golden('full_b3'); cfg, ac, voc = states_for(g, 'full_b3')\n"
        "eng = engine_from_states(cfg, cfgmod.DEFAULT_STATS, ac, voc, device=0)\n"
        f"eng.set_precision({precision!r})\n"
        "spk = np.array([int(g['speaker'])], np.int64); d, p, e = (float(x) for x in g['controls'])\n"
        "r = eng.acoustic(g['ids'], g['lens'], spk, d, p, e, want=('dur', 'mel_lens', 'pitch_idx', 'energy_idx'))\n"
        "mel, mel_post = eng.fetch_mel(r['B'], r['T'])\n"
        "wav, pcm = eng.vocoder(None, r['B'], r['T'], pcm=True)\n"
        "np.savez(sys.argv[1], mel=mel, mel_post=mel_post, dur=r['dur'], pitch_idx=r['pitch_idx'], energy_idx=r['energy_idx'], wav=wav, pcm=pcm)\n")
    out = tmp_path / "att.npz"
    env = dict(os.environ, E2ETTS_ATT_SPLIT_MAX="0", E2ETTS_ROWS="0", E2ETTS_VOC_CONC_FRAMES="0")
    rr = subprocess.run([sys.executable, str(script), str(out)], env=env, capture_output=True, text=True, timeout=900)
    assert rr.returncode == 0, rr.stderr[-3000:]
    o = np.load(out)
    for k in ("dur", "pitch_idx", "energy_idx"):
        np.testing.assert_array_equal(o[k], r[k])
    np.testing.assert_array_equal(o["mel"], mel)
    np.testing.assert_array_equal(o["mel_post"], mel_post)
    np.testing.assert_array_equal(o["wav"], wav)
    np.testing.assert_array_equal(o["pcm"], pcm)
    assert np.abs(mel).max() > 0.1


def test_fused_kernels_on_unusual_resblock_geometry():
    """The fused ResBlock kernels take kernel sizes and dilations from the config (reference V/generator.py:27-31, V/layers.py:11-31), not
    the V1 constants: a generator with kernel sizes (3, 5, 9), dilations (1, 2, 4) / (2, 6, 3) / (1, 1, 1) and stages of 64 / 32 / 16 / 8
    channels -- the k = 3 ResBlock runs as a chain at 64 and 32 channels with halo 2 + 3 + 5, the others as pairs -- against the numpy
    oracle (tolerance) and across the fusion levels (bit for bit), on a length that is no multiple of any tile."""
    from e2e_tts_amd import synth_weights as sw
    from e2e_tts_amd.models import HifiGan
    from oracle import ref_numpy as orc
    cfg = cfgmod.default_config()
    cfg["models"]["hifigan"].update(upsample_initial_channel=128, resblock_kernel_sizes=[3, 5, 9],
                                    resblock_dilation_sizes=[[1, 2, 4], [2, 6, 3], [1, 1, 1]])
    voc = sw.make_vocoder_state(cfg, seed=77)
    v = HifiGan(cfg["models"]["hifigan"])
    v.load_state_dict(sw.to_torch(voc))
    eng = v.eval().to(0).engine
    rng = np.random.Generator(np.random.PCG64(78))
    B, T = 3, 37
    mel = rng.standard_normal((B, T, 80)).astype(np.float32)
    ref = orc.VocoderOracle(voc, cfg).forward(mel.transpose(0, 2, 1))[:, 0]
    for prec, bar in (("fp32", 1e-5), ("bf16x3", 1e-5), ("bf16", 2e-3)):
        eng.set_precision(prec)
        outs = []
        for level in (0, 1, 2):
            eng.set_fused_resblocks(level)
            wav, pcm = eng.vocoder(mel, B, T, channels_first=False, pcm=True)
            outs.append((wav, pcm))
        assert mean_l1(outs[2][0], ref) < bar, (prec, mean_l1(outs[2][0], ref))
        for level in (1, 2):
            np.testing.assert_array_equal(outs[level][0], outs[0][0])
            np.testing.assert_array_equal(outs[level][1], outs[0][1])
    eng.set_fused_resblocks(True)
    eng.close()


def test_runaway_durations_are_rejected_not_wrapped():
    """exp(log_d) that is huge or infinite must not wrap the int32 repeat counts / the frame total into a small T that passes the
    size checks.  The duration kernel caps a phoneme at 2^20 frames and sums in 64 bits, so the host sees the real magnitude and refuses."""
    from e2e_tts_amd import synth_weights as sw
    from e2e_tts_amd.runtime import engine_from_states
    cfg = cfgmod.tiny_config()
    stats = cfgmod.DEFAULT_STATS
    ids = np.random.Generator(np.random.PCG64(5)).integers(4, 131, size=(2, 9)).astype(np.int64)
    lens = np.array([9, 4], np.int64)
    spk = np.array([1], np.int64)
    for bias in (np.log(3.0e6), 60.0, 100.0):   # 3e6 frames per phoneme; 1e26; exp() = inf in fp32
        ac = sw.make_acoustic_state(cfg, stats, 4, seed=3, mode="fixed")
        ac["variance_adaptor.duration_predictor.linear.bias"] = np.array([bias], np.float32)
        eng = engine_from_states(cfg, stats, ac, sw.make_vocoder_state(cfg, seed=4), device=0)
        with pytest.raises(ValueError, match="exceeds|unreasonably large"):
            eng.acoustic(ids, lens, spk)
        with pytest.raises(ValueError, match="exceeds|unreasonably large"):
            eng.synthesize(ids, lens, spk)
        eng.close()


@pytest.mark.parametrize("B", [64, 70])
def test_ragged_batches_at_and_beyond_the_compact_grid_capacity(B):
    """Ragged launches find their utterance through a table in the kernel arguments that holds 64 utterances (kernels.h: RowMap,
    round 3); B = 64 fills it, B = 70 falls back to the padded grid (workgroups beyond an utterance's rows exit).  Either way the valid
    samples must equal the padded computation bit for bit, in both arithmetic modes, and an utterance must not depend on the batch
    around it more than the reference's padded batch does -- checked here as ragged == padded only (tiny config, mixed lengths 3 .. 29)."""
    g = load_golden("tiny_b3")
    cfg, eng = engine_for(g, "tiny_b3")
    rng = np.random.Generator(np.random.PCG64(64 + B))
    lens = rng.integers(3, 30, size=B).astype(np.int64)
    lens[0] = 29
    L = int(lens.max())
    ids = np.zeros((B, L), np.int64)
    for b, n in enumerate(lens):
        ids[b, :n] = rng.integers(4, 131, size=n)
    spk = np.array([1], np.int64)
    hop = cfg["audio"]["stft"]["hop_length"]
    for prec in PRECISIONS:
        eng.set_precision(prec)
        eng.set_ragged(False)
        full, mel_lens, T = eng.synthesize(ids, lens, spk)
        eng.set_ragged(True)
        eng.poison_workspace()
        rag, mel_lens2, T2 = eng.synthesize(ids, lens, spk)
        assert T == T2 and np.array_equal(mel_lens, mel_lens2) and int(mel_lens.min()) < T
        for b, n in enumerate(mel_lens * hop):
            np.testing.assert_array_equal(rag[b, :n], full[b, :n])
    eng.set_ragged(True)


@pytest.mark.parametrize("name", ["tiny_cf_b3", "full_cf_b2"])
def test_fused_glu_depthwise_kernel_is_bit_identical_to_the_two_kernel_form(tmp_path, name):
    """Conformer convolution module: dwconv_glu_swish_kernel (GLU of a tile staged in LDS once, weights in registers, rows streamed past the
    accumulators; k = 7 at 64 channels in the tiny config, k = 31 at 384 in the default one) against glu_kernel + dwconv_swish_kernel
    (E2ETTS_DWGLU=0, read once per process: a child runs it): same products added in the same order, so the mel must be equal bit for bit."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    g = load_golden(name)
    cfg, eng = engine_for(g, name)
    eng.set_precision("fp32")
    spk = np.array([int(g["speaker"])], np.int64)
    d, p, e = (float(x) for x in g["controls"])
    r = eng.acoustic(g["ids"], g["lens"], spk, d, p, e, want=("dur", "mel_lens"))
    _, mel_post = eng.fetch_mel(r["B"], r["T"], mel=False)
    script = tmp_path / "dwglu_child.py"
    script.write_text(
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})\n"
        "from conftest import load_golden, states_for\n"
        "from e2e_tts_amd import config as cfgmod\n"
        "from e2e_tts_amd.runtime import engine_from_states\n"
        f"g = load_golden({name!r}); cfg, ac, voc = states_for(g, {name!r})\n"
        "eng = engine_from_states(cfg, cfgmod.DEFAULT_STATS, ac, voc, device=0)\n"
        "eng.set_precision('fp32')\n"
        "spk = np.array([int(g['speaker'])], np.int64); d, p, e = (float(x) for x in g['controls'])\n"
        "r = eng.acoustic(g['ids'], g['lens'], spk, d, p, e, want=('dur', 'mel_lens'))\n"
        "_, mel_post = eng.fetch_mel(r['B'], r['T'], mel=False)\n"
        "np.savez(sys.argv[1], mel_post=mel_post, dur=r['dur'])\n")
    out = tmp_path / "dwglu.npz"
    rr = subprocess.run([sys.executable, str(script), str(out)], env=dict(os.environ, E2ETTS_DWGLU="0"), capture_output=True, text=True, timeout=900)
    assert rr.returncode == 0, rr.stderr[-3000:]
    o = np.load(out)
    np.testing.assert_array_equal(o["dur"], r["dur"])
    np.testing.assert_array_equal(o["mel_post"], mel_post)


def test_parallel_key_segments_match_the_in_register_merge(tmp_path):
    """Exact-fp32 attention runs its online softmax per segment of 256 keys and merges the segments in key order (attention.hip,
    ATT_SEG_CHUNKS).  On a small padded grid -- this B = 1, T = 768 fixture: 24 query blocks -- every segment gets a workgroup of its own
    and attention_combine_kernel merges them; a large grid (and the child process here, E2ETTS_ATT_PAR=0) merges in registers.  The same
    operations in the same order: mel and PCM must be equal bit for bit, and both still meet the fixture the reference produced."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    g = load_golden("c2_latency")
    cfg, eng = engine_for(g, "c2_latency")
    eng.set_precision("fp32")
    spk = np.array([int(g["speaker"])], np.int64)
    r = eng.acoustic(g["ids"], g["lens"], spk, want=("dur", "mel_lens"))
    assert r["T"] > 256 and r["B"] == 1
    _, mel_post = eng.fetch_mel(r["B"], r["T"], mel=False)
    wav, pcm = eng.vocoder(None, r["B"], r["T"], pcm=True)
    assert mean_l1(mel_post, g["mel_post"]) < 1e-4
    script = tmp_path / "attpar_child.py"
    script.write_text(
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})\n"
        "from conftest import load_golden, states_for\n"
        "from e2e_tts_amd import config as cfgmod\n"
        "from e2e_tts_amd.runtime import engine_from_states\n"
        "g = load_golden('c2_latency'); cfg, ac, voc = states_for(g, 'c2_latency')\n"
        "eng = engine_from_states(cfg, cfgmod.DEFAULT_STATS, ac, voc, device=0)\n"
        "eng.set_precision('fp32')\n"
        "spk = np.array([int(g['speaker'])], np.int64)\n"
        "r = eng.acoustic(g['ids'], g['lens'], spk, want=('dur', 'mel_lens'))\n"
        "_, mel_post = eng.fetch_mel(r['B'], r['T'], mel=False)\n"
        "wav, pcm = eng.vocoder(None, r['B'], r['T'], pcm=True)\n"
        "np.savez(sys.argv[1], mel_post=mel_post, pcm=pcm)\n")
    out = tmp_path / "attpar.npz"
    rr = subprocess.run([sys.executable, str(script), str(out)], env=dict(os.environ, E2ETTS_ATT_PAR="0"), capture_output=True, text=True, timeout=900)
    assert rr.returncode == 0, rr.stderr[-3000:]
    o = np.load(out)
    np.testing.assert_array_equal(o["mel_post"], mel_post)
    np.testing.assert_array_equal(o["pcm"], pcm)


def _random_geometry(seed):
    """A small model whose every dimension differs from the two fixture configurations: head dim 32 .. 192, FFN / predictor / postnet widths
    that are no multiple of 32, kernel sizes 3 .. 9, 2 .. 4 vocoder stages of assorted rates, 1 .. 3 ResBlocks with 1 .. 3 dilations."""
    rng = np.random.Generator(np.random.PCG64(seed))
    pick = lambda xs: xs[int(rng.integers(0, len(xs)))]   # noqa: E731
    cfg = cfgmod.tiny_config()
    fs = cfg["models"]["fastspeech2"]
    hidden, heads = pick([(64, 2), (96, 1), (128, 2), (192, 1), (128, 1), (96, 3), (192, 2)])
    fs["encoder_hidden"] = fs["decoder_hidden"] = hidden
    fs["encoder_layers"], fs["decoder_layers"] = int(rng.integers(1, 3)), int(rng.integers(1, 3))
    tr = fs["building_block"]["transformer"]
    tr["encoder_head"] = tr["decoder_head"] = heads
    if seed % 3 == 0:   # decoder_head of its own (U/blocks/transformer.py:105): any count whose head dim the attention kernels have
        tr["decoder_head"] = pick([n for n in (1, 2, 3, 4, 6) if hidden % n == 0 and hidden // n in (32, 64, 96, 128, 192) and n != heads])
    tr["conv_filter_size"] = pick([64, 100, 136, 160])
    tr["conv_kernel_size"] = [pick([3, 5, 9]), 1]
    vp = fs["variance"]["variance_predictor"]
    vp["filter_size"] = pick([32, 44, 64])
    vp["dur_predictor_layers"] = int(rng.integers(1, 4))
    vp["pit_predictor_layers"] = vp["ener_predictor_layers"] = int(rng.integers(1, 4))
    vp["dur_predictor_kernel"] = pick([3, 5])
    vp["pit_predictor_kernel"] = vp["ener_predictor_kernel"] = pick([3, 5])
    if seed % 2 == 0:   # energy predictor of its own depth / kernel (U/layers.py:92,96)
        vp["ener_predictor_layers"], vp["ener_predictor_kernel"] = int(rng.integers(1, 4)), pick([3, 5, 7])
    fs["postnet"].update(embedding_dim=pick([32, 44, 60]), conv_layers=int(rng.integers(2, 6)), kernel_size=pick([3, 5, 7]))
    rates = pick([[4, 4], [2, 2, 2], [8, 2], [4, 2, 2, 2], [2, 4, 2]])
    nk = int(rng.integers(1, 4))
    nd = int(rng.integers(1, 4))
    hg = cfg["models"]["hifigan"]
    hg.update(upsample_rates=rates, upsample_kernel_sizes=[2 * r for r in rates], upsample_initial_channel=pick([64, 128]),
              resblock_kernel_sizes=[pick([3, 5, 7, 11]) for _ in range(nk)],
              resblock_dilation_sizes=[[pick([1, 2, 3, 5]) for _ in range(nd)] for _ in range(nk)])
    cfg["audio"]["stft"]["hop_length"] = int(np.prod(rates))
    # the variance adaptor's configuration knobs (U/layers.py:48-104,136-160,226-257,400-402), one per seed
    ve = fs["variance"]["variance_embedding"]
    if seed == 2:
        ve["use_uv"] = False
    if seed == 4:
        vp["ffn_padding"] = "LEFT"
    if seed == 5:
        ve["pitch_feature"] = ve["energy_feature"] = "frame_level"
    if seed == 6:
        ve["pitch_quantization"] = "log"
    if seed == 103:
        ve["energy_feature"] = "frame_level"
    if seed >= 100:   # Conformer blocks (U/blocks/conformer.py:31-36): head dims 8 .. 96, depthwise kernels with and without a fused instantiation
        hidden, heads = pick([(64, 8), (64, 4), (96, 2), (128, 4), (96, 1), (192, 4)])
        fs["encoder_hidden"] = fs["decoder_hidden"] = hidden
        fs["building_block"]["block_type"] = "conformer"
        fs["building_block"]["conformer"].update(encoder_head=heads, decoder_head=heads, ffn_expansion_factor=pick([2, 4]),
                                                 conv_kernel_size=pick([5, 7, 9, 15, 31]))
        if seed % 2 == 0:
            fs["building_block"]["conformer"]["decoder_head"] = pick([n for n in (1, 2, 4, 8, 12) if hidden % n == 0 and hidden // n in (8, 16, 32, 48, 64, 96)
                                                                       and n != heads])
    return cfg


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6, 101, 102, 103])
def test_random_model_geometries_match_the_oracle(seed):
    """HIP vs the numpy oracle on model geometries no fixture has (the oracle is pinned by the fixtures' two configurations; the kernels
    take every dimension from the config, reference U/model.py:20-60, V/generator.py:14-35; seeds >= 100: Conformer blocks): discrete outputs exact on ids chosen away from
    the rounding boundaries (the fixtures' own margin search), mel / waveform within the fp32 bars in both arithmetic modes, and ragged
    compute bit-identical to the padded batch on valid samples."""
    from e2e_tts_amd import synth_weights as sw
    from e2e_tts_amd.runtime import engine_from_states
    from oracle import ref_numpy as orc
    from oracle.make_goldens import search_ids
    cfg = _random_geometry(seed)
    stats = cfgmod.DEFAULT_STATS
    ac = sw.make_acoustic_state(cfg, stats, 4, seed=700 + seed, mode="varied")
    voc = sw.make_vocoder_state(cfg, seed=800 + seed)
    o = orc.AcousticOracle(ac, cfg, stats)
    lens = np.array([[29, 7, 18], [33, 33, 1, 12], [5, 40]][seed % 3], np.int64)
    spk_id = seed % 4
    _, ids = search_ids(o, [int(x) for x in lens], spk_id, stats, (1.0, 1.0, 1.0), 2e-3, 40, 5000 + 100 * seed, False)
    spk = np.array([spk_id], np.int64)
    (omel, omel_post, odur), omel_lens = o.inference(spk, ids, lens)
    owav = orc.VocoderOracle(voc, cfg).forward(omel_post.transpose(0, 2, 1))[:, 0]
    hop = cfg["audio"]["stft"]["hop_length"]
    eng = engine_from_states(cfg, stats, ac, voc)
    try:
        for prec in PRECISIONS:
            eng.set_precision(prec)
            r = eng.acoustic(ids, lens, spk, want=("dur", "mel_lens", "pitch_idx", "energy_idx"))
            np.testing.assert_array_equal(r["dur"], odur)
            np.testing.assert_array_equal(r["mel_lens"], omel_lens)
            np.testing.assert_array_equal(r["pitch_idx"], o.trace["pitch_idx"])
            np.testing.assert_array_equal(r["energy_idx"], o.trace["energy_idx"])
            mel, mel_post = eng.fetch_mel(r["B"], r["T"])
            assert mean_l1(mel_post, omel_post) < MEL_L1, (seed, prec, mean_l1(mel_post, omel_post))
            wav, _ = eng.vocoder(None, r["B"], r["T"])
            assert mean_l1(wav, owav) < WAV_L1, (seed, prec, mean_l1(wav, owav))
            eng.set_ragged(False)
            full, ml, T = eng.synthesize(ids, lens, spk)
            eng.set_ragged(True)
            eng.poison_workspace()
            rag, ml2, T2 = eng.synthesize(ids, lens, spk)
            assert T == T2 and np.array_equal(ml, ml2) and np.array_equal(ml, omel_lens)
            for b, n in enumerate(ml * hop):
                np.testing.assert_array_equal(rag[b, :n], full[b, :n])
    finally:
        eng.close()


def test_frame_level_features_up_to_the_last_row_of_the_position_table():
    """VERDICT r3 item 8 / commit 71c90f4: with frame_level features the variance predictors' fairseq position table (U/sublayers.py:28-60,
    which the reference grows on demand) is indexed by FRAME, so it must reach as far as the decoder's own table lets T go (pos_table_rows =
    4 096): one utterance of 682 phonemes x exactly 6 frames = 4 092 frames runs and matches the numpy oracle (discrete outputs exact, frame
    level taps included); 683 phonemes = 4 098 frames is refused with E2ETTS_EINVAL and a message naming the table, not computed past it."""
    from e2e_tts_amd import synth_weights as sw
    from e2e_tts_amd.packer import variance_position_table
    from e2e_tts_amd.runtime import engine_from_states
    from oracle import ref_numpy as orc
    from oracle.make_goldens import pv_variant
    cfg = pv_variant(cfgmod.tiny_config(), "frame")
    stats = cfgmod.DEFAULT_STATS
    ac = sw.make_acoustic_state(cfg, stats, 4, seed=77, mode="fixed", frames_per_phoneme=6)
    voc = sw.make_vocoder_state(cfg, seed=78)
    eng = engine_from_states(cfg, stats, ac, voc, device=0)
    try:
        rng = np.random.Generator(np.random.PCG64(79))
        spk = np.array([1], np.int64)
        L = 682
        ids = rng.integers(4, 131, size=(1, L)).astype(np.int64)
        lens = np.array([L], np.int64)
        eng.set_precision("fp32")
        r = eng.acoustic(ids, lens, spk, want=("dur", "mel_lens", "pitch_idx", "energy_idx"))
        assert r["T"] == 6 * L == 4092 and r["pitch_idx"].shape == (1, 4092) and r["energy_idx"].shape == (1, 4092)
        mel, mel_post = eng.fetch_mel(1, r["T"])
        H = cfg["models"]["fastspeech2"]["encoder_hidden"]
        o = orc.AcousticOracle(ac, cfg, stats, var_pos_table=variance_position_table(4098, H))
        (omel, omel_post, odur), omel_lens = o.inference(spk, ids, lens)
        np.testing.assert_array_equal(r["dur"], odur)
        np.testing.assert_array_equal(r["mel_lens"], omel_lens)
        np.testing.assert_array_equal(r["pitch_idx"], o.trace["pitch_idx"])
        np.testing.assert_array_equal(r["energy_idx"], o.trace["energy_idx"])
        assert mean_l1(mel_post, omel_post) < MEL_L1
        L2 = 683
        ids2 = rng.integers(4, 131, size=(1, L2)).astype(np.int64)
        with pytest.raises(ValueError, match="position table"):
            eng.acoustic(ids2, np.array([L2], np.int64), spk, want=("mel_lens",))
    finally:
        eng.close()


def test_frame_level_outputs_are_taps_not_acoustic_outputs():
    """variance_embedding.*_feature "frame_level" (reference U/layers.py:249-257): the feature's index / prediction arrays have T columns, which
    the caller cannot size before the call -- e2etts_acoustic refuses a buffer for them and the taps hand them out afterwards."""
    import ctypes as C
    g = load_golden("tiny_pframe_b3")
    cfg, eng = engine_for(g, "tiny_pframe_b3")
    spk = np.array([int(g["speaker"])], np.int64)
    d, p, e = (float(x) for x in g["controls"])
    r = eng.acoustic(g["ids"], g["lens"], spk, d, p, e, want=("pitch_idx", "energy_idx", "pitch_pred", "energy_pred", "mel_lens"))
    B, L, T = g["ids"].shape[0], g["ids"].shape[1], r["T"]
    assert r["pitch_idx"].shape == (B, T) and r["pitch_pred"].shape == (B, T, 2)      # frame level
    assert r["energy_idx"].shape == (B, L) and r["energy_pred"].shape == (B, L)       # phoneme level
    np.testing.assert_array_equal(r["pitch_idx"], g["pitch_idx"])
    np.testing.assert_array_equal(r["energy_idx"], g["energy_idx"])
    ids, lens = np.ascontiguousarray(g["ids"]), np.ascontiguousarray(g["lens"])
    buf = np.zeros((B, L), np.int32)
    Tc = C.c_int(0)
    rc = eng.lib.e2etts_acoustic(eng._h, ids.ctypes.data, lens.ctypes.data, B, L, spk.ctypes.data, 1, d, p, e,
                                 None, None, C.byref(Tc), buf.ctypes.data, None, None, None, None)
    assert rc != 0 and b"frame_level" in eng.lib.e2etts_last_error(eng._h)
    wrong = np.zeros(5, np.int32)
    assert eng.lib.e2etts_fetch_tap_i32(eng._h, b"pitch_idx", wrong.ctypes.data, wrong.size) != 0
