#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own modules on CPU.

Runs only in the build container (needs /root/reference; the GPU box never sees
it).  The reference source is imported from where it lies and never copied:
fixtures hold inputs and the reference's outputs only.

Recipe: a stand-in ``numba`` module (identity ``jit``,
``prange = range``) is put on sys.path because U/function.py:2 imports numba for
two training-only functions; then ``models`` is imported from
/root/reference/e2e_tts.  Synthetic state dicts come from
e2e_tts_amd.synth_weights and are loaded with ``strict=True`` -- which proves
the manifest exact.

The host-loop methods (TTS.arrange_text / input_parse / combine_audio,
API/utils.py:64-117) live in a module whose import needs ffmpy/pydub/g2p and
instantiates an uploader; those three pure methods are therefore extracted
from the reference file with ``ast`` and executed as-is with a toy tokenizer.

Usage:  python oracle/make_goldens.py [--only NAME] [--skip-large]
"""
from __future__ import annotations

import argparse
import ast
import os
import sys
import tempfile
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from e2e_tts_amd import config as cfgmod  # noqa: E402
from e2e_tts_amd import synth_weights as sw  # noqa: E402
from oracle import ref_numpy as orc  # noqa: E402


def import_reference():
    tmp = tempfile.mkdtemp(prefix="numba_standin_")
    os.makedirs(os.path.join(tmp, "numba"))
    with open(os.path.join(tmp, "numba", "__init__.py"), "w") as f:
        f.write("def jit(*a, **k):\n"
                "    if len(a) == 1 and callable(a[0]) and not k: return a[0]\n"
                "    return lambda f: f\n"
                "prange = range\n")
    sys.path.insert(0, tmp)
    sys.path.insert(0, os.path.join(REF, "e2e_tts"))
    warnings.filterwarnings("ignore")
    import models  # noqa
    return models


def build_reference(models, config, stats, n_speakers, ac_state, voc_state):
    import torch
    m = models.UnsupervisedFastSpeech2(n_symbols=cfgmod.N_SYMBOLS, n_speakers=n_speakers,
                                       n_channels=config["audio"]["mel"]["channels"],
                                       config=config["models"]["fastspeech2"], stats=stats)
    m.load_state_dict(sw.to_torch(ac_state), strict=True)
    m.eval()
    v = models.HifiGan(config["models"]["hifigan"])
    v.load_state_dict(sw.to_torch(voc_state), strict=True)
    v.eval()
    torch.set_grad_enabled(False)
    return m, v


def run_reference(m, v, ids, lens, speaker, controls=(1.0, 1.0, 1.0), run_vocoder=True):
    """controls = (d, p, e).  Returns dict of numpy outputs incl. hooked intermediates."""
    import torch
    trace = {}
    hooks = []
    hooks.append(m.encoder.register_forward_hook(lambda mod, i, o: trace.__setitem__("enc_out", o[0].numpy().copy())))
    hooks.append(m.decoder.register_forward_hook(lambda mod, i, o: trace.__setitem__("dec_out", o[0].numpy().copy())))

    def va_hook(mod, i, o):
        x, log_d, dur, p_pred, e_pred, mel_lens, mel_mask, _ = o[0]
        trace.update(lr_out=x.numpy().copy(), log_d=log_d.numpy().copy(), pitch_pred=p_pred.numpy().copy(),
                     energy_pred=e_pred.numpy().copy())
    hooks.append(m.variance_adaptor.register_forward_hook(va_hook))
    hooks.append(m.variance_adaptor.pitch_embedding.register_forward_hook(
        lambda mod, i, o: trace.__setitem__("pitch_idx", i[0].numpy().copy())))
    hooks.append(m.variance_adaptor.energy_embedding.register_forward_hook(
        lambda mod, i, o: trace.__setitem__("energy_idx", i[0].numpy().copy())))
    t_ids = torch.from_numpy(np.asarray(ids, dtype=np.int64))
    t_lens = torch.from_numpy(np.asarray(lens, dtype=np.int64))
    t0 = time.time()
    (mel, mel_post, dur), mel_lens = m.inference(speaker=torch.tensor([speaker]), texts=t_ids, txt_lens=t_lens,
                                                 max_txt_len=int(t_ids.shape[1]),
                                                 d_control=controls[0], p_control=controls[1], e_control=controls[2])
    t_ac = time.time() - t0
    for h in hooks:
        h.remove()
    out = dict(trace)
    out.update(mel=mel.numpy(), mel_post=mel_post.numpy(), dur=dur.numpy(), mel_lens=mel_lens.numpy())
    if run_vocoder:
        t0 = time.time()
        wav = v(mel_post.transpose(1, 2)).squeeze(1)
        out["wav"] = wav.numpy()
        out["t_vocoder_s"] = time.time() - t0
    out["t_acoustic_s"] = t_ac
    return out


def margins(out, energy_bins, stats, controls=(1.0, 1.0, 1.0), lens=None, ve=None, pitch_bins=None):
    """Distance of every discrete decision to its rounding boundary (fp32 reference values).  `ve`: the config's variance_embedding
    section when it is not the shipped one (use_uv False: pitch bucketized on `pitch_bins`; pitch_quantization "log": f0 = 2 ** p)."""
    L = out["log_d"].shape[1]
    valid = np.arange(L)[None, :] < np.asarray(lens)[:, None]
    d = np.exp(out["log_d"].astype(np.float64)) - 1
    m_dur = np.abs((d - np.floor(d)) - 0.5)
    m_dur = np.where(valid, m_dur, 1.0)
    # frame_level features (U/layers.py:249-257) have T columns; EVERY row counts there -- the embedding is added on padded rows too,
    # and although the decoder masks those, the fixture stores their indices
    valid_p = valid if out["pitch_pred"].shape[1] == L and not (ve is not None and ve["pitch_feature"] == "frame_level") else np.ones(out["pitch_pred"].shape[:2], bool)
    valid_e = valid if out["energy_pred"].shape[1] == L and not (ve is not None and ve["energy_feature"] == "frame_level") else np.ones(out["energy_pred"].shape[:2], bool)
    p = out["pitch_pred"].astype(np.float64)  # already multiplied by p_control (U/layers.py:147)
    if ve is not None and not ve["use_uv"]:
        pb = np.asarray(pitch_bins, np.float64)
        p = p * controls[1]   # (this branch hands back the prediction BEFORE p_control, U/layers.py:156,162)
        m_f0 = np.abs(p[..., None] - pb[None, None, :]).min(axis=-1) / float(pb[1] - pb[0])   # in bucket units
        m_uv = np.ones_like(m_f0)
    else:
        f0 = np.power(2.0, p[..., 0]) if (ve is not None and ve["pitch_quantization"] == "log") else p[..., 0] * stats["f0"]["std"] + stats["f0"]["mean"]
        uv = p[..., 1] > 0
        m_uv = np.abs(p[..., 1])
        mel = 1127 * np.log(1 + np.maximum(f0, -699.0) / 700)
        b = np.where(mel > 0, (mel - orc.F0_MEL_MIN) * 254 / (orc.F0_MEL_MAX - orc.F0_MEL_MIN) + 1, mel)
        # idx = trunc(clip(b, 1, 255) + 0.5) is continuous at both clip points and at mel == 0,
        # so the only boundaries are the half-integers of b and the uv threshold.
        fb = np.clip(b, 1, 255) + 0.5
        m_f0 = np.where(uv, 1.0, np.minimum(fb - np.floor(fb), np.ceil(fb) - fb))
    m_f0 = np.where(valid_p, m_f0, 1.0)
    m_uv = np.where(valid_p, m_uv, 1.0)
    e = out["energy_pred"].astype(np.float64) * controls[2]
    m_en = np.abs(e[..., None] - energy_bins[None, None, :].astype(np.float64)).min(axis=-1)
    m_en = m_en / float(energy_bins[1] - energy_bins[0])  # in bucket units
    m_en = np.where(valid_e, m_en, 1.0)
    return dict(dur=float(m_dur.min()), uv=float(m_uv.min()), f0=float(m_f0.min()), energy=float(m_en.min()))


def oracle_margin(ac_oracle, ids, lens, speaker, stats, controls=(1.0, 1.0, 1.0)):
    """Cheap margin probe with the numpy oracle's encoder + variance adaptor only (for the seed search)."""
    ids = np.asarray(ids, np.int64)
    lens = np.asarray(lens, np.int64)
    pad = orc.get_mask_from_lengths(lens, ids.shape[1])
    x = ac_oracle.encoder(ids, pad) + ac_oracle.sd["speaker_emb.weight"][[speaker]][:, None, :]
    ve = ac_oracle.fs["variance"]["variance_embedding"]
    log_d = ac_oracle.duration_predictor(x, pad)
    xp = xe = x
    if ve["pitch_feature"] == "frame_level" or ve["energy_feature"] == "frame_level":   # those predictors read the regulator's output
        dur = np.maximum(np.round(np.exp(log_d) - np.float32(1)) * np.float32(controls[0]), np.float32(0))
        x_tmp = x
        if ve["pitch_feature"] != "frame_level":
            x_tmp = x_tmp + ac_oracle.pitch_embedding(x, controls[1])[2]
        if ve["energy_feature"] != "frame_level":
            x_tmp = x_tmp + ac_oracle.energy_embedding(x, controls[2])[2]
        xf, _ = ac_oracle.length_regulator(x_tmp, dur)
        xp = xf if ve["pitch_feature"] == "frame_level" else x
        xe = xf if ve["energy_feature"] == "frame_level" else x
    pp = ac_oracle.variance_predictor("pitch", xp)
    out = dict(log_d=log_d,
               pitch_pred=pp * np.float32(controls[1]) if ve["use_uv"] else pp[..., 0],
               energy_pred=ac_oracle.variance_predictor("energy", xe)[..., 0])
    return margins(out, ac_oracle.sd["variance_adaptor.energy_bins"], stats, controls, lens, ve, ac_oracle.sd["variance_adaptor.pitch_bins"])


def make_ids(seed, lens, L=None):
    """ids uniform in [4, 130], pad = 0."""
    rng = np.random.Generator(np.random.PCG64(seed))
    L = L or int(max(lens))
    ids = np.zeros((len(lens), L), dtype=np.int64)
    for b, n in enumerate(lens):
        ids[b, :n] = rng.integers(4, 131, size=n)
    return ids


def search_ids(ac_oracle, lens, speaker, stats, controls, want, max_tries, first_seed, fixed_durations):
    best = None
    for s in range(first_seed, first_seed + max_tries):
        ids = make_ids(s, lens)
        mg = oracle_margin(ac_oracle, ids, lens, speaker, stats, controls)
        keys = ("uv", "f0", "energy") if fixed_durations else ("dur", "uv", "f0", "energy")
        worst = min(mg[k] for k in keys)
        if best is None or worst > best[0]:
            best = (worst, s, ids, mg)
        if worst >= want:
            break
    print(f"    ids seed {best[1]}: min margin {best[0]:.2e} ({best[3]})", flush=True)
    return best[1], best[2]


def wav_digest(wav, lens_samples):
    """Strided / cropped view of a waveform batch + float64 checksums (keeps fixtures small)."""
    d = {}
    d["wav_stride"] = np.int64(16)
    d["wav_strided"] = wav[:, ::16].copy()
    d["wav_head"] = wav[:, :2048].copy()
    d["wav_sum"] = np.array([wav[b, :n].astype(np.float64).sum() for b, n in enumerate(lens_samples)])
    d["wav_abs_sum"] = np.array([np.abs(wav[b, :n].astype(np.float64)).sum() for b, n in enumerate(lens_samples)])
    return d


def save(name, **arrays):
    os.makedirs(GOLD, exist_ok=True)
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)", flush=True)


# --------------------------------------------------------------------------- cases

def case_model(models, name, config, mode, lens, speaker, controls, ids_seed, want_margin, store, w_seed=(1234, 4321),
               max_tries=40):
    print(f"[{name}]", flush=True)
    stats = cfgmod.DEFAULT_STATS
    n_spk = 4
    ac_state = sw.make_acoustic_state(config, stats, n_spk, seed=w_seed[0], mode=mode)
    voc_state = sw.make_vocoder_state(config, seed=w_seed[1])
    ac_or = orc.AcousticOracle(ac_state, config, stats)
    seed, ids = search_ids(ac_or, lens, speaker, stats, controls, want_margin, max_tries, ids_seed, mode == "fixed")
    m, v = build_reference(models, config, stats, n_spk, ac_state, voc_state)
    out = run_reference(m, v, ids, lens, speaker, controls)
    mg = margins(out, ac_state["variance_adaptor.energy_bins"], stats, controls, lens, config["models"]["fastspeech2"]["variance"]["variance_embedding"],
                 ac_state["variance_adaptor.pitch_bins"])
    print(f"    reference margins {mg}; T={out['mel'].shape[1]} acoustic {out['t_acoustic_s']:.2f}s vocoder {out['t_vocoder_s']:.2f}s", flush=True)
    hop = config["audio"]["stft"]["hop_length"]
    meta = dict(ids=ids, lens=np.asarray(lens, np.int64), speaker=np.int64(speaker),
                controls=np.asarray(controls, np.float64), ids_seed=np.int64(seed),
                weight_seeds=np.asarray(w_seed, np.int64), mode=np.array(mode),
                margin_dur=mg["dur"], margin_uv=mg["uv"], margin_f0=mg["f0"], margin_energy=mg["energy"],
                ref_t_acoustic_s=out["t_acoustic_s"], ref_t_vocoder_s=out["t_vocoder_s"])
    arrays = dict(meta)
    for k in ("dur", "mel_lens", "pitch_idx", "energy_idx", "log_d", "pitch_pred", "energy_pred"):
        arrays[k] = out[k]
    if store == "full":
        for k in ("enc_out", "lr_out", "dec_out", "mel", "mel_post", "wav"):
            arrays[k] = out[k]
    elif store == "medium":
        arrays["mel_post"] = out["mel_post"]
        arrays["mel"] = out["mel"]
        arrays.update(wav_digest(out["wav"], out["mel_lens"] * hop))
    else:  # "digest": a few utterances, strided frames
        sel = np.array(sorted({int(np.argmax(lens)), int(np.argmin(lens)), len(lens) // 2}))
        arrays["sel"] = sel
        arrays["mel_post_sel"] = out["mel_post"][sel][:, ::4].copy()
        arrays["mel_frame_stride"] = np.int64(4)
        arrays["mel_post_sum"] = np.array([out["mel_post"][b, :n].astype(np.float64).sum() for b, n in enumerate(out["mel_lens"])])
        arrays["mel_post_abs_sum"] = np.array([np.abs(out["mel_post"][b, :n].astype(np.float64)).sum() for b, n in enumerate(out["mel_lens"])])
        wd = wav_digest(out["wav"], out["mel_lens"] * hop)
        arrays["wav_strided_sel"] = out["wav"][sel][:, ::64].copy()
        arrays["wav_stride"] = np.int64(64)
        arrays["wav_sum"], arrays["wav_abs_sum"] = wd["wav_sum"], wd["wav_abs_sum"]
    save(name, **arrays)


def case_vocoder_micro(models):
    """Per-op vocoder fixtures on the tiny config: conv_pre, every upsampler, every ResBlock1, conv_post."""
    import torch
    import torch.nn.functional as F
    print("[voc_micro_tiny]", flush=True)
    config = cfgmod.tiny_config()
    voc_state = sw.make_vocoder_state(config, seed=4321)
    v = models.HifiGan(config["models"]["hifigan"])
    v.load_state_dict(sw.to_torch(voc_state), strict=True)
    v.eval()
    torch.set_grad_enabled(False)
    rng = np.random.Generator(np.random.PCG64(77))
    mel = rng.standard_normal((2, 80, 10)).astype(np.float32)
    arrays = dict(mel=mel)
    x = v.conv_pre(torch.from_numpy(mel))
    arrays["conv_pre"] = x.numpy().copy()
    for i in range(v.num_upsamples):
        x = F.leaky_relu(x, 0.1)
        x = v.ups[i](x)
        arrays[f"ups{i}"] = x.numpy().copy()
        xs = None
        for j in range(v.num_kernels):
            r = v.resblocks[i * v.num_kernels + j](x)
            arrays[f"rb{i * v.num_kernels + j}"] = r.numpy().copy()
            xs = r if xs is None else xs + r
        x = xs / v.num_kernels
        arrays[f"stage{i}"] = x.numpy().copy()
    x = torch.tanh(v.conv_post(F.leaky_relu(x)))
    arrays["wav"] = x.numpy().copy()
    # weight-norm fold: the effective weight the reference convolves with
    arrays["ups0_weight"] = v.ups[0].weight.detach().numpy().copy()
    arrays["conv_pre_weight"] = v.conv_pre.weight.detach().numpy().copy()
    save("voc_micro_tiny", **arrays)


def case_istft(models):
    """iSTFTNet generator (V/generator.py:65-113) + the inverse STFT of src/tools/stft.py:138-148.  ``inverse_stft`` is taken
    from the reference file with ``ast`` and executed as-is (the module itself imports librosa, which this image lacks)."""
    import torch
    print("[istft]", flush=True)
    src = open(os.path.join(REF, "e2e_tts", "src", "tools", "stft.py")).read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "inverse_stft"]
    ns = {"torch": torch}
    exec(compile(ast.Module(body=fn, type_ignores=[]), "reference:stft.py:inverse_stft", "exec"), ns)
    inverse_stft = ns["inverse_stft"]
    torch.set_grad_enabled(False)
    arrays = {}
    cases = (("tiny_rb2", cfgmod.tiny_config(), 1, (2, 12), 51), ("tiny_rb1", cfgmod.tiny_config(), "1", (1, 9), 52),
             ("full_rb2", cfgmod.default_config(), 1, (2, 10), 53))
    for tag, config, resblock, (B, T), seed in cases:
        config["models"]["istft"]["resblock"] = resblock
        hg = config["models"]["istft"]
        state = sw.make_vocoder_state(config, seed=4000 + seed, vocoder="istft")
        g = models.iSTFT(hg)
        g.load_state_dict(sw.to_torch(state), strict=True)   # proves the iSTFT state-dict manifest exact
        g.eval()
        mel = np.random.Generator(np.random.PCG64(seed)).standard_normal((B, 80, T)).astype(np.float32)
        spec, phase = g(torch.from_numpy(mel))
        wav = inverse_stft(spec, phase, n_fft=hg["gen_istft_n_fft"], hop_size=hg["gen_istft_hop_size"], win_size=hg["gen_istft_win_size"])
        assert wav.shape == (B, 1, T * 256), wav.shape
        arrays[f"{tag}.mel"] = mel
        arrays[f"{tag}.spec"] = spec.numpy().copy()
        arrays[f"{tag}.phase"] = phase.numpy().copy()
        arrays[f"{tag}.wav"] = wav.numpy().copy()
        arrays[f"{tag}.seed"] = np.array([4000 + seed])
        arrays[f"{tag}.resblock_is_str"] = np.array([isinstance(resblock, str)])
        o = orc.IstftOracle(state, config)
        s2, p2 = o.forward(mel)
        print(f"  {tag}: oracle vs reference spec {np.abs(s2 - arrays[f'{tag}.spec']).max():.2e} phase {np.abs(p2 - arrays[f'{tag}.phase']).max():.2e} "
              f"wav {np.abs(o.inverse(s2, p2) - arrays[f'{tag}.wav']).max():.2e}", flush=True)
    save("istft", **arrays)


def case_tables(models):
    """Both sinusoid tables as the reference builds them (U/blocks/utils.py:14-34, U/sublayers.py:28-44)."""
    print("[tables]", flush=True)
    from models.acoustic.unsupervised_fastspeech2.blocks.utils import get_sinusoid_encoding_table
    from models.acoustic.unsupervised_fastspeech2.sublayers import SinusoidalPositionalEmbedding
    from models.acoustic.unsupervised_fastspeech2.function import make_positions
    import torch
    t1 = get_sinusoid_encoding_table(300, 384).numpy()
    t1b = get_sinusoid_encoding_table(1300, 64).numpy()
    t2 = SinusoidalPositionalEmbedding.get_embedding(2048, 384, 0).numpy()
    t2b = SinusoidalPositionalEmbedding.get_embedding(600, 64, 0).numpy()
    x0 = np.array([[0.5, 0.0, -1.0, 2.0, 0.0, 0.0, 3.0], [0.0, 0.0, 1.0, 1.0, 1.0, 0.0, 0.0]], np.float32)
    pos = make_positions(torch.from_numpy(x0), 0).numpy()
    save("tables", fft_384_rows300=t1, fft_64_rows1300_tail=t1b[1000:], var_384_rows2048_stride8=t2[::8].copy(),
         var_64_rows600=t2b, positions_in=x0, positions_out=pos)


def case_host_loop():
    """arrange_text / input_parse / combine_audio run from the reference source (API/utils.py:64-117)."""
    print("[host_loop]", flush=True)
    import torch
    import torch.nn as nn
    src = open(os.path.join(REF, "e2e_tts/src/api/utils.py")).read()
    tree = ast.parse(src)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "TTS")
    keep = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in ("arrange_text", "input_parse", "combine_audio")]
    mod = ast.Module(body=[ast.ClassDef(name="HostLoop", bases=[], keywords=[], body=keep, decorator_list=[])], type_ignores=[])
    ast.fix_missing_locations(mod)

    def toy_tokenizer(txt):  # stand-in for g2p.text_to_sequence: one id per character
        return [4 + (ord(c) % 127) for c in txt]

    ns = {"torch": torch, "nn": nn, "np": np, "text_to_sequence": toy_tokenizer}
    exec(compile(mod, "<reference API/utils.py TTS methods>", "exec"), ns)
    hl = ns["HostLoop"]()
    hl.max_len, hl.hop_length, hl.max_wav_value = 300, 256, 32768.0
    rng = np.random.Generator(np.random.PCG64(5))
    words = ["xin", "chao", "viet", "nam", "tieng", "noi", "tong", "hop", "am", "thanh"]

    def sentence(n):
        return " ".join(words[int(i)] for i in rng.integers(0, len(words), n))

    texts = [sentence(6), " , ".join(sentence(int(rng.integers(8, 30))) for _ in range(9)), sentence(40), sentence(3),
             " , ".join(sentence(12) for _ in range(5)), sentence(70), sentence(1)]
    arranged = hl.arrange_text(list(texts))
    batches, revert = hl.input_parse(list(texts))
    arrays = dict(texts=np.array(texts), arranged=np.array(arranged), revert=revert.numpy(), n_batches=np.int64(len(batches)))
    for i, (ids, lens) in enumerate(batches):
        arrays[f"ids{i}"] = ids.numpy()
        arrays[f"lens{i}"] = lens.numpy()
    # a second, budget-stressing length list through the batching arithmetic only
    lens_list = [120, 119, 100, 90, 90, 61, 60, 33, 30, 30, 12, 7, 300, 299, 150, 150, 1]
    fake = ["a" * n for n in lens_list]
    hl2 = ns["HostLoop"]()
    hl2.max_len = 300
    hl2.arrange_text = lambda t: t
    b2, r2 = hl2.input_parse(fake)
    arrays["stress_lens"] = np.asarray(lens_list, np.int64)
    arrays["stress_revert"] = r2.numpy()
    arrays["stress_batch_sizes"] = np.asarray([len(l) for _, l in b2], np.int64)
    arrays["stress_batch_first_len"] = np.asarray([int(l[0]) for _, l in b2], np.int64)
    # combine_audio
    audios = [rng.uniform(-1.2, 1.2, 256 * 9).astype(np.float32), rng.uniform(-1, 1, 256 * 9).astype(np.float32),
              np.array([0.99999, -0.99999, 1.0, -1.0, 0.5 / 32768, -0.5 / 32768, 1.5 / 32768] + [0.0] * (256 * 9 - 7), np.float32)]
    lengths = [torch.tensor(7), torch.tensor(9), torch.tensor(2)]
    pcm = hl.combine_audio(audios, lengths, 1000)
    arrays.update(ca_audio0=audios[0], ca_audio1=audios[1], ca_audio2=audios[2], ca_lengths=np.array([7, 9, 2], np.int64),
                  ca_distance=np.int64(1000), ca_pcm=pcm)
    save("host_loop", **arrays)


def case_g2p():
    """Vietnamese grapheme -> phoneme front-end (reference e2e_tts/models/g2p/g2p.py:58-176, symbols.py:19-50,
    __init__.py:5-31), run from the reference source.  Two absent third-party imports get stand-ins: `g2p_en.G2p`
    (English fallback, never reached: dict/foreign_words.json is empty) and `unidecode.unidecode`, for which NFD
    decomposition minus combining marks (+ d-stroke -> d) is used -- identical to unidecode on the Vietnamese alphabet.
    `text_to_sequence` itself cannot run as shipped (cleaners.py:12,26-30 shadows the import and recurses), so the
    fixture calls what it evidently intends: normalize_phonemes(text.lower(), is_training=False) + the symbol -> id map."""
    print("[g2p]", flush=True)
    import importlib
    import types
    import unicodedata
    g2p_en = types.ModuleType("g2p_en")
    g2p_en.G2p = type("G2p", (), {"__call__": lambda self, t: []})
    ud = types.ModuleType("unidecode")

    def unidecode(x):
        x = x.replace("\u0111", "d").replace("\u0110", "D")
        return "".join(c for c in unicodedata.normalize("NFD", x) if not unicodedata.combining(c))
    ud.unidecode = unidecode
    sys.modules["g2p_en"], sys.modules["unidecode"] = g2p_en, ud
    sys.path.insert(0, os.path.join(REF, "e2e_tts", "models"))
    ref_g2p = importlib.import_module("g2p.g2p")
    ref_sym = importlib.import_module("g2p.symbols")
    symbols = list(ref_sym.symbols)
    sym2id = {s: i for i, s in enumerate(symbols)}
    words = [w for w in ref_g2p.vn_words]
    # every word of dict/fix_words.txt the reference's converter accepts (17 977 entries; round 2 sampled 1 500)
    pick = range(len(words))
    sample, phon = [], []
    skipped = 0
    for i in pick:
        try:
            ph = ref_g2p.vi_convert(words[i])
        except Exception:   # a handful of dictionary entries are not single syllables the converter accepts
            skipped += 1
            continue
        sample.append(words[i])
        phon.append(" ".join(ph))
    print(f"    {len(sample)} words ({skipped} raise in the reference and are left out)")
    sentences = [
        "xin ch\u00e0o vi\u1ec7t nam",
        "h\u00f4m nay tr\u1eddi \u0111\u1eb9p , ch\u00fang ta \u0111i ch\u01a1i nh\u00e9 .",
        "qu\u1ed1c gia , gi\u00e1o d\u1ee5c v\u00e0 khoa-h\u1ecdc c\u00f4ng-ngh\u1ec7 ?",
        "ngh\u1ec7 thu\u1eadt t\u1ed5ng h\u1ee3p ti\u1ebfng n\u00f3i",
        "u\u1ed1ng n\u01b0\u1edbc nh\u1edb ngu\u1ed3n !",
        "kh\u00f4ng-gian th\u1eddi-gian",
        "a",
    ]
    import contextlib
    import io
    sent_ids, sent_ph = [], []
    for t in sentences:
        with contextlib.redirect_stdout(io.StringIO()):
            seq, boundaries = ref_g2p.normalize_phonemes(t.lower(), is_training=False)
        sent_ph.append(" ".join(seq))
        sent_ids.append(np.array([sym2id[w[:-1] if w.startswith("@") and w[-1].isdigit() else w] for w in seq], np.int64))
    arrays = dict(symbols=np.array(symbols), words=np.array(sample), phonemes=np.array(phon), sentences=np.array(sentences),
                  sentence_phonemes=np.array(sent_ph))
    for i, ids in enumerate(sent_ids):
        arrays[f"ids{i}"] = ids
    save("g2p", **arrays)


def case_hifigan48k(models):
    """BASELINE config 5's generator -- upsample_rates [8, 8, 4, 2], kernels [16, 16, 8, 4] (hop 512) -- instantiated from the reference's
    config-driven HifiGan class (V/generator.py:14-35) at widths 64 and 512, T = 90 frames of random mel.  The reference ships no
    48 kHz yaml; the class is what pins the arithmetic of this configuration."""
    import torch
    print("[hifigan_48k]", flush=True)
    torch.set_grad_enabled(False)
    arrays = {}
    for tag, width, B, T, wseed, mseed in (("w64", 64, 2, 90, 31, 5), ("w512", 512, 1, 90, 34, 8)):
        config = cfgmod.default_config()
        config["models"]["hifigan"].update(upsample_rates=[8, 8, 4, 2], upsample_kernel_sizes=[16, 16, 8, 4], upsample_initial_channel=width)
        config["audio"]["stft"]["hop_length"] = 512
        config["audio"]["signal"]["sampling_rate"] = 48000
        state = sw.make_vocoder_state(config, seed=wseed)
        v = models.HifiGan(config["models"]["hifigan"])
        v.load_state_dict(sw.to_torch(state), strict=True)
        v.eval()
        mel = np.random.Generator(np.random.PCG64(mseed)).standard_normal((B, T, 80)).astype(np.float32)   # channels-last, as the engine takes it
        wav = v(torch.from_numpy(np.ascontiguousarray(mel.transpose(0, 2, 1)))).squeeze(1).numpy()
        assert wav.shape == (B, T * 512), wav.shape
        arrays[f"{tag}.mel"] = mel
        arrays[f"{tag}.wav"] = wav.copy()
        # The reference's own class run in bfloat16 (module.bfloat16() on a bf16 mel: weight_g / weight_v, every activation and every
        # layer output in bf16, torch-CPU kernels): what "config 5's arithmetic" is when the REFERENCE does it.  Its distance from the
        # reference's fp32 output is the yardstick for the engine's plain-bf16 mode (tests/test_gpu_longform.py).
        import copy
        v16 = copy.deepcopy(v).bfloat16()
        wav16 = v16(torch.from_numpy(np.ascontiguousarray(mel.transpose(0, 2, 1))).bfloat16()).squeeze(1).float().numpy()
        assert wav16.shape == wav.shape and np.isfinite(wav16).all()
        arrays[f"{tag}.wav_ref_bf16"] = wav16.copy()
        arrays[f"{tag}.ref_bf16_mean_l1"] = np.float64(np.abs(wav16.astype(np.float64) - wav.astype(np.float64)).mean())
        print(f"    {tag}: reference in bf16 vs reference in fp32: wav mean-L1 {float(arrays[f'{tag}.ref_bf16_mean_l1']):.3e}", flush=True)
        arrays[f"{tag}.width"] = np.int64(width)
        arrays[f"{tag}.weight_seed"] = np.int64(wseed)
        print(f"    {tag}: wav {wav.shape} |wav| mean {np.abs(wav).mean():.3e}", flush=True)
    save("hifigan_48k", **arrays)


def case_bench_b32(models):
    """The headline workload itself (bench.py): B = 32, L = 128 phonemes each, "fixed" weights 1234 / 4321 -> T = 768.  Row 0 holds the
    ids of c2_latency (the B = 1 latency case) so that one utterance is pinned both alone and inside the batch; rows 1..31 come from a
    seed search like c3_mixed's.  Stored with c3_mixed's digest scheme."""
    print("[bench_b32]", flush=True)
    c2 = np.load(os.path.join(GOLD, "c2_latency.npz"))
    config, stats, n_spk, speaker, controls = cfgmod.default_config(), cfgmod.DEFAULT_STATS, 4, 1, (1.0, 1.0, 1.0)
    ac_state = sw.make_acoustic_state(config, stats, n_spk, seed=1234, mode="fixed")
    voc_state = sw.make_vocoder_state(config, seed=4321)
    ac_or = orc.AcousticOracle(ac_state, config, stats)
    lens = [128] * 32
    best = None
    for seed in range(3000, 3040):
        ids = np.concatenate([c2["ids"], make_ids(seed, lens[1:])], axis=0)
        mg = oracle_margin(ac_or, ids, lens, speaker, stats, controls)
        worst = min(mg[k] for k in ("uv", "f0", "energy"))
        if best is None or worst > best[0]:
            best = (worst, seed, ids)
        if worst >= 1e-4:
            break
    print(f"    ids seed {best[1]}: min margin {best[0]:.2e}", flush=True)
    seed, ids = best[1], best[2]
    m, v = build_reference(models, config, stats, n_spk, ac_state, voc_state)
    out = run_reference(m, v, ids, lens, speaker, controls)
    mg = margins(out, ac_state["variance_adaptor.energy_bins"], stats, controls, lens, config["models"]["fastspeech2"]["variance"]["variance_embedding"],
                 ac_state["variance_adaptor.pitch_bins"])
    print(f"    reference margins {mg}; T={out['mel'].shape[1]} acoustic {out['t_acoustic_s']:.2f}s vocoder {out['t_vocoder_s']:.2f}s", flush=True)
    hop = config["audio"]["stft"]["hop_length"]
    arrays = dict(ids=ids, lens=np.asarray(lens, np.int64), speaker=np.int64(speaker), controls=np.asarray(controls, np.float64),
                  ids_seed=np.int64(seed), weight_seeds=np.asarray((1234, 4321), np.int64), mode=np.array("fixed"),
                  margin_dur=mg["dur"], margin_uv=mg["uv"], margin_f0=mg["f0"], margin_energy=mg["energy"],
                  ref_t_acoustic_s=out["t_acoustic_s"], ref_t_vocoder_s=out["t_vocoder_s"])
    for k in ("dur", "mel_lens", "pitch_idx", "energy_idx", "log_d", "pitch_pred", "energy_pred"):
        arrays[k] = out[k]
    sel = np.array([0, 15, 31])
    arrays["sel"] = sel
    arrays["mel_post_sel"] = out["mel_post"][sel][:, ::4].copy()
    arrays["mel_frame_stride"] = np.int64(4)
    arrays["mel_post_sum"] = np.array([out["mel_post"][b, :n].astype(np.float64).sum() for b, n in enumerate(out["mel_lens"])])
    arrays["mel_post_abs_sum"] = np.array([np.abs(out["mel_post"][b, :n].astype(np.float64)).sum() for b, n in enumerate(out["mel_lens"])])
    wd = wav_digest(out["wav"], out["mel_lens"] * hop)
    arrays["wav_strided_sel"] = out["wav"][sel][:, ::64].copy()
    arrays["wav_stride"] = np.int64(64)
    arrays["wav_sum"], arrays["wav_abs_sum"] = wd["wav_sum"], wd["wav_abs_sum"]
    # int16 PCM of the selected utterances as TTS.combine_audio forms it (x 32768, truncation; API/utils.py:111-117), strided
    arrays["pcm_strided_sel"] = (out["wav"][sel][:, ::64] * np.float32(32768.0)).astype(np.int16)
    save("bench_b32", **arrays)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--skip-large", action="store_true")
    args = ap.parse_args()
    models = import_reference()
    tiny, full = cfgmod.tiny_config(), cfgmod.default_config()
    tiny_cf, full_cf = cfgmod.tiny_config(), cfgmod.default_config()   # block_type "conformer" (U/model.py:26-27)
    for c in (tiny_cf, full_cf):
        c["models"]["fastspeech2"]["building_block"]["block_type"] = "conformer"
    tiny_hv, tiny_cf_hv = hv_variant(cfgmod.tiny_config()), hv_variant(cfgmod.tiny_config())
    tiny_cf_hv["models"]["fastspeech2"]["building_block"]["block_type"] = "conformer"
    jobs = {
        "tables": lambda: case_tables(models),
        "host_loop": case_host_loop,
        "g2p": case_g2p,
        "voc_micro_tiny": lambda: case_vocoder_micro(models),
        "istft": lambda: case_istft(models),
        "hifigan_48k": lambda: case_hifigan48k(models),
        "tiny_b3": lambda: case_model(models, "tiny_b3", tiny, "varied", [23, 17, 9], 1, (1.0, 1.0, 1.0), 100, 2e-3, "full"),
        "tiny_long": lambda: case_model(models, "tiny_long", tiny, "varied", [70, 33], 2, (1.0, 1.0, 1.0), 200, 2e-3, "full"),
        "tiny_ctl": lambda: case_model(models, "tiny_ctl", tiny, "varied", [12, 30, 30, 5], 0, (1.3, 0.9, 1.1), 300, 2e-3, "full"),
        "tiny_b1": lambda: case_model(models, "tiny_b1", tiny, "varied", [1], 3, (1.0, 1.0, 1.0), 400, 2e-3, "full"),
        # Conformer blocks: tiny_cf_long runs past max_seq_len = 60 (regenerated tables in encoder, decoder and every MHSA module)
        "tiny_cf_b3": lambda: case_model(models, "tiny_cf_b3", tiny_cf, "varied", [23, 17, 9], 1, (1.0, 1.0, 1.0), 600, 2e-3, "full"),
        "tiny_cf_long": lambda: case_model(models, "tiny_cf_long", tiny_cf, "varied", [70, 33], 2, (1.1, 0.9, 1.2), 700, 2e-3, "full"),
        "full_cf_b2": lambda: case_model(models, "full_cf_b2", full_cf, "varied", [40, 27], 1, (1.0, 1.0, 1.0), 800, 1e-3, "medium"),
        "tiny_hv_b3": lambda: case_model(models, "tiny_hv_b3", tiny_hv, "varied", [21, 13, 30], 1, (1.0, 1.0, 1.0), 1100, 2e-3, "full"),
        "tiny_cf_hv_b3": lambda: case_model(models, "tiny_cf_hv_b3", tiny_cf_hv, "varied", [21, 13, 30], 2, (1.0, 1.0, 1.0), 1200, 2e-3, "full"),
        "tiny_nouv_b3": lambda: case_model(models, "tiny_nouv_b3", pv_variant(cfgmod.tiny_config(), "nouv"), "varied", [19, 26, 8], 1, (1.0, 1.1, 0.9), 1300, 2e-3, "full"),
        "tiny_plog_b3": lambda: case_model(models, "tiny_plog_b3", pv_variant(cfgmod.tiny_config(), "plog"), "varied", [19, 26, 8], 3, (1.0, 1.0, 1.0), 1400, 2e-3, "full"),
        "tiny_lpad_b3": lambda: case_model(models, "tiny_lpad_b3", pv_variant(cfgmod.tiny_config(), "lpad"), "varied", [17, 28, 5], 2, (1.0, 1.0, 1.0), 1500, 2e-3, "full"),
        "tiny_frame_b3": lambda: case_model(models, "tiny_frame_b3", pv_variant(cfgmod.tiny_config(), "frame"), "varied", [14, 22, 6], 1, (1.0, 1.0, 1.0), 1600, 1e-3, "full", max_tries=80),
        # energy alone at the frame level (pitch stays on the phonemes), and frame-level features under Conformer blocks (whose
        # unmasked attention reads the embeddings added on padded rows)
        "tiny_eframe_b3": lambda: case_model(models, "tiny_eframe_b3", pv_variant(cfgmod.tiny_config(), "eframe"), "varied", [14, 22, 6], 2, (1.0, 0.95, 1.05), 1900, 1e-3, "full", max_tries=80),
        "tiny_cf_frame_b3": lambda: case_model(models, "tiny_cf_frame_b3", pv_variant(cf_tiny(), "frame"), "varied", [14, 22, 6], 1, (1.0, 1.0, 1.0), 2000, 1e-3, "full", max_tries=80),
        "tiny_pframe_b3": lambda: case_model(models, "tiny_pframe_b3", pv_variant(cfgmod.tiny_config(), "pframe"), "varied", [14, 22, 6], 0, (1.0, 1.05, 0.95), 1700, 1e-3, "full", max_tries=80),
        # the frame-level features at FULL dimensions (384 hidden, 256-channel predictors over T rows)
        "full_frame_b2": lambda: case_model(models, "full_frame_b2", pv_variant(cfgmod.default_config(), "frame"), "varied", [40, 27], 1, (1.0, 1.0, 1.0), 1800, 1e-3, "medium", max_tries=80),
        "c1_plumbing": lambda: case_model(models, "c1_plumbing", full, "varied", [40], 1, (1.0, 1.0, 1.0), 1, 1e-3, "medium"),
        "full_b3": lambda: case_model(models, "full_b3", full, "varied", [48, 31, 20], 1, (1.0, 1.0, 1.0), 500, 1e-3, "medium"),
    }
    large = {
        "c2_latency": lambda: case_model(models, "c2_latency", full, "fixed", [128], 1, (1.0, 1.0, 1.0), 1, 1e-3, "medium"),
        "c3_mixed": lambda: case_model(models, "c3_mixed", full, "fixed", c3_lengths(), 1, (1.0, 1.0, 1.0), 2, 1e-4, "digest", max_tries=60),
        "bench_b32": lambda: case_bench_b32(models),   # after c2_latency: its row 0 is that fixture's utterance
        # one long utterance next to a short one: T = 3 072 frames = 12 key segments of the attention kernels (c3_mixed has 5), a position
        # table regenerated for 3 x max_seq_len, ragged limits that leave 93 % of the second row untouched
        "full_long": lambda: case_model(models, "full_long", full, "fixed", [512, 37], 2, (1.0, 1.0, 1.0), 900, 1e-4, "digest", max_tries=20),
    }
    if not args.skip_large:
        jobs.update(large)
    for name, fn in jobs.items():
        if args.only and name != args.only:
            continue
        fn()


def cf_tiny():
    cfg = cfgmod.tiny_config()
    cfg["models"]["fastspeech2"]["building_block"]["block_type"] = "conformer"
    return cfg


def hv_variant(cfg):
    """Fixtures `*_hv_*`: decoder_head != encoder_head (U/blocks/transformer.py:29,105, conformer.py:31,108) and an energy predictor whose
    depth / kernel differ from the pitch predictor's (U/layers.py:54-58,92-96) -- knobs the reference's config has and its shipped yaml leaves equal."""
    fs = cfg["models"]["fastspeech2"]
    bb = fs["building_block"]
    bb["transformer"].update(encoder_head=2, decoder_head=1)     # hidden 64: head dims 32 / 64
    bb["conformer"].update(encoder_head=4, decoder_head=2)       # head dims 16 / 32
    fs["variance"]["variance_predictor"].update(ener_predictor_layers=3, ener_predictor_kernel=3)
    return cfg


def pv_variant(cfg, which):
    """Fixtures `*_nouv_*` / `*_plog_*`: variance_embedding.use_uv False (one pitch output bucketized on pitch_bins, f0_bins embedding rows,
    U/layers.py:60-63,136,155-157) and pitch_quantization "log" with use_uv (f0 = 2 ** prediction, U/layers.py:148-149)."""
    ve = cfg["models"]["fastspeech2"]["variance"]["variance_embedding"]
    if which == "nouv":
        ve["use_uv"] = False
    elif which == "plog":
        ve["pitch_quantization"] = "log"
    elif which == "lpad":   # variance_predictor.ffn_padding "LEFT" -- causal predictor convolutions (U/layers.py:400-402,479-481)
        cfg["models"]["fastspeech2"]["variance"]["variance_predictor"]["ffn_padding"] = "LEFT"
    elif which == "frame":  # both features at the frame level (U/layers.py:249-257)
        ve["pitch_feature"] = ve["energy_feature"] = "frame_level"
    elif which == "eframe":  # energy at the frame level, pitch on the phonemes
        ve["energy_feature"] = "frame_level"
    else:                   # "pframe": pitch at the frame level, energy on the phonemes
        ve["pitch_feature"] = "frame_level"
    return cfg


def c3_lengths():
    """B = 32, lengths linspace(40, 200, 32) rounded, shuffled with PCG64(2)."""
    lens = np.round(np.linspace(40, 200, 32)).astype(np.int64)
    np.random.Generator(np.random.PCG64(2)).shuffle(lens)
    return [int(x) for x in lens]


if __name__ == "__main__":
    main()
