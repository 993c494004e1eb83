"""Deterministic synthetic checkpoints for tests and benchmarks.

No trained checkpoint ships with the reference (its .gitignore excludes *.pt),
so parity and throughput are measured on synthetic weights.
The *manifest* below is this project's own statement of the reference
state-dict layout (SURVEY.md 8(a) "State-dict manifest"); oracle/make_goldens.py
proves it exact by ``load_state_dict(strict=True)`` into the reference modules.

Generator: numpy ``Generator(PCG64(seed))``, tensors drawn in sorted-key order.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np

from .config import N_SYMBOLS

Shape = Tuple[int, ...]


def sinusoid_table(n_position: int, d_hid: int) -> np.ndarray:
    """FFT-block position table: float64 numpy, sin on even / cos on odd columns, cast to fp32.

    Restates reference U/blocks/utils.py:14-34 in vectorised form (same float64
    operations: ``pos / 10000 ** (2 * (j // 2) / d_hid)`` then sin / cos).
    """
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid)
    denom = np.power(10000, 2 * (j // 2) / d_hid)
    table = pos / denom[None, :]
    table[:, 0::2] = np.sin(table[:, 0::2])
    table[:, 1::2] = np.cos(table[:, 1::2])
    return table.astype(np.float32)


def acoustic_manifest(config: dict, n_speakers: int, n_symbols: int = N_SYMBOLS) -> "OrderedDict[str, Tuple[Shape, str]]":
    """name -> (shape, kind) for UnsupervisedFastSpeech2.state_dict()."""
    fs = config["models"]["fastspeech2"]
    bt = fs["building_block"]["block_type"]
    tr = fs["building_block"][bt]
    H = fs["encoder_hidden"]
    if bt == "conformer":
        F, k1, k2 = H * tr["ffn_expansion_factor"], tr["conv_kernel_size"], 1
    else:
        F = tr["conv_filter_size"]
        k1, k2 = tr["conv_kernel_size"]
    n_mel = config["audio"]["mel"]["channels"]
    vp = fs["variance"]["variance_predictor"]
    ve = fs["variance"]["variance_embedding"]
    pn = fs["postnet"]
    m: "OrderedDict[str, Tuple[Shape, str]]" = OrderedDict()

    def add(name, shape, kind):
        m[name] = (tuple(shape), kind)

    for side, layers in (("encoder", fs["encoder_layers"]), ("decoder", fs["decoder_layers"])):
        add(f"{side}.position_enc", (1, fs["max_seq_len"] + 1, H), "posenc")
        if side == "encoder":
            add("encoder.src_word_emb.weight", (n_symbols + 1, H), "emb_pad0")
        for l in range(layers if bt == "conformer" else 0):
            # ConformerBlock.state_dict() (U/blocks/conformer.py:214-249): FFN, rel-pos MHSA, conv module, FFN, LayerNorm
            p = f"{side}.layer_stack.{l}.sequential"
            nh = tr[f"{side}_head"]
            for i in (0, 3):
                q = f"{p}.{i}.module.sequential"
                add(f"{q}.0.weight", (H,), "gamma")
                add(f"{q}.0.bias", (H,), "beta")
                add(f"{q}.1.linear.weight", (F, H), "w")
                add(f"{q}.1.linear.bias", (F,), "b")
                add(f"{q}.4.linear.weight", (H, F), "w")
                add(f"{q}.4.linear.bias", (H,), "b")
            q = f"{p}.1.module"
            add(f"{q}.positional_encoding", (1, fs["max_seq_len"] + 1, H), "posenc")  # the side's table, registered again (:330)
            add(f"{q}.layer_norm.weight", (H,), "gamma")
            add(f"{q}.layer_norm.bias", (H,), "beta")
            add(f"{q}.attention.u_bias", (nh, H // nh), "emb")
            add(f"{q}.attention.v_bias", (nh, H // nh), "emb")
            for w in ("query_proj", "key_proj", "value_proj", "pos_proj", "out_proj"):
                add(f"{q}.attention.{w}.linear.weight", (H, H), "w")  # LinearNorm default: no bias (U/blocks/utils.py:182)
            q = f"{p}.2.module.sequential"
            add(f"{q}.0.weight", (H,), "gamma")
            add(f"{q}.0.bias", (H,), "beta")
            add(f"{q}.2.conv.weight", (2 * H, H, 1), "w")
            add(f"{q}.2.conv.bias", (2 * H,), "b")
            add(f"{q}.4.conv.weight", (H, 1, k1), "w")
            add(f"{q}.5.weight", (H,), "gamma")
            add(f"{q}.5.bias", (H,), "beta")
            add(f"{q}.5.running_mean", (H,), "bn_mean")
            add(f"{q}.5.running_var", (H,), "bn_var")
            add(f"{q}.5.num_batches_tracked", (), "i64")
            add(f"{q}.7.conv.weight", (H, H, 1), "w")
            add(f"{q}.7.conv.bias", (H,), "b")
            add(f"{p}.4.weight", (H,), "gamma")
            add(f"{p}.4.bias", (H,), "beta")
        for l in range(0 if bt == "conformer" else layers):
            p = f"{side}.layer_stack.{l}"
            for w in ("w_qs", "w_ks", "w_vs", "fc"):
                add(f"{p}.slf_attn.{w}.weight", (H, H), "w")
                add(f"{p}.slf_attn.{w}.bias", (H,), "b")
            add(f"{p}.slf_attn.layer_norm.weight", (H,), "gamma")
            add(f"{p}.slf_attn.layer_norm.bias", (H,), "beta")
            add(f"{p}.pos_ffn.w_1.weight", (F, H, k1), "w")
            add(f"{p}.pos_ffn.w_1.bias", (F,), "b")
            add(f"{p}.pos_ffn.w_2.weight", (H, F, k2), "w")
            add(f"{p}.pos_ffn.w_2.bias", (H,), "b")
            add(f"{p}.pos_ffn.layer_norm.weight", (H,), "gamma")
            add(f"{p}.pos_ffn.layer_norm.bias", (H,), "beta")
    va = "variance_adaptor"
    add(f"{va}.pitch_bins", (ve["n_bins"] - 1,), "pitch_bins")
    add(f"{va}.energy_bins", (ve["n_bins"] - 1,), "energy_bins")
    # aligner: training only, unused at inference, but part of the state dict (U/layers.py:275-369)
    add(f"{va}.aligner.key_proj.0.conv.weight", (2 * H, H, 3), "w")
    add(f"{va}.aligner.key_proj.0.conv.bias", (2 * H,), "b")
    add(f"{va}.aligner.key_proj.2.conv.weight", (n_mel, 2 * H, 1), "w")
    add(f"{va}.aligner.key_proj.2.conv.bias", (n_mel,), "b")
    add(f"{va}.aligner.query_proj.0.conv.weight", (2 * n_mel, n_mel, 3), "w")
    add(f"{va}.aligner.query_proj.0.conv.bias", (2 * n_mel,), "b")
    add(f"{va}.aligner.query_proj.2.conv.weight", (n_mel, 2 * n_mel, 1), "w")
    add(f"{va}.aligner.query_proj.2.conv.bias", (n_mel,), "b")
    add(f"{va}.aligner.query_proj.4.conv.weight", (n_mel, n_mel, 1), "w")
    add(f"{va}.aligner.query_proj.4.conv.bias", (n_mel,), "b")
    add(f"{va}.aligner.key_spk_proj.linear.weight", (H, H), "w")
    add(f"{va}.aligner.query_spk_proj.linear.weight", (n_mel, H), "w")
    dc = n_mel  # duration predictor width = mel channels (U/layers.py:39)
    for i in range(vp["dur_predictor_layers"]):
        cin = H if i == 0 else dc
        add(f"{va}.duration_predictor.conv.{i}.1.weight", (dc, cin, vp["dur_predictor_kernel"]), "w")
        add(f"{va}.duration_predictor.conv.{i}.1.bias", (dc,), "b")
        add(f"{va}.duration_predictor.conv.{i}.3.weight", (dc,), "gamma")
        add(f"{va}.duration_predictor.conv.{i}.3.bias", (dc,), "beta")
    add(f"{va}.duration_predictor.linear.weight", (1, dc), "dur_w")
    add(f"{va}.duration_predictor.linear.bias", (1,), "dur_b")
    vc = vp["filter_size"]
    for which, odim, layers, kern in (("pitch", 2 if ve["use_uv"] else 1, vp["pit_predictor_layers"], vp["pit_predictor_kernel"]),
                                      ("energy", 1, vp["ener_predictor_layers"], vp["ener_predictor_kernel"])):
        p = f"{va}.{which}_predictor"
        add(f"{p}.pos_embed_alpha", (1,), "alpha")
        for i in range(layers):
            cin = H if i == 0 else vc
            add(f"{p}.conv.{i}.1.weight", (vc, cin, kern), "w")
            add(f"{p}.conv.{i}.1.bias", (vc,), "b")
            add(f"{p}.conv.{i}.3.weight", (vc,), "gamma")
            add(f"{p}.conv.{i}.3.bias", (vc,), "beta")
        pb = "pitch_b" if ve["use_uv"] and ve["pitch_quantization"] != "log" else ("pitch_b_log" if ve["use_uv"] else "pitch_b_nouv")
        add(f"{p}.linear.weight", (odim, vc), f"{which}_w")
        add(f"{p}.linear.bias", (odim,), pb if which == "pitch" else f"{which}_b")
        add(f"{p}.embed_positions._float_tensor", (1,), "zero")
        rows = ve["n_bins"] if (which == "energy" or ve["use_uv"]) else ve["f0_bins"]   # U/layers.py:60-63
        add(f"{va}.{which}_embedding.weight", (rows, H), "emb")
    add("mel_linear.weight", (n_mel, H), "w")
    add("mel_linear.bias", (n_mel,), "b")
    P = pn["embedding_dim"]
    for i in range(pn["conv_layers"]):
        cin = n_mel if i == 0 else P
        cout = n_mel if i == pn["conv_layers"] - 1 else P
        p = f"postnet.convolutions.{i}"
        add(f"{p}.0.conv.weight", (cout, cin, pn["kernel_size"]), "w")
        add(f"{p}.0.conv.bias", (cout,), "b")
        add(f"{p}.1.weight", (cout,), "gamma")
        add(f"{p}.1.bias", (cout,), "beta")
        add(f"{p}.1.running_mean", (cout,), "bn_mean")
        add(f"{p}.1.running_var", (cout,), "bn_var")
        add(f"{p}.1.num_batches_tracked", (), "i64")
    add("speaker_emb.weight", (n_speakers, H), "emb")
    return m


def vocoder_manifest(config: dict, vocoder: str = "hifigan") -> "OrderedDict[str, Tuple[Shape, str]]":
    """name -> (shape, kind) for HifiGan.state_dict() / iSTFT.state_dict() (weight-normed: weight_g / weight_v).

    ``vocoder="istft"``: reference V/generator.py:65-94 -- same trunk from ``config["models"]["istft"]``, conv_post with
    n_fft + 2 output channels, and ResBlock2 (``resblocks.N.convs.{0,1}``) unless ``resblock == '1'`` (the string; :71)."""
    hg = config["models"][vocoder]
    n_mel = config["audio"]["mel"]["channels"]
    C0 = hg["upsample_initial_channel"]
    rb1 = (hg["resblock"] == "1") if vocoder == "istft" else (hg["resblock"] == 1)
    m: "OrderedDict[str, Tuple[Shape, str]]" = OrderedDict()

    def add_wn(prefix, vshape, kind="wn"):
        m[f"{prefix}.bias"] = ((vshape[0],) if kind != "wn_t" else (vshape[1],), "b")
        m[f"{prefix}.weight_g"] = ((vshape[0], 1, 1), "wn_g")
        m[f"{prefix}.weight_v"] = (tuple(vshape), kind)

    add_wn("conv_pre", (C0, n_mel, 7))
    ch = C0
    for i, (u, k) in enumerate(zip(hg["upsample_rates"], hg["upsample_kernel_sizes"])):
        add_wn(f"ups.{i}", (C0 // 2 ** i, C0 // 2 ** (i + 1), k), kind="wn_t")
    nk = len(hg["resblock_kernel_sizes"])
    for i in range(len(hg["upsample_rates"])):
        ch = C0 // 2 ** (i + 1)
        for j, (k, dil) in enumerate(zip(hg["resblock_kernel_sizes"], hg["resblock_dilation_sizes"])):
            if rb1:
                for cs in ("convs1", "convs2"):
                    for d in range(len(dil)):
                        add_wn(f"resblocks.{i * nk + j}.{cs}.{d}", (ch, ch, k))
            else:
                for d in range(2):
                    add_wn(f"resblocks.{i * nk + j}.convs.{d}", (ch, ch, k))
    add_wn("conv_post", ((hg["gen_istft_n_fft"] + 2) if vocoder == "istft" else 1, ch, 7))
    return m


def _draw(rng: np.random.Generator, name: str, shape: Shape, kind: str, *, stats: dict, mode: str,
          frames_per_phoneme: int) -> np.ndarray:
    f32 = np.float32
    if kind == "i64":
        return np.zeros(shape, dtype=np.int64)
    if kind == "zero":
        return np.zeros(shape, dtype=f32)
    if kind == "posenc":
        return sinusoid_table(shape[1], shape[2])[None]
    if kind in ("w", "wn"):
        fan_in = int(np.prod(shape[1:]))
        return (rng.standard_normal(shape) * (1.0 / math.sqrt(fan_in))).astype(f32)
    if kind == "wn_t":  # ConvTranspose1d weight [Cin, Cout, k]; each output sees Cin * k / stride taps
        fan_in = shape[0] * 2
        return (rng.standard_normal(shape) * (1.0 / math.sqrt(fan_in))).astype(f32)
    if kind == "wn_g":
        return None  # filled after weight_v (needs its norm)
    if kind == "b":
        return (0.05 * rng.standard_normal(shape)).astype(f32)
    if kind == "gamma":
        return (1.0 + 0.1 * rng.standard_normal(shape)).astype(f32)
    if kind == "beta":
        return (0.1 * rng.standard_normal(shape)).astype(f32)
    if kind == "bn_mean":
        return (0.1 * rng.standard_normal(shape)).astype(f32)
    if kind == "bn_var":
        return (1.0 + 0.1 * rng.random(shape)).astype(f32)
    if kind == "emb":
        return (0.3 * rng.standard_normal(shape)).astype(f32)
    if kind == "emb_pad0":
        w = rng.standard_normal(shape).astype(f32)
        w[0] = 0.0  # padding_idx row (U/blocks/transformer.py:41-43)
        return w
    if kind == "alpha":
        return np.ones(shape, dtype=f32)
    if kind == "dur_w":
        w = rng.standard_normal(shape)
        return ((0.06 * w) if mode == "varied" else (0.0 * w)).astype(f32)
    if kind == "dur_b":
        rng.standard_normal(shape)  # keep the stream aligned between modes
        if mode == "varied":
            return np.full(shape, math.log(4.6), dtype=f32)
        # exp(b) - 1 == frames_per_phoneme + 0.25: far from a rounding boundary
        return np.full(shape, math.log(frames_per_phoneme + 1.25), dtype=f32)
    if kind in ("pitch_w", "energy_w"):
        fan_in = shape[-1]
        return (rng.standard_normal(shape) * (0.4 / math.sqrt(fan_in))).astype(f32)
    if kind == "pitch_b":
        b = (0.05 * rng.standard_normal(shape)).astype(f32)
        b[-1] -= 0.8  # bias the uv logit towards "voiced" so the f0 buckets are exercised
        return b
    if kind == "pitch_b_log":   # f0 = 2 ** prediction: centre the f0 channel on log2(190 Hz)
        b = (0.05 * rng.standard_normal(shape)).astype(f32)
        b[0] += 7.57
        b[-1] -= 0.8
        return b
    if kind == "pitch_b_nouv":  # one output bucketized on pitch_bins (stats "pitch" min .. max): centre it inside the range
        return (4.0 + 0.05 * rng.standard_normal(shape)).astype(f32)
    if kind == "energy_b":
        return (2.5 + 0.05 * rng.standard_normal(shape)).astype(f32)
    if kind == "pitch_bins":   # (the reference's "log" variant needs positive stats; the synthetic ones are not: linear bins)
        return np.linspace(stats["pitch"]["min"], stats["pitch"]["max"], shape[0]).astype(f32)
    if kind == "energy_bins":
        return np.linspace(stats["energy"]["min"], stats["energy"]["max"], shape[0]).astype(f32)
    raise KeyError(kind)


def _fill(manifest, seed: int, **kw) -> Dict[str, np.ndarray]:
    rng = np.random.Generator(np.random.PCG64(seed))
    out: Dict[str, np.ndarray] = {}
    for name in sorted(manifest):
        shape, kind = manifest[name]
        out[name] = _draw(rng, name, shape, kind, **kw)
    # weight_g = ||v|| * (1 + 0.1 N) over all dims but 0 (weight_norm dim=0; V/generator.py:18,23,33)
    for name in sorted(manifest):
        if manifest[name][1] == "wn_g":
            v = out[name[: -len("weight_g")] + "weight_v"].astype(np.float64)
            norm = np.sqrt((v * v).sum(axis=(1, 2), keepdims=True))
            out[name] = (norm * (1.0 + 0.1 * rng.standard_normal(norm.shape))).astype(np.float32)
    return OrderedDict((k, out[k]) for k in manifest)


def make_acoustic_state(config: dict, stats: dict, n_speakers: int, seed: int = 1234, mode: str = "fixed",
                        frames_per_phoneme: int = 6, n_symbols: int = N_SYMBOLS) -> "OrderedDict[str, np.ndarray]":
    """mode='fixed': exactly ``frames_per_phoneme`` frames per real phoneme (benchmarks);
    mode='varied': data-dependent durations of roughly 1..9 frames (correctness fixtures)."""
    return _fill(acoustic_manifest(config, n_speakers, n_symbols), seed, stats=stats, mode=mode,
                 frames_per_phoneme=frames_per_phoneme)


def make_vocoder_state(config: dict, seed: int = 4321, vocoder: str = "hifigan") -> "OrderedDict[str, np.ndarray]":
    sd = _fill(vocoder_manifest(config, vocoder), seed, stats=None, mode="fixed", frames_per_phoneme=0)
    # keep the pre-tanh signal in the unsaturated range so waveform parity is informative (iSTFT: log-magnitudes of O(1))
    sd["conv_post.weight_g"] = (sd["conv_post.weight_g"] * 0.25).astype(np.float32)
    return sd


def to_torch(state: Dict[str, np.ndarray]):
    import torch

    return OrderedDict((k, torch.from_numpy(np.ascontiguousarray(v))) for k, v in state.items())
