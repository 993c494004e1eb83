"""state_dict -> packed weight blob for the HIP engine.

This is the "PyTorch-ROCm only for weight load / layout" part of the design:
it reads the two reference checkpoints' state dicts (the formats written by
reference e2e_tts/src/tools/tools_for_model.py:155-177 and loaded at
API/utils.py:48-49,54-55) and lays the tensors out the way the kernels read
them:

* Conv1d weights [Cout, Cin, K] -> [Cout, K, Cin] (tap-major rows: one GEMM K axis);
* w_qs / w_ks / w_vs stacked into one [3H, H] projection;
* weight_norm folded, w = g * v / ||v|| (reference V/generator.py:18,23,33; fp32 via
  torch._weight_norm, the same routine the reference's parametrisation calls);
* eval-mode BatchNorm1d folded into the Postnet convolutions (U/layers.py:530-553):
  w' = w * gamma / sqrt(var + 1e-5), b' = (b - mean) * gamma / sqrt(var + 1e-5) + beta;
* ConvTranspose1d(k = 2s, stride s, pad s/2) rewritten as a 3-tap convolution with s * Cout output
  channels (polyphase form; zero where a tap does not reach a phase);
* both sinusoid tables precomputed on the host (U/blocks/utils.py:14-34 in float64 numpy;
  U/sublayers.py:28-44 with the reference's own torch fp32 operations).

Blob layout (little endian): 32-byte header {magic "E2ETTSW1", u32 version = 1, u32 n_entries,
u64 data_offset, u64 total_bytes}, n_entries x {char name[64], u64 offset, u64 numel}, then fp32
tensors at 256-byte aligned offsets.
"""
from __future__ import annotations

import math
import struct
from collections import OrderedDict
from typing import Dict, Mapping, Optional

import numpy as np

from .config import EngineDims
from .synth_weights import sinusoid_table

MAGIC = b"E2ETTSW1"
ALIGN = 256
VAR_POS_INIT_ROWS = 4096  # SinusoidalPositionalEmbedding(idim, 0, init_size=4096), reference U/layers.py:488


def _np(t) -> np.ndarray:
    if isinstance(t, np.ndarray):
        return t
    return t.detach().cpu().numpy()


def fold_weight_norm(g, v) -> np.ndarray:
    """w = g * v / ||v||, norm over all dims but 0 (weight_norm default dim=0)."""
    try:
        import torch
        return torch._weight_norm(torch.as_tensor(_np(v)), torch.as_tensor(_np(g)), 0).numpy()
    except ImportError:  # pragma: no cover - torch is part of the image
        v64 = _np(v).astype(np.float64)
        norm = np.sqrt((v64 * v64).sum(axis=tuple(range(1, v64.ndim)), keepdims=True))
        return (_np(g).astype(np.float64) * v64 / norm).astype(np.float32)


def variance_position_table(rows: int, dim: int) -> np.ndarray:
    """The fairseq-style table of reference U/sublayers.py:28-44, built with the same torch fp32 ops
    (the fp32 angle pos * exp(-k * c) is sensitive to the last bit of exp, so torch is used, not numpy)."""
    import torch
    half = dim // 2
    emb = math.log(10000) / (half - 1)
    emb = torch.exp(torch.arange(half, dtype=torch.float) * -emb)
    emb = torch.arange(rows, dtype=torch.float).unsqueeze(1) * emb.unsqueeze(0)
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=1).view(rows, -1)
    if dim % 2 == 1:
        emb = torch.cat([emb, torch.zeros(rows, 1)], dim=1)
    emb[0, :] = 0  # padding_idx = 0
    return emb.numpy()


def conv_rows(w: np.ndarray) -> np.ndarray:
    """[Cout, Cin, K] -> [Cout, K * Cin] tap-major."""
    return np.ascontiguousarray(w.transpose(0, 2, 1)).reshape(w.shape[0], -1)


def pack_x3(w2: np.ndarray, KW: int, Cin: int) -> np.ndarray:
    """fp32 tap-major conv weight [Cout, KW*Cin] -> split-precision image for the bf16x3 MFMA path:
    [Cout, KW, ceil(Cin/32), 32 bf16 hi | 32 bf16 lo] with hi = bf16(w), lo = bf16(w - hi) (round to nearest even),
    returned as 32-bit words [Cout, KW * nchunk * 32] (one LDS row of the kernel = one 32-channel chunk)."""
    cout = w2.shape[0]
    nchunk = (Cin + 31) // 32
    w = np.zeros((cout, KW, nchunk * 32), dtype=np.float32)
    w[:, :, :Cin] = w2.reshape(cout, KW, Cin)

    def bf16_bits(x):
        u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
        return ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)).astype(np.uint16)

    hi = bf16_bits(w)
    hi_f = (hi.astype(np.uint32) << np.uint32(16)).view(np.float32)
    lo = bf16_bits(w - hi_f)
    out = np.concatenate([hi.reshape(cout, KW, nchunk, 32), lo.reshape(cout, KW, nchunk, 32)], axis=-1)
    return np.ascontiguousarray(out).reshape(cout, -1).view(np.float32)


def split_rows_x3(x: np.ndarray) -> np.ndarray:
    """fp32 [..., D] -> split-precision rows [..., D] of 32-bit words: D bf16 hi (= bf16(x), round to nearest even) followed by D bf16 lo
    (= bf16(x - hi)); the operand format of the bf16x3 attention kernels (D even)."""
    def bf16_bits(v):
        u = np.ascontiguousarray(v, dtype=np.float32).view(np.uint32)
        return ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)).astype(np.uint16)
    x = np.ascontiguousarray(x, dtype=np.float32)
    hi = bf16_bits(x)
    lo = bf16_bits(x - (hi.astype(np.uint32) << np.uint32(16)).view(np.float32))
    return np.ascontiguousarray(np.concatenate([hi, lo], axis=-1)).view(np.float32)


def polyphase_upsampler(w: np.ndarray, b: np.ndarray, stride: int):
    """ConvTranspose1d weight [Cin, Cout, K = 2s] (pad s/2) -> 3-tap conv weight [s*Cout, 3*Cin], bias [s*Cout].

    out[s*q + r] = sum_i x[i] . w[:, :, s*q + r + pad - s*i]; with taps m = 0, 1, 2 <-> i = q - 1 + m the
    kernel index is kk = r + pad + s * (1 - m), used when 0 <= kk < K.
    """
    cin, cout, K = w.shape
    s = stride
    pad = (K - s) // 2
    assert K == 2 * s and s % 2 == 0
    out = np.zeros((s, cout, 3, cin), dtype=np.float32)
    for r in range(s):
        for m in range(3):
            kk = r + pad + s * (1 - m)
            if 0 <= kk < K:
                out[r, :, m, :] = w[:, :, kk].T
    return out.reshape(s * cout, 3 * cin), np.tile(b, s).astype(np.float32)


def pack_tensors(dims: EngineDims, acoustic: Optional[Mapping[str, object]], vocoder: Optional[Mapping[str, object]]) -> "OrderedDict[str, np.ndarray]":
    """Engine tensor name -> fp32 array (consumers: bind_acoustic() / bind_vocoder() in csrc/engine.hip).
    Either state dict may be None: the blob then carries only the other model."""
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    if acoustic is not None:
        _pack_acoustic(dims, {k: _np(v) for k, v in acoustic.items()}, out)
    if vocoder is not None:
        _pack_vocoder(dims, {k: _np(v) for k, v in vocoder.items()}, out)
    if not out:
        raise ValueError("nothing to pack: both state dicts are None")
    return OrderedDict((k, np.ascontiguousarray(v, dtype=np.float32)) for k, v in out.items())


def _need(sd, key, shape=None):
    if key not in sd:
        raise KeyError(f"checkpoint has no tensor {key!r}")
    t = sd[key]
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{key}: shape {tuple(t.shape)} != expected {tuple(shape)}")
    return t.astype(np.float32) if t.dtype.kind == "f" else t


def _pack_conformer(dims: EngineDims, A, out) -> None:
    """Conformer blocks (reference U/blocks/conformer.py:171-255) -> engine tensors `enc.N.*` / `dec.N.*` (bind_conformer()).

    Folded here: the half-step factor into the second FFN Linear (x 0.5 is exact); BatchNorm (eval) into the depthwise conv;
    pos_proj(position table) per layer -- it does not depend on the input -- laid out per head [n_head][rows][d_head], once from
    the stored table (`att.pos`, N <= max_seq_len) and once from the regenerated one (`att.posr`, conformer.py:339-344).  u_bias and
    v_bias travel as they are (`att.u`, `att.v`, heads flattened): the attention kernel adds them to its query fragments."""
    H, F, k = dims.hidden, dims.ffn_dim, dims.ffn_k1
    need = _need
    regen = sinusoid_table(dims.pos_table_rows, H)
    for side, short, n in (("encoder", "enc", dims.enc_layers), ("decoder", "dec", dims.dec_layers)):
        nh = (dims.dec_n_head or dims.n_head) if short == "dec" else dims.n_head   # encoder_head / decoder_head (conformer.py:31,108)
        dh = H // nh
        for l in range(n):
            s = f"{side}.layer_stack.{l}.sequential"
            q = f"{short}.{l}."
            gemms = []
            for i, nm in ((0, "ff1"), (3, "ff2")):
                m = f"{s}.{i}.module.sequential"
                out[q + nm + ".ln.g"] = need(A, f"{m}.0.weight", (H,))
                out[q + nm + ".ln.b"] = need(A, f"{m}.0.bias", (H,))
                out[q + nm + ".w1"] = need(A, f"{m}.1.linear.weight", (F, H))
                out[q + nm + ".b1"] = need(A, f"{m}.1.linear.bias", (F,))
                out[q + nm + ".w2"] = need(A, f"{m}.4.linear.weight", (H, F)) * np.float32(dims.cf_ffn_factor)
                out[q + nm + ".b2"] = need(A, f"{m}.4.linear.bias", (H,)) * np.float32(dims.cf_ffn_factor)
                gemms += [(nm + ".w1", H), (nm + ".w2", F)]
            a = f"{s}.1.module"
            out[q + "att.ln.g"] = need(A, f"{a}.layer_norm.weight", (H,))
            out[q + "att.ln.b"] = need(A, f"{a}.layer_norm.bias", (H,))
            out[q + "att.wqkv"] = np.concatenate(
                [need(A, f"{a}.attention.{w}.linear.weight", (H, H)) for w in ("query_proj", "key_proj", "value_proj")], 0)
            out[q + "att.wo"] = need(A, f"{a}.attention.out_proj.linear.weight", (H, H))
            out[q + "att.u"] = need(A, f"{a}.attention.u_bias", (nh, dh)).reshape(H)
            out[q + "att.v"] = need(A, f"{a}.attention.v_bias", (nh, dh)).reshape(H)
            wp = need(A, f"{a}.attention.pos_proj.linear.weight", (H, H))
            stored = need(A, f"{a}.positional_encoding", (1, dims.max_seq_len + 1, H))[0]
            for tag, table in (("pos", stored), ("posr", regen)):
                rows = table.shape[0]
                ph = np.ascontiguousarray((table @ wp.T).astype(np.float32).reshape(rows, nh, dh).transpose(1, 0, 2))
                out[q + f"att.{tag}"] = ph
                if short == "dec" and dh % 16 == 0:  # split-precision image for rel_attention_x3_kernel: rows of dh bf16 hi | dh bf16 lo
                    out[q + f"att.{tag}.x3"] = split_rows_x3(ph)
            m = f"{s}.2.module.sequential"
            out[q + "cv.ln.g"] = need(A, f"{m}.0.weight", (H,))
            out[q + "cv.ln.b"] = need(A, f"{m}.0.bias", (H,))
            out[q + "cv.pw1.w"] = need(A, f"{m}.2.conv.weight", (2 * H, H, 1))[:, :, 0]
            out[q + "cv.pw1.b"] = need(A, f"{m}.2.conv.bias", (2 * H,))
            dw = need(A, f"{m}.4.conv.weight", (H, 1, k))[:, 0, :].astype(np.float64)
            scale = need(A, f"{m}.5.weight", (H,)).astype(np.float64) / np.sqrt(need(A, f"{m}.5.running_var", (H,)).astype(np.float64) + 1e-5)
            out[q + "cv.dw.w"] = np.ascontiguousarray((dw * scale[:, None]).T).astype(np.float32)            # [k][H]
            out[q + "cv.dw.b"] = (need(A, f"{m}.5.bias", (H,)).astype(np.float64)
                                  - need(A, f"{m}.5.running_mean", (H,)).astype(np.float64) * scale).astype(np.float32)
            out[q + "cv.pw2.w"] = need(A, f"{m}.7.conv.weight", (H, H, 1))[:, :, 0]
            out[q + "cv.pw2.b"] = need(A, f"{m}.7.conv.bias", (H,))
            out[q + "ln.g"] = need(A, f"{s}.4.weight", (H,))
            out[q + "ln.b"] = need(A, f"{s}.4.bias", (H,))
            gemms += [("att.wqkv", H), ("att.wo", H), ("cv.pw1.w", H), ("cv.pw2.w", H)]
            if short == "dec":  # as for the FFT blocks: only the decoder may run split-precision
                for nm, cin in gemms:
                    out[q + nm + ".x3"] = pack_x3(np.ascontiguousarray(out[q + nm]), 1, cin)


def _pack_acoustic(dims: EngineDims, A, out) -> None:
    H = dims.hidden
    need = _need
    out["enc.emb"] = need(A, "encoder.src_word_emb.weight", (dims.n_symbols + 1, H))
    out["enc.pos"] = need(A, "encoder.position_enc", (1, dims.max_seq_len + 1, H))[0]
    out["dec.pos"] = need(A, "decoder.position_enc", (1, dims.max_seq_len + 1, H))[0]
    out["pos.regen"] = sinusoid_table(dims.pos_table_rows, H)
    out["spk.emb"] = need(A, "speaker_emb.weight", (dims.n_speakers, H))
    if dims.block_type == 1:
        _pack_conformer(dims, A, out)
    for side, short, n in (("encoder", "enc", dims.enc_layers), ("decoder", "dec", dims.dec_layers)):
        for l in range(n if dims.block_type == 0 else 0):
            p = f"{side}.layer_stack.{l}"
            q = f"{short}.{l}."
            out[q + "wqkv"] = np.concatenate([need(A, f"{p}.slf_attn.{w}.weight", (H, H)) for w in ("w_qs", "w_ks", "w_vs")], 0)
            out[q + "bqkv"] = np.concatenate([need(A, f"{p}.slf_attn.{w}.bias", (H,)) for w in ("w_qs", "w_ks", "w_vs")], 0)
            out[q + "wo"] = need(A, f"{p}.slf_attn.fc.weight", (H, H))
            out[q + "bo"] = need(A, f"{p}.slf_attn.fc.bias", (H,))
            out[q + "ln1.g"] = need(A, f"{p}.slf_attn.layer_norm.weight", (H,))
            out[q + "ln1.b"] = need(A, f"{p}.slf_attn.layer_norm.bias", (H,))
            out[q + "w1"] = conv_rows(need(A, f"{p}.pos_ffn.w_1.weight", (dims.ffn_dim, H, dims.ffn_k1)))
            out[q + "b1"] = need(A, f"{p}.pos_ffn.w_1.bias", (dims.ffn_dim,))
            out[q + "w2"] = conv_rows(need(A, f"{p}.pos_ffn.w_2.weight", (H, dims.ffn_dim, dims.ffn_k2)))
            out[q + "b2"] = need(A, f"{p}.pos_ffn.w_2.bias", (H,))
            out[q + "ln2.g"] = need(A, f"{p}.pos_ffn.layer_norm.weight", (H,))
            out[q + "ln2.b"] = need(A, f"{p}.pos_ffn.layer_norm.bias", (H,))
            if short == "dec":  # the decoder may run split-precision (nothing discrete depends on it); the encoder never does
                out[q + "wqkv.x3"] = pack_x3(out[q + "wqkv"], 1, H)
                out[q + "wo.x3"] = pack_x3(out[q + "wo"], 1, H)
                out[q + "w1.x3"] = pack_x3(out[q + "w1"], dims.ffn_k1, H)
                out[q + "w2.x3"] = pack_x3(out[q + "w2"], dims.ffn_k2, dims.ffn_dim)
    va = "variance_adaptor"
    for name, short, layers, kern, chans, odim in (
            ("duration_predictor", "dur", dims.dur_layers, dims.dur_kernel, dims.dur_chans, 1),
            ("pitch_predictor", "pitch", dims.var_layers, dims.var_kernel, dims.var_chans, 1 if dims.pitch_no_uv else 2),
            ("energy_predictor", "energy", dims.energy_layers or dims.var_layers, dims.energy_kernel or dims.var_kernel, dims.var_chans, 1)):
        for i in range(layers):
            cin = H if i == 0 else chans
            p = f"{va}.{name}.conv.{i}"
            out[f"{short}.{i}.w"] = conv_rows(need(A, f"{p}.1.weight", (chans, cin, kern)))
            out[f"{short}.{i}.b"] = need(A, f"{p}.1.bias", (chans,))
            out[f"{short}.{i}.g"] = need(A, f"{p}.3.weight", (chans,))
            out[f"{short}.{i}.beta"] = need(A, f"{p}.3.bias", (chans,))
        out[f"{short}.lin.w"] = need(A, f"{va}.{name}.linear.weight", (odim, chans))
        out[f"{short}.lin.b"] = need(A, f"{va}.{name}.linear.bias", (odim,))
        if short != "dur":
            out[f"{short}.alpha"] = need(A, f"{va}.{name}.pos_embed_alpha", (1,))
    # (frame_level features index it by frame: rows for every T the decoder's own table allows -- the reference grows its table on demand,
    # U/sublayers.py:56-60, with the same function, so the rows below 4096 are the same)
    var_rows = max(VAR_POS_INIT_ROWS, dims.pos_table_rows + 2) if (dims.pitch_frame or dims.energy_frame) else VAR_POS_INIT_ROWS
    out["var.pos"] = variance_position_table(var_rows, H)
    out["pitch.emb"] = need(A, f"{va}.pitch_embedding.weight", (dims.pitch_emb_rows or dims.n_bins, H))
    if dims.pitch_no_uv:
        out["pitch.bins"] = need(A, f"{va}.pitch_bins", (dims.n_bins - 1,))
    out["energy.emb"] = need(A, f"{va}.energy_embedding.weight", (dims.n_bins, H))
    out["energy.bins"] = need(A, f"{va}.energy_bins", (dims.n_bins - 1,))
    out["mel.w"] = need(A, "mel_linear.weight", (dims.n_mel, H))
    out["mel.b"] = need(A, "mel_linear.bias", (dims.n_mel,))
    out["mel.w.x3"] = pack_x3(out["mel.w"], 1, H)
    for i in range(dims.postnet_layers):
        p = f"postnet.convolutions.{i}"
        w = need(A, f"{p}.0.conv.weight").astype(np.float64)
        b = need(A, f"{p}.0.conv.bias").astype(np.float64)
        scale = need(A, f"{p}.1.weight").astype(np.float64) / np.sqrt(need(A, f"{p}.1.running_var").astype(np.float64) + 1e-5)
        wf = (w * scale[:, None, None]).astype(np.float32)
        bf = ((b - need(A, f"{p}.1.running_mean").astype(np.float64)) * scale + need(A, f"{p}.1.bias").astype(np.float64)).astype(np.float32)
        out[f"post.{i}.w"] = conv_rows(wf)
        out[f"post.{i}.b"] = bf
        out[f"post.{i}.w.x3"] = pack_x3(out[f"post.{i}.w"], wf.shape[2], wf.shape[1])


def _pack_vocoder(dims: EngineDims, V, out) -> None:
    need = _need

    def voc_weight(prefix):
        if prefix + ".weight_v" in V:
            return fold_weight_norm(need(V, prefix + ".weight_g"), need(V, prefix + ".weight_v")), need(V, prefix + ".bias")
        return need(V, prefix + ".weight"), need(V, prefix + ".bias")  # remove_weight_norm()'d checkpoint

    def put(name, w2, b, KW, cin):
        out[name + ".w"], out[name + ".b"] = w2, b
        out[name + ".wx3"] = pack_x3(w2, KW, cin)   # split-precision copy for the bf16x3 path

    w, b = voc_weight("conv_pre")
    put("voc.pre", conv_rows(w), b, w.shape[2], w.shape[1])
    nk = len(dims.voc_rb_kernel)
    for i, s in enumerate(dims.voc_up_rate):
        w, b = voc_weight(f"ups.{i}")
        w3, b3 = polyphase_upsampler(w, b, s)
        put(f"voc.up.{i}", w3, b3, 3, w.shape[0])
        for j in range(nk):
            idx = i * nk + j
            for m in range(len(dims.voc_rb_dil[j])):
                if dims.voc_resblock == 2:  # ResBlock2: `convs.{m}` (reference V/layers.py:52-56)
                    w, b = voc_weight(f"resblocks.{idx}.convs.{m}")
                    put(f"voc.rb.{idx}.c.{m}", conv_rows(w), b, w.shape[2], w.shape[1])
                    continue
                for cs, short in (("convs1", "c1"), ("convs2", "c2")):
                    w, b = voc_weight(f"resblocks.{idx}.{cs}.{m}")
                    put(f"voc.rb.{idx}.{short}.{m}", conv_rows(w), b, w.shape[2], w.shape[1])
    w, b = voc_weight("conv_post")
    if dims.voc_istft_nfft:
        # iSTFTNet: conv_post has n_fft + 2 output channels (V/generator.py:92); zero rows pad them to a multiple of 4 so that
        # the convolution kernel's float4 epilogue applies
        pc = dims.voc_post_channels
        w2 = np.zeros((pc, w.shape[1] * w.shape[2]), np.float32)
        w2[:w.shape[0]] = conv_rows(w)
        b2 = np.zeros((pc,), np.float32)
        b2[:w.shape[0]] = b
        put("voc.post", w2, b2, w.shape[2], w.shape[1])
    else:
        out["voc.post.w"], out["voc.post.b"] = conv_rows(w), b


def build_blob(tensors: Mapping[str, np.ndarray]) -> np.ndarray:
    """Serialise to the flat image e2etts_load_weights() takes.  Returns a uint8 array."""
    n = len(tensors)
    header_size = 32
    dir_size = n * 80
    data_offset = (header_size + dir_size + ALIGN - 1) // ALIGN * ALIGN
    entries = []
    off = data_offset
    for name, t in tensors.items():
        raw = name.encode()
        if len(raw) > 63:
            raise ValueError(f"tensor name too long: {name}")
        entries.append((raw, off, t.size))
        off = (off + t.size * 4 + ALIGN - 1) // ALIGN * ALIGN
    total = off
    blob = np.zeros(total, dtype=np.uint8)
    blob[:header_size] = np.frombuffer(struct.pack("<8sIIQQ", MAGIC, 1, n, data_offset, total), dtype=np.uint8)
    pos = header_size
    for (raw, o, numel), t in zip(entries, tensors.values()):
        blob[pos:pos + 80] = np.frombuffer(struct.pack("<64sQQ", raw, o, numel), dtype=np.uint8)
        pos += 80
        blob[o:o + numel * 4] = t.reshape(-1).view(np.uint8)
    return blob


def pack(dims: EngineDims, acoustic: Optional[Mapping[str, object]], vocoder: Optional[Mapping[str, object]]) -> np.ndarray:
    return build_blob(pack_tensors(dims, acoustic, vocoder))
