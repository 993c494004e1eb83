"""CPU oracle: a numpy restatement of the reference's inference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``e2e_tts_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg do, and there only as the checker / the reported CPU
baseline -- never as the product path.

Every function cites the reference file:line it restates (paths relative to
/root/reference; U/ = e2e_tts/models/acoustic/unsupervised_fastspeech2/,
V/ = e2e_tts/models/vocoder/, API/ = e2e_tts/src/api/).  Layouts and operation
order follow the reference (channels-first convolutions, [B, L, H] sequences)
so that the restatement can be read side by side with it.

Parity pinning: the reference holds no tests or golden vectors,
so this oracle is pinned by fixtures generated from the reference's own
modules run in the build container (oracle/make_goldens.py ->
tests/golden/*.npz); tests/test_oracle_golden.py checks every fixture.

All arithmetic is numpy; ``dtype`` selects float32 (reference-equivalent) or
float64 (used to measure the fp32 noise floor).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

LRELU_SLOPE = 0.1  # V/generator.py:10, V/layers.py:7

# U/function.py:9-13
F0_BIN = 256
F0_MAX = 1100.0
F0_MIN = 50.0
F0_MEL_MIN = 1127 * np.log(1 + F0_MIN / 700)
F0_MEL_MAX = 1127 * np.log(1 + F0_MAX / 700)


# --------------------------------------------------------------------------- primitives

def linear(x: np.ndarray, w: np.ndarray, b: Optional[np.ndarray]) -> np.ndarray:
    """torch.nn.Linear: y = x W^T + b."""
    y = x @ w.T
    if b is not None:
        y = y + b
    return y


_C_CONV = None
_CONV_BACKEND = None  # None: the C / OpenMP backend when built, else numpy; "numpy"; "torch": ATen's CPU kernels (fp32 only)


def set_conv_backend(name: Optional[str]) -> None:
    """Which implementation conv1d / conv_transpose1d use for fp32 inputs.  "torch" runs the two primitives through
    torch.nn.functional on the CPU -- the very kernels the reference's own CPU path executes (V/generator.py:37-53 calls
    nn.Conv1d / nn.ConvTranspose1d) -- and exists for bench.py's cpu_baseline, which reports the faster of the backends so that
    the stand-in is not slower than the reference's own path.  Everything else of the oracle is unchanged."""
    global _CONV_BACKEND
    if name not in (None, "numpy", "torch", "c"):
        raise ValueError(name)
    _CONV_BACKEND = name



def _c_conv():
    """Optional plain-C / OpenMP backend of conv1d (oracle/conv1d.c, built by __graft_entry__.build()); fp32 only.
    Same arithmetic up to summation order; the numpy path below is the fallback and the float64 path."""
    global _C_CONV
    if _C_CONV is None:
        import ctypes
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libref_conv1d.so")
        _C_CONV = False
        if os.path.exists(path) and not os.environ.get("E2ETTS_ORACLE_NUMPY"):
            try:
                lib = ctypes.CDLL(path)
                lib.ref_conv1d_f32.restype = ctypes.c_int
                lib.ref_conv1d_f32.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 7
                _C_CONV = lib
            except OSError:
                _C_CONV = False
    return _C_CONV


def conv1d(x: np.ndarray, w: np.ndarray, b: Optional[np.ndarray], padding: int = 0, dilation: int = 1,
           chunk: int = 8192) -> np.ndarray:
    """torch.nn.Conv1d (stride 1, zero padding).  x [B, Cin, T], w [Cout, Cin, K] -> [B, Cout, T'].

    fp32 inputs go to the C backend when it is built; otherwise (and for float64) numpy: one [Cout, K*Cin] x [K*Cin, n]
    product per chunk of n output positions (im2col by chunks keeps the BLAS call large even when Cin is 32 and the
    column buffer small enough to stay in cache).
    """
    B, Cin, T = x.shape
    f32 = x.dtype == np.float32 and w.dtype == np.float32
    if f32 and _CONV_BACKEND == "torch":
        import torch
        with torch.no_grad():
            y = torch.nn.functional.conv1d(torch.from_numpy(np.ascontiguousarray(x)), torch.from_numpy(np.ascontiguousarray(w)),
                                           None if b is None else torch.from_numpy(np.ascontiguousarray(b, dtype=np.float32)),
                                           padding=padding, dilation=dilation)
        return y.numpy()
    lib = _c_conv() if (f32 and _CONV_BACKEND in (None, "c")) else None
    if lib:
        Cout, _, K = w.shape
        Tout = T + 2 * padding - dilation * (K - 1)
        xc, wc = np.ascontiguousarray(x), np.ascontiguousarray(w)
        bc = None if b is None else np.ascontiguousarray(b, dtype=np.float32)
        out = np.empty((B, Cout, Tout), dtype=np.float32)
        rc = lib.ref_conv1d_f32(xc.ctypes.data, wc.ctypes.data, None if bc is None else bc.ctypes.data, out.ctypes.data,
                                B, Cin, T, Cout, K, padding, dilation)
        if rc == 0:
            return out
    Cout, _, K = w.shape
    xp = np.zeros((B, Cin, T + 2 * padding), dtype=x.dtype)
    xp[:, :, padding:padding + T] = x
    Tout = T + 2 * padding - dilation * (K - 1)
    out = np.empty((B, Cout, Tout), dtype=x.dtype)
    w2 = np.ascontiguousarray(w.transpose(0, 2, 1).reshape(Cout, K * Cin))
    cols = np.empty((K * Cin, min(chunk, max(Tout, 1))), dtype=x.dtype)
    for bi in range(B):
        for t0 in range(0, Tout, chunk):
            n = min(chunk, Tout - t0)
            for k in range(K):
                cols[k * Cin:(k + 1) * Cin, :n] = xp[bi, :, t0 + k * dilation:t0 + k * dilation + n]
            np.matmul(w2, cols[:, :n], out=out[bi, :, t0:t0 + n])
    if b is not None:
        out += b[None, :, None]
    return out


def conv_transpose1d(x: np.ndarray, w: np.ndarray, b: Optional[np.ndarray], stride: int, padding: int) -> np.ndarray:
    """torch.nn.ConvTranspose1d.  x [B, Cin, T], w [Cin, Cout, K] -> [B, Cout, (T-1)*stride - 2*padding + K]."""
    B, Cin, T = x.shape
    _, Cout, K = w.shape
    if _CONV_BACKEND == "torch" and x.dtype == np.float32 and w.dtype == np.float32:
        import torch
        with torch.no_grad():
            y = torch.nn.functional.conv_transpose1d(torch.from_numpy(np.ascontiguousarray(x)), torch.from_numpy(np.ascontiguousarray(w)),
                                                     None if b is None else torch.from_numpy(np.ascontiguousarray(b, dtype=np.float32)),
                                                     stride=stride, padding=padding)
        return y.numpy()
    full = np.zeros((B, Cout, (T - 1) * stride + K), dtype=x.dtype)
    wt = np.ascontiguousarray(w.transpose(2, 1, 0))  # [K, Cout, Cin]: contiguous per tap, or matmul leaves BLAS
    xc = np.ascontiguousarray(x)
    for k in range(K):
        full[:, :, k:k + (T - 1) * stride + 1:stride] += np.matmul(wt[k], xc)
    out = full[:, :, padding:full.shape[2] - padding]
    if b is not None:
        out = out + b[None, :, None]
    return np.ascontiguousarray(out)


def layer_norm(x: np.ndarray, gamma: np.ndarray, beta: np.ndarray, eps: float) -> np.ndarray:
    """torch.nn.LayerNorm over the last dim (biased variance)."""
    mean = x.mean(axis=-1, keepdims=True)
    var = ((x - mean) ** 2).mean(axis=-1, keepdims=True)
    return (x - mean) / np.sqrt(var + x.dtype.type(eps)) * gamma + beta


def leaky_relu(x: np.ndarray, slope: float) -> np.ndarray:
    """F.leaky_relu for 0 <= slope <= 1: max(x, slope * x) (same values as where(x >= 0, x, slope * x))."""
    return np.maximum(x, x * x.dtype.type(slope))


def softmax_lastdim(x: np.ndarray) -> np.ndarray:
    """A row of -inf only (every key masked: an utterance of zero frames) gives NaN, as torch's softmax does; the FFT block's
    masked_fill replaces those rows afterwards (U/blocks/transformer.py:182-183)."""
    with np.errstate(invalid="ignore"):
        m = x.max(axis=-1, keepdims=True)
        e = np.exp(x - m)
        return e / e.sum(axis=-1, keepdims=True)


def get_mask_from_lengths(lengths: np.ndarray, max_len: Optional[int] = None) -> np.ndarray:
    """U/function.py:17-25.  True = padding."""
    if max_len is None:
        max_len = int(lengths.max())
    ids = np.arange(max_len)[None, :]
    return ids >= lengths[:, None]


def sinusoid_table(n_position: int, d_hid: int) -> np.ndarray:
    """U/blocks/utils.py:14-34: float64 numpy table, sin on even / cos on odd columns -> fp32."""
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid)
    table = pos / np.power(10000, 2 * (j // 2) / d_hid)[None, :]
    table[:, 0::2] = np.sin(table[:, 0::2])
    table[:, 1::2] = np.cos(table[:, 1::2])
    return table.astype(np.float32)


def fairseq_sinusoid_table(num_embeddings: int, embedding_dim: int, padding_idx: Optional[int] = 0) -> np.ndarray:
    """U/sublayers.py:28-44: [sin | cos] halves, frequencies exp(-k ln(1e4)/(half-1)), fp32 arithmetic."""
    half = embedding_dim // 2
    f32 = np.float32
    step = f32(math.log(10000) / (half - 1))
    freq = np.exp(np.arange(half, dtype=f32) * -step).astype(f32)
    ang = (np.arange(num_embeddings, dtype=f32)[:, None] * freq[None, :]).astype(f32)
    emb = np.concatenate([np.sin(ang), np.cos(ang)], axis=1).astype(f32)
    if embedding_dim % 2 == 1:
        emb = np.concatenate([emb, np.zeros((num_embeddings, 1), f32)], axis=1)
    if padding_idx is not None:
        emb[padding_idx, :] = 0
    return emb


def make_positions(x0: np.ndarray, padding_idx: int = 0) -> np.ndarray:
    """U/function.py:28-38: cumulative count of non-padding entries, zero at padding."""
    mask = (x0 != padding_idx).astype(np.int64)
    return np.cumsum(mask, axis=1) * mask + padding_idx


# --------------------------------------------------------------------------- acoustic model

class AcousticOracle:
    """UnsupervisedFastSpeech2.inference restated (U/model.py:155-194)."""

    def __init__(self, state: Dict[str, np.ndarray], config: dict, stats: dict, dtype=np.float32,
                 var_pos_table: Optional[np.ndarray] = None):
        """``var_pos_table``: optional [>= L+1, H] table for the variance predictors.  The reference builds
        it with torch fp32 exp/sin (U/sublayers.py:28-44); its angles are 1-ulp sensitive to the exp
        implementation, so a caller holding the torch-built table can inject it; the default is the
        numpy restatement (checked against the reference's table in tests/test_oracle_golden.py)."""
        self.dt = np.dtype(dtype).type
        self.sd = {k: (v.astype(dtype) if v.dtype.kind == "f" else v) for k, v in state.items()}
        self.fs = config["models"]["fastspeech2"]
        self.bt = self.fs["building_block"]["block_type"]  # "transformer" | "conformer" (U/model.py:24-27)
        self.tr = self.fs["building_block"][self.bt]
        self.stats = stats
        self.H = self.fs["encoder_hidden"]
        self.n_head = self.tr["encoder_head"]
        self.max_seq_len = self.fs["max_seq_len"]
        self.var_pos_table = (var_pos_table if var_pos_table is not None
                              else fairseq_sinusoid_table(4096, self.H, 0))  # U/layers.py:488 init_size=4096
        self.trace: Dict[str, np.ndarray] = {}

    # U/blocks/transformer.py:213-240 + :251-261
    def mha(self, p: str, x: np.ndarray, key_pad: np.ndarray) -> np.ndarray:
        sd, nh = self.sd, self.n_head
        B, N, H = x.shape
        dk = H // nh
        q = linear(x, sd[p + ".w_qs.weight"], sd[p + ".w_qs.bias"]).reshape(B, N, nh, dk).transpose(2, 0, 1, 3)
        k = linear(x, sd[p + ".w_ks.weight"], sd[p + ".w_ks.bias"]).reshape(B, N, nh, dk).transpose(2, 0, 1, 3)
        v = linear(x, sd[p + ".w_vs.weight"], sd[p + ".w_vs.bias"]).reshape(B, N, nh, dk).transpose(2, 0, 1, 3)
        attn = np.matmul(q, k.transpose(0, 1, 3, 2))                 # [nh, B, N, N]
        attn = attn / self.dt(np.power(dk, 0.5))                    # temperature :201
        attn = np.where(key_pad[None, :, None, :], self.dt(-np.inf), attn)
        attn = softmax_lastdim(attn)
        out = np.matmul(attn, v)                                     # [nh, B, N, dk]
        out = out.transpose(1, 2, 0, 3).reshape(B, N, nh * dk)
        out = linear(out, sd[p + ".fc.weight"], sd[p + ".fc.bias"])
        return layer_norm(out + x, sd[p + ".layer_norm.weight"], sd[p + ".layer_norm.bias"], 1e-5)

    # U/blocks/transformer.py:289-297
    def pos_ffn(self, p: str, x: np.ndarray) -> np.ndarray:
        sd = self.sd
        k1 = sd[p + ".w_1.weight"].shape[2]
        k2 = sd[p + ".w_2.weight"].shape[2]
        h = conv1d(x.transpose(0, 2, 1), sd[p + ".w_1.weight"], sd[p + ".w_1.bias"], padding=(k1 - 1) // 2)
        h = np.maximum(h, 0)
        h = conv1d(h, sd[p + ".w_2.weight"], sd[p + ".w_2.bias"], padding=(k2 - 1) // 2)
        return layer_norm(h.transpose(0, 2, 1) + x, sd[p + ".layer_norm.weight"], sd[p + ".layer_norm.bias"], 1e-5)

    # U/blocks/transformer.py:178-189
    def fft_block(self, p: str, x: np.ndarray, pad: np.ndarray) -> np.ndarray:
        x = self.mha(p + ".slf_attn", x, pad)
        x = np.where(pad[:, :, None], self.dt(0), x)
        x = self.pos_ffn(p + ".pos_ffn", x)
        return np.where(pad[:, :, None], self.dt(0), x)

    # ---- Conformer block (U/blocks/conformer.py:171-255); every sub-module is wrapped in ResidualConnectionModule (:258-270)
    # U/blocks/conformer.py:273-304: LayerNorm -> Linear -> Swish -> Linear (bias on both: LinearNorm(bias=True))
    def cf_ffn(self, p: str, x: np.ndarray) -> np.ndarray:
        sd = self.sd
        h = layer_norm(x, sd[p + ".0.weight"], sd[p + ".0.bias"], 1e-5)
        h = linear(h, sd[p + ".1.linear.weight"], sd[p + ".1.linear.bias"])
        h = h * (self.dt(1) / (self.dt(1) + np.exp(-h)))                 # Swish, U/blocks/utils.py:204-205
        return linear(h, sd[p + ".4.linear.weight"], sd[p + ".4.linear.bias"])

    # U/blocks/conformer.py:335-353 + 399-440.  nn.Sequential calls the module with ONE argument (:252), so mask is None: no key is
    # masked, padded rows take part.  Projections have no bias (LinearNorm default); the score is divided by sqrt(d_model) (:384, :418).
    def cf_mhsa(self, p: str, x: np.ndarray) -> np.ndarray:
        sd, nh = self.sd, self.n_head
        B, N, H = x.shape
        dh = H // nh
        pos = (sinusoid_table(N, H).astype(self.dt) if N > self.max_seq_len else sd[p + ".positional_encoding"][0, :N])  # :339-348
        y = layer_norm(x, sd[p + ".layer_norm.weight"], sd[p + ".layer_norm.bias"], 1e-5)
        a = p + ".attention."
        q = linear(y, sd[a + "query_proj.linear.weight"], None).reshape(B, N, nh, dh)
        k = linear(y, sd[a + "key_proj.linear.weight"], None).reshape(B, N, nh, dh).transpose(0, 2, 1, 3)
        v = linear(y, sd[a + "value_proj.linear.weight"], None).reshape(B, N, nh, dh).transpose(0, 2, 1, 3)
        pe = linear(pos, sd[a + "pos_proj.linear.weight"], None).reshape(N, nh, dh)
        content = np.matmul((q + sd[a + "u_bias"]).transpose(0, 2, 1, 3), k.transpose(0, 1, 3, 2))        # [B, nh, N, N]
        pscore = np.matmul((q + sd[a + "v_bias"]).transpose(0, 2, 1, 3), pe.transpose(1, 2, 0)[None])    # [B, nh, N, N]
        # _relative_shift (:432-440): prepend a zero column, view as [N + 1, N], drop the first row, view as [N, N]
        padded = np.concatenate([np.zeros((B, nh, N, 1), self.dt), pscore], axis=-1).reshape(B, nh, N + 1, N)
        pscore = padded[:, :, 1:].reshape(B, nh, N, N)
        score = (content + pscore) / self.dt(math.sqrt(H))
        ctx = np.matmul(softmax_lastdim(score), v).transpose(0, 2, 1, 3).reshape(B, N, H)
        return linear(ctx, sd[a + "out_proj.linear.weight"], None)

    # U/blocks/conformer.py:468-481: LayerNorm -> pointwise 2H -> GLU -> depthwise k (no bias) -> BatchNorm (eval) -> Swish -> pointwise
    def cf_conv(self, p: str, x: np.ndarray) -> np.ndarray:
        sd = self.sd
        h = layer_norm(x, sd[p + ".0.weight"], sd[p + ".0.bias"], 1e-5).transpose(0, 2, 1)
        h = conv1d(h, sd[p + ".2.conv.weight"], sd[p + ".2.conv.bias"])
        C = h.shape[1] // 2
        h = h[:, :C] * (self.dt(1) / (self.dt(1) + np.exp(-h[:, C:])))  # GLU(dim=1), U/blocks/utils.py:217-219
        w = sd[p + ".4.conv.weight"]                                    # [C, 1, k], groups = C
        k = w.shape[2]
        hp = np.pad(h, ((0, 0), (0, 0), ((k - 1) // 2, (k - 1) // 2)))
        acc = np.zeros_like(h)
        for j in range(k):
            acc += hp[:, :, j:j + h.shape[2]] * w[None, :, 0, j, None]
        inv = self.dt(1) / np.sqrt(sd[p + ".5.running_var"] + self.dt(1e-5))
        h = (acc - sd[p + ".5.running_mean"][None, :, None]) * inv[None, :, None] * sd[p + ".5.weight"][None, :, None] \
            + sd[p + ".5.bias"][None, :, None]
        h = h * (self.dt(1) / (self.dt(1) + np.exp(-h)))
        return conv1d(h, sd[p + ".7.conv.weight"], sd[p + ".7.conv.bias"]).transpose(0, 2, 1)

    def conformer_block(self, p: str, x: np.ndarray, pad: np.ndarray) -> np.ndarray:
        sd, s = self.sd, p + ".sequential"
        f = self.dt(0.5 if self.tr["half_step_residual"] else 1.0)      # :209-212
        x = self.cf_ffn(s + ".0.module.sequential", x) * f + x
        x = self.cf_mhsa(s + ".1.module", x) + x
        x = self.cf_conv(s + ".2.module.sequential", x) + x
        x = self.cf_ffn(s + ".3.module.sequential", x) * f + x
        x = layer_norm(x, sd[s + ".4.weight"], sd[s + ".4.bias"], 1e-5)
        return np.where(pad[:, :, None], self.dt(0), x)                 # :253-254

    def block(self, p: str, x: np.ndarray, pad: np.ndarray) -> np.ndarray:
        return self.conformer_block(p, x, pad) if self.bt == "conformer" else self.fft_block(p, x, pad)

    def _pos_enc(self, side: str, n: int) -> np.ndarray:
        # eval-time regeneration when the sequence exceeds max_seq_len
        # (U/blocks/transformer.py:68-77 encoder, :138-153 decoder)
        if n > self.max_seq_len:
            return sinusoid_table(n, self.H).astype(self.dt)
        return self.sd[side + ".position_enc"][0, :n]

    # U/blocks/transformer.py:58-86
    def encoder(self, ids: np.ndarray, pad: np.ndarray) -> np.ndarray:
        self.n_head = self.tr["encoder_head"]                            # U/blocks/transformer.py:29, conformer.py:31
        x = self.sd["encoder.src_word_emb.weight"][ids] + self._pos_enc("encoder", ids.shape[1])[None]
        for l in range(self.fs["encoder_layers"]):
            x = self.block(f"encoder.layer_stack.{l}", x, pad)
        return x

    # U/blocks/transformer.py:132-164
    def decoder(self, x: np.ndarray, pad: np.ndarray) -> np.ndarray:
        self.n_head = self.tr["decoder_head"]                            # U/blocks/transformer.py:105, conformer.py:108
        x = x + self._pos_enc("decoder", x.shape[1])[None]
        for l in range(self.fs["decoder_layers"]):
            x = self.block(f"decoder.layer_stack.{l}", x, pad)
        return x

    # ConstantPad1d in front of every predictor convolution (U/layers.py:400-402, :479-481): ((k-1)//2, (k-1)//2) for ffn_padding "SAME",
    # (k - 1, 0) -- causal -- otherwise
    def _pred_conv(self, xs: np.ndarray, w: np.ndarray, b: np.ndarray, k: int) -> np.ndarray:
        if self.fs["variance"]["variance_predictor"]["ffn_padding"] == "SAME":
            return conv1d(xs, w, b, padding=(k - 1) // 2)
        xp = np.concatenate([np.zeros(xs.shape[:2] + (k - 1,), xs.dtype), xs], axis=2)
        return conv1d(xp, w, b, padding=0)

    # U/layers.py:410-420 (ctor :382-408); channel LayerNorm eps 1e-12: U/sublayers.py:151-170
    def duration_predictor(self, x: np.ndarray, pad: np.ndarray) -> np.ndarray:
        sd, p = self.sd, "variance_adaptor.duration_predictor"
        keep = (1 - pad.astype(self.dt))
        xs = x.transpose(0, 2, 1)
        for i in range(self.fs["variance"]["variance_predictor"]["dur_predictor_layers"]):
            k = sd[f"{p}.conv.{i}.1.weight"].shape[2]
            xs = self._pred_conv(xs, sd[f"{p}.conv.{i}.1.weight"], sd[f"{p}.conv.{i}.1.bias"], k)
            xs = np.maximum(xs, 0)
            xs = layer_norm(xs.transpose(0, 2, 1), sd[f"{p}.conv.{i}.3.weight"], sd[f"{p}.conv.{i}.3.bias"], 1e-12).transpose(0, 2, 1)
            xs = xs * keep[:, None, :]
        out = linear(xs.transpose(0, 2, 1), sd[p + ".linear.weight"], sd[p + ".linear.bias"])
        out = out * keep[:, :, None]
        return out[..., 0]

    # U/layers.py:491-505 (ctor :463-489); positions: U/sublayers.py:46-65, U/function.py:28-38
    def variance_predictor(self, which: str, x: np.ndarray) -> np.ndarray:
        sd, p = self.sd, f"variance_adaptor.{which}_predictor"
        positions = make_positions(x[..., 0], 0)
        table = self.var_pos_table
        need = 1 + x.shape[1]
        if need > table.shape[0]:
            table = fairseq_sinusoid_table(need, self.H, 0)
        pos_emb = table[positions].astype(self.dt)
        xs = x + sd[p + ".pos_embed_alpha"] * pos_emb
        xs = xs.transpose(0, 2, 1)
        n_layers = self.fs["variance"]["variance_predictor"]["pit_predictor_layers" if which == "pitch" else "ener_predictor_layers"]
        for i in range(n_layers):
            k = sd[f"{p}.conv.{i}.1.weight"].shape[2]
            xs = self._pred_conv(xs, sd[f"{p}.conv.{i}.1.weight"], sd[f"{p}.conv.{i}.1.bias"], k)
            xs = np.maximum(xs, 0)
            xs = layer_norm(xs.transpose(0, 2, 1), sd[f"{p}.conv.{i}.3.weight"], sd[f"{p}.conv.{i}.3.bias"], 1e-12).transpose(0, 2, 1)
        return linear(xs.transpose(0, 2, 1), sd[p + ".linear.weight"], sd[p + ".linear.bias"])

    # U/function.py:178-187
    def f0_to_coarse(self, f0: np.ndarray) -> np.ndarray:
        dt = self.dt
        f0_mel = dt(1127) * np.log(dt(1) + f0 / dt(700))
        pos = f0_mel > 0
        scaled = (f0_mel - dt(F0_MEL_MIN)) * dt(F0_BIN - 2) / dt(F0_MEL_MAX - F0_MEL_MIN) + dt(1)
        f0_mel = np.where(pos, scaled, f0_mel)
        f0_mel = np.where(f0_mel <= 1, dt(1), f0_mel)
        f0_mel = np.where(f0_mel > F0_BIN - 1, dt(F0_BIN - 1), f0_mel)
        return (f0_mel + dt(0.5)).astype(np.int64)  # .long(): truncation

    # U/layers.py:136-162
    def pitch_embedding(self, x: np.ndarray, control: float) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        dt = self.dt
        ve = self.fs["variance"]["variance_embedding"]
        if not ve["use_uv"]:                                              # :138,155-157: squeeze, bucketize(prediction * control, pitch_bins);
            pred = self.variance_predictor("pitch", x)[..., 0]            # the prediction handed back is the one BEFORE the control (:162)
            idx = np.searchsorted(self.sd["variance_adaptor.pitch_bins"], pred * dt(control), side="left").astype(np.int64)
            return pred, idx, self.sd["variance_adaptor.pitch_embedding.weight"][idx]
        pred = self.variance_predictor("pitch", x) * dt(control)
        f0 = pred[:, :, 0]
        uv = pred[:, :, 1] > 0
        if ve["pitch_quantization"] == "log":                             # :148-149
            f0_denorm = np.power(dt(2), f0).astype(self.dt)
        else:
            f0_denorm = f0 * dt(self.stats["f0"]["std"]) + dt(self.stats["f0"]["mean"])
        f0_denorm = np.where(uv, dt(0), f0_denorm)
        idx = self.f0_to_coarse(f0_denorm)
        return pred, idx, self.sd["variance_adaptor.pitch_embedding.weight"][idx]

    # U/layers.py:164-173
    def energy_embedding(self, x: np.ndarray, control: float) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        pred = self.variance_predictor("energy", x)[..., 0]
        energy = pred * self.dt(control)
        idx = np.searchsorted(self.sd["variance_adaptor.energy_bins"], energy, side="left").astype(np.int64)  # bucketize(right=False)
        return pred, idx, self.sd["variance_adaptor.energy_embedding.weight"][idx]

    # U/layers.py:423-457 + U/function.py:75-93
    @staticmethod
    def length_regulator(x: np.ndarray, duration: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        B, L, H = x.shape
        reps = np.maximum(duration.astype(np.int64), 0)   # max(int(expand_size), 0)
        mel_lens = reps.sum(axis=1)
        T = int(mel_lens.max())
        out = np.zeros((B, T, H), dtype=x.dtype)
        for b in range(B):
            out[b, :mel_lens[b]] = np.repeat(x[b], reps[b], axis=0)
        return out, mel_lens.astype(np.int64)

    # U/layers.py:556-563 (ctor :514-554); BatchNorm1d eval with running stats, eps 1e-5
    def postnet(self, mel: np.ndarray) -> np.ndarray:
        sd = self.sd
        n = self.fs["postnet"]["conv_layers"]
        k = self.fs["postnet"]["kernel_size"]
        x = mel.transpose(0, 2, 1)
        for i in range(n):
            p = f"postnet.convolutions.{i}"
            x = conv1d(x, sd[p + ".0.conv.weight"], sd[p + ".0.conv.bias"], padding=int((k - 1) / 2))
            x = (x - sd[p + ".1.running_mean"][None, :, None]) / np.sqrt(sd[p + ".1.running_var"][None, :, None] + self.dt(1e-5))
            x = x * sd[p + ".1.weight"][None, :, None] + sd[p + ".1.bias"][None, :, None]
            if i < n - 1:
                x = np.tanh(x)
        return x.transpose(0, 2, 1)

    def inference(self, speaker: np.ndarray, ids: np.ndarray, lens: np.ndarray,
                  d_control: float = 1.0, p_control: float = 1.0, e_control: float = 1.0):
        """-> ((mel, mel_post, duration_rounded), mel_lens); intermediates in ``self.trace``."""
        dt = self.dt
        ids = np.asarray(ids, dtype=np.int64)
        lens = np.asarray(lens, dtype=np.int64)
        pad = get_mask_from_lengths(lens, ids.shape[1])
        x = self.encoder(ids, pad)
        self.trace["enc_out"] = x
        spk = self.sd["speaker_emb.weight"][np.asarray(speaker, dtype=np.int64)]        # [1 or B, H]
        x = x + spk[:, None, :]                                                           # U/layers.py:195
        log_d = self.duration_predictor(x, pad)
        # U/layers.py:218-221: round half-to-even, control applied after rounding, clamp >= 0
        dur = np.maximum(np.round(np.exp(log_d) - dt(1)) * dt(d_control), dt(0))
        ve = self.fs["variance"]["variance_embedding"]
        p_frame, e_frame = ve["pitch_feature"] == "frame_level", ve["energy_feature"] == "frame_level"
        x_tmp = x                                                                         # U/layers.py:225-239: phoneme_level features
        if not p_frame:
            p_pred, p_idx, p_emb = self.pitch_embedding(x, p_control)
            x_tmp = x_tmp + p_emb
        if not e_frame:
            e_pred, e_idx, e_emb = self.energy_embedding(x, e_control)
            x_tmp = x_tmp + e_emb
        x, mel_lens = self.length_regulator(x_tmp, dur)
        x_tmp = x                                                                         # :248-257: frame_level features, both predicted from
        if p_frame:                                                                       # the regulator's output, padded rows included
            p_pred, p_idx, p_emb = self.pitch_embedding(x, p_control)
            x_tmp = x_tmp + p_emb
        if e_frame:
            e_pred, e_idx, e_emb = self.energy_embedding(x, e_control)
            x_tmp = x_tmp + e_emb
        x = x_tmp
        self.trace.update(log_d=log_d, pitch_pred=p_pred, pitch_idx=p_idx, energy_pred=e_pred, energy_idx=e_idx, lr_out=x)
        mel_pad = get_mask_from_lengths(mel_lens)
        x = self.decoder(x, mel_pad)
        self.trace["dec_out"] = x
        mel = linear(x, self.sd["mel_linear.weight"], self.sd["mel_linear.bias"])       # U/model.py:186
        mel_post = self.postnet(mel) + mel                                               # U/model.py:188
        return (mel, mel_post, dur), mel_lens


# --------------------------------------------------------------------------- vocoder

def fold_weight_norm(g: np.ndarray, v: np.ndarray) -> np.ndarray:
    """weight_norm(dim=0): w = g * v / ||v|| with the norm over all dims but 0 (V/generator.py:18,23,33)."""
    v64 = v.astype(np.float64)
    norm = np.sqrt((v64 * v64).sum(axis=tuple(range(1, v.ndim)), keepdims=True))
    return (g.astype(np.float64) * v64 / norm).astype(v.dtype)


class VocoderOracle:
    """HifiGan.forward restated (V/generator.py:37-53) with ResBlock1 (V/layers.py:33-40)."""

    def __init__(self, state: Dict[str, np.ndarray], config: dict, dtype=np.float32):
        self.dt = np.dtype(dtype).type
        hg = config["models"]["hifigan"]
        self.hg = hg
        sd = {k: v.astype(dtype) for k, v in state.items()}
        self.w: Dict[str, np.ndarray] = {}
        for k in sd:
            if k.endswith(".weight_v"):
                p = k[: -len(".weight_v")]
                self.w[p + ".weight"] = fold_weight_norm(sd[p + ".weight_g"], sd[k])
                self.w[p + ".bias"] = sd[p + ".bias"]
            elif k.endswith(".weight"):      # already-folded checkpoints (remove_weight_norm, V/generator.py:55-62)
                self.w[k] = sd[k]
                self.w[k[:-7] + ".bias"] = sd[k[:-7] + ".bias"]

    def resblock(self, idx: int, x: np.ndarray, k: int, dils: Sequence[int]) -> np.ndarray:
        w = self.w
        for m, d in enumerate(dils):
            xt = leaky_relu(x, LRELU_SLOPE)
            xt = conv1d(xt, w[f"resblocks.{idx}.convs1.{m}.weight"], w[f"resblocks.{idx}.convs1.{m}.bias"],
                        padding=int((k * d - d) / 2), dilation=d)
            xt = leaky_relu(xt, LRELU_SLOPE)
            xt = conv1d(xt, w[f"resblocks.{idx}.convs2.{m}.weight"], w[f"resblocks.{idx}.convs2.{m}.bias"],
                        padding=int((k - 1) / 2), dilation=1)
            x = xt + x
        return x

    def forward(self, mel_bct: np.ndarray) -> np.ndarray:
        """mel [B, 80, T] -> wav [B, 1, T * hop]."""
        hg, w = self.hg, self.w
        x = conv1d(mel_bct.astype(self.dt), w["conv_pre.weight"], w["conv_pre.bias"], padding=3)
        nk = len(hg["resblock_kernel_sizes"])
        for i, (u, k) in enumerate(zip(hg["upsample_rates"], hg["upsample_kernel_sizes"])):
            x = leaky_relu(x, LRELU_SLOPE)
            x = conv_transpose1d(x, w[f"ups.{i}.weight"], w[f"ups.{i}.bias"], stride=u, padding=(k - u) // 2)
            xs = None
            for j in range(nk):
                r = self.resblock(i * nk + j, x, hg["resblock_kernel_sizes"][j], hg["resblock_dilation_sizes"][j])
                xs = r if xs is None else xs + r
            x = xs / self.dt(nk)
        x = leaky_relu(x, 0.01)  # F.leaky_relu default slope (V/generator.py:49)
        x = conv1d(x, w["conv_post.weight"], w["conv_post.bias"], padding=3)
        return np.tanh(x)


class IstftOracle(VocoderOracle):
    """iSTFT.forward restated (V/generator.py:96-113) plus the inverse STFT the reference applies to its outputs
    (src/tools/stft.py:138-148 = torch.istft with a periodic Hann window, center=True).

    ResBlock1 only when ``config['resblock'] == '1'`` -- the STRING, as the reference compares (V/generator.py:71); the yaml
    ships the int 1, which selects ResBlock2 (V/layers.py:49-66, dilations d[0], d[1])."""

    def __init__(self, state: Dict[str, np.ndarray], config: dict, dtype=np.float32):
        cfg = {"models": {"hifigan": config["models"]["istft"]}}
        super().__init__(state, cfg, dtype)
        self.n_fft = int(self.hg["gen_istft_n_fft"])
        self.hop = int(self.hg["gen_istft_hop_size"])
        self.rb1 = self.hg["resblock"] == "1"

    def resblock2(self, idx: int, x: np.ndarray, k: int, dils: Sequence[int]) -> np.ndarray:
        w = self.w
        for m in range(2):
            d = dils[m]
            xt = leaky_relu(x, LRELU_SLOPE)
            xt = conv1d(xt, w[f"resblocks.{idx}.convs.{m}.weight"], w[f"resblocks.{idx}.convs.{m}.bias"],
                        padding=int((k * d - d) / 2), dilation=d)
            x = xt + x
        return x

    def forward(self, mel_bct: np.ndarray):
        """mel [B, 80, T] -> (spec [B, n_fft/2 + 1, F], phase [B, n_fft/2 + 1, F]), F = T * prod(upsample_rates) + 1."""
        hg, w = self.hg, self.w
        x = conv1d(mel_bct.astype(self.dt), w["conv_pre.weight"], w["conv_pre.bias"], padding=3)
        nk = len(hg["resblock_kernel_sizes"])
        for i, (u, k) in enumerate(zip(hg["upsample_rates"], hg["upsample_kernel_sizes"])):
            x = leaky_relu(x, LRELU_SLOPE)
            x = conv_transpose1d(x, w[f"ups.{i}.weight"], w[f"ups.{i}.bias"], stride=u, padding=(k - u) // 2)
            xs = None
            for j in range(nk):
                rb = self.resblock if self.rb1 else self.resblock2
                r = rb(i * nk + j, x, hg["resblock_kernel_sizes"][j], hg["resblock_dilation_sizes"][j])
                xs = r if xs is None else xs + r
            x = xs / self.dt(nk)
        x = leaky_relu(x, 0.01)                                   # F.leaky_relu default slope (:107)
        x = np.concatenate([x[:, :, 1:2], x], axis=2)             # nn.ReflectionPad1d((1, 0)) (:94, :108)
        x = conv1d(x, w["conv_post.weight"], w["conv_post.bias"], padding=3)
        bins = self.n_fft // 2 + 1
        return np.exp(x[:, :bins, :]), np.sin(x[:, bins:, :])      # (:110-111)

    def inverse(self, spec: np.ndarray, phase: np.ndarray) -> np.ndarray:
        """torch.istft(spec * exp(1j * phase), n_fft, hop, win_length = n_fft, window = hann_window(n_fft)) -> [B, 1, hop (F - 1)]
        (src/tools/stft.py:138-148): per-frame irfft, times the window, overlap-add, divided by the overlap-added squared
        window, n_fft // 2 samples trimmed at both ends (center=True)."""
        n, hop = self.n_fft, self.hop
        B, _, F = spec.shape
        X = spec.astype(np.float64) * np.exp(1j * phase.astype(np.float64))
        frames = np.fft.irfft(X, n=n, axis=1)                       # [B, n, F]
        win = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)    # torch.hann_window(n): periodic
        total = n + hop * (F - 1)
        y = np.zeros((B, total))
        env = np.zeros(total)
        for m in range(n):                                          # overlap-add, one window position at a time
            y[:, m:m + hop * F:hop][:, :F] += frames[:, m, :] * win[m]
            env[m:m + hop * F:hop][:F] += win[m] ** 2
        y = y[:, n // 2:n // 2 + hop * (F - 1)] / env[n // 2:n // 2 + hop * (F - 1)]
        return y[:, None, :].astype(self.dt)

    def wav(self, mel_bct: np.ndarray) -> np.ndarray:
        return self.inverse(*self.forward(mel_bct))


# --------------------------------------------------------------------------- host loop (API/utils.py)

def arrange_text(text: List[str], max_len: int = 300) -> List[str]:
    """API/utils.py:64-80."""
    arranged: List[str] = []
    for line in text:
        if round(len(line) / max_len) != 1:
            parts = line.split(" , ")
            arranged.append(parts[0])
            parts.pop(0)
            while len(parts) > 0:
                if len(arranged[-1]) >= max_len:
                    arranged.append(parts[0])
                else:
                    arranged[-1] = " , ".join([arranged[-1], parts[0]])
                parts.pop(0)
        else:
            arranged.append(line)
    return arranged


def pack_batches(seq_lens: Sequence[int], max_len: int = 300):
    """Batching of API/utils.py:84-99 on sequence lengths only.

    Returns (order, revert_indices, [(s, e), ...]) where ``order`` is the stable
    descending sort, and each (s, e) slices the sorted list.  Reproduces the
    reference quirk that the item which overflows the token budget is not
    counted into the next batch's total (:96).
    """
    lens = np.asarray(seq_lens, dtype=np.int64)
    order = np.argsort(-lens, kind="stable")
    revert = np.argsort(order, kind="stable")
    sorted_lens = lens[order]
    spans = []
    s = e = total = 0
    for i, n in enumerate(sorted_lens):
        if s == e or total + n <= max_len:
            e = i + 1
            total += int(n)
        else:
            spans.append((s, e))
            s, total = e, 0
    if not spans or spans[-1][1] != len(lens):
        spans.append((s, len(lens)))
    return order, revert, spans


def combine_audio(audios: Sequence[np.ndarray], lengths: Sequence[int], distance: int, hop_length: int = 256,
                  max_wav_value: float = 32768.0) -> np.ndarray:
    """API/utils.py:108-117: trim to len*hop, x32768, append ``distance`` zeros after every utterance, -> int16."""
    out = []
    for i, audio in enumerate(audios):
        a = audio[: int(lengths[i]) * hop_length]
        a = a * max_wav_value
        out.extend([a, np.zeros(distance)])
    return np.concatenate(out).astype("int16")


def synthesize(acoustic: AcousticOracle, vocoder: VocoderOracle, batches, speaker: int, revert: np.ndarray,
               silence: int, hop: int = 256, controls=(1.0, 1.0, 1.0)) -> np.ndarray:
    """TTS.inference restated (API/utils.py:119-160) on pre-tokenised batches [(ids [B,L], lens [B]), ...]."""
    audios, lengths = [], []
    for ids, lens in batches:
        (mel, mel_post, dur), mel_lens = acoustic.inference(np.array([speaker]), ids, lens,
                                                            d_control=controls[2], p_control=controls[0], e_control=controls[1])
        wav = vocoder.forward(mel_post.transpose(0, 2, 1))[:, 0]
        audios.extend(list(wav))
        lengths.extend(list(mel_lens))
    audios = [audios[i] for i in revert.tolist()]
    lengths = [lengths[i] for i in revert.tolist()]
    return combine_audio(audios, lengths, silence, hop)
