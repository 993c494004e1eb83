"""Drop-in mirrors of the reference's two model classes, backed by the HIP engine.

``UnsupervisedFastSpeech2`` mirrors reference e2e_tts/models/acoustic/unsupervised_fastspeech2/model.py:8-68,
155-194 and ``HifiGan`` mirrors e2e_tts/models/vocoder/generator.py:13-62: same constructor arguments, same
``load_state_dict`` / ``eval`` / ``to`` call sequence that ``TTS.__init__`` performs (API/utils.py:41-56), same
``inference`` / ``forward`` signatures and return structure (torch tensors, on the engine's GPU).

They are NOT nn.Modules and hold no torch parameters: ``load_state_dict`` packs the checkpoint into the
engine's HBM image.  Training-side methods (``forward`` with targets, ``parse_batch``) are out of scope.
"""
from __future__ import annotations

from typing import Mapping, Optional

import numpy as np

from . import packer
from ._lib import Engine
from .config import EngineDims, dims_from_config, default_config


def _torch():
    import torch
    return torch


def _device_index(device) -> int:
    if device is None:
        return 0
    if isinstance(device, int):
        return device
    torch = _torch()
    d = torch.device(device)
    if d.type != "cuda":
        raise RuntimeError(f"e2e_tts_amd runs on MI355X GPUs only; device={device!r} was requested (no CPU fallback)")
    return d.index or 0


class _EngineBacked:
    """Shared plumbing: one Engine, (re)created lazily on the requested GPU, weights packed on load."""

    def __init__(self):
        self._engine: Optional[Engine] = None
        self._device = 0
        self._blob = None
        self.training = False

    def _dims(self) -> EngineDims:
        raise NotImplementedError

    def _pack(self, state) -> np.ndarray:
        raise NotImplementedError

    def _ensure_engine(self) -> Engine:
        if self._engine is None:
            self._engine = Engine(self._dims(), self._device)
            if self._blob is not None:
                self._engine.load_weights(self._blob)
        return self._engine

    # nn.Module-style surface used by TTS.__init__
    def load_state_dict(self, state_dict: Mapping[str, object], strict: bool = True):
        self._blob = self._pack(state_dict)
        if self._engine is not None:
            self._engine.load_weights(self._blob)
        return self

    def eval(self):
        self.training = False
        return self

    def train(self, mode: bool = True):
        if mode:
            raise NotImplementedError("the HIP engine is inference only")
        return self

    def to(self, device):
        idx = _device_index(device)
        if idx != self._device and self._engine is not None:
            self._engine.close()
            self._engine = None
        self._device = idx
        self._ensure_engine()
        return self

    def cuda(self, device=None):
        return self.to(device if device is not None else 0)

    @property
    def engine(self) -> Engine:
        return self._ensure_engine()


class UnsupervisedFastSpeech2(_EngineBacked):
    def __init__(self, n_symbols: int, n_speakers: int, n_channels: int, config: dict, stats: dict, device=None,
                 hop_length: int = 256, sampling_rate: int = 22050, hifigan_config: Optional[dict] = None,
                 pos_table_rows: int = 4096):
        """``config`` is the ``models.fastspeech2`` sub-dictionary, exactly as the reference passes it
        (API/utils.py:41-47).  The extra keyword arguments only matter when this object shares an engine with a
        vocoder (see ``e2e_tts_amd.api.TTS``)."""
        super().__init__()
        self.config = config
        self.stats = stats
        self.n_symbols, self.n_speakers, self.n_channels = n_symbols, n_speakers, n_channels
        full = default_config()
        full["models"]["fastspeech2"] = config
        if hifigan_config is not None:
            full["models"]["hifigan"] = hifigan_config
        full["audio"]["mel"]["channels"] = n_channels
        full["audio"]["stft"]["hop_length"] = hop_length
        full["audio"]["signal"]["sampling_rate"] = sampling_rate
        self._full_config = full
        self._dims_cache = dims_from_config(full, stats, n_speakers, n_symbols, pos_table_rows)
        self._device = _device_index(device)

    def _dims(self) -> EngineDims:
        return self._dims_cache

    def _pack(self, state) -> np.ndarray:
        return packer.pack(self._dims_cache, state, None)

    def inference(self, speaker, texts, txt_lens, max_txt_len=None, d_control: float = 1.0, p_control: float = 1.0,
                  e_control: float = 1.0):
        """-> ((mel [B, T, n_mel], mel_post [B, T, n_mel], duration_rounded [B, L] fp32), mel_lens [B] int64),
        torch tensors on the engine's GPU (reference U/model.py:155-194).  ``max_txt_len`` is accepted for
        signature compatibility; like the reference's mask it must equal texts.shape[1]."""
        torch = _torch()
        eng = self._ensure_engine()
        dev = torch.device("cuda", self._device)
        ids = torch.as_tensor(texts, dtype=torch.int64).contiguous()
        lens = torch.as_tensor(txt_lens, dtype=torch.int64).contiguous()
        spk = torch.as_tensor(speaker, dtype=torch.int64).reshape(-1).contiguous()
        if max_txt_len is not None and int(max_txt_len) != ids.shape[1]:
            raise ValueError(f"max_txt_len={int(max_txt_len)} != texts.shape[1]={ids.shape[1]}")
        with eng.lock:  # the mel tensors are the RESIDENT result of this acoustic() call: no other thread's call in between
            r = eng.acoustic(ids, lens, spk, d_control, p_control, e_control, want=("dur", "mel_lens"))
            B, T = r["B"], r["T"]
            mel = torch.empty((B, T, self.n_channels), dtype=torch.float32, device=dev)
            mel_post = torch.empty_like(mel)
            eng.fetch_mel(B, T, out_mel=mel, out_mel_post=mel_post)
        dur = torch.from_numpy(r["dur"]).to(dev)
        mel_lens = torch.from_numpy(r["mel_lens"]).to(dev)
        return (mel, mel_post, dur), mel_lens

    def forward(self, *a, **k):
        raise NotImplementedError("training forward is out of scope; use .inference()")

    __call__ = forward


def _vocoder_only_dims(hifigan_config: dict, n_mel: int = 80, vocoder: str = "hifigan") -> EngineDims:
    cfg = default_config()
    cfg["models"][vocoder] = hifigan_config
    cfg["audio"]["mel"]["channels"] = n_mel
    hop = 1
    for r in hifigan_config["upsample_rates"]:
        hop *= r
    if vocoder == "istft":
        hop *= hifigan_config["gen_istft_hop_size"]
    cfg["audio"]["stft"]["hop_length"] = hop
    from .config import DEFAULT_STATS
    return dims_from_config(cfg, DEFAULT_STATS, n_speakers=1, vocoder=vocoder)


class HifiGan(_EngineBacked):
    def __init__(self, config: dict, device=None, _shared: Optional[_EngineBacked] = None):
        """``config`` is the ``models.hifigan`` sub-dictionary (reference V/generator.py:14)."""
        super().__init__()
        self.config = config
        self.num_kernels = len(config["resblock_kernel_sizes"])
        self.num_upsamples = len(config["upsample_rates"])
        self._dims_cache = _vocoder_only_dims(config)
        self._device = _device_index(device)

    def _dims(self) -> EngineDims:
        return self._dims_cache

    def _pack(self, state) -> np.ndarray:
        return packer.pack(self._dims_cache, None, state)

    def remove_weight_norm(self):
        """No-op: weight norm is folded when the checkpoint is packed (reference V/generator.py:55-62)."""
        return None

    def forward(self, x):
        """x [B, 80, T] (torch tensor, any device, or numpy) -> wav [B, 1, T * hop] on the GPU (V/generator.py:37-53)."""
        torch = _torch()
        eng = self._ensure_engine()
        dev = torch.device("cuda", self._device)
        x = torch.as_tensor(x, dtype=torch.float32).contiguous()
        if x.dim() != 3 or x.shape[1] != self._dims_cache.n_mel:
            raise ValueError(f"expected mel of shape [B, {self._dims_cache.n_mel}, T], got {tuple(x.shape)}")
        B, _, T = x.shape
        wav = torch.empty((B, T * self._dims_cache.hop_length), dtype=torch.float32, device=dev)
        eng.vocoder(x, B, T, channels_first=True, out_wav=wav)
        return wav.unsqueeze(1)

    __call__ = forward


class iSTFT(HifiGan):
    """Mirror of reference ``models.vocoder.iSTFT`` (V/generator.py:65-118; iSTFTNet): same constructor,
    ``load_state_dict`` / ``eval`` / ``to`` / ``remove_weight_norm``, ``forward(x) -> (spec, phase)``.

    ``inference(x)`` additionally returns the waveform the reference obtains with ``inverse_stft(spec, phase, n_fft, hop,
    win)`` (src/tools/stft.py:138-148) -- on the engine the exp / sin heads, the per-frame inverse DFT, the window and the
    overlap-add are one tail after conv_post, so the waveform needs no second call.  ResBlock selection reproduces the
    reference's comparison with the string '1' (V/generator.py:71): the shipped yaml's integer 1 selects ResBlock2."""

    def __init__(self, config: dict, device=None):
        _EngineBacked.__init__(self)
        self.config = config
        self.num_kernels = len(config["resblock_kernel_sizes"])
        self.num_upsamples = len(config["upsample_rates"])
        self.post_n_fft = config["gen_istft_n_fft"]
        self._dims_cache = _vocoder_only_dims(config, vocoder="istft")
        self._device = _device_index(device)

    def _run(self, x):
        torch = _torch()
        eng = self._ensure_engine()
        dev = torch.device("cuda", self._device)
        x = torch.as_tensor(x, dtype=torch.float32).contiguous()
        if x.dim() != 3 or x.shape[1] != self._dims_cache.n_mel:
            raise ValueError(f"expected mel of shape [B, {self._dims_cache.n_mel}, T], got {tuple(x.shape)}")
        B, _, T = x.shape
        wav = torch.empty((B, T * self._dims_cache.hop_length), dtype=torch.float32, device=dev)
        eng.vocoder(x, B, T, channels_first=True, out_wav=wav)
        return eng, B, T, wav

    def forward(self, x):
        """x [B, 80, T] -> (spec [B, n_fft/2 + 1, F], phase [B, n_fft/2 + 1, F]) on the GPU, F = T * prod(upsample_rates) + 1."""
        torch = _torch()
        up = 1
        for r in self.config["upsample_rates"]:
            up *= r
        with self._ensure_engine().lock:  # the tap belongs to this vocoder call
            eng, B, T, _ = self._run(x)
            F, bins = T * up + 1, self.post_n_fft // 2 + 1
            sp = torch.empty((B, F, 2 * bins), dtype=torch.float32, device=torch.device("cuda", self._device))
            eng.fetch_tap_into("istft_spec_phase", sp)
        sp = sp.transpose(1, 2)
        return sp[:, :bins, :], sp[:, bins:, :]

    __call__ = forward

    def inference(self, x):
        """x [B, 80, T] -> wav [B, 1, T * hop] (= inverse_stft(*forward(x)))."""
        return self._run(x)[3].unsqueeze(1)
