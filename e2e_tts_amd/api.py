"""Drop-in mirrors of the reference's host loop: ``TTS`` (reference e2e_tts/src/api/utils.py:22-160) and the
service wrapper ``Synthesizer`` (e2e_tts/src/api/inference.py:12-50).

Same constructor arguments, attributes (``hop_length``, ``sample_rate``, ``max_wav_value``, ``speakers``,
``config``, ``stats``, ``max_len``) and method signatures.  Differences, all deliberate:

* one HIP engine holds both models, so the mel never leaves HBM between acoustic model and vocoder and the
  only device->host traffic per batch is the int16 PCM (the reference copies fp32 audio, API/utils.py:145);
* ``input_parse`` sorts with a *stable* descending sort (torch.sort(descending=True) at :84 leaves the order of
  equal-length sentences unspecified; batches and audio are identical, only ties may be numbered differently);
* text -> phoneme ids defaults to ``e2e_tts_amd.g2p.text_to_sequence`` (a working restatement of the reference's
  Vietnamese g2p, pinned by a fixture converted with the reference's own code; the reference's ``text_to_sequence``
  cannot run as shipped: cleaners.py:12,26-30 recurses with a bad kwarg) and stays pluggable (``text_to_sequence=``);
* no network call to a text normaliser (API/inference.py:28-33 swallows its failure anyway), no upload;
* ``speed != 1`` is served by the model's duration control instead of an ffmpeg ``atempo`` subprocess (``audio_speed_change``
  keeps the reference's file-level signature on a WSOLA restatement; ffmpeg is absent, so that function is parity-unpinned).
"""
from __future__ import annotations

import json
import os
import time
import wave
from datetime import datetime
from typing import Callable, List, Optional, Sequence

import numpy as np

from . import packer
from ._lib import Engine
from .config import N_SYMBOLS, dims_from_config
from .models import HifiGan, UnsupervisedFastSpeech2, _device_index


def _load_state(path: str):
    import torch
    ckpt = torch.load(path, map_location="cpu", weights_only=True)  # tensors only: nothing in the file is executed
    return ckpt["state_dict"]


class TTS:
    def __init__(self, acoustic_path: str, vocoder_path: str, max_len: int = 300, device=None,
                 text_to_sequence: Optional[Callable[[str], Sequence[int]]] = None, n_symbols: int = N_SYMBOLS,
                 pos_table_rows: int = 4096):
        import yaml
        self.device = device
        self._device_index = _device_index(device)
        base = os.path.dirname(acoustic_path)
        # the three side-car files written by save_information (reference tools_for_model.py:143-152)
        self.config = yaml.safe_load(open(os.path.join(base, "config.yaml"), "r"))  # plain scalars / lists / dicts only
        self.speakers = json.load(open(os.path.join(base, "speakers.json"), "r"))
        self.stats = json.load(open(os.path.join(base, "stats.json"), "r"))
        if self.config["models"]["fastspeech2"]["variance"]["duration_modelling"]["learn_alignment"] is not True:
            raise NotImplementedError("SupervisedFastSpeech2 checkpoints are out of scope (reference API/utils.py:37-40)")
        self._dims = dims_from_config(self.config, self.stats, len(self.speakers), n_symbols, pos_table_rows)
        self._acoustic_state = _load_state(acoustic_path)
        self._vocoder_state = _load_state(vocoder_path)
        self.engine = Engine(self._dims, self._device_index)
        self.engine.load_weights(packer.pack(self._dims, self._acoustic_state, self._vocoder_state))
        self.hop_length = self.config["audio"]["stft"]["hop_length"]
        self.sample_rate = self.config["audio"]["signal"]["sampling_rate"]
        self.max_wav_value = 32768.0
        self.max_len = max_len
        if text_to_sequence is None:
            from .g2p import text_to_sequence as _default_g2p
            text_to_sequence = _default_g2p
        self.text_to_sequence = text_to_sequence
        self._acoustic = None
        self._vocoder = None

    # the reference exposes the two modules as attributes; build the stand-alone mirrors only on demand
    @property
    def acoustic(self) -> UnsupervisedFastSpeech2:
        if self._acoustic is None:
            m = UnsupervisedFastSpeech2(self._dims.n_symbols, len(self.speakers), self._dims.n_mel,
                                        self.config["models"]["fastspeech2"], self.stats, device=self._device_index,
                                        hop_length=self.hop_length, sampling_rate=self.sample_rate)
            m.load_state_dict(self._acoustic_state)
            self._acoustic = m.eval().to(self._device_index)
        return self._acoustic

    @property
    def vocoder(self) -> HifiGan:
        if self._vocoder is None:
            v = HifiGan(self.config["models"]["hifigan"], device=self._device_index)
            v.load_state_dict(self._vocoder_state)
            self._vocoder = v.eval().to(self._device_index)
        return self._vocoder

    # ---- reference API/utils.py:64-80
    def arrange_text(self, text: List[str]) -> List[str]:
        arranged_text: List[str] = []
        for line in text:
            if round(len(line) / self.max_len) != 1:
                pieces = line.split(" , ")
                arranged_text.append(pieces[0])
                for piece in pieces[1:]:
                    if len(arranged_text[-1]) >= self.max_len:
                        arranged_text.append(piece)
                    else:
                        arranged_text[-1] = " , ".join([arranged_text[-1], piece])
            else:
                arranged_text.append(line)
        return arranged_text

    @staticmethod
    def pack_sequences(sequences: Sequence[Sequence[int]], max_len: int):
        """The batching arithmetic of reference API/utils.py:84-104 on already-tokenised sequences.

        Sort by length (descending, stable), then cut greedily into batches whose length sum stays <= max_len;
        the sentence that overflows a batch opens the next one WITHOUT being counted in its total (the
        reference's `s, total_lens = e, 0` at :96-99 -- reproduced, it decides the batch composition and the
        padded-batch results depend on it).  Returns ([(ids [B, L] int64, lens [B] int64), ...], revert_indices).
        """
        lens = np.asarray([len(s) for s in sequences], dtype=np.int64)
        order = np.argsort(-lens, kind="stable")
        revert = np.argsort(order, kind="stable")
        spans = []
        s = e = total = 0
        for i, n in enumerate(lens[order]):
            if s == e or total + n <= max_len:
                e = i + 1
                total += int(n)
            else:
                spans.append((s, e))
                s, total = e, 0
        if not spans or spans[-1][1] != len(lens):
            spans.append((s, len(lens)))
        batches = []
        for s, e in spans:
            rows = [sequences[j] for j in order[s:e]]
            L = max(len(r) for r in rows)
            ids = np.zeros((len(rows), L), dtype=np.int64)  # pad_sequence(batch_first=True): zero padding
            for r, row in enumerate(rows):
                ids[r, :len(row)] = np.asarray(row, dtype=np.int64)
            batches.append((ids, lens[order[s:e]].copy()))
        return batches, revert

    def input_parse(self, input_texts: List[str]):
        """-> (list of [ids LongTensor [B, L], lens LongTensor [B]], revert_indices LongTensor) like the reference."""
        import torch
        if self.text_to_sequence is None:
            raise RuntimeError("TTS was built without a text front-end: pass text_to_sequence=... or call inference_ids()")
        seqs = [list(self.text_to_sequence(t)) for t in self.arrange_text(input_texts)]
        batches, revert = self.pack_sequences(seqs, self.max_len)
        return [[torch.from_numpy(i), torch.from_numpy(l)] for i, l in batches], torch.from_numpy(revert)

    # ---- reference API/utils.py:108-117
    def combine_audio(self, audios, lengths, distance: int) -> np.ndarray:
        output_audio = []
        for i, audio in enumerate(audios):
            audio = np.asarray(audio)[: int(lengths[i]) * self.hop_length]
            audio = audio * self.max_wav_value
            output_audio.extend([audio, np.zeros(distance)])
        return np.concatenate(output_audio).astype("int16")

    def _combine_pcm(self, pcms: List[np.ndarray], lengths: List[int], distance: int) -> np.ndarray:
        """Same result as combine_audio, from the int16 the GPU already produced (trunc(wav * 32768))."""
        total = sum(int(n) * self.hop_length + distance for n in lengths)
        out = np.zeros(total, dtype=np.int16)
        pos = 0
        for pcm, n in zip(pcms, lengths):
            k = int(n) * self.hop_length
            out[pos:pos + k] = pcm[:k]
            pos += k + distance
        return out

    def inference_ids(self, sequences: Sequence[Sequence[int]], speaker_id: str, pitch_control: float = 1.0,
                      energy_control: float = 1.0, duration_control: float = 1.0, silence_distance: float = 0.5) -> np.ndarray:
        """TTS.inference from phoneme-id sequences (the part after text_to_sequence)."""
        batches, revert = self.pack_sequences([list(s) for s in sequences], self.max_len)
        spk = np.array([self.speakers[speaker_id]], dtype=np.int64)  # KeyError for an unknown speaker, as the reference
        pcms, lengths = [], []
        for ids, lens in batches:
            pcm, mel_lens, T = self.engine.synthesize(ids, lens, spk, duration_control, pitch_control, energy_control)
            pcms.extend(list(pcm))
            lengths.extend(int(x) for x in mel_lens)
        pcms = [pcms[i] for i in revert.tolist()]
        lengths = [lengths[i] for i in revert.tolist()]
        return self._combine_pcm(pcms, lengths, int(silence_distance * self.sample_rate))

    def inference(self, texts: list, speaker_id: str, pitch_control: float = 1.0, energy_control: float = 1.0,
                  duration_control: float = 1.0, silence_distance: float = 0.5) -> np.ndarray:
        """reference API/utils.py:119-160 -> 1-D np.int16."""
        if isinstance(texts, str):
            texts = [texts]  # API/inference.py:39-40 passes a bare str, which the reference would iterate per character
        if self.text_to_sequence is None:
            raise RuntimeError("TTS was built without a text front-end: pass text_to_sequence=... or call inference_ids()")
        seqs = [list(self.text_to_sequence(t)) for t in self.arrange_text(texts)]
        generated_audio = self.inference_ids(seqs, speaker_id, pitch_control, energy_control, duration_control, silence_distance)
        print(f"Audio Saved: {time.strftime('%H:%M:%S', time.gmtime(generated_audio.size / self.sample_rate))}")
        return generated_audio


def write_wav(path: str, audio: np.ndarray, samplerate: int) -> None:
    """16-bit mono PCM WAV (what soundfile.write(path, int16 array, sr) produces at API/inference.py:47)."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with wave.open(path, "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(int(samplerate))
        f.writeframes(np.ascontiguousarray(audio, dtype="<i2").tobytes())


def read_wav(path: str):
    """-> (int16 mono samples, sample rate) of a 16-bit PCM WAV (multi-channel files are averaged)."""
    with wave.open(path, "rb") as f:
        if f.getsampwidth() != 2:
            raise ValueError("only 16-bit PCM WAV files are supported")
        sr, ch = f.getframerate(), f.getnchannels()
        a = np.frombuffer(f.readframes(f.getnframes()), dtype="<i2")
    if ch > 1:
        a = a.reshape(-1, ch).astype(np.int32).mean(axis=1).astype(np.int16)
    return a, sr


def time_stretch_wsola(x: np.ndarray, speed: float, sr: int = 22050, frame_ms: float = 40.0, search_ms: float = 10.0) -> np.ndarray:
    """Tempo change without pitch change by waveform-similarity overlap-add (WSOLA, Verhelst & Roelands 1993) -- the family
    of algorithm behind ffmpeg's ``atempo`` filter, which the reference shells out to (API/utils.py:163-172).  ffmpeg is not
    part of this build, so this is a restatement of the published method, not of ffmpeg's code: *parity unpinned*.
    Host-side utility, not on the GPU hot path (see ``Synthesizer.synthesis`` for the tempo control that is)."""
    if not (0.25 <= speed <= 4.0):
        raise ValueError("speed must lie in [0.25, 4]")
    x = np.asarray(x, dtype=np.float64)
    if speed == 1.0 or x.size == 0:
        return x.copy()
    n = max(int(sr * frame_ms / 1000.0) // 2 * 2, 64)      # frame length (even)
    hop_out = n // 2                                       # 50 % overlap of Hann windows sums to one
    hop_in = hop_out * speed
    delta = max(int(sr * search_ms / 1000.0), 1)
    win = np.hanning(n + 1)[:n]
    n_frames = max(int(np.ceil((x.size / speed) / hop_out)), 1)
    xp = np.concatenate([np.zeros(delta + n), x, np.zeros(2 * n + delta + int(hop_in) + 1)])
    out = np.zeros(n_frames * hop_out + n)
    pos = delta + n  # index in xp of the natural continuation of the previous frame
    for i in range(n_frames):
        target = int(round(i * hop_in)) + delta + n
        if i == 0:
            best = target
        else:
            # the candidate within +-delta of `target` that best continues what was just written (template = natural successor)
            tmpl = xp[pos:pos + n]
            lo = target - delta
            seg = xp[lo:lo + n + 2 * delta]
            corr = np.correlate(seg, tmpl, mode="valid")    # 2 delta + 1 lags
            best = lo + int(np.argmax(corr))
        out[i * hop_out:i * hop_out + n] += xp[best:best + n] * win
        pos = best + hop_out
    return out[: int(round(x.size / speed))]


def audio_speed_change(input_path: str, output_path: str = None, speed_rate: float = 1.0, engine: Optional[Engine] = None) -> str:
    """Same signature, output naming and return value as reference API/utils.py:163-172, without the ffmpeg subprocess:
    reads the WAV, time-stretches it and writes ``<input>_<speed>.<ext>``.  With ``engine`` the stretch runs on the GPU
    (``Engine.tempo`` -> ``e2etts_tempo``, the WSOLA kernel); without one, on the host (``time_stretch_wsola``, the same algorithm in
    numpy).  Parity unpinned either way: the reference's ffmpeg ``atempo`` filter is not available here."""
    if output_path is None:
        file_type = input_path.split(".")[-1]
        output_path = f"{input_path[:-len(file_type) - 1]}_{round(speed_rate, 2)}.{file_type}"
    audio, sr = read_wav(input_path)
    if engine is not None:
        pcm = engine.tempo(audio, float(speed_rate), sr)
    else:
        y = time_stretch_wsola(audio.astype(np.float64), float(speed_rate), sr)
        pcm = np.clip(np.rint(y), -32768, 32767).astype(np.int16)
    write_wav(output_path, pcm, sr)
    return output_path


class Synthesizer:
    """reference e2e_tts/src/api/inference.py:12-50."""

    def __init__(self, acoustic_path: str, vocoder_path: str, output_dir: str = "outputs", **tts_kwargs) -> None:
        self.model = TTS(acoustic_path=acoustic_path, vocoder_path=vocoder_path, **tts_kwargs)
        os.makedirs(output_dir, exist_ok=True)
        self.output_dir = output_dir

    def tts_to_file(self, text: str, file_path: str, speed: float = 1):
        return self.synthesis(text, file_path, speed)

    def synthesis(self, text: str, save_filepath: str = None, speed: float = 1, speaker_id: str = "hn_minhphuong", sr: int = 22050,
                  speed_mode: str = "duration"):
        """reference API/inference.py:24-50.  ``speed != 1``: the reference synthesises at normal tempo and then runs ffmpeg's
        ``atempo`` on the file (API/utils.py:163-172).  Here, by default (``speed_mode="duration"``), the tempo goes into the
        model instead -- ``duration_control = 1 / speed`` (U/layers.py:218-221), i.e. the phonemes are simply generated shorter
        or longer on the GPU path, with no vocoded-audio artefacts; ``speed_mode="wsola"`` post-processes the file like the
        reference does (``audio_speed_change``).  Either way the returned path is named ``<file>_<speed>.wav`` as in the reference,
        and -- as in the reference, which writes the file before it runs ffmpeg on it -- ``save_filepath`` itself exists too
        (callers such as the top-level ``synthesizer.Synthesizer`` hand that path on); in duration mode it holds the same
        tempo-adjusted audio."""
        assert len(text) > 0
        if speed_mode not in ("duration", "wsola"):
            raise ValueError("speed_mode must be 'duration' or 'wsola'")
        if not save_filepath:
            save_filepath = os.path.join(self.output_dir, datetime.now().strftime("%m_%d_%Y_%H_%M_%S") + ".wav")
        in_model = speed != 1 and speed_mode == "duration"
        audio = self.model.inference(texts=[text], speaker_id=speaker_id, pitch_control=1.0, energy_control=1.0,
                                     duration_control=(1.0 / float(speed)) if in_model else 1.0, silence_distance=0.5)
        write_wav(save_filepath, audio, sr)
        if in_model:
            file_type = save_filepath.split(".")[-1]
            save_filepath = f"{save_filepath[:-len(file_type) - 1]}_{round(speed, 2)}.{file_type}"
            write_wav(save_filepath, audio, sr)
        if speed != 1 and not in_model:
            save_filepath = audio_speed_change(save_filepath, speed_rate=speed, engine=getattr(self.model, "engine", None))
        return save_filepath
