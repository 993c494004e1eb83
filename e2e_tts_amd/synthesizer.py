"""Drop-in mirror of the reference's top-level façade (reference synthesizer.py:10-65).

``Synthesizer(output_dir).synthesis(text, language, target_filepath, speed) -> (tts_wav_path, vc_wav_path | None)``.
Vietnamese ("vie") is served natively by the HIP engine.  English / Burmese / voice conversion are delegated by
the reference to the third-party Coqui ``TTS`` package, fetched by model NAME from the network
(synthesizer.py:12-13,27); they are constructed lazily here and raise a clear error when that package or the
network is unavailable -- they are outside the hot path this project rebuilds.
"""
from __future__ import annotations

import os
from datetime import datetime
from typing import Optional, Tuple

from .api import Synthesizer as SynthesizerVN


def gen_filename() -> str:
    return datetime.now().strftime("%Y_%m_%d-%I_%M_%S_%p") + ".wav"


class _LazyCoqui:
    def __init__(self, model_name: str):
        self.model_name = model_name
        self._model = None

    def get(self):
        if self._model is None:
            try:
                from TTS.api import TTS  # Coqui
            except ImportError as e:
                raise RuntimeError(f"language backed by Coqui TTS model {self.model_name!r} needs the `TTS` package "
                                   "(not part of this build; only 'vie' runs on the HIP engine)") from e
            self._model = TTS(model_name=self.model_name)
        return self._model

    def tts_to_file(self, text, file_path, speed=1.0):
        return self.get().tts_to_file(text, file_path=file_path, speed=speed)


class Synthesizer:
    def __init__(self, output_dir: str = "outputs", acoustic_path: str = "e2e_tts/exps/acoustic/statedict.pt",
                 vocoder_path: str = "e2e_tts/exps/vocoder/statedict.pt", use_returned_path: bool = False, **tts_kwargs) -> None:
        # use_returned_path: see synthesis().  False (default) = the reference's behaviour, path for path.
        self.use_returned_path = bool(use_returned_path)
        vie_model = SynthesizerVN(acoustic_path=acoustic_path, vocoder_path=vocoder_path, output_dir=output_dir, **tts_kwargs)
        self.model_dict = {
            "eng": _LazyCoqui("tts_models/en/ljspeech/vits"),
            "mya": _LazyCoqui("tts_models/mya/fairseq/vits"),
            "vie": vie_model,
        }
        self._vc = _LazyCoqui("tts_models/en/ljspeech/vits")
        os.makedirs(output_dir, exist_ok=True)
        self.output_dir = output_dir

    @property
    def voice_conversion_model(self):
        m = self._vc.get()
        if not getattr(m, "_e2e_vc_loaded", False):
            m.load_vc_model_by_name("voice_conversion_models/multilingual/vctk/freevc24")
            m._e2e_vc_loaded = True
        return m

    def synthesis(self, text: str, language: str, target_filepath: Optional[str] = None, speed: float = 1.0) -> Tuple[str, Optional[str]]:
        if not isinstance(speed, float):
            speed = float(speed)
        language = language.split()[0]  # "<code> <Name>" (synthesizer.py:43)
        tts_output_filepath = os.path.join(self.output_dir, gen_filename())
        # The reference ignores what tts_to_file returns (synthesizer.py:47) and hands on the path it passed in -- for speed != 1 that is
        # NOT the file named <file>_<speed>.wav.  Kept as is by default (a drop-in returns what the reference returns; with the service's
        # default speed_mode="duration" that file already carries the tempo, api.Synthesizer.synthesis).  use_returned_path=True opts
        # into the path tts_to_file reports instead ("Behaviour the drop-in keeps").
        made = self.model_dict[language].tts_to_file(text, file_path=tts_output_filepath, speed=speed)
        if getattr(self, "use_returned_path", False) and isinstance(made, str) and os.path.exists(made):
            tts_output_filepath = made
        vc_output_filepath = None
        if target_filepath:
            vc_output_filepath = os.path.join(self.output_dir, gen_filename())
            self.voice_conversion_model.voice_conversion_to_file(source_wav=tts_output_filepath, target_wav=target_filepath,
                                                                file_path=vc_output_filepath)
        return tts_output_filepath, vc_output_filepath

    def voice_conversion(self, src_filepath: str, target_filepath: str) -> str:
        save_filepath = os.path.join(self.output_dir, gen_filename())
        self.voice_conversion_model.voice_conversion_to_file(source_wav=src_filepath, target_wav=target_filepath, file_path=save_filepath)
        return save_filepath
