"""Glue between reference-style inputs (config dict, stats, state dicts) and the HIP engine."""
from __future__ import annotations

from typing import Mapping, Optional

from . import packer
from ._lib import Engine
from .config import N_SYMBOLS, EngineDims, dims_from_config


def engine_from_states(config: dict, stats: dict, acoustic_state: Mapping[str, object], vocoder_state: Mapping[str, object],
                       device: int = 0, n_symbols: int = N_SYMBOLS, pos_table_rows: int = 4096,
                       dims: Optional[EngineDims] = None) -> Engine:
    """Build an engine on `device` and load the packed weights (what TTS.__init__ does in the reference,
    API/utils.py:41-56)."""
    n_speakers = int(acoustic_state["speaker_emb.weight"].shape[0])
    dims = dims or dims_from_config(config, stats, n_speakers, n_symbols, pos_table_rows)
    eng = Engine(dims, device)
    eng.load_weights(packer.pack(dims, acoustic_state, vocoder_state))
    return eng
