"""Vietnamese text front-end: syllable -> phonemes (C1 w V_T C2 scheme) -> symbol ids.

Host-side (CPU, strings) counterpart of the reference's ``g2p`` package -- SURVEY.md 8(f) item 1:
``vi_convert`` / ``normalize_phonemes`` (reference e2e_tts/models/g2p/g2p.py:58-176), the 131-symbol table
(symbols.py:19-50) and ``text_to_sequence`` (__init__.py:11-31).  Written from the behaviour of those functions and
pinned by tests/golden/g2p.npz (1 500 dictionary syllables + sentences converted by the reference's own code).

Deliberate differences: no dependency on ``unidecode`` / ``g2p_en`` (absent here): ASCII folding is NFD minus
combining marks, which equals unidecode on the Vietnamese alphabet; the English fallback is not implemented (the
shipped foreign_words.json is empty); ``text_to_sequence`` works (the reference's cannot run as shipped: its cleaner
shadows the function it wants to call, cleaners.py:12,26-30); a lone "-" token is skipped instead of raising IndexError.
"""
from __future__ import annotations

import re
import string
import unicodedata
from typing import Dict, List, Optional, Sequence, Tuple, Union

# ---- symbol table (reference symbols.py:19-50): 4 specials + 23 onsets + 2 medials + 15 x 6 toned nuclei + 12 codas
_ONSETS = ["b", "ch", "d", "dd", "g", "h", "k", "kh", "kw", "l", "m", "n", "ng", "nh", "p", "ph", "r", "s", "t", "th", "tr", "v", "x"]
_MEDIALS = ["wo", "wu"]
_NUCLEI = ["a", "aa", "aw", "e", "ee", "i", "o", "oa", "oo", "ow", "u", "uw", "ie", "uo", "wa"]
_CODAS = ["cz", "iz", "kz", "mz", "ngz", "nhz", "nz", "oz", "pz", "tz", "uz", "yz"]
_TONES = "012345"
symbols: List[str] = [s.upper() for s in (["<pad>", "<silent>", "<s>", "</s>"] + _ONSETS + _MEDIALS +
                                          [f"{v}_{t}" for v in _NUCLEI for t in _TONES] + _CODAS)]
_symbol_to_id: Dict[str, int] = {s: i for i, s in enumerate(symbols)}
_id_to_symbol: Dict[int, str] = {i: s for i, s in enumerate(symbols)}

# ---- orthography -> phoneme tables (reference g2p.py:17-29)
_ONSET_MAP = {"b": "b", "ch": "ch", "đ": "dd", "ph": "ph", "h": "h", "d": "d", "k": "k", "qu": "kw", "q": "k", "c": "k", "l": "l",
              "m": "m", "n": "n", "nh": "nh", "ng": "ng", "ngh": "ng", "p": "p", "x": "x", "s": "s", "t": "t", "th": "th",
              "tr": "tr", "v": "v", "kh": "kh", "g": "g", "gh": "g", "gi": "d", "r": "r"}
_MEDIAL_MAP = {"u": "wu", "o": "wo"}
_MONOPHTHONG = {"ă": "aw", "ê": "ee", "e": "e", "â": "aa", "ơ": "ow", "y": "i", "i": "i", "ư": "uw", "ô": "oo", "u": "u",
                "oo": "o", "o": "oa", "a": "a"}
_DIPHTHONG = {"yê": "ie", "iê": "ie", "ya": "ie", "ia": "ie", "ươ": "wa", "ưa": "wa", "uô": "uo", "ua": "uo"}
_CODA_MAP = {"m": "mz", "n": "nz", "ng": "ngz", "nh": "nhz", "p": "pz", "t": "tz", "ch": "kz", "k": "cz", "c": "cz", "u": "uz",
             "o": "oz", "y": "yz", "i": "iz"}
_VOWEL_LETTERS = set("aeiouy")

# tone marks: combining character -> tone number (sắc 1, huyền 2, hỏi 3, ngã 4, nặng 5)
_TONE_MARK = {"́": "1", "̀": "2", "̉": "3", "̃": "4", "̣": "5"}


def _fold(s: str) -> str:
    """ASCII folding of Vietnamese letters (what the reference gets from unidecode)."""
    s = s.replace("đ", "d").replace("Đ", "D")
    return "".join(c for c in unicodedata.normalize("NFD", s) if not unicodedata.combining(c))


def _strip_tone(ch: str) -> Tuple[str, Optional[str]]:
    """(letter without its tone mark, tone or None); the quality marks (breve, circumflex, horn) stay."""
    parts = unicodedata.normalize("NFD", ch)
    tone = None
    kept = []
    for c in parts:
        if c in _TONE_MARK and tone is None:
            tone = _TONE_MARK[c]
        else:
            kept.append(c)
    return unicodedata.normalize("NFC", "".join(kept)), tone


def vi_convert(graph: str) -> List[str]:
    """One lower-case Vietnamese syllable -> [onset, medial, nucleus_tone, coda] (empty parts dropped)."""
    if len(graph) == 1 and graph in _ONSET_MAP:
        return [_ONSET_MAP[graph]]
    # tone = the first tone-marked letter (g2p.py:72-78)
    tone = "0"
    letters = list(unicodedata.normalize("NFC", graph))
    for i, ch in enumerate(letters):
        base, t = _strip_tone(ch)
        if t is not None and _fold(ch) in _VOWEL_LETTERS:
            tone, letters[i] = t, base
            break
    word = "".join(letters)
    # alternate consonant / vowel runs (g2p.py:81-93)
    runs: List[str] = [word[0]]
    for prev, ch in zip(word, word[1:]):
        if (_fold(ch) in _VOWEL_LETTERS) != (_fold(prev) in _VOWEL_LETTERS):
            runs.append(ch)
        else:
            runs[-1] += ch
    if _fold(runs[0][0]) in _VOWEL_LETTERS:
        runs.insert(0, "")
    runs += [""] * (3 - len(runs))
    onset, nucleus, coda = runs[0], runs[1], runs[2]
    folded = [_fold(r) for r in runs]
    if nucleus:
        # "gi" and "qu" share a letter between onset and nucleus (g2p.py:99-104)
        if folded[0] == "g" and folded[1][0] == "i":
            onset = "d"
            if not (folded[1] in ("i", "ieu") or (nucleus == "iê" and coda)):
                nucleus = nucleus[1:]
        elif folded[0] == "q" and folded[1][0] == "u":
            onset = "qu" if nucleus != "u" else "c"
            if folded[1] != "u":
                nucleus = nucleus[1:]
        if len(nucleus) > 1:
            # a trailing glide becomes the coda; a leading u / o becomes the medial (g2p.py:106-111)
            if nucleus[-1] in "uoiy" and nucleus not in _DIPHTHONG and not coda:
                coda, nucleus = nucleus[-1], nucleus[:-1]
            if nucleus[0] in "uo" and nucleus not in _DIPHTHONG and nucleus != "oo":
                nucleus = nucleus[0] + " " + nucleus[1:]
    c1 = _ONSET_MAP.get(onset, "")
    medial = vowel = ""
    if nucleus:
        parts = nucleus.split()
        if len(parts) == 1 and parts[-1] == "o" and coda in ("n", "t", "i"):
            parts[-1] = "oo"
        if len(parts) == 2:
            medial = _MEDIAL_MAP[parts[0]]
        last = parts[-1]
        vowel = _DIPHTHONG[last] if len(last) == 2 and last != "oo" else _MONOPHTHONG[last]
    c2 = _CODA_MAP.get(coda, "")
    return [x for x in (c1, medial, f"{vowel}_{tone}", c2) if x]


def normalize_phonemes(text: Union[str, Sequence[str]], foreign_dict: Optional[dict] = None, is_training: bool = True):
    """Sentence (string or token list) -> (upper-case phoneme list, boundaries) (g2p.py:135-176)."""
    tokens = text.split() if isinstance(text, str) else list(text)
    if not tokens or tokens[-1] not in string.punctuation or len(tokens[-1]) != 1:
        tokens.append(".")
    groups: List[list] = []
    for i, word in enumerate(tokens):
        entry = foreign_dict.get(word) if foreign_dict else None
        if entry is not None:
            if entry.get("phonemes") is not None:
                def arpabet(seq):
                    return [f"@{ph[:-1] if ph[-1].isdigit() else ph}" for ph in seq.split()]
                ph = entry["phonemes"]
                groups.append([arpabet(x.strip()) for x in ph.split("|")] if "|" in ph else arpabet(ph))
            else:
                groups.append([vi_convert(x) for x in entry["subtitle"].split("-")])
        elif "-" in word:
            parts = [vi_convert(x) for x in word.split("-") if x]
            if parts:
                groups.append(parts)
        elif len(word) == 1 and word in string.punctuation:
            groups.append(["</s>"] if i == len(tokens) - 1 else ["<silent>"])
        else:
            groups.append(vi_convert(word))
    phonemes: List[str] = []
    boundaries: list = []
    for grp in groups:
        if grp and isinstance(grp[0], list):
            phonemes.extend(ph for w in grp for ph in w)
            if is_training:
                boundaries.extend(len(w) for w in grp)
            else:
                boundaries.append([len(w) for w in grp])
        else:
            phonemes.extend(grp)
            boundaries.append(len(grp))
    return [x.upper() for x in phonemes], boundaries


_WS = re.compile(r"\s+")


def text_to_sequence(text: str, cleaner_names=("normalize_phonemes",), return_boundary: bool = False):
    """String -> list of symbol ids; KeyError for a phoneme outside the table, as the reference (__init__.py:26)."""
    cleaned = _WS.sub(" ", text.lower()).strip()
    phonemes, boundaries = normalize_phonemes(cleaned, is_training=False)
    seq = [_symbol_to_id[w[:-1] if w.startswith("@") and w[-1].isdigit() else w] for w in phonemes]
    return (seq, boundaries) if return_boundary else seq


def sequence_to_text(sequence: Sequence[int]) -> str:
    return " ".join(_id_to_symbol[int(i)] for i in sequence)
