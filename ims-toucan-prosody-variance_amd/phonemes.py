"""Phoneme-string -> articulatory feature tensor (the ``input_is_phones=True`` branch).

Behavioural mirror of Preprocessing/TextFrontend.py:213-288 (``string_to_tensor`` with
``input_phonemes=True``): every IPA symbol is looked up in the 62-dim articulatory table,
modifier symbols flip a bit on the neighbouring phoneme.  The table itself is data captured
from ``Preprocessing/articulatory_features.generate_feature_table()`` (:904-949) by
``tests/golden/make_golden.py`` and shipped as ``data/phone_table.json``.

Grapheme-to-phoneme conversion (espeak-ng via phonemizer) is outside the hot path and is
not available offline; ``string_to_tensor(text, input_phonemes=False)`` raises.
"""
import json
import os

import numpy as np

N_FEATS = 62
IDX = dict(stressed=0, very_high_tone=1, high_tone=2, mid_tone=3, low_tone=4, very_low_tone=5, rising_tone=6,
           falling_tone=7, peaking_tone=8, dipping_tone=9, lengthened=10, half_length=11, shortened=12,
           consonant=13, vowel=14, phoneme=15, silence=16, end_of_sentence=17, questionmark=18,
           exclamationmark=19, fullstop=20, word_boundary=21, nasal=51, unvoiced=60, voiced=61)

# symbol -> feature bit set on the PREVIOUS phoneme (TextFrontend.py:236-277)
_POST_MODIFIERS = {
    "ː": IDX["lengthened"], "ˑ": IDX["half_length"], "̆": IDX["shortened"], "̃": IDX["nasal"],
    "˥": IDX["very_high_tone"], "˦": IDX["high_tone"], "˧": IDX["mid_tone"], "˨": IDX["low_tone"],
    "˩": IDX["very_low_tone"], "⭧": IDX["rising_tone"], "⭨": IDX["falling_tone"], "⮁": IDX["peaking_tone"],
    "⮃": IDX["dipping_tone"],
}
_STRESS = "ˈ"

_TABLE = None


def phone_table():
    """dict symbol -> np.float32[62]."""
    global _TABLE
    if _TABLE is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "phone_table.json")
        with open(path, encoding="utf-8") as f:
            raw = json.load(f)
        _TABLE = {k: np.array([int(c) for c in v], dtype=np.float32) for k, v in raw.items()}
    return _TABLE


def phones_to_features(phones: str, handle_missing: bool = True) -> np.ndarray:
    """[L, 62] float32 feature matrix for a phoneme string."""
    table = phone_table()
    phones = phones.replace("ɚ", "ə").replace("ᵻ", "ɨ")  # TextFrontend.py:223
    rows = []
    stressed = False
    for ch in phones:
        if ch == _STRESS:
            stressed = True
        elif ch in _POST_MODIFIERS:
            rows[-1][_POST_MODIFIERS[ch]] = 1.0
        else:
            vec = table.get(ch)
            if vec is None:
                if not handle_missing:
                    raise KeyError(ch)
                print("unknown phoneme: {}".format(ch))
            else:
                rows.append(vec.copy())
            # the reference consumes a pending stress mark here even when the symbol was unknown, so the stress then lands
            # on the PREVIOUS phoneme (TextFrontend.py:275-286; an IndexError if there is none, as there)
            if stressed:
                stressed = False
                rows[-1][IDX["stressed"]] = 1.0
    if not rows:
        return np.zeros((0, N_FEATS), dtype=np.float32)
    return np.stack(rows)


# Preprocessing/TextFrontend.py:490-524
LANGUAGE_IDS = {"de": 1, "el": 2, "es": 3, "fi": 4, "ru": 5, "hu": 6, "nl": 7, "fr": 8, "pt": 9, "pl": 10, "it": 11,
                "en": 12, "cmn": 13, "vi": 14, "uk": 15, "fa": 16, "pt-br": 17}


def get_language_id(language: str):
    """Language shorthand -> id (None for an unknown language, as the reference's if/elif chain)."""
    return LANGUAGE_IDS.get(language)


class ArticulatoryCombinedTextFrontend:
    """Minimal counterpart of Preprocessing/TextFrontend.ArticulatoryCombinedTextFrontend."""

    def __init__(self, language="en", add_silence_to_end=True):
        self.language = language
        self.add_silence_to_end = add_silence_to_end

    def string_to_tensor(self, text, view=False, device="cpu", handle_missing=True, input_phonemes=False):
        import torch
        if not input_phonemes:
            raise RuntimeError("grapheme-to-phoneme conversion needs espeak-ng/phonemizer, which are not part of the "
                               "MI355X hot path; pass phoneme strings with input_is_phones=True")
        if view:
            print("Phonemes: \n{}\n".format(text))
        return torch.from_numpy(phones_to_features(text, handle_missing))

    def get_phone_string(self, text, include_eos_symbol=True, for_feature_extraction=False, for_plot_labels=False):
        raise RuntimeError("grapheme-to-phoneme conversion is not available (no espeak-ng offline)")
