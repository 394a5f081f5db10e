"""Phoneme-string front end vs golden vectors captured from the reference's own ``string_to_tensor(input_phonemes=True)`` and
``get_language_id`` (tests/golden/make_frontend_golden.py; Preprocessing/TextFrontend.py:213-288, 490-524).  Bit-exact."""
import json
import os

import numpy as np
import pytest

from ims_toucan_prosody_variance_amd import phonemes

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frontend.json"), encoding="utf-8") as f:
    GOLD = json.load(f)


@pytest.mark.parametrize("case", GOLD["cases"], ids=[str(i) for i in range(len(GOLD["cases"]))])
def test_phones_to_features_matches_reference(case, capsys):
    feats = phonemes.phones_to_features(case["phones"])
    want = np.array([[int(ch) for ch in row] for row in case["rows"]], dtype=np.float32).reshape(len(case["rows"]), phonemes.N_FEATS)
    assert feats.shape == want.shape
    assert np.array_equal(feats, want)
    assert capsys.readouterr().out == case["printed"]  # same "unknown phoneme" diagnostics, in the same order


def test_frontend_class_returns_the_same_tensor():
    tf = phonemes.ArticulatoryCombinedTextFrontend(language="en")
    case = GOLD["cases"][0]
    t = tf.string_to_tensor(case["phones"], input_phonemes=True)
    assert tuple(t.shape) == (len(case["rows"]), 62)
    assert ["".join(str(int(v)) for v in r) for r in t.tolist()] == case["rows"]
    with pytest.raises(KeyError):
        tf.string_to_tensor("a§", input_phonemes=True, handle_missing=False)


def test_language_ids_match_reference():
    for lang, want in GOLD["language_ids"].items():
        assert phonemes.get_language_id(lang) == want
