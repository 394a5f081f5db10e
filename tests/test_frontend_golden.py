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


def test_file_reader_harness_phoneme_lines_are_fully_covered_by_the_table(capsys):
    """run_phoneme_file_reader.py (counterpart of the reference's run_text_to_file_reader.py:8-41) carries the fourteen lines of the poem as phoneme strings:
    every symbol must be one the articulatory table knows (no "unknown phoneme" diagnostics), one string per text line."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("reader_harness", os.path.join(root, "run_phoneme_file_reader.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert len(mod.THE_RAVEN) == len(mod.THE_RAVEN_PHONES) == 14
    for line in mod.THE_RAVEN_PHONES:
        feats = phonemes.phones_to_features(line)
        assert capsys.readouterr().out == "", line
        assert 20 < feats.shape[0] < 128 and line.startswith("~") and line.endswith("#")
        assert feats[-1, phonemes.IDX["end_of_sentence"]] == 1.0
