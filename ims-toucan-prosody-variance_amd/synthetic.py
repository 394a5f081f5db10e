"""Seeded synthetic inputs of the benchmark / parity shapes.

Utterance ``u`` of a batch: position 0 = '~', positions L-2, L-1 = '~', '#'; every 8th position is a
word boundary ' '; the remaining positions are drawn uniformly from the phoneme-type rows of the
articulatory table with the counter RNG of ``fixture_weights`` (seed 1000+u); the ``stressed`` bit is
set with p = 0.15 on vowels.  utt_emb ~ N(0,1)[64] (seed 2000+u); PostFlow noise
z = 0.8 * N(0,1)[80, T] (seed 3000+u), matching Glow.py:363's temperature.
"""
import numpy as np

from . import fixture_weights as fw
from .phonemes import IDX, phone_table

LANG_EN = 12


def _phoneme_rows():
    table = phone_table()
    syms = sorted(s for s, v in table.items() if v[IDX["phoneme"]] == 1.0)
    return syms, table


def utterance_features(u: int, L: int, word_boundaries: bool = True) -> np.ndarray:
    syms, table = _phoneme_rows()
    pick = fw.uniform01(f"utt{u}.sym", L, 1000 + u)
    stress = fw.uniform01(f"utt{u}.stress", L, 1000 + u, stream=5)
    rows = np.zeros((L, 62), dtype=np.float32)
    for i in range(L):
        if i == 0 or (i == L - 2 and L >= 4):
            rows[i] = table["~"]
        elif i == L - 1 and L >= 4:
            rows[i] = table["#"]
        elif word_boundaries and i % 8 == 0:
            rows[i] = table[" "]
        else:
            rows[i] = table[syms[int(pick[i] * len(syms)) % len(syms)]]
            if rows[i, IDX["vowel"]] == 1.0 and stress[i] < 0.15:
                rows[i, IDX["stressed"]] = 1.0
    return rows


def utterance_embedding(u: int) -> np.ndarray:
    return fw.normal(f"utt{u}.emb", (64,), 2000 + u)


def postflow_noise(u: int, T: int) -> np.ndarray:
    """[80, T] = 0.8 * N(0,1); column t is drawn independently of T so prefixes agree."""
    return (fw.normal(f"utt{u}.z", (T, 80), 3000 + u) * np.float32(0.8)).T.copy()


def gold_durations(feats: np.ndarray, frames_per_phone: int = 5) -> np.ndarray:
    d = np.full((feats.shape[0],), frames_per_phone, dtype=np.int64)
    d[feats[:, IDX["word_boundary"]] == 1.0] = 0
    return d


def ragged_durations(u: int, feats: np.ndarray, lo: int = 2, hi: int = 8) -> np.ndarray:
    r = fw.uniform01(f"utt{u}.dur", feats.shape[0], 4000 + u)
    d = (lo + np.floor(r * (hi - lo + 1))).astype(np.int64)
    d[feats[:, IDX["word_boundary"]] == 1.0] = 0
    return d
