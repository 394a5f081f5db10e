"""Packed ragged batches: utterances are concatenated along the row (time) axis.

``Ragged`` records where each utterance lives and builds the tile tables (``TtsTile[]``,
include/toucan_tts.h) the kernels use to stay inside one utterance.  This is what turns the
reference's strictly batch-1 inference (InferenceToucanTTS.py:293-316, ToucanTTSInterface.py:269-280)
into a batch: every kernel sees per-utterance begin/end rows, so zero padding, GroupNorm statistics,
attention keys and the Glow squeeze behave exactly as if each utterance ran alone.
"""
import numpy as np
import torch


class Ragged:
    _cache = {}

    @classmethod
    def cached(cls, lengths, device, align=1):
        """Layouts (and their device-side tile tables) are reused across calls with the same shape signature."""
        key = (tuple(int(n) for n in lengths), str(device), int(align))
        hit = cls._cache.get(key)
        if hit is None:
            if len(cls._cache) > 64:
                cls._cache.clear()
            hit = cls._cache[key] = cls(lengths, device, align)
        return hit

    def __init__(self, lengths, device, align=1, begins=None):
        self.lengths = [int(n) for n in lengths]
        self.device = device
        if begins is None:
            begins, off = [], 0
            for n in self.lengths:
                begins.append(off)
                off += (n + align - 1) // align * align
            self.total_rows = off
        else:
            begins = [int(b) for b in begins]
            self.total_rows = max([b + n for b, n in zip(begins, self.lengths)] + [0])
        self.begins = begins
        self.n_seq = len(self.lengths)
        self.max_len = max(self.lengths) if self.lengths else 0
        self._tiles = {}
        self._bounds = None
        self._derived = {}

    def bounds(self):
        """(seq_begin, seq_end) int32 device tensors."""
        if self._bounds is None:
            b = torch.tensor(self.begins, dtype=torch.int32)
            e = torch.tensor([x + n for x, n in zip(self.begins, self.lengths)], dtype=torch.int32)
            self._bounds = (b.to(self.device), e.to(self.device))
        return self._bounds

    def tiles(self, rows_per_tile):
        """(device int32 tensor [n,4] = (row0, seq_begin, seq_end, seq_id), n)."""
        key = int(rows_per_tile)
        if key not in self._tiles:
            b = np.asarray(self.begins, dtype=np.int64)
            n = np.asarray(self.lengths, dtype=np.int64)
            per = (n + key - 1) // key  # tiles per utterance
            sid = np.repeat(np.arange(len(n)), per)
            first = np.cumsum(per) - per
            k = np.arange(int(per.sum())) - np.repeat(first, per)  # tile index inside its utterance
            arr = np.stack([b[sid] + k * key, b[sid], b[sid] + n[sid], sid], axis=1).astype(np.int32).reshape(-1, 4)
            assert self.total_rows < 2 ** 31
            self._tiles[key] = (torch.from_numpy(np.ascontiguousarray(arr)).to(self.device), arr.shape[0])
        return self._tiles[key]

    def scaled(self, factor):
        """Layout after a transposed conv of stride ``factor``: [rows, f*C] viewed as [rows*f, C]."""
        key = ("scaled", factor)
        if key not in self._derived:
            self._derived[key] = Ragged([n * factor for n in self.lengths], self.device, begins=[b * factor for b in self.begins])
        return self._derived[key]

    def halved(self):
        """Glow squeeze (glow_utils.py:28-40): pairs of frames become one row; an odd last frame is dropped.
        Requires even begins (construct the frame layout with align=2)."""
        if "halved" not in self._derived:
            assert all(b % 2 == 0 for b in self.begins)
            self._derived["halved"] = Ragged([n // 2 for n in self.lengths], self.device, begins=[b // 2 for b in self.begins])
        return self._derived["halved"]

    def doubled(self):
        """Inverse re-view of ``halved`` (glow_utils.py:43-53): every squeezed row is two frames again."""
        if "doubled" not in self._derived:
            self._derived["doubled"] = Ragged([2 * n for n in self.lengths], self.device, begins=[2 * b for b in self.begins])
        return self._derived["doubled"]
