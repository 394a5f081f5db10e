"""Debug probe: every Ops.empty() buffer is filled with NaN, so any kernel that mixes never-written memory into a result shows up."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import ims_toucan_prosody_variance_amd  # noqa
from ims_toucan_prosody_variance_amd import engine, fixture_weights as fw, synthetic as syn
from ims_toucan_prosody_variance_amd.ragged import Ragged

dev = torch.device("cuda:0")
orig = engine.Ops.empty
def nan_empty(self, *shape, dtype=torch.float32):
    t = orig(self, *shape, dtype=dtype)
    if t.is_floating_point():
        t.fill_(float("nan"))
    return t
engine.Ops.empty = nan_empty
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
lens = [565, 433, 279, 87]
for kind, sd in (("bigvgan", fw.bigvgan_state_dict()), ("hifigan", fw.hifigan_state_dict())):
    voc = engine.VocoderEngine(sd, kind, dev, precision=prec)
    rag = Ragged(lens, dev, align=2)
    mel = torch.full((rag.total_rows, 80), float("nan"), device=dev)
    for b, n in zip(rag.begins, rag.lengths):
        mel[b:b + n] = torch.randn(n, 80, device=dev)
    taps = {}
    wav, rw = voc.forward(mel, rag, taps=taps) if "taps" in voc.forward.__code__.co_varnames else voc.forward(mel, rag)
    torch.cuda.synchronize()
    for k, v in taps.items():
        print(kind, k, "finite" if bool(torch.isfinite(v).all()) else "HAS NaN (may be padding rows)")
    for u, (b, n) in enumerate(zip(rw.begins, rw.lengths)):
        w = wav[b:b + n]
        print(kind, prec, "utt", u, "finite" if bool(torch.isfinite(w).all()) else f"NaN count {int((~torch.isfinite(w)).sum())} of {n}")
