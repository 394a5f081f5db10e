"""Debug probe: the same 16-bit BigVGAN pass N times; every output must equal the first bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import ims_toucan_prosody_variance_amd  # noqa
from ims_toucan_prosody_variance_amd import engine, fixture_weights as fw
from ims_toucan_prosody_variance_amd.ragged import Ragged

dev = torch.device("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
lens = [565, 433, 279, 87]
voc = engine.VocoderEngine(fw.bigvgan_state_dict(), "bigvgan", dev, precision=prec)
rag = Ragged(lens, dev, align=2)
torch.manual_seed(0)
mel = torch.randn(rag.total_rows, 80, device=dev)
taps0 = {}
w0, rw = voc.forward(mel, rag, taps=taps0)
torch.cuda.synchronize()
w0 = w0.clone(); taps0 = {k: v.clone() for k, v in taps0.items()}
bad = 0
for i in range(N):
    taps = {}
    w, _ = voc.forward(mel, rag, taps=taps)
    torch.cuda.synchronize()
    if not all(torch.equal(w[b:b + n], w0[b:b + n]) for b, n in zip(rw.begins, rw.lengths)):
        bad += 1
        first = [k for k in taps if not torch.equal(torch.nan_to_num(taps[k]), torch.nan_to_num(taps0[k]))]
        d = (w - w0)
        print(f"iteration {i}: differs; first differing tap {first[:1]}, max |d| {float(torch.nan_to_num(d).abs().max()):.3e}, nan {int(torch.isnan(w).sum())}", flush=True)
print(prec, "mismatching passes:", bad, "of", N)
