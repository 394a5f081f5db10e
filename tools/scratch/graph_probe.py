import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import ims_toucan_prosody_variance_amd  # noqa
from ims_toucan_prosody_variance_amd import engine, fixture_weights as fw
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from tests.test_gpu_e2e import _gold, _inputs, DEV
gs, gl = _gold("L20_pred"), _gold("L128_gold5")
eager = engine.AcousticEngine(fw.acoustic_state_dict(), DEV)
graphed = engine.AcousticEngine(fw.acoustic_state_dict(), DEV, use_graphs=True)
def run(eng, g, tag):
    sys.stderr.write(f"=== {tag}\n"); sys.stderr.flush()
    texts, embs, langs, zs = _inputs([g])
    kw = {"durations": [torch.from_numpy(g["gold_durations"])]} if "gold_durations" in g.files else {}
    out = eng.forward(texts, embs, langs, z_noise=zs, **kw)
    torch.cuda.synchronize()
    return out["mel"][0].clone(), out
ref_s, o_ref = run(eager, gs, "eager short")
ref_l, _ = run(eager, gl, "eager long")
a, _ = run(graphed, gs, "graph short 1"); print("1", torch.equal(a, ref_s))
b, _ = run(graphed, gl, "graph long"); print("2", torch.equal(b, ref_l))
c, o_c = run(graphed, gs, "graph short 2"); print("3", torch.equal(c, ref_s), float((c - ref_s).abs().max()))
for k in ("pitch", "energy"):
    print(k, torch.equal(o_c[k][0], o_ref[k][0]))
print("durations", torch.equal(o_c["durations_packed"], o_ref["durations_packed"]))
