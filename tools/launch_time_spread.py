"""Spread of the launch time of tts_resblock_step over 200 back-to-back launches per shape (run on the MI355X box): median, p99,
maximum, outliers - the persistent work-queue kernel must not have any (TOUCAN_RB_WG_PER_CU=-1: one workgroup per tile)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ims_toucan_prosody_variance_amd
from ims_toucan_prosody_variance_amd import capi, engine, packing
from ims_toucan_prosody_variance_amd.ragged import Ragged
dev = torch.device("cuda:0"); ops = engine.Ops(dev)
filt = torch.from_numpy(packing.kaiser_sinc_filter12()).to(dev)
for C, mult, k, dil in ((64, 192, 11, 5), (32, 384, 3, 1), (128, 48, 7, 3)):
    rows = 640 * mult; rag = Ragged([rows] * 32, dev); R = rag.total_rows
    x = torch.randn(R, C, device=dev).to(torch.bfloat16); y = torch.empty_like(x)
    sn = (torch.zeros(C, device=dev), torch.zeros(C, device=dev))
    rs = np.random.RandomState(0)
    c1 = packing.pack_conv((rs.randn(C, C, k) / np.sqrt(C * k)).astype(np.float32), np.zeros(C, np.float32), dev, dil=dil, bf16=True)
    c2 = packing.pack_conv((rs.randn(C, C, k) / np.sqrt(C * k)).astype(np.float32), np.zeros(C, np.float32), dev, dil=1, bf16=True)
    run = lambda: ops.resblock_step(c1, c2, x, y, rag, capi.PRE_SNAKE, 0.1, sn, sn, filt)
    run(); torch.cuda.synchronize()
    ts = []
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
    for e0, e1 in evs:
        e0.record(); run(); e1.record()
    torch.cuda.synchronize()
    ts = np.array([e0.elapsed_time(e1) * 1e3 for e0, e1 in evs])
    print(f"C={C} k={k}: median {np.median(ts):.0f} us, p99 {np.percentile(ts, 99):.0f}, max {ts.max():.0f}, >2x median: {(ts > 2 * np.median(ts)).sum()}", flush=True)
