"""Does the fused residual step's time depend on where its output lies relative to its input (HBM channel / bank aliasing)?
x at the start of one big allocation, y `gap + skew` bytes behind the end of x, for a list of skews; C = 32 (k = 3) and C = 64 (k = 11),
batch 32.  us per launch per skew, three passes over the list (run-to-run spread)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import ims_toucan_prosody_variance_amd  # noqa: F401
from ims_toucan_prosody_variance_amd import capi, engine, packing
from ims_toucan_prosody_variance_amd.ragged import Ragged

dev = torch.device("cuda:0")
ops = engine.Ops(dev)
filt = torch.from_numpy(packing.kaiser_sinc_filter12()).to(dev)
skews = [0, 256, 1024, 4096, 8192, 16384, 65536, 262144, 1 << 20, (1 << 20) + 4096, 3 << 20]
for C, mult, k, dil in ((32, 384, 3, 1), (64, 192, 11, 5), (128, 48, 7, 3)):
    rag = Ragged([640 * mult] * 32, dev)
    R = rag.total_rows
    nbytes = R * C * 2
    big = torch.empty(2 * nbytes + (8 << 20), dtype=torch.uint8, device=dev)
    base = big.data_ptr()
    pad = (-base) % (1 << 21)  # x starts on a 2 MiB boundary
    x = big[pad:pad + nbytes].view(torch.bfloat16).view(R, C)
    x.copy_(torch.randn(R, C, device=dev).to(torch.bfloat16))
    rs = np.random.RandomState(0)
    c1 = packing.pack_conv((rs.randn(C, C, k) / np.sqrt(C * k)).astype(np.float32), np.zeros(C, np.float32), dev, dil=dil, bf16=True)
    c2 = packing.pack_conv((rs.randn(C, C, k) / np.sqrt(C * k)).astype(np.float32), np.zeros(C, np.float32), dev, dil=1, bf16=True)
    sn = (torch.zeros(C, device=dev), torch.zeros(C, device=dev))
    print(f"C={C} k={k}: x {nbytes / 2**20:.0f} MiB at a 2 MiB boundary; y at x_end + skew")
    for rep in range(3):
        line = []
        for sk in skews:
            off = pad + nbytes + sk
            y = big[off:off + nbytes].view(torch.bfloat16).view(R, C)
            run = lambda: ops.resblock_step(c1, c2, x, y, rag, capi.PRE_SNAKE, 0.1, sn, sn, filt)
            run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                run()
            e1.record()
            torch.cuda.synchronize()
            line.append(f"{sk}:{1e3 * e0.elapsed_time(e1) / 8:.0f}")
        print("   " + "  ".join(line), flush=True)
