"""Diagnostic (library built with -DCONV_DIAG=6): clock stamps of the 128 x 128 conv tile on the stage-1 shape - main loop, epilogue issue,
epilogue drain per workgroup, in shader-clock-independent 100 MHz ticks (s_memtime).  TOUCAN_HIP_LIB=.../libcd6.so python tools/conv_trace.py"""
import ctypes as C
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
lib = capi.lib()
for k, dil, with_res in ((3, 1, False), (3, 1, True), (11, 5, False)):
    rag = Ragged([640 * 8] * 32, dev)
    R = rag.total_rows
    rs = np.random.RandomState(0)
    cw = packing.pack_conv((rs.randn(256, 256, k) / np.sqrt(256 * k)).astype(np.float32), np.zeros(256, np.float32), dev, dil=dil, bf16=True)
    x = torch.randn(R, 256, device=dev).to(torch.bfloat16)
    y = torch.empty(R, 256, device=dev, dtype=torch.bfloat16)
    res = torch.randn(R, 256, device=dev).to(torch.bfloat16) if with_res else None
    for _ in range(3):
        ops.conv(cw, x, y, rag, res=res, compute=capi.COMPUTE_BF16)
    torch.cuda.synchronize()
    buf = (C.c_uint64 * (4096 * 4))()
    lib.tts_conv_diag_trace.argtypes = [C.c_void_p]
    assert lib.tts_conv_diag_trace(buf) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 4).astype(np.int64)
    t = t[:2560]
    main, issue, drain, total = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 3] - t[:, 0]
    span = t[:, 3].max() - t[:, 0].min()
    q = lambda a: f"{np.median(a) / 100:.2f} us (p10 {np.percentile(a, 10) / 100:.2f}, p90 {np.percentile(a, 90) / 100:.2f})"
    print(f"k={k} res={with_res}: main loop {q(main)}; epilogue issue {q(issue)}; drain {q(drain)}; workgroup {q(total)}; launch span {span / 100:.1f} us")
