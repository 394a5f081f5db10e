"""Micro-benchmark of tts_resblock_step on the vocoder stage shapes (run on the MI355X box)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import ims_toucan_prosody_variance_amd  # noqa: F401
from ims_toucan_prosody_variance_amd import capi, engine, packing
from ims_toucan_prosody_variance_amd.ragged import Ragged


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=640)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--store", default="fp32", choices=["fp32", "bf16", "f16"])
    ap.add_argument("--channels", default="256,128,64,32", help="comma list of channel counts to run")
    ap.add_argument("--taps", default="3,7,11", help="comma list of kernel sizes to run")
    ap.add_argument("--acts", default="lrelu,snake")
    ap.add_argument("--dump", default=None, help="directory for the raw per-workgroup traces of a -DRB_DIAG_CLOCK library")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    ops = engine.Ops(dev)
    filt = torch.from_numpy(packing.kaiser_sinc_filter12()).to(dev)
    print(f"{'C':>4} {'k':>3} {'dil':>3} {'act':>6} {'us':>9} {'TFLOP/s':>8} {'GB/s(x+y)':>9}")
    trace = getattr(ops.lib, "tts_rb_diag_trace", None)
    if trace is not None:
        import ctypes as C_
        trace.argtypes, trace.restype = [C_.c_void_p, C_.c_int], C_.c_int
    want = {int(c) for c in args.channels.split(",")}
    pk = "f16" if args.store == "f16" else True
    for C, mult in ((256, 8), (128, 48), (64, 192), (32, 384)):
        if C not in want:
            continue
        rows = args.frames * mult
        rag = Ragged([rows] * args.batch, dev)
        R = rag.total_rows
        sdt = {"bf16": torch.bfloat16, "f16": torch.float16, "fp32": torch.float32}[args.store]
        x = torch.randn(R, C, device=dev).to(sdt)
        y = torch.empty(R, C, device=dev, dtype=sdt)
        sn = (torch.zeros(C, device=dev), torch.zeros(C, device=dev))
        for k, dil in ((3, 1), (7, 3), (11, 5)):
            if str(k) not in args.taps.split(","):
                continue
            rs = np.random.RandomState(0)
            c1 = packing.pack_conv((rs.randn(C, C, k) / np.sqrt(C * k)).astype(np.float32), np.zeros(C, np.float32), dev, dil=dil, bf16=pk)
            c2 = packing.pack_conv((rs.randn(C, C, k) / np.sqrt(C * k)).astype(np.float32), np.zeros(C, np.float32), dev, dil=1, bf16=pk)
            for act, name in ((capi.PRE_LRELU, "lrelu"), (capi.PRE_SNAKE, "snake")):
                if name not in args.acts.split(","):
                    continue
                run = lambda: ops.resblock_step(c1, c2, x, y, rag, act, 0.1, sn, sn, filt)
                tile_rows = ops.lib.tts_resblock_tile_rows(C)
                run()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    run()
                e1.record()
                torch.cuda.synchronize()
                us = 1e3 * e0.elapsed_time(e1) / args.reps
                if trace is not None and name == "snake":  # (diagnostic library built with -DRB_DIAG_CLOCK=<wave>)
                    nwg = min(40000, -(-rows // tile_rows) * args.batch, int(os.environ.get('RB_TRACE_WGS', '256' if C == 128 else '512')))
                    tb = np.zeros((nwg, 16), dtype=np.uint64)
                    assert trace(tb.ctypes.data, nwg) == nwg
                    if args.dump:
                        os.makedirs(args.dump, exist_ok=True)
                        np.save(os.path.join(args.dump, f"trace_C{C}_k{k}.npy"), tb)
                    ph = tb[:, :9].astype(np.float64) / np.maximum(1, tb[:, 9:10].astype(np.float64))  # cycles per tile, per workgroup
                    names = ("stage", "sync+begin", "sweep1", "conv1", "t1", "begin2", "sweep2", "conv2", "epilogue")
                    print("      phases (median cycles per tile): " + "  ".join(f"{nm} {np.median(ph[:, i]):.0f}" for i, nm in enumerate(names))
                          + f"  | total {np.median(ph.sum(axis=1)):.0f}; tiles per workgroup {tb[:, 9].min()}..{tb[:, 9].max()}", flush=True)
                    t0, t1_, hw = tb[:, 10].astype(np.int64), tb[:, 11].astype(np.int64), tb[:, 12]
                    dur = (t1_ - t0) * 10e-3  # us (100 MHz ticks)
                    ghz = tb[:, :9].astype(np.float64).sum() / ((t1_ - t0).sum() * 10.0)
                    key = (hw >> np.uint64(32)) * np.uint64(1 << 16) + (hw & np.uint64(0xFF00))
                    cus = np.unique(key)
                    busy, span = [], []
                    for k_ in cus:
                        m = key == k_
                        span.append((t1_[m].max() - t0[m].min()) * 10e-3)
                        busy.append((t1_[m] - t0[m]).sum() * 10e-3)
                    print(f"      trace: {nwg} workgroups on {len(cus)} CUs; lifetime median {np.median(dur):.1f} us (min {dur.min():.1f}, max {dur.max():.1f}); "
                          f"shader clock {ghz:.2f} GHz; per CU: span {np.mean(span):.0f} us, lifetimes {np.mean(busy):.0f} us "
                          f"(= {np.mean(busy) / np.mean(span):.2f} resident); launch span {(t1_.max() - t0.min()) * 10e-3:.0f} us", flush=True)
                    xcc = (hw >> np.uint64(32)).astype(np.int64) & 15
                    print("      per XCD (tiles per workgroup mean | first start, last end in us after the launch's first start): "
                          + "  ".join(f"{x}: {tb[xcc == x, 9].astype(np.float64).mean():.1f} | {(t0[xcc == x].min() - t0.min()) * 10e-3:.0f}, {(t1_[xcc == x].max() - t0.min()) * 10e-3:.0f}"
                                      for x in np.unique(xcc)), flush=True)
                flops = 2 * 2.0 * R * C * C * k
                print(f"{C:>4} {k:>3} {dil:>3} {name:>6} {us:9.1f} {flops / us / 1e6:8.1f} {2.0 * R * C * 4 / 1e9 / (us * 1e-6):9.0f}", flush=True)


if __name__ == "__main__":
    main()
