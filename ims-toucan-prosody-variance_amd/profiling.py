"""HIP-event timing of selected conv launches on the stream they are launched on (bench.py's roofline leg).

A ``ConvTimer`` is attached to an ``engine.Ops``; for every conv launch whose kernel class it selects it
records an event pair around the launch (torch.cuda.Event on the current stream == the launch stream).
Algorithmic work per launch = 2 * rows * cin * cout_total * taps FLOPs (each conv is
counted once with the reference op's own tap count) and rows * (cin + cout_total) * 4 bytes of activations.
"""
import torch

from . import capi


def kernel_class(cw, compute, tile_rows=None):
    """Name of the conv1d instantiation a launch runs: precision, tile (the 64 x 64 small-batch form when the host asked
    for 64-row tiles on a shape whose regular form is larger) and dual-accumulator mode."""
    dual = cw.mode != capi.MODE_LINEAR
    bm, bn = cw.tile_rows, cw.n_tile
    if tile_rows is not None and tile_rows != cw.tile_rows:
        bm, bn = tile_rows, 64
    prec = {capi.COMPUTE_F32: "f32", capi.COMPUTE_BF16: "bf16", capi.COMPUTE_F16: "f16", capi.COMPUTE_F32X3: "f32x3"}[compute]
    return "conv1d_%s<%dx%d%s>" % (prec, bm, bn, ",dual" if dual else "")


class ConvTimer:
    def __init__(self, select=None):
        self.select = select  # None: every class; else a set of class names
        self.records = []
        self.enabled = True

    def wants(self, cw, compute, tile_rows=None):
        return self.enabled and (self.select is None or kernel_class(cw, compute, tile_rows) in self.select)

    def wants_name(self, name):
        return self.enabled and (self.select is None or name in self.select)

    def add_named(self, name, flops, ev0, ev1, nbytes=0.0, elems=0.0):
        self.records.append((name, flops, ev0, ev1, nbytes, elems))

    def events(self):
        return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def add(self, cw, compute, rows, ev0, ev1, tile_rows=None, x_bytes=4, y_bytes=4):
        """Algorithmic bytes of a conv launch: its input and output tensors once (in their HBM element size) + its weights once."""
        ctot = cw.cout * (2 if cw.mode != capi.MODE_LINEAR else 1)
        flops = 2.0 * rows * cw.cin * ctot * cw.algo_taps
        wbytes = cw.taps * cw.cin * ctot * (4 if compute == capi.COMPUTE_F32 else 2)
        nbytes = float(rows) * (cw.cin * x_bytes + cw.cout * y_bytes) + wbytes
        self.records.append((kernel_class(cw, compute, tile_rows), flops, ev0, ev1, nbytes, float(rows) * cw.cout))

    def summary(self):
        """class -> dict(launches, total_ms, avg_us, flops_per_launch, tflops). Call after a device sync."""
        out = {}
        for name, flops, e0, e1, nbytes, elems in self.records:
            ms = e0.elapsed_time(e1)
            s = out.setdefault(name, dict(launches=0, total_ms=0.0, flops=0.0, bytes=0.0, elems=0.0))
            s["launches"] += 1
            s["total_ms"] += ms
            s["flops"] += flops
            s["bytes"] += nbytes
            s["elems"] += elems
        for s in out.values():
            s["avg_us"] = 1e3 * s["total_ms"] / s["launches"]
            s["flops_per_launch"] = s["flops"] / s["launches"]
            s["bytes_per_launch"] = s["bytes"] / s["launches"]
            s["elems_per_launch"] = s["elems"] / s["launches"]
            s["tflops"] = s["flops"] / (s["total_ms"] * 1e-3) / 1e12 if s["total_ms"] > 0 else 0.0
        return out

    def reset(self):
        self.records = []
