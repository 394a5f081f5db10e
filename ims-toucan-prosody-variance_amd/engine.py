"""Host-side sequencing of the HIP kernels: acoustic model and vocoders.

Python only orders launches and owns the buffers (torch tensors = device memory); every arithmetic
operation of the forward pass is a call into libtoucan_hip.so (capi.py).  There is no torch compute
on the path and no CPU fallback.

Layout: activations are time-major [rows, channels], all utterances of the batch packed along rows
(ragged.py).  The reference mirrors are cited per method.
"""
import ctypes as C
import os
import math

import numpy as np
import torch

from . import capi, packing
from .capi import (ACT_NONE, ACT_RELU, ACT_TANH, COMPUTE_BF16, COMPUTE_F16, COMPUTE_F32, COMPUTE_F32X3, MODE_COUPLING, MODE_GATED, MODE_GLU,
                   MODE_LINEAR, PRE_LRELU, PRE_NONE, PRE_SNAKE)
from .ragged import Ragged

ATT, HEADS, DK = 192, 4, 48


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _is_bf16(t):
    """t is a 16-bit tensor in HBM (bf16 in the bf16 configuration, fp16 in the fp16 one)."""
    return t is not None and t.dtype in (torch.bfloat16, torch.float16)


def _f16_flag(*ts):
    return capi.IO_F16 if any(t is not None and t.dtype == torch.float16 for t in ts) else 0


def precision_of(bf16, precision):
    """(`bf16` flag of the older API, `precision` in {None, "f32", "f32x3", "bf16", "f16"}) -> (name, pack argument, compute, torch
    dtype of the 16-bit tensors).  "f32x3": fp32 tensors everywhere, the dense products of the frame stages as three fp16 MFMAs on
    split operands (TTS_COMPUTE_F32X3: ~22 bits per product at 16/3 of the fp32 matrix rate) - a 32-bit configuration in every
    other respect (no fused 16-bit kernels, fp32 attention, exact fp32 phoneme stages)."""
    name = precision if precision is not None else ("bf16" if bf16 else "f32")
    if name in ("fp32", "f32"):
        return "f32", False, COMPUTE_F32, torch.float32
    if name in ("f32x3", "fp32x3"):
        return "f32x3", "x3", COMPUTE_F32X3, torch.float32
    if name == "bf16":
        return "bf16", "bf16", COMPUTE_BF16, torch.bfloat16
    if name in ("fp16", "f16"):
        return "f16", "f16", COMPUTE_F16, torch.float16
    raise ValueError(f"precision {name!r}: expected f32, bf16 or f16")


def _ld(t):
    if t is None:
        return 0
    if t.dim() == 1:
        return 1
    assert t.stride(-1) == 1, "kernels need unit channel stride"
    return t.stride(0)


class Ops:
    """Thin typed wrappers over the C ABI; all launches go to the current torch stream of `device`."""

    def __init__(self, device, lib=None):
        """lib: the loaded libtoucan_hip.so (default) - or, in tests only, the numpy ABI emulator working on host memory."""
        self.lib = capi.lib() if lib is None else lib
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        if isinstance(self.lib, C.CDLL) and self.device.type != "cuda":
            # the kernels dereference device pointers: host tensors would end in a GPU memory fault, not in a Python error
            raise capi.ToucanHipError(f"device {str(self.device)!r}: libtoucan_hip.so has no CPU path - construct the engines / the "
                                      f"interface with device='cuda' (the reference's default 'cpu' cannot be served)")
        self.timer = None  # optional profiling.ConvTimer (bench.py): HIP events around selected conv launches
        # fp32 configuration only, set per stage by the acoustic engine (same rule as pipeline.hip's Handle::split_mode): 2 = split-K
        # convs / key-split attention at every grid size (phoneme stages: everything upstream of the rounded durations keeps one
        # arithmetic whatever the batch), 1 = on small grids only (frame stages), 0 = never (vocoder: chunked == whole)
        self.split_k = 0
        self.small_tile_blocks = int(os.environ.get("TOUCAN_SMALL_TILE_BLOCKS", "1536"))  # regular conv grids below this many workgroups switch to the 64 x 64 small-batch form (0: never)
        self._fir_tabs = {}
        self.default_compute = COMPUTE_F32  # convs whose weights carry a 16-bit copy run on bf16 / fp16 MFMA when this is not COMPUTE_F32

    def stream(self):
        if self.device.type == "cuda":
            return torch.cuda.current_stream(self.device).cuda_stream
        return 0

    def empty(self, *shape, dtype=torch.float32):
        return torch.empty(*shape, dtype=dtype, device=self.device)

    def conv(self, cw, x, y, rag, pre=PRE_NONE, pre_slope=0.0, act=ACT_NONE, alpha=1.0, seqvec=None, preadd=None, res=None,
             res_scale=1.0, aux=None, accumulate=False, compute=None, snake=None, split_k=True):
        if compute is None:
            compute = self.default_compute
        tile_rows = cw.tile_rows
        if self.small_tile_blocks and cw.small_tile_rows and not cw.small_only:
            # grid of the regular form; when it cannot fill the chip, the 64 x 64 form runs ~3x more workgroups
            cols = cw.wn if cw.mode == MODE_LINEAR else cw.half_pad
            if -(-rag.total_rows // cw.tile_rows) * (cols // cw.n_tile) < self.small_tile_blocks:
                tile_rows = cw.small_tile_rows
        tiles, n_tiles = rag.tiles(tile_rows)
        d = capi.TtsConvDesc()
        d.x, d.ldx, d.cin = x.data_ptr(), _ld(x), cw.cin
        use_bf16 = compute != COMPUTE_F32 and cw.w16 is not None  # the conv then runs in the format of its own 16-bit copy
        if use_bf16 and cw.compute16 == COMPUTE_F32X3:
            # the split fp32 product (same rule as pipeline.hip conv()): not in the phoneme stages (exact fp32 upstream of the rounded
            # durations), and not where the exact fp32 split-K form is the fast one (frame stages on grids of a few workgroups)
            cols = cw.wn if cw.mode == MODE_LINEAR else cw.half_pad
            small = n_tiles * (tile_rows // 64) * (-(-cols // 64)) <= 128
            if self.split_k == 2 or (self.split_k == 1 and small):
                use_bf16 = False
        d.w = cw.w16.data_ptr() if use_bf16 else cw.w.data_ptr()
        d.cin_pad, d.wn, d.half_pad = cw.cin_pad, cw.wn, cw.half_pad
        d.bias = _ptr(cw.bias)
        d.y, d.ldy, d.cout = y.data_ptr(), _ld(y), cw.cout
        d.taps, d.dil, d.pad_left = cw.taps, cw.dil, cw.pad_left
        d.pre_act, d.pre_slope = pre, pre_slope
        if snake is not None:  # (alpha[cin], beta[cin], filter[12]) for PRE_SNAKE
            d.snake_alpha, d.snake_beta, d.snake_filt = snake[0].data_ptr(), snake[1].data_ptr(), snake[2].data_ptr()
        d.mode, d.act, d.alpha = cw.mode, act, alpha
        d.seqvec, d.ld_seqvec = _ptr(seqvec), _ld(seqvec)
        d.preadd, d.ld_preadd = _ptr(preadd), _ld(preadd)
        d.res, d.ld_res, d.res_scale = _ptr(res), _ld(res), res_scale
        d.aux, d.ld_aux = _ptr(aux), _ld(aux)
        d.accumulate = 1 if accumulate else 0
        d.compute = cw.compute16 if use_bf16 else COMPUTE_F32
        # 16-bit tensors in HBM are recognised by dtype (strides are already in elements)
        d.io_flags = (capi.IO_X_BF16 if _is_bf16(x) else 0) | (capi.IO_Y_BF16 if _is_bf16(y) else 0) | (capi.IO_RES_BF16 if _is_bf16(res) else 0) \
            | _f16_flag(x, y, res)
        if split_k and self.split_k and self.default_compute in (COMPUTE_F32, COMPUTE_F32X3) and d.compute == COMPUTE_F32:
            # same rule as pipeline.hip conv(): the fp32 configuration only - the fp32 layers of a 16-bit configuration keep one
            # accumulation order at every batch size (an utterance's result there does not depend on the batch it is in, bit for bit)
            d.io_flags |= capi.IO_SPLIT_K_ALWAYS if self.split_k == 2 else capi.IO_SPLIT_K
        d.tiles, d.n_tiles, d.tile_rows = tiles.data_ptr(), n_tiles, tile_rows
        tm = self.timer
        if tm is not None and tm.wants(cw, d.compute, tile_rows):
            ev0, ev1 = tm.events()
            ev0.record()
            capi.check(self.lib.tts_conv1d(C.byref(d), self.stream()), "tts_conv1d")
            ev1.record()
            tm.add(cw, d.compute, sum(rag.lengths), ev0, ev1, tile_rows, x.element_size(), y.element_size())
        else:
            capi.check(self.lib.tts_conv1d(C.byref(d), self.stream()), "tts_conv1d")
        return y

    def resblock_step(self, c1, c2, x, y, rag, act, slope=0.1, snake1=None, snake2=None, filt=None, alpha=1.0, res_scale=1.0,
                      accumulate=False, fir_tab=None):
        """Fused residual step (tts_resblock_step): y = alpha*conv2(act(conv1(act(x)))) + res_scale*x (+ y)."""
        tile_rows = self.lib.tts_resblock_tile_rows(c1.cin)
        tiles, n_tiles = rag.tiles(tile_rows)
        d = capi.TtsResblockDesc()
        d.x, d.ldx, d.y, d.ldy = x.data_ptr(), _ld(x), y.data_ptr(), _ld(y)
        d.c, d.taps, d.dil = c1.cin, c1.taps, c1.dil
        d.w1, d.b1, d.w2, d.b2 = c1.w16.data_ptr(), c1.bias.data_ptr(), c2.w16.data_ptr(), c2.bias.data_ptr()
        assert c1.compute16 == c2.compute16 != COMPUTE_F32
        d.compute = c1.compute16
        d.act, d.slope = act, slope
        if snake1 is not None:
            d.alpha1, d.beta1 = snake1[0].data_ptr(), snake1[1].data_ptr()
            d.alpha2, d.beta2 = snake2[0].data_ptr(), snake2[1].data_ptr()
            d.filt = filt.data_ptr()
            if fir_tab is None:  # callers without a prepared table (tests, micro-benchmarks): built once per filter tensor
                key = (filt.data_ptr(), str(filt.device))
                if key not in self._fir_tabs:
                    self._fir_tabs[key] = (packing.snake_fir_table(filt.detach().cpu().numpy(), filt.device), filt)
                fir_tab = self._fir_tabs[key][0]
            d.fir_tab = fir_tab.data_ptr()
        d.alpha, d.res_scale, d.accumulate = alpha, res_scale, 1 if accumulate else 0
        assert x.dtype == y.dtype and (not _is_bf16(x) or (x.dtype == torch.float16) == (c1.compute16 == COMPUTE_F16))
        d.io_bf16 = 1 if _is_bf16(x) else 0
        d.tiles, d.n_tiles, d.tile_rows = tiles.data_ptr(), n_tiles, tile_rows
        tm = self.timer
        if tm is not None and tm.wants_name("resblock_step<%d>" % c1.cin):
            ev0, ev1 = tm.events()
            ev0.record()
            capi.check(self.lib.tts_resblock_step(C.byref(d), self.stream()), "tts_resblock_step")
            ev1.record()
            rows = sum(rag.lengths)
            # algorithmic bytes: x read once + y written once (in their HBM element size) + both weight sets once
            nbytes = 2.0 * rows * c1.cin * x.element_size() + 2.0 * c1.taps * c1.cin * c1.cin * 2
            tm.add_named("resblock_step<%d>" % c1.cin, 2.0 * rows * c1.cin * c1.cin * c1.taps * 2, ev0, ev1, nbytes, float(rows) * c1.cin)
        else:
            capi.check(self.lib.tts_resblock_step(C.byref(d), self.stream()), "tts_resblock_step")
        return y

    def wavenet_layer(self, inl, res_skip, hs_in, hs_out, cond, rag):
        """Fused WaveNet layer (tts_wavenet_layer): hs_out = hs_in + res_skip(tanh.sigmoid(in_layer(h) + cond)); 16-bit modes only."""
        tiles, n = rag.tiles(64)
        d = capi.TtsWavenetDesc()
        d.hs_in, d.ld_in, d.hs_out, d.ld_out = hs_in.data_ptr(), _ld(hs_in), hs_out.data_ptr(), _ld(hs_out)
        d.cond, d.ld_cond = cond.data_ptr(), _ld(cond)
        d.w1, d.b1, d.w2, d.b2 = inl.w16.data_ptr(), inl.bias.data_ptr(), res_skip.w16.data_ptr(), res_skip.bias.data_ptr()
        assert inl.compute16 == res_skip.compute16 != COMPUTE_F32 and inl.wn == 384 and inl.taps == 5 and res_skip.wn == res_skip.cout
        d.cout2, d.compute = res_skip.cout, inl.compute16
        d.tiles, d.n_tiles, d.tile_rows = tiles.data_ptr(), n, 64
        capi.check(self.lib.tts_wavenet_layer(C.byref(d), self.stream()), "tts_wavenet_layer")
        return hs_out

    def ffn_fused(self, x, y, norm, fused, b2, rows, compute, post=None, alpha=0.5, eps=1e-12):
        """Fused feed-forward module (tts_ffn_fused): y = [LN_post](x + alpha (W2 relu(W1 LN(x) + b1) + b2)); 16-bit modes, kernel size 1.
        fused: packing.pack_ffn's tensor; norm / post: (gamma, beta)."""
        d = capi.TtsFfnDesc()
        d.x, d.ldx, d.y, d.ldy, d.rows, d.channels = x.data_ptr(), _ld(x), y.data_ptr(), _ld(y), rows, x.shape[1]
        d.ln_g, d.ln_b = norm[0].data_ptr(), norm[1].data_ptr()
        d.w, d.b2 = fused.data_ptr(), b2.data_ptr()
        d.post_g, d.post_b = (post[0].data_ptr(), post[1].data_ptr()) if post is not None else (None, None)
        d.hidden, d.compute, d.alpha, d.eps = 32 * (fused.numel() // (14 * 1024)), compute, alpha, eps  # (28 KB of int16 per 32 hidden channels)
        capi.check(self.lib.tts_ffn_fused(C.byref(d), self.stream()), "tts_ffn_fused")
        return y

    def layernorm(self, x, y, gamma, beta, rows, c, eps=1e-12):
        capi.check(self.lib.tts_layernorm(x.data_ptr(), _ld(x), y.data_ptr(), _ld(y), gamma.data_ptr(), beta.data_ptr(), rows, c, eps,
                                          self.stream()), "tts_layernorm")
        return y

    def cond_layernorm(self, x, y, scale, shift, c, rag):
        tiles, n = rag.tiles(64)
        capi.check(self.lib.tts_cond_layernorm(x.data_ptr(), _ld(x), y.data_ptr(), _ld(y), scale.data_ptr(), shift.data_ptr(), c,
                                               tiles.data_ptr(), n, 64, self.stream()), "tts_cond_layernorm")
        return y

    def cln_mlp(self, e_norm, weights, n_mlp, d_in, d_out):
        out = self.empty(n_mlp, e_norm.shape[0], d_out)
        capi.check(self.lib.tts_cln_mlp(e_norm.data_ptr(), e_norm.shape[0], d_in, d_out, weights.data_ptr(), n_mlp, out.data_ptr(),
                                        self.stream()), "tts_cln_mlp")
        return out

    def l2_normalize(self, x, y):
        capi.check(self.lib.tts_l2_normalize(x.data_ptr(), y.data_ptr(), x.shape[0], x.shape[1], self.stream()), "tts_l2_normalize")
        return y

    def groupnorm(self, x, y, gamma, beta, c, groups, rag, tanh, res=None, eps=1e-5):
        sb, se = rag.bounds()
        ws = self.empty(max(1, int(self.lib.tts_groupnorm_workspace_floats(rag.n_seq, rag.max_len, groups))))
        capi.check(self.lib.tts_groupnorm(x.data_ptr(), _ld(x), y.data_ptr(), _ld(y), gamma.data_ptr(), beta.data_ptr(), c, groups, eps,
                                          1 if tanh else 0, _ptr(res), _ld(res), sb.data_ptr(), se.data_ptr(), rag.n_seq, rag.max_len,
                                          ws.data_ptr(), self.stream()), "tts_groupnorm")
        return y

    def attention(self, qkv, ptab, pmax, bias_u, bias_v, ctx, rag, tile_rows=128, f16=False, flags=None):
        """Matrix-core kernels: fp32-input MFMA (exact fp32 products), or f16: the fp16-MFMA form of the 16-bit configurations.
        flags (fp32 kernel): capi.ATT_KEY_SPLIT / ATT_KEY_SPLIT_ALWAYS; default: by the stage's split mode in the fp32 configuration,
        0 (the plain form) in the 16-bit ones."""
        tiles, n = rag.tiles(tile_rows)
        args = (qkv.data_ptr(), _ld(qkv), ptab.data_ptr(), pmax, bias_u.data_ptr(), bias_v.data_ptr(), ctx.data_ptr(), _ld(ctx), HEADS, DK,
                tiles.data_ptr(), n, tile_rows)
        if f16:
            capi.check(self.lib.tts_relpos_attention_f16(*args, self.stream()), "tts_relpos_attention_f16")
            return ctx
        if flags is None:
            flags = 0 if self.default_compute not in (COMPUTE_F32, COMPUTE_F32X3) else {0: 0, 1: capi.ATT_KEY_SPLIT, 2: capi.ATT_KEY_SPLIT_ALWAYS}[self.split_k]
        capi.check(self.lib.tts_relpos_attention(*args, flags, self.stream()), "tts_relpos_attention")
        return ctx

    def dwconv_swish(self, x, y, w, b, c, k, rag):
        tiles, n = rag.tiles(64)
        capi.check(self.lib.tts_dwconv_swish(x.data_ptr(), _ld(x), y.data_ptr(), _ld(y), w.data_ptr(), b.data_ptr(), c, k,
                                             tiles.data_ptr(), n, 64, self.stream()), "tts_dwconv_swish")
        return y

    def duration_from_log(self, logd, dur):
        capi.check(self.lib.tts_duration_from_log(logd.data_ptr(), dur.data_ptr(), logd.numel(), self.stream()), "tts_duration_from_log")

    def prosody_control(self, text, pitch, energy, dur, rag, duration_scale, pitch_scale, energy_scale, pause_scale):
        sb, se = rag.bounds()
        capi.check(self.lib.tts_prosody_control(text.data_ptr(), _ld(text), pitch.data_ptr(), energy.data_ptr(), dur.data_ptr(),
                                                sb.data_ptr(), se.data_ptr(), rag.n_seq, duration_scale, pitch_scale, energy_scale,
                                                pause_scale, self.stream()), "tts_prosody_control")

    def length_regulate(self, enc, pitch, energy, wp, bp, we, be, dur, rag_phone, rag_frame, up, dec_in, dec_scale):
        pb, pe = rag_phone.bounds()
        fb, _ = rag_frame.bounds()
        capi.check(self.lib.tts_length_regulate(enc.data_ptr(), _ld(enc), pitch.data_ptr(), energy.data_ptr(), wp.data_ptr(),
                                                bp.data_ptr(), we.data_ptr(), be.data_ptr(), dur.data_ptr(), pb.data_ptr(), pe.data_ptr(),
                                                fb.data_ptr(), rag_phone.n_seq, rag_frame.max_len, rag_phone.max_len, enc.shape[1],
                                                up.data_ptr(), _ld(up), _ptr(dec_in), _ld(dec_in), dec_scale, self.stream()),
                   "tts_length_regulate")

    def glow_invconv_actnorm(self, x, rows, c, winv, an_bias, an_logs):
        capi.check(self.lib.tts_glow_invconv_actnorm(x.data_ptr(), _ld(x), rows, c, winv.data_ptr(), an_bias.data_ptr(), an_logs.data_ptr(),
                                                     self.stream()), "tts_glow_invconv_actnorm")

    def snake_aa(self, x, y, alpha, beta, filt, c, rag):
        tiles, n = rag.tiles(256)  # 8 streamed groups of 32 frames per tile: 256 work items already at c = 32
        flags = (capi.IO_X_BF16 if _is_bf16(x) else 0) | (capi.IO_Y_BF16 if _is_bf16(y) else 0) | _f16_flag(x, y)
        capi.check(self.lib.tts_snake_aa(x.data_ptr(), _ld(x), y.data_ptr(), _ld(y), alpha.data_ptr(), beta.data_ptr(), filt.data_ptr(), c,
                                         tiles.data_ptr(), n, 256, flags, self.stream()), "tts_snake_aa")
        return y

    def conv_post(self, x, cin, w, bias, pre, slope, wav, rag):
        tiles, n = rag.tiles(256)
        capi.check(self.lib.tts_conv_post(x.data_ptr(), _ld(x), cin, w.data_ptr(), bias, pre, slope, wav.data_ptr(), tiles.data_ptr(), n,
                                          256, (capi.IO_X_BF16 if _is_bf16(x) else 0) | _f16_flag(x), self.stream()), "tts_conv_post")
        return wav

    def conv_post_snake(self, x, cin, w, bias, alpha, beta, filt, wav, rag):
        tr = self.lib.tts_conv_post_snake_tile_rows()
        tiles, n = rag.tiles(tr)
        capi.check(self.lib.tts_conv_post_snake(x.data_ptr(), _ld(x), cin, w.data_ptr(), bias, alpha.data_ptr(), beta.data_ptr(), filt.data_ptr(),
                                                wav.data_ptr(), tiles.data_ptr(), n, tr, (capi.IO_X_BF16 if _is_bf16(x) else 0) | _f16_flag(x), self.stream()),
                   "tts_conv_post_snake")
        return wav

    def gather_rows(self, src, idx, dst):
        capi.check(self.lib.tts_gather_rows(src.data_ptr(), _ld(src), idx.data_ptr(), dst.data_ptr(), _ld(dst), idx.numel(), src.shape[1],
                                            self.stream()), "tts_gather_rows")
        return dst


def _dev(a, device):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)


class GraphCache:
    """HIP-graph capture/replay of shape-static launch sequences (torch.cuda.CUDAGraph captures every kernel that
    libtoucan_hip.so enqueues on the capturing stream).  One graph per key; inputs are copied into graph-owned buffers
    before each replay, outputs are graph-owned tensors (valid until the next replay of the same key)."""

    def __init__(self, device, max_entries=16):
        self.device = torch.device(device)
        self.entries = {}
        self.max_entries = max_entries

    def run(self, key, inputs, fn):
        entry = self.entries.get(key)
        if entry is None:
            if len(self.entries) >= self.max_entries:
                self.entries.clear()
            static = {k: (None if v is None else v.clone()) for k, v in inputs.items()}
            side = torch.cuda.Stream(self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):  # eager warm-up: builds tile tables / position tables outside the capture
                fn(**static)
            torch.cuda.current_stream(self.device).wait_stream(side)
            torch.cuda.synchronize(self.device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                outs = fn(**static)
            # `fn` stays referenced by the entry: its closure owns every external device buffer the captured launches read by raw
            # pointer (the Ragged layouts with their tile tables and bounds) - they must outlive Ragged's own layout cache
            entry = self.entries[key] = (graph, static, outs, fn)
            graph.replay()
            return outs
        graph, static, outs, _keep = entry
        for k, v in inputs.items():
            if v is not None:
                static[k].copy_(v)
        graph.replay()
        return outs


NO_F16_ATTENTION = bool(os.environ.get("TOUCAN_NO_F16_ATTENTION"))  # (A/B switch, csrc/pipeline.hip reads the same variable)


class ConformerWeights:
    """Packed weights of one Layers/Conformer.py stack (6 EncoderLayers)."""

    def __init__(self, sd, prefix, kernel, device, bf16=False):
        self.kernel = kernel
        self.blocks = []
        pack = lambda *a, **k: packing.pack_conv(*a, bf16=bf16, **k)
        for b in range(6):
            p = f"{prefix}.encoders.{b}."
            blk = {}
            for ln in ("norm_ff_macaron", "norm_mha", "norm_conv", "norm_ff", "norm_final"):
                blk[ln] = (_dev(sd[p + ln + ".weight"], device), _dev(sd[p + ln + ".bias"], device))
            for ff in ("feed_forward_macaron", "feed_forward"):
                blk[ff + ".w1"] = pack(sd[p + ff + ".w_1.weight"], sd[p + ff + ".w_1.bias"], device)
                blk[ff + ".w2"] = pack(sd[p + ff + ".w_2.weight"], sd[p + ff + ".w_2.bias"], device)
                # 16-bit configurations, kernel size 1: the whole module is one launch (tts_ffn_fused) on weights in its fragment order
                w1 = np.asarray(sd[p + ff + ".w_1.weight"])
                if bf16 in ("bf16", "f16", True) and not os.environ.get("TOUCAN_NO_FUSED_FFN") and (w1.ndim == 2 or w1.shape[2] == 1) and w1.shape[1] == ATT and w1.shape[0] % 32 == 0:
                    blk[ff + ".fused"] = packing.pack_ffn(w1, sd[p + ff + ".w_1.bias"], sd[p + ff + ".w_2.weight"], device, "f16" if bf16 == "f16" else "bf16")
            a = p + "self_attn."
            wqkv = np.concatenate([sd[a + f"linear_{n}.weight"] for n in "qkv"], axis=0)
            bqkv = np.concatenate([sd[a + f"linear_{n}.bias"] for n in "qkv"], axis=0)
            blk["qkv"] = pack(wqkv, bqkv, device)
            blk["out"] = pack(sd[a + "linear_out.weight"], sd[a + "linear_out.bias"], device)
            blk["pos"] = packing.pack_conv(sd[a + "linear_pos.weight"], None, device)  # table built once, fp32
            blk["u"] = _dev(sd[a + "pos_bias_u"].reshape(-1), device)
            blk["v"] = _dev(sd[a + "pos_bias_v"].reshape(-1), device)
            c = p + "conv_module."
            blk["pw1"] = pack(sd[c + "pointwise_conv1.weight"], sd[c + "pointwise_conv1.bias"], device, mode=MODE_GLU)
            # BatchNorm1d eval (running stats, eps 1e-5) folded into the depthwise conv: Convolution.py:26-27,50-51
            g = sd[c + "norm.weight"].astype(np.float64) / np.sqrt(sd[c + "norm.running_var"].astype(np.float64) + 1e-5)
            dw = sd[c + "depthwise_conv.weight"][:, 0, :].astype(np.float64) * g[:, None]  # [c, k]
            db = (sd[c + "depthwise_conv.bias"].astype(np.float64) - sd[c + "norm.running_mean"].astype(np.float64)) * g \
                + sd[c + "norm.bias"].astype(np.float64)
            blk["dw_w"] = _dev(dw.T, device)  # [k][c]
            blk["dw_b"] = _dev(db, device)
            blk["pw2"] = pack(sd[c + "pointwise_conv2.weight"], sd[c + "pointwise_conv2.bias"], device)
            self.blocks.append(blk)
        self.pmax = 0
        self.ptabs = None


class AcousticEngine:
    """InferenceToucanTTS.ToucanTTS (:16-319) for a ragged batch of utterances."""

    def __init__(self, state_dict, device, bf16=False, use_graphs=False, precision=None, pack_only=False):
        """precision "bf16" (or bf16=True): Conformer / PostNet / PostFlow GEMMs on bf16 MFMA with fp32 accumulation and fp32
        activations (BASELINE.json configs[2]); "f16": the same GEMMs on fp16 MFMA (configs[4]).  In both, the variance predictors,
        all norms, softmax, the coupling output conv and the flow state stay fp32 (fp16 exp(-logs) chains
        over 18 blocks overflow otherwise)."""
        # pack_only: only the weight preparation (host tensors for native.NativePipeline to upload), no launch machinery
        self.ops = None if pack_only else Ops(device)
        self.device = torch.device(device) if pack_only else self.ops.device
        self.precision, bf16, compute16, self.dt16 = precision_of(bf16, precision)
        self.bf16 = bf16 in ("bf16", "f16", True)  # a 16-bit MFMA configuration (either format); "x3" keeps fp32 tensors
        self.use_graphs = use_graphs
        self._graphs = GraphCache(self.device)
        if bf16 and self.ops is not None:
            self.ops.default_compute = compute16
        dev = self.device
        sd = packing.fold_weight_norm(state_dict)
        self.multilingual = "encoder.language_embedding.weight" in sd
        self.multispeaker = "encoder.hs_emb_projection.weight" in sd  # else: utt_embed_dim=None variant (ToucanTTSInterface.py:61-63)
        pc = packing.pack_conv  # fp32 only (embedding, predictors, per-utterance vectors)
        pcb = lambda *a, **k: packing.pack_conv(*a, bf16=bf16, **k)
        self.embed0 = pc(sd["encoder.embed.0.weight"], sd["encoder.embed.0.bias"], dev)
        self.embed2 = pc(sd["encoder.embed.2.weight"], sd["encoder.embed.2.bias"], dev)
        self.lang_table = _dev(sd["encoder.language_embedding.weight"], dev) if self.multilingual else None
        self.enc = ConformerWeights(sd, "encoder", 7, dev, bf16)
        self.dec = ConformerWeights(sd, "decoder", 31, dev, bf16)
        self.out_norm = (_dev(sd["encoder.output_norm.weight"], dev), _dev(sd["encoder.output_norm.bias"], dev))
        if self.multispeaker:
            hs = sd["encoder.hs_emb_projection.weight"]
            self.hs_h = pcb(hs[:, :ATT], None, dev)  # acts on the hidden states
            self.hs_e = pc(hs[:, ATT:], sd["encoder.hs_emb_projection.bias"], dev)  # acts on the utterance embedding (+ bias)
        self.pred = {}
        cln_blocks = []
        for name, layers, k in (("pitch_predictor", 7, 5), ("energy_predictor", 2, 3), ("duration_predictor", 3, 3)):
            convs = [pc(sd[f"{name}.conv.{i}.0.weight"], sd[f"{name}.conv.{i}.0.bias"], dev) for i in range(layers)]
            first = len(cln_blocks) // 2  # index of this predictor's first conditional layer norm
            if not self.multispeaker:  # plain LayerNorm(256) per layer (VariancePredictor.py:47-48 / DurationPredictor.py:57-58)
                norms = [(_dev(sd[f"{name}.norms.{i}.weight"], dev), _dev(sd[f"{name}.norms.{i}.bias"], dev)) for i in range(layers)]
                self.pred[name] = (convs, norms, pc(sd[name + ".linear.weight"], sd[name + ".linear.bias"], dev))
                continue
            for i in range(layers):
                for which in ("W_scale", "W_bias"):  # ConditionalLayerNorm.py:26-35: Linear, Tanh, Linear, Tanh, Linear
                    q = f"{name}.norms.{i}.{which}."
                    parts = []
                    for j in (0, 2, 4):
                        parts += [np.ascontiguousarray(np.asarray(sd[q + f"{j}.weight"], dtype=np.float32).T).reshape(-1),
                                  np.asarray(sd[q + f"{j}.bias"], dtype=np.float32).reshape(-1)]
                    cln_blocks.append(np.concatenate(parts))
            lin = pc(sd[name + ".linear.weight"], sd[name + ".linear.bias"], dev)
            self.pred[name] = (convs, first, lin)
        # every scale / shift MLP of the 12 conditional layer norms in one buffer -> one tts_cln_mlp launch per forward
        self.n_cln_mlp = len(cln_blocks)
        if cln_blocks:
            self.cln_weights = _dev(np.concatenate(cln_blocks), dev)
            assert cln_blocks[0].size == capi.lib().tts_cln_mlp_weight_floats(64, 256)
        self.pitch_w = _dev(sd["pitch_embed.0.weight"].reshape(-1), dev)
        self.pitch_b = _dev(sd["pitch_embed.0.bias"], dev)
        self.energy_w = _dev(sd["energy_embed.0.weight"].reshape(-1), dev)
        self.energy_b = _dev(sd["energy_embed.0.bias"], dev)
        self.feat_out = pcb(sd["feat_out.weight"], sd["feat_out.bias"], dev)
        self.postnet = [(pcb(sd[f"conv_postnet.postnet.{i}.0.weight"], None, dev), _dev(sd[f"conv_postnet.postnet.{i}.1.weight"], dev),
                         _dev(sd[f"conv_postnet.postnet.{i}.1.bias"], dev)) for i in range(5)]
        # PostFlow
        self.g_proj = pcb(sd["post_flow.g_proj.weight"], sd["post_flow.g_proj.bias"], dev)
        self.flow = []
        for b in range(18):
            pa, pn, pcp = (f"post_flow.flows.{3 * b + i}." for i in range(3))
            blk = dict(an_bias=_dev(sd[pa + "bias"].reshape(-1), dev), an_logs=_dev(sd[pa + "logs"].reshape(-1), dev),
                       winv=_dev(packing.invconv_inverse(sd, pn), dev),
                       start=pcb(*self._start_with_zeroed_skip(sd[pcp + "start.weight"], sd[pcp + "start.bias"]), dev),
                       end=pc(sd[pcp + "end.weight"], sd[pcp + "end.bias"], dev, mode=MODE_COUPLING, small_only=True),  # m, logs: keep fp32
                       cond=pcb(sd[pcp + "wn.cond_layer.weight"], sd[pcp + "wn.cond_layer.bias"], dev))
            if b % 4 == 0 or not self.flow:  # in/res-skip layers are shared inside groups of 4 blocks (Glow.py:325-327)
                shared = dict(inl=[], res_skip=[])
                for i in range(4):
                    shared["inl"].append(pcb(sd[pcp + f"wn.in_layers.{i}.weight"], sd[pcp + f"wn.in_layers.{i}.bias"], dev, mode=MODE_GATED))
                    # layers 0-2: 384 outputs, first half feeds the residual stream, second half the skip sum (wavenet.py:112-118);
                    # the hidden state and the skip sum live side by side in one [rows, 384] tensor, so ONE accumulating conv
                    # updates both.  Layer 3: 192 outputs, skip only.
                    shared["res_skip"].append(pcb(sd[pcp + f"wn.res_skip_layers.{i}.weight"], sd[pcp + f"wn.res_skip_layers.{i}.bias"], dev))
            blk.update(shared)
            self.flow.append(blk)
        self._pe_cache = {}

    @staticmethod
    def _start_with_zeroed_skip(w, b):
        """CouplingBlock.start (Glow.py:232-241) widened from 192 to 384 outputs with zero weights: the extra half lands in the
        skip-sum columns that sit next to the hidden state, so the launch that starts a block also clears its skip sum."""
        w, b = np.asarray(w, dtype=np.float32), np.asarray(b, dtype=np.float32)
        return np.concatenate([w, np.zeros_like(w)], axis=0), np.concatenate([b, np.zeros_like(b)], axis=0)

    # ---- relative position tables -----------------------------------------------------------------
    def _ensure_ptabs(self, cw, pmax):
        """ptab_l[pmax-1+p] = linear_pos_l(pe(p)) for every block l (Attention.py:177, PositionalEncoding.py:90-130)."""
        if cw.pmax >= pmax:
            return
        # captured graphs hold the old tables' addresses and the old pmax: drop them before the tables are released
        self._graphs.entries.clear()
        pmax = max(pmax, 2 * cw.pmax, 256)
        ops = self.ops
        pe = _dev(packing.rel_pos_encoding(pmax), self.device)
        rag = Ragged.cached([2 * pmax - 1], self.device)
        # (never the split-K form: a table is shared by calls with different pmax, and its rows must not depend on the table's size)
        cw.ptabs = [ops.conv(blk["pos"], pe, ops.empty(2 * pmax - 1, ATT), rag, split_k=False) for blk in cw.blocks]
        cw.pmax = pmax

    # ---- Conformer stack ---------------------------------------------------------------------------
    def _conformer(self, cw, x, rag, taps=None, tap_name=None):
        """Layers/EncoderLayer.py:62-144 x6; x is the residual stream [rows,192], already scaled by sqrt(192)."""
        ops = self.ops
        R = x.shape[0]
        self._ensure_ptabs(cw, rag.max_len)
        ln = ops.empty(R, ATT)
        # tensors consumed only by bf16-MFMA convs are kept as bf16 in HBM: the consumer would round them to bf16 while staging
        # anyway (same round-to-nearest-even), so the result is bit-identical and the round trip costs half the bytes
        hid = ops.empty(R, 1536, dtype=self.dt16)
        qkv = ops.empty(R, 3 * ATT)
        ctx = ops.empty(R, ATT)
        glu = ops.empty(R, ATT)
        dwo = ops.empty(R, ATT)
        for li, blk in enumerate(cw.blocks):
            # (one launch per module whatever the number of rows: its time is flat at ~60 us - 48 dependent chunk steps - against 52 us
            # of the unfused launches at 4 096 rows and 212 us at 20 480; a row-count threshold would make an utterance's result
            # depend on the batch it is in)
            if "feed_forward_macaron.fused" in blk:
                ops.ffn_fused(x, x, blk["norm_ff_macaron"], blk["feed_forward_macaron.fused"], blk["feed_forward_macaron.w2"].bias, R,
                              blk["feed_forward_macaron.w1"].compute16)
            else:
                ops.layernorm(x, ln, *blk["norm_ff_macaron"], R, ATT)
                ops.conv(blk["feed_forward_macaron.w1"], ln, hid, rag, act=ACT_RELU)
                ops.conv(blk["feed_forward_macaron.w2"], hid, x, rag, alpha=0.5, res=x)
            ops.layernorm(x, ln, *blk["norm_mha"], R, ATT)
            ops.conv(blk["qkv"], ln, qkv, rag)
            ops.attention(qkv, cw.ptabs[li], cw.pmax, blk["u"], blk["v"], ctx, rag, f16=self.bf16 and not NO_F16_ATTENTION)
            ops.conv(blk["out"], ctx, x, rag, res=x)
            ops.layernorm(x, ln, *blk["norm_conv"], R, ATT)
            ops.conv(blk["pw1"], ln, glu, rag)
            ops.dwconv_swish(glu, dwo, blk["dw_w"], blk["dw_b"], ATT, cw.kernel, rag)
            ops.conv(blk["pw2"], dwo, x, rag, res=x)
            if "feed_forward.fused" in blk:  # (with the block's final norm in its epilogue)
                ops.ffn_fused(x, x, blk["norm_ff"], blk["feed_forward.fused"], blk["feed_forward.w2"].bias, R, blk["feed_forward.w1"].compute16,
                              post=blk["norm_final"])
            else:
                ops.layernorm(x, ln, *blk["norm_ff"], R, ATT)
                ops.conv(blk["feed_forward.w1"], ln, hid, rag, act=ACT_RELU)
                ops.conv(blk["feed_forward.w2"], hid, x, rag, alpha=0.5, res=x)
                ops.layernorm(x, x, *blk["norm_final"], R, ATT)
            if taps is not None:
                taps[f"{tap_name}_block{li}"] = x.clone()
        return x

    def _predictor(self, name, enc, cln, rag):
        """Layers/VariancePredictor.py:65-80 / DurationPredictor.py:63-74 with ConditionalLayerNorm.py:52-67.
        cln: [n_mlp, B, 256] scale / shift vectors of every conditional layer norm (tts_cln_mlp)."""
        ops = self.ops
        convs, first, lin = self.pred[name]
        R = enc.shape[0]
        h = enc
        a, bbuf = ops.empty(R, 256), ops.empty(R, 256)
        for i, cw in enumerate(convs):
            ops.conv(cw, h, a, rag, act=ACT_RELU)
            if self.multispeaker:
                ops.cond_layernorm(a, bbuf, cln[2 * (first + i)], cln[2 * (first + i) + 1], 256, rag)
            else:
                ops.layernorm(a, bbuf, *first[i], R, 256)  # `first` is the list of (weight, bias) pairs in this variant
            h = bbuf  # the next conv reads bbuf into a, then the norm overwrites bbuf: no aliasing
        out = ops.empty(R, 1)
        ops.conv(lin, h, out, rag)
        return out.view(-1)

    # ---- stage A: everything up to the final per-phoneme durations (no data-dependent shapes) ----------------
    def _stage_a(self, text, emb, lang_idx, gold_p, gold_e, gold_d, rag_p, rag_b, scales, taps=None):
        """Conformer.py:92-134, VariancePredictor / DurationPredictor, InferenceToucanTTS.py:214-227."""
        ops = self.ops
        ops.split_k = 2  # phoneme stages (fp32 configuration): the split forms at every grid size - see Ops.__init__
        R, B = text.shape[0], emb.shape[0]
        e_norm = ops.l2_normalize(emb, ops.empty(B, 64))
        h100 = ops.conv(self.embed0, text, ops.empty(R, 100), rag_p, act=ACT_TANH)
        seqvec = None
        if lang_idx is not None:
            seqvec = ops.gather_rows(self.lang_table, lang_idx, ops.empty(B, ATT))
        x = ops.conv(self.embed2, h100, ops.empty(R, ATT), rag_p, seqvec=seqvec, alpha=math.sqrt(ATT))
        if taps is not None:
            taps["enc_embed_scaled"] = x.clone()
        x = self._conformer(self.enc, x, rag_p, taps, "enc")
        ops.layernorm(x, x, *self.out_norm, R, ATT)
        if self.multispeaker:  # Conformer.py:130-134: projection of [hidden | normalised utterance embedding]
            e_proj = ops.conv(self.hs_e, e_norm, ops.empty(B, ATT), rag_b)
            enc = ops.conv(self.hs_h, x, ops.empty(R, ATT), rag_p, seqvec=e_proj)
        else:
            enc = x
        cln = None
        if self.multispeaker and (gold_p is None or gold_e is None or gold_d is None):
            cln = ops.cln_mlp(e_norm, self.cln_weights, self.n_cln_mlp, 64, 256)  # [n_mlp, B, 256]: all scale / shift vectors at once
        if gold_p is None:
            p = self._predictor("pitch_predictor", enc, cln, rag_p)
        else:
            p = ops.empty(R)
            p.copy_(gold_p)
        if gold_e is None:
            en = self._predictor("energy_predictor", enc, cln, rag_p)
        else:
            en = ops.empty(R)
            en.copy_(gold_e)
        d = ops.empty(R, dtype=torch.int32)
        if gold_d is None:
            logd = self._predictor("duration_predictor", enc, cln, rag_p)
            ops.duration_from_log(logd, d)
            if taps is not None:
                taps["log_dur"] = logd.clone()
        else:
            d.copy_(gold_d)
        if taps is not None:
            taps.update(enc_out=enc.clone(), pitch_raw=p.clone(), energy_raw=en.clone())
        ops.prosody_control(text, p, en, d, rag_p, *scales)
        return enc, p, en, d

    # ---- stage B: length regulator -> decoder -> PostNet -> PostFlow (shapes fixed by the durations) ----------
    def _stage_b(self, enc, p, en, d, z_sq, rag_p, rag_f, taps=None):
        ops, dev = self.ops, self.device
        ops.split_k = 1  # frame stages (fp32 configuration): the split forms on small grids only
        RF = rag_f.total_rows
        cat = torch.zeros(RF, 80 + ATT, dtype=torch.float32, device=dev)  # [refined mel | upsampled text] = g_proj input
        up = cat[:, 80:]
        dec_x = ops.empty(RF, ATT)
        ops.length_regulate(enc, p, en, self.pitch_w, self.pitch_b, self.energy_w, self.energy_b, d, rag_p, rag_f, up, dec_x,
                            math.sqrt(ATT))
        if taps is not None:
            taps["upsampled"] = up.clone()
        # decoder + feat_out (InferenceToucanTTS.py:238-239)
        dec_x = self._conformer(self.dec, dec_x, rag_f, taps, "dec")
        mel0 = ops.conv(self.feat_out, dec_x, ops.empty(RF, 80), rag_f)
        # PostNet (PostNet.py:62-74) + residual (InferenceToucanTTS.py:241)
        a, bb = ops.empty(RF, 256), ops.empty(RF, 256)
        src = mel0
        for i, (cw, gw, gb) in enumerate(self.postnet):
            if i < 4:
                ops.conv(cw, src, a, rag_f)  # reads src (mel0 or bb) -> a
                ops.groupnorm(a, bb, gw, gb, 256, 32, rag_f, tanh=True)  # a -> bb
                src = bb
            else:
                y80 = ops.conv(cw, src, ops.empty(RF, 80), rag_f)
                ops.groupnorm(y80, cat[:, :80], gw, gb, 80, 20, rag_f, tanh=False, res=mel0)
        mel = self._postflow(cat, rag_f, z_sq, taps) if z_sq is not None else cat[:, :80]
        return mel0, cat, mel

    @torch.inference_mode()
    def forward(self, texts, utt_embs, lang_ids=None, durations=None, pitch=None, energy=None, z_noise=None,
                duration_scaling_factor=1.0, pitch_variance_scale=1.0, energy_variance_scale=1.0,
                pause_duration_scaling_factor=1.0, run_postflow=True, taps=None, generator=None):
        """texts: list of [L_u,62] float tensors; utt_embs: [B,64]; lang_ids: list of int or None;
        durations/pitch/energy: optional lists (gold values, InferenceToucanTTS.py:209-211);
        z_noise: optional list of [80, T_u] tensors = 0.8*N(0,1) (Glow.py:363) - drawn on the device if omitted.
        Returns dict(mel=[list of [T'_u,80]], durations, pitch, energy, plus packed tensors).

        With ``use_graphs`` (CUDA devices, no taps) the two shape-static halves of the pass are captured once per shape
        signature into HIP graphs and replayed: ~900 launches become two graph launches, which is what batch-1 latency
        needs.  Tensors in the returned dict are then views of graph-owned buffers, valid until the next call."""
        if self.device.type == "cuda" and torch.cuda.current_device() != self.device.index:
            with torch.cuda.device(self.device):  # launches go to streams of self.device: it must be the current HIP device
                return self.forward(texts, utt_embs, lang_ids, durations, pitch, energy, z_noise, duration_scaling_factor,
                                    pitch_variance_scale, energy_variance_scale, pause_duration_scaling_factor, run_postflow, taps, generator)
        ops, dev = self.ops, self.device
        B = len(texts)
        assert duration_scaling_factor > 0
        Ls = [int(t.shape[0]) for t in texts]
        rag_p = Ragged.cached(Ls, dev)
        rag_b = Ragged.cached([B], dev)
        text = torch.cat([t.reshape(-1, 62).to(torch.float32) for t in texts], dim=0).to(dev).contiguous()
        emb = utt_embs.to(dev, torch.float32).reshape(B, 64).contiguous()

        def packed_gold(lst, dtype):
            return None if lst is None else torch.cat([torch.as_tensor(v).reshape(-1).to(dtype) for v in lst]).to(dev).contiguous()

        lang_idx = None
        if self.multilingual and lang_ids is not None:
            lang_idx = torch.tensor([int(i) for i in lang_ids], dtype=torch.int32).to(dev)
        gp, ge, gd = packed_gold(pitch, torch.float32), packed_gold(energy, torch.float32), packed_gold(durations, torch.int32)
        scales = (float(duration_scaling_factor), float(pitch_variance_scale), float(energy_variance_scale),
                  float(pause_duration_scaling_factor))
        graphs = self.use_graphs and taps is None and dev.type == "cuda"

        if graphs:
            key = ("A", tuple(Ls), lang_idx is not None, gp is not None, ge is not None, gd is not None, scales)
            ins = dict(text=text, emb=emb, lang_idx=lang_idx, gold_p=gp, gold_e=ge, gold_d=gd)
            enc, p, en, d = self._graphs.run(key, ins, lambda **kw: self._stage_a(rag_p=rag_p, rag_b=rag_b, scales=scales, **kw))
        else:
            enc, p, en, d = self._stage_a(text, emb, lang_idx, gp, ge, gd, rag_p, rag_b, scales, taps)

        # ---- the one host round trip: frame counts fix every later buffer size ----
        d_host = d.cpu().numpy()
        Ts = []
        for b0, n in zip(rag_p.begins, rag_p.lengths):
            t = int(d_host[b0:b0 + n].sum())
            Ts.append(t if t > 0 else n)  # LengthRegulator.py:52-53 (all-zero utterance -> all ones)
        rag_f = Ragged.cached(Ts, dev, align=2)  # even begins so that the Glow squeeze is a pure re-view
        RS = rag_f.total_rows // 2

        z_sq = None
        if run_postflow:
            rag_s = rag_f.halved()
            if z_noise is None:  # Glow.py:363: z ~ 0.8 * N(0,1), drawn per squeezed row on the device
                z_sq = torch.randn(RS, 160, device=dev, dtype=torch.float32, generator=generator) * 0.8
            else:
                z_sq = torch.zeros(RS, 160, dtype=torch.float32, device=dev)
                for zu, b0, n in zip(z_noise, rag_s.begins, rag_s.lengths):
                    z_sq[b0:b0 + n].copy_(torch.as_tensor(zu, dtype=torch.float32).t()[: 2 * n].reshape(n, 160))

        if graphs:
            key = ("B", tuple(Ls), tuple(Ts), z_sq is not None)
            ins = dict(enc=enc, p=p, en=en, d=d, z_sq=z_sq)
            mel0, cat, mel_packed = self._graphs.run(key, ins, lambda **kw: self._stage_b(rag_p=rag_p, rag_f=rag_f, **kw))
        else:
            mel0, cat, mel_packed = self._stage_b(enc, p, en, d, z_sq, rag_p, rag_f, taps)

        rag_out = rag_f.halved().doubled() if run_postflow else rag_f
        out = dict(durations_packed=d, pitch_packed=p, energy_packed=en, rag_phone=rag_p, rag_frame=rag_f, decoded_packed=mel0,
                   refined_packed=cat[:, :80], mel_packed=mel_packed, rag_mel=rag_out)
        out["mel"] = [mel_packed[b0:b0 + n] for b0, n in zip(rag_out.begins, rag_out.lengths)]
        out["durations"] = [d[b0:b0 + n] for b0, n in zip(rag_p.begins, rag_p.lengths)]
        out["pitch"] = [p[b0:b0 + n] for b0, n in zip(rag_p.begins, rag_p.lengths)]
        out["energy"] = [en[b0:b0 + n] for b0, n in zip(rag_p.begins, rag_p.lengths)]
        return out

    @torch.inference_mode()
    def predict_frame_counts(self, texts, utt_embs, lang_ids=None, pitch=None, energy=None, duration_scaling_factor=1.0,
                             pitch_variance_scale=1.0, energy_variance_scale=1.0, pause_duration_scaling_factor=1.0):
        """Stage A alone (encoder, predictors, control: InferenceToucanTTS.py:206-227): mel frames each utterance will have.
        Used to balance a multi-GPU deal by vocoder work before anything expensive runs (distributed.py)."""
        dev = self.device
        Ls = [int(t.shape[0]) for t in texts]
        rag_p, rag_b = Ragged.cached(Ls, dev), Ragged.cached([len(texts)], dev)
        text = torch.cat([t.reshape(-1, 62).to(torch.float32) for t in texts], dim=0).to(dev).contiguous()
        emb = utt_embs.to(dev, torch.float32).reshape(len(texts), 64).contiguous()
        cat = lambda lst: None if lst is None else torch.cat([torch.as_tensor(v).reshape(-1).to(torch.float32) for v in lst]).to(dev).contiguous()
        lang_idx = None
        if self.multilingual and lang_ids is not None:
            lang_idx = torch.tensor([int(i) for i in lang_ids], dtype=torch.int32).to(dev)
        scales = (float(duration_scaling_factor), float(pitch_variance_scale), float(energy_variance_scale), float(pause_duration_scaling_factor))
        _, _, _, d = self._stage_a(text, emb, lang_idx, cat(pitch), cat(energy), None, rag_p, rag_b, scales)
        d_host = d.cpu().numpy()
        return [int(d_host[b0:b0 + n].sum()) or n for b0, n in zip(rag_p.begins, rag_p.lengths)]

    def _postflow(self, cat, rag_f, z_sq, taps):
        """Glow.forward(infer=True) + _forward(reverse=True): Glow.py:342-391.  z_sq: the noise in squeezed layout [RS,160]."""
        ops, dev = self.ops, self.device
        RF = cat.shape[0]
        g = ops.conv(self.g_proj, cat, ops.empty(RF, ATT), rag_f)
        if taps is not None:
            taps["glow_g"] = g.clone()
        rag_s = rag_f.halved()
        RS = RF // 2
        g_sq = g.view(RS, 2 * ATT)  # squeeze == re-view in time-major layout (glow_utils.py:28-40)
        x = ops.empty(RS, 160)
        x.copy_(z_sq)
        hs = ops.empty(RS, 2 * ATT)  # [hidden state | skip sum] side by side: one accumulating conv per WaveNet layer updates both
        # 16-bit configurations: one launch per WaveNet layer (tts_wavenet_layer), the state ping-pongs between two buffers
        fused = self.bf16 and os.environ.get("TOUCAN_NO_FUSED_WAVENET") is None
        hs2 = ops.empty(RS, 2 * ATT) if fused else None
        h, skip = hs[:, :ATT], hs[:, ATT:]
        acts = ops.empty(RS, ATT, dtype=self.dt16)  # read only by the res/skip conv (16-bit MFMA)
        cond = ops.empty(RS, 8 * ATT)
        for b in reversed(range(18)):
            blk = self.flow[b]
            ops.conv(blk["start"], x[:, :80], hs, rag_s)  # h = start(x0); the zero-weight second half clears the skip sum
            ops.conv(blk["cond"], g_sq, cond, rag_s)
            cur = hs
            for i in range(4):
                if fused:
                    nxt = hs2 if cur is hs else hs
                    ops.wavenet_layer(blk["inl"][i], blk["res_skip"][i], cur, nxt, cond[:, i * 2 * ATT:(i + 1) * 2 * ATT], rag_s)
                    cur = nxt
                    continue
                ops.conv(blk["inl"][i], h, acts, rag_s, preadd=cond[:, i * 2 * ATT:(i + 1) * 2 * ATT])
                ops.conv(blk["res_skip"][i], acts, hs if i < 3 else skip, rag_s, accumulate=True)  # h += res, skip += skip_out
            x1 = x[:, 80:]
            ops.conv(blk["end"], cur[:, ATT:], x1, rag_s, aux=x1)  # (four fused layers end in `hs` again)
            ops.glow_invconv_actnorm(x, RS, 160, blk["winv"], blk["an_bias"], blk["an_logs"])
            if taps is not None and b in (17, 8, 0):
                taps[f"glow_z_after_block{b}"] = x.clone()
        return x.view(2 * RS, 80)  # unsqueeze == re-view (glow_utils.py:43-53)


class VocoderEngine:
    """BigVGAN (InferenceBigVGAN.py:72-95) / Avocodo-HiFiGAN generator (InferenceAvocodo.py:69-80) on packed mels."""

    UP = ((8, 16), (6, 12), (4, 8), (2, 4))
    KS = (3, 7, 11)
    DIL = (1, 3, 5)

    def __init__(self, state_dict, kind, device, bf16=False, fuse_snake=False, fuse_step=None, store_bf16=None, use_graphs=False,
                 precision=None, pack_only=False):
        assert kind in ("bigvgan", "hifigan")
        self.precision, bf16, compute16, self.dt16 = precision_of(bf16, precision)
        self.kind = kind
        self.use_graphs = use_graphs
        self._graphs = GraphCache(device)
        # BigVGAN: run the anti-aliased snake inside the convs' input staging (TTS_PRE_SNAKE, no extra HBM round trip) or as
        # its own kernel.  Measured on MI355X (batch 32, bf16): 122.5 ms/step fused vs 119.0 ms/step unfused - the fused
        # variant needs 111-131 VGPRs and loses occupancy, so the stand-alone kernel is the default for now.
        self.fuse_snake = fuse_snake
        # 16-bit modes only: one fused kernel per residual dilation step (tts_resblock_step) for the stages with C <= 128
        is16 = bf16 in ("bf16", "f16", True)  # ("x3": fp32 tensors, unfused path, every conv on the split fp32 product)
        self.fuse_step = is16 if fuse_step is None else bool(fuse_step and is16)
        # C = 256 (stage 1) is supported by the fused kernel too, but with one 4-wave workgroup per CU it only ties the
        # unfused convs + stand-alone snakes (9.1 vs 9.0 ms per step measured), so stage 1 keeps the unfused path
        self.fuse_max_channels = 128
        self.store_bf16 = self.fuse_step if store_bf16 is None else (store_bf16 and self.fuse_step)
        self.ops = None if pack_only else Ops(device)
        self.device = torch.device(device) if pack_only else self.ops.device
        self.compute = compute16
        dev = self.device
        sd = packing.fold_weight_norm(state_dict)
        if kind == "bigvgan":
            pre, ups, blk, c1, c2, post = "conv_pre", "ups.{}.0", "resblocks.{}.", "convs1.{}", "convs2.{}", "conv_post"
        else:
            pre, ups, blk, c1, c2, post = "input_conv", "upsamples.{}.1", "blocks.{}.", "convs1.{}.1", "convs2.{}.1", "output_conv.1"
        pc = packing.pack_conv
        self.pre = pc(sd[pre + ".weight"], sd[pre + ".bias"], dev, bf16=bf16)
        self.ups, self.blocks, self.snakes = [], [], []
        for i, (u, k) in enumerate(self.UP):
            self.ups.append(packing.pack_conv_transpose(sd[ups.format(i) + ".weight"], sd[ups.format(i) + ".bias"], u, dev, bf16=bf16))
            stage, stage_sn = [], []
            for j, kk in enumerate(self.KS):
                b = blk.format(3 * i + j)
                convs, sn = [], []
                for dd, dil in enumerate(self.DIL):
                    convs.append((pc(sd[b + c1.format(dd) + ".weight"], sd[b + c1.format(dd) + ".bias"], dev, dil=dil, bf16=bf16),
                                  pc(sd[b + c2.format(dd) + ".weight"], sd[b + c2.format(dd) + ".bias"], dev, dil=1, bf16=bf16)))
                    if kind == "bigvgan":
                        sn.append(tuple((_dev(sd[b + f"activations.{2 * dd + q}.act.alpha"], dev),
                                         _dev(sd[b + f"activations.{2 * dd + q}.act.beta"], dev)) for q in (0, 1)))
                stage.append(convs)
                stage_sn.append(sn)
            self.blocks.append(stage)
            self.snakes.append(stage_sn)
        pw = sd[post + ".weight"]  # [1, 32, 7]
        self.post_w = _dev(np.ascontiguousarray(pw[0].T), dev)  # [7][32]
        self.post_b = float(sd[post + ".bias"][0])
        if kind == "bigvgan":
            self.post_snake = (_dev(sd["activation_post.act.alpha"], dev), _dev(sd["activation_post.act.beta"], dev))
            stored = packing.stored_antialias_filter(sd)  # real checkpoints carry the filter as buffers; fixtures do not
            taps = packing.kaiser_sinc_filter12() if stored is None else stored
            self.filt = _dev(taps, dev)
            self.fir_tab = packing.snake_fir_table(taps, dev)  # the same filter as matrix-core operands (fused residual steps)

    @torch.inference_mode()
    def forward(self, mel_packed, rag, taps=None):
        """mel_packed [rows,80] time-major (utterance u at rag.begins[u], rag.lengths[u] frames) -> (wav packed, Ragged)."""
        if self.device.type == "cuda" and torch.cuda.current_device() != self.device.index:
            with torch.cuda.device(self.device):
                return self.forward(mel_packed, rag, taps)
        if self.use_graphs and taps is None and self.device.type == "cuda":
            key = (tuple(rag.lengths), tuple(rag.begins), int(mel_packed.shape[0]))
            wav = self._graphs.run(key, dict(mel_packed=mel_packed.contiguous()), lambda mel_packed: self._forward(mel_packed, rag, None)[0])
            return wav, rag.scaled(8).scaled(6).scaled(4).scaled(2)
        return self._forward(mel_packed, rag, taps)

    def _forward(self, mel_packed, rag, taps=None):
        ops, cp = self.ops, self.compute
        big = self.kind == "bigvgan"
        R = mel_packed.shape[0]
        x = ops.conv(self.pre, mel_packed, ops.empty(R, 512), rag, compute=cp)
        ch = 512
        for i, (u, k) in enumerate(self.UP):
            ch //= 2
            # transposed conv as a 3-tap polyphase conv; [R, u*ch] re-viewed as [R*u, ch]
            # stages that run the fused residual step keep their residual stream as bf16 in HBM (bandwidth-bound kernels)
            fused = self.fuse_step and ch <= self.fuse_max_channels
            sdt = self.dt16 if self.store_bf16 else torch.float32  # (the unfused C = 256 stage included: its convs and snakes take 16-bit I/O)
            y = ops.conv(self.ups[i], x, ops.empty(R, u * ch, dtype=sdt), rag, pre=PRE_NONE if big else PRE_LRELU, pre_slope=0.1, compute=cp)
            R, rag = R * u, rag.scaled(u)
            xs = y.view(R, ch)
            stage_out = ops.empty(R, ch, dtype=sdt)
            t1 = None if fused else ops.empty(R, ch, dtype=sdt)
            t2, sa = (ops.empty(R, ch, dtype=sdt), ops.empty(R, ch, dtype=sdt)) if (big and not self.fuse_snake and not fused) else (None, None)
            for j in range(3):
                cur = xs
                bufs = [ops.empty(R, ch, dtype=sdt), ops.empty(R, ch, dtype=sdt)]
                for dd in range(3):
                    c1, c2 = self.blocks[i][j][dd]
                    last = dd == 2
                    if fused:
                        # one launch per dilation step: act, conv(dil), act, conv(1), + x (and the stage mean on the last step)
                        sn1, sn2 = self.snakes[i][j][dd] if big else (None, None)
                        dst = stage_out if last else bufs[dd % 2]
                        ops.resblock_step(c1, c2, cur, dst, rag, PRE_SNAKE if big else PRE_LRELU, 0.1, sn1, sn2, self.filt if big else None,
                                          alpha=1.0 / 3.0 if last else 1.0, res_scale=1.0 / 3.0 if last else 1.0, accumulate=last and j > 0,
                                          fir_tab=self.fir_tab if big else None)
                        cur = dst
                        continue
                    if big:  # AMP.py:51-60: a1 -> c1 -> a2 -> c2 -> + x; both activations run inside the convs' input staging
                        (a1, b1), (a2, b2) = self.snakes[i][j][dd]
                        if self.fuse_snake:
                            ops.conv(c1, cur, t1, rag, pre=PRE_SNAKE, snake=(a1, b1, self.filt), compute=cp)
                            src2, pre2, sn2 = t1, PRE_SNAKE, (a2, b2, self.filt)
                        else:
                            ops.snake_aa(cur, sa, a1, b1, self.filt, ch, rag)
                            ops.conv(c1, sa, t1, rag, compute=cp)
                            ops.snake_aa(t1, t2, a2, b2, self.filt, ch, rag)
                            src2, pre2, sn2 = t2, PRE_NONE, None
                    else:  # ResidualBlock.py:83-98 with LeakyReLU(0.1)
                        ops.conv(c1, cur, t1, rag, pre=PRE_LRELU, pre_slope=0.1, compute=cp)
                        src2, pre2, sn2 = t1, PRE_LRELU, None
                    if not last:
                        nxt = bufs[dd % 2]
                        ops.conv(c2, src2, nxt, rag, pre=pre2, pre_slope=0.1, snake=sn2, res=cur, compute=cp)
                        cur = nxt
                    else:  # stage output = mean of the three blocks (InferenceBigVGAN.py:82-88)
                        ops.conv(c2, src2, stage_out, rag, pre=pre2, pre_slope=0.1, snake=sn2, alpha=1.0 / 3.0, res=cur,
                                 res_scale=1.0 / 3.0, accumulate=(j > 0), compute=cp)
            x = stage_out
            if taps is not None:
                taps[f"voc_stage{i}"] = x.float()
        wav = ops.empty(R)
        if big:
            # activation_post + conv_post + tanh in one launch: the activated [R, 32] tensor never reaches HBM
            ops.conv_post_snake(x, ch, self.post_w, self.post_b, *self.post_snake, self.filt, wav, rag)
        else:
            ops.conv_post(x, ch, self.post_w, self.post_b, PRE_LRELU, 0.01, wav, rag)  # InferenceAvocodo.py:53
        return wav, rag
