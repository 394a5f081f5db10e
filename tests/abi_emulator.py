"""TEST INFRASTRUCTURE ONLY: a numpy restatement of the C ABI in include/toucan_tts.h, operating on HOST
pointers.  It lets the `-m "not gpu"` suite drive the real host code (engine.py, packing.py, ragged.py,
interface, distributed sharding) end to end on CPU tensors and compare with the oracle, so that sequencing
and weight-layout bugs are caught without a GPU.  It doubles as the executable specification the GPU
kernel tests compare each HIP kernel against.

It is never imported by the package: tests install it by monkeypatching ``capi._LIB``.  The product path
loads libtoucan_hip.so and nothing else.
"""
import ctypes as C
import math

import numpy as np

from ims_toucan_prosody_variance_amd import capi


def _arr(ptr, n, dtype=np.float32):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    ct = {np.float32: C.c_float, np.int32: C.c_int32, np.uint16: C.c_uint16}[dtype]
    return np.ctypeslib.as_array((ct * int(n)).from_address(int(ptr)))


def _mat(ptr, rows, cols, ld, dtype=np.float32):
    """Strided [rows, cols] view on host memory."""
    if rows == 0:
        return np.zeros((0, cols), dtype=dtype)
    flat = _arr(ptr, (rows - 1) * ld + cols, dtype)
    return np.lib.stride_tricks.as_strided(flat, shape=(rows, cols), strides=(ld * flat.itemsize, flat.itemsize))


def _tiles(ptr, n):
    return _arr(ptr, 4 * n, np.int32).reshape(n, 4)


def _seqs_from_tiles(t):
    """unique (seq_begin, seq_end, seq_id) in order."""
    seen, out = set(), []
    for row0, sb, se, sid in t:
        if (sb, se, sid) not in seen:
            seen.add((sb, se, sid))
            out.append((int(sb), int(se), int(sid)))
    return out


def _bf16_round(a):
    """round-to-nearest-even to bfloat16, returned as float32."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def _round16(a, f16):
    """round-to-nearest-even to bf16 (or IEEE fp16 when `f16`), returned as float32."""
    if f16:
        return np.ascontiguousarray(a, dtype=np.float32).astype(np.float16).astype(np.float32)
    return _bf16_round(a)


def _widen16(raw, f16):
    raw = np.ascontiguousarray(raw)
    if f16:
        return raw.view(np.float16).astype(np.float32)
    return (raw.astype(np.uint32) << 16).view(np.float32)


def _fmt(is16, f16):
    """tensor format code: 0 fp32, 1 bf16, 2 fp16."""
    return 0 if not is16 else (2 if f16 else 1)


def _load(ptr, r0, r1, cols, ld, bf16=0):
    """rows [r0, r1) of a strided tensor as float32 (16-bit tensors are widened; bf16 = 0 fp32 / 1 bf16 / 2 fp16)."""
    if bf16:
        raw = _mat(ptr, r1, cols, ld, np.uint16)[r0:r1]
        return _widen16(raw, bf16 == 2).reshape(r1 - r0, cols)
    return _mat(ptr, r1, cols, ld)[r0:r1].astype(np.float32)


def _store(ptr, r0, r1, cols, ld, values, bf16=0):
    if bf16 == 2:
        _mat(ptr, r1, cols, ld, np.uint16)[r0:r1] = np.ascontiguousarray(values, dtype=np.float32).astype(np.float16).view(np.uint16)
    elif bf16:
        v = _bf16_round(np.ascontiguousarray(values, dtype=np.float32))
        _mat(ptr, r1, cols, ld, np.uint16)[r0:r1] = (v.view(np.uint32) >> 16).astype(np.uint16)
    else:
        _mat(ptr, r1, cols, ld)[r0:r1] = values


def _snake_seq(X, alpha, beta, filt):
    """Activation1d(SnakeBeta) on one utterance X [T, c] (fp64): replicate pad, 2x transposed conv, snake, 2x decimation."""
    T, c = X.shape
    f = filt.astype(np.float64)
    ea = np.exp(alpha.astype(np.float64))
    ib = 1.0 / (np.exp(beta.astype(np.float64)) + 1e-9)
    xp = np.concatenate([np.repeat(X[:1], 5, 0), X, np.repeat(X[-1:], 5, 0)], 0)  # replicate pad 5/5
    full = np.zeros((2 * (T + 10) + 10, c))
    for m in range(T + 10):  # conv_transpose1d stride 2
        full[2 * m:2 * m + 12] += xp[m][None, :] * f[:, None]
    u = 2.0 * full[15:15 + 2 * T]
    s = u + ib * np.sin(u * ea) ** 2
    sp = np.concatenate([np.repeat(s[:1], 5, 0), s, np.repeat(s[-1:], 6, 0)], 0)
    return np.stack([(sp[2 * t:2 * t + 12] * f[:, None]).sum(0) for t in range(T)], 0)


class Emulator:
    def __init__(self):
        self._real = None
        self._err = b""
        self.calls = {}

    def _count(self, name):
        self.calls[name] = self.calls.get(name, 0) + 1

    # host-only helpers come from the real library (they do not touch a GPU)
    def _reallib(self):
        if self._real is None:
            self._real = C.CDLL(capi.LIB_PATH)
        return self._real

    def tts_last_error(self):
        return self._err

    def tts_abi_version(self):
        return capi.ABI_VERSION

    def tts_diag_queue_nonzero(self):
        return 0  # (the emulator has no work queues)

    def tts_diag_queue_slots_used(self):
        return 0

    def tts_conv1d_tile_rows(self, cout, mode):
        return self._reallib().tts_conv1d_tile_rows(cout, mode)

    def tts_conv1d_small_tile_rows(self, cout, mode, cols):
        return self._reallib().tts_conv1d_small_tile_rows(cout, mode, cols)

    def tts_conv1d_n_tile(self, cout, mode):
        return self._reallib().tts_conv1d_n_tile(cout, mode)

    # ------------------------------------------------------------------------------------------
    def tts_conv1d(self, dref, stream):
        self._count("conv1d")
        d = dref._obj
        t = _tiles(d.tiles, d.n_tiles)
        dual = d.mode != capi.MODE_LINEAR
        f16 = bool(d.io_flags & capi.IO_F16) or d.compute == capi.COMPUTE_F16
        fx, fy, fr = (_fmt(d.io_flags & b, f16) for b in (capi.IO_X_BF16, capi.IO_Y_BF16, capi.IO_RES_BF16))
        if d.compute == capi.COMPUTE_F32X3:  # two fp16 planes [hi | lo']: w = hi + 2^-11 lo' (what the three products add up to)
            wraw = _arr(d.w, 2 * d.taps * d.cin_pad * d.wn, np.uint16).reshape(2, d.taps, d.cin_pad // 8, d.wn, 8)
            planes = [_widen16(wraw[i], True).reshape(wraw[i].shape).transpose(0, 1, 3, 2).reshape(d.taps, d.cin_pad, d.wn).astype(np.float64) for i in (0, 1)]
            w = planes[0] + planes[1] / 2048.0
        elif d.compute != capi.COMPUTE_F32:
            wraw = _arr(d.w, d.taps * d.cin_pad * d.wn, np.uint16).reshape(d.taps, d.cin_pad // 8, d.wn, 8)
            w = _widen16(wraw, d.compute == capi.COMPUTE_F16).reshape(wraw.shape).transpose(0, 1, 3, 2).reshape(d.taps, d.cin_pad, d.wn)
        else:
            w = _arr(d.w, d.taps * d.cin_pad * d.wn).reshape(d.taps, d.cin_pad, d.wn)
        bias = _arr(d.bias, d.cout * (2 if dual else 1)) if d.bias else None
        for sb, se, sid in _seqs_from_tiles(t):
            n = se - sb
            x = _load(d.x, sb, se, d.cin, d.ldx, fx)
            if d.pre_act == capi.PRE_LRELU:
                x = np.where(x > 0, x, x * np.float32(d.pre_slope))
            elif d.pre_act == capi.PRE_SNAKE:
                x = _snake_seq(x.astype(np.float64), _arr(d.snake_alpha, d.cin), _arr(d.snake_beta, d.cin),
                               _arr(d.snake_filt, 12)).astype(np.float32)
            if d.compute == capi.COMPUTE_F32X3:  # x = hi + 2^-11 lo' with the same two fp16 roundings as the kernel
                hi = _round16(x, True)
                x = hi.astype(np.float64) + _round16(((x - hi) * np.float32(2048.0)).astype(np.float32), True).astype(np.float64) / 2048.0
            elif d.compute != capi.COMPUTE_F32:
                x = _round16(x, d.compute == capi.COMPUTE_F16)
            halo = (d.taps - 1) * d.dil
            xp = np.zeros((n + halo, d.cin), dtype=np.float64 if d.compute == capi.COMPUTE_F32X3 else np.float32)
            xp[d.pad_left:d.pad_left + n] = x
            acc = np.zeros((n, d.wn), dtype=np.float64)
            for j in range(d.taps):
                acc += xp[j * d.dil:j * d.dil + n].astype(np.float64) @ w[j, :d.cin].astype(np.float64)
            acc = acc.astype(np.float32)
            a = acc[:, :d.cout]
            if bias is not None:
                a = a + bias[:d.cout]
            if d.seqvec:
                a = a + _mat(d.seqvec, sid + 1, d.cout, d.ld_seqvec)[sid]
            if d.preadd:
                a = a + _mat(d.preadd, se, d.cout, d.ld_preadd)[sb:se]
            if dual:
                g = acc[:, d.half_pad:d.half_pad + d.cout]
                if bias is not None:
                    g = g + bias[d.cout:]
                if d.preadd:
                    g = g + _mat(d.preadd + 4 * d.cout, se, d.cout, d.ld_preadd)[sb:se]
                sig = 1.0 / (1.0 + np.exp(-g.astype(np.float64)))
                if d.mode == capi.MODE_GLU:
                    v = a * sig
                elif d.mode == capi.MODE_GATED:
                    v = np.tanh(a.astype(np.float64)) * sig
                else:
                    v = (_mat(d.aux, se, d.cout, d.ld_aux)[sb:se] - a) * np.exp(-g.astype(np.float64))
            else:
                v = a
                if d.act == capi.ACT_RELU:
                    v = np.maximum(v, 0)
                elif d.act == capi.ACT_TANH:
                    v = np.tanh(v.astype(np.float64))
            v = (np.asarray(v, dtype=np.float32) * np.float32(d.alpha)).astype(np.float32)
            if d.res:
                v = v + np.float32(d.res_scale) * _load(d.res, sb, se, d.cout, d.ld_res, fr)
            if d.accumulate:
                v = v + _load(d.y, sb, se, d.cout, d.ldy, fy)
            _store(d.y, sb, se, d.cout, d.ldy, v, fy)
        return 0

    def tts_wavenet_layer(self, dref, stream):
        """16-bit rounding points as in csrc/wavenet.hip: h -> 16 bit; acts -> 16 bit; fp32 accumulation and epilogues."""
        self._count("wavenet_layer")
        d = dref._obj
        f16 = d.compute == capi.COMPUTE_F16
        H, n2 = 192, d.cout2
        w1 = _widen16(_arr(d.w1, 5 * H * 384, np.uint16), f16).reshape(5, H // 8, 384, 8).transpose(0, 1, 3, 2).reshape(5, H, 384).astype(np.float64)
        w2 = _widen16(_arr(d.w2, H * n2, np.uint16), f16).reshape(H // 8, n2, 8).transpose(0, 2, 1).reshape(H, n2).astype(np.float64)
        b1, b2 = _arr(d.b1, 384).astype(np.float32), _arr(d.b2, n2).astype(np.float32)
        col0 = 0 if n2 == 384 else H
        for sb, se, sid in _seqs_from_tiles(_tiles(d.tiles, d.n_tiles)):
            n = se - sb
            hs = _mat(d.hs_in, se, 384, d.ld_in)[sb:se].astype(np.float32)
            h = _round16(hs[:, :H], f16).astype(np.float64)
            hp = np.zeros((n + 4, H))
            hp[2:2 + n] = h
            acc = sum(hp[j:j + n] @ w1[j] for j in range(5)).astype(np.float32)
            cond = _mat(d.cond, se, 384, d.ld_cond)[sb:se]
            a = (acc[:, :H] + b1[:H]) + cond[:, :H]
            g = (acc[:, H:] + b1[H:]) + cond[:, H:]
            acts = _round16((np.tanh(a.astype(np.float64)) / (1.0 + np.exp(-g.astype(np.float64)))).astype(np.float32), f16).astype(np.float64)
            out = ((acts @ w2).astype(np.float32) + b2) + hs[:, col0:col0 + n2]
            _mat(d.hs_out, se, 384, d.ld_out)[sb:se, col0:col0 + n2] = out
        return 0

    def tts_ffn_fused(self, dref, stream):
        """Rounding points as in csrc/ffn.hip: LN(x) -> 16 bit; relu(W1 . + b1) -> 16 bit; fp32 accumulation, statistics and epilogue.
        The weights arrive in the kernel's fragment order (include/toucan_tts.h) and are unpacked here."""
        self._count("ffn_fused")
        d = dref._obj
        f16 = d.compute == capi.COMPUTE_F16
        Cc, Hh = 192, d.hidden
        n = Hh // 32
        raw = _arr(d.w, n * 14 * 1024, np.uint16).reshape(n, 14 * 1024)
        frag = _widen16(raw[:, :24 * 512].reshape(-1), f16).reshape(n, 24, 64, 8)
        bias = raw[:, 24 * 512:].copy().view(np.float32).reshape(n, 4, 64, 4)
        lane = np.arange(64)
        lk, r = lane // 32, lane % 32
        w1 = np.zeros((Hh, Cc), np.float64)
        w2 = np.zeros((Cc, Hh), np.float64)
        b1 = np.zeros(Hh, np.float32)
        for c in range(n):
            for ks in range(12):
                for i in range(8):
                    w1[32 * c + r, 16 * ks + 8 * lk + i] = frag[c, ks, lane, i]
            for j in range(6):
                for ab in range(2):
                    for i in range(8):
                        slot = 4 * lk + i if i < 4 else 8 + 4 * lk + i - 4
                        w2[32 * j + r, 32 * c + 16 * ab + slot] = frag[c, 12 + 2 * j + ab, lane, i]
            for q in range(4):
                for i in range(4):
                    b1[32 * c + 8 * q + 4 * lk + i] = bias[c, q, lane, i]
        x = _mat(d.x, d.rows, Cc, d.ldx)[:d.rows].astype(np.float32)
        ln = lambda v, g, b: (((v - v.mean(1, keepdims=True, dtype=np.float64)) / np.sqrt(v.astype(np.float64).var(1, keepdims=True) + d.eps))
                              * _arr(g, Cc).astype(np.float64) + _arr(b, Cc).astype(np.float64)).astype(np.float32)
        xn = _round16(ln(x, d.ln_g, d.ln_b), f16).astype(np.float64)
        h = _round16(np.maximum((xn @ w1.T).astype(np.float32) + b1, 0.0), f16).astype(np.float64)
        out = x + np.float32(d.alpha) * ((h @ w2.T).astype(np.float32) + _arr(d.b2, Cc).astype(np.float32))
        if d.post_g:
            out = ln(out, d.post_g, d.post_b)
        _mat(d.y, d.rows, Cc, d.ldy)[:d.rows] = out
        return 0

    def tts_snake_fir_table(self, filt, table):
        return self._reallib().tts_snake_fir_table(filt, table)  # host-only arithmetic: the real library's (the emulator ignores the table)

    def tts_resblock_tile_rows(self, c):
        return self._reallib().tts_resblock_tile_rows(c)

    def tts_resblock_step(self, dref, stream):
        """bf16 rounding points as in csrc/resblock.hip: act1(x) -> bf16; conv1 (+b1) -> bf16 (LeakyReLU before the
        rounding, snake after it) -> act2 -> bf16; conv2; fp32 epilogue."""
        self._count("resblock_step")
        d = dref._obj
        C_, k = d.c, d.taps

        f16 = d.compute == capi.COMPUTE_F16
        io = _fmt(d.io_bf16, f16)
        rnd = lambda a: _round16(a, f16)

        def wload(ptr):
            raw = _arr(ptr, k * C_ * C_, np.uint16).reshape(k, C_ // 8, C_, 8)
            return _widen16(raw, f16).reshape(raw.shape).transpose(0, 1, 3, 2).reshape(k, C_, C_).astype(np.float64)

        w1, w2 = wload(d.w1), wload(d.w2)
        b1, b2 = _arr(d.b1, C_).astype(np.float64), _arr(d.b2, C_).astype(np.float64)
        snake = d.act == capi.PRE_SNAKE
        filt = _arr(d.filt, 12) if snake else None

        def act(v, al, be):
            if snake:
                return _snake_seq(v, _arr(al, C_), _arr(be, C_), filt)
            return np.where(v > 0, v, v * np.float64(np.float32(d.slope)))

        def conv(v, w, dil):
            n = v.shape[0]
            h = (k - 1) // 2 * dil
            vp = np.zeros((n + 2 * h, C_))
            vp[h:h + n] = v
            return sum(vp[j * dil:j * dil + n] @ w[j] for j in range(k))

        for sb, se, sid in _seqs_from_tiles(_tiles(d.tiles, d.n_tiles)):
            x = _load(d.x, sb, se, C_, d.ldx, io).astype(np.float64)
            a1 = rnd(act(x, d.alpha1, d.beta1).astype(np.float32)).astype(np.float64)
            t = conv(a1, w1, d.dil) + b1
            if snake:
                t = rnd(t.astype(np.float32)).astype(np.float64)
                a2 = rnd(act(t, d.alpha2, d.beta2).astype(np.float32)).astype(np.float64)
            else:
                a2 = rnd(act(t, None, None).astype(np.float32)).astype(np.float64)
            v = np.float32(d.alpha) * (conv(a2, w2, 1) + b2).astype(np.float32) + np.float32(d.res_scale) * x.astype(np.float32)
            if d.accumulate:
                v = v + _load(d.y, sb, se, C_, d.ldy, io)
            _store(d.y, sb, se, C_, d.ldy, v.astype(np.float32), io)
        return 0

    # ---- per-speaker path (csrc/style.hip) -------------------------------------------------------------------
    def tts_gru_layer(self, x, ldx, batch, steps, in_dim, hidden, w_ih_t, w_hh_t, b_ih, b_hh, y, ldy, stream):
        self._count("gru_layer")
        X = _mat(x, batch * steps, in_dim, ldx).astype(np.float64)
        Wi = _arr(w_ih_t, in_dim * 3 * hidden).reshape(in_dim, 3 * hidden).astype(np.float64)
        Wh = _arr(w_hh_t, hidden * 3 * hidden).reshape(hidden, 3 * hidden).astype(np.float64)
        bi, bh = _arr(b_ih, 3 * hidden).astype(np.float64), _arr(b_hh, 3 * hidden).astype(np.float64)
        Y = _mat(y, batch * steps, hidden, ldy)
        H = hidden
        sig = lambda v: 1.0 / (1.0 + np.exp(-v))
        for b in range(batch):
            h = np.zeros(H)
            for t in range(steps):
                gi, gh = X[b * steps + t] @ Wi + bi, h @ Wh + bh
                r, z = sig(gi[:H] + gh[:H]), sig(gi[H:2 * H] + gh[H:2 * H])
                n = np.tanh(gi[2 * H:] + r * gh[2 * H:])
                h = (1 - z) * n + z * h
                Y[b * steps + t] = h.astype(np.float32)
        return 0

    def tts_style_tokens(self, q, k, v, batch, n_tokens, heads, dk, ctx, stream):
        self._count("style_tokens")
        ld = heads * dk
        Q = _mat(q, batch, ld, ld).astype(np.float64).reshape(batch, heads, dk)
        K = _mat(k, n_tokens, ld, ld).astype(np.float64).reshape(n_tokens, heads, dk)
        V = _mat(v, n_tokens, ld, ld).astype(np.float64).reshape(n_tokens, heads, dk)
        s = np.einsum("bhd,nhd->bhn", Q, K) / math.sqrt(dk)
        p = np.exp(s - s.max(-1, keepdims=True))
        p /= p.sum(-1, keepdims=True)
        _mat(ctx, batch, ld, ld)[:] = np.einsum("bhn,nhd->bhd", p, V).reshape(batch, ld).astype(np.float32)
        return 0

    def tts_complex_magnitude(self, x, ldx, y, ldy, rows, bins, stream):
        self._count("complex_magnitude")
        X = _mat(x, rows, 2 * bins, ldx).astype(np.float64)
        _mat(y, rows, bins, ldy)[:] = np.sqrt(X[:, :bins] ** 2 + X[:, bins:] ** 2).astype(np.float32)
        return 0

    def tts_log10_floor(self, x, ldx, y, ldy, rows, c, eps, stream):
        self._count("log10_floor")
        _mat(y, rows, c, ldy)[:] = np.log10(np.maximum(np.float32(eps), _mat(x, rows, c, ldx))).astype(np.float32)
        return 0

    def tts_layernorm(self, x, ldx, y, ldy, gamma, beta, rows, c, eps, stream):
        self._count("layernorm")
        X = _mat(x, rows, c, ldx).astype(np.float64)
        m = X.mean(1, keepdims=True)
        v = ((X - m) ** 2).mean(1, keepdims=True)
        out = (X - m) / np.sqrt(v + eps) * _arr(gamma, c) + _arr(beta, c)
        _mat(y, rows, c, ldy)[:] = out.astype(np.float32)
        return 0

    def tts_cond_layernorm(self, x, ldx, y, ldy, scale, shift, c, tiles, n_tiles, tile_rows, stream):
        self._count("cond_layernorm")
        for sb, se, sid in _seqs_from_tiles(_tiles(tiles, n_tiles)):
            X = _mat(x, se, c, ldx)[sb:se].astype(np.float64)
            m = X.mean(1, keepdims=True)
            v = ((X - m) ** 2).mean(1, keepdims=True)
            sc = _mat(scale, sid + 1, c, c)[sid]
            sh = _mat(shift, sid + 1, c, c)[sid]
            _mat(y, se, c, ldy)[sb:se] = (sc * ((X - m) / v) + sh).astype(np.float32)
        return 0

    def tts_cln_mlp_weight_floats(self, d_in, d_out):
        return d_in * d_in + d_in + d_in * d_out + d_out + d_out * d_out + d_out

    def tts_cln_mlp(self, e, n_seq, d_in, d_out, weights, n_mlp, out, stream):
        self._count("cln_mlp")
        per = self.tts_cln_mlp_weight_floats(d_in, d_out)
        E = _mat(e, n_seq, d_in, d_in).astype(np.float64)
        W = _arr(weights, n_mlp * per).astype(np.float64).reshape(n_mlp, per)
        O = _arr(out, n_mlp * n_seq * d_out).reshape(n_mlp, n_seq, d_out)
        for m in range(n_mlp):
            o = 0
            w0 = W[m, o:o + d_in * d_in].reshape(d_in, d_in); o += d_in * d_in
            b0 = W[m, o:o + d_in]; o += d_in
            w1 = W[m, o:o + d_in * d_out].reshape(d_in, d_out); o += d_in * d_out
            b1 = W[m, o:o + d_out]; o += d_out
            w2 = W[m, o:o + d_out * d_out].reshape(d_out, d_out); o += d_out * d_out
            b2 = W[m, o:o + d_out]
            O[m] = (np.tanh(np.tanh(E @ w0 + b0) @ w1 + b1) @ w2 + b2).astype(np.float32)
        return 0

    def tts_l2_normalize(self, x, y, rows, c, stream):
        self._count("l2_normalize")
        X = _mat(x, rows, c, c).astype(np.float64)
        n = np.maximum(np.sqrt((X ** 2).sum(1, keepdims=True)), 1e-12)
        _mat(y, rows, c, c)[:] = (X / n).astype(np.float32)
        return 0

    def tts_groupnorm_workspace_floats(self, n_seq, max_len, groups):
        return n_seq * ((max_len + 15) // 16) * groups * 2

    def tts_groupnorm(self, x, ldx, y, ldy, gamma, beta, c, groups, eps, apply_tanh, res, ld_res, seq_begin, seq_end, n_seq, max_len,
                      workspace, stream):
        self._count("groupnorm")
        sb, se = _arr(seq_begin, n_seq, np.int32), _arr(seq_end, n_seq, np.int32)
        g, b = _arr(gamma, c).astype(np.float64), _arr(beta, c).astype(np.float64)
        for u in range(n_seq):
            r0, r1 = int(sb[u]), int(se[u])
            X = _mat(x, r1, c, ldx)[r0:r1].astype(np.float64)
            Xg = X.reshape(r1 - r0, groups, c // groups)
            m = Xg.mean(axis=(0, 2), keepdims=True)
            v = ((Xg - m) ** 2).mean(axis=(0, 2), keepdims=True)
            out = ((Xg - m) / np.sqrt(v + eps)).reshape(r1 - r0, c) * g + b
            if apply_tanh:
                out = np.tanh(out)
            if res:
                out = out + _mat(res, r1, c, ld_res)[r0:r1]
            _mat(y, r1, c, ldy)[r0:r1] = out.astype(np.float32)
        return 0

    def tts_relpos_attention_f16(self, qkv, ld_qkv, ptab, pmax, bias_u, bias_v, ctx, ld_ctx, heads, dk, tiles, n_tiles, tile_rows, stream):
        """fp16 rounding points as in relpos_attention_f16_kernel: q + u, q + v, k, v, the table -> fp16; the probabilities
        exp(s - max) -> fp16 before the product with v (the normaliser sums the unrounded ones)."""
        return self.tts_relpos_attention(qkv, ld_qkv, ptab, pmax, bias_u, bias_v, ctx, ld_ctx, heads, dk, tiles, n_tiles, tile_rows, 0, stream, f16=True)

    def tts_relpos_attention(self, qkv, ld_qkv, ptab, pmax, bias_u, bias_v, ctx, ld_ctx, heads, dk, tiles, n_tiles, tile_rows, flags, stream, f16=False):
        """(flags: the key-split forms differ from the plain one in rounding order only - float64 here either way)"""
        self._count("relpos_attention")
        assert tile_rows == 128 and 0 <= flags <= 3
        hd = heads * dk
        r16 = (lambda a: _round16(np.asarray(a, dtype=np.float32), True).astype(np.float64)) if f16 else (lambda a: a)
        P = r16(_mat(ptab, 2 * pmax - 1, hd, hd).astype(np.float64))
        u = _arr(bias_u, hd).astype(np.float64)
        v = _arr(bias_v, hd).astype(np.float64)
        for sb, se, sid in _seqs_from_tiles(_tiles(tiles, n_tiles)):
            n = se - sb
            assert n <= pmax
            Q = _mat(qkv, se, 3 * hd, ld_qkv)[sb:se].astype(np.float64)
            q, k, val = Q[:, :hd], Q[:, hd:2 * hd], Q[:, 2 * hd:]
            rel = np.arange(n)[:, None] - np.arange(n)[None, :] + (pmax - 1)  # table row of p = i - j
            out = np.zeros((n, hd))
            for h in range(heads):
                sl = slice(h * dk, (h + 1) * dk)
                ac = r16((q[:, sl] + u[sl]).astype(np.float32)) @ r16(k[:, sl]).T
                bd = np.einsum("id,ijd->ij", r16((q[:, sl] + v[sl]).astype(np.float32)), P[rel][:, :, sl])
                s = (ac + bd) / math.sqrt(dk)
                s = np.exp(s - s.max(1, keepdims=True))
                out[:, sl] = (r16(s) @ r16(val[:, sl])) / s.sum(1, keepdims=True)
            _mat(ctx, se, hd, ld_ctx)[sb:se] = out.astype(np.float32)
        return 0

    def tts_dwconv_swish(self, x, ldx, y, ldy, w, b, c, k, tiles, n_tiles, tile_rows, stream):
        self._count("dwconv_swish")
        W = _mat(w, k, c, c).astype(np.float64)
        B = _arr(b, c).astype(np.float64)
        h = (k - 1) // 2
        for sb, se, sid in _seqs_from_tiles(_tiles(tiles, n_tiles)):
            n = se - sb
            xp = np.zeros((n + k - 1, c))
            xp[h:h + n] = _mat(x, se, c, ldx)[sb:se]
            a = B + sum(xp[j:j + n] * W[j] for j in range(k))
            _mat(y, se, c, ldy)[sb:se] = (a / (1.0 + np.exp(-a))).astype(np.float32)
        return 0

    def tts_duration_from_log(self, logd, dur, n, stream):
        self._count("duration_from_log")
        v = np.rint(np.exp(_arr(logd, n)).astype(np.float32) - np.float32(1.0))
        _arr(dur, n, np.int32)[:] = np.clip(v, 0, 1e6).astype(np.int32)
        return 0

    def tts_prosody_control(self, text, ld_text, pitch, energy, dur, seq_begin, seq_end, n_seq, ds, ps, es, pause, stream):
        self._count("prosody_control")
        sb, se = _arr(seq_begin, n_seq, np.int32), _arr(seq_end, n_seq, np.int32)
        for u in range(n_seq):
            r0, r1 = int(sb[u]), int(se[u])
            f = _mat(text, r1, 62, ld_text)[r0:r1]
            p, e, d = _arr(pitch, r1)[r0:r1], _arr(energy, r1)[r0:r1], _arr(dur, r1, np.int32)[r0:r1]
            p[f[:, 61] == 0] = 0
            e[f[:, 15] == 0] = 0
            d[f[:, 21] == 1] = 0
            if pause != 1.0:
                m = f[:, 16] == 1
                d[m] = np.rint(d[m].astype(np.float32) * np.float32(pause)).astype(np.int32)
            if ds != 1.0:
                d[:] = np.rint(d.astype(np.float32) * np.float32(ds)).astype(np.int32)
            for seq, sc in ((p, ps), (e, es)):
                if sc != 1.0:
                    avg = seq[seq != 0].mean(dtype=np.float32) if (seq != 0).any() else np.float32(np.nan)
                    v = ((seq - avg) * np.float32(sc)) + avg
                    seq[:] = np.where(v < 0, 0, v)
        return 0

    def tts_length_regulate(self, enc, ld_enc, pitch, energy, wp, bp, we, be, dur, phone_begin, phone_end, frame_begin, n_seq,
                            max_frames, max_phones, c, up, ld_up, dec_in, ld_dec, dec_scale, stream):
        self._count("length_regulate")
        pb, pe, fb = (_arr(a, n_seq, np.int32) for a in (phone_begin, phone_end, frame_begin))
        Wp, Bp, We, Be = (_arr(a, c) for a in (wp, bp, we, be))
        for u in range(n_seq):
            p0, p1 = int(pb[u]), int(pe[u])
            d = _arr(dur, p1, np.int32)[p0:p1].copy()
            if d.sum() == 0:
                d[:] = 1
            src = np.repeat(np.arange(p0, p1), d)
            E = _mat(enc, p1, c, ld_enc)
            P, En = _arr(pitch, p1), _arr(energy, p1)
            v = E[src] + (P[src, None] * Wp + Bp) + (En[src, None] * We + Be)
            T = len(src)
            f0 = int(fb[u])
            _mat(up, f0 + T, c, ld_up)[f0:f0 + T] = v
            if dec_in:
                _mat(dec_in, f0 + T, c, ld_dec)[f0:f0 + T] = v * np.float32(dec_scale)
        return 0

    def tts_glow_invconv_actnorm(self, x, ldx, rows, c, winv, an_bias, an_logs, stream):
        self._count("glow_invconv_actnorm")
        X = _mat(x, rows, c, ldx)
        W = _arr(winv, 16).reshape(4, 4).astype(np.float64)
        v = X.astype(np.float64).reshape(rows, 2, c // 4, 2).transpose(0, 1, 3, 2).reshape(rows, 4, c // 4)  # [r, n=(a,r), g]
        z = np.einsum("on,rng->rog", W, v)
        z = z.reshape(rows, 2, 2, c // 4).transpose(0, 1, 3, 2).reshape(rows, c)
        X[:] = ((z - _arr(an_bias, c)) * np.exp(-_arr(an_logs, c).astype(np.float64))).astype(np.float32)
        return 0

    def tts_snake_aa(self, x, ldx, y, ldy, alpha, beta, filt, c, tiles, n_tiles, tile_rows, io_flags, stream):
        self._count("snake_aa")
        for sb, se, sid in _seqs_from_tiles(_tiles(tiles, n_tiles)):
            X = _load(x, sb, se, c, ldx, _fmt(io_flags & capi.IO_X_BF16, io_flags & capi.IO_F16)).astype(np.float64)
            out = _snake_seq(X, _arr(alpha, c), _arr(beta, c), _arr(filt, 12)).astype(np.float32)
            _store(y, sb, se, c, ldy, out, _fmt(io_flags & capi.IO_Y_BF16, io_flags & capi.IO_F16))
        return 0

    def tts_conv_post(self, x, ldx, cin, w, bias, pre_act, pre_slope, wav, tiles, n_tiles, tile_rows, io_flags, stream):
        self._count("conv_post")
        W = _mat(w, 7, cin, cin).astype(np.float64)
        for sb, se, sid in _seqs_from_tiles(_tiles(tiles, n_tiles)):
            n = se - sb
            X = _load(x, sb, se, cin, ldx, _fmt(io_flags & capi.IO_X_BF16, io_flags & capi.IO_F16)).astype(np.float64)
            if pre_act == capi.PRE_LRELU:
                X = np.where(X > 0, X, X * pre_slope)
            xp = np.zeros((n + 6, cin))
            xp[3:3 + n] = X
            a = bias + sum((xp[j:j + n] * W[j]).sum(1) for j in range(7))
            _arr(wav, se)[sb:se] = np.tanh(a).astype(np.float32)
        return 0

    def tts_conv_post_snake_tile_rows(self):
        return 250

    def tts_conv_post_snake(self, x, ldx, cin, w, bias, alpha, beta, filt, wav, tiles, n_tiles, tile_rows, io_flags, stream):
        self._count("conv_post_snake")
        W = _mat(w, 7, cin, cin).astype(np.float64)
        for sb, se, sid in _seqs_from_tiles(_tiles(tiles, n_tiles)):
            n = se - sb
            X = _load(x, sb, se, cin, ldx, _fmt(io_flags & capi.IO_X_BF16, io_flags & capi.IO_F16)).astype(np.float64)
            X = _snake_seq(X, _arr(alpha, cin), _arr(beta, cin), _arr(filt, 12))
            xp = np.zeros((n + 6, cin))
            xp[3:3 + n] = X
            a = bias + sum((xp[j:j + n] * W[j]).sum(1) for j in range(7))
            _arr(wav, se)[sb:se] = np.tanh(a).astype(np.float32)
        return 0

    def tts_gather_rows(self, src, ld_src, idx, dst, ld_dst, n, c, stream):
        self._count("gather_rows")
        ii = _arr(idx, n, np.int32)
        S = _mat(src, int(ii.max()) + 1, c, ld_src)
        _mat(dst, n, c, ld_dst)[:] = S[ii]
        return 0



def install(monkeypatch=None):
    """Replace the loaded library by the emulator (tests only). Returns the emulator."""
    emu = Emulator()
    if monkeypatch is not None:
        monkeypatch.setattr(capi, "_LIB", emu)
    else:
        capi._LIB = emu
    return emu
