"""Weight preparation: reference-schema state_dicts -> kernel-friendly device layouts.

Done once at load time, the counterpart of the reference's ``store_inverse_all``
(InferenceToucanTTS.py:321-330), ``remove_weight_norm`` (InferenceBigVGAN.py:97-105,
InferenceAvocodo.py:82-89) and ``InvConvNear.store_inverse`` (Glow.py:130-139):

* weight-norm folded: w = g * v / ||v||
* BatchNorm(eval) of the Conformer conv module folded into the depthwise conv (Convolution.py:26-27)
* the 18 LU-parametrised 4x4 flow matrices inverted (fp64 on the host, stored fp32)
* Conv1d / Linear weights re-laid as [tap][cin_pad][cout_pad] (the B operand of the implicit GEMM),
  optionally also as bf16 or fp16 [tap][cin_pad/8][cout_pad][8]
* ConvTranspose1d (k = 2*stride, padding = stride/2) rewritten as a 3-tap polyphase conv whose output
  row holds the `stride` output samples of one input frame
* q/k/v projections concatenated into one [192 -> 576] GEMM; GLU / gated / coupling convs split in halves

This is host-side layout work on weights (numpy), not inference arithmetic.
"""
import math

import numpy as np
import torch

from . import capi


def _np(t):
    if isinstance(t, np.ndarray):
        return t
    return t.detach().cpu().numpy()


def fold_weight_norm(sd):
    """{..weight_g, ..weight_v} -> {..weight}; other entries pass through (numpy arrays, fp32)."""
    out = {}
    for k, v in sd.items():
        if k.endswith(".weight_g"):
            base = k[: -len(".weight_g")]
            vv = _np(sd[base + ".weight_v"]).astype(np.float64)
            g = _np(v).astype(np.float64)
            nrm = np.sqrt((vv.reshape(vv.shape[0], -1) ** 2).sum(axis=1)).reshape([-1] + [1] * (vv.ndim - 1))
            out[base + ".weight"] = (vv * (g / nrm)).astype(np.float32)
        elif k.endswith(".weight_v"):
            continue
        else:
            out[k] = _np(v)
    return out


def _roundup(a, b):
    return (a + b - 1) // b * b


class ConvWeights:
    """One packed conv: device tensors + the static fields of TtsConvDesc."""

    def __init__(self, w_kio, bias, mode, dil, pad_left, device, bf16=False, small_only=False):
        # w_kio: numpy [taps, cin, cout_total] (dual modes: cout_total = 2*cout, halves [a | g])
        lib = capi.lib()
        taps, cin, ctot = w_kio.shape
        self.mode = mode
        self.taps, self.dil, self.pad_left = int(taps), int(dil), int(pad_left)
        self.algo_taps = int(taps)  # taps of the reference op (a transposed conv packed as 3 taps only has 2)
        self.cin = int(cin)
        self.cin_pad = _roundup(cin, 32)
        dual = mode != capi.MODE_LINEAR
        self.cout = ctot // 2 if dual else ctot
        self.n_tile = lib.tts_conv1d_n_tile(self.cout, mode)
        self.tile_rows = lib.tts_conv1d_tile_rows(self.cout, mode)
        # small_only: pad the columns to the 64-column small-batch form instead of the regular N tile and always run that form
        # (narrow dual-mode convs such as the 80-channel coupling output would otherwise get one 128 x 96 workgroup per 128 rows)
        self.small_only = bool(small_only)
        if self.small_only:
            self.n_tile = 64
        half = _roundup(self.cout, self.n_tile)
        self.half_pad = half if dual else 0
        self.wn = 2 * half if dual else half
        self.small_tile_rows = lib.tts_conv1d_small_tile_rows(self.cout, mode, half)  # 64 or 0 (small-batch form)
        if self.small_only:
            assert self.small_tile_rows == 64, "this shape has no small-batch form"
            self.tile_rows = 64
        packed = np.zeros((taps, self.cin_pad, self.wn), dtype=np.float32)
        if dual:
            packed[:, :cin, : self.cout] = w_kio[:, :, : self.cout]
            packed[:, :cin, half: half + self.cout] = w_kio[:, :, self.cout:]
        else:
            packed[:, :cin, : self.cout] = w_kio
        self.w = torch.from_numpy(packed).to(device)
        # optional 16-bit copy for the bf16 / fp16 MFMA paths: `bf16` is False, True / "bf16", "f16" or "x3" (one format per conv)
        self.w16 = None
        self.compute16 = capi.COMPUTE_F32
        if bf16 == "x3":
            # the split fp32 product (TTS_COMPUTE_F32X3): two fp16 planes [hi | lo'], hi = fp16(w), lo' = fp16((w - hi) 2^11), each in
            # the 16-bit fragment layout below
            self.compute16 = capi.COMPUTE_F32X3
            t32 = torch.from_numpy(packed)
            hi = t32.to(torch.float16)
            lo = ((t32 - hi.to(torch.float32)) * 2048.0).to(torch.float16)
            frag = lambda t: t.reshape(taps, self.cin_pad // 8, 8, self.wn).permute(0, 1, 3, 2).contiguous()
            self.w16 = torch.cat([frag(hi), frag(lo)], dim=0).contiguous().to(device)  # [2 taps][cin_pad/8][wn][8]: the hi plane, then the lo' plane
        elif bf16:
            dt = torch.float16 if bf16 == "f16" else torch.bfloat16
            self.compute16 = capi.COMPUTE_F16 if bf16 == "f16" else capi.COMPUTE_BF16
            # [taps][cin_pad/8][wn][8]: a B fragment (8 consecutive k for one column) is one 16-byte read
            t = torch.from_numpy(packed).to(dt).reshape(taps, self.cin_pad // 8, 8, self.wn).permute(0, 1, 3, 2).contiguous()
            self.w16 = t.to(device)
        self.w_bf16 = self.w16  # (older name)
        self.bias = None if bias is None else torch.from_numpy(np.ascontiguousarray(bias, dtype=np.float32)).to(device)


def pack_conv(weight, bias, device, dil=1, mode=capi.MODE_LINEAR, bf16=False, small_only=False):
    """torch Conv1d weight [cout, cin, k] ('same' padding (k-1)/2*dil) or Linear weight [cout, cin]."""
    w = _np(weight)
    if w.ndim == 2:
        w = w[:, :, None]
    k = w.shape[2]
    w_kio = np.ascontiguousarray(np.transpose(w, (2, 1, 0)))
    return ConvWeights(w_kio, None if bias is None else _np(bias), mode, dil, (k - 1) // 2 * dil, device, bf16, small_only)


def pack_ffn(w1, b1, w2, device, fmt):
    """Both weight matrices and the first bias of a kernel-size-1 feed-forward module in tts_ffn_fused's fragment order
    (include/toucan_tts.h): w1 = w_1.weight [hidden, 192(, 1)], b1 = w_1.bias, w2 = w_2.weight [192, hidden(, 1)]; fmt "bf16" /
    "f16".  28 KB per 32 hidden channels, returned as a flat int16 tensor (the bias fragments are fp32 bit patterns)."""
    w1, w2, b1 = _np(w1).astype(np.float32), _np(w2).astype(np.float32), _np(b1).astype(np.float32)
    if w1.ndim == 3:
        assert w1.shape[2] == 1 and w2.shape[2] == 1, "the fused feed-forward kernel covers kernel size 1"
        w1, w2 = w1[:, :, 0], w2[:, :, 0]
    hidden, c = w1.shape
    assert c == 192 and w2.shape == (c, hidden) and b1.shape == (hidden,) and hidden % 32 == 0
    n = hidden // 32
    lane = np.arange(64)
    lk, r = lane // 32, lane % 32
    i = np.arange(8)
    frag = np.zeros((n, 24, 64, 8), dtype=np.float32)
    for ks in range(12):  # W1 fragment ks: [lane][i] = W1[32 c + r][16 ks + 8 lk + i]
        cols = 16 * ks + 8 * lk[:, None] + i[None, :]
        frag[:, ks] = w1.reshape(n, 32, c)[:, r[:, None], cols]
    slot = np.where(i[None, :] < 4, 4 * lk[:, None] + i[None, :], 8 + 4 * lk[:, None] + i[None, :] - 4)  # [lane][i]
    w2c = w2.reshape(c, n, 32)
    for j in range(6):  # W2 fragment (j, ab): [lane][i] = W2[32 j + r][32 c + 16 ab + slot]
        for ab in range(2):
            frag[:, 12 + 2 * j + ab] = np.transpose(w2c[32 * j + r[:, None], :, 16 * ab + slot], (2, 0, 1))
    dt = torch.float16 if fmt == "f16" else torch.bfloat16
    w16 = torch.from_numpy(frag).to(dt).view(torch.int16).numpy()  # [n][24][64][8] 16-bit patterns
    bias = np.zeros((n, 4, 64, 4), dtype=np.float32)  # b1 fragment q: [lane][4] = b1[32 c + 8 q + 4 lk + 0..3]
    for q in range(4):
        bias[:, q] = b1.reshape(n, 32)[:, 8 * q + 4 * lk[:, None] + np.arange(4)[None, :]]
    out = np.concatenate([w16.reshape(n, -1), bias.view(np.int16).reshape(n, -1)], axis=1)  # [n][28 KB / 2]
    assert out.shape[1] * 2 == 28 * 1024
    return torch.from_numpy(np.ascontiguousarray(out).reshape(-1)).to(device)


def pack_conv_transpose(weight, bias, stride, device, bf16=False):
    """ConvTranspose1d weight [cin, cout, k] with k = 2*stride, padding = (k-stride)//2 (InferenceBigVGAN.py:41-46).

    out[s*q + r] = sum_t x[t] . w[:, :, s*(q-t) + r + pad]; only t in {q-1, q, q+1} can hit a valid tap, so this is
    a 3-tap conv (pad_left 1) with `stride*cout` output channels, column r*cout + co.  One third of the packed taps
    are structural zeros (each phase uses 2 of the 3 frames)."""
    w = _np(weight)
    cin, cout, k = w.shape
    s = int(stride)
    assert k == 2 * s, "polyphase packing assumes kernel = 2*stride"
    pad = (k - s) // 2
    w_kio = np.zeros((3, cin, s * cout), dtype=np.float32)
    for j in range(3):  # tap j reads input frame q + j - 1
        for r in range(s):
            kidx = s * (1 - j) + r + pad
            if 0 <= kidx < k:
                w_kio[j, :, r * cout:(r + 1) * cout] = w[:, :, kidx]
    b = None if bias is None else np.tile(_np(bias), s)
    cw = ConvWeights(w_kio, b, capi.MODE_LINEAR, 1, 1, device, bf16)
    cw.algo_taps = 2
    return cw


def invconv_inverse(sd, prefix):
    """Glow.py:130-139: W = P (L*mask + I) (U*mask^T + diag(sign*exp(log_s))), inverse stored for the reverse pass."""
    f = lambda k: _np(sd[prefix + k]).astype(np.float64)
    l = f("l") * f("l_mask") + f("eye")
    u = f("u") * f("l_mask").T + np.diag(f("sign_s") * np.exp(f("log_s")))
    w = f("p") @ (l @ u)
    return np.linalg.inv(w.astype(np.float32).astype(np.float64)).astype(np.float32)


def rel_pos_encoding(pmax, d=192):
    """Rows p = -(pmax-1) .. pmax-1 of the sinusoid table (Layers/PositionalEncoding.py:90-117), fp32 like the reference."""
    pos = np.arange(-(pmax - 1), pmax, dtype=np.float32)[:, None]
    div = np.exp(np.arange(0, d, 2, dtype=np.float32) * np.float32(-(math.log(10000.0) / d))).astype(np.float32)
    pe = np.zeros((2 * pmax - 1, d), dtype=np.float32)
    arg = (pos * div).astype(np.float32)
    pe[:, 0::2] = np.sin(arg)
    pe[:, 1::2] = np.cos(arg)
    return pe


def stored_antialias_filter(sd):
    """The anti-alias filter a BigVGAN checkpoint stores as buffers of its Activation1d modules
    (``...upsample.filter`` / ``...downsample.lowpass.filter``, [1,1,12]) - None when the state dict has none (then
    ``kaiser_sinc_filter12`` restates the package's design formula).  The kernels use ONE 12-tap filter for the 2x
    up- and down-sampler of every activation, which is what the package builds; a checkpoint whose stored filters differ
    from each other is refused instead of being silently approximated."""
    keys = sorted(k for k in sd if k.endswith("upsample.filter") or k.endswith("downsample.lowpass.filter"))
    if not keys:
        return None
    first = _np(sd[keys[0]]).reshape(-1).astype(np.float32)
    if first.size != 12:
        raise NotImplementedError(f"{keys[0]}: {first.size}-tap anti-alias filter, the kernels implement the 12-tap 2x design")
    for k in keys[1:]:
        other = _np(sd[k]).reshape(-1).astype(np.float32)
        if other.shape != first.shape or not np.allclose(other, first, rtol=0.0, atol=1e-7):
            raise NotImplementedError(f"{k} differs from {keys[0]}: per-activation anti-alias filters are not supported")
    return first


def snake_fir_table(filt, device):
    """The anti-alias filter as matrix-core operands for the fused residual step (tts_snake_fir_table, csrc/snake_mfma.h)."""
    import ctypes
    f = np.ascontiguousarray(filt, dtype=np.float32)
    assert f.size == 12
    buf = np.zeros(4096, dtype=np.uint8)
    capi.check(capi.lib().tts_snake_fir_table(f.ctypes.data_as(ctypes.c_void_p), buf.ctypes.data_as(ctypes.c_void_p)), "tts_snake_fir_table")
    return torch.from_numpy(buf).to(device)


def kaiser_sinc_filter12():
    """alias_free_torch's kaiser_sinc_filter1d(cutoff 0.25, half_width 0.3, 12 taps) - third party, PARITY UNPINNED.
    A = 2.285*(K/2-1)*pi*4*half_width + 7.95; beta from Kaiser's formula; h = 2c*w*sinc(2c*t), normalised to sum 1."""
    k, cutoff, half_width = 12, 0.25, 0.3
    a = 2.285 * (k // 2 - 1) * math.pi * (4 * half_width) + 7.95
    if a > 50.0:
        beta = 0.1102 * (a - 8.7)
    elif a >= 21.0:
        beta = 0.5842 * (a - 21.0) ** 0.4 + 0.07886 * (a - 21.0)
    else:
        beta = 0.0
    n = np.arange(k, dtype=np.float64)
    win = np.i0(beta * np.sqrt(1.0 - (2.0 * n / (k - 1) - 1.0) ** 2)) / np.i0(beta)
    t = np.arange(-(k // 2), k // 2, dtype=np.float64) + 0.5
    h = 2 * cutoff * win * np.sinc(2 * cutoff * t)
    return (h / h.sum()).astype(np.float32)
