"""Kernel-level parity on a real MI355X: every entry point of libtoucan_hip.so against the numpy ABI
emulator (tests/abi_emulator.py) on seeded random inputs, ragged batches and edge cases.
Tolerances: fp32 kernels 2e-5 relative to the output scale (rounding order only); bf16 MFMA 2e-2."""
import numpy as np
import pytest
import torch

import ims_toucan_prosody_variance_amd  # noqa: F401
from ims_toucan_prosody_variance_amd import capi, engine, packing
from ims_toucan_prosody_variance_amd.ragged import Ragged
from tests import abi_emulator

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    ops = engine.Ops("cuda:0")
    assert not isinstance(ops.lib, abi_emulator.Emulator)
    return ops


@pytest.fixture(scope="module")
def cpu():
    return engine.Ops("cpu", lib=abi_emulator.Emulator())


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def close(a, b, tol=2e-5):
    a, b = a.detach().cpu().float().numpy(), b.detach().cpu().float().numpy()
    scale = max(1.0, float(np.abs(b).max()))
    err = float(np.abs(a - b).max())
    assert err <= tol * scale, f"max abs err {err:.3e} vs tol {tol * scale:.3e}"


PACK16 = {capi.COMPUTE_F32: True, capi.COMPUTE_BF16: "bf16", capi.COMPUTE_F16: "f16", capi.COMPUTE_F32X3: "x3"}  # pack_conv's 16-bit copy per compute mode
DT16 = {capi.COMPUTE_BF16: torch.bfloat16, capi.COMPUTE_F16: torch.float16}
# relative to the output scale (F32X3: the split fp32 product - ~22 bits per product, fp32 accumulation: the fp32 tolerance holds)
TOL = {capi.COMPUTE_F32: 2e-5, capi.COMPUTE_BF16: 2e-2, capi.COMPUTE_F16: 3e-3, capi.COMPUTE_F32X3: 2e-5}
ALL_COMPUTE = [capi.COMPUTE_F32, capi.COMPUTE_BF16, capi.COMPUTE_F16, capi.COMPUTE_F32X3]


def both(gpu, cpu, fn):
    """fn(ops, to) runs the op with tensors moved by `to`; returns output tensor(s)."""
    out_g = fn(gpu, lambda t: t.to("cuda:0").contiguous())
    torch.cuda.synchronize()
    out_c = fn(cpu, lambda t: t.clone().contiguous())
    return out_g, out_c


CONV_CASES = [
    # cin, cout, k, dil, mode, lengths
    (62, 100, 1, 1, capi.MODE_LINEAR, [7, 20]),
    (192, 1536, 1, 1, capi.MODE_LINEAR, [128, 97, 1]),
    (1536, 192, 1, 1, capi.MODE_LINEAR, [130]),
    (192, 576, 1, 1, capi.MODE_LINEAR, [300, 5]),
    (192, 256, 5, 1, capi.MODE_LINEAR, [64, 3, 129]),
    (80, 512, 7, 1, capi.MODE_LINEAR, [42, 126]),
    (64, 64, 11, 5, capi.MODE_LINEAR, [700, 30]),
    (32, 32, 3, 3, capi.MODE_LINEAR, [1000, 257]),
    (256, 1, 1, 1, capi.MODE_LINEAR, [20, 7]),
    (272, 192, 5, 1, capi.MODE_LINEAR, [88]),
    (192, 256, 7, 3, capi.MODE_LINEAR, [100, 37]),   # small form: halo 18 (bf16 window falls back to synchronous staging, fp32 prefetches)
    (96, 128, 11, 5, capi.MODE_LINEAR, [90]),        # small form: halo 50 (both fall back)
    (192, 1536, 1, 1, capi.MODE_LINEAR, [700, 700, 100]),   # small form, grid >= 512 workgroups: 64 x 128 tiles
    (192, 384, 3, 2, capi.MODE_LINEAR, [3000, 2600, 70]),   # 〃 with taps
    (192, 384, 1, 1, capi.MODE_GLU, [128, 33]),
    (192, 384, 5, 1, capi.MODE_GATED, [63, 21]),
    (192, 160, 1, 1, capi.MODE_COUPLING, [44, 63]),
]


@pytest.mark.parametrize("cin,cout,k,dil,mode,lengths", CONV_CASES)
@pytest.mark.parametrize("compute", ALL_COMPUTE)
@pytest.mark.parametrize("form", ["regular", "small"])
def test_conv1d(gpu, cpu, cin, cout, k, dil, mode, lengths, compute, form):
    # form: the regular 128/256-row tiles, or the 64 x 64 small-batch tiles the host picks when the grid is small
    w = rnd(cout, cin, k, seed=1, scale=1.0 / np.sqrt(cin * k)).numpy()
    b = rnd(cout, seed=2, scale=0.1).numpy()
    if form == "small" and not packing.pack_conv(w, b, "cpu", dil=dil, mode=mode).small_tile_rows:
        pytest.skip("shape has no small-batch form")
    gpu.small_tile_blocks = 0 if form == "regular" else 1 << 30
    dual = mode != capi.MODE_LINEAR
    co = cout // 2 if dual else cout

    def run(ops, to):
        rag = Ragged(lengths, ops.device, align=2)
        R = rag.total_rows
        cw = packing.pack_conv(w, b, ops.device, dil=dil, mode=mode, bf16=PACK16[compute])
        x = to(rnd(R, cin + 3, seed=3))[:, :cin]  # strided input view
        y = to(rnd(R, co + 5, seed=4))
        res = to(rnd(R, co, seed=5))
        pre = to(rnd(R, 2 * co if dual else co, seed=6))
        sv = to(rnd(len(lengths), co, seed=7))
        aux = to(rnd(R, co, seed=8)) if mode == capi.MODE_COUPLING else None
        ops.conv(cw, x, y[:, :co], rag, pre=capi.PRE_LRELU, pre_slope=0.1, act=capi.ACT_TANH, alpha=0.5, seqvec=sv, preadd=pre, res=res,
                 res_scale=0.25, aux=aux, accumulate=True, compute=compute)
        return y

    try:
        g, c = both(gpu, cpu, run)
    finally:
        gpu.small_tile_blocks = 1536
    close(g, c, TOL[compute])


SPLIT_K_CASES = [
    # cin, cout, k, dil, mode, lengths        (fp32, <= 128 workgroups if cut into 64 x 64 tiles, taps * cin >= 64)
    (192, 384, 5, 1, capi.MODE_GATED, [320]),            # the flow's WaveNet in-layer at batch 1
    (192, 384, 5, 1, capi.MODE_GATED, [63, 21, 1, 130]),
    (1536, 192, 1, 1, capi.MODE_LINEAR, [128]),          # second feed-forward conv of an encoder block at batch 1
    (384, 1536, 1, 1, capi.MODE_LINEAR, [128, 5]),
    (256, 256, 3, 1, capi.MODE_LINEAR, [70, 33, 1]),     # variance-predictor conv
    (80, 512, 5, 1, capi.MODE_LINEAR, [100]),            # PostNet input conv: cin 80 < cin_pad 96
    (272, 128, 7, 3, capi.MODE_LINEAR, [64, 65]),        # 17 channel groups (not a multiple of the four wavefronts), halo 18
    (32, 64, 3, 1, capi.MODE_LINEAR, [50]),              # two channel groups per tap: fewer k-steps per tap than wavefronts
    (384, 384, 1, 1, capi.MODE_GLU, [128, 33]),
    (192, 256, 3, 1, capi.MODE_COUPLING, [44, 63]),
    (192, 576, 1, 1, capi.MODE_LINEAR, [128]),           # q/k/v projection: 192 products per output
    (80, 384, 1, 1, capi.MODE_LINEAR, [320, 2]),         # the flow's input conv: 80 products, 5 channel groups
    (256, 80, 5, 1, capi.MODE_LINEAR, [640]),            # no small-batch form (packed width 96): row blocks cut out of 128-row tiles
    (256, 1, 1, 1, capi.MODE_LINEAR, [128, 9]),          # predictor output: one column of a 256 x 32 tile
    (192, 160, 1, 1, capi.MODE_COUPLING, [320]),         # coupling conv of the flow (half width 80, packed 96)
]


def _split_k_run(w, b, cin, cout, k, dil, mode, lengths):
    dual = mode != capi.MODE_LINEAR
    co = cout // 2 if dual else cout

    def run(ops, to):
        rag = Ragged(lengths, ops.device, align=2)
        R = rag.total_rows
        cw = packing.pack_conv(w, b, ops.device, dil=dil, mode=mode)
        x = to(rnd(R, cin, seed=3))
        y = to(rnd(R, co + 5, seed=4))
        res = to(rnd(R, co, seed=5))
        pre = to(rnd(R, 2 * co if dual else co, seed=6))
        sv = to(rnd(len(lengths), co, seed=7))
        aux = to(rnd(R, co, seed=8)) if mode == capi.MODE_COUPLING else None
        ops.conv(cw, x, y[:, :co], rag, pre=capi.PRE_LRELU, pre_slope=0.1, act=capi.ACT_TANH, alpha=0.5, seqvec=sv, preadd=pre, res=res,
                 res_scale=0.25, aux=aux, accumulate=True, compute=capi.COMPUTE_F32)
        return y
    return run


@pytest.mark.parametrize("cin,cout,k,dil,mode,lengths", SPLIT_K_CASES)
def test_conv1d_split_k_form(gpu, cpu, cin, cout, k, dil, mode, lengths, monkeypatch):
    """The split-K form of the fp32 convs (TTS_IO_SPLIT_K: on grids of a few workgroups - these are; TTS_IO_SPLIT_K_ALWAYS: at any
    size): against the emulator, and against the same launch without the flag (rounding order only; the flag must really change
    the kernel: the results differ in the last bits).  The 16 x 16 and the 32 x 32 tile forms (TOUCAN_NO_SPLIT_K16 forces the
    latter) sum in ONE order: bit-identical - which form the grid heuristics pick is a speed choice only."""
    w = rnd(cout, cin, k, seed=1, scale=1.0 / np.sqrt(cin * k)).numpy()
    b = rnd(cout, seed=2, scale=0.1).numpy()
    run = _split_k_run(w, b, cin, cout, k, dil, mode, lengths)
    dev = lambda t: t.to("cuda:0").contiguous()
    gpu.small_tile_blocks = 1 << 30
    try:
        gpu.split_k = 1
        g, c = both(gpu, cpu, run)
        gpu.split_k = 2
        always = run(gpu, dev)
        monkeypatch.setenv("TOUCAN_NO_SPLIT_K16", "1")
        g32 = run(gpu, dev)
        monkeypatch.delenv("TOUCAN_NO_SPLIT_K16")
        gpu.split_k = 0
        plain = run(gpu, dev)
        gpu.split_k = 1
        monkeypatch.setenv("TOUCAN_NO_SPLIT_K", "1")  # the escape hatch gives the other forms back, bit for bit
        off = run(gpu, dev)
    finally:
        gpu.small_tile_blocks = 1536
        gpu.split_k = 0
    close(g, c, 2e-5)
    close(g, plain, 1e-5)
    assert not torch.equal(g, plain), "the split-K form did not run"
    assert torch.equal(g, g32), "16 x 16 and 32 x 32 split-K tiles must sum in the same order"
    assert torch.equal(g, always)
    assert torch.equal(off, plain)


def test_conv1d_split_k_always_is_independent_of_the_grid(gpu, cpu):
    """TTS_IO_SPLIT_K_ALWAYS (phoneme stages of the fp32 acoustic model): a 32-utterance batch - a grid far beyond the small-grid
    limit, where TTS_IO_SPLIT_K falls back to the tiled form - gives every utterance the bits it gets alone."""
    cin, cout, k = 192, 1536, 1
    w = rnd(cout, cin, k, seed=1, scale=1.0 / np.sqrt(cin * k)).numpy()
    b = rnd(cout, seed=2, scale=0.1).numpy()
    lengths = [128 - 3 * u for u in range(32)]
    dev = lambda t: t.to("cuda:0").contiguous()
    cw = packing.pack_conv(w, b, gpu.device, mode=capi.MODE_LINEAR)
    rag = Ragged(lengths, gpu.device, align=2)
    x = dev(rnd(rag.total_rows, cin, seed=3))

    def conv(rg, xx):
        y = dev(torch.zeros(rg.total_rows, cout))
        gpu.conv(cw, xx, y, rg, act=capi.ACT_RELU, compute=capi.COMPUTE_F32)
        return y
    try:
        gpu.split_k = 2
        big = conv(rag, x)
        for u in (0, 7, 31):
            b0, n = rag.begins[u], rag.lengths[u]
            one = conv(Ragged([n], gpu.device, align=2), x[b0:b0 + n].contiguous())
            assert torch.equal(one[:n], big[b0:b0 + n]), u
        gpu.split_k = 1
        auto = conv(rag, x)
        gpu.split_k = 0
        plain = conv(rag, x)
    finally:
        gpu.split_k = 0
    assert torch.equal(auto, plain) and not torch.equal(big, plain)
    close(big, plain, 1e-5)


@pytest.mark.parametrize("c,k,dil,lengths", [(32, 3, 1, [1000, 9, 257]), (64, 11, 5, [300, 40]), (256, 7, 3, [130, 1, 2]), (128, 3, 5, [64, 8])])
@pytest.mark.parametrize("compute", ALL_COMPUTE[:3])  # (the snake prologue does not exist for the split fp32 product)
def test_conv1d_with_fused_antialiased_snake(gpu, cpu, c, k, dil, lengths, compute):
    """TTS_PRE_SNAKE: the conv's input staging applies Activation1d(SnakeBeta) (AMP.py:53-56) - ragged edges included."""
    w = rnd(c, c, k, seed=1, scale=1.0 / np.sqrt(c * k)).numpy()
    b = rnd(c, seed=2, scale=0.1).numpy()

    def run(ops, to):
        rag = Ragged(lengths, ops.device, align=2)
        R = rag.total_rows
        cw = packing.pack_conv(w, b, ops.device, dil=dil, bf16=PACK16[compute])
        x = to(rnd(R, c, seed=3))
        y = to(torch.zeros(R, c))
        res = to(rnd(R, c, seed=5))
        sn = (to(rnd(c, seed=6, scale=0.3)), to(rnd(c, seed=7, scale=0.3)), to(torch.from_numpy(packing.kaiser_sinc_filter12())))
        return ops.conv(cw, x, y, rag, pre=capi.PRE_SNAKE, snake=sn, res=res, compute=compute)

    g, cc = both(gpu, cpu, run)
    close(g, cc, 3e-5 if compute == capi.COMPUTE_F32 else TOL[compute])


@pytest.mark.parametrize("c,k,dil,lengths", [(32, 3, 1, [1000, 9, 225]), (32, 11, 5, [700, 30]), (64, 7, 3, [224, 449, 1]), (64, 11, 5, [300]),
                                             (128, 3, 1, [500, 17]), (128, 11, 5, [260, 100]), (128, 7, 1, [2, 223]),
                                             (256, 3, 1, [200, 97]), (256, 11, 5, [130, 96, 1]), (256, 7, 3, [95])])
@pytest.mark.parametrize("act", [capi.PRE_LRELU, capi.PRE_SNAKE])
@pytest.mark.parametrize("store,compute", [(torch.float32, capi.COMPUTE_BF16), (torch.bfloat16, capi.COMPUTE_BF16),
                                           (torch.float32, capi.COMPUTE_F16), (torch.float16, capi.COMPUTE_F16)])
def test_fused_resblock_step(gpu, cpu, c, k, dil, lengths, act, store, compute):
    """tts_resblock_step against the emulator (same 16-bit rounding points) on ragged batches incl. tile-boundary lengths,
    in both 16-bit formats (bf16 MFMA / fp16 MFMA) with fp32 or 16-bit tensors in HBM."""
    w1 = rnd(c, c, k, seed=1, scale=1.0 / np.sqrt(c * k)).numpy()
    w2 = rnd(c, c, k, seed=2, scale=1.0 / np.sqrt(c * k)).numpy()
    b1, b2 = rnd(c, seed=3, scale=0.1).numpy(), rnd(c, seed=4, scale=0.1).numpy()

    def run(ops, to):
        rag = Ragged(lengths, ops.device, align=2)
        R = rag.total_rows
        c1 = packing.pack_conv(w1, b1, ops.device, dil=dil, bf16=PACK16[compute])
        c2 = packing.pack_conv(w2, b2, ops.device, dil=1, bf16=PACK16[compute])
        x = to(rnd(R, c, seed=5).to(store))
        y = to(rnd(R, c, seed=6).to(store))
        sn1 = (to(rnd(c, seed=7, scale=0.3)), to(rnd(c, seed=8, scale=0.3)))
        sn2 = (to(rnd(c, seed=9, scale=0.3)), to(rnd(c, seed=10, scale=0.3)))
        filt = to(torch.from_numpy(packing.kaiser_sinc_filter12()))
        return ops.resblock_step(c1, c2, x, y, rag, act, 0.1, sn1, sn2, filt, alpha=1.0 / 3.0, res_scale=1.0 / 3.0, accumulate=True)

    g, cc = both(gpu, cpu, run)
    if compute == capi.COMPUTE_F16:
        close(g, cc, 2e-3 if store == torch.float32 else 3e-3)
    else:
        close(g, cc, 1e-2 if store == torch.float32 else 2e-2)


@pytest.mark.parametrize("compute", [capi.COMPUTE_BF16, capi.COMPUTE_F16])
def test_conv1d_bf16_tensors_in_hbm(gpu, cpu, compute):
    """TTS_IO_*_BF16 (+ TTS_IO_F16): x / y / res stored as 16-bit tensors (transposed-conv outputs and residual streams of the
    bf16 / fp16 vocoder)."""
    dt = DT16[compute]
    w = rnd(64, 64, 3, seed=1, scale=0.1).numpy()

    def run(ops, to):
        rag = Ragged([300, 17], ops.device, align=2)
        R = rag.total_rows
        cw = packing.pack_conv(w, rnd(64, seed=2, scale=0.1).numpy(), ops.device, bf16=PACK16[compute])
        x = to(rnd(R, 64, seed=3).to(dt))
        y = to(rnd(R, 64, seed=4).to(dt))
        res = to(rnd(R, 64, seed=5).to(dt))
        sn = (to(rnd(64, seed=6, scale=0.3)), to(rnd(64, seed=7, scale=0.3)), to(torch.from_numpy(packing.kaiser_sinc_filter12())))
        ops.conv(cw, x, y, rag, pre=capi.PRE_SNAKE, snake=sn, res=res, accumulate=True, compute=compute)
        y2 = to(torch.zeros(R, 64))
        ops.conv(cw, x, y2, rag, pre=capi.PRE_LRELU, pre_slope=0.1, res=res, compute=compute)  # 16-bit in, fp32 out
        s_out = to(torch.zeros(R, 64))
        ops.snake_aa(x, s_out, sn[0], sn[1], sn[2], 64, rag)
        s16 = to(torch.zeros(R, 64, dtype=dt))
        ops.snake_aa(x, s16, sn[0], sn[1], sn[2], 64, rag)  # 16-bit in, 16-bit out
        return torch.cat([y.float(), y2, s_out, s16.float()])

    g, cc = both(gpu, cpu, run)
    close(g, cc, TOL[compute])


def test_conv1d_rows_outside_utterances_are_untouched(gpu):
    rag = Ragged([5, 3], gpu.device, align=2)  # rows 5 and 9 are alignment padding
    cw = packing.pack_conv(rnd(32, 32, 3).numpy(), None, gpu.device)
    x = torch.ones(rag.total_rows, 32, device="cuda")
    y = torch.full((rag.total_rows, 32), 7.0, device="cuda")
    gpu.conv(cw, x, y, rag)
    torch.cuda.synchronize()
    assert torch.all(y[5] == 7.0) and torch.all(y[9] == 7.0)
    assert not torch.any(y[:5] == 7.0)


@pytest.mark.parametrize("stride,k,cin,cout", [(8, 16, 512, 256), (2, 4, 64, 32)])
def test_conv_transpose_polyphase_matches_torch_definition(gpu, stride, k, cin, cout):
    """The 3-tap polyphase packing against the textbook ConvTranspose1d definition (fp64 on the host)."""
    T = 37
    w = rnd(cin, cout, k, seed=11, scale=1.0 / np.sqrt(2 * cin))
    b = rnd(cout, seed=12, scale=0.1)
    x = rnd(T, cin, seed=13)
    ref = torch.nn.functional.conv_transpose1d(x.t().unsqueeze(0).double(), w.double(), b.double(), stride=stride, padding=(k - stride) // 2)[0].t()
    cw = packing.pack_conv_transpose(w.numpy(), b.numpy(), stride, gpu.device)
    rag = Ragged([T], gpu.device)
    y = gpu.conv(cw, x.cuda(), gpu.empty(T, stride * cout), rag)
    torch.cuda.synchronize()
    close(y.view(T * stride, cout), ref.float(), 2e-5)


def test_layernorm_cln_l2_groupnorm(gpu, cpu):
    def ln(ops, to):
        x = to(rnd(300, 192, seed=1))
        return ops.layernorm(x, ops.empty(300, 192), to(rnd(192, seed=2)), to(rnd(192, seed=3)), 300, 192)
    close(*both(gpu, cpu, ln))

    def cln(ops, to):
        rag = Ragged([70, 5, 128], ops.device)
        x = to(rnd(rag.total_rows, 256, seed=1).abs())
        return ops.cond_layernorm(x, ops.empty(rag.total_rows, 256), to(rnd(3, 256, seed=2)), to(rnd(3, 256, seed=3)), 256, rag)
    close(*both(gpu, cpu, cln), tol=1e-4)

    def l2(ops, to):
        return ops.l2_normalize(to(rnd(5, 64, seed=1)), ops.empty(5, 64))
    close(*both(gpu, cpu, l2))

    for c, groups, tanh in ((256, 32, True), (80, 20, False)):
        def gn(ops, to):
            rag = Ragged([101, 7, 640], ops.device, align=2)
            x = to(rnd(rag.total_rows, c, seed=1))
            res = to(rnd(rag.total_rows, c, seed=4))
            y = to(torch.zeros(rag.total_rows, c))
            return ops.groupnorm(x, y, to(rnd(c, seed=2)), to(rnd(c, seed=3)), c, groups, rag, tanh=tanh, res=res)
        close(*both(gpu, cpu, gn), tol=5e-5)


@pytest.mark.parametrize("lengths", [[7], [64], [65, 128, 1], [640, 333], [129, 31, 33]])
@pytest.mark.parametrize("flags", [0, capi.ATT_KEY_SPLIT_ALWAYS])
def test_relpos_attention(gpu, cpu, lengths, flags):
    """The fp32 matrix-core kernel in its plain form and in the key-split form forced at every grid size (the encoder of the fp32
    configuration), ragged batches incl. multi-tile utterances, against the emulator."""
    pmax = 700

    def run(ops, to):
        rag = Ragged(lengths, ops.device, align=2)
        qkv = to(rnd(rag.total_rows, 576, seed=1, scale=0.7))
        ptab = to(rnd(2 * pmax - 1, 192, seed=2, scale=0.5))
        ctx = to(torch.zeros(rag.total_rows, 192))
        return ops.attention(qkv, ptab, pmax, to(rnd(192, seed=3, scale=0.3)), to(rnd(192, seed=4, scale=0.3)), ctx, rag, 128, flags=flags)
    close(*both(gpu, cpu, run), tol=5e-5)


@pytest.mark.parametrize("lengths", [[7], [640], [333, 129], [160, 33, 1]])
def test_relpos_attention_key_split_form(gpu, cpu, lengths):
    """fp32 attention, key-split form: the four wavefronts of a workgroup split the keys of one 32-query block and merge their
    running (max, sum, output) in LDS.  The CALLER opts in (flags): TTS_ATT_KEY_SPLIT takes the form on small grids (these are),
    TTS_ATT_KEY_SPLIT_ALWAYS at any size, 0 never.  Against the emulator, and against the plain form: rounding order only - and
    really another kernel (the last bits differ); the two opt-in flags run the same kernel (bit-identical)."""
    pmax = 700

    def run(ops, to, flags=capi.ATT_KEY_SPLIT):
        rag = Ragged(lengths, ops.device, align=2)
        qkv = to(rnd(rag.total_rows, 576, seed=1, scale=0.7))
        ptab = to(rnd(2 * pmax - 1, 192, seed=2, scale=0.5))
        ctx = to(torch.zeros(rag.total_rows, 192))
        return ops.attention(qkv, ptab, pmax, to(rnd(192, seed=3, scale=0.3)), to(rnd(192, seed=4, scale=0.3)), ctx, rag, 128, flags=flags)
    g, c = both(gpu, cpu, run)
    dev = lambda t: t.to("cuda:0").contiguous()
    plain = run(gpu, dev, flags=0)
    always = run(gpu, dev, flags=capi.ATT_KEY_SPLIT_ALWAYS)
    close(g, c, tol=5e-5)
    close(g, plain, tol=1e-5)
    assert torch.equal(g, always)
    if max(lengths) > 32:  # (with a single 32-key step the merge is exact: wavefront 0 holds everything)
        assert not torch.equal(g, plain), "the key-split form did not run"
    else:
        assert torch.equal(g, plain)


@pytest.mark.parametrize("lengths", [[640, 37, 128, 1], [129], [700], [97, 33, 65, 32]])  # (two 32-key tiles per step: every way a step can end early)
def test_relpos_attention_f16(gpu, cpu, lengths):
    """tts_relpos_attention_f16 (the three contractions on the fp16 matrix cores) against the emulator with the same rounding points,
    and against the exact fp32 kernel within the fp16 tolerance."""
    pmax = 700

    def run(ops, to, f16=True):
        rag = Ragged(lengths, ops.device, align=2)
        qkv = to(rnd(rag.total_rows, 576, seed=1, scale=0.7))
        ptab = to(rnd(2 * pmax - 1, 192, seed=2, scale=0.5))
        ctx = to(torch.zeros(rag.total_rows, 192))
        return ops.attention(qkv, ptab, pmax, to(rnd(192, seed=3, scale=0.3)), to(rnd(192, seed=4, scale=0.3)), ctx, rag, 128, f16=f16)
    g, c = both(gpu, cpu, run)
    rag = Ragged(lengths, "cpu", align=2)
    rows = torch.cat([torch.arange(b, b + n) for b, n in zip(rag.begins, rag.lengths)])
    close(g.cpu()[rows], c[rows], tol=1e-3)
    exact = run(gpu, lambda t: t.to("cuda:0").contiguous(), f16=False)
    torch.cuda.synchronize()
    close(g.cpu()[rows], exact.cpu()[rows], tol=TOL[capi.COMPUTE_F16])


@pytest.mark.parametrize("k", [7, 31])
def test_dwconv_swish(gpu, cpu, k):
    def run(ops, to):
        rag = Ragged([100, 3, 65], ops.device, align=2)
        x = to(rnd(rag.total_rows, 192, seed=1))
        y = to(torch.zeros(rag.total_rows, 192))
        return ops.dwconv_swish(x, y, to(rnd(k, 192, seed=2, scale=0.3)), to(rnd(192, seed=3, scale=0.1)), 192, k, rag)
    close(*both(gpu, cpu, run))


def test_duration_control_length_regulator(gpu, cpu):
    lengths = [20, 7, 33]

    def run(ops, to):
        rag = Ragged(lengths, ops.device)
        R = rag.total_rows
        g = torch.Generator().manual_seed(5)
        text = to((torch.rand(R, 62, generator=g) < 0.3).float())
        logd = to(rnd(R, seed=1, scale=0.5) + 1.6)
        # exact .5 cases for round-half-even: exp(x)-1 = 2.5 / 3.5
        logd[0], logd[1] = float(np.log(3.5)), float(np.log(4.5))
        d = to(torch.zeros(R, dtype=torch.int32))
        ops.duration_from_log(logd, d)
        p, e = to(rnd(R, seed=2)), to(rnd(R, seed=3).abs())
        ops.prosody_control(text, p, e, d, rag, 1.2, 1.3, 0.7, 1.2)
        if ops.device.type == "cuda":
            torch.cuda.synchronize()
        dh = d.cpu().numpy()
        Ts = [int(dh[b:b + n].sum()) for b, n in zip(rag.begins, rag.lengths)]
        ragf = Ragged(Ts, ops.device, align=2)
        enc = to(rnd(R, 192, seed=4))
        up = to(torch.zeros(ragf.total_rows, 192))
        dec = to(torch.zeros(ragf.total_rows, 192))
        ops.length_regulate(enc, p, e, to(rnd(192, seed=6)), to(rnd(192, seed=7)), to(rnd(192, seed=8)), to(rnd(192, seed=9)), d, rag, ragf,
                            up, dec, 13.0)
        return torch.cat([d.float(), p, e, up.reshape(-1), dec.reshape(-1)])
    g, c = both(gpu, cpu, run)
    assert torch.equal(g.cpu()[: sum(lengths)], c[: sum(lengths)])  # integer durations: bit exact
    close(g, c, 1e-5)


def test_length_regulator_all_zero_utterance_becomes_all_ones(gpu):
    rag = Ragged([4, 3], gpu.device)
    d = torch.tensor([0, 0, 0, 0, 2, 0, 1], dtype=torch.int32, device="cuda")
    ragf = Ragged([4, 3], gpu.device, align=2)
    enc = torch.arange(7, dtype=torch.float32, device="cuda")[:, None].repeat(1, 192).contiguous()
    z = torch.zeros(7, device="cuda")
    zero = torch.zeros(192, device="cuda")
    up = torch.zeros(ragf.total_rows, 192, device="cuda")
    gpu.length_regulate(enc, z, z, zero, zero, zero, zero, d, rag, ragf, up, None, 1.0)
    torch.cuda.synchronize()
    assert up[:, 0].tolist() == [0, 1, 2, 3, 4, 4, 6, 0]


def test_glow_invconv_actnorm(gpu, cpu):
    def run(ops, to):
        x = to(rnd(77, 160, seed=1))
        ops.glow_invconv_actnorm(x, 77, 160, to(rnd(16, seed=2)), to(rnd(160, seed=3, scale=0.1)), to(rnd(160, seed=4, scale=0.1)))
        return x
    close(*both(gpu, cpu, run))


@pytest.mark.parametrize("c,lengths", [(32, [1000, 9]), (256, [336, 1, 2, 70]), (64, [64, 8])])
def test_snake_aa(gpu, cpu, c, lengths):
    def run(ops, to):
        rag = Ragged(lengths, ops.device, align=2)
        x = to(rnd(rag.total_rows, c, seed=1))
        y = to(torch.zeros(rag.total_rows, c))
        filt = to(torch.from_numpy(packing.kaiser_sinc_filter12()))
        return ops.snake_aa(x, y, to(rnd(c, seed=2, scale=0.3)), to(rnd(c, seed=3, scale=0.3)), filt, c, rag)
    close(*both(gpu, cpu, run), tol=2e-5)


@pytest.mark.parametrize("pre", [capi.PRE_NONE, capi.PRE_LRELU])
def test_conv_post(gpu, cpu, pre):
    def run(ops, to):
        rag = Ragged([1000, 3, 256], ops.device, align=2)
        x = to(rnd(rag.total_rows, 32, seed=1))
        wav = to(torch.zeros(rag.total_rows))
        return ops.conv_post(x, 32, to(rnd(7, 32, seed=2, scale=0.1)), 0.05, pre, 0.01, wav, rag)
    close(*both(gpu, cpu, run))


def test_library_rejects_bad_arguments(gpu):
    rag = Ragged([10], gpu.device)
    cw = packing.pack_conv(rnd(32, 32, 3).numpy(), None, gpu.device)
    cw.tile_rows = 64  # tile table built for the wrong tile height
    with pytest.raises(capi.ToucanHipError, match="tile table"):
        gpu.conv(cw, gpu.empty(10, 32), gpu.empty(10, 32), rag)


@pytest.mark.parametrize("cin,cout,k,dil,mode,lengths", [
    (64, 192, 1, 1, capi.MODE_LINEAR, [1, 1, 1, 1, 1]),       # one row per utterance (the conditional-layer-norm MLPs)
    (192, 192, 1, 1, capi.MODE_LINEAR, [7]),
    (1536, 192, 1, 1, capi.MODE_LINEAR, [130]),
    (192, 576, 1, 1, capi.MODE_LINEAR, [97, 5]),
    (256, 256, 1, 1, capi.MODE_LINEAR, [1, 1, 1]),
    (192, 384, 1, 1, capi.MODE_GLU, [33, 128]),
    (192, 1536, 1, 1, capi.MODE_LINEAR, [64, 65]),
    (192, 384, 5, 1, capi.MODE_GATED, [63, 21, 1]),           # PostFlow WaveNet layer: taps shift the A rows, zero outside the utterance
    (192, 256, 3, 1, capi.MODE_LINEAR, [64, 3, 129]),
    (256, 256, 5, 2, capi.MODE_LINEAR, [7, 1, 40]),
])
@pytest.mark.parametrize("compute", ALL_COMPUTE)
def test_conv1d_rows_kernel(gpu, cpu, cin, cout, k, dil, mode, lengths, compute, monkeypatch):
    """1-tap convs with 16-byte aligned contiguous rows take the LDS-free kernel in the small-batch form (fp32 always, bf16 on
    latency-bound grids or when TOUCAN_GEMM_ROWS_BF16 is set); the multi-tap cases run the LDS-staged small form on the same
    aligned inputs (double-buffered window).  Every epilogue feature, against the emulator."""
    monkeypatch.setenv("TOUCAN_GEMM_ROWS_BF16", "1")
    w = rnd(cout, cin, k, seed=1, scale=1.0 / np.sqrt(cin * k)).numpy()
    b = rnd(cout, seed=2, scale=0.1).numpy()
    dual = mode != capi.MODE_LINEAR
    co = cout // 2 if dual else cout

    def run(ops, to):
        rag = Ragged(lengths, ops.device, align=2)
        R = rag.total_rows
        cw = packing.pack_conv(w, b, ops.device, dil=dil, mode=mode, bf16=PACK16[compute])
        x = to(rnd(R, cin, seed=3))
        y = to(rnd(R, co, seed=4))
        res = to(rnd(R, co, seed=5))
        pre = to(rnd(R, 2 * co if dual else co, seed=6))
        sv = to(rnd(len(lengths), co, seed=7))
        ops.conv(cw, x, y, rag, pre=capi.PRE_LRELU, pre_slope=0.1, act=capi.ACT_TANH, alpha=0.5, seqvec=sv, preadd=pre, res=res,
                 res_scale=0.25, accumulate=True, compute=compute)
        return y

    gpu.small_tile_blocks = 1 << 30
    try:
        g, c = both(gpu, cpu, run)
    finally:
        gpu.small_tile_blocks = 1536
    close(g, c, TOL[compute])


@pytest.mark.parametrize("n_seq,n_mlp,d_in,d_out", [(1, 24, 64, 256), (5, 3, 64, 256), (32, 24, 64, 256), (2, 2, 32, 48)])
def test_cln_mlp(gpu, cpu, n_seq, n_mlp, d_in, d_out):
    """All conditional-layer-norm scale / shift MLPs in one launch vs the fp64 emulator."""
    per = d_in * d_in + d_in + d_in * d_out + d_out + d_out * d_out + d_out
    assert capi.lib().tts_cln_mlp_weight_floats(d_in, d_out) == per
    w = rnd(n_mlp, per, seed=1, scale=0.2)
    e = torch.nn.functional.normalize(rnd(n_seq, d_in, seed=2), dim=1)

    def run(ops, to):
        return ops.cln_mlp(to(e), to(w).reshape(-1), n_mlp, d_in, d_out)

    g, c = both(gpu, cpu, run)
    assert tuple(g.shape) == (n_mlp, n_seq, d_out)
    close(g, c, 2e-6)


@pytest.mark.parametrize("cin,cout,mode,lengths", [(1536, 192, capi.MODE_LINEAR, [130, 7]), (192, 384, capi.MODE_GLU, [33]), (256, 256, capi.MODE_LINEAR, [1, 1])])
@pytest.mark.parametrize("pre", [capi.PRE_NONE, capi.PRE_LRELU])
@pytest.mark.parametrize("compute", [capi.COMPUTE_BF16, capi.COMPUTE_F16])
def test_conv1d_rows_kernel_bf16_input(gpu, cpu, cin, cout, mode, lengths, pre, compute, monkeypatch):
    """The LDS-free 1-tap kernel reading a 16-bit tensor (the FFN hidden state / WaveNet gate activations of the bf16 / fp16 configurations)."""
    monkeypatch.setenv("TOUCAN_GEMM_ROWS_BF16", "1")
    w = rnd(cout, cin, 1, seed=1, scale=1.0 / np.sqrt(cin)).numpy()
    b = rnd(cout, seed=2, scale=0.1).numpy()
    co = cout // 2 if mode != capi.MODE_LINEAR else cout

    def run(ops, to):
        rag = Ragged(lengths, ops.device, align=2)
        cw = packing.pack_conv(w, b, ops.device, mode=mode, bf16=PACK16[compute])
        x = to(rnd(rag.total_rows, cin, seed=3)).to(DT16[compute])
        y = to(rnd(rag.total_rows, co, seed=4))
        ops.conv(cw, x, y, rag, pre=pre, pre_slope=0.1, res=to(rnd(rag.total_rows, co, seed=5)), compute=compute)
        return y

    gpu.small_tile_blocks = 1 << 30
    try:
        g, c = both(gpu, cpu, run)
    finally:
        gpu.small_tile_blocks = 1536
    close(g, c, TOL[compute])


@pytest.mark.parametrize("lengths", [[1000, 9, 257], [250], [251, 1, 2, 499]])
@pytest.mark.parametrize("store", [torch.float32, torch.bfloat16, torch.float16])
def test_conv_post_snake(gpu, cpu, lengths, store):
    """activation_post (anti-aliased snake) + output conv + tanh fused (BigVGAN's last two ops) vs the fp64 emulator."""
    c = 32
    filt = torch.from_numpy(packing.kaiser_sinc_filter12())

    def run(ops, to):
        rag = Ragged(lengths, ops.device)
        x = to(rnd(rag.total_rows, c, seed=1)).to(store)
        wav = to(torch.full((rag.total_rows,), 9.0))
        return ops.conv_post_snake(x, c, to(rnd(7, c, seed=2, scale=0.1)), 0.05, to(rnd(c, seed=3, scale=0.3)), to(rnd(c, seed=4, scale=0.3)),
                                   to(filt), wav, rag)

    g, cc = both(gpu, cpu, run)
    close(g, cc, 2e-5)


@pytest.mark.parametrize("compute", [capi.COMPUTE_BF16, capi.COMPUTE_F16])
@pytest.mark.parametrize("cout2,lengths", [(384, [320, 63, 1, 130]), (192, [64, 65, 7])])
def test_fused_wavenet_layer(gpu, cpu, compute, cout2, lengths):
    """tts_wavenet_layer (gated 5-tap conv + cond, tanh.sigmoid, res/skip conv, state update) against the emulator on ragged
    batches incl. tile-boundary lengths; and against the two-launch form it replaces (same rounding points, other summation order)."""
    H = 192
    w_in = rnd(2 * H, H, 5, seed=1, scale=1.0 / np.sqrt(5 * H)).numpy()
    b_in = rnd(2 * H, seed=2, scale=0.1).numpy()
    w_rs = rnd(cout2, H, 1, seed=3, scale=1.0 / np.sqrt(H)).numpy()
    b_rs = rnd(cout2, seed=4, scale=0.1).numpy()

    def run(ops, to, fused=True):
        rag = Ragged(lengths, ops.device, align=2)
        R = rag.total_rows
        inl = packing.pack_conv(w_in, b_in, ops.device, mode=capi.MODE_GATED, bf16=PACK16[compute])
        rs = packing.pack_conv(w_rs, b_rs, ops.device, bf16=PACK16[compute])
        hs = to(rnd(R, 2 * H, seed=5))
        cond = to(rnd(R, 8 * H, seed=6, scale=0.5))[:, 2 * H:4 * H]  # a 384-column slice of the [R, 1536] conditioning
        out = to(rnd(R, 2 * H, seed=7))
        if fused:
            ops.wavenet_layer(inl, rs, hs, out, cond, rag)
            return out if cout2 == 384 else out[:, H:]
        acts = to(torch.zeros(R, H, dtype=DT16[compute]))
        ops.conv(inl, hs[:, :H], acts, rag, preadd=cond, compute=compute)
        ops.conv(rs, acts, hs if cout2 == 384 else hs[:, H:], rag, accumulate=True, compute=compute)
        return hs if cout2 == 384 else hs[:, H:]

    g, c = both(gpu, cpu, run)
    rag = Ragged(lengths, "cpu", align=2)
    rows = torch.cat([torch.arange(b, b + n) for b, n in zip(rag.begins, rag.lengths)])  # alignment rows are never written
    close(g.cpu()[rows], c[rows], TOL[compute])
    two = run(gpu, lambda t: t.to("cuda:0").contiguous(), fused=False)
    torch.cuda.synchronize()
    close(g.cpu()[rows], two.cpu()[rows], TOL[compute])


@pytest.mark.parametrize("compute", [capi.COMPUTE_BF16, capi.COMPUTE_F16])
@pytest.mark.parametrize("rows,hidden,post", [(300, 1536, True), (128, 1536, False), (1, 64, True), (4096 + 37, 1536, True)])
def test_fused_feed_forward(gpu, cpu, compute, rows, hidden, post):
    """tts_ffn_fused (LayerNorm, w_1, ReLU, w_2, half-step residual, optional final LayerNorm in one launch; in place) against the
    emulator - partial last workgroup, one row, a short hidden axis - and against the launches it replaces (same rounding
    points, other summation order)."""
    Cc = 192
    w1 = rnd(hidden, Cc, 1, seed=1, scale=1.0 / np.sqrt(Cc)).numpy()
    b1 = rnd(hidden, seed=2, scale=0.1).numpy()
    w2 = rnd(Cc, hidden, 1, seed=3, scale=1.0 / np.sqrt(hidden)).numpy()
    b2 = rnd(Cc, seed=4, scale=0.1).numpy()
    fmt = PACK16[compute]

    def run(ops, to, fused=True):
        x = to(rnd(rows, Cc, seed=5, scale=2.0))
        norm = (to(1.0 + rnd(Cc, seed=6, scale=0.1)), to(rnd(Cc, seed=7, scale=0.1)))
        fin = (to(1.0 + rnd(Cc, seed=8, scale=0.1)), to(rnd(Cc, seed=9, scale=0.1))) if post else None
        c1 = packing.pack_conv(w1, b1, ops.device, bf16=fmt)
        c2 = packing.pack_conv(w2, b2, ops.device, bf16=fmt)
        if fused:
            pk = packing.pack_ffn(w1, b1, w2, ops.device, "f16" if fmt == "f16" else "bf16")
            return ops.ffn_fused(x, x, norm, pk, c2.bias, rows, compute, post=fin)
        rag = Ragged([rows], ops.device)
        ln = to(torch.zeros(rows, Cc))
        hid = to(torch.zeros(rows, hidden, dtype=DT16[compute]))
        ops.layernorm(x, ln, *norm, rows, Cc)
        ops.conv(c1, ln, hid, rag, act=capi.ACT_RELU, compute=compute)
        ops.conv(c2, hid, x, rag, alpha=0.5, res=x, compute=compute)
        if fin is not None:
            ops.layernorm(x, x, *fin, rows, Cc)
        return x

    g, c = both(gpu, cpu, run)
    close(g, c, TOL[compute])
    if rows <= 300:
        two = run(gpu, lambda t: t.to("cuda:0").contiguous(), fused=False)
        torch.cuda.synchronize()
        close(g.cpu(), two.cpu(), TOL[compute])


def test_concurrent_fused_residual_steps_do_not_share_work_queues(gpu):
    """Two tts_resblock_step launches in flight at once on two streams (each draws its tiles from its own queue slot), several times
    over: every output is bit for bit what the same launch produces alone."""
    dev = "cuda:0"
    filt = torch.from_numpy(packing.kaiser_sinc_filter12()).to(dev)
    cases = []
    for c, k, dil, lengths, seed in ((64, 7, 3, [9000, 4000, 333], 1), (32, 3, 1, [20000, 77], 2)):
        w1 = rnd(c, c, k, seed=seed, scale=1.0 / np.sqrt(c * k)).numpy()
        w2 = rnd(c, c, k, seed=seed + 10, scale=1.0 / np.sqrt(c * k)).numpy()
        b = rnd(c, seed=seed + 20, scale=0.1).numpy()
        rag = Ragged(lengths, gpu.device, align=2)
        c1 = packing.pack_conv(w1, b, gpu.device, dil=dil, bf16="f16")
        c2 = packing.pack_conv(w2, b, gpu.device, dil=1, bf16="f16")
        x = rnd(rag.total_rows, c, seed=seed + 30).to(dev).to(torch.float16)
        sn = (rnd(c, seed=seed + 40, scale=0.3).to(dev), rnd(c, seed=seed + 50, scale=0.3).to(dev))
        cases.append((c1, c2, x, rag, sn))
    run = lambda cs, y: gpu.resblock_step(cs[0], cs[1], cs[2], y, cs[3], capi.PRE_SNAKE, 0.1, cs[4], cs[4], filt)
    alone = []
    for cs in cases:
        y = torch.zeros_like(cs[2])
        run(cs, y)
        alone.append(y)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    for _ in range(5):
        outs = [torch.zeros_like(cs[2]) for cs in cases]
        for st, cs, y in zip(streams, cases, outs):
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                run(cs, y)
                run(cs, y)  # (the second launch of a stream follows the first: same result)
        torch.cuda.synchronize()
        for y, want, rag in zip(outs, alone, (cs[3] for cs in cases)):
            rows = torch.cat([torch.arange(b0, b0 + n) for b0, n in zip(rag.begins, rag.lengths)]).to(dev)
            assert torch.equal(y[rows], want[rows])


def test_graph_replay_beside_eager_residual_steps_on_another_stream(gpu):
    """Work-queue slots cannot alias between launches that may be in flight together: a slot belongs to a stream (eager launches of
    one stream run in order), and every launch recorded into a HIP graph owns a slot of its own.  A graph of three fused residual
    steps is replayed on one stream while the same steps run eagerly on another - also on the very stream the graph was captured
    on - several times over; every output is bit for bit what the launches give alone.  (With slots chosen round-robin at launch
    time and baked into the graph - the previous scheme - a replay could meet an eager launch on its slot.)"""
    dev = "cuda:0"
    lib = capi.lib()
    filt = torch.from_numpy(packing.kaiser_sinc_filter12()).to(dev)
    c, k, dil, lengths = 64, 7, 3, [9000, 4000, 333]
    w1 = rnd(c, c, k, seed=1, scale=1.0 / np.sqrt(c * k)).numpy()
    w2 = rnd(c, c, k, seed=11, scale=1.0 / np.sqrt(c * k)).numpy()
    b = rnd(c, seed=21, scale=0.1).numpy()
    rag = Ragged(lengths, gpu.device, align=2)
    c1 = packing.pack_conv(w1, b, gpu.device, dil=dil, bf16="f16")
    c2 = packing.pack_conv(w2, b, gpu.device, dil=1, bf16="f16")
    x = rnd(rag.total_rows, c, seed=31).to(dev).to(torch.float16)
    sn = (rnd(c, seed=41, scale=0.3).to(dev), rnd(c, seed=51, scale=0.3).to(dev))
    rows = torch.cat([torch.arange(b0, b0 + n) for b0, n in zip(rag.begins, rag.lengths)]).to(dev)

    def chain(bufs):  # three dependent steps: x -> bufs[0] -> bufs[1] -> bufs[2]
        src = x
        for y in bufs:
            gpu.resblock_step(c1, c2, src, y, rag, capi.PRE_SNAKE, 0.1, sn, sn, filt)
            src = y
    want = [torch.zeros_like(x) for _ in range(3)]
    chain(want)
    torch.cuda.synchronize()
    s_cap, s_other = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    g_out = [torch.zeros_like(x) for _ in range(3)]
    used0 = lib.tts_diag_queue_slots_used()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s_cap):
        chain(g_out)  # (the capture stream's own eager slot exists before the capture)
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=s_cap):
            chain(g_out)
    torch.cuda.synchronize()
    assert lib.tts_diag_queue_slots_used() == used0 + 1 + 3, "one slot for the stream, one per recorded launch"
    for eager_stream in (s_other, s_cap):
        replay_stream = s_other if eager_stream is s_cap else s_cap
        for _ in range(4):
            for t in g_out:
                t.zero_()
            e_out = [torch.zeros_like(x) for _ in range(3)]
            torch.cuda.synchronize()
            with torch.cuda.stream(replay_stream):
                graph.replay()
            with torch.cuda.stream(eager_stream):
                chain(e_out)
                chain(e_out)
            torch.cuda.synchronize()
            for got_g, got_e, w in zip(g_out, e_out, want):
                assert torch.equal(got_g[rows], w[rows]) and torch.equal(got_e[rows], w[rows])
    assert lib.tts_diag_queue_slots_used() == used0 + 1 + 3 + 1, "eager launches of a stream reuse its slot"


@pytest.mark.parametrize("cin,cout,k,mode,rows", [(192, 1536, 1, capi.MODE_LINEAR, 4096), (1536, 192, 1, capi.MODE_LINEAR, 4096),
                                                  (192, 384, 5, capi.MODE_GATED, 2048), (256, 256, 5, capi.MODE_LINEAR, 3000)])
def test_split_fp32_product_is_as_close_to_float64_as_the_fp32_kernel(gpu, cin, cout, k, mode, rows):
    """TTS_COMPUTE_F32X3 (three fp16 MFMAs on split operands, fp32 accumulation) against a float64 product of the same fp32
    operands, beside the exact fp32 kernel: its error stays within 4x the fp32 kernel's (22 vs 24 bits per product; the
    accumulation is fp32 in both), far inside the 16-bit modes' - on operands of mixed magnitude (incl. values below fp16's
    normal range, which the scaled low plane still carries)."""
    w = rnd(cout, cin, k, seed=1, scale=1.0 / np.sqrt(cin * k)).numpy()
    dual = mode != capi.MODE_LINEAR
    co = cout // 2 if dual else cout
    rag = Ragged([rows], gpu.device, align=2)
    x = rnd(rows, cin, seed=3) * torch.logspace(-5, 1.5, cin)[None, :]  # per-channel magnitudes 1e-5 .. 30
    xd = x.to("cuda:0").contiguous()
    outs = {}
    gpu.small_tile_blocks = 0
    try:
        for compute in (capi.COMPUTE_F32, capi.COMPUTE_F32X3, capi.COMPUTE_F16):
            cw = packing.pack_conv(w, None, gpu.device, mode=capi.MODE_LINEAR, bf16=PACK16[compute])
            y = torch.zeros(rows, cout, device="cuda:0")
            gpu.conv(cw, xd, y, rag, compute=compute)
            outs[compute] = y.cpu().double()
    finally:
        gpu.small_tile_blocks = 1536
    xp = torch.zeros(rows + k - 1, cin, dtype=torch.float64)
    xp[(k - 1) // 2:(k - 1) // 2 + rows] = x.double()
    ref = sum(xp[j:j + rows] @ torch.from_numpy(w[:, :, j].T.astype(np.float64)) for j in range(k))
    scale = float(ref.abs().max())
    e32, ex3, e16 = (float((outs[c] - ref).abs().max()) / scale for c in (capi.COMPUTE_F32, capi.COMPUTE_F32X3, capi.COMPUTE_F16))
    print(f"max error / output scale: fp32 {e32:.2e}, split fp32 {ex3:.2e}, fp16 {e16:.2e}")
    assert ex3 < max(4 * e32, 2e-6) and ex3 < 1e-2 * e16 + 2e-6
