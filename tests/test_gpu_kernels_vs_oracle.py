"""The fused kernels against the ORACLE's own functions (oracle/toucan_oracle.py: `activation1d` + `snake_beta` + F.conv1d of
BigVGAN/AMP.py:53-58, `layer_norm` + `ffn` of Layers/EncoderLayer.py:84-90,128-136), not against the numpy ABI emulator that
tests/test_gpu_kernels.py uses - the emulator places its 16-bit rounding points where the kernels do, the oracle is plain fp32.
Tolerances are therefore those of the 16-bit formats relative to the output scale (fp16 3e-3, bf16 2e-2)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import ims_toucan_prosody_variance_amd  # noqa: F401
from ims_toucan_prosody_variance_amd import capi, engine, packing
from ims_toucan_prosody_variance_amd.ragged import Ragged

from oracle import toucan_oracle as oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = {capi.COMPUTE_BF16: 2e-2, capi.COMPUTE_F16: 3e-3}
FMT = {capi.COMPUTE_BF16: "bf16", capi.COMPUTE_F16: "f16"}


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def close(got, want, tol):
    got, want = got.detach().cpu().float().numpy(), want.detach().cpu().float().numpy()
    scale = max(1.0, float(np.abs(want).max()))
    err = float(np.abs(got - want).max())
    assert err <= tol * scale, f"max abs err {err:.3e} vs tol {tol * scale:.3e}"


@pytest.mark.parametrize("compute", [capi.COMPUTE_F16, capi.COMPUTE_BF16])
@pytest.mark.parametrize("c,k,dil,lengths", [(64, 7, 3, [500, 37]), (32, 11, 5, [700, 30]), (128, 3, 1, [260, 224, 1]), (64, 3, 1, [4000])])
def test_resblock_step_matches_the_oracles_amp_step(compute, c, k, dil, lengths):
    """One dilation step of an AMP block, y = c2(a2(c1(a1(x)))) + x (BigVGAN/AMP.py:53-58), per utterance of a ragged batch - incl.
    a batch large enough for the persistent workgroups to walk several tiles each and a 5-tile batch (fewer workgroups than XCDs)."""
    ops = engine.Ops(torch.device(DEV))
    w1, w2 = rnd(c, c, k, seed=1, scale=1.0 / np.sqrt(c * k)), rnd(c, c, k, seed=2, scale=1.0 / np.sqrt(c * k))
    b1, b2 = rnd(c, seed=3, scale=0.1), rnd(c, seed=4, scale=0.1)
    a1, be1, a2, be2 = (rnd(c, seed=s, scale=0.3) for s in (7, 8, 9, 10))
    filt = oracle.kaiser_sinc_filter()
    rag = Ragged(lengths, ops.device, align=2)
    x = rnd(rag.total_rows, c, seed=5)
    c1 = packing.pack_conv(w1.numpy(), b1.numpy(), ops.device, dil=dil, bf16=FMT[compute])
    c2 = packing.pack_conv(w2.numpy(), b2.numpy(), ops.device, dil=1, bf16=FMT[compute])
    xd = x.to(DEV).contiguous()
    yd = torch.zeros_like(xd)
    ops.resblock_step(c1, c2, xd, yd, rag, capi.PRE_SNAKE, 0.1, (a1.to(DEV), be1.to(DEV)), (a2.to(DEV), be2.to(DEV)), filt.to(DEV))
    torch.cuda.synchronize()
    for b0, n in zip(rag.begins, rag.lengths):
        xu = x[b0:b0 + n].t().contiguous()  # [C, T]
        t = oracle.activation1d(xu, lambda v: oracle.snake_beta(v, a1, be1), filt).unsqueeze(0)
        t = F.conv1d(t, w1, b1, padding=(k - 1) // 2 * dil, dilation=dil)
        t = oracle.activation1d(t[0], lambda v: oracle.snake_beta(v, a2, be2), filt).unsqueeze(0)
        want = (F.conv1d(t, w2, b2, padding=(k - 1) // 2)[0] + xu).t()
        close(yd[b0:b0 + n], want, TOL[compute])


@pytest.mark.parametrize("compute", [capi.COMPUTE_F16, capi.COMPUTE_BF16])
@pytest.mark.parametrize("rows,post", [(300, True), (129, False)])
def test_fused_feed_forward_matches_the_oracles_ffn(compute, rows, post):
    """x + 0.5 ffn(layer_norm(x)) (+ the block's final layer_norm): Layers/EncoderLayer.py:84-90 / :128-136 as the oracle states them."""
    ops = engine.Ops(torch.device(DEV))
    Cc, H = 192, 1536
    sd = {"w_1.weight": rnd(H, Cc, 1, seed=1, scale=1.0 / np.sqrt(Cc)), "w_1.bias": rnd(H, seed=2, scale=0.1),
          "w_2.weight": rnd(Cc, H, 1, seed=3, scale=1.0 / np.sqrt(H)), "w_2.bias": rnd(Cc, seed=4, scale=0.1)}
    g, b = 1.0 + rnd(Cc, seed=6, scale=0.1), rnd(Cc, seed=7, scale=0.1)
    g2, b2 = 1.0 + rnd(Cc, seed=8, scale=0.1), rnd(Cc, seed=9, scale=0.1)
    x = rnd(rows, Cc, seed=5, scale=2.0)
    want = x + 0.5 * oracle.ffn(oracle.layer_norm(x, g, b), sd, "")
    if post:
        want = oracle.layer_norm(want, g2, b2)
    pk = packing.pack_ffn(sd["w_1.weight"].numpy(), sd["w_1.bias"].numpy(), sd["w_2.weight"].numpy(), ops.device, FMT[compute])
    xd = x.to(DEV).contiguous()
    got = ops.ffn_fused(xd, xd, (g.to(DEV), b.to(DEV)), pk, sd["w_2.bias"].to(DEV), rows, compute,
                        post=(g2.to(DEV), b2.to(DEV)) if post else None)
    torch.cuda.synchronize()
    close(got, want, TOL[compute])


@pytest.mark.parametrize("compute", [capi.COMPUTE_F16, capi.COMPUTE_BF16])
def test_fused_wavenet_layers_match_the_oracles_wavenet(compute):
    """The four layers of one PostFlow WaveNet (wavenet.py:89-122 as the oracle states it: in_layer + conditioning, tanh . sigmoid,
    res_skip layer, state / skip update) through four tts_wavenet_layer launches on a ragged batch; the hidden state ping-pongs
    between two buffers as in the pipeline."""
    ops = engine.Ops(torch.device(DEV))
    H = 192
    lengths = [300, 65, 7]
    sd = {}
    for i in range(4):
        sd[f"in_layers.{i}.weight"] = rnd(2 * H, H, 5, seed=10 + i, scale=1.0 / np.sqrt(5 * H))
        sd[f"in_layers.{i}.bias"] = rnd(2 * H, seed=20 + i, scale=0.1)
        co = 2 * H if i < 3 else H
        sd[f"res_skip_layers.{i}.weight"] = rnd(co, H, 1, seed=30 + i, scale=1.0 / np.sqrt(H))
        sd[f"res_skip_layers.{i}.bias"] = rnd(co, seed=40 + i, scale=0.1)
    sd["cond_layer.weight"] = rnd(8 * H, 2 * H, 1, seed=50, scale=1.0 / np.sqrt(2 * H))
    sd["cond_layer.bias"] = rnd(8 * H, seed=51, scale=0.1)
    rag = Ragged(lengths, ops.device, align=2)
    R = rag.total_rows
    x = rnd(R, H, seed=1)
    g = rnd(R, 2 * H, seed=2)
    cond = F.conv1d(g.t().unsqueeze(0), sd["cond_layer.weight"], sd["cond_layer.bias"])[0].t().contiguous()  # [R, 1536] (rows are independent)
    hs = [torch.zeros(R, 2 * H), torch.zeros(R, 2 * H)]
    hs[0][:, :H] = x
    hs = [h.to(DEV).contiguous() for h in hs]
    cond_d = cond.to(DEV).contiguous()
    for i in range(4):
        inl = packing.pack_conv(sd[f"in_layers.{i}.weight"].numpy(), sd[f"in_layers.{i}.bias"].numpy(), ops.device, mode=capi.MODE_GATED, bf16=FMT[compute])
        rs = packing.pack_conv(sd[f"res_skip_layers.{i}.weight"].numpy(), sd[f"res_skip_layers.{i}.bias"].numpy(), ops.device, bf16=FMT[compute])
        src, dst = hs[i & 1], hs[(i + 1) & 1]  # (a layer writes [h + res | skip + skip'] of its input buffer into the other one)
        ops.wavenet_layer(inl, rs, src, dst, cond_d[:, i * 2 * H:(i + 1) * 2 * H], rag)
    torch.cuda.synchronize()
    got = hs[0][:, H:]  # after four layers the result sits in buffer 0's skip half
    for b0, n in zip(rag.begins, rag.lengths):
        want = oracle.wavenet(x[b0:b0 + n].t().contiguous(), g[b0:b0 + n].t().contiguous(), sd, "").t()
        close(got[b0:b0 + n], want, TOL[compute])
