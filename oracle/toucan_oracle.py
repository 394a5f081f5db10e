"""CPU ORACLE - TEST INFRASTRUCTURE ONLY.

A from-scratch CPU restatement (plain PyTorch fp32 CPU ops, functional style, one utterance
at a time exactly like the reference's batch-1 inference) of the ToucanTTS inference hot path.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this file; the product path (``ims-toucan-prosody-variance_amd``) never does and fails loudly when
its HIP extension is missing.

Pinning: the reference ships no tests or golden vectors for this path, so
this oracle is pinned against outputs of the reference's own modules, produced in the survey
container by ``tests/golden/make_golden.py`` (which imports /root/reference) and committed under
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` replays them.  One piece is
"parity unpinned": BigVGAN's anti-aliased activation comes from the third-party package
``alias_free_torch ~= 0.0.6`` (requirements.txt, last line; InferenceBigVGAN.py:8, AMP.py:8-9),
which is neither vendored in the reference nor installed.  ``activation1d`` below restates that
package's published algorithm (2x Kaiser-sinc up-sampling, activation, 2x low-pass decimation,
12 taps each); the goldens for BigVGAN were produced with the same restatement injected into the
reference's BigVGAN class, so they pin everything around the filter but not the filter itself.

Every function cites the reference file:line it follows (paths relative to /root/reference).
"""
import math

import torch
import torch.nn.functional as F

F_STRESSED, F_PHONEME, F_SILENCE, F_WORD_BOUNDARY, F_VOICED = 0, 15, 16, 21, 61  # articulatory_features.py:817-901


# --------------------------------------------------------------------------------------
# weight preparation (InferenceToucanTTS.py:321-330, Glow.py:130-139, InferenceBigVGAN.py:97-105)
# --------------------------------------------------------------------------------------
def fold_weight_norm(sd):
    """w = g * v / ||v|| over all dims but 0 (torch.nn.utils.weight_norm default dim=0)."""
    out = {}
    for k, v in sd.items():
        if k.endswith(".weight_g"):
            base = k[: -len(".weight_g")]
            vv = sd[base + ".weight_v"]
            nrm = vv.reshape(vv.shape[0], -1).norm(dim=1).reshape([-1] + [1] * (vv.dim() - 1))
            out[base + ".weight"] = vv * (v / nrm)
        elif k.endswith(".weight_v"):
            continue
        else:
            out[k] = v
    return out


def invconv_inverse(sd, prefix):
    """Glow.py:130-139: W = P (L*mask + I) (U*mask^T + diag(sign*exp(log_s))); stored inverse."""
    l = sd[prefix + "l"] * sd[prefix + "l_mask"] + sd[prefix + "eye"]
    u = sd[prefix + "u"] * sd[prefix + "l_mask"].t() + torch.diag(sd[prefix + "sign_s"] * torch.exp(sd[prefix + "log_s"]))
    w = sd[prefix + "p"] @ (l @ u)
    return torch.inverse(w.float())


def to_torch(sd_np):
    return {k: torch.from_numpy(v) if not torch.is_tensor(v) else v for k, v in sd_np.items()}


# --------------------------------------------------------------------------------------
# Conformer (Layers/Conformer.py, EncoderLayer.py, Attention.py, Convolution.py, ...)
# --------------------------------------------------------------------------------------
def rel_pos_table(n, d=192):
    """Layers/PositionalEncoding.py:90-117,129: rows are relative positions n-1 ... -(n-1)."""
    pos = torch.arange(n - 1, -n, -1, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    pe = torch.zeros(2 * n - 1, d)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def layer_norm(x, w, b):
    return F.layer_norm(x, (x.shape[-1],), w, b, 1e-12)  # Layers/LayerNorm.py:17


def ffn(x, sd, p):
    """Layers/MultiLayeredConv1d.py:50-51 with kernel size 1 == two Linear layers."""
    h = torch.relu(F.linear(x, sd[p + "w_1.weight"][:, :, 0], sd[p + "w_1.bias"]))
    return F.linear(h, sd[p + "w_2.weight"][:, :, 0], sd[p + "w_2.bias"])


def rel_attention(x, pe, sd, p, taps=None):
    """Layers/Attention.py:159-198 (+ rel_shift :138-157, forward_attention :66-92); x [N,192]."""
    n, h, dk = x.shape[0], 4, 48
    q = F.linear(x, sd[p + "linear_q.weight"], sd[p + "linear_q.bias"]).view(n, h, dk)
    k = F.linear(x, sd[p + "linear_k.weight"], sd[p + "linear_k.bias"]).view(n, h, dk).transpose(0, 1)
    v = F.linear(x, sd[p + "linear_v.weight"], sd[p + "linear_v.bias"]).view(n, h, dk).transpose(0, 1)
    pp = F.linear(pe, sd[p + "linear_pos.weight"]).view(2 * n - 1, h, dk).transpose(0, 1)  # [h, 2n-1, dk]
    qu = (q + sd[p + "pos_bias_u"]).transpose(0, 1)  # [h, n, dk]
    qv = (q + sd[p + "pos_bias_v"]).transpose(0, 1)
    ac = qu @ k.transpose(1, 2)
    bd = qv @ pp.transpose(1, 2)  # [h, n, 2n-1]; column m is relative position n-1-m
    # rel_shift: bd'[i, j] = bd[i, n-1-i+j]
    idx = (n - 1) - torch.arange(n).unsqueeze(1) + torch.arange(n).unsqueeze(0)
    bd = torch.gather(bd, 2, idx.unsqueeze(0).expand(h, n, n))
    scores = (ac + bd) / math.sqrt(dk)
    if taps is not None:
        taps["scores"] = scores
    attn = torch.softmax(scores, dim=-1)
    ctx = (attn @ v).transpose(0, 1).reshape(n, h * dk)
    return F.linear(ctx, sd[p + "linear_out.weight"], sd[p + "linear_out.bias"])


def conv_module(x, sd, p, kernel):
    """Layers/Convolution.py:31-55: pw(192->384), GLU, depthwise k, BatchNorm(eval), Swish, pw."""
    y = x.t().unsqueeze(0)
    y = F.conv1d(y, sd[p + "pointwise_conv1.weight"], sd[p + "pointwise_conv1.bias"])
    y = F.glu(y, dim=1)
    y = F.conv1d(y, sd[p + "depthwise_conv.weight"], sd[p + "depthwise_conv.bias"], padding=(kernel - 1) // 2, groups=192)
    y = F.batch_norm(y, sd[p + "norm.running_mean"], sd[p + "norm.running_var"], sd[p + "norm.weight"], sd[p + "norm.bias"],
                     False, 0.0, 1e-5)
    y = y * torch.sigmoid(y)  # Layers/Swish.py:18
    y = F.conv1d(y, sd[p + "pointwise_conv2.weight"], sd[p + "pointwise_conv2.bias"])
    return y[0].t()


def conformer_block(x, pe, sd, p, kernel, taps=None):
    """Layers/EncoderLayer.py:62-144 (macaron, pre-norm, cnn module, final norm)."""
    x = x + 0.5 * ffn(layer_norm(x, sd[p + "norm_ff_macaron.weight"], sd[p + "norm_ff_macaron.bias"]), sd, p + "feed_forward_macaron.")
    x = x + rel_attention(layer_norm(x, sd[p + "norm_mha.weight"], sd[p + "norm_mha.bias"]), pe, sd, p + "self_attn.", taps)
    x = x + conv_module(layer_norm(x, sd[p + "norm_conv.weight"], sd[p + "norm_conv.bias"]), sd, p + "conv_module.", kernel)
    x = x + 0.5 * ffn(layer_norm(x, sd[p + "norm_ff.weight"], sd[p + "norm_ff.bias"]), sd, p + "feed_forward.")
    return layer_norm(x, sd[p + "norm_final.weight"], sd[p + "norm_final.bias"])


def conformer_stack(x, sd, prefix, kernel, taps=None, tap_name=None):
    """Layers/Conformer.py:116-118: x*sqrt(192), rel-pos table, 6 blocks."""
    x = x * math.sqrt(192.0)  # PositionalEncoding.py:128
    pe = rel_pos_table(x.shape[0])
    for b in range(6):
        t = {} if (taps is not None and b == 0) else None
        x = conformer_block(x, pe, sd, f"{prefix}.encoders.{b}.", kernel, t)
        if taps is not None:
            taps[f"{tap_name}_block{b}"] = x
            if t:
                taps[f"{tap_name}_scores0"] = t["scores"]
    return x


def encoder(text, utt_emb, lang_id, sd, taps=None):
    """Layers/Conformer.py:92-134 for the encoder instance (InferenceToucanTTS.py:87-105)."""
    x = F.linear(torch.tanh(F.linear(text, sd["encoder.embed.0.weight"], sd["encoder.embed.0.bias"])),
                 sd["encoder.embed.2.weight"], sd["encoder.embed.2.bias"])
    if lang_id is not None:
        x = x + sd["encoder.language_embedding.weight"][int(lang_id)]
    if taps is not None:
        taps["enc_embed"] = x
    x = conformer_stack(x, sd, "encoder", 7, taps, "enc")
    x = layer_norm(x, sd["encoder.output_norm.weight"], sd["encoder.output_norm.bias"])
    if "encoder.hs_emb_projection.weight" not in sd:  # single-speaker variant (utt_embed=None): Conformer.py:125-126 is skipped
        return x
    e = F.normalize(utt_emb.unsqueeze(0))[0]  # Conformer.py:132 (idempotent with InferenceToucanTTS.py:202)
    x = F.linear(torch.cat([x, e.unsqueeze(0).expand(x.shape[0], -1)], dim=-1),
                 sd["encoder.hs_emb_projection.weight"], sd["encoder.hs_emb_projection.bias"])
    return x


# --------------------------------------------------------------------------------------
# variance adaptor
# --------------------------------------------------------------------------------------
def cln_mlp(e, sd, p):
    h = torch.tanh(F.linear(e, sd[p + "0.weight"], sd[p + "0.bias"]))
    h = torch.tanh(F.linear(h, sd[p + "2.weight"], sd[p + "2.bias"]))
    return F.linear(h, sd[p + "4.weight"], sd[p + "4.bias"])


def conditional_layer_norm(x, e, sd, p):
    """Layers/ConditionalLayerNorm.py:52-67; x [L,C]; NOTE divides by the variance, no eps/sqrt."""
    mean = x.mean(dim=-1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=-1, keepdim=True)
    return cln_mlp(e, sd, p + "W_scale.") * ((x - mean) / var) + cln_mlp(e, sd, p + "W_bias.")


def predictor(x, e, sd, prefix, n_layers, kernel):
    """Layers/VariancePredictor.py:65-80 and Layers/DurationPredictor.py:63-74; returns [L]."""
    h = x
    for i in range(n_layers):
        y = F.conv1d(h.t().unsqueeze(0), sd[f"{prefix}.conv.{i}.0.weight"], sd[f"{prefix}.conv.{i}.0.bias"], padding=(kernel - 1) // 2)
        if f"{prefix}.norms.{i}.weight" in sd:  # single-speaker variant: LayerNorm(n_chans, dim=1) (VariancePredictor.py:47-48)
            h = layer_norm(torch.relu(y)[0].t(), sd[f"{prefix}.norms.{i}.weight"], sd[f"{prefix}.norms.{i}.bias"])
        else:
            h = conditional_layer_norm(torch.relu(y)[0].t(), e, sd, f"{prefix}.norms.{i}.")
    return F.linear(h, sd[prefix + ".linear.weight"], sd[prefix + ".linear.bias"])[:, 0]


def duration_from_log(x):
    return torch.clamp(torch.round(x.exp() - 1.0), min=0).long()  # DurationPredictor.py:79


def scale_variance(seq, scale):
    """InferenceToucanTTS.py:333-343 (mean over non-zeros, every entry shifted, negatives clamped)."""
    if scale == 1.0:
        return seq
    avg = seq[seq != 0.0].mean()
    seq = (seq - avg) * scale + avg
    return torch.where(seq < 0.0, torch.zeros_like(seq), seq)


def control(text, pitch, energy, dur, duration_scale, pitch_scale, energy_scale, pause_scale):
    """InferenceToucanTTS.py:214-227."""
    pitch = torch.where(text[:, F_VOICED] == 0, torch.zeros_like(pitch), pitch)
    energy = torch.where(text[:, F_PHONEME] == 0, torch.zeros_like(energy), energy)
    dur = torch.where(text[:, F_WORD_BOUNDARY] == 1, torch.zeros_like(dur), dur)
    if pause_scale != 1.0:
        dur = torch.where(text[:, F_SILENCE] == 1, torch.round(dur.float() * pause_scale).long(), dur)
    if duration_scale != 1.0:
        dur = torch.round(dur.float() * duration_scale).long()
    return scale_variance(pitch, pitch_scale), scale_variance(energy, energy_scale), dur


def length_regulate(x, dur):
    """Layers/LengthRegulator.py:37-61 for a single utterance."""
    if int(dur.sum()) == 0:
        dur = torch.ones_like(dur)
    return torch.repeat_interleave(x, dur, dim=0)


# --------------------------------------------------------------------------------------
# PostNet + PostFlow
# --------------------------------------------------------------------------------------
def postnet(mel, sd):
    """Layers/PostNet.py:62-74; mel [T,80] -> residual [T,80]."""
    y = mel.t().unsqueeze(0)
    for i in range(5):
        y = F.conv1d(y, sd[f"conv_postnet.postnet.{i}.0.weight"], None, padding=2)
        y = F.group_norm(y, 32 if i < 4 else 20, sd[f"conv_postnet.postnet.{i}.1.weight"], sd[f"conv_postnet.postnet.{i}.1.bias"], 1e-5)
        if i < 4:
            y = torch.tanh(y)
    return y[0].t()


def glow_squeeze(x):
    """glow_utils.py:28-40: [C,T] -> [2C, T//2], channel = s*C + c, drops an odd last frame."""
    c, t = x.shape
    t2 = t // 2
    return x[:, : 2 * t2].reshape(c, t2, 2).permute(2, 0, 1).reshape(2 * c, t2)


def glow_unsqueeze(x):
    """glow_utils.py:43-53."""
    c2, t2 = x.shape
    return x.reshape(2, c2 // 2, t2).permute(1, 2, 0).reshape(c2 // 2, 2 * t2)


def wavenet(x, g, sd, p):
    """wavenet.py:89-122; x [192,T], g [384,T]."""
    cond = F.conv1d(g.unsqueeze(0), sd[p + "cond_layer.weight"], sd[p + "cond_layer.bias"])[0]
    out = torch.zeros_like(x)
    for i in range(4):
        a = F.conv1d(x.unsqueeze(0), sd[p + f"in_layers.{i}.weight"], sd[p + f"in_layers.{i}.bias"], padding=2)[0]
        a = a + cond[i * 384:(i + 1) * 384]
        acts = torch.tanh(a[:192]) * torch.sigmoid(a[192:])
        rs = F.conv1d(acts.unsqueeze(0), sd[p + f"res_skip_layers.{i}.weight"], sd[p + f"res_skip_layers.{i}.bias"])[0]
        if i < 3:
            x = x + rs[:192]
            out = out + rs[192:]
        else:
            out = out + rs
    return out


def postflow(mel, upsampled, z, sd, winv, taps=None):
    """Glow.py:342-391 reverse pass; mel [T,80], upsampled [T,192], z [80,T] (= 0.8*randn, :363)."""
    g = torch.cat([mel.t(), upsampled.t()], dim=0).unsqueeze(0)
    g = F.conv1d(g, sd["post_flow.g_proj.weight"], sd["post_flow.g_proj.bias"], padding=2)[0]
    if taps is not None:
        taps["glow_g"] = g
    x = glow_squeeze(z)
    g = glow_squeeze(g)
    for b in reversed(range(18)):
        pa, pc = f"post_flow.flows.{3 * b}.", f"post_flow.flows.{3 * b + 2}."
        # CouplingBlock reverse, Glow.py:248-269
        x0, x1 = x[:80], x[80:]
        h = F.conv1d(x0.unsqueeze(0), sd[pc + "start.weight"], sd[pc + "start.bias"])[0]
        h = wavenet(h, g, sd, pc + "wn.")
        o = F.conv1d(h.unsqueeze(0), sd[pc + "end.weight"], sd[pc + "end.bias"])[0]
        x = torch.cat([x0, (x1 - o[:80]) * torch.exp(-o[80:])], dim=0)
        # InvConvNear reverse, Glow.py:93-128 (channel regrouping :102-103, :126-127)
        c, t = x.shape
        y = x.reshape(2, c // 4, 2, t).permute(0, 2, 1, 3).reshape(4, c // 4, t)
        y = torch.einsum("on,ngt->ogt", winv[b], y)
        x = y.reshape(2, 2, c // 4, t).permute(0, 2, 1, 3).reshape(c, t)
        # ActNorm reverse, Glow.py:30-31
        x = (x - sd[pa + "bias"][0]) * torch.exp(-sd[pa + "logs"][0])
        if taps is not None and b in (17, 8, 0):
            taps[f"glow_z_after_block{b}"] = x
    return glow_unsqueeze(x).t()


# --------------------------------------------------------------------------------------
# acoustic model driver
# --------------------------------------------------------------------------------------
class AcousticOracle:
    """InferenceToucanTTS.ToucanTTS (forward :252-319, _forward :183-250) for one utterance."""

    def __init__(self, sd_np):
        self.sd = fold_weight_norm(to_torch(sd_np))
        self.winv = [invconv_inverse(self.sd, f"post_flow.flows.{3 * b + 1}.") for b in range(18)]

    @torch.inference_mode()
    def __call__(self, text, utt_emb, lang_id, z_noise=None, durations=None, pitch=None, energy=None,
                 duration_scaling_factor=1.0, pitch_variance_scale=1.0, energy_variance_scale=1.0,
                 pause_duration_scaling_factor=1.0, taps=None, run_postflow=True):
        sd = self.sd
        if "encoder.language_embedding.weight" not in sd:
            lang_id = None  # InferenceToucanTTS.py:196-197
        e = F.normalize(utt_emb.unsqueeze(0))[0] if utt_emb is not None else None
        enc = encoder(text, utt_emb, lang_id, sd, taps)
        p = predictor(enc, e, sd, "pitch_predictor", 7, 5) if pitch is None else pitch.reshape(-1).float()
        en = predictor(enc, e, sd, "energy_predictor", 2, 3) if energy is None else energy.reshape(-1).float()
        if durations is None:
            logd = predictor(enc, e, sd, "duration_predictor", 3, 3)
            d = duration_from_log(logd)
        else:
            logd, d = None, durations.long()
        if taps is not None:
            taps.update(enc_out=enc, pitch_raw=p.clone(), energy_raw=en.clone())
            if logd is not None:
                taps["log_dur"] = logd
        p, en, d = control(text, p, en, d, duration_scaling_factor, pitch_variance_scale, energy_variance_scale,
                           pause_duration_scaling_factor)
        enriched = enc + p.unsqueeze(1) * sd["pitch_embed.0.weight"][:, 0, 0] + sd["pitch_embed.0.bias"] \
            + en.unsqueeze(1) * sd["energy_embed.0.weight"][:, 0, 0] + sd["energy_embed.0.bias"]
        up = length_regulate(enriched, d)
        dec = conformer_stack(up, sd, "decoder", 31, taps, "dec")
        mel0 = F.linear(dec, sd["feat_out.weight"], sd["feat_out.bias"])
        mel1 = mel0 + postnet(mel0, sd)
        out = dict(durations=d, pitch=p, energy=en, decoded=mel0, refined=mel1, upsampled=up)
        if run_postflow:
            assert z_noise is not None, "PostFlow noise must be an explicit input (Glow.py:363 is stochastic)"
            out["mel"] = postflow(mel1, up, z_noise, sd, self.winv, taps)
        else:
            out["mel"] = mel1
        return out


# --------------------------------------------------------------------------------------
# vocoders
# --------------------------------------------------------------------------------------
UP_RATES = (8, 6, 4, 2)
UP_KERNELS = (16, 12, 8, 4)
RES_KERNELS = (3, 7, 11)
RES_DIL = (1, 3, 5)


def kaiser_sinc_filter(cutoff=0.25, half_width=0.3, k=12):
    """alias_free_torch (third party, ~=0.0.6) kaiser_sinc_filter1d - PARITY UNPINNED, see header."""
    half = k // 2
    a = 2.285 * (half - 1) * math.pi * (4 * half_width) + 7.95
    if a > 50.0:
        beta = 0.1102 * (a - 8.7)
    elif a >= 21.0:
        beta = 0.5842 * (a - 21.0) ** 0.4 + 0.07886 * (a - 21.0)
    else:
        beta = 0.0
    win = torch.kaiser_window(k, periodic=False, beta=beta, dtype=torch.float32)
    t = torch.arange(-half, half, dtype=torch.float32) + 0.5
    f = 2 * cutoff * win * torch.sinc(2 * cutoff * t)
    return f / f.sum()


def activation1d(x, act, filt):
    """alias_free_torch Activation1d(up 2, down 2, 12 taps) - PARITY UNPINNED; x [C,T]."""
    c = x.shape[0]
    w = filt.view(1, 1, -1).expand(c, 1, -1)
    y = F.pad(x.unsqueeze(0), (5, 5), mode="replicate")
    y = 2.0 * F.conv_transpose1d(y, w, stride=2, groups=c)[..., 15:-15]
    y = act(y)
    y = F.pad(y, (5, 6), mode="replicate")
    return F.conv1d(y, w, stride=2, groups=c)[0]


def snake_beta(x, alpha, beta):
    """BigVGAN/Snake.py:56-69 (log-scale alpha/beta)."""
    a = torch.exp(alpha).view(1, -1, 1)
    b = torch.exp(beta).view(1, -1, 1)
    return x + (1.0 / (b + 1e-9)) * torch.sin(x * a) ** 2


class VocoderOracle:
    """InferenceBigVGAN.BigVGAN.forward :72-95 / InferenceAvocodo.HiFiGANGenerator.forward :69-80."""

    def __init__(self, sd_np, kind):
        assert kind in ("bigvgan", "hifigan")
        self.kind = kind
        self.sd = fold_weight_norm(to_torch(sd_np))
        # a checkpoint that stores the Activation1d filter buffers (alias_free_torch registers them) overrides the restated design
        stored = sorted(k for k in self.sd if k.endswith("upsample.filter") or k.endswith("downsample.lowpass.filter"))
        self.filt = self.sd[stored[0]].reshape(-1).to(torch.float32) if stored else kaiser_sinc_filter()

    def _names(self):
        if self.kind == "bigvgan":
            return "conv_pre", "ups.{}.0", "resblocks.{}.", "convs1.{}", "convs2.{}", "conv_post"
        return "input_conv", "upsamples.{}.1", "blocks.{}.", "convs1.{}.1", "convs2.{}.1", "output_conv.1"

    @torch.inference_mode()
    def __call__(self, mel, taps=None):
        """mel [80,T] -> wav [384*T]."""
        sd = self.sd
        pre, ups, blk, c1, c2, post = self._names()
        big = self.kind == "bigvgan"
        x = F.conv1d(mel.unsqueeze(0), sd[pre + ".weight"], sd[pre + ".bias"], padding=3)
        for i, (u, k) in enumerate(zip(UP_RATES, UP_KERNELS)):
            if not big:
                x = F.leaky_relu(x, 0.1)  # InferenceAvocodo.py:38
            x = F.conv_transpose1d(x, sd[ups.format(i) + ".weight"], sd[ups.format(i) + ".bias"], stride=u, padding=(k - u) // 2)
            acc = None
            for j, kk in enumerate(RES_KERNELS):
                b = blk.format(3 * i + j)
                y = x
                for d, dil in enumerate(RES_DIL):  # AMP.py:51-60 / ResidualBlock.py:83-98
                    if big:
                        a = b + f"activations.{2 * d}.act."
                        t = activation1d(y[0], lambda v: snake_beta(v, sd[a + "alpha"], sd[a + "beta"]), self.filt).unsqueeze(0)
                    else:
                        t = F.leaky_relu(y, 0.1)
                    t = F.conv1d(t, sd[b + c1.format(d) + ".weight"], sd[b + c1.format(d) + ".bias"], padding=(kk - 1) // 2 * dil, dilation=dil)
                    if big:
                        a = b + f"activations.{2 * d + 1}.act."
                        t = activation1d(t[0], lambda v: snake_beta(v, sd[a + "alpha"], sd[a + "beta"]), self.filt).unsqueeze(0)
                    else:
                        t = F.leaky_relu(t, 0.1)
                    t = F.conv1d(t, sd[b + c2.format(d) + ".weight"], sd[b + c2.format(d) + ".bias"], padding=(kk - 1) // 2)
                    y = t + y
                acc = y if acc is None else acc + y
            x = acc / 3.0
            if taps is not None:
                taps[f"voc_stage{i}"] = x[0]
        if big:
            x = activation1d(x[0], lambda v: snake_beta(v, sd["activation_post.act.alpha"], sd["activation_post.act.beta"]), self.filt).unsqueeze(0)
        else:
            x = F.leaky_relu(x, 0.01)  # InferenceAvocodo.py:53 (torch default slope)
        x = torch.tanh(F.conv1d(x, sd[post + ".weight"], sd[post + ".bias"], padding=3))
        return x.reshape(-1)


# --------------------------------------------------------------------------------------
# Style embedding (GST) and the log-mel front end behind set_utterance_embedding(path)
# (ToucanTTSInterface.py:103-114)
# --------------------------------------------------------------------------------------
class StyleOracle:
    """StyleEmbedding.forward (StyleEmbedding.py:21-57) + StyleEncoder / ReferenceEncoder / StyleTokenLayer (GST.py:59-243)."""

    def __init__(self, state_dict):
        self.sd = {k: (v if torch.is_tensor(v) else torch.from_numpy(v)) for k, v in state_dict.items()}

    @staticmethod
    def tile_to_812(spec):
        """StyleEmbedding.py:40-52: repeat the spectrogram (at least once) until it has 812 frames, keep the first 812."""
        spec = spec.repeat((2, 1))
        while len(spec) < 812:
            spec = spec.repeat((2, 1))
        return spec[:812]

    def reference_encoder(self, spec):
        """GST.py:144-161: 8 x (Conv2d k3 s2 p1 no bias, BatchNorm2d eval, ReLU) on [1, 1, 812, 80], then a 2-layer GRU over the
        remaining time steps; the last hidden state of the top layer is the reference embedding."""
        sd = self.sd
        h = spec[None, None]
        for i in range(8):
            p, q = f"gst.ref_enc.convs.{3 * i}", f"gst.ref_enc.convs.{3 * i + 1}"
            h = F.conv2d(h, sd[p + ".weight"], None, stride=2, padding=1)
            h = F.batch_norm(h, sd[q + ".running_mean"], sd[q + ".running_var"], sd[q + ".weight"], sd[q + ".bias"], False, 0.0, 1e-5)
            h = torch.relu(h)
        hs = h.transpose(1, 2).contiguous().view(1, h.shape[2], -1)[0]  # [T', C * F']
        x = hs
        for layer in range(2):  # torch.nn.GRU equations, gate order r, z, n
            w_ih, w_hh = sd[f"gst.ref_enc.gst.weight_ih_l{layer}"], sd[f"gst.ref_enc.gst.weight_hh_l{layer}"]
            b_ih, b_hh = sd[f"gst.ref_enc.gst.bias_ih_l{layer}"], sd[f"gst.ref_enc.gst.bias_hh_l{layer}"]
            hdim = w_hh.shape[1]
            hcur = torch.zeros(hdim)
            outs = []
            for t in range(x.shape[0]):
                gi, gh = w_ih @ x[t] + b_ih, w_hh @ hcur + b_hh
                r = torch.sigmoid(gi[:hdim] + gh[:hdim])
                z = torch.sigmoid(gi[hdim:2 * hdim] + gh[hdim:2 * hdim])
                n = torch.tanh(gi[2 * hdim:] + r * gh[2 * hdim:])
                hcur = (1 - z) * n + z * hcur
                outs.append(hcur)
            x = torch.stack(outs)
        return x[-1]

    def style_tokens(self, ref):
        """GST.py:205-219 + Layers/Attention.py:42-92: one query (the reference embedding) over tanh(2000 style tokens), 8 heads of 8."""
        sd = self.sd
        m = "gst.stl.mha."
        toks = torch.tanh(sd["gst.stl.gst_embs"])
        q = (sd[m + "linear_q.weight"] @ ref + sd[m + "linear_q.bias"]).view(8, 8)
        k = (toks @ sd[m + "linear_k.weight"].t() + sd[m + "linear_k.bias"]).view(-1, 8, 8)
        v = (toks @ sd[m + "linear_v.weight"].t() + sd[m + "linear_v.bias"]).view(-1, 8, 8)
        scores = torch.einsum("hd,nhd->hn", q, k) / math.sqrt(8)
        ctx = torch.einsum("hn,nhd->hd", torch.softmax(scores, dim=-1), v).reshape(-1)
        return sd[m + "linear_out.weight"] @ ctx + sd[m + "linear_out.bias"]

    def __call__(self, spec, return_ref=False):
        ref = self.reference_encoder(self.tile_to_812(spec))
        return (self.style_tokens(ref), ref) if return_ref else self.style_tokens(ref)


def mel_filterbank(sr=16000, n_fft=1024, n_mels=80, fmin=40.0, fmax=8000.0):
    """librosa.filters.mel with its defaults (Slaney scale, htk=False, norm='slaney') restated in numpy float64.  Third party
    (librosa is neither vendored nor installed): PARITY UNPINNED, restated from the package's documented algorithm -
    linear below 1 kHz (200/3 Hz per mel), logarithmic above (step log(6.4)/27), triangles normalised to constant area."""
    import numpy as np

    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        lin = f / (200.0 / 3)
        return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-10) / 1000.0) / (np.log(6.4) / 27.0), lin)

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), m * (200.0 / 3))

    fft_f = np.linspace(0.0, sr / 2.0, n_fft // 2 + 1)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fft_f[None, :]
    w = np.maximum(0.0, np.minimum(-ramps[:-2] / fdiff[:-1, None], ramps[2:] / fdiff[1:, None]))
    return (w * (2.0 / (mel_f[2:] - mel_f[:-2]))[:, None]).astype(np.float32)  # [n_mels, n_fft/2 + 1], float32 like librosa


def logmel(audio, sr=16000, n_fft=1024, hop=256, n_mels=80, fmin=40.0, fmax=8000.0, eps=1e-10):
    """AudioPreprocessor.logmelfilterbank (AudioPreprocessor.py:96-117): librosa.stft(n_fft 1024, hop 256, hann, centred with
    reflect padding) -> magnitude -> mel basis -> log10(max(eps, .)); returns [frames, n_mels] (the reference returns the transpose).
    float64 numpy restatement; the librosa pieces are PARITY UNPINNED (see mel_filterbank)."""
    import numpy as np
    x = np.asarray(audio, dtype=np.float64)
    x = np.pad(x, n_fft // 2, mode="reflect")
    n = np.arange(n_fft)
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)  # scipy.signal.get_window("hann", n_fft, fftbins=True)
    frames = 1 + (len(x) - n_fft) // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(frames)[:, None]
    spec = np.abs(np.fft.rfft(x[idx] * win[None, :], axis=1))
    return np.log10(np.maximum(eps, spec @ mel_filterbank(sr, n_fft, n_mels, fmin, fmax).astype(np.float64).T)).astype(np.float32)
