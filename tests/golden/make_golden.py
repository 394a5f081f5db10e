"""Golden-vector generator.  Runs ONLY in the survey/build container (needs /root/reference).

It imports the reference's own inference classes (with empty stand-ins for third-party packages
that are imported at module top level but unused on this path), loads the build's seeded fixture
weights through the reference's strict ``load_state_dict``, runs the reference forward on seeded
synthetic inputs and stores inputs + outputs (+ intermediate taps) as small ``.npz`` fixtures.
While doing so it asserts that ``oracle/toucan_oracle.py`` reproduces every stored tensor, i.e.
the committed fixtures pin the oracle.

It also captures the 62-dim articulatory feature table (data) into
``ims-toucan-prosody-variance_amd/data/phone_table.json``.

Nothing from the reference is copied: fixtures are inputs and expected outputs only.

    python tests/golden/make_golden.py            # regenerate everything (about a minute)
"""
import json
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

import torch  # noqa: E402

torch.set_num_threads(8)

from oracle import toucan_oracle as orc  # noqa: E402


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Dummy:
    def __init__(self, *a, **k):
        pass


def install_stubs():
    lib = _stub("librosa")
    lib.display = _stub("librosa.display")
    lib.core = _stub("librosa.core")
    _stub("phonemizer")
    _stub("phonemizer.backend", EspeakBackend=_Dummy)
    _stub("dragonmapper")
    _stub("dragonmapper.transcriptions", pinyin_to_ipa=lambda x: x)
    _stub("pypinyin", pinyin=lambda *a, **k: [])
    _stub("soundfile")
    _stub("sounddevice")
    _stub("pyloudnorm")
    _stub("torchaudio")
    _stub("torchaudio.transforms", Resample=_Dummy)

    # alias_free_torch is absent: stand-in restating its published Activation1d (PARITY UNPINNED for
    # the filter itself - see oracle/toucan_oracle.py header).  AMP.py:8 relies on the star import to
    # bring ``torch`` and ``nn`` into scope.
    class Activation1d(torch.nn.Module):
        def __init__(self, activation, up_ratio=2, down_ratio=2, up_kernel_size=12, down_kernel_size=12):
            super().__init__()
            self.act = activation
            self._filt = orc.kaiser_sinc_filter()

        def forward(self, x):
            return torch.stack([orc.activation1d(xi, self.act, self._filt) for xi in x])

    _stub("alias_free_torch", Activation1d=Activation1d, torch=torch, nn=torch.nn)


install_stubs()
import ims_toucan_prosody_variance_amd  # noqa: E402,F401
from ims_toucan_prosody_variance_amd import fixture_weights as fw  # noqa: E402


def dump_phone_table():
    from Preprocessing.articulatory_features import generate_feature_table
    table = generate_feature_table()
    out = {k: "".join(str(int(b)) for b in v) for k, v in table.items()}
    assert all(len(v) == 62 for v in out.values())
    path = os.path.join(REPO, "ims-toucan-prosody-variance_amd", "data", "phone_table.json")
    with open(path, "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=False, indent=0, sort_keys=True)
    print(f"phone table: {len(out)} symbols -> {path}")


def close(name, a, b, tol):
    a = a.detach().float().reshape(-1)
    b = b.detach().float().reshape(-1)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    err = (a - b).abs().max().item() if a.numel() else 0.0
    scale = max(1.0, b.abs().max().item()) if b.numel() else 1.0
    assert err <= tol * scale, f"oracle != reference at {name}: max abs err {err:.3e} (scale {scale:.2f})"
    return err


class RefAcoustic:
    def __init__(self, sd_np, **model_kwargs):
        """model_kwargs: e.g. lang_embs=None / utt_embed_dim=None for the checkpoint variants of ToucanTTSInterface.py:55-63."""
        from InferenceInterfaces.InferenceArchitectures.InferenceToucanTTS import ToucanTTS
        sd = {k: torch.from_numpy(np.array(v)) for k, v in sd_np.items()}
        self.m = ToucanTTS(weights=sd, **model_kwargs)  # strict load validates the fixture schema
        with torch.no_grad():
            self.m.store_inverse_all()
        self.m.eval()
        self.taps = {}
        m = self.m
        for b in range(6):
            m.encoder.encoders[b].register_forward_hook(self._hook(f"enc_block{b}", lambda o: o[0][0][0]))
            m.decoder.encoders[b].register_forward_hook(self._hook(f"dec_block{b}", lambda o: o[0][0][0]))
        m.encoder.register_forward_hook(self._hook("enc_out", lambda o: o[0][0]))
        m.pitch_predictor.register_forward_hook(self._hook("pitch_raw", lambda o: o[0, :, 0].clone()))
        m.energy_predictor.register_forward_hook(self._hook("energy_raw", lambda o: o[0, :, 0].clone()))
        m.duration_predictor.linear.register_forward_hook(self._hook("log_dur", lambda o: o[0, :, 0]))
        m.feat_out.register_forward_hook(self._hook("decoded", lambda o: o[0]))
        m.length_regulator.register_forward_hook(self._hook("upsampled", lambda o: o[0]))
        m.conv_postnet.register_forward_hook(self._hook("postnet_res", lambda o: o[0].t()))
        m.post_flow.g_proj.register_forward_hook(self._hook("glow_g", lambda o: o[0]))
        for b in (17, 8, 0):
            m.post_flow.flows[3 * b].register_forward_hook(self._hook(f"glow_z_after_block{b}", lambda o: o[0][0]))

    def _hook(self, name, fn):
        def h(mod, inp, out):
            self.taps[name] = fn(out).detach().clone()
        return h

    def __call__(self, text, utt_emb, lang_id, z, **kw):
        """z [80,T'] is injected in place of torch.randn at Glow.py:363 (already scaled by 0.8 there,
        so we hand randn the unscaled noise)."""
        self.taps = {}
        real_randn = torch.randn
        used = {}

        def fake_randn(*shape, **k):
            shp = tuple(shape[0]) if len(shape) == 1 and not isinstance(shape[0], int) else tuple(shape)
            assert shp == (1, 80, z.shape[1]), (shp, z.shape)
            used["ok"] = True
            return (z / 0.8).reshape(shp).clone()

        torch.randn = fake_randn
        try:
            mel, dur, pitch, energy = self.m(text, utterance_embedding=utt_emb, lang_id=torch.LongTensor([lang_id]),
                                             return_duration_pitch_energy=True, **kw)
        finally:
            torch.randn = real_randn
        assert used.get("ok")
        return mel, dur, pitch, energy


def ref_vocoder(kind, sd_np):
    path = f"/tmp/_golden_{kind}.pt"
    torch.save({"generator": {k: torch.from_numpy(np.array(v)) for k, v in sd_np.items()}}, path)
    if kind == "bigvgan":
        from InferenceInterfaces.InferenceArchitectures.InferenceBigVGAN import BigVGAN
        m = BigVGAN(path_to_weights=path)
    else:
        from InferenceInterfaces.InferenceArchitectures.InferenceAvocodo import HiFiGANGenerator
        m = HiFiGANGenerator(path_to_weights=path)
    m.remove_weight_norm()
    m.eval()
    taps = {}
    return m, taps


def gold_prosody(u, L):
    """Seeded gold pitch / energy curves of the UtteranceCloner kind (UtteranceCloner.py:147-194): positive values on every
    phoneme - also on unvoiced ones, silences and word boundaries, which the control loop (InferenceToucanTTS.py:214-222) must
    zero - so the fixture exercises the zeroing of GOLD values and the variance scaling on top of them."""
    p = 0.2 + 1.6 * fw.uniform01(f"utt{u}.gold_pitch", L, 5000 + u)
    e = 0.1 + 2.0 * fw.uniform01(f"utt{u}.gold_energy", L, 6000 + u)
    return p.astype(np.float32), e.astype(np.float32)


def main(only=None):
    if only is None:
        dump_phone_table()
    from ims_toucan_prosody_variance_amd import synthetic as syn

    ac_sd = fw.acoustic_state_dict()
    hf_sd = fw.hifigan_state_dict()
    bv_sd = fw.bigvgan_state_dict()
    ref_ac = RefAcoustic(ac_sd)
    orc_ac = orc.AcousticOracle(ac_sd)
    ref_hf, _ = ref_vocoder("hifigan", hf_sd)
    ref_bv, _ = ref_vocoder("bigvgan", bv_sd)
    orc_hf = orc.VocoderOracle(hf_sd, "hifigan")
    orc_bv = orc.VocoderOracle(bv_sd, "bigvgan")
    out_dir = os.path.dirname(os.path.abspath(__file__))

    def dur_guess(u, L, feats, mode):
        if mode == "gold5":
            return syn.gold_durations(feats)
        if mode == "ragged":
            return syn.ragged_durations(u, feats)
        return None

    cases = [
        # name, utt id, L, duration mode, control scales, full taps?
        ("L7_pred", 7, 7, None, {}, True),
        ("L20_pred", 20, 20, None, {}, True),
        ("L20_ctrl", 21, 20, None, dict(duration_scaling_factor=1.2, pitch_variance_scale=1.3, energy_variance_scale=0.7,
                                        pause_duration_scaling_factor=1.2), True),
        ("L20_gold_odd", 22, 20, "ragged", dict(pitch_variance_scale=0.7), True),
        # the fork's namesake path: gold durations AND gold pitch / energy (InferenceToucanTTS.py:209-210) with every scale != 1
        ("L20_gold_prosody", 23, 20, "ragged", dict(pitch_variance_scale=1.4, energy_variance_scale=0.6,
                                                    pause_duration_scaling_factor=1.5, duration_scaling_factor=1.1), True),
        ("L128_gold5", 0, 128, "gold5", {}, False),
        ("R128", 100, 128, "ragged", {}, False),
        ("R97", 101, 97, "ragged", {}, False),
        ("R64", 102, 64, "ragged", {}, False),
        ("R20", 103, 20, "ragged", {}, False),
    ]
    summary = {}
    if only is not None:
        with open(os.path.join(out_dir, "SUMMARY.json")) as f:
            summary = json.load(f)
    for name, u, L, mode, ctrl, full in cases:
        if only is not None and name not in only:
            continue
        feats = syn.utterance_features(u, L)
        emb = syn.utterance_embedding(u)
        text = torch.from_numpy(feats)
        e = torch.from_numpy(emb)
        durs = dur_guess(u, L, feats, mode)
        if name == "L20_gold_odd":  # force an odd frame count: the Glow squeeze drops the last frame
            if int(durs.sum()) % 2 == 0:
                durs[1] += 1
        kw = dict(ctrl)
        if durs is not None:
            kw["durations"] = torch.from_numpy(durs)
        gp = ge = None
        if name == "L20_gold_prosody":
            gp, ge = gold_prosody(u, L)
            assert (feats[:, 61] == 0).any() and (feats[:, 16] == 1).any() and (feats[:, 21] == 1).any()  # unvoiced, silence, boundary
            kw["pitch"], kw["energy"] = torch.from_numpy(gp), torch.from_numpy(ge)
        # frame count is only known after the duration stage: run once with a long noise buffer
        if durs is None:
            probe = orc_ac(text, e, syn.LANG_EN, run_postflow=False, **{k: v for k, v in kw.items()})
            T = int(probe["durations"].sum())
        else:
            probe = orc_ac(text, e, syn.LANG_EN, run_postflow=False, **kw)
            T = int(probe["durations"].sum())
        z = torch.from_numpy(syn.postflow_noise(u, T))
        with torch.inference_mode():
            # the reference writes its zeroing into the tensors it was handed (InferenceToucanTTS.py:216-222 index-assign into
            # views of the gold arguments) and wants gold pitch / energy as [L, 1] (it transposes them at :230-231): hand it copies
            rkw = {k: (v.clone().reshape(-1, 1) if k in ("pitch", "energy") else v.clone() if torch.is_tensor(v) else v) for k, v in kw.items()}
            mel_r, dur_r, pitch_r, energy_r = ref_ac(text, e, syn.LANG_EN, z, **rkw)
        rt = dict(ref_ac.taps)
        ot = {}
        o = orc_ac(text, e, syn.LANG_EN, z_noise=z, taps=ot, **kw)
        errs = {}
        errs["mel"] = close("mel", o["mel"], mel_r, 2e-5)
        assert torch.equal(o["durations"], dur_r.reshape(-1)), (o["durations"], dur_r)
        errs["pitch"] = close("pitch", o["pitch"], pitch_r, 1e-5)
        errs["energy"] = close("energy", o["energy"], energy_r, 1e-5)
        for b in range(6):
            errs[f"enc_block{b}"] = close(f"enc_block{b}", ot[f"enc_block{b}"], rt[f"enc_block{b}"], 1e-5)
            errs[f"dec_block{b}"] = close(f"dec_block{b}", ot[f"dec_block{b}"], rt[f"dec_block{b}"], 1e-5)
        close("enc_out", ot["enc_out"], rt["enc_out"], 1e-5)
        close("upsampled", o["upsampled"], rt["upsampled"], 1e-5)
        close("postnet", o["refined"] - o["decoded"], rt["postnet_res"], 1e-5)
        close("glow_g", ot["glow_g"], rt["glow_g"], 1e-5)
        for b in (17, 8, 0):
            close(f"glow_z{b}", ot[f"glow_z_after_block{b}"], rt[f"glow_z_after_block{b}"], 2e-5)
        if "pitch_raw" in rt:
            close("pitch_raw", ot["pitch_raw"], rt["pitch_raw"], 1e-5)
            close("energy_raw", ot["energy_raw"], rt["energy_raw"], 1e-5)
        if "log_dur" in rt and "log_dur" in ot:
            close("log_dur", ot["log_dur"], rt["log_dur"], 1e-5)

        mel_in = mel_r.t().contiguous()  # [80, T']
        with torch.inference_mode():
            wav_hf_r = ref_hf(mel_in)
            wav_bv_r = ref_bv(mel_in)
        vt_h, vt_b = {}, {}
        wav_hf_o = orc_hf(mel_in, vt_h)
        wav_bv_o = orc_bv(mel_in, vt_b)
        errs["wav_hifigan"] = close("wav_hifigan", wav_hf_o, wav_hf_r, 1e-4)
        errs["wav_bigvgan"] = close("wav_bigvgan", wav_bv_o, wav_bv_r, 1e-4)

        g = dict(text=feats, utt_emb=emb, lang_id=np.int64(syn.LANG_EN), z=z.numpy(),
                 ctrl=json.dumps(ctrl), mel=mel_r.numpy(), durations=dur_r.reshape(-1).numpy(),
                 pitch=pitch_r.reshape(-1).numpy(), energy=energy_r.reshape(-1).numpy(),
                 decoded=rt["decoded"].numpy(),
                 enc_out=rt["enc_out"].numpy())
        if durs is not None:
            g["gold_durations"] = durs
        if gp is not None:
            g["gold_pitch"], g["gold_energy"] = gp, ge
        if full:
            for k, v in rt.items():
                g["tap_" + k] = v.numpy()
            g["wav_hifigan"] = wav_hf_r.numpy()
            g["wav_bigvgan"] = wav_bv_r.numpy()
            if L <= 7:  # the coarsest vocoder stage of the shortest case is small enough to keep
                g["tap_hifigan_stage0"] = vt_h["voc_stage0"].numpy()
                g["tap_bigvgan_stage0"] = vt_b["voc_stage0"].numpy()
        else:
            n = 8192
            g["wav_hifigan_head"] = wav_hf_r[:n].numpy()
            g["wav_hifigan_tail"] = wav_hf_r[-n:].numpy()
            g["wav_bigvgan_head"] = wav_bv_r[:n].numpy()
            g["wav_bigvgan_tail"] = wav_bv_r[-n:].numpy()
            g["wav_len"] = np.int64(wav_hf_r.numel())
            # checksums let a full-length run be compared without shipping a megabyte per case
            g["wav_hifigan_abs_sum"] = np.float64(wav_hf_r.double().abs().sum().item())
            g["wav_bigvgan_abs_sum"] = np.float64(wav_bv_r.double().abs().sum().item())
        np.savez_compressed(os.path.join(out_dir, f"{name}.npz"), **g)
        summary[name] = dict(L=L, T=int(T), T_out=int(mel_r.shape[0]), mel_abs_mean=float(mel_r.abs().mean()),
                             wav_hf_abs_mean=float(wav_hf_r.abs().mean()), wav_bv_abs_mean=float(wav_bv_r.abs().mean()),
                             dur_min=int(dur_r.min()), dur_max=int(dur_r.max()),
                             oracle_vs_reference_max_abs_err=errs)
        print(name, json.dumps(summary[name])[:400])
    with open(os.path.join(out_dir, "SUMMARY.json"), "w") as f:
        json.dump(summary, f, indent=1)


if __name__ == "__main__":
    main(only=sys.argv[1:] or None)  # e.g. `make_golden.py L20_gold_prosody`: add one case, leave the other fixtures as they are
