"""Host side of the stage API (include/toucan_tts.h, csrc/pipeline.hip): packs the weights with the same code the
Python-sequenced engines use (engine.py / packing.py), uploads them into a ``TtsHandle`` and runs a ragged batch through
``tts_encoder`` ... ``tts_vocoder_*`` - a dozen C calls per pass instead of ~500 kernel-level ones.

This is the product path of ``ToucanTTSInterface`` on a GPU.  The Python-sequenced engines stay for the tap-by-tap parity
tests, for HIP-graph capture and for the CPU host-logic tests that drive the numpy ABI emulator.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import capi, engine, packing
from .ragged import Ragged

_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2, torch.uint8: 4}
_PRED = (("pitch_predictor", "pitch", 7), ("energy_predictor", "energy", 2), ("duration_predictor", "duration", 3))
_NORMS = ("norm_ff_macaron", "norm_mha", "norm_conv", "norm_ff", "norm_final")


class NativePipeline:
    """One handle per (process, device): acoustic model + (optionally) one vocoder."""

    def __init__(self, acoustic_sd, vocoder_sd=None, vocoder_kind=None, device="cuda", precision="f32", pmax=1024, vocoder_precision=None):
        """precision: of the acoustic model ("f32" | "bf16" | "f16").  vocoder_precision (default: the same): a different one gives
        the vocoder its own handle - e.g. precision="f32", vocoder_precision="f16": the mel keeps the exact-parity arithmetic
        (mel L1 <= 1e-5 against the reference) and the vocoder, the bulk of the work, runs on the 16-bit matrix cores."""
        self.lib = capi.lib()
        if not isinstance(self.lib, C.CDLL):
            raise capi.ToucanHipError("the stage API needs the real libtoucan_hip.so (the test emulator only restates kernels)")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise capi.ToucanHipError(f"device {str(self.device)!r}: libtoucan_hip.so has no CPU path - use device='cuda'")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.precision, _, compute, self.dt16 = engine.precision_of(False, precision)
        self.vocoder_precision, _, compute_v, _ = engine.precision_of(False, precision if vocoder_precision is None else vocoder_precision)
        split = vocoder_sd is not None and self.vocoder_precision != self.precision
        ac = engine.AcousticEngine(acoustic_sd, "cpu", precision=self.precision, pack_only=True)
        self.multilingual, self.multispeaker = ac.multilingual, ac.multispeaker
        voc = None
        self.kind = vocoder_kind
        self._streams = None  # (acoustic, vocoder) stream pair of forward_pipelined, made on first use
        if vocoder_sd is not None:
            assert vocoder_kind in ("hifigan", "bigvgan")
            voc = engine.VocoderEngine(vocoder_sd, vocoder_kind, "cpu", precision=self.vocoder_precision, pack_only=True)
        kind_code = {None: 0, "hifigan": 1, "bigvgan": 2}[vocoder_kind]
        post_b = float(voc.post_b) if voc is not None else 0.0
        stb = int(os.environ.get("TOUCAN_SMALL_TILE_BLOCKS", "0"))  # same A/B switch as engine.Ops (0: the library's default, 1536)
        cfg = capi.TtsConfig(int(self.multilingual), int(self.multispeaker), 0 if split else kind_code, compute, stb, post_b)
        self.h = C.c_void_p()
        self.h_voc = self.h  # the handle the vocoder entries are called on (its own when the precisions differ)
        with torch.cuda.device(self.device):
            capi.check(self.lib.tts_create(C.byref(cfg), C.byref(self.h)), "tts_create")
            self._target = self.h  # the handle _load() uploads to
            self._upload_acoustic(ac)
            if split:
                self.h_voc = C.c_void_p()
                cfg_v = capi.TtsConfig(int(self.multilingual), int(self.multispeaker), kind_code, compute_v, stb, post_b)
                capi.check(self.lib.tts_create(C.byref(cfg_v), C.byref(self.h_voc)), "tts_create (vocoder)")
            else:
                self.h_voc = self.h
            if voc is not None:
                self._target = self.h_voc
                self._upload_vocoder(voc)
                self._target = self.h
            self._pmax = 0
            self._ensure_pe(pmax)

    def __del__(self):
        h, hv = getattr(self, "h", None), getattr(self, "h_voc", None)
        for handle in ([hv] if (hv is not None and hv is not h) else []) + [h]:
            if handle is not None and handle.value:
                try:
                    self.lib.tts_destroy(handle)
                except Exception:
                    pass
        self.h = self.h_voc = None

    # ---- weight upload ------------------------------------------------------------------------------------------------
    def _load(self, name, t):
        t = t.detach().cpu().contiguous()
        if t.dtype == torch.int32:
            code, raw = 3, t.numpy()
        elif t.dtype == torch.int16:  # (raw 16-bit patterns: a kernel's own fragment order, e.g. packing.pack_ffn)
            code, raw = 1, t.numpy()
        else:
            code = _DT[t.dtype]
            raw = t.view(torch.int16).numpy() if t.dtype in (torch.bfloat16, torch.float16) else t.numpy()
        shape = (C.c_int64 * max(1, t.dim()))(*([int(s) for s in t.shape] or [1]))
        capi.check(self.lib.tts_load_weights(self._target, name.encode(), raw.ctypes.data_as(C.c_void_p), shape, max(1, t.dim()), code),
                   f"tts_load_weights({name})")

    def _conv(self, name, cw):
        self._load(name + ".w", cw.w)
        if cw.w16 is not None:
            self._load(name + ".w16", cw.w16)
        if cw.bias is not None:
            self._load(name + ".bias", cw.bias)
        meta = [cw.mode, cw.taps, cw.dil, cw.pad_left, cw.cin, cw.cin_pad, cw.cout, cw.wn, cw.half_pad, cw.tile_rows, cw.small_tile_rows,
                int(cw.small_only), cw.n_tile, cw.compute16, cw.algo_taps, 0]
        self._load(name + ".meta", torch.tensor(meta, dtype=torch.int32))

    def _upload_acoustic(self, ac):
        self._conv("embed0", ac.embed0)
        self._conv("embed2", ac.embed2)
        if ac.lang_table is not None:
            self._load("lang_table", ac.lang_table)
        for stack, cw in (("enc", ac.enc), ("dec", ac.dec)):
            for b, blk in enumerate(cw.blocks):
                p = f"{stack}.{b}."
                for ln in _NORMS:
                    self._load(p + ln + ".g", blk[ln][0])
                    self._load(p + ln + ".b", blk[ln][1])
                for src, dst in (("feed_forward_macaron.w1", "ffm.w1"), ("feed_forward_macaron.w2", "ffm.w2"), ("feed_forward.w1", "ff.w1"),
                                 ("feed_forward.w2", "ff.w2"), ("qkv", "qkv"), ("out", "out"), ("pos", "pos"), ("pw1", "pw1"), ("pw2", "pw2")):
                    self._conv(p + dst, blk[src])
                for src, dst in (("feed_forward_macaron.fused", "ffm.fused"), ("feed_forward.fused", "ff.fused")):
                    if src in blk:
                        self._load(p + dst, blk[src])
                for v in ("u", "v", "dw_w", "dw_b"):
                    self._load(p + v, blk[v])
        self._load("out_norm.g", ac.out_norm[0])
        self._load("out_norm.b", ac.out_norm[1])
        if ac.multispeaker:
            self._conv("hs_h", ac.hs_h)
            self._conv("hs_e", ac.hs_e)
            self._load("cln_weights", ac.cln_weights)
        for src, dst, layers in _PRED:
            convs, norms, lin = ac.pred[src]
            for i, c in enumerate(convs):
                self._conv(f"{dst}.conv.{i}", c)
                if not ac.multispeaker:
                    self._load(f"{dst}.norm.{i}.g", norms[i][0])
                    self._load(f"{dst}.norm.{i}.b", norms[i][1])
            self._conv(f"{dst}.lin", lin)
        for n in ("pitch_w", "pitch_b", "energy_w", "energy_b"):
            self._load(n, getattr(ac, n))
        self._conv("feat_out", ac.feat_out)
        for i, (cw, g, b) in enumerate(ac.postnet):
            self._conv(f"postnet.{i}.conv", cw)
            self._load(f"postnet.{i}.g", g)
            self._load(f"postnet.{i}.b", b)
        self._conv("g_proj", ac.g_proj)
        for b, blk in enumerate(ac.flow):
            for n in ("start", "end", "cond"):
                self._conv(f"flow.{b}.{n}", blk[n])
            for n in ("winv", "an_bias", "an_logs"):
                self._load(f"flow.{b}.{n}", blk[n])
            if b % 4 == 0:  # in / res-skip layers are shared inside groups of 4 blocks (Glow.py:325-327)
                for i in range(4):
                    self._conv(f"flowgrp.{b // 4}.inl.{i}", blk["inl"][i])
                    self._conv(f"flowgrp.{b // 4}.res_skip.{i}", blk["res_skip"][i])

    def _upload_vocoder(self, voc):
        self._conv("voc.pre", voc.pre)
        for i in range(4):
            self._conv(f"voc.ups.{i}", voc.ups[i])
            for j in range(3):
                for dd in range(3):
                    c1, c2 = voc.blocks[i][j][dd]
                    p = f"voc.blk.{i}.{j}.{dd}"
                    self._conv(p + ".c1", c1)
                    self._conv(p + ".c2", c2)
                    if voc.kind == "bigvgan":
                        (a1, b1), (a2, b2) = voc.snakes[i][j][dd]
                        for n, t in (("a1", a1), ("b1", b1), ("a2", a2), ("b2", b2)):
                            self._load(f"{p}.{n}", t)
        self._load("voc.post_w", voc.post_w)
        if voc.kind == "bigvgan":
            self._load("voc.post_a", voc.post_snake[0])
            self._load("voc.post_b", voc.post_snake[1])
            self._load("voc.filt", voc.filt)
            self._load("voc.fir_tab", voc.fir_tab)

    def _ensure_pe(self, pmax):
        """The sinusoid table for relative positions -(pmax-1) .. pmax-1 (PositionalEncoding.py:90-117); the handle turns it
        into the per-block position tables.  Grows by re-upload when a batch has longer utterances."""
        if pmax <= self._pmax:
            return
        pmax = max(pmax, 2 * self._pmax, 256)
        self._load("pe", torch.from_numpy(packing.rel_pos_encoding(pmax)))
        self._pmax = pmax

    def profile(self, enable, select=None):
        """Roofline leg: HIP events around the launches of one matrix-core kernel class (None: all) inside the stage entries."""
        for handle in self._handles():
            capi.check(self.lib.tts_profile(handle, int(enable), None if select is None else select.encode()), "tts_profile")  # 2: per-shape conv classes

    def _handles(self):
        return [self.h] if self.h_voc is self.h else [self.h, self.h_voc]

    def profile_summary(self):
        """class -> dict(launches, total_ms, avg_us, flops_per_launch, bytes_per_launch, elems_per_launch, tflops) (waits for the events)."""
        out = {}
        name = C.create_string_buffer(128)
        ms, fl, by, el = C.c_double(), C.c_double(), C.c_double(), C.c_double()
        for handle in self._handles():
            for i in range(int(self.lib.tts_profile_count(handle))):
                capi.check(self.lib.tts_profile_read(handle, i, name, 128, C.byref(ms), C.byref(fl), C.byref(by), C.byref(el)), "tts_profile_read")
                s = out.setdefault(name.value.decode(), dict(launches=0, total_ms=0.0, flops=0.0, bytes=0.0, elems=0.0))
                s["launches"] += 1
                s["total_ms"] += ms.value
                s["flops"] += fl.value
                s["bytes"] += by.value
                s["elems"] += el.value
        for s in out.values():
            n = s["launches"]
            s["avg_us"] = 1e3 * s["total_ms"] / n
            s["flops_per_launch"], s["bytes_per_launch"], s["elems_per_launch"] = s["flops"] / n, s["bytes"] / n, s["elems"] / n
            s["tflops"] = s["flops"] / (s["total_ms"] * 1e-3) / 1e12 if s["total_ms"] > 0 else 0.0
        return out

    def workspace_bytes(self, B, Lmax, Tmax):
        """Upper bound (bytes) of the workspace a batch of that shape will claim - what a caller budgets HBM from."""
        return sum(int(self.lib.tts_workspace_bytes(handle, B, Lmax, Tmax)) for handle in self._handles())

    def table_stats(self):
        """(tile tables built into batch arenas, tile tables given a permanent allocation) over this pipeline's handles."""
        a = b = 0
        for handle in self._handles():
            x, y = C.c_int64(), C.c_int64()
            capi.check(self.lib.tts_table_stats(handle, C.byref(x), C.byref(y)), "tts_table_stats")
            a, b = a + x.value, b + y.value
        return a, b

    def workspace_claimed(self):
        """Bytes the handles' arenas hold right now."""
        return sum(int(self.lib.tts_workspace_claimed(handle)) for handle in self._handles())

    # ---- one ragged batch ---------------------------------------------------------------------------------------------
    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    @torch.inference_mode()
    def pack_inputs(self, texts, utt_embs, lang_ids=None, durations=None, pitch=None, energy=None):
        """The C ABI's input format: everything packed along the phoneme axis, on the device.  ``forward`` does this itself; a
        caller that reuses a batch (bench.py: inputs resident in HBM before the timed region) packs once and passes ``packed=``."""
        dev = self.device
        B = len(texts)
        Ls = [int(t.shape[0]) for t in texts]
        text = torch.cat([t.reshape(-1, 62).to(torch.float32) for t in texts], dim=0).to(dev).contiguous()
        emb = utt_embs.to(dev, torch.float32).reshape(B, 64).contiguous() if utt_embs is not None else None
        lang = None
        if self.multilingual and lang_ids is not None:
            lang = torch.tensor([int(i) for i in lang_ids], dtype=torch.int32).to(dev)
        cat = lambda lst, dt: None if lst is None else torch.cat([torch.as_tensor(v).reshape(-1).to(dt) for v in lst]).to(dev).contiguous()
        return dict(Ls=Ls, text=text, emb=emb, lang=lang, gp=cat(pitch, torch.float32), ge=cat(energy, torch.float32), gd=cat(durations, torch.int32))

    def squeeze_noise(self, z_noise, frame_counts):
        """Per-utterance noise [80, T_u] -> the squeezed packed layout tts_postflow takes ([total_frames / 2, 160], 2-aligned begins)."""
        rag_f = Ragged(frame_counts, self.device, align=2)
        rag_s = rag_f.halved()
        z_sq = torch.zeros(rag_f.total_rows // 2, 160, dtype=torch.float32, device=self.device)  # (incl. the row an odd last utterance leaves unused)
        for zu, b0, n in zip(z_noise, rag_s.begins, rag_s.lengths):
            z_sq[b0:b0 + n].copy_(torch.as_tensor(zu, dtype=torch.float32).t()[: 2 * n].reshape(n, 160))
        return z_sq

    @torch.inference_mode()
    def forward(self, texts, utt_embs, lang_ids=None, durations=None, pitch=None, energy=None, z_noise=None, duration_scaling_factor=1.0,
                pitch_variance_scale=1.0, energy_variance_scale=1.0, pause_duration_scaling_factor=1.0, run_postflow=True, vocode=True,
                generator=None, packed=None, z_sq=None):
        """Same arguments and result keys as engine.AcousticEngine.forward (+ ``wav`` / ``wav_spans`` when a vocoder is loaded and
        ``vocode``).  Every stage is one call into libtoucan_hip.so."""
        if torch.cuda.current_device() != self.device.index:
            with torch.cuda.device(self.device):
                return self.forward(texts, utt_embs, lang_ids, durations, pitch, energy, z_noise, duration_scaling_factor, pitch_variance_scale,
                                    energy_variance_scale, pause_duration_scaling_factor, run_postflow, vocode, generator, packed, z_sq)
        dev, lib, st = self.device, self.lib, self._stream()
        assert duration_scaling_factor > 0
        if packed is None:
            packed = self.pack_inputs(texts, utt_embs, lang_ids, durations, pitch, energy)
        Ls, text, emb, lang, gp, ge, gd = (packed[k] for k in ("Ls", "text", "emb", "lang", "gp", "ge", "gd"))
        B = len(Ls)
        self._ensure_pe(max(Ls))
        ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        lens = (C.c_int32 * B)(*Ls)
        capi.check(lib.tts_encoder(self.h, ptr(text), ptr(emb), ptr(lang), lens, B, st), "tts_encoder")
        capi.check(lib.tts_variance_predictors(self.h, ptr(gp), ptr(ge), ptr(gd), st), "tts_variance_predictors")
        frames = (C.c_int32 * B)()
        capi.check(lib.tts_control_and_regulate(self.h, float(duration_scaling_factor), float(pitch_variance_scale), float(energy_variance_scale),
                                                float(pause_duration_scaling_factor), frames, st), "tts_control_and_regulate")
        Ts = [int(f) for f in frames]
        if max(Ts) > self._pmax:  # longer than the position table: enlarge it and redo the (cheap) phoneme stages
            self._ensure_pe(max(Ts))
            return self.forward(texts, utt_embs, lang_ids, durations, pitch, energy, z_noise, duration_scaling_factor, pitch_variance_scale,
                                energy_variance_scale, pause_duration_scaling_factor, run_postflow, vocode, generator, packed, z_sq)
        rag_p = Ragged(Ls, dev)
        rag_f = Ragged(Ts, dev, align=2)
        capi.check(lib.tts_decoder(self.h, st), "tts_decoder")
        capi.check(lib.tts_postnet(self.h, st), "tts_postnet")
        RF = rag_f.total_rows
        decoded = refined = None
        if run_postflow:
            rag_s = rag_f.halved()
            RS = RF // 2
            if z_sq is not None:
                assert tuple(z_sq.shape) == (RS, 160) and z_sq.is_contiguous()
            elif z_noise is None:  # Glow.py:363: z ~ 0.8 N(0,1), drawn per squeezed row on the device
                z_sq = torch.randn(RS, 160, device=dev, dtype=torch.float32, generator=generator) * 0.8
            else:
                z_sq = self.squeeze_noise(z_noise, Ts)
            capi.check(lib.tts_postflow(self.h, ptr(z_sq), st), "tts_postflow")
        mel_packed = torch.empty(RF, 80, dtype=torch.float32, device=dev)
        capi.check(lib.tts_copy_mel(self.h, ptr(mel_packed), 80, st), "tts_copy_mel")
        rag_out = rag_f.halved().doubled() if run_postflow else rag_f
        R = rag_p.total_rows
        d = torch.empty(R, dtype=torch.int32, device=dev)
        p = torch.empty(R, dtype=torch.float32, device=dev)
        en = torch.empty(R, dtype=torch.float32, device=dev)
        capi.check(lib.tts_copy_prosody(self.h, ptr(d), ptr(p), ptr(en), st), "tts_copy_prosody")
        out = dict(durations_packed=d, pitch_packed=p, energy_packed=en, rag_phone=rag_p, rag_frame=rag_f, mel_packed=mel_packed, rag_mel=rag_out)
        out["mel"] = [mel_packed[b0:b0 + n] for b0, n in zip(rag_out.begins, rag_out.lengths)]
        for key, src in (("durations", d), ("pitch", p), ("energy", en)):
            out[key] = [src[b0:b0 + n] for b0, n in zip(rag_p.begins, rag_p.lengths)]
        if vocode and self.kind is not None:
            out["wav"], out["wav_spans"] = self._vocode_internal(rag_out, st)
        return out

    @torch.inference_mode()
    def forward_pipelined(self, batches):
        """Several batches, two HIP streams: the acoustic model of batch k+1 is enqueued beside the vocoder of batch k (the handle's
        phoneme / frame arenas belong to the acoustic stages, its vocoder arenas to the vocoder; the mel crosses in forward()'s own
        copy behind an event).  `batches` yields dicts of forward() keyword arguments; yields forward()'s result per batch with
        ``wav`` / ``wav_spans``, ordered on the caller's current stream.  Same results as forward() batch by batch."""
        assert self.kind is not None, "no vocoder was loaded into this handle"
        with torch.cuda.device(self.device):
            caller = torch.cuda.current_stream(self.device)
            if self._streams is None:
                self._streams = (torch.cuda.Stream(self.device, priority=-1), torch.cuda.Stream(self.device))  # (acoustic: high priority)
            s_ac, s_voc = self._streams
            s_ac.wait_stream(caller)
            s_voc.wait_stream(caller)
            pending = None

            def finish(p):
                out, done = p
                caller.wait_event(done)
                for key in ("wav", "mel_packed", "durations_packed", "pitch_packed", "energy_packed"):
                    out[key].record_stream(caller)
                return out

            for kw in batches:  # (may be a generator that builds a batch's device tensors on the caller's stream while we iterate)
                kw = dict(kw)
                kw["vocode"] = False
                s_ac.wait_stream(caller)  # everything the caller enqueued for this batch so far is ordered in front of its acoustic pass
                for v in kw.values():     # the inputs are read on s_ac: the allocator must not hand their memory out before that
                    for t in (v if isinstance(v, (list, tuple)) else [v]):
                        if torch.is_tensor(t) and t.is_cuda:
                            t.record_stream(s_ac)
                    if isinstance(v, dict):
                        for t in v.values():
                            if torch.is_tensor(t) and t.is_cuda:
                                t.record_stream(s_ac)
                with torch.cuda.stream(s_ac):
                    out = self.forward(**kw)
                    mel_ready = torch.cuda.Event()
                    mel_ready.record(s_ac)
                with torch.cuda.stream(s_voc):
                    s_voc.wait_event(mel_ready)
                    for key in ("mel_packed", "durations_packed", "pitch_packed", "energy_packed"):
                        out[key].record_stream(s_voc)
                    out["wav"], _ = self.vocode(out["mel_packed"], out["rag_mel"])
                    out["wav_spans"] = [(384 * int(b0), 384 * int(n)) for b0, n in zip(out["rag_mel"].begins, out["rag_mel"].lengths)]
                    done = torch.cuda.Event()
                    done.record(s_voc)
                if pending is not None:
                    yield finish(pending)
                pending = (out, done)
            if pending is not None:
                yield finish(pending)

    @torch.inference_mode()
    def predict_frame_counts(self, texts, utt_embs, lang_ids=None, pitch=None, energy=None, duration_scaling_factor=1.0,
                             pitch_variance_scale=1.0, energy_variance_scale=1.0, pause_duration_scaling_factor=1.0):
        """Stage A alone (tts_encoder, tts_variance_predictors, tts_control_and_regulate): mel frames per utterance - the balancing
        key of the multi-GPU deal (distributed.py)."""
        with torch.cuda.device(self.device):
            dev, lib, st = self.device, self.lib, self._stream()
            B = len(texts)
            Ls = [int(t.shape[0]) for t in texts]
            self._ensure_pe(max(Ls))
            text = torch.cat([t.reshape(-1, 62).to(torch.float32) for t in texts], dim=0).to(dev).contiguous()
            emb = utt_embs.to(dev, torch.float32).reshape(B, 64).contiguous() if utt_embs is not None else None
            lang = torch.tensor([int(i) for i in lang_ids], dtype=torch.int32).to(dev) if (self.multilingual and lang_ids is not None) else None
            cat = lambda lst: None if lst is None else torch.cat([torch.as_tensor(v).reshape(-1).to(torch.float32) for v in lst]).to(dev).contiguous()
            gp, ge = cat(pitch), cat(energy)
            ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
            capi.check(lib.tts_encoder(self.h, ptr(text), ptr(emb), ptr(lang), (C.c_int32 * B)(*Ls), B, st), "tts_encoder")
            capi.check(lib.tts_variance_predictors(self.h, ptr(gp), ptr(ge), None, st), "tts_variance_predictors")
            frames = (C.c_int32 * B)()
            capi.check(lib.tts_control_and_regulate(self.h, float(duration_scaling_factor), float(pitch_variance_scale),
                                                    float(energy_variance_scale), float(pause_duration_scaling_factor), frames, st),
                       "tts_control_and_regulate")
            return [int(f) for f in frames]

    @torch.inference_mode()
    def vocode_batch(self, rag_mel):
        """Vocoder on the mel of the batch `forward(..., vocode=False)` just produced (it still sits in the handle's workspace)."""
        with torch.cuda.device(self.device):
            return self._vocode_internal(rag_mel, self._stream())

    def _vocode_internal(self, rag_mel, st):
        """Vocoder on the mel that sits in the handle's workspace (no copy)."""
        B = rag_mel.n_seq
        mel, ld = C.c_void_p(), C.c_int32()
        fb, fc = (C.c_int32 * B)(), (C.c_int32 * B)()
        capi.check(self.lib.tts_mel(self.h, C.byref(mel), C.byref(ld), fb, fc), "tts_mel")
        return self._run_vocoder(mel, int(ld.value), fb, fc, B, st)

    def _run_vocoder(self, mel_ptr, ld, fb, fc, B, st):
        total = max([int(b) + int(n) for b, n in zip(fb, fc)] + [0])
        wav = torch.empty(384 * max(total, 1), dtype=torch.float32, device=self.device)
        fn = self.lib.tts_vocoder_bigvgan if self.kind == "bigvgan" else self.lib.tts_vocoder_hifigan
        capi.check(fn(self.h_voc, mel_ptr, ld, fb, fc, B, C.c_void_p(wav.data_ptr()), st), "tts_vocoder_" + str(self.kind))
        return wav, [(384 * int(b), 384 * int(n)) for b, n in zip(fb, fc)]

    @torch.inference_mode()
    def vocode(self, mel_packed, rag):
        """mel_packed [rows, 80] (utterance u at rag.begins[u]) -> (packed waveform, Ragged of samples), like VocoderEngine.forward."""
        assert self.kind is not None, "no vocoder was loaded into this handle"
        with torch.cuda.device(self.device):
            mel = mel_packed.to(self.device, torch.float32)
            assert mel.stride(-1) == 1
            B = rag.n_seq
            fb, fc = (C.c_int32 * B)(*rag.begins), (C.c_int32 * B)(*rag.lengths)
            wav, _ = self._run_vocoder(C.c_void_p(mel.data_ptr()), int(mel.stride(0)), fb, fc, B, self._stream())
        return wav, rag.scaled(384)
