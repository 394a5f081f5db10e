"""bench.py - throughput of the ToucanTTS hot path on MI355X (metric of BASELINE.json).

One step = one pass of the hot path over one batch of synthetic utterances per GPU:
acoustic model (Conformer enc -> pitch / energy predictors -> control -> length regulator -> Conformer dec -> PostNet
-> PostFlow) + BigVGAN vocoder, batch 32 x 128 phonemes per GPU (BASELINE.json configs[2]; at N GPUs the job is N x 32
utterances, weak scaling, waveforms all-gathered over RCCL).  Gold durations fix the frame count (the duration predictor is
bypassed: with random-init weights its exp() output is numerically wild, SURVEY.md fact 4 - the API offers `durations=` for
exactly this, ToucanTTSInterface.py:139-141).  Inputs are resident in HBM before the timed region; the PCIe-inclusive rate
(host phoneme tensors in, host waveforms out) is measured separately after it and reported beside `value`, never as `value`.
Prints ONE JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 32] [--phones 128] [--vocoder bigvgan|hifigan]
                    [--dtype fp32|bf16|fp16] [--pitch-scale S] [--energy-scale S] [--no-cpu-baseline]

  --dtype bf16 (default)                                    BASELINE.json configs[2]
  --dtype fp16 --pitch-scale 1.3 --energy-scale 0.7         the per-GPU shard of configs[4] (fp16 MFMA, PostFlow on, variance scaling)
  --gpus N (N > 1) without a torch.distributed environment  re-launches itself under `python -m torch.distributed.run`
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_TFLOPS = 157.3    # MI355X_MICROARCH.md: fp32 matrix peak (v_mfma_f32_32x32x2_f32)
PEAK_16BIT_TFLOPS = 2500.0  # dense bf16 / fp16 MFMA peak
PEAK_HBM_GBS = 8000.0      # HBM3E spec peak (6.3 TB/s is what a streaming copy reaches)
# vector lane-operations/s: 256 CUs x 4 SIMDs x 16 lanes per cycle at 2.4 GHz.  Measured on these kernels (profiles/r02_*_SQ*):
# SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 1.1 quad-cycles, i.e. a wave64 vector instruction holds its SIMD's issue for ~4 cycles
PEAK_VALU_LANE_OPS = 256 * 4 * 16 * 2.4e9


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU")
    ap.add_argument("--phones", type=int, default=128)
    ap.add_argument("--frames-per-phone", type=int, default=5)
    ap.add_argument("--vocoder", default="bigvgan", choices=["bigvgan", "hifigan"])
    ap.add_argument("--dtype", default="bf16", choices=["fp32", "bf16", "fp16", "mixed", "fp32x3", "mixed3"],
                    help="bf16 = BASELINE.json configs[2] (bf16 MFMA GEMMs, fp32 statistics); fp16 = configs[4]'s fp16 MFMA path; "
                         "fp32 = exact-parity configuration (fp32 matrix instructions); mixed = that acoustic model + fp16 vocoder; "
                         "fp32x3 = fp32 tensors, dense products as three fp16 MFMAs on split operands (~22 bits per product: keeps the "
                         "fp32 tolerances against the reference); mixed3 = fp32x3 acoustic model + fp16 vocoder")
    ap.add_argument("--pitch-scale", type=float, default=1.0, help="pitch_variance_scale (configs[4]: 1.3)")
    ap.add_argument("--energy-scale", type=float, default=1.0, help="energy_variance_scale (configs[4]: 0.7)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true",
                    help="skip the B = 1 check run after the timed region (kernel traces: its short launches would dilute the averages)")
    ap.add_argument("--fuse-snake", action="store_true", help="BigVGAN: anti-aliased snake inside the conv input staging")
    ap.add_argument("--graphs", action="store_true", help="replay the shape-static stages as HIP graphs (no per-kernel roofline leg)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="one HIP stream: every step runs acoustic model, then vocoder.  Default (native sequencer): two streams - the "
                         "acoustic model of step k+1 runs beside the vocoder of step k (the small acoustic kernels fill the gaps the "
                         "vocoder leaves); all K steps still complete inside the timed region")
    ap.add_argument("--sequencer", default="native", choices=["native", "python"],
                    help="native: the stage API (csrc/pipeline.hip sequences the kernels in C++; the product path); python: engine.py issues "
                         "every kernel-level call itself")
    return ap.parse_args()


def relaunch_distributed(args):
    """`python bench.py --gpus N` from a plain shell: start the N ranks as a child job (one process per GPU over RCCL) and
    exit with its code.  Nothing in this process has touched the GPU yet (no torch.cuda call, no HIP library loaded)."""
    port = 29500 + os.getpid() % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def cpu_baseline(phones, frames_per_phone, vocoder, scales=None):
    """The CPU oracle (kind 'port') on a bounded sample: ONE utterance of the same workload, processed the way the
    reference's read_to_file does (one utterance at a time), all host cores."""
    import numpy as np
    import torch
    from ims_toucan_prosody_variance_amd import fixture_weights as fw, synthetic as syn
    from oracle import toucan_oracle as orc
    # the GPU box exposes many more logical CPUs than its cgroup share (16 per GPU): oversubscribing stalls torch
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))
    torch.set_num_threads(cores)
    oa = orc.AcousticOracle(fw.acoustic_state_dict())
    ov = orc.VocoderOracle(fw.bigvgan_state_dict() if vocoder == "bigvgan" else fw.hifigan_state_dict(), vocoder)
    feats = torch.from_numpy(syn.utterance_features(0, phones, word_boundaries=False))
    emb = torch.from_numpy(syn.utterance_embedding(0))
    dur = torch.full((phones,), frames_per_phone, dtype=torch.long)
    T = phones * frames_per_phone
    z = torch.from_numpy(syn.postflow_noise(0, T))
    times_a, times_v = [], []
    for it in range(3):
        t0 = time.perf_counter()
        o = oa(feats, emb, syn.LANG_EN, z_noise=z, durations=dur, **(scales or {}))
        t1 = time.perf_counter()
        w = ov(o["mel"].t().contiguous())
        t2 = time.perf_counter()
        if it > 0:
            times_a.append(t1 - t0)
            times_v.append(t2 - t1)
    ta, tv = float(np.median(times_a)), float(np.median(times_v))
    frames = int(o["mel"].shape[0])
    return dict(value=frames / (ta + tv), unit="mel-frames/s", cores=cores, kind="port",
                sample=f"1 utterance x {phones} phonemes ({frames} frames, {w.numel() / 24000.0:.2f} s audio), "
                       f"acoustic + {vocoder}, fp32, median of 2 after 1 warm-up",
                acoustic_mel_frames_per_s=frames / ta, vocoder_rtf=tv / (w.numel() / 24000.0),
                e2e_rtf=(ta + tv) / (w.numel() / 24000.0)), (o["mel"], w)


def committed_counters(kernel, algorithmic_bytes=None):
    """HBM bytes per launch and VALU lane-operations per element of `kernel` from this round's committed rocprofv3 PMC passes
    (separate --pmc runs over tools/microbench_resblock.py at the bench's shapes; FETCH_SIZE doubled per MI355X_MICROARCH.md).
    Counter passes cannot share a process with the timed region, so these are read from profiles/; the file is named in the
    line.  (None, None, None) when no pass covers the kernel."""
    for tag in ("r03", "r02", "r01_v18", "r01"):
        path = os.path.join(ROOT, "profiles", f"{tag}_pmc_resblock_traffic.json")
        if not os.path.exists(path):
            continue
        with open(path) as f:
            rows = [r for r in json.load(f)["launches"] if r["kernel"] == kernel and r["act"] == "snake"]
        if not rows:
            continue
        if algorithmic_bytes is not None:
            # the counter pass ran the micro-benchmark, not this process: its launch must be the bench's launch (same rows x channels
            # x 16-bit read + write; the bench's figure also counts the weights, < 0.1 %), or the traffic would describe another shape
            assert all(abs(r["algorithmic_bytes"] - algorithmic_bytes) <= 2e-3 * algorithmic_bytes for r in rows), \
                f"{path}: counter pass shape ({rows[0]['algorithmic_bytes']} B per launch) is not the bench's launch ({algorithmic_bytes} B)"
        traffic = sum(r["hbm_bytes_corrected"] for r in rows) / len(rows)
        valu = None
        sq = os.path.join(ROOT, "profiles", f"{tag}_pmc_resblock_SQ_summary.json")
        if os.path.exists(sq):
            with open(sq) as f:
                v = [r["valu_per_elem"] for r in json.load(f)["launches"] if r["kernel"] == kernel and r["act"] == "snake"]
            valu = sum(v) / len(v) if v else None
        return traffic, valu, os.path.relpath(path, ROOT)
    return None, None, None


class TwoStreamRunner:
    """bench.py's default step on one GPU: the acoustic model of step k+1 is enqueued on one HIP stream beside the vocoder of step k on
    another (the mel crosses in forward()'s own copy behind an event; the handle keeps the acoustic stages' arenas and the vocoder's
    apart).  tests/test_gpu_product_path.py drives this very class at full size against the Python sequencer, bit for bit."""

    def __init__(self, pipe, packed, z_sq, scales, dev, priority=-1):
        import torch
        self.torch, self.pipe, self.packed, self.z_sq, self.scales = torch, pipe, packed, z_sq, dict(scales)
        # the acoustic stream has the higher priority: its short kernels take the CUs a vocoder launch frees in its tail before
        # the next vocoder launch's persistent workgroups do (42.0 vs 42.25 ms measured; TOUCAN_BENCH_AC_PRIORITY=0 for the A/B run)
        self.s_ac, self.s_voc = torch.cuda.Stream(dev, priority=priority), torch.cuda.Stream(dev)
        cur = torch.cuda.current_stream(dev)  # the resident inputs were written on the caller's stream
        self.s_ac.wait_stream(cur)
        self.s_voc.wait_stream(cur)

    def step(self, ev=None, after_vocoder=None):
        """ev: four timing events (acoustic begin / end, vocoder end / begin) recorded on the stage's own stream; after_vocoder(wav):
        called inside the vocoder stream's context (the multi-GPU exchange step)."""
        torch, pipe = self.torch, self.pipe
        with torch.cuda.stream(self.s_ac):
            if ev:
                ev[0].record(self.s_ac)
            out = pipe.forward(None, None, packed=self.packed, z_sq=self.z_sq, vocode=False, **self.scales)
            if ev:
                ev[1].record(self.s_ac)
            done = torch.cuda.Event()
            done.record(self.s_ac)
        with torch.cuda.stream(self.s_voc):
            self.s_voc.wait_event(done)
            out["mel_packed"].record_stream(self.s_voc)
            if ev:
                ev[3].record(self.s_voc)
            wav, _ = pipe.vocode(out["mel_packed"], out["rag_mel"])
            if ev:
                ev[2].record(self.s_voc)
            if after_vocoder is not None:
                after_vocoder(wav)
        return out, wav


def verify_against_single_run(pipe, out, wav, packed, z_sq, scales, B, oracle=None):
    """The `verify` object of the line: utterance 0 of a finished step (mel and waveform as the timed path produced them) against
    the SAME utterance run alone (B = 1) through the same handle - computed after the timed region.  16-bit configurations:
    bit-identical by construction (an utterance's arithmetic never depends on the batch); fp32: the frame stages may take the
    split forms on the small grid of a B = 1 run, so the difference is rounding order (reported).  oracle: (mel, wav) of the CPU
    oracle on the same utterance - the checker, run by the cpu_baseline leg - adds the error against it."""
    import torch
    from ims_toucan_prosody_variance_amd.ragged import Ragged
    L0 = packed["Ls"][0]
    cut = lambda t, n: None if t is None else t[:n].contiguous()
    one = dict(Ls=[L0], text=cut(packed["text"], L0), emb=cut(packed["emb"], 1), lang=cut(packed["lang"], 1), gp=cut(packed["gp"], L0),
               ge=cut(packed["ge"], L0), gd=cut(packed["gd"], L0))
    T0 = int(out["rag_frame"].lengths[0])
    rows_sq = Ragged([T0], pipe.device, align=2).total_rows // 2  # (utterance 0 starts at row 0 of the batch's squeezed noise too)
    o1 = pipe.forward(None, None, packed=one, z_sq=z_sq[:rows_sq].contiguous(), vocode=False, **scales)
    w1, _ = pipe.vocode(o1["mel_packed"], o1["rag_mel"])
    torch.cuda.synchronize()
    n = int(out["rag_mel"].lengths[0])
    b0 = int(out["rag_mel"].begins[0])
    mel_b, mel_1 = out["mel_packed"][b0:b0 + n], o1["mel_packed"][:n]
    wav_b, wav_1 = wav[384 * b0:384 * (b0 + n)], w1[: 384 * n]
    v = {"utterance": 0, "frames": n, "against": f"the same utterance run alone (B = 1) vs in the batch of {B}",
         "finite": bool(torch.isfinite(mel_b).all() and torch.isfinite(wav_b).all()),
         "mel_bit_identical": bool(torch.equal(mel_b, mel_1)), "wav_bit_identical": bool(torch.equal(wav_b, wav_1)),
         "mel_max_abs_diff": float((mel_b - mel_1).abs().max()), "wav_max_abs_diff": float((wav_b - wav_1).abs().max())}
    if oracle is not None:
        mel_o, wav_o = (t.to(mel_b.device) for t in oracle)
        if tuple(mel_o.shape) == tuple(mel_b.shape):
            v["vs_cpu_oracle_fp32"] = {"mel_mean_abs_err": float((mel_b - mel_o).abs().mean()), "mel_max_abs_err": float((mel_b - mel_o).abs().max()),
                                       "mel_mean_abs": float(mel_o.abs().mean()), "wav_mean_abs_err": float((wav_b - wav_o).abs().mean()),
                                       "wav_max_abs_err": float((wav_b - wav_o).abs().max()), "wav_mean_abs": float(wav_o.abs().mean())}
    return v


def algorithmic_flops_per_utterance(L, T, vocoder):
    """FLOPs (2 x MAC) of one utterance of L phonemes -> T frames.  Conformer block of length N:
    3 022 848 N + 1 536 N^2 + 384 K N (K = 7 encoder / 31 decoder); predictors 0.66 G at L = 128; feat_out + PostNet 1.54 G and
    PostFlow 27.63 G at T = 640 (linear in T); vocoder 677.15 (BigVGAN) / 648.24 (Avocodo) MFLOP per mel frame."""
    block = lambda n, k: 3022848.0 * n + 1536.0 * n * n + 384.0 * k * n
    acoustic = 6 * block(L, 7) + 6 * block(T, 31) + 0.66e9 * L / 128 + (1.54e9 + 27.63e9) * T / 640
    voc = (677.15e6 if vocoder == "bigvgan" else 648.24e6) * (T - T % 2)
    return acoustic, voc


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch_distributed(args))

    import numpy as np
    import torch
    import torch.distributed as dist

    import ims_toucan_prosody_variance_amd  # noqa: F401
    from ims_toucan_prosody_variance_amd import capi, engine, fixture_weights as fw, native, profiling, synthetic as syn

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # Rehearsal on a one-GPU box (never used by the driver): TOUCAN_BENCH_REHEARSAL=1 maps every rank to device 0 and runs the
    # collectives over gloo on host copies, so that the launch / barrier / max-over-ranks / rank-0 JSON logic can be exercised
    # end to end without a second GPU.  The numbers of such a run mean nothing.
    rehearsal = os.environ.get("TOUCAN_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    precision = {"fp32": "f32", "bf16": "bf16", "fp16": "f16", "mixed": "f32", "fp32x3": "f32x3", "mixed3": "f32x3"}[args.dtype]
    voc_precision = "f16" if args.dtype in ("mixed", "mixed3") else precision  # mixed: fp32-tolerance acoustic model + fp16 vocoder
    log(f"building engines (fixture weights, {precision})")
    voc_sd = fw.bigvgan_state_dict() if args.vocoder == "bigvgan" else fw.hifigan_state_dict()
    use_native = args.sequencer == "native" and not args.graphs and not args.fuse_snake
    if use_native:
        pipe = native.NativePipeline(fw.acoustic_state_dict(), voc_sd, args.vocoder, dev, precision=precision, vocoder_precision=voc_precision)
    else:
        ac = engine.AcousticEngine(fw.acoustic_state_dict(), dev, precision=precision, use_graphs=args.graphs)
        voc = engine.VocoderEngine(voc_sd, args.vocoder, dev, precision=voc_precision, fuse_snake=args.fuse_snake, use_graphs=args.graphs)

    B, L, T = args.batch, args.phones, args.phones * args.frames_per_phone
    log(f"synthetic inputs: {B} x {L} phonemes -> {T} frames per utterance")
    ids = [rank * B + u for u in range(B)]
    host_texts = [torch.from_numpy(syn.utterance_features(u, L, word_boundaries=False)).pin_memory() for u in ids]
    host_embs = torch.from_numpy(np.stack([syn.utterance_embedding(u) for u in ids])).pin_memory()
    host_zs = [torch.from_numpy(syn.postflow_noise(u, T)).pin_memory() for u in ids]
    texts = [t.to(dev) for t in host_texts]
    embs = host_embs.to(dev)
    durs = [torch.full((L,), args.frames_per_phone, dtype=torch.int32, device=dev) for _ in ids]
    zs = [z.to(dev) for z in host_zs]
    langs = [syn.LANG_EN] * B
    scales = dict(pitch_variance_scale=args.pitch_scale, energy_variance_scale=args.energy_scale)
    # 1-D concatenation form of the gather output (accepted by every backend)
    gathered = torch.empty(world * B * T * 384, device="cpu" if rehearsal else dev) if world > 1 else None
    gathered2 = [gathered, torch.empty_like(gathered)] if world > 1 else None  # (two-stream form: the exchange of step k runs beside step k+1)
    gather_no = [0]

    packed = z_sq = None
    if use_native:  # the stage API's input format (packed along the phoneme axis): resident in HBM before the timed region
        packed = pipe.pack_inputs(texts, embs, langs, durations=durs)
        z_sq = pipe.squeeze_noise(zs, [T] * B)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    overlap = use_native and not args.no_overlap
    runner = s_comm = None
    if overlap:
        s_comm = torch.cuda.Stream(dev) if world > 1 else None
        runner = TwoStreamRunner(pipe, packed, z_sq, scales, dev, priority=int(os.environ.get("TOUCAN_BENCH_AC_PRIORITY", "-1")))

    def exchange(wav):
        """The exchange step on its own stream: the vocoder of step k+1 does not wait for the waveforms of step k to cross xGMI
        (31.5 MB per rank and step); all of it is inside the timed region (the final device synchronise).  A failing collective
        ends the run (no retry on another stream: a broken communicator must not hide behind a slower path)."""
        block = wav[: B * T * 384].contiguous()
        dst = gathered2[gather_no[0] & 1]
        gather_no[0] += 1
        if rehearsal:
            dist.all_gather_into_tensor(dst, block.cpu())
            return
        voc_done = torch.cuda.Event()
        voc_done.record(runner.s_voc)
        with torch.cuda.stream(s_comm):
            s_comm.wait_event(voc_done)
            block.record_stream(s_comm)
            dist.all_gather_into_tensor(dst, block)

    def step_overlapped(record=False):
        return runner.step(ev if record else None, exchange if world > 1 else None)

    def step(record=False, tx=texts, em=embs, zz=zs, resident=True):
        if overlap and resident:
            return step_overlapped(record)
        if record:
            ev[0].record()
        if use_native and resident:
            out = pipe.forward(None, None, packed=packed, z_sq=z_sq, vocode=False, **scales)
        elif use_native:
            out = pipe.forward(tx, em, langs, durations=durs, z_noise=zz, vocode=False, **scales)
        else:
            out = ac.forward(tx, em, langs, durations=durs, z_noise=zz, **scales)
        if record:
            ev[1].record()
        if use_native:
            wav, _ = pipe.vocode_batch(out["rag_mel"])
        else:
            wav, rag = voc.forward(out["mel_packed"], out["rag_mel"])
        if record:
            ev[2].record()
        if world > 1:  # one exchange step: waveforms of all ranks (equal length here) over RCCL/xGMI
            block = wav[: B * T * 384].contiguous()
            dist.all_gather_into_tensor(gathered, block.cpu() if rehearsal else block)
        return out, wav

    # ---- warm-up; the second warm-up step times every MFMA kernel class to find the dominant one ----
    log("warm-up step 1 (untimed: first-launch costs)")
    step()
    torch.cuda.synchronize()
    classes, dominant = {}, None
    if use_native:
        pipe.profile(True, None)
        log("warm-up step 2 (all MFMA kernel classes timed inside the stage entries)")
        calls0 = capi.CALLS
        step()
        torch.cuda.synchronize()
        abi_calls = capi.CALLS - calls0
        classes = pipe.profile_summary()
        dominant = max(classes, key=lambda k: classes[k]["total_ms"]) if classes else None
        pipe.profile(False)
    elif not args.graphs:
        timer = profiling.ConvTimer()
        ac.ops.timer = voc.ops.timer = timer
        log("warm-up step 2 (all MFMA kernel classes timed)")
        calls0 = capi.CALLS
        step()
        torch.cuda.synchronize()
        abi_calls = capi.CALLS - calls0
        classes = timer.summary()
        dominant = max(classes, key=lambda k: classes[k]["total_ms"]) if classes else None
        ac.ops.timer = voc.ops.timer = None
    else:
        abi_calls = None
        step()
    for _ in range(max(0, args.warmup - 2)):
        step()
    # ---- timed region: only the dominant class carries event pairs (a few dozen launches per step) ----
    timer = None
    if use_native:
        if dominant:
            pipe.profile(True, dominant)
    else:
        timer = profiling.ConvTimer(select={dominant}) if dominant else None
        ac.ops.timer = voc.ops.timer = timer
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    t_ac = t_voc = 0.0
    step_done = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]  # completion of every timed step, on the stream that ends it
    end_stream = lambda: runner.s_voc if overlap else torch.cuda.current_stream(dev)
    step_done[0].record(end_stream())
    for it in range(args.steps):
        log(f"timed step {it}")
        if overlap:  # no host wait between steps: the streams overlap consecutive steps; stage times from the last step only
            out, wav = step(record=(it == args.steps - 1))
        else:
            out, wav = step(record=True)
            ev[2].synchronize()
            t_ac += ev[0].elapsed_time(ev[1]) * 1e-3
            t_voc += ev[1].elapsed_time(ev[2]) * 1e-3
        step_done[it + 1].record(end_stream())
    torch.cuda.synchronize()
    if overlap:  # (the vocoder of one step, running beside the acoustic model of the next)
        t_voc = ev[3].elapsed_time(ev[2]) * 1e-3 * args.steps
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device="cpu" if rehearsal else dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    dom_summary = None
    if use_native:
        if dominant:
            dom_summary = pipe.profile_summary()[dominant]
        pipe.profile(False)
    else:
        if dominant:
            dom_summary = timer.summary()[dominant]
        ac.ops.timer = voc.ops.timer = None

    frames_out = int(sum(m.shape[0] for m in out["mel"]))
    audio_s = frames_out * 384 / 24000.0
    # HIP-event time between the completions of consecutive timed steps (event timing, median beside the mean)
    step_ms_in_order = [step_done[i].elapsed_time(step_done[i + 1]) for i in range(args.steps)]
    step_ms = sorted(step_ms_in_order)
    step_ms_median = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])

    # ---- after the timed region: the acoustic model's own rate (one stream, nothing beside it) and the parity check of the
    #      timed path's output ----
    acoustic_alone_ms = None
    last_out, last_wav = out, wav
    if use_native:
        torch.cuda.synchronize()
        pe = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        n_probe = 3
        pipe.forward(None, None, packed=packed, z_sq=z_sq, vocode=False, **scales)
        pe[0].record()
        for _ in range(n_probe):
            pipe.forward(None, None, packed=packed, z_sq=z_sq, vocode=False, **scales)
        pe[1].record()
        torch.cuda.synchronize()
        acoustic_alone_ms = pe[0].elapsed_time(pe[1]) / n_probe
        if overlap:
            t_ac = acoustic_alone_ms * 1e-3 * args.steps

    # ---- PCIe-inclusive rate (SURVEY.md 8(d) protocol): phoneme tensors / embeddings / noise from pinned host memory in,
    #      waveforms to pinned host memory out, every step; reported beside `value` ----
    pcie = None
    if world == 1:
        host_wav = torch.empty(wav.numel(), dtype=torch.float32).pin_memory()
        n_pcie = max(2, min(args.steps, 3))
        torch.cuda.synchronize()
        tp0 = time.perf_counter()
        for _ in range(n_pcie):
            tx = [t.to(dev, non_blocking=True) for t in host_texts]
            em = host_embs.to(dev, non_blocking=True)
            zz = [z.to(dev, non_blocking=True) for z in host_zs]
            _, w = step(tx=tx, em=em, zz=zz, resident=False)  # (one stream: host tensors -> acoustic model -> vocoder -> host)
            host_wav.copy_(w, non_blocking=True)
        torch.cuda.synchronize()
        tp = (time.perf_counter() - tp0) / n_pcie
        pcie = {"value": frames_out / tp, "unit": "mel-frames/s", "ms_per_step": 1e3 * tp, "steps": n_pcie,
                "h2d_bytes": int(sum(t.numel() for t in host_texts) * 4 + host_embs.numel() * 4 + sum(z.numel() for z in host_zs) * 4),
                "d2h_bytes": int(wav.numel() * 4)}

    if rank == 0:
        roof = None
        if dominant:
            dom = dom_summary
            is16 = ("bf16" in dominant or "f16" in dominant or "resblock" in dominant) and "f32<" not in dominant and "f32x3" not in dominant
            peak = PEAK_16BIT_TFLOPS if is16 else (PEAK_16BIT_TFLOPS / 3 if "f32x3" in dominant else PEAK_F32_TFLOPS)  # (split fp32: three MFMAs per product)
            traffic, valu_per_elem, src = committed_counters(dominant, dom["bytes_per_launch"])
            mfma_frac = dom["tflops"] / peak
            gbs = dom["bytes_per_launch"] / (dom["avg_us"] * 1e-6) / 1e9
            hbm_frac = gbs / PEAK_HBM_GBS
            valu_frac = None
            if valu_per_elem is not None and dom.get("elems_per_launch"):
                valu_frac = valu_per_elem * dom["elems_per_launch"] / (dom["avg_us"] * 1e-6) / PEAK_VALU_LANE_OPS
            # the binding resource is the one with the highest utilisation (SQ counters for the fused step: VALU issue, not
            # the matrix pipe and not HBM - DESIGN.md section 5); `frac` / `achieved` / `peak` describe THAT resource's roofline
            # among the two the contract names, the other fraction and the VALU view are given next to it
            bound = "mfma" if mfma_frac >= hbm_frac else "hbm"
            roof = {"bound": bound, "kernel": dominant,
                    "achieved": dom["tflops"] if bound == "mfma" else gbs, "peak": peak if bound == "mfma" else PEAK_HBM_GBS,
                    "unit": "TFLOP/s" if bound == "mfma" else "GB/s", "frac": max(mfma_frac, hbm_frac),
                    "mfma_frac": mfma_frac, "mfma_tflops": dom["tflops"], "mfma_peak_tflops": peak,
                    "hbm_frac": hbm_frac, "hbm_gbs_algorithmic": gbs, "hbm_peak_gbs": PEAK_HBM_GBS,
                    "valu_frac": valu_frac, "valu_lane_ops_per_element": valu_per_elem,
                    "limiter": ("valu-issue" if valu_frac is not None and valu_frac > max(mfma_frac, hbm_frac) else bound),
                    "traffic": traffic, "traffic_source": src, "algorithmic_bytes_per_launch": dom["bytes_per_launch"],
                    "avg_launch_us": dom["avg_us"], "launches_per_step": dom["launches"] / args.steps,
                    "flops_per_launch": dom["flops_per_launch"], "share_of_step": dom["total_ms"] / (1e3 * elapsed)}
        fa, fv = algorithmic_flops_per_utterance(L, frames_out // B, args.vocoder)
        step_flops = B * (fa + fv)
        step_peak = PEAK_F32_TFLOPS if args.dtype == "fp32" else (PEAK_16BIT_TFLOPS / 3 if args.dtype == "fp32x3" else PEAK_16BIT_TFLOPS)
        step_roof = {"flops_per_step": step_flops, "acoustic_flops_per_utterance": fa, "vocoder_flops_per_utterance": fv,
                     "achieved_tflops": step_flops / (elapsed / args.steps) / 1e12, "peak_tflops": step_peak,
                     "frac": step_flops / (elapsed / args.steps) / 1e12 / step_peak,
                     "note": "algorithmic FLOPs of the whole step (SURVEY.md 8(d)) / step time / dense MFMA peak of the vocoder's dtype"}
        cfg_name = {"bf16": "configs[2]", "fp16": "configs[4] (per-GPU shard)", "fp32": "configs[2] shape in fp32",
                    "mixed": "configs[2] shape, acoustic model in fp32 (exact mel parity) + fp16 vocoder",
                    "fp32x3": "configs[2] shape, fp32 tensors, dense products as three fp16 MFMAs on split operands",
                    "mixed3": "configs[2] shape, split-fp32 acoustic model (fp32 tolerances) + fp16 vocoder"}[args.dtype]
        line = {
            "metric": "mel-frames/sec + vocoder RTF @24kHz, batch=32, 1/2/4/8 MI355X",
            "value": world * frames_out * args.steps / elapsed,
            "unit": "mel-frames/s (acoustic + vocoder end to end)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32", "bf16": "bf16", "fp16": "f16", "mixed": "f32 (acoustic) + f16 (vocoder)", "fp32x3": "f32 as 3 x f16",
                      "mixed3": "f32 as 3 x f16 (acoustic) + f16 (vocoder)"}[args.dtype], "data": "synthetic",
            "config": {"workload": f"{cfg_name}: batch={B}/GPU x {L} phonemes -> {T} frames, acoustic (PostFlow on) + {args.vocoder}, "
                                   f"gold durations {args.frames_per_phone}/phoneme (duration predictor bypassed; pitch / energy predicted), "
                                   f"pitch scale {args.pitch_scale}, energy scale {args.energy_scale}, fixture weights",
                       "global_batch": world * B, "phones": L, "frames_per_utt": frames_out // B, "vocoder": args.vocoder,
                       "acoustic_dtype": f"{args.dtype} MFMA / f32 activations" if args.dtype in ("bf16", "fp16") else "f32",
                       "vocoder_dtype": {"fp32": "f32", "bf16": "bf16", "fp16": "fp16", "mixed": "fp16", "fp32x3": "f32 as 3 x f16", "mixed3": "fp16"}[args.dtype],
                       "parallelism": f"dp{world}",
                       "hip_graphs": bool(args.graphs),
                       "sequencer": "native stage API (csrc/pipeline.hip)" if use_native else "python (engine.py)",
                       "streams": "2 HIP streams: acoustic model of step k+1 beside the vocoder of step k" if overlap else "1 HIP stream"},
            # the acoustic model alone on the GPU (one stream, measured after the timed region); in the two-stream form the vocoder's
            # time is that of the last step's vocoder running beside the next step's acoustic model
            "acoustic_mel_frames_per_s": world * frames_out * args.steps / t_ac,
            "acoustic_ms_alone": acoustic_alone_ms,
            "vocoder_rtf": t_voc / (args.steps * audio_s),
            "ms_per_step_median_events": step_ms_median, "ms_per_step_min_max_events": [step_ms[0], step_ms[-1]],
            "ms_per_step_events": [round(v, 2) for v in step_ms_in_order],
            "step_roofline": step_roof,
            "e2e_rtf": elapsed / (args.steps * audio_s * 1.0),
            "abi_calls_per_step": abi_calls,
            "pcie_inclusive": pcie,
            "roofline": roof,
            "kernel_classes_warmup_step": {k: {"ms": round(v["total_ms"], 3), "launches": v["launches"], "tflops": round(v["tflops"], 2)}
                                          for k, v in sorted(classes.items(), key=lambda kv: -kv[1]["total_ms"])},
        }
        log(f"timed region done: {1e3 * elapsed / args.steps:.1f} ms/step")
        if world == 1 and not args.no_cpu_baseline:
            log("cpu baseline (oracle, one utterance)")
            line["cpu_baseline"], oracle_out = cpu_baseline(L, args.frames_per_phone, args.vocoder, scales)
        else:
            oracle_out = None
        if use_native and not args.no_verify:
            log("verify: utterance 0 of the last timed step against a B = 1 run" + (" and the CPU oracle" if oracle_out else ""))
            line["verify"] = verify_against_single_run(pipe, last_out, last_wav, packed, z_sq, scales, B, oracle_out)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
