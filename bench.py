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
    ap.add_argument("--dtype", default="bf16", choices=["fp32", "bf16", "fp16", "mixed"],
                    help="bf16 = BASELINE.json configs[2] (bf16 MFMA GEMMs, fp32 statistics); fp16 = configs[4]'s fp16 MFMA path; "
                         "fp32 = exact-parity configuration; mixed = acoustic model in fp32 (exact mel parity) + fp16 vocoder")
    ap.add_argument("--pitch-scale", type=float, default=1.0, help="pitch_variance_scale (configs[4]: 1.3)")
    ap.add_argument("--energy-scale", type=float, default=1.0, help="energy_variance_scale (configs[4]: 0.7)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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


def cpu_baseline(phones, frames_per_phone, vocoder):
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
        o = oa(feats, emb, syn.LANG_EN, z_noise=z, durations=dur)
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
                e2e_rtf=(ta + tv) / (w.numel() / 24000.0))


def committed_counters(kernel):
    """HBM bytes per launch and VALU lane-operations per element of `kernel` from this round's committed rocprofv3 PMC passes
    (separate --pmc runs over tools/microbench_resblock.py at the bench's shapes; FETCH_SIZE doubled per MI355X_MICROARCH.md).
    Counter passes cannot share a process with the timed region, so these are read from profiles/; the file is named in the
    line.  (None, None, None) when no pass covers the kernel."""
    for tag in ("r02", "r01_v18", "r01"):
        path = os.path.join(ROOT, "profiles", f"{tag}_pmc_resblock_traffic.json")
        if not os.path.exists(path):
            continue
        with open(path) as f:
            rows = [r for r in json.load(f)["launches"] if r["kernel"] == kernel and r["act"] == "snake"]
        if not rows:
            continue
        traffic = sum(r["hbm_bytes_corrected"] for r in rows) / len(rows)
        valu = None
        sq = os.path.join(ROOT, "profiles", f"{tag}_pmc_resblock_SQ_summary.json")
        if os.path.exists(sq):
            with open(sq) as f:
                v = [r["valu_per_elem"] for r in json.load(f)["launches"] if r["kernel"] == kernel and r["act"] == "snake"]
            valu = sum(v) / len(v) if v else None
        return traffic, valu, os.path.relpath(path, ROOT)
    return None, None, None


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

    precision = {"fp32": "f32", "bf16": "bf16", "fp16": "f16", "mixed": "f32"}[args.dtype]
    voc_precision = "f16" if args.dtype == "mixed" else precision  # mixed: exact-parity acoustic model (fp32) + fp16 vocoder
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
    comm_stream_ok = [True]

    packed = z_sq = None
    if use_native:  # the stage API's input format (packed along the phoneme axis): resident in HBM before the timed region
        packed = pipe.pack_inputs(texts, embs, langs, durations=durs)
        z_sq = pipe.squeeze_noise(zs, [T] * B)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    overlap = use_native and not args.no_overlap
    s_ac = s_voc = s_comm = None
    if overlap:
        s_comm = torch.cuda.Stream(dev) if world > 1 else None
        # the acoustic stream has the higher priority: its short kernels take the CUs a vocoder launch frees in its tail before
        # the next vocoder launch's persistent workgroups do (42.0 vs 42.25 ms measured; TOUCAN_BENCH_AC_PRIORITY=0 for the A/B run)
        s_ac, s_voc = torch.cuda.Stream(dev, priority=int(os.environ.get("TOUCAN_BENCH_AC_PRIORITY", "-1"))), torch.cuda.Stream(dev)

    def step_overlapped(record=False):
        """Two streams: the acoustic model of this step is enqueued on s_ac, its vocoder on s_voc behind an event; the next
        step's acoustic model does not wait for this step's vocoder.  The mel handed over is forward()'s own copy."""
        with torch.cuda.stream(s_ac):
            if record:
                ev[0].record(s_ac)
            out = pipe.forward(None, None, packed=packed, z_sq=z_sq, vocode=False, **scales)
            if record:
                ev[1].record(s_ac)
            done = torch.cuda.Event()
            done.record(s_ac)
        with torch.cuda.stream(s_voc):
            s_voc.wait_event(done)
            out["mel_packed"].record_stream(s_voc)
            if record:
                ev[3].record(s_voc)
            wav, _ = pipe.vocode(out["mel_packed"], out["rag_mel"])
            if record:
                ev[2].record(s_voc)
            if world > 1:
                # the exchange step on its own stream: the vocoder of step k+1 does not wait for the waveforms of step k to cross
                # xGMI (31.5 MB per rank and step); all of it is inside the timed region (the final device synchronise)
                block = wav[: B * T * 384].contiguous()
                voc_done = torch.cuda.Event()
                voc_done.record(s_voc)
                dst = gathered2[gather_no[0] & 1]
                gather_no[0] += 1
                if rehearsal:
                    dist.all_gather_into_tensor(dst, block.cpu())
                elif comm_stream_ok[0]:
                    try:
                        with torch.cuda.stream(s_comm):
                            s_comm.wait_event(voc_done)
                            block.record_stream(s_comm)
                            dist.all_gather_into_tensor(dst, block)
                    except Exception as e:  # (never seen; the plain form below is the one round 1 measured)
                        log(f"exchange on its own stream failed ({e!r}): falling back to the vocoder stream")
                        comm_stream_ok[0] = False
                        dist.all_gather_into_tensor(dst, block)
                else:
                    dist.all_gather_into_tensor(dst, block)
        return out, wav

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
    for it in range(args.steps):
        log(f"timed step {it}")
        if overlap:  # no host wait between steps: the streams overlap consecutive steps; stage times from the last step only
            out, wav = step(record=(it == args.steps - 1))
        else:
            out, wav = step(record=True)
            ev[2].synchronize()
            t_ac += ev[0].elapsed_time(ev[1]) * 1e-3
            t_voc += ev[1].elapsed_time(ev[2]) * 1e-3
    torch.cuda.synchronize()
    if overlap:  # (the two stages of one step, each running beside the other stage of a neighbouring step)
        t_ac = ev[0].elapsed_time(ev[1]) * 1e-3 * args.steps
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
            _, w = step(tx=tx, em=em, zz=zz, resident=False)
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
            is16 = ("bf16" in dominant or "f16" in dominant or "resblock" in dominant) and "f32" not in dominant
            peak = PEAK_16BIT_TFLOPS if is16 else PEAK_F32_TFLOPS
            traffic, valu_per_elem, src = committed_counters(dominant)
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
        cfg_name = {"bf16": "configs[2]", "fp16": "configs[4] (per-GPU shard)", "fp32": "configs[2] shape in fp32",
                    "mixed": "configs[2] shape, acoustic model in fp32 (exact mel parity) + fp16 vocoder"}[args.dtype]
        line = {
            "metric": "mel-frames/sec + vocoder RTF @24kHz, batch=32, 1/2/4/8 MI355X",
            "value": world * frames_out * args.steps / elapsed,
            "unit": "mel-frames/s (acoustic + vocoder end to end)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32", "bf16": "bf16", "fp16": "f16", "mixed": "f32 (acoustic) + f16 (vocoder)"}[args.dtype], "data": "synthetic",
            "config": {"workload": f"{cfg_name}: batch={B}/GPU x {L} phonemes -> {T} frames, acoustic (PostFlow on) + {args.vocoder}, "
                                   f"gold durations {args.frames_per_phone}/phoneme (duration predictor bypassed; pitch / energy predicted), "
                                   f"pitch scale {args.pitch_scale}, energy scale {args.energy_scale}, fixture weights",
                       "global_batch": world * B, "phones": L, "frames_per_utt": frames_out // B, "vocoder": args.vocoder,
                       "acoustic_dtype": f"{args.dtype} MFMA / f32 activations" if args.dtype in ("bf16", "fp16") else "f32",
                       "vocoder_dtype": {"fp32": "f32", "bf16": "bf16", "fp16": "fp16", "mixed": "fp16"}[args.dtype], "parallelism": f"dp{world}",
                       "hip_graphs": bool(args.graphs),
                       "sequencer": "native stage API (csrc/pipeline.hip)" if use_native else "python (engine.py)",
                       "streams": "2 HIP streams: acoustic model of step k+1 beside the vocoder of step k" if overlap else "1 HIP stream"},
            "acoustic_mel_frames_per_s": world * frames_out * args.steps / t_ac,
            "vocoder_rtf": t_voc / (args.steps * audio_s),
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
            line["cpu_baseline"] = cpu_baseline(L, args.frames_per_phone, args.vocoder)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
