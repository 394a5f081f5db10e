"""bench.py - throughput of the ToucanTTS hot path on MI355X (metric of BASELINE.json).

One step = one pass of the hot path over one batch of synthetic utterances per GPU:
acoustic model (Conformer enc -> predictors bypassed by gold durations -> length regulator -> Conformer dec
-> PostNet -> PostFlow) + BigVGAN vocoder, batch 32 x 128 phonemes per GPU (BASELINE.json configs[2]; at N
GPUs the job is N x 32 utterances, weak scaling, waveforms all-gathered over RCCL).  Inputs are resident in
HBM before the timed region.  Prints ONE JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 32] [--phones 128] [--vocoder bigvgan|hifigan]
                    [--dtype fp32|bf16] [--no-cpu-baseline]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import ims_toucan_prosody_variance_amd  # noqa: E402,F401
from ims_toucan_prosody_variance_amd import engine, fixture_weights as fw, profiling, synthetic as syn  # noqa: E402

PEAK_F32_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 matrix peak (v_mfma_f32_32x32x2_f32)
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak


def cpu_baseline(phones, frames_per_phone, vocoder):
    """The CPU oracle (kind 'port') on a bounded sample: ONE utterance of the same workload, processed the way the
    reference's read_to_file does (one utterance at a time), all host cores."""
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


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/r01_pmc_resblock_traffic.json:
    FETCH_SIZE and WRITE_SIZE in separate --pmc runs, FETCH_SIZE doubled per MI355X_MICROARCH.md).  None if not collected."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_resblock_traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        rows = [r for r in json.load(f)["launches"] if r["kernel"] == kernel and r["act"] == "snake"]
    if not rows:
        return None
    return sum(r["hbm_bytes_corrected"] for r in rows) / len(rows)


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU")
    ap.add_argument("--phones", type=int, default=128)
    ap.add_argument("--frames-per-phone", type=int, default=5)
    ap.add_argument("--vocoder", default="bigvgan", choices=["bigvgan", "hifigan"])
    ap.add_argument("--dtype", default="bf16", choices=["fp32", "bf16"],
                    help="bf16 = BASELINE.json configs[2] (bf16 MFMA GEMMs, fp32 statistics); fp32 = exact-parity configuration")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fuse-snake", action="store_true", help="BigVGAN: anti-aliased snake inside the conv input staging")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
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

    bf16 = args.dtype == "bf16"
    log("building engines (fixture weights)")
    ac = engine.AcousticEngine(fw.acoustic_state_dict(), dev, bf16=bf16)
    voc_sd = fw.bigvgan_state_dict() if args.vocoder == "bigvgan" else fw.hifigan_state_dict()
    voc = engine.VocoderEngine(voc_sd, args.vocoder, dev, bf16=bf16, fuse_snake=args.fuse_snake)

    B, L, T = args.batch, args.phones, args.phones * args.frames_per_phone
    log(f"synthetic inputs: {B} x {L} phonemes -> {T} frames per utterance")
    ids = [rank * B + u for u in range(B)]
    texts = [torch.from_numpy(syn.utterance_features(u, L, word_boundaries=False)).to(dev) for u in ids]
    embs = torch.from_numpy(np.stack([syn.utterance_embedding(u) for u in ids])).to(dev)
    durs = [torch.full((L,), args.frames_per_phone, dtype=torch.int32, device=dev) for _ in ids]
    zs = [torch.from_numpy(syn.postflow_noise(u, T)).to(dev) for u in ids]
    langs = [syn.LANG_EN] * B
    # 1-D concatenation form of the gather output (accepted by every backend)
    gathered = torch.empty(world * B * T * 384, device="cpu" if rehearsal else dev) if world > 1 else None

    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]

    def step(record=False):
        if record:
            ev[0].record()
        out = ac.forward(texts, embs, langs, durations=durs, z_noise=zs)
        if record:
            ev[1].record()
        wav, rag = voc.forward(out["mel_packed"], out["rag_mel"])
        if record:
            ev[2].record()
        if world > 1:  # one exchange step: waveforms of all ranks (equal length here) over RCCL/xGMI
            block = wav[: B * T * 384].contiguous()
            dist.all_gather_into_tensor(gathered, block.cpu() if rehearsal else block)
        return out, wav

    # ---- warm-up; the first warm-up step times every conv class to find the dominant kernel ----
    log("warm-up step 1 (untimed: first-launch costs)")
    step()
    torch.cuda.synchronize()
    timer = profiling.ConvTimer()
    ac.ops.timer = voc.ops.timer = timer
    log("warm-up step 2 (all MFMA kernel classes timed)")
    step()
    torch.cuda.synchronize()
    classes = timer.summary()
    dominant = max(classes, key=lambda k: classes[k]["total_ms"])
    ac.ops.timer = voc.ops.timer = None
    for _ in range(max(0, args.warmup - 2)):
        step()
    # ---- timed region: only the dominant class carries event pairs (a few dozen launches per step) ----
    timer = profiling.ConvTimer(select={dominant})
    ac.ops.timer = voc.ops.timer = timer
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    t_ac = t_voc = 0.0
    for it in range(args.steps):
        log(f"timed step {it}")
        out, wav = step(record=True)
        ev[2].synchronize()
        t_ac += ev[0].elapsed_time(ev[1]) * 1e-3
        t_voc += ev[1].elapsed_time(ev[2]) * 1e-3
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device="cpu" if rehearsal else dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    frames_out = int(sum(m.shape[0] for m in out["mel"]))
    audio_s = frames_out * 384 / 24000.0
    dom = timer.summary()[dominant]
    peak = PEAK_BF16_TFLOPS if ("bf16" in dominant or "resblock" in dominant) else PEAK_F32_TFLOPS
    if rank == 0:
        line = {
            "metric": "mel-frames/sec + vocoder RTF @24kHz, batch=32, 1/2/4/8 MI355X",
            "value": world * frames_out * args.steps / elapsed,
            "unit": "mel-frames/s (acoustic + vocoder end to end)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if bf16 else "f32", "data": "synthetic",
            "config": {"workload": f"configs[2]: batch={B}/GPU x {L} phonemes -> {T} frames, acoustic (PostFlow on) + {args.vocoder}, "
                                   f"gold durations {args.frames_per_phone}/phoneme, fixture weights",
                       "global_batch": world * B, "phones": L, "frames_per_utt": frames_out // B, "vocoder": args.vocoder,
                       "acoustic_dtype": "bf16 MFMA / f32 activations" if bf16 else "f32", "vocoder_dtype": "bf16" if bf16 else "f32", "parallelism": f"dp{world}"},
            "acoustic_mel_frames_per_s": world * frames_out * args.steps / t_ac,
            "vocoder_rtf": t_voc / (args.steps * audio_s),
            "e2e_rtf": elapsed / (args.steps * audio_s * 1.0),
            "roofline": {"bound": "mfma", "kernel": dominant, "achieved": dom["tflops"], "peak": peak, "unit": "TFLOP/s",
                         "frac": dom["tflops"] / peak, "traffic": pmc_traffic(dominant), "avg_launch_us": dom["avg_us"],
                         "launches_per_step": dom["launches"] / args.steps, "flops_per_launch": dom["flops_per_launch"],
                         "share_of_step": dom["total_ms"] / (1e3 * elapsed)},
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
