"""Multi-GPU path: utterances are independent (the reference synthesises them one at a time,
ToucanTTSInterface.py:269-280), so a batch is dealt over the ranks of one node (one process per GPU, weights
replicated) and the only exchange step of the data path is the all-gather of the decoded waveforms over RCCL/xGMI
(backend "nccl" on ROCm; "gloo" for the CPU tests).

Exchange: TWO collectives per batch, both `all_gather_into_tensor` on the compute stream -
  1. `i32[n_max, 2]` per rank: (first sample, sample count) of each local utterance inside the rank's packed waveform;
  2. the packed waveform itself, `f32[S_max]` per rank (S_max = the longest packed waveform of any rank; only that tail is
     padding - the vocoder's own packed output is sent as it is, no per-utterance [n_max, S_max] staging block)
- and ONE host read (of collective 1's result, which sizes collective 2).  Each peer's block travels on its own xGMI link
(direct all-gather, 7 links x ~153 GB/s per GPU); at batch 32 x 10.24 s that is 31.5 MB per rank, ~0.2 ms.

Sharding key: the number of mel FRAMES of each utterance (vocoder work is proportional to it, 90 % of the step), not its
phoneme count.  With gold durations the count is known on the host; with predicted durations every rank first runs the cheap
stage A (encoder + predictors, < 1 % of an utterance's work) on a phoneme-balanced deal, the frame counts are all-gathered
(one more `i32[n_max]` collective) and the batch is dealt again by frames.

What a shard changes: nothing but the grid sizes (ragged tiles never mix utterances).  In the 16-bit configurations (bf16 / fp16)
an utterance's result is bit-identical whatever batch or shard it is in.  In the fp32 configuration the PHONEME stages (encoder,
predictors - everything upstream of the rounded durations) keep one arithmetic at every grid size, so frame counts, pitch and
energy are bit-identical too; the frame stages may take the split-K / key-split forms on the small grids of a small shard, so the
mel - and the waveform made from it - agree with the 1-GPU run to fp32 rounding order (max-abs ~3e-5 measured), not bit for bit.
(The gloo tests on the CPU emulator see bit-identity because the emulator has no split forms.)
"""
import torch
import torch.distributed as dist

from .phonemes import IDX


def deal_by_length(lengths, world):
    """Cost-balanced assignment: sort by cost (descending) and deal snake-wise.  Returns `world` lists of indices."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    shards = [[] for _ in range(world)]
    for pos, idx in enumerate(order):
        rnd, off = divmod(pos, world)
        r = off if rnd % 2 == 0 else world - 1 - off
        shards[r].append(idx)
    return shards


def _gather_i32(local, n_max, width, device):
    """all_gather_into_tensor of an i32[n_max, width] block per rank; `local` is a host list of rows.  Returns a host tensor
    [world, n_max, width] (the one device->host read of the exchange)."""
    world = dist.get_world_size()
    block = torch.zeros(max(n_max, 1), width, dtype=torch.int32)
    if local:
        block[: len(local)] = torch.tensor(local, dtype=torch.int32).reshape(len(local), width)
    block = block.to(device)
    out = torch.empty(world * block.numel(), dtype=torch.int32, device=device)
    dist.all_gather_into_tensor(out, block.reshape(-1))
    return out.cpu().reshape(world, max(n_max, 1), width)


def all_gather_packed_waveforms(wav, spans, n_per_rank, device):
    """wav: this rank's packed waveform (1-D float tensor on `device`, possibly empty); spans: host list of (first sample, count)
    per local utterance.  n_per_rank: utterances per rank (known to every rank from the deterministic deal).
    Returns, per rank, the list of that rank's waveforms (views of the gathered buffer)."""
    world = dist.get_world_size()
    n_max = max(n_per_rank) if n_per_rank else 0
    table = _gather_i32([list(s) for s in spans], n_max, 2, device)  # collective 1 + the host read
    totals = [max([int(table[r, i, 0] + table[r, i, 1]) for i in range(n_per_rank[r])] + [0]) for r in range(world)]
    s_max = max(1, max(totals))
    block = wav.new_zeros(s_max) if wav.numel() != s_max else wav
    if wav.numel() != s_max and wav.numel():
        block[: min(wav.numel(), s_max)].copy_(wav[:s_max])
    gathered = torch.empty(world * s_max, dtype=torch.float32, device=device)
    dist.all_gather_into_tensor(gathered, block.contiguous())  # collective 2
    out = []
    for r in range(world):
        base = r * s_max
        out.append([gathered[base + int(table[r, i, 0]): base + int(table[r, i, 0]) + int(table[r, i, 1])] for i in range(n_per_rank[r])])
    return out


def _host_frame_counts(feats, durations):
    """Σ gold durations per utterance with the word-boundary zeroing the control step applies (InferenceToucanTTS.py:219-220)."""
    out = []
    for f, d in zip(feats, durations):
        d = torch.as_tensor(d).reshape(-1).to(torch.int64).cpu()
        wb = torch.as_tensor(f)[:, IDX["word_boundary"]].cpu() == 1
        out.append(int(d[~wb].sum()))
    return out


def frame_costs(iface, feats, embs, durations, pitch, energy, kw):
    """Frames per utterance, identical on every rank.  Gold durations: computed on the host.  Predicted durations: stage A on a
    phoneme-balanced deal, then one i32 all-gather."""
    if durations is not None:
        return _host_frame_counts(feats, durations)
    world, rank = dist.get_world_size(), dist.get_rank()
    shards = deal_by_length([f.shape[0] for f in feats], world)
    mine = shards[rank]
    pick = lambda lst: None if lst is None else [lst[i] for i in mine]
    local = []
    if mine:
        local = iface.predict_frame_counts([feats[i] for i in mine], [embs[i] for i in mine], pitch=pick(pitch), energy=pick(energy), **kw)
    n_per = [len(s) for s in shards]
    table = _gather_i32([[c] for c in local], max(n_per), 1, torch.device(iface.device))
    costs = [0] * len(feats)
    for r in range(world):
        for j, i in enumerate(shards[r]):
            costs[i] = int(table[r, j, 0])
    return costs


def synthesize_sharded(iface, feats, embs, z_noise, durations, pitch, energy, kw):
    """Shard `feats` over the ranks by frame count, synthesise the local shard, all-gather; every rank returns all waveforms in
    input order."""
    assert dist.is_available() and dist.is_initialized(), "distributed=True needs an initialised process group"
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = torch.device(iface.device)
    with torch.inference_mode():
        costs = frame_costs(iface, feats, embs, durations, pitch, energy, kw)
        shards = deal_by_length(costs, world)
        mine = shards[rank]
        pick = lambda lst: None if lst is None else [lst[i] for i in mine]
        if mine:
            wav, spans = iface._synthesize_packed([feats[i] for i in mine], [embs[i] for i in mine], [iface._lang()] * len(mine),
                                                  z_noise=pick(z_noise), durations=pick(durations), pitch=pick(pitch),
                                                  energy=pick(energy), **kw)
        else:
            wav, spans = torch.zeros(0, dtype=torch.float32, device=dev), []
        per_rank = all_gather_packed_waveforms(wav, spans, [len(s) for s in shards], dev)
    out = [None] * len(feats)
    for r in range(world):
        for i, w in zip(shards[r], per_rank[r]):
            out[i] = w
    return out
