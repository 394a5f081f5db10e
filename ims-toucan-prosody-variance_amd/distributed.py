"""Multi-GPU path: utterances are independent (the reference synthesises them one at a time,
ToucanTTSInterface.py:269-280), so a batch is dealt over the ranks of one node (one process per GPU, weights
replicated) and the only exchange step is one all-gather of the decoded waveforms over RCCL/xGMI
(backend "nccl" on ROCm; "gloo" for the CPU tests).  No collective touches the data path before that.

Results are bit-identical to the 1-GPU run: every utterance goes through the same kernels with the same
per-utterance arithmetic (ragged tiles never mix utterances).
"""
import torch
import torch.distributed as dist


def deal_by_length(lengths, world):
    """Length-balanced assignment: sort by length (descending) and deal snake-wise.  Returns world lists of indices."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    shards = [[] for _ in range(world)]
    for pos, idx in enumerate(order):
        rnd, off = divmod(pos, world)
        r = off if rnd % 2 == 0 else world - 1 - off
        shards[r].append(idx)
    return shards


def all_gather_waveforms(local_waves, device):
    """local_waves: list of 1-D float tensors on `device`.  Returns (per-rank list of lists of tensors)."""
    world = dist.get_world_size()
    n_local = torch.tensor([len(local_waves)], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local)
    n_max = max(int(c.item()) for c in counts)
    lens = torch.zeros(max(n_max, 1), dtype=torch.int64, device=device)
    for i, w in enumerate(local_waves):
        lens[i] = w.numel()
    all_lens = [torch.zeros_like(lens) for _ in range(world)]
    dist.all_gather(all_lens, lens)
    s_max = max(1, max(int(l.max().item()) for l in all_lens))
    # one padded [n_max, s_max] block per rank: a direct all-gather puts each peer's block on its own xGMI link
    block = torch.zeros(max(n_max, 1), s_max, dtype=torch.float32, device=device)
    for i, w in enumerate(local_waves):
        block[i, : w.numel()] = w
    gathered = [torch.empty_like(block) for _ in range(world)]
    dist.all_gather(gathered, block)
    out = []
    for r in range(world):
        out.append([gathered[r][i, : int(all_lens[r][i].item())] for i in range(int(counts[r].item()))])
    return out


def synthesize_sharded(iface, feats, embs, z_noise, durations, pitch, energy, kw):
    """Shard `feats` over the ranks, synthesise the local shard, all-gather; every rank returns all waveforms in input order."""
    assert dist.is_available() and dist.is_initialized(), "distributed=True needs an initialised process group"
    world, rank = dist.get_world_size(), dist.get_rank()
    shards = deal_by_length([f.shape[0] for f in feats], world)
    mine = shards[rank]
    pick = lambda lst: None if lst is None else [lst[i] for i in mine]
    with torch.inference_mode():
        if mine:
            local = iface._synthesize([feats[i] for i in mine], [embs[i] for i in mine], [iface._lang()] * len(mine),
                                      z_noise=pick(z_noise), durations=pick(durations), pitch=pick(pitch), energy=pick(energy), **kw)
        else:
            local = []
        dev = torch.device(iface.device)
        per_rank = all_gather_waveforms(local, dev)
    out = [None] * len(feats)
    for r in range(world):
        for i, w in zip(shards[r], per_rank[r]):
            out[i] = w
    return out
