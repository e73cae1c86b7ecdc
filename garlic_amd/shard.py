"""Individual sharding across the GPUs of a node.

Individuals are independent on this path (reference src/garlic-roh.cpp:46 iterates them one by
one), so rank r owns a contiguous block of the TFAM order, gets ALL SNPs for it plus replicated
per-SNP tables, and there is no collective on the data path.  The only communication is the
host-side gather of per-individual score rows, in rank order, because the KDE feed iterates
chromosome -> individual -> locus (src/garlic-data.cpp:2033-2037)."""
import numpy as np


def shard_range(nind, world, rank):
    """Contiguous block [begin, end) of individuals for `rank`; blocks of ceil(nind/world)."""
    per = (nind + world - 1) // world
    begin = min(nind, rank * per)
    return begin, min(nind, begin + per)


def gather_rows(local_rows, nind, group=None):
    """local_rows: list (one per chromosome) of float64 [n_local][nloci_c] arrays of this rank's
    individuals.  Returns on rank 0 the list of [nind][nloci_c] arrays in TFAM order (None on the
    other ranks).  Works with any torch.distributed backend (gloo on CPU, nccl = RCCL on GPUs)."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    out = [] if rank == 0 else None
    for rows in local_rows:
        rows = np.ascontiguousarray(rows, dtype=np.float64)
        nloci = rows.shape[1]
        per = (nind + world - 1) // world
        # equal-sized buffers for gather: pad the last block
        buf = torch.zeros((per, nloci), dtype=torch.float64)
        buf[: rows.shape[0]] = torch.from_numpy(rows)
        parts = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
        dist.gather(buf, parts, dst=0, group=group)
        if rank == 0:
            full = np.empty((nind, nloci), dtype=np.float64)
            for r in range(world):
                b, e = shard_range(nind, world, r)
                full[b:e] = parts[r][: e - b].numpy()
            out.append(full)
    return out


def gather_segments(local_segs, nind, group=None):
    """local_segs: this rank's garlic_roh_segments result, int32 [n][4] of (shard-local individual, chromosome, first
    SNP, last SNP), ordered by individual.  Returns on rank 0 the panel-wide list -- individuals offset by the shard's
    first one, shards in rank order = the order assembleROHWindows appends in (src/garlic-roh.cpp:425, individual by
    individual) -- and None on the other ranks.  A few KB per rank: no score row or coverage count crosses ranks."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    seg = np.ascontiguousarray(local_segs, dtype=np.int32).reshape(-1, 4)
    counts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)] if rank == 0 else None
    dist.gather(torch.tensor([seg.shape[0]], dtype=torch.int64), counts, dst=0, group=group)
    cap = torch.zeros(1, dtype=torch.int64)
    if rank == 0:
        cap[0] = max(int(c.item()) for c in counts)
    dist.broadcast(cap, src=0, group=group)
    buf = torch.zeros((max(int(cap.item()), 1), 4), dtype=torch.int32)
    buf[: seg.shape[0]] = torch.from_numpy(seg)
    parts = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, parts, dst=0, group=group)
    if rank != 0:
        return None
    out = []
    for r in range(world):
        part = parts[r][: int(counts[r].item())].numpy().copy()
        part[:, 0] += shard_range(nind, world, r)[0]
        out.append(part)
    return np.concatenate(out, axis=0)


def split_subsample(sub_idx, nind, world, rank):
    """The part of a panel-wide LD subsample (sorted individual indices, src/garlic-data.cpp:361)
    that lives in `rank`'s block, as shard-local indices; None (= everyone) stays None."""
    if sub_idx is None:
        return None
    sub = np.asarray(sub_idx, dtype=np.int64)
    b, e = shard_range(nind, world, rank)
    return (sub[(sub >= b) & (sub < e)] - b).astype(np.int32)


def allreduce_ld_counts(locus_counts, pair_counts, group=None):
    """The one collective of the weighted path: LD weights count individuals (hr2,
    src/garlic-data.cpp:558-583), and a shard only sees its own.  Sums the integer count arrays of
    garlic_ld_counts element-wise over all ranks, in place -- exact in any order, so every rank then
    finishes (garlic_ld_finish) to bit-identical weights.  Accepts torch tensors (int32; device
    tensors with the nccl = RCCL backend, host tensors with gloo) or numpy arrays (summed through a
    host tensor)."""
    import torch
    import torch.distributed as dist

    outs = []
    for a in (locus_counts, pair_counts):
        t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32))
        assert t.dtype == torch.int32
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        outs.append(t if isinstance(a, torch.Tensor) else t.numpy())
    return outs[0], outs[1]
