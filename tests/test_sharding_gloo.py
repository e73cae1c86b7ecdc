"""CPU, world_size 2, gloo: individuals shard across ranks with no data-path collective; the
gather of per-individual rows reproduces the single-process result bit for bit (the per-rank
compute here is the oracle -- the GPU path itself is covered by tests/test_gpu_parity.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nind, W, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import oracle_lib as ol
    from garlic_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(42)                       # same panel on every rank
    chroms = [ol.random_panel(rng, n, nind, max_gap=200000) for n in (400, 250)]
    b, e = shard.shard_range(nind, world, rank)
    local = [ol.oracle_calc_lod(g[:, b:e], f, p, cs, ce, W, 0.001, 200000) for g, f, p, cs, ce in chroms]
    dist.barrier()
    full = shard.gather_rows(local, nind)
    if rank == 0:
        want = [ol.oracle_calc_lod(g, f, p, cs, ce, W, 0.001, 200000) for g, f, p, cs, ce in chroms]
        ok = all(ol.bits_equal(a, w) for a, w in zip(full, want))
        # KDE-feed order chr -> ind -> locus is preserved by the rank-ordered gather
        feed = np.concatenate([ol.oracle_flatten(a, W) for a in full])
        feed_want = np.concatenate([ol.oracle_flatten(w, W) for w in want])
        ok = ok and ol.bits_equal(feed, feed_want)
        with open(out_path, "w") as fh:
            fh.write("ok" if ok else "mismatch")
    dist.destroy_process_group()


@pytest.mark.parametrize("nind", [10, 7])
def test_two_rank_shard_and_gather(tmp_path, nind):
    from garlic_amd import shard
    assert shard.shard_range(10, 2, 0) == (0, 5) and shard.shard_range(10, 2, 1) == (5, 10)
    assert shard.shard_range(7, 2, 1) == (4, 7) and shard.shard_range(3, 8, 5) == (3, 3)
    out = tmp_path / "result.txt"
    mp.spawn(_worker, args=(2, _free_port(), nind, 20, str(out)), nprocs=2, join=True)
    assert out.read_text() == "ok"


def _seg_worker(rank, world, port, nind, W, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import oracle_lib as ol
    from garlic_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(43)
    chroms = [ol.random_panel(rng, n, nind, max_gap=200000) for n in (900, 350)]
    cutoff, frac = -8.0, 0.25

    def segments(lo, hi):      # what garlic_roh_segments returns for the individuals [lo, hi): local indices, sorted
        out = []
        for c, (g, f, p, cs, ce) in enumerate(chroms):
            cov = ol.oracle_roh_coverage(ol.oracle_calc_lod(g[:, lo:hi], f, p, cs, ce, W, 0.001, 200000), W, cutoff)
            out += [(i, c, a, b) for i, a, b in ol.oracle_roh_segments(cov, p, cs, ce, W, 200000, frac)]
        return np.array(sorted(out), dtype=np.int32).reshape(-1, 4)

    b, e = shard.shard_range(nind, world, rank)
    full = shard.gather_segments(segments(b, e), nind)
    if rank == 0:
        want = segments(0, nind)
        with open(out_path, "w") as fh:
            fh.write("ok" if want.shape[0] > 3 and np.array_equal(full, want) else "mismatch")
    else:
        assert full is None
    dist.destroy_process_group()


@pytest.mark.parametrize("nind", [10, 7])
def test_two_rank_roh_segments_gather(tmp_path, nind):
    out = tmp_path / "result.txt"
    mp.spawn(_seg_worker, args=(2, _free_port(), nind, 20, str(out)), nprocs=2, join=True)
    assert out.read_text() == "ok"


def _ld_counts_numpy(geno, W, sub):
    """what garlic_ld_counts returns for a shard: integer counts only"""
    nloci = geno.shape[0]
    nm = geno != -9
    hom = nm & (geno != 1)
    loc = np.stack([hom.sum(1), nm.sum(1)], axis=1).astype(np.int32)
    pair = np.zeros((nloci, W, 2), dtype=np.int32)
    s_nm, s_hom = nm[:, sub], hom[:, sub]
    for d in range(1, W):
        pair[:nloci - d, d, 0] = (s_nm[:nloci - d] & s_nm[d:]).sum(1)
        pair[:nloci - d, d, 1] = (s_hom[:nloci - d] & s_hom[d:]).sum(1)
    return loc, pair


def _ld_finish_numpy(loc, pair, W):
    """what garlic_ld_finish does with the summed counts (garlic-data.cpp:474-583), scalar FP64"""
    nloci = loc.shape[0]
    with np.errstate(invalid="ignore", divide="ignore"):
        hf = loc[:, 0].astype(np.float64) / loc[:, 1].astype(np.float64)

        def hr2(i, j):
            HA, HB = hf[i], hf[j]
            if not (0 < HA < 1 and 0 < HB < 1):
                return 0.0
            lo, d = min(i, j), abs(i - j)
            HAB = np.float64(pair[lo, d, 1]) / np.float64(pair[lo, d, 0])
            H = HAB - HA * HB
            v = H * H / (HA * (1 - HA) * HB * (1 - HB))
            return 1.0 if v > 1 else v

        ld = np.zeros((nloci, W))
        for s in range(nloci - W + 1):
            for k in range(W):
                acc = np.float64(0.0)
                for i in range(s, s + W):
                    acc = acc + (1.0 if i == s + k else hr2(i, s + k))
                ld[s, k] = acc
    return ld


def _ld_worker(rank, world, port, nind, W, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import oracle_lib as ol
    from garlic_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(7)
    geno = ol.random_panel(rng, 90, nind, max_gap=10 ** 9, gaps=0, miss=0.05)[0]
    sub = np.sort(rng.choice(nind, size=nind // 2, replace=False)).astype(np.int32)
    b, e = shard.shard_range(nind, world, rank)
    loc, pair = _ld_counts_numpy(geno[:, b:e], W, shard.split_subsample(sub, nind, world, rank))
    loc, pair = shard.allreduce_ld_counts(loc, pair)            # the path's one collective
    ld = _ld_finish_numpy(loc, pair, W)
    ok = ol.bits_equal(ld, ol.oracle_hr2_ld(geno, W, idx=sub))  # every rank holds the full-panel weights
    flags = [None] * world
    dist.all_gather_object(flags, bool(ok))
    if rank == 0:
        with open(out_path, "w") as fh:
            fh.write("ok" if all(flags) else "mismatch")
    dist.destroy_process_group()


def test_two_rank_ld_counts_allreduce(tmp_path):
    """LD weights shard as integer counts + one all-reduce; the replicated floating-point finish then
    equals the single-process oracle bit for bit on every rank"""
    out = tmp_path / "ld.txt"
    mp.spawn(_ld_worker, args=(2, _free_port(), 21, 6, str(out)), nprocs=2, join=True)
    assert out.read_text() == "ok"
