"""GPU parity: LD weights (calcHR2LD) computed on the device vs the CPU oracle, bit for bit, and
the sharded two-step form (integer counts summed over shards)."""
import numpy as np
import pytest

import oracle_lib as ol
from garlic_amd import abi

pytestmark = pytest.mark.gpu


def make_panel(ctx, chroms, nind):
    panel = abi.Panel(ctx, [c[0].shape[0] for c in chroms], nind)
    panel.set_map(np.concatenate([c[2] for c in chroms]), [c[3] for c in chroms], [c[4] for c in chroms],
                  gpos=np.concatenate([c[2] for c in chroms]) * 1e-6)
    panel.set_freq(np.concatenate([c[1] for c in chroms]))
    panel.set_genotypes(np.concatenate([c[0] for c in chroms], axis=0))
    return panel


def oracle_ld(chroms, W, sub=None):
    out = []
    for g, *_ in chroms:
        ld = ol.oracle_hr2_ld(g, W, idx=sub)
        out.append(ld)
    return np.concatenate(out, axis=0)


def same(a, b):
    """bit-equal, NaNs included: a pair no individual of the subsample has both genotypes for is 0/0 =
    x86's default NaN (sign bit set), which the kernels reproduce"""
    return ol.bits_equal(a, b)


@pytest.mark.parametrize("W", [2, 7, 30, 100])
@pytest.mark.parametrize("nind", [40, 64, 150])
def test_ld_matches_oracle(gpu_ctx, W, nind):
    rng = np.random.default_rng(17 * W + nind)
    sizes = [400, 1, max(1, W - 1), W, W + 1, 257]
    chroms = [ol.random_panel(rng, n, nind, max_gap=10 ** 9, gaps=0, miss=0.05) for n in sizes]
    # monomorphic and all-heterozygous SNPs (homFreq 1 / 0 -> hr2 = 0), an all-missing SNP (NaN homFreq)
    g0 = chroms[0][0]
    g0[5, :] = 2
    g0[9, :] = 1
    g0[13, :] = -9
    with make_panel(gpu_ctx, chroms, nind) as panel:
        got = panel.compute_ld(W)
        assert same(got, oracle_ld(chroms, W))
        sub = np.sort(rng.choice(nind, size=max(2, nind // 3), replace=False)).astype(np.int32)
        got = panel.compute_ld(W, sub_idx=sub)
        assert same(got, oracle_ld(chroms, W, sub))


def test_ld_feeds_wlod(gpu_ctx):
    """compute_ld installs the weights: wLOD from them == oracle wLOD from oracle LD"""
    rng = np.random.default_rng(3)
    W, nind, mg = 20, 70, 200000
    chroms = [ol.random_panel(rng, n, nind, max_gap=mg, mono=0.0) for n in (600, 300)]
    gpos = [c[2] * 1e-6 for c in chroms]
    with make_panel(gpu_ctx, chroms, nind) as panel:
        ld = panel.compute_ld(W)
        out = panel.wlod_windows(W, 0.001, mg, 7, 1e-9, pitch_align=32)
    off = 0
    for c, (g, f, p, cs, ce) in enumerate(chroms):
        ldc = ld[off:off + g.shape[0]]
        off += g.shape[0]
        assert same(ldc, ol.oracle_hr2_ld(g, W))
        want = ol.oracle_calc_wlod(g, f, p, gpos[c], ldc, cs, ce, W, 0.001, mg, 1e-9, 7)
        assert same(out[c], want)


def test_ld_sharded_counts_sum_to_the_whole(gpu_ctx):
    """two shards of individuals: counts add up, both shards finish to the full-panel LD"""
    rng = np.random.default_rng(11)
    W, nind = 25, 130
    chroms = [ol.random_panel(rng, n, nind, max_gap=10 ** 9, gaps=0, miss=0.03) for n in (300, 180)]
    sub = np.sort(rng.choice(nind, size=60, replace=False)).astype(np.int32)
    want = oracle_ld(chroms, W, sub)
    cut = 70
    parts = []
    for lo, hi in ((0, cut), (cut, nind)):
        shard = [(g[:, lo:hi].copy(), f, p, cs, ce) for g, f, p, cs, ce in chroms]
        parts.append((make_panel(gpu_ctx, shard, hi - lo), sub[(sub >= lo) & (sub < hi)] - lo))
    counts = [panel.ld_counts(W, sub_idx=s) for panel, s in parts]
    loc = counts[0][0] + counts[1][0]
    pair = counts[0][1] + counts[1][1]
    for panel, _ in parts:
        assert same(panel.ld_finish(W, loc, pair), want)
        panel.close()


def test_nan_weights_reach_wlod_with_the_x86_sign(gpu_ctx):
    """two neighbouring SNPs genotyped in disjoint halves of the panel: hr2 = 0/0.  The NaN is x86's
    default one (-nan) in the LD weights and in every wLOD score that uses them, bit for bit"""
    rng = np.random.default_rng(31)
    W, nind, mg = 20, 70, 200000
    chroms = [ol.random_panel(rng, n, nind, max_gap=mg, mono=0.0, gaps=0, centro=False) for n in (300, 120)]
    g0 = chroms[0][0]
    g0[40, :35] = -9
    g0[41, 35:] = -9
    gpos = [c[2] * 1e-6 for c in chroms]
    with make_panel(gpu_ctx, chroms, nind) as panel:
        ld = panel.compute_ld(W)
        out = panel.wlod_windows(W, 0.001, mg, 7, 1e-9, pitch_align=32)
    want_ld = oracle_ld(chroms, W)
    assert np.isnan(want_ld).any() and (np.signbit(want_ld[np.isnan(want_ld)])).all()
    assert same(ld, want_ld)
    off = 0
    for c, (g, f, p, cs, ce) in enumerate(chroms):
        ldc = want_ld[off:off + g.shape[0]]
        off += g.shape[0]
        want = ol.oracle_calc_wlod(g, f, p, gpos[c], ldc, cs, ce, W, 0.001, mg, 1e-9, 7)
        assert c != 0 or np.isnan(want).any()
        assert same(out[c], want), c


def test_ld_shard_without_any_subsample_member(gpu_ctx):
    """a panel-wide LD subsample that lies entirely in the first shard: the second shard passes an EMPTY
    list (non-NULL, n_sub = 0) -- zero pair counts, its locus counts still over all its individuals --
    and the finished weights equal the single-panel ones (NULL would have meant "everyone")"""
    rng = np.random.default_rng(13)
    W, nind, cut = 15, 130, 70
    chroms = [ol.random_panel(rng, n, nind, max_gap=10 ** 9, gaps=0, miss=0.03) for n in (260, 140)]
    sub = np.sort(rng.choice(cut, size=30, replace=False)).astype(np.int32)      # all below the cut
    want = oracle_ld(chroms, W, sub)
    parts = []
    for lo, hi in ((0, cut), (cut, nind)):
        shard = [(g[:, lo:hi].copy(), f, p, cs, ce) for g, f, p, cs, ce in chroms]
        parts.append((make_panel(gpu_ctx, shard, hi - lo), (sub[(sub >= lo) & (sub < hi)] - lo).astype(np.int32)))
    assert parts[1][1].shape[0] == 0
    counts = [panel.ld_counts(W, sub_idx=s) for panel, s in parts]
    assert not counts[1][1].any() and counts[1][0][:, 1].max() > 0
    loc = counts[0][0] + counts[1][0]
    pair = counts[0][1] + counts[1][1]
    for panel, _ in parts:
        assert same(panel.ld_finish(W, loc, pair), want)
        panel.close()


def test_ld_bad_subsample_is_refused(gpu_ctx):
    rng = np.random.default_rng(1)
    chroms = [ol.random_panel(rng, 50, 10, max_gap=10 ** 9, gaps=0)]
    with make_panel(gpu_ctx, chroms, 10) as panel:
        for bad in ([0, 10], [-1], [3, 3]):
            with pytest.raises(abi.GarlicError) as e:
                panel.compute_ld(5, sub_idx=np.array(bad, dtype=np.int32))
            assert e.value.code == abi.ERR_INVALID


def test_ld_sharded_counts_on_device(gpu_ctx):
    """the device-resident form of the sharded LD step: int32 count tensors stay on the GPU, are summed
    there (what an RCCL all-reduce does across GPUs) and finished in place"""
    import torch
    rng = np.random.default_rng(12)
    W, nind = 12, 100
    chroms = [ol.random_panel(rng, n, nind, max_gap=10 ** 9, gaps=0, miss=0.02) for n in (150, 90)]
    nloci = sum(c[0].shape[0] for c in chroms)
    want = oracle_ld(chroms, W)
    shards = []
    for lo, hi in ((0, 37), (37, 100)):
        shard = [(g[:, lo:hi].copy(), f, p, cs, ce) for g, f, p, cs, ce in chroms]
        panel = make_panel(gpu_ctx, shard, hi - lo)
        loc = torch.zeros((nloci, 2), dtype=torch.int32, device="cuda")
        pair = torch.zeros((nloci, W, 2), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        panel.ld_counts_device(W, loc.data_ptr(), pair.data_ptr())
        shards.append((panel, loc, pair))
    loc = shards[0][1] + shards[1][1]
    pair = shards[0][2] + shards[1][2]
    torch.cuda.synchronize()
    for panel, _, _ in shards:
        ld = torch.empty((nloci, W), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        panel.ld_finish_device(W, loc.data_ptr(), pair.data_ptr(), ld.data_ptr())
        assert same(ld.cpu().numpy(), want)
        panel.close()


def test_ld_golden_from_the_reference_build(gpu_ctx):
    """tests/golden/wlod.npz holds calcHR2LD's own output (tools/make_golden.py, real reference build)"""
    import os
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wlod.npz"))
    cs, ce = (int(x) for x in d["centro"])
    n, nind = d["geno"].shape
    with abi.Panel(gpu_ctx, [n], nind) as panel:
        panel.set_map(d["pos"], [cs], [ce], gpos=d["gpos"])
        panel.set_freq(d["freq"])
        panel.set_genotypes(d["geno"])
        for W in (10, 30):
            assert same(panel.compute_ld(W), d[f"ld_W{W}"]), W



def oracle_r2(chroms, phase, W, sub=None):
    out, l0 = [], 0
    for g, f, *_ in chroms:
        out.append(ol.oracle_r2_ld(g, phase[l0:l0 + g.shape[0]], f, W, idx=sub))
        l0 += g.shape[0]
    return np.concatenate(out, axis=0)


@pytest.mark.parametrize("W", [2, 9, 40])
@pytest.mark.parametrize("nind", [33, 64, 150])
def test_phased_ld_matches_oracle(gpu_ctx, W, nind):
    """--phased: calcR2LD / r2 (garlic-data.cpp:426-535, 585-617) from the firstCopy bits"""
    rng = np.random.default_rng(5 * W + nind)
    sizes = [300, 1, max(1, W - 1), W, W + 1, 129]
    chroms = [list(ol.random_panel(rng, n, nind, max_gap=10 ** 9, gaps=0, miss=0.05)) for n in sizes]
    chroms[0][1][3] = 0.0                                  # frequency 0 / 1: r2 = 0 whatever the genotypes
    chroms[0][1][4] = 1.0
    chroms[0][0][13, :] = -9                               # no pair has both genotypes: 0/0
    nloci = sum(sizes)
    phase = rng.integers(0, 2, size=(nloci, nind)).astype(np.uint8)
    with make_panel(gpu_ctx, chroms, nind) as panel:
        with pytest.raises(abi.GarlicError):
            panel.compute_ld(W, phased=True)               # no phase yet
        panel.set_phase(phase[:100])                       # streamed in two chunks
        panel.set_phase(phase[100:], locus_begin=100)
        assert same(panel.compute_ld(W, phased=True), oracle_r2(chroms, phase, W))
        sub = np.sort(rng.choice(nind, size=max(2, nind // 3), replace=False)).astype(np.int32)
        assert same(panel.compute_ld(W, sub_idx=sub, phased=True), oracle_r2(chroms, phase, W, sub))
        assert same(panel.compute_ld(W), oracle_ld(chroms, W))     # the unphased weights are still hr2
        panel.release_scratch()                                     # cached LD buffers gone: allocated again
        assert same(panel.compute_ld(W, phased=True), oracle_r2(chroms, phase, W))


def test_phased_ld_sharded(gpu_ctx):
    rng = np.random.default_rng(12)
    W, nind = 20, 130
    chroms = [ol.random_panel(rng, n, nind, max_gap=10 ** 9, gaps=0, miss=0.03) for n in (200, 90)]
    phase = rng.integers(0, 2, size=(290, nind)).astype(np.uint8)
    want = oracle_r2(chroms, phase, W)
    parts = []
    for lo, hi in ((0, 70), (70, nind)):
        shard = [(g[:, lo:hi].copy(), f, p, cs, ce) for g, f, p, cs, ce in chroms]   # freq: whole panel's
        panel = make_panel(gpu_ctx, shard, hi - lo)
        panel.set_phase(phase[:, lo:hi].copy())
        parts.append(panel)
    counts = [panel.ld_counts(W, phased=True) for panel in parts]
    loc, pair = counts[0][0] + counts[1][0], counts[0][1] + counts[1][1]
    for panel in parts:
        assert same(panel.ld_finish(W, loc, pair, phased=True), want)
        panel.close()


@pytest.mark.parametrize("W", [32, 33, 34, 63, 64, 65, 200, 256, 257, 300, 512, 513])
def test_ld_wide_windows_and_kernel_switch(gpu_ctx, W, monkeypatch):
    """the ordered sums: 33 <= W <= 512 one thread per SNP of the window (ld_sum_col_kernel, the combined hr2 rows
    streamed by LDS-DMA; W around the wave size, the accumulator count and the widest it takes), below that and
    when switched off one thread per column (ld_sum_tiled_kernel), wider windows the plain kernel"""
    rng = np.random.default_rng(W)
    nind = 24
    sizes = [W + 70, W, W - 1, 2 * W + 3, 5 * W + 1]
    chroms = [ol.random_panel(rng, n, nind, max_gap=10 ** 9, gaps=0, miss=0.1) for n in sizes]
    want = oracle_ld(chroms, W)
    with make_panel(gpu_ctx, chroms, nind) as panel:
        assert same(panel.compute_ld(W), want)
        monkeypatch.setenv("GARLIC_LD_SUM_BY_COLUMN", "1")
        assert same(panel.compute_ld(W), want)
        monkeypatch.setenv("GARLIC_LD_SUM_L2", "1")
        assert same(panel.compute_ld(W), want)


def test_phased_ld_golden_from_the_reference_build(gpu_ctx):
    """tests/golden/phased_ld.npz holds calcR2LD's own output (tools/make_golden.py, real reference build)"""
    import os
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "phased_ld.npz"))
    n, nind = d["geno"].shape
    with abi.Panel(gpu_ctx, [n], nind) as panel:
        panel.set_map(np.arange(1, n + 1, dtype=np.int32) * 1000, [0], [0])
        panel.set_freq(d["freq"])
        panel.set_genotypes(d["geno"])
        panel.set_phase(d["first_copy"])
        for W in (10, 30):
            assert same(panel.compute_ld(W, phased=True), d[f"ld_W{W}"]), W
            assert same(panel.compute_ld(W, sub_idx=d["sub"], phased=True), d[f"ldsub_W{W}"]), W


def test_ld_plane_cache_follows_genotypes_and_subsample(gpu_ctx):
    """the bit planes are kept across calls: new genotypes, another subsample or another window size must not see stale
    ones; W = 40 and 100 run the matrix-core pair counts (ld_pair_mfma_kernel), chromosomes around its 128-SNP tiles"""
    rng = np.random.default_rng(2026)
    nind = 200
    sizes = [127, 128, 129, 300, 41]
    chroms = [ol.random_panel(rng, n, nind, max_gap=10 ** 9, gaps=0, miss=0.04) for n in sizes]
    sub = np.sort(rng.choice(nind, size=90, replace=False)).astype(np.int32)
    with make_panel(gpu_ctx, chroms, nind) as panel:
        for W in (40, 100, 40):
            assert same(panel.compute_ld(W), oracle_ld(chroms, W)), W
            assert same(panel.compute_ld(W, sub_idx=sub), oracle_ld(chroms, W, sub)), W
        chroms2 = [ol.random_panel(rng, n, nind, max_gap=10 ** 9, gaps=0, miss=0.04) for n in sizes]
        chroms2 = [(c2[0],) + tuple(c[1:]) for c, c2 in zip(chroms, chroms2)]       # same map, new genotypes
        panel.set_genotypes(np.concatenate([c[0] for c in chroms2], axis=0))
        assert same(panel.compute_ld(40, sub_idx=sub), oracle_ld(chroms2, 40, sub))
        assert same(panel.compute_ld(40), oracle_ld(chroms2, 40))
