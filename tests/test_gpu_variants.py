"""GPU parity of the TGLS (per-genotype likelihood) and wLOD (gap-weighted) variants, bit for bit
against the reference's golden outputs and the oracle."""
import os

import numpy as np
import pytest

import oracle_lib as ol
from garlic_amd import abi

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_tgls_golden(gpu_ctx):
    d = np.load(os.path.join(G, "tgls_lod.npz"))
    cs, ce = (int(x) for x in d["centro"])
    n, nind = d["geno"].shape
    with abi.Panel(gpu_ctx, [n], nind) as panel:
        panel.set_map(d["pos"], [cs], [ce])
        panel.set_freq(d["freq"])
        panel.set_genotypes(d["geno"])
        panel.set_gl(d["gl_error"])
        for W in (10, 60):
            for pa in (1, 32):
                out = panel.lod_windows(W, 0.001, int(d["max_gap"]), use_gl=True, pitch_align=pa)
                assert ol.bits_equal(np.ascontiguousarray(out[0]), d[f"win_W{W}"]), (W, pa)
        # the --error path of the same panel is unaffected by the GL data
        want = ol.oracle_calc_lod(d["geno"], d["freq"], d["pos"], cs, ce, 10, 0.001, int(d["max_gap"]))
        assert ol.bits_equal(np.ascontiguousarray(panel.lod_windows(10, 0.001, int(d["max_gap"]))[0]), want)


def test_tgls_random_multichr_and_chunked_upload(gpu_ctx):
    rng = np.random.default_rng(77)
    mg = 200000
    chroms = [ol.random_panel(rng, n, 70, max_gap=mg) for n in (600, 333)]
    gq = [rng.integers(3, 61, size=c[0].shape).astype(np.float64) for c in chroms]
    err = [np.vectorize(lambda x: ol.oracle().oracle_tgls_to_error(float(x), 0))(q) for q in gq]
    with abi.Panel(gpu_ctx, [600, 333], 70) as panel:
        panel.set_map(np.concatenate([c[2] for c in chroms]), [c[3] for c in chroms], [c[4] for c in chroms])
        panel.set_freq(np.concatenate([c[1] for c in chroms]))
        panel.set_genotypes(np.concatenate([c[0] for c in chroms], axis=0))
        allerr = np.concatenate(err, axis=0)
        for l0 in range(0, 933, 200):
            panel.set_gl(allerr[l0:l0 + 200], locus_begin=l0)
        out = panel.lod_windows(25, 0.5, mg, use_gl=True, pitch_align=32)
        for c, (g, f, p, cs, ce) in enumerate(chroms):
            want = ol.oracle_calc_lod(g, f, p, cs, ce, 25, 0.5, mg, gl=err[c])
            assert ol.bits_equal(np.ascontiguousarray(out[c]), want), c


def test_wlod_golden(gpu_ctx):
    d = np.load(os.path.join(G, "wlod.npz"))
    cs, ce = (int(x) for x in d["centro"])
    n, nind = d["geno"].shape
    with abi.Panel(gpu_ctx, [n], nind) as panel:
        panel.set_map(d["pos"], [cs], [ce], gpos=d["gpos"])
        panel.set_freq(d["freq"])
        panel.set_genotypes(d["geno"])
        for W in (10, 30):
            panel.set_ld(W, d[f"ldsafe_W{W}"])
            for pa in (1, 32):
                out = panel.wlod_windows(W, float(d["error"]), int(d["max_gap"]), int(d["M"]), float(d["mu"]),
                                         pitch_align=pa)
                assert ol.bits_equal(np.ascontiguousarray(out[0]), d[f"win_W{W}"]), (W, pa)
        with pytest.raises(abi.GarlicError):     # LD weights are for W=30 now
            panel.wlod_windows(10, 0.001, 200000, 7, 1e-9)


def test_wlod_random_multichr_with_gl(gpu_ctx):
    rng = np.random.default_rng(8)
    mg, W = 200000, 40
    chroms = [ol.random_panel(rng, n, 66, max_gap=mg, mono=0.0) for n in (500, 260)]
    gpos = [np.cumsum(np.diff(c[2], prepend=0) * 1e-6 * rng.uniform(0.8, 1.2, size=c[2].shape[0])) for c in chroms]
    lds = []
    for c in chroms:
        ld = ol.oracle_hr2_ld(c[0], W)
        lds.append(np.where(np.isfinite(ld) & (ld > 0), ld, 1.0))
    err = [rng.choice([1e-16, 1e-3, 0.01, 0.5, 1.0], size=c[0].shape) for c in chroms]
    with abi.Panel(gpu_ctx, [500, 260], 66) as panel:
        panel.set_map(np.concatenate([c[2] for c in chroms]), [c[3] for c in chroms], [c[4] for c in chroms],
                      gpos=np.concatenate(gpos))
        panel.set_freq(np.concatenate([c[1] for c in chroms]))
        panel.set_genotypes(np.concatenate([c[0] for c in chroms], axis=0))
        panel.set_gl(np.concatenate(err, axis=0))
        panel.set_ld(W, np.concatenate(lds, axis=0))
        for use_gl in (False, True):
            out = panel.wlod_windows(W, 0.001, mg, 7, 1e-9, use_gl=use_gl, pitch_align=32)
            for c, (g, f, p, cs, ce) in enumerate(chroms):
                want = ol.oracle_calc_wlod(g, f, p, gpos[c], lds[c], cs, ce, W, 0.001, mg, 1e-9, 7,
                                           gl=err[c] if use_gl else None)
                assert ol.bits_equal(np.ascontiguousarray(out[c]), want), (use_gl, c)


def test_wlod_wide_window(gpu_ctx):
    """W = 1500: score rows need > 48 KB of LDS and the reads run 1500 SNPs past chromosome ends"""
    rng = np.random.default_rng(5)
    W, nind, mg = 1500, 70, 10 ** 9
    sizes = [4000, 1600]
    chroms = [ol.random_panel(rng, n, nind, max_gap=mg, gaps=0) for n in sizes]
    gpos = [c[2] * 1e-6 for c in chroms]
    lds = [rng.uniform(1.0, 50.0, size=(n, W)) for n in sizes]
    with abi.Panel(gpu_ctx, sizes, nind) as panel:
        panel.set_map(np.concatenate([c[2] for c in chroms]), [c[3] for c in chroms], [c[4] for c in chroms],
                      gpos=np.concatenate(gpos))
        panel.set_freq(np.concatenate([c[1] for c in chroms]))
        panel.set_genotypes(np.concatenate([c[0] for c in chroms], axis=0))
        panel.set_ld(W, np.concatenate(lds, axis=0))
        out = panel.wlod_windows(W, 0.001, mg, 7, 1e-9, pitch_align=32)
        for c, (g, f, p, cs, ce) in enumerate(chroms):
            want = ol.oracle_calc_wlod(g, f, p, gpos[c], lds[c], cs, ce, W, 0.001, mg, 1e-9, 7)
            assert ol.bits_equal(np.ascontiguousarray(out[c]), want), c
        # wider than 4096 with windows to compute: refused (chromosome 0 has 4000 SNPs: W = 3990 still runs)
        panel.set_ld(4200, np.ones((sum(sizes), 4200)))
        out = panel.wlod_windows(4200, 0.001, mg, 7, 1e-9)     # no chromosome holds a window: all MISSING
        assert all((o == ol.MISSING).all() for o in out)


@pytest.mark.parametrize("W", [2, 3, 7, 10, 15, 16, 17, 31, 100, 250])
def test_wlod_tile_kernel_shapes(gpu_ctx, W):
    """tuned wLOD kernels (W >= 16 the hand-scheduled loops, narrower windows -- GARLIC's default is 10 -- the
    unrolled wlod_group_small of the same tile kernel): chromosomes shorter than / equal to /
    just above the window, gaps and centromeres inside tiles, ragged individual counts, unaligned
    sub-ranges, dense and padded output layouts"""
    rng = np.random.default_rng(900 + W)
    mg = 60000
    sizes = [700, 1, W - 1, W, W + 1, 333, 1029]
    nind = 150
    chroms = [ol.random_panel(rng, n, nind, max_gap=mg, gaps=3 if n > 300 else 0) for n in sizes]
    gpos = [np.cumsum(np.diff(c[2], prepend=0) * 1e-6 * rng.uniform(0.8, 1.2, size=c[2].shape[0])) for c in chroms]
    lds = [rng.uniform(1.0, max(2.0, W / 4.0), size=(n, W)) for n in sizes]
    with abi.Panel(gpu_ctx, sizes, nind) as panel:
        panel.set_map(np.concatenate([c[2] for c in chroms]), [c[3] for c in chroms], [c[4] for c in chroms],
                      gpos=np.concatenate(gpos))
        panel.set_freq(np.concatenate([c[1] for c in chroms]))
        panel.set_genotypes(np.concatenate([c[0] for c in chroms], axis=0))
        panel.set_ld(W, np.concatenate(lds, axis=0))
        want = [ol.oracle_calc_wlod(g, f, p, gpos[c], lds[c], cs, ce, W, 0.001, mg, 1e-9, 7)
                for c, (g, f, p, cs, ce) in enumerate(chroms)]
        for pa, i0, cnt in ((1, 0, nind), (32, 0, nind), (2, 37, 70), (32, 64, 86)):
            out = panel.wlod_windows(W, 0.001, mg, 7, 1e-9, pitch_align=pa, ind_begin=i0, ind_count=cnt)
            for c in range(len(sizes)):
                assert ol.bits_equal(np.ascontiguousarray(out[c]), want[c][i0:i0 + cnt]), (pa, i0, c)
        # the same with per-genotype likelihoods (scores from the TGLS term matrix)
        err = [rng.choice([1e-16, 1e-3, 0.01, 0.2, 1.0], size=c[0].shape) for c in chroms]
        panel.set_gl(np.concatenate(err, axis=0))
        want = [ol.oracle_calc_wlod(g, f, p, gpos[c], lds[c], cs, ce, W, 0.001, mg, 1e-9, 7, gl=err[c])
                for c, (g, f, p, cs, ce) in enumerate(chroms)]
        for pa, i0, cnt in ((32, 0, nind), (1, 37, 70)):
            out = panel.wlod_windows(W, 0.001, mg, 7, 1e-9, pitch_align=pa, ind_begin=i0, ind_count=cnt, use_gl=True)
            for c in range(len(sizes)):
                assert ol.bits_equal(np.ascontiguousarray(out[c]), want[c][i0:i0 + cnt]), ("gl", pa, i0, c)
        # the weighted kernel scaled the term matrix in place: the unweighted TGLS chain gets the raw
        # terms back, other decay parameters rescale from them, and the first form comes back too
        tg = panel.lod_windows(W, 0.001, mg, use_gl=True, pitch_align=32)
        for c, (g, f, p, cs, ce) in enumerate(chroms):
            assert ol.bits_equal(np.ascontiguousarray(tg[c]), ol.oracle_calc_lod(g, f, p, cs, ce, W, 0.001, mg, gl=err[c])), c
        out = panel.wlod_windows(W, 0.001, mg, 3, 2e-9, pitch_align=32, use_gl=True)
        for c, (g, f, p, cs, ce) in enumerate(chroms):
            w2 = ol.oracle_calc_wlod(g, f, p, gpos[c], lds[c], cs, ce, W, 0.001, mg, 2e-9, 3, gl=err[c])
            assert ol.bits_equal(np.ascontiguousarray(out[c]), w2), ("gl, other decay", c)
        out = panel.wlod_windows(W, 0.001, mg, 7, 1e-9, pitch_align=32, use_gl=True)
        for c in range(len(sizes)):
            assert ol.bits_equal(np.ascontiguousarray(out[c]), want[c]), ("gl again", c)


def test_kde_feed_flatten_on_device(gpu_ctx):
    """garlic_lod_flatten = convertWinData2DoubleData (garlic-data.cpp:2026): order chr -> ind -> locus,
    every step-th window, MISSING and NaN dropped -- against the golden and the oracle."""
    import torch
    rng = np.random.default_rng(21)
    mg, W = 200000, 30
    chroms = [ol.random_panel(rng, n, 70, max_gap=mg) for n in (900, 410, 64)]
    # a NaN-producing frequency (only reachable through --freq-file): log10 of a negative ratio
    chroms[1][1][100] = -0.25
    with abi.Panel(gpu_ctx, [c[0].shape[0] for c in chroms], 70) as panel:
        panel.set_map(np.concatenate([c[2] for c in chroms]), [c[3] for c in chroms], [c[4] for c in chroms])
        panel.set_freq(np.concatenate([c[1] for c in chroms]))
        panel.set_genotypes(np.concatenate([c[0] for c in chroms], axis=0))
        base, pitch, total = panel.out_layout(32, 70)
        out = torch.empty(total, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        panel.lod_windows_device(out.data_ptr(), W, 0.001, mg, pitch_align=32)
        wins = [ol.oracle_calc_lod(g, f, p, cs, ce, W, 0.001, mg) for g, f, p, cs, ce in chroms]
        assert np.isnan(wins[1]).any()
        for step in (1, W, 7):
            want = np.concatenate([ol.oracle_flatten(w, step) for w in wins])
            n = panel.flatten_device(out.data_ptr(), step, None, 0)          # count only
            assert n == want.shape[0]
            feed = torch.empty(n, dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()
            assert panel.flatten_device(out.data_ptr(), step, feed.data_ptr(), n) == n
            torch.cuda.synchronize()
            assert ol.bits_equal(feed.cpu().numpy(), want), step
    d = np.load(os.path.join(G, "flatten.npz"))
    assert ol.bits_equal(ol.oracle_flatten(d["win"], 30), d["flat_step30"])


@pytest.mark.parametrize("W", [2, 30, 250])
def test_roh_coverage_counts_on_device(gpu_ctx, W):
    """garlic_roh_coverage = the inWin[] loop of assembleROHWindows (garlic-roh.cpp:446-454) over
    device-resident scores: several cutoffs incl. one below MISSING, NaN scores, segment borders
    (2048-SNP segments), chromosomes shorter than the window"""
    import torch
    rng = np.random.default_rng(70 + W)
    mg, nind = 200000, 37
    sizes = [5000, 1, max(1, W - 1), W + 3, 2048, 2049]
    chroms = [ol.random_panel(rng, n, nind, max_gap=mg, gaps=2 if n > 1000 else 0) for n in sizes]
    with abi.Panel(gpu_ctx, sizes, nind) as panel:
        panel.set_map(np.concatenate([c[2] for c in chroms]), [c[3] for c in chroms], [c[4] for c in chroms])
        panel.set_freq(np.concatenate([c[1] for c in chroms]))
        panel.set_genotypes(np.concatenate([c[0] for c in chroms], axis=0))
        base, pitch, total = panel.out_layout(32, nind)
        dev = torch.empty(total, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        panel.lod_windows_device(dev.data_ptr(), W, 0.001, mg, pitch_align=32)
        host = dev.cpu().numpy().copy()
        # plant a few NaNs (never >= cutoff) inside chromosome 0
        for i, l in ((0, 10), (3, 2047), (3, 2048), (36, 4000)):
            host[base[0] + i * pitch[0] + l] = np.nan
        dev.copy_(torch.from_numpy(host))
        torch.cuda.synchronize()
        rows = [host[base[c]: base[c] + nind * pitch[c]].reshape(nind, pitch[c])[:, :n] for c, n in enumerate(sizes)]
        for cutoff in (0.0, -2.5, 3.0, -10000.0):
            got = panel.roh_coverage(dev.data_ptr(), W, cutoff, pitch_align=32)
            got8 = panel.roh_coverage(dev.data_ptr(), W, cutoff, pitch_align=32, inwin_pitch_align=8)   # eight counts per store
            for c in range(len(sizes)):
                want = ol.oracle_roh_coverage(rows[c], W, cutoff)
                assert np.array_equal(got[c], want), (W, cutoff, c)
                assert np.array_equal(got8[c][:, :sizes[c]], want), (W, cutoff, c, "16-B aligned rows")


@pytest.mark.parametrize("W", [2, 30, 100, 250])
def test_roh_coverage_fused_equals_oracle_scores_then_counts(gpu_ctx, W):
    """garlic_roh_coverage_fused: chain + compare + sliding count in one kernel, no score matrix -- against the oracle's
    scores run through the oracle's inWin[] loop; gaps and centromeres (runs a window apart), chromosomes shorter
    than the window and around the 32-window tiles, ragged individual counts, dense and 16-byte-aligned rows, and
    the cases that fall back to scores + garlic_roh_coverage (cutoff below MISSING)"""
    rng = np.random.default_rng(170 + W)
    mg = 200000
    sizes = [5000, 1, max(1, W - 1), W, W + 3, 31, 32, 33, 2048 + W]
    for nind in (1, 37, 130):
        chroms = [ol.random_panel(rng, n, nind, max_gap=mg, gaps=3 if n > 1000 else 0) for n in sizes]
        with abi.Panel(gpu_ctx, sizes, nind) as panel:
            panel.set_map(np.concatenate([c[2] for c in chroms]), [c[3] for c in chroms], [c[4] for c in chroms])
            panel.set_freq(np.concatenate([c[1] for c in chroms]))
            panel.set_genotypes(np.concatenate([c[0] for c in chroms], axis=0))
            scores = [ol.oracle_calc_lod(g, f, p, cs, ce, W, 0.001, mg) for (g, f, p, cs, ce) in chroms]
            for cutoff, align in ((0.0, 1), (-2.5, 8), (3.0, 32), (-10000.0, 8)):
                got = panel.roh_coverage_fused(W, 0.001, mg, cutoff, pitch_align=align)
                for c, n in enumerate(sizes):
                    want = ol.oracle_roh_coverage(np.ascontiguousarray(scores[c]), W, cutoff)
                    assert np.array_equal(got[c][:, :n], want), (W, nind, cutoff, align, c)


def test_roh_coverage_fused_counts_in_the_chain_kernels_queue(gpu_ctx, monkeypatch):
    """GARLIC_COVERAGE_OVERLAP=1 (off by default: measured slower): the count items wait in the
    chain kernel's queue for their chromosome's chains -- same counts as the two launches"""
    rng = np.random.default_rng(99)
    mg, W = 200000, 40
    sizes = [40000, 1, W - 1, W + 3, 9000, 33, 20000]
    for nind in (37, 300):
        chroms = [ol.random_panel(rng, n, nind, max_gap=mg, gaps=3 if n > 1000 else 0) for n in sizes]
        with abi.Panel(gpu_ctx, sizes, nind) as panel:
            panel.set_map(np.concatenate([c[2] for c in chroms]), [c[3] for c in chroms], [c[4] for c in chroms])
            panel.set_freq(np.concatenate([c[1] for c in chroms]))
            panel.set_genotypes(np.concatenate([c[0] for c in chroms], axis=0))
            for cutoff, align in ((0.0, 1), (-2.5, 8)):
                monkeypatch.delenv("GARLIC_COVERAGE_OVERLAP", raising=False)
                want = panel.roh_coverage_fused(W, 0.001, mg, cutoff, pitch_align=align)
                monkeypatch.setenv("GARLIC_COVERAGE_OVERLAP", "1")
                for _ in range(2):
                    got = panel.roh_coverage_fused(W, 0.001, mg, cutoff, pitch_align=align)
                    for c, n in enumerate(sizes):
                        assert np.array_equal(got[c][:, :n], want[c][:, :n]), (nind, cutoff, align, c)
            monkeypatch.delenv("GARLIC_COVERAGE_OVERLAP", raising=False)


@pytest.mark.parametrize("W", [5, 16, 40, 100])
def test_roh_coverage_fused_weighted(gpu_ctx, W):
    """garlic_roh_coverage_fused with --weighted (plain and with per-genotype likelihoods): the tuned wLOD kernels leave
    16 bits per individual and group instead of 16 scores -- against the oracle's wLOD scores through the oracle's
    inWin[] loop; also unweighted scores with likelihoods (the two-step path)"""
    rng = np.random.default_rng(270 + W)
    mg = 200000
    sizes = [3000, max(1, W - 1), W + 3, 33, 700]
    for nind in (37, 130):
        chroms = [ol.random_panel(rng, n, nind, max_gap=mg, gaps=2 if n > 1000 else 0) for n in sizes]
        gpos = [c[2] * 1e-6 for c in chroms]
        lds = [rng.uniform(1.0, max(2.0, W / 4.0), size=(n, W)) for n in sizes]
        gl = [rng.choice([1e-16, 1e-3, 0.01, 0.2, 1.0], size=c[0].shape) for c in chroms]
        with abi.Panel(gpu_ctx, sizes, nind) as panel:
            panel.set_map(np.concatenate([c[2] for c in chroms]), [c[3] for c in chroms], [c[4] for c in chroms],
                          gpos=np.concatenate(gpos))
            panel.set_freq(np.concatenate([c[1] for c in chroms]))
            panel.set_genotypes(np.concatenate([c[0] for c in chroms], axis=0))
            panel.set_ld(W, np.concatenate(lds, axis=0))
            panel.set_gl(np.concatenate(gl, axis=0))
            for use_gl in (False, True):
                want = [ol.oracle_calc_wlod(g, f, p, gpos[c], lds[c], cs, ce, W, 0.001, mg, 1e-9, 7, gl=gl[c] if use_gl else None)
                        for c, (g, f, p, cs, ce) in enumerate(chroms)]
                for cutoff, align in ((0.0, 1), (-1.5, 8), (-10000.0, 8)):
                    got = panel.roh_coverage_fused(W, 0.001, mg, cutoff, pitch_align=align, use_gl=use_gl, weighted=True)
                    for c, n in enumerate(sizes):
                        assert np.array_equal(got[c][:, :n], ol.oracle_roh_coverage(np.ascontiguousarray(want[c]), W, cutoff)), \
                            (W, nind, use_gl, cutoff, align, c)
            want = [ol.oracle_calc_lod(g, f, p, cs, ce, W, 0.001, mg, gl=gl[c]) for c, (g, f, p, cs, ce) in enumerate(chroms)]
            got = panel.roh_coverage_fused(W, 0.001, mg, 0.5, pitch_align=8, use_gl=True)
            for c, n in enumerate(sizes):
                assert np.array_equal(got[c][:, :n], ol.oracle_roh_coverage(np.ascontiguousarray(want[c]), W, 0.5)), (W, nind, c)


def test_lod_feed_one_call(gpu_ctx):
    """garlic_lod_feed = scores + convertWinData2DoubleData on the device, unweighted / TGLS / wLOD"""
    rng = np.random.default_rng(21)
    mg, W, nind = 200000, 30, 70
    sizes = [900, 400]
    chroms = [ol.random_panel(rng, n, nind, max_gap=mg) for n in sizes]
    gpos = [c[2] * 1e-6 for c in chroms]
    lds = [rng.uniform(1.0, 8.0, size=(n, W)) for n in sizes]
    err = [rng.choice([1e-3, 0.01, 0.2], size=c[0].shape) for c in chroms]
    with abi.Panel(gpu_ctx, sizes, nind) as panel:
        panel.set_map(np.concatenate([c[2] for c in chroms]), [c[3] for c in chroms], [c[4] for c in chroms],
                      gpos=np.concatenate(gpos))
        panel.set_freq(np.concatenate([c[1] for c in chroms]))
        panel.set_genotypes(np.concatenate([c[0] for c in chroms], axis=0))
        panel.set_gl(np.concatenate(err, axis=0))
        panel.set_ld(W, np.concatenate(lds, axis=0))
        for step in (1, W, 7):
            for mode in ("lod", "tgls", "wlod"):
                feed, per_chr = panel.lod_feed(W, 0.001, mg, step, use_gl=(mode == "tgls"), weighted=(mode == "wlod"))
                want = []
                for c, (g, f, p, cs, ce) in enumerate(chroms):
                    if mode == "wlod":
                        win = ol.oracle_calc_wlod(g, f, p, gpos[c], lds[c], cs, ce, W, 0.001, mg, 1e-9, 7)
                    else:
                        win = ol.oracle_calc_lod(g, f, p, cs, ce, W, 0.001, mg, gl=err[c] if mode == "tgls" else None)
                    want.append(ol.oracle_flatten(win, step))
                assert [len(w) for w in want] == list(per_chr), (mode, step)
                assert ol.bits_equal(feed, np.concatenate(want)), (mode, step)



def test_lod_feed_of_a_subsample(gpu_ctx):
    """garlic_lod_feed_subset = convertSubsetWinData2DoubleData (garlic-data.cpp:2071-2150, --kde-subsample)
    with the drawn individuals supplied: chromosome -> listed individual -> locus, for the thinned chain
    write-out (only the blocks that hold a listed individual are scored), the full-score paths (TGLS, wLOD,
    small steps), ascending draws as gsl_ran_choose leaves them and an arbitrary order"""
    rng = np.random.default_rng(23)
    mg, W, nind = 200000, 30, 330
    sizes = [2500, 400, 64]
    chroms = [ol.random_panel(rng, n, nind, max_gap=mg, gaps=2 if n > 1000 else 0) for n in sizes]
    gpos = [c[2] * 1e-6 for c in chroms]
    lds = [rng.uniform(1.0, 8.0, size=(n, W)) for n in sizes]
    err = [rng.choice([1e-3, 0.01, 0.2], size=c[0].shape) for c in chroms]
    draws = [np.sort(rng.choice(nind, size=20, replace=False)), np.array([329, 0, 64, 63, 200]), np.array([128]),
             np.arange(nind)]
    with abi.Panel(gpu_ctx, sizes, nind) as panel:
        panel.set_map(np.concatenate([c[2] for c in chroms]), [c[3] for c in chroms], [c[4] for c in chroms],
                      gpos=np.concatenate(gpos))
        panel.set_freq(np.concatenate([c[1] for c in chroms]))
        panel.set_genotypes(np.concatenate([c[0] for c in chroms], axis=0))
        panel.set_gl(np.concatenate(err, axis=0))
        panel.set_ld(W, np.concatenate(lds, axis=0))
        wins = {}
        for mode in ("lod", "tgls", "wlod"):
            wins[mode] = []
            for c, (g, f, p, cs, ce) in enumerate(chroms):
                if mode == "wlod":
                    wins[mode].append(ol.oracle_calc_wlod(g, f, p, gpos[c], lds[c], cs, ce, W, 0.001, mg, 1e-9, 7))
                else:
                    wins[mode].append(ol.oracle_calc_lod(g, f, p, cs, ce, W, 0.001, mg, gl=err[c] if mode == "tgls" else None))
        for idx in draws:
            for mode, steps in (("lod", (W, 7, 1)), ("tgls", (W,)), ("wlod", (W,))):
                for step in steps:
                    feed, per_chr = panel.lod_feed(W, 0.001, mg, step, use_gl=(mode == "tgls"), weighted=(mode == "wlod"),
                                                   ind_idx=idx)
                    want = [ol.oracle_flatten_subset(w, step, idx) for w in wins[mode]]
                    assert [len(w) for w in want] == list(per_chr), (mode, step, idx[:3])
                    assert ol.bits_equal(feed, np.concatenate(want)), (mode, step, idx[:3])
            # the whole-panel feed afterwards (other block set: a new plan)
            feed, _ = panel.lod_feed(W, 0.001, mg, W)
            assert ol.bits_equal(feed, np.concatenate([ol.oracle_flatten(w, W) for w in wins["lod"]]))
        for bad in ([0, 0], [nind], [-1]):
            with pytest.raises(abi.GarlicError):
                panel.lod_feed(W, 0.001, mg, W, ind_idx=np.array(bad))


@pytest.mark.parametrize("W,step", [(30, 30), (100, 100), (100, 7), (50, 33), (64, 64), (20, 4), (100, 1000)])
def test_lod_feed_thinned_write_out(gpu_ctx, W, step):
    """unweighted feed with step >= 4: the chain kernel itself stores only the sampled windows (POST's
    thinned role in the hand-scheduled loop, the masked head/tail tiles elsewhere).  Runs long enough
    for the loop, several chromosomes, gaps, a ragged last block -- equal to flatten(oracle scores)."""
    rng = np.random.default_rng(1000 * W + step)
    mg, nind = 200000, 150
    sizes = [5000, 37, 2111, W, W + 1]
    chroms = [ol.random_panel(rng, n, nind, max_gap=mg, gaps=3 if n > 1000 else 0) for n in sizes]
    with abi.Panel(gpu_ctx, sizes, nind) as panel:
        panel.set_map(np.concatenate([c[2] for c in chroms]), [c[3] for c in chroms], [c[4] for c in chroms])
        panel.set_freq(np.concatenate([c[1] for c in chroms]))
        panel.set_genotypes(np.concatenate([c[0] for c in chroms], axis=0))
        for _ in range(2):                                   # second call: resident plan
            feed, per_chr = panel.lod_feed(W, 0.001, mg, step)
            want = [ol.oracle_flatten(ol.oracle_calc_lod(g, f, p, cs, ce, W, 0.001, mg), step)
                    for g, f, p, cs, ce in chroms]
            assert [len(w) for w in want] == list(per_chr)
            assert ol.bits_equal(feed, np.concatenate(want))
        # the full scores afterwards are not disturbed by the thinned plan
        got = panel.lod_windows(W, 0.001, mg)
        for c, (g, f, p, cs, ce) in enumerate(chroms):
            assert ol.bits_equal(got[c], ol.oracle_calc_lod(g, f, p, cs, ce, W, 0.001, mg)), c


def test_lod_feed_multi_equals_single_calls_and_the_oracle(gpu_ctx, monkeypatch):
    """garlic_lod_feed_multi (--winsize-multi: exploreWinsizes / selectWinsizeFromList, garlic-roh.cpp:726-751,
    881-920): every size on its own stream, feeds fetched in order -- the values of one garlic_lod_feed call per
    size and of flatten(oracle scores); with an individual list; with a step the chain kernel does not thin for
    (falls back to single calls); repeated (resident scratch); the serial path forced"""
    rng = np.random.default_rng(77)
    mg, nind = 200000, 200
    sizes = [6000, 45, 1500, 301]
    chroms = [ol.random_panel(rng, n, nind, max_gap=mg, gaps=3 if n > 1000 else 0) for n in sizes]
    Ws = [50, 100, 200, 300, 20]
    with abi.Panel(gpu_ctx, sizes, nind) as panel:
        panel.set_map(np.concatenate([c[2] for c in chroms]), [c[3] for c in chroms], [c[4] for c in chroms])
        panel.set_freq(np.concatenate([c[1] for c in chroms]))
        panel.set_genotypes(np.concatenate([c[0] for c in chroms], axis=0))
        wins = {W: [ol.oracle_calc_lod(g, f, p, cs, ce, W, 0.001, mg) for g, f, p, cs, ce in chroms] for W in Ws}
        for rep in range(2):
            feeds, per_chr = panel.lod_feed_multi(Ws, 0.001, mg)
            for i, W in enumerate(Ws):
                want = [ol.oracle_flatten(w, W) for w in wins[W]]
                assert [len(w) for w in want] == list(per_chr[i]), (W, rep)
                assert ol.bits_equal(feeds[i], np.concatenate(want)), (W, rep)
                single, _ = panel.lod_feed(W, 0.001, mg, W)
                assert ol.bits_equal(feeds[i], single), (W, rep)
        idx = np.array([199, 3, 64, 130])
        feeds, per_chr = panel.lod_feed_multi(Ws[:3], 0.001, mg, steps=[7, 100, 33], ind_idx=idx)
        for i, (W, step) in enumerate(zip(Ws[:3], (7, 100, 33))):
            want = [ol.oracle_flatten_subset(w, step, idx) for w in wins[W]]
            assert [len(w) for w in want] == list(per_chr[i]), (W, step)
            assert ol.bits_equal(feeds[i], np.concatenate(want)), (W, step)
        feeds, _ = panel.lod_feed_multi([100, 50], 0.001, mg, steps=[1, 50])     # step 1: full scores, sampled
        assert ol.bits_equal(feeds[0], np.concatenate([ol.oracle_flatten(w, 1) for w in wins[100]]))
        assert ol.bits_equal(feeds[1], np.concatenate([ol.oracle_flatten(w, 50) for w in wins[50]]))
        monkeypatch.setenv("GARLIC_FEED_SERIAL", "1")
        feeds, _ = panel.lod_feed_multi(Ws, 0.001, mg)
        for i, W in enumerate(Ws):
            assert ol.bits_equal(feeds[i], np.concatenate([ol.oracle_flatten(w, W) for w in wins[W]])), W
        with pytest.raises(abi.GarlicError):
            panel.lod_feed_multi([100, 1], 0.001, mg)


@pytest.mark.parametrize("W,step,nind", [(2, 4, 64), (5, 31, 65), (33, 32, 1), (129, 5, 130), (1100, 64, 70)])
def test_lod_feed_thinned_edge_shapes(gpu_ctx, W, step, nind):
    """thinned write-out at the edges: windows narrower than a tile and than the step, a step of one
    tile, a single individual, a window wider than the hand-scheduled loop's genotype ring (generic
    tile path only)"""
    rng = np.random.default_rng(7 * W + step)
    mg = 200000
    sizes = [3 * W + 700, 1, W + 40]
    chroms = [ol.random_panel(rng, n, nind, max_gap=mg, gaps=2 if n > 500 else 0) for n in sizes]
    with abi.Panel(gpu_ctx, sizes, nind) as panel:
        panel.set_map(np.concatenate([c[2] for c in chroms]), [c[3] for c in chroms], [c[4] for c in chroms])
        panel.set_freq(np.concatenate([c[1] for c in chroms]))
        panel.set_genotypes(np.concatenate([c[0] for c in chroms], axis=0))
        feed, per_chr = panel.lod_feed(W, 0.001, mg, step)
        want = [ol.oracle_flatten(ol.oracle_calc_lod(g, f, p, cs, ce, W, 0.001, mg), step) for g, f, p, cs, ce in chroms]
        assert [len(w) for w in want] == list(per_chr)
        assert ol.bits_equal(feed, np.concatenate(want))


def test_tgls_from_dictionary_codes(gpu_ctx):
    """garlic_panel_set_gl_codes: likelihoods that arrive as one-byte codes + a value table, a
    different table per chunk (as per chromosome in the host adapter) -> the scores of the doubles"""
    rng = np.random.default_rng(404)
    mg, W, nind = 200000, 20, 90
    sizes = [500, 260]
    chroms = [ol.random_panel(rng, n, nind, max_gap=mg) for n in sizes]
    tables = [np.array([1e-16, 1e-3, 0.01, 0.2]), np.array([0.5, 1e-3, 1.0, 0.2, 0.03])]   # overlapping, reordered
    codes = [rng.integers(0, len(t), size=c[0].shape).astype(np.uint8) for t, c in zip(tables, chroms)]
    with abi.Panel(gpu_ctx, sizes, nind) as panel:
        panel.set_map(np.concatenate([c[2] for c in chroms]), [c[3] for c in chroms], [c[4] for c in chroms])
        panel.set_freq(np.concatenate([c[1] for c in chroms]))
        panel.set_genotypes(np.concatenate([c[0] for c in chroms], axis=0))
        panel.set_gl_codes(codes[0], tables[0])
        panel.set_gl_codes(codes[1], tables[1], locus_begin=sizes[0])
        out = panel.lod_windows(W, 0.5, mg, use_gl=True, pitch_align=32)
        for c, (g, f, p, cs, ce) in enumerate(chroms):
            want = ol.oracle_calc_lod(g, f, p, cs, ce, W, 0.5, mg, gl=tables[c][codes[c]])
            assert ol.bits_equal(np.ascontiguousarray(out[c]), want), c
        with pytest.raises(abi.GarlicError):      # a caller's table holds at most 256 values (one-byte codes)
            panel.set_gl_codes(codes[0], np.linspace(0.1, 0.9, 257))


def _oracle_segments(chroms, scores, W, cutoff, mg, frac):
    """the oracle's scores through the oracle's inWin[] loop and its restatement of the segment walk (pinned against the
    real assembleROHWindows in tests/test_oracle_vs_ref.py) -> [(individual, chromosome, first SNP, last SNP)] in the
    reference's order"""
    out = []
    for c, (g, f, p, cs, ce) in enumerate(chroms):
        cov = ol.oracle_roh_coverage(np.ascontiguousarray(scores[c]), W, cutoff)
        out += [(i, c, a, b) for i, a, b in ol.oracle_roh_segments(cov, p, cs, ce, W, mg, frac)]
    return sorted(out)


@pytest.mark.parametrize("W", [2, 30, 100, 250])
def test_roh_segments_equal_the_oracles(gpu_ctx, W):
    """garlic_roh_segments (unweighted --error scores): the segments of assembleROHWindows straight from the genotypes --
    gaps and centromeres, chromosomes shorter than the window and around the 32-SNP words, ragged individual counts,
    thresholds from one SNP to the whole window, a cutoff low enough for long segments and one below MISSING (the
    scores path), capacity too small"""
    rng = np.random.default_rng(470 + W)
    mg = 200000
    sizes = [5000, 1, max(1, W - 1), W, W + 3, 31, 32, 33, 2048 + W]
    for nind in (1, 37, 130):
        chroms = [ol.random_panel(rng, n, nind, max_gap=mg, gaps=3 if n > 1000 else 0) for n in sizes]
        with abi.Panel(gpu_ctx, sizes, nind) as panel:
            panel.set_map(np.concatenate([c[2] for c in chroms]), [c[3] for c in chroms], [c[4] for c in chroms])
            panel.set_freq(np.concatenate([c[1] for c in chroms]))
            panel.set_genotypes(np.concatenate([c[0] for c in chroms], axis=0))
            scores = [ol.oracle_calc_lod(g, f, p, cs, ce, W, 0.001, mg) for (g, f, p, cs, ce) in chroms]
            total = 0
            for cutoff in (0.0, -2.5, 3.0, -10000.0):
                for frac in (1e-9, 0.25, 0.6, 1.0):
                    want = _oracle_segments(chroms, scores, W, cutoff, mg, frac)
                    got = panel.roh_segments(W, 0.001, mg, cutoff, frac)
                    assert [tuple(int(v) for v in r) for r in got] == want, (W, nind, cutoff, frac, len(got), len(want))
                    total += len(want)
            assert total > 0
            if len(want) > 1:
                with pytest.raises(abi.GarlicError):
                    panel.roh_segments(W, 0.001, mg, cutoff, frac, capacity=len(want) - 1)


@pytest.mark.parametrize("W", [5, 40, 100])
def test_roh_segments_weighted_and_with_likelihoods(gpu_ctx, W):
    """garlic_roh_segments with --weighted (plain and with per-genotype likelihoods) and for unweighted scores with
    likelihoods: the bits of the wLOD kernels / the TGLS chain into the same segment kernels"""
    rng = np.random.default_rng(570 + W)
    mg = 200000
    sizes = [3000, max(1, W - 1), W + 3, 33, 700]
    nind = 70
    chroms = [ol.random_panel(rng, n, nind, max_gap=mg, gaps=2 if n > 1000 else 0) for n in sizes]
    gpos = [c[2] * 1e-6 for c in chroms]
    lds = [rng.uniform(1.0, max(2.0, W / 4.0), size=(n, W)) for n in sizes]
    gl = [rng.choice([1e-16, 1e-3, 0.01, 0.2, 1.0], size=c[0].shape) for c in chroms]
    with abi.Panel(gpu_ctx, sizes, nind) as panel:
        panel.set_map(np.concatenate([c[2] for c in chroms]), [c[3] for c in chroms], [c[4] for c in chroms],
                      gpos=np.concatenate(gpos))
        panel.set_freq(np.concatenate([c[1] for c in chroms]))
        panel.set_genotypes(np.concatenate([c[0] for c in chroms], axis=0))
        panel.set_ld(W, np.concatenate(lds, axis=0))
        panel.set_gl(np.concatenate(gl, axis=0))
        for weighted, use_gl in ((True, False), (True, True), (False, True)):
            if weighted:
                scores = [ol.oracle_calc_wlod(g, f, p, gpos[c], lds[c], cs, ce, W, 0.001, mg, 1e-9, 7, gl=gl[c] if use_gl else None)
                          for c, (g, f, p, cs, ce) in enumerate(chroms)]
            else:
                scores = [ol.oracle_calc_lod(g, f, p, cs, ce, W, 0.001, mg, gl=gl[c]) for c, (g, f, p, cs, ce) in enumerate(chroms)]
            for cutoff in (0.0, -1.5):
                for frac in (0.25, 1.0):
                    want = _oracle_segments(chroms, scores, W, cutoff, mg, frac)
                    got = panel.roh_segments(W, 0.001, mg, cutoff, frac, use_gl=use_gl, weighted=weighted)
                    assert [tuple(int(v) for v in r) for r in got] == want, (W, weighted, use_gl, cutoff, frac, len(got), len(want))


def test_roh_segments_and_counts_over_many_small_chromosomes(gpu_ctx):
    """150 chromosomes of 1 .. 90 SNPs (scaffolds): every word of the bit matrices belongs to another chromosome than its
    neighbours -- the fused counts and the segments against the oracle"""
    rng = np.random.default_rng(77)
    mg, W, nind = 200000, 6, 45
    sizes = [int(x) for x in rng.integers(1, 91, size=150)]
    chroms = [ol.random_panel(rng, n, nind, max_gap=mg, gaps=0) for n in sizes]
    with abi.Panel(gpu_ctx, sizes, nind) as panel:
        panel.set_map(np.concatenate([c[2] for c in chroms]), [c[3] for c in chroms], [c[4] for c in chroms])
        panel.set_freq(np.concatenate([c[1] for c in chroms]))
        panel.set_genotypes(np.concatenate([c[0] for c in chroms], axis=0))
        scores = [ol.oracle_calc_lod(g, f, p, cs, ce, W, 0.001, mg) for (g, f, p, cs, ce) in chroms]
        for cutoff in (-1.0, -4.0):
            got = panel.roh_coverage_fused(W, 0.001, mg, cutoff, pitch_align=8)
            for c, n in enumerate(sizes):
                assert np.array_equal(got[c][:, :n], ol.oracle_roh_coverage(np.ascontiguousarray(scores[c]), W, cutoff)), (cutoff, c)
            for frac in (0.2, 1.0):
                want = _oracle_segments(chroms, scores, W, cutoff, mg, frac)
                segs = panel.roh_segments(W, 0.001, mg, cutoff, frac)
                assert [tuple(int(v) for v in r) for r in segs] == want, (cutoff, frac, len(segs), len(want))
        assert len(want) > 50


@pytest.mark.parametrize("W", [4, 30])
def test_roh_segments_of_chromosomes_that_start_at_position_zero(gpu_ctx, W):
    """0-based maps: a stretch opened at SNP 0 is neither open nor closed to the reference's tests on its first position
    (garlic-roh.cpp:456, 493, 514; pinned against the real assembleROHWindows in tests/test_oracle_vs_ref.py): it ends at
    the first covered SNP behind a break and everything in front of that is one segment or none.  Chromosomes that start
    at 0 beside ones that do not, with breaks early, late and not at all; a negative position is refused"""
    rng = np.random.default_rng(640 + W)
    mg, nind = 200000, 70
    sizes = [700, 40, 1, 3000, 33, 900]
    chroms = []
    for k, n in enumerate(sizes):
        g, f, p, cs, ce = ol.random_panel(rng, n, nind, max_gap=mg, gaps=(0, 1, 0, 4, 2, 3)[k], centro=(k % 2 == 0))
        if k != 5:
            shift = int(p[0])
            p = (p - shift).astype(np.int32)
            cs, ce = (cs - shift, ce - shift) if (cs or ce) else (0, 0)
        chroms.append((g, f, p, cs, ce))
    with abi.Panel(gpu_ctx, sizes, nind) as panel:
        panel.set_map(np.concatenate([c[2] for c in chroms]), [c[3] for c in chroms], [c[4] for c in chroms])
        panel.set_freq(np.concatenate([c[1] for c in chroms]))
        panel.set_genotypes(np.concatenate([c[0] for c in chroms], axis=0))
        scores = [ol.oracle_calc_lod(g, f, p, cs, ce, W, 0.001, mg) for (g, f, p, cs, ce) in chroms]
        from_zero = 0
        for cutoff in (-2.0, 0.0, -20000.0):
            for frac in (1e-9, 0.25, 1.0):          # (SNP 0 lies in one window: it is "in ROH" only at a threshold of one SNP)
                want = _oracle_segments(chroms, scores, W, cutoff, mg, frac)
                got = panel.roh_segments(W, 0.001, mg, cutoff, frac)
                assert [tuple(int(v) for v in r) for r in got] == want, (W, cutoff, frac, len(got), len(want))
                from_zero += sum(1 for r in want if r[2] == 0 and r[1] != 5)
        assert from_zero > 20
    g, f, p, cs, ce = chroms[0]
    with abi.Panel(gpu_ctx, [sizes[0]], nind) as panel:
        panel.set_map((p - 1).astype(np.int32), [cs], [ce])
        panel.set_freq(f)
        panel.set_genotypes(g)
        with pytest.raises(abi.GarlicError, match="positions >= 0"):
            panel.roh_segments(20, 0.001, mg, 0.0, 0.25)
