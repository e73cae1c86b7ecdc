"""CPU: the oracle (oracle/lod_oracle.c) against golden vectors produced by the real reference
(tools/make_golden.py).  Bit-exact everywhere; where libm is involved (log10/exp/pow) the
fixtures were made with the same image's glibc, and per-term tables are stored so the chain
arithmetic is also checked independently of libm."""
import os

import numpy as np
import pytest

import oracle_lib as ol

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name))


def test_lod_known_answers():
    d = load("lod_known_answers.npz")
    o = ol.oracle()
    for i, g in enumerate(d["genotype"]):
        for j, f in enumerate(d["freq"]):
            for k, e in enumerate(d["error"]):
                got = np.float64(o.oracle_lod(int(g), float(f), float(e)))
                assert got.view(np.uint64) == d["lod"][i, j, k].view(np.uint64), (g, f, e)
    # model table of the manual (garlic-manual.tex:131-143): monomorphic or unknown genotype -> +0.0
    assert o.oracle_lod(0, 0.0, 0.001) == 0.0 and o.oracle_lod(-9, 0.3, 0.001) == 0.0
    assert not np.signbit(o.oracle_lod(7, 0.3, 0.001))


def test_tgls_conversion():
    d = load("tgls_conversion.npz")
    o = ol.oracle()
    for code, t in enumerate(("GQ", "GL", "PL")):
        got = np.array([o.oracle_tgls_to_error(float(x), code) for x in d[t + "_in"]])
        assert ol.bits_equal(got, d[t + "_out"]), t
    # README:31 worked example: p=0.999 <-> GQ=30
    assert abs(o.oracle_tgls_to_error(30.0, 0) - 1e-3) < 1e-18
    assert o.oracle_tgls_to_error(0.0, 1) == 1e-16       # GL=0 -> 1-1=0 -> clamped up
    assert o.oracle_tgls_to_error(-5.0, 0) == 1.0        # GQ<0 -> >1 -> clamped down


def test_unweighted_all_cases():
    d = load("unweighted.npz")
    err, mg = float(d["error"]), int(d["max_gap"])
    for c in range(int(d["nchr"])):
        cs, ce, _known = (int(x) for x in d[f"centro{c}"])
        for W in d["winsizes"]:
            want = d[f"win{c}_W{W}"]
            got = ol.oracle_calc_lod(d[f"geno{c}"], d[f"freq{c}"], d[f"pos{c}"], cs, ce, int(W), err, mg)
            assert ol.bits_equal(got, want), (c, W)
            # the mask is shared by all individuals and matches oracle_mask
            valid = ol.oracle_mask(d[f"pos{c}"], cs, ce, int(W), mg).astype(bool)
            assert ((want != ol.MISSING) == valid[None, :]).all()
            # multi-threaded variant is the same computation split over individuals
            assert ol.bits_equal(ol.oracle_calc_lod(d[f"geno{c}"], d[f"freq{c}"], d[f"pos{c}"], cs, ce,
                                                    int(W), err, mg, threads=3), want)


def test_unweighted_terms_and_chain_independent_of_libm():
    """Replays the rolling sum from the stored per-SNP term table with numpy float64 only."""
    d = load("unweighted.npz")
    c, W = 0, 30
    g = d[f"geno{c}"]
    code = np.where((g >= 0) & (g <= 2), g, 3)
    terms = np.take_along_axis(d[f"terms{c}"], code, axis=1)  # [locus][ind]
    want = d[f"win{c}_W{W}"]
    valid = want[0] != ol.MISSING
    got = np.full_like(want, ol.MISSING)
    s = 0
    n = g.shape[0]
    while s < n:
        if not valid[s]:
            s += 1
            continue
        acc = np.zeros(g.shape[1])
        for l in range(s, s + W):
            acc = acc + terms[l]
        got[:, s] = acc
        s += 1
        while s < n and valid[s]:
            acc = (acc - terms[s - 1]) + terms[s + W - 1]
            got[:, s] = acc
            s += 1
    assert ol.bits_equal(got, want)


def test_tgls_lod():
    d = load("tgls_lod.npz")
    cs, ce = (int(x) for x in d["centro"])
    gq_err = np.array([[ol.oracle().oracle_tgls_to_error(float(x), 0) for x in row] for row in d["gq"]])
    assert ol.bits_equal(gq_err, d["gl_error"])
    for W in (10, 60):
        got = ol.oracle_calc_lod(d["geno"], d["freq"], d["pos"], cs, ce, W, 0.001, int(d["max_gap"]),
                                 gl=d["gl_error"])
        assert ol.bits_equal(got, d[f"win_W{W}"]), W


def test_wlod_and_ld():
    d = load("wlod.npz")
    cs, ce = (int(x) for x in d["centro"])
    for W in (10, 30):
        assert ol.bits_equal(ol.oracle_geno_freq(d["geno"]), d[f"hom_W{W}"])
        assert ol.bits_equal(ol.oracle_hr2_ld(d["geno"], W), d[f"ld_W{W}"])
        for threads in (1, 4):
            got = ol.oracle_calc_wlod(d["geno"], d["freq"], d["pos"], d["gpos"], d[f"ldsafe_W{W}"], cs, ce,
                                      W, float(d["error"]), int(d["max_gap"]), float(d["mu"]), int(d["M"]),
                                      threads=threads)
            assert ol.bits_equal(got, d[f"win_W{W}"]), (W, threads)


def test_phased_ld():
    """--phased LD weights (calcR2LD / r2): oracle == what the reference build produced"""
    d = load("phased_ld.npz")
    for W in (10, 30):
        for key, idx in ((f"ld_W{W}", None), (f"ldsub_W{W}", d["sub"])):
            mine = ol.oracle_r2_ld(d["geno"], d["first_copy"], d["freq"], W, idx=idx)
            nan = np.isnan(d[key])
            assert np.array_equal(nan, np.isnan(mine)) and ol.bits_equal(d[key][~nan], mine[~nan]), key


def test_flatten():
    d = load("flatten.npz")
    for step in (1, 30):
        assert ol.bits_equal(ol.oracle_flatten(d["win"], step), d[f"flat_step{step}"])


@pytest.mark.parametrize("W", [2, 17])
def test_edge_shapes(W):
    """nloci < W, nloci == W, a single SNP, everything inside the centromere."""
    rng = np.random.default_rng(9)
    for n in (1, W - 1, W, W + 1):
        if n < 1:
            continue
        g, f, p, cs, ce = ol.random_panel(rng, n, 3, centro=False, gaps=0)
        win = ol.oracle_calc_lod(g, f, p, 0, 0, W, 0.001, 200000)
        assert ((win != ol.MISSING).sum(axis=1) == max(0, n - W + 1)).all()
        inside = ol.oracle_calc_lod(g, f, p, int(p[0]) - 1, int(p[-1]) + 1, W, 0.001, 200000)
        assert (inside == ol.MISSING).all()


def test_roh_coverage_against_brute_force():
    """oracle_roh_coverage (garlic-roh.cpp:446-454) vs the loop written out in numpy; this helper is the
    one part of the oracle that is not pinned to the reference build (no separate function there)"""
    rng = np.random.default_rng(0)
    win = rng.normal(size=(5, 300))
    win[rng.random(win.shape) < 0.1] = -9999.0
    win[0, 5] = np.nan
    for W in (2, 7, 50):
        for cut in (0.0, -10000.0, 1.0):
            got = ol.oracle_roh_coverage(win, W, cut)
            q = win >= cut
            want = np.zeros_like(got)
            for w in range(300):
                for i in range(W):
                    if w + i < 300:
                        want[:, w + i] += q[:, w]
            assert np.array_equal(got, want)

