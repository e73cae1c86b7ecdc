"""GPU: TGLS with continuous likelihoods (--gl-type GL / PL: more distinct values than the 256-entry
dictionary holds; reference src/garlic-data.cpp:1555-1577 -> src/garlic-roh.cpp:68,91-95,117,245).
lod() then runs on the device with glibc's log10 restated (garlic_amd/csrc/tgls_math.hpp); everything
here is compared bit for bit with the oracle, whose log10 is the host libm's."""
import os
from contextlib import contextmanager

import numpy as np
import pytest

import oracle_lib as ol
from garlic_amd import abi

pytestmark = pytest.mark.gpu
MG = 200000


@contextmanager
def env(**kv):
    old = {k: os.environ.get(k) for k in kv}
    os.environ.update({k: str(v) for k, v in kv.items()})
    try:
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def to_error(x, gl_type):
    """readTGLSData's conversion (garlic-data.cpp:1557-1576) through the oracle's restatement"""
    f = ol.oracle().oracle_tgls_to_error
    return np.array([f(float(v), gl_type) for v in np.ravel(x)], dtype=np.float64).reshape(np.shape(x))


def raw_likelihoods(rng, shape, gl_type):
    """what a TGLS file of that type holds, README:31's examples and the clamp cases included"""
    if gl_type == 0:      # GQ: phred-scaled, here with decimals (a file may carry any number)
        x = rng.uniform(0.0, 99.0, size=shape)
        special = [30.0, 0.0, 150.0, -3.0]
    elif gl_type == 1:    # GL: log10 likelihood of the called genotype
        x = -rng.exponential(0.05, size=shape)
        special = [-0.000434511774018, 0.0, 0.2, -12.0]
    else:                 # PL
        x = rng.exponential(3.0, size=shape)
        special = [0.00434511774018, 0.0, -1.0, 250.0]
    flat = x.reshape(-1)
    flat[rng.choice(flat.shape[0], size=4 * len(special), replace=False)] = np.repeat(special, 4)
    return x


def make_panel(ctx, chroms, nind, gpos=None):
    panel = abi.Panel(ctx, [c[0].shape[0] for c in chroms], nind)
    panel.set_map(np.concatenate([c[2] for c in chroms]), [c[3] for c in chroms], [c[4] for c in chroms],
                  gpos=None if gpos is None else np.concatenate(gpos))
    panel.set_freq(np.concatenate([c[1] for c in chroms]))
    panel.set_genotypes(np.concatenate([c[0] for c in chroms], axis=0))
    return panel


def check_tgls(panel, chroms, err, W, error=0.001, **kw):
    out = panel.lod_windows(W, error, MG, use_gl=True, **kw)
    i0 = kw.get("ind_begin", 0)
    for c, (g, f, p, cs, ce) in enumerate(chroms):
        want = ol.oracle_calc_lod(g, f, p, cs, ce, W, error, MG, gl=err[c])
        want = want[i0:i0 + kw.get("ind_count", want.shape[0] - i0)]
        assert ol.bits_equal(np.ascontiguousarray(out[c]), want), ("tgls", W, kw, c)


def check_wlod(panel, chroms, err, gpos, lds, W, M=7, mu=1e-9, **kw):
    out = panel.wlod_windows(W, 0.001, MG, M, mu, use_gl=True, **kw)
    for c, (g, f, p, cs, ce) in enumerate(chroms):
        want = ol.oracle_calc_wlod(g, f, p, gpos[c], lds[c], cs, ce, W, 0.001, MG, mu, M, gl=err[c])
        assert ol.bits_equal(np.ascontiguousarray(out[c]), want), ("wlod", W, kw, c)


@pytest.mark.parametrize("gl_type", [0, 1, 2])
def test_continuous_likelihoods_tgls_and_weighted(gpu_ctx, gl_type):
    """the test that used to pin the refusal: thousands of distinct values per panel, every --gl-type"""
    rng = np.random.default_rng(40 + gl_type)
    nind, sizes = 150, [1500, 37, 640]
    chroms = [ol.random_panel(rng, n, nind, max_gap=MG, gaps=2 if n > 500 else 0) for n in sizes]
    err = [to_error(raw_likelihoods(rng, c[0].shape, gl_type), gl_type) for c in chroms]
    assert np.unique(np.concatenate([e.ravel() for e in err])).shape[0] > 10000
    gpos = [np.cumsum(np.diff(c[2], prepend=0) * 1e-6 * rng.uniform(0.8, 1.2, size=c[2].shape[0])) for c in chroms]
    with make_panel(gpu_ctx, chroms, nind, gpos) as panel:
        allerr = np.concatenate(err, axis=0)
        for l0 in range(0, allerr.shape[0], 500):      # uploaded in slabs
            panel.set_gl(allerr[l0:l0 + 500], locus_begin=l0)
        assert panel.tgls_mode()[0] == 2
        for W in (10, 60):
            for pa in (1, 32):
                check_tgls(panel, chroms, err, W, pitch_align=pa)
        assert panel.tgls_mode() == (2, 1), "the terms must come from the device's log10 on this host"
        check_tgls(panel, chroms, err, 33, ind_begin=37, ind_count=70, pitch_align=32)   # unaligned sub-range
        for W in (10, 40):          # generic kernel (W < 16) and the tile kernel, scores from the scaled terms
            lds = [rng.uniform(1.0, max(2.0, W / 4.0), size=(n, W)) for n in sizes]
            panel.set_ld(W, np.concatenate(lds, axis=0))
            check_wlod(panel, chroms, err, gpos, lds, W, pitch_align=32)
            check_wlod(panel, chroms, err, gpos, lds, W, M=3, mu=2e-9, pitch_align=1)
            check_tgls(panel, chroms, err, W, pitch_align=32)      # back to the raw terms
        # the --error scores of the same panel do not see the likelihoods
        got = panel.lod_windows(60, 0.001, MG)
        for c, (g, f, p, cs, ce) in enumerate(chroms):
            assert ol.bits_equal(np.ascontiguousarray(got[c]), ol.oracle_calc_lod(g, f, p, cs, ce, 60, 0.001, MG))


def test_dictionary_overflows_into_continuous_mid_upload(gpu_ctx):
    """the first slabs hold GQ integers (dictionary codes), a later one continuous values: what was coded
    is decoded, the rest is stored as values; also through the one-byte-code upload"""
    rng = np.random.default_rng(51)
    nind, sizes = 70, [900, 300]
    chroms = [ol.random_panel(rng, n, nind, max_gap=MG) for n in sizes]
    allerr = np.empty((1200, nind))
    allerr[:600] = to_error(rng.integers(3, 61, size=(600, nind)).astype(np.float64), 0)
    allerr[600:] = to_error(-rng.exponential(0.05, size=(600, nind)), 1)
    err = [allerr[:900], allerr[900:]]
    with make_panel(gpu_ctx, chroms, nind) as panel:
        panel.set_gl(allerr[:300])
        assert panel.tgls_mode()[0] == 1
        panel.set_gl(allerr[300:700], locus_begin=300)      # overflows inside this slab
        assert panel.tgls_mode()[0] == 2
        panel.set_gl(allerr[700:], locus_begin=700)
        check_tgls(panel, chroms, err, 25, pitch_align=32)
    tables = [np.unique(allerr[:600]), rng.uniform(0.001, 0.9, size=256), rng.uniform(0.001, 0.9, size=200)]
    codes = [np.searchsorted(tables[0], allerr[:600]).astype(np.uint8),
             rng.integers(0, 256, size=(300, nind)).astype(np.uint8), rng.integers(0, 200, size=(300, nind)).astype(np.uint8)]
    allerr = np.concatenate([tables[k][codes[k]] for k in range(3)], axis=0)
    err = [allerr[:900], allerr[900:]]
    with make_panel(gpu_ctx, chroms, nind) as panel:
        panel.set_gl_codes(codes[0], tables[0])
        assert panel.tgls_mode()[0] == 1
        panel.set_gl_codes(codes[1], tables[1], locus_begin=600)    # 58 + 256 values: no longer a dictionary
        assert panel.tgls_mode()[0] == 2
        panel.set_gl_codes(codes[2], tables[2], locus_begin=900)
        check_tgls(panel, chroms, err, 25, pitch_align=32)


def test_terms_on_the_host_when_forced(gpu_ctx):
    """the fallback for hosts whose libm the device does not reproduce: same scores, made by the host"""
    rng = np.random.default_rng(52)
    nind, sizes = 130, [700, 129]
    chroms = [ol.random_panel(rng, n, nind, max_gap=MG) for n in sizes]
    err = [to_error(-rng.exponential(0.05, size=c[0].shape), 1) for c in chroms]
    with env(GARLIC_TGLS_HOST_TERMS=1), make_panel(gpu_ctx, chroms, nind) as panel:
        panel.set_gl(np.concatenate(err, axis=0))
        check_tgls(panel, chroms, err, 30, pitch_align=32)
        assert panel.tgls_mode() == (2, 2)


def test_values_converted_in_place_when_memory_is_short(gpu_ctx):
    """GARLIC_TGLS_INPLACE=1 forces what a 10M x 1250 shard does: the value matrix becomes the term matrix.
    Raw -> weighted works (scaled in place); going back needs the likelihoods again, all of them."""
    rng = np.random.default_rng(53)
    nind, sizes, W = 100, [800, 200], 20
    chroms = [ol.random_panel(rng, n, nind, max_gap=MG) for n in sizes]
    err = [to_error(rng.exponential(3.0, size=c[0].shape), 2) for c in chroms]
    allerr = np.concatenate(err, axis=0)
    gpos = [c[2] * 1e-6 for c in chroms]
    lds = [rng.uniform(1.0, 5.0, size=(n, W)) for n in sizes]
    with env(GARLIC_TGLS_INPLACE=1), make_panel(gpu_ctx, chroms, nind, gpos) as panel:
        panel.set_gl(allerr)
        panel.set_ld(W, np.concatenate(lds, axis=0))
        check_tgls(panel, chroms, err, W, pitch_align=32)
        check_tgls(panel, chroms, err, 50, pitch_align=1)            # other window sizes reuse the terms
        check_wlod(panel, chroms, err, gpos, lds, W, pitch_align=32)
        with pytest.raises(abi.GarlicError) as e:                    # the raw terms are gone
            panel.lod_windows(W, 0.001, MG, use_gl=True)
        assert e.value.code == abi.ERR_STATE and "uploaded again" in str(e.value)
        panel.set_gl(allerr[:500])                                    # restart: every locus has to come
        with pytest.raises(abi.GarlicError) as e:
            panel.lod_windows(W, 0.001, MG, use_gl=True)
        assert e.value.code == abi.ERR_STATE and "locus 500" in str(e.value)
        panel.set_gl(allerr[500:], locus_begin=500)
        check_tgls(panel, chroms, err, W, pitch_align=32)


def test_continuous_special_values(gpu_ctx):
    """frequencies outside (0, 1) (reachable through --freq-file), NaNs, errors of 0 / above 1 / infinite:
    the NaNs and infinities of the reference, sign and payload included.  One special per chromosome -- a
    NaN that enters a rolling sum never leaves it (garlic-roh.cpp:98-100), so it would hide the others"""
    rng = np.random.default_rng(54)
    nind, n, W = 64, 120, 8
    specials = [("f", -0.25), ("f", 1.5), ("f", np.nan), ("f", -np.nan), ("f", 1e-200), ("e", 0.0), ("e", 7.0),
                ("e", np.inf), ("e", np.nan), ("e", -np.nan), ("e", 1.0), ("e", -0.5), ("e", -np.inf)]
    chroms, err = [], []
    for kind, v in specials:
        g, f, p, cs, ce = ol.random_panel(rng, n, nind, max_gap=MG, gaps=0, centro=False, mono=0.0)
        e = rng.uniform(1e-6, 0.9, size=(n, nind))
        if kind == "f":
            f[60] = v
        else:
            e[60, ::2] = v
        chroms.append((g, f, p, cs, ce))
        err.append(e)
    with make_panel(gpu_ctx, chroms, nind) as panel:
        panel.set_gl(np.concatenate(err, axis=0))
        assert panel.tgls_mode()[0] == 2
        out = panel.lod_windows(W, 0.001, MG, use_gl=True, pitch_align=32)
        seen_nan = seen_inf = 0
        for c, (g, f, p, cs, ce) in enumerate(chroms):
            want = ol.oracle_calc_lod(g, f, p, cs, ce, W, 0.001, MG, gl=err[c])
            seen_nan += int(np.isnan(want).any())
            seen_inf += int(np.isinf(want).any())
            assert ol.bits_equal(np.ascontiguousarray(out[c]), want), specials[c]
        assert seen_nan >= 6 and seen_inf >= 2


def test_weighted_terms_follow_a_new_map(gpu_ctx):
    """garlic_panel_set_map after a weighted TGLS call: the scaled term matrix held products with the
    old decay factors and must be rebuilt (dictionary and continuous likelihoods alike)"""
    rng = np.random.default_rng(55)
    nind, sizes, W = 70, [600], 20
    chroms = [ol.random_panel(rng, n, nind, max_gap=MG) for n in sizes]
    lds = [rng.uniform(1.0, 5.0, size=(n, W)) for n in sizes]
    for kind in ("dictionary", "continuous"):
        err = [rng.choice([1e-3, 0.01, 0.2], size=c[0].shape) if kind == "dictionary"
               else rng.uniform(1e-4, 0.5, size=c[0].shape) for c in chroms]
        gpos = [c[2] * 1e-6 for c in chroms]
        with make_panel(gpu_ctx, chroms, nind, gpos) as panel:
            panel.set_gl(np.concatenate(err, axis=0))
            panel.set_ld(W, np.concatenate(lds, axis=0))
            check_wlod(panel, chroms, err, gpos, lds, W, pitch_align=32)
            gpos2 = [c[2] * 2.5e-6 for c in chroms]
            panel.set_map(np.concatenate([c[2] for c in chroms]), [c[3] for c in chroms], [c[4] for c in chroms],
                          gpos=np.concatenate(gpos2))
            check_wlod(panel, chroms, err, gpos2, lds, W, pitch_align=32)


def test_a_window_sum_of_exactly_minus_9999(gpu_ctx):
    """garlic-roh.cpp:79 decides "the previous window holds no score" by VALUE: a scored window that sums to
    exactly -9999.0 makes the next one a fresh left-to-right sum instead of (previous - leaving) + entering.
    Constructed here: 40 heterozygous SNPs at frequency 0.5 (lod = log10(error)) with per-genotype errors
    near 1e-250, the last one chosen so that the left-to-right sum of the run's first window is -9999.0 to
    the bit.  The library notices that W * (most negative term) can reach -9999, scans the tuned chain's scored
    windows for the value, finds it and runs its by-value chain."""
    o = ol.oracle()
    W, n, nind = 40, 200, 3
    rng = np.random.default_rng(0)
    e0 = 10.0 ** (-rng.uniform(2, 6, size=n))
    e0[:39] = 10.0 ** (-rng.uniform(249.5, 250.5, size=39))
    lod1 = lambda x: o.oracle_lod(1, 0.5, float(x))
    S = 0.0
    for x in e0[:39]:
        S += lod1(x)
    target = -9999.0 - S
    lo, hi = int(np.float64(1e-300).view(np.uint64)), int(np.float64(1e-200).view(np.uint64))
    while lo < hi:                      # lod is monotone in the error: bisection on the bit pattern
        mid = (lo + hi) // 2
        if lod1(np.uint64(mid).view(np.float64)) < target:
            lo = mid + 1
        else:
            hi = mid
    e0[39] = np.uint64(lo).view(np.float64)
    assert lod1(e0[39]) == target and S + target == -9999.0
    g = rng.integers(0, 3, size=(n, nind)).astype(np.int16)
    g[:, 0] = 1
    f = np.full(n, 0.5)
    pos = (np.arange(n, dtype=np.int64) * 1000 + 1000).astype(np.int32)
    e = 10.0 ** (-rng.uniform(2, 6, size=(n, nind)))
    e[:, 0] = e0
    want = ol.oracle_calc_lod(g, f, pos, 0, 0, W, 0.001, MG, gl=e)
    t = [lod1(x) for x in e0]
    fresh = 0.0
    for i in range(1, W + 1):
        fresh += t[i]
    assert want[0, 0] == -9999.0 and want[0, 1] == fresh != (-9999.0 - t[0]) + t[W]   # the by-value branch is live
    with make_panel(gpu_ctx, [(g, f, pos, 0, 0)], nind) as panel:
        panel.set_gl(e)
        for pa in (1, 32):
            out = panel.lod_windows(W, 0.001, MG, use_gl=True, pitch_align=pa)
            assert ol.bits_equal(np.ascontiguousarray(out[0]), want), pa
            assert panel.chain_kind() == 2          # the tuned chain ran, the scan found the value, the by-value chain ran
        feed, _ = panel.lod_feed(W, 0.001, MG, 7, use_gl=True)
        assert ol.bits_equal(feed, ol.oracle_flatten(want, 7))     # -9999.0 is dropped as MISSING, as in the reference


@pytest.mark.parametrize("switch,kind_expected", [("GARLIC_EXACT_CHAIN_ONLY", 2), ("GARLIC_EXACT_CHAIN", 1)])
def test_exact_chain_forced(gpu_ctx, switch, kind_expected):
    """GARLIC_EXACT_CHAIN_ONLY=1: the to-the-letter kernel on ordinary panels (unweighted, dictionary TGLS,
    continuous TGLS, the feed) equals the oracle like the tuned chains do.  GARLIC_EXACT_CHAIN=1: a sum of
    -9999.0 is taken to be possible -- tuned chain, scan, nothing found (kind 1), the same scores."""
    rng = np.random.default_rng(61)
    nind, sizes, W = 130, [700, 45, 300], 30
    chroms = [ol.random_panel(rng, n, nind, max_gap=MG, gaps=2 if n > 500 else 0) for n in sizes]
    with env(**{switch: 1}):
        for kind in ("dictionary", "continuous"):
            err = [rng.choice([1e-3, 0.01, 0.2], size=c[0].shape) if kind == "dictionary"
                   else rng.uniform(1e-4, 0.5, size=c[0].shape) for c in chroms]
            with make_panel(gpu_ctx, chroms, nind) as panel:
                panel.set_gl(np.concatenate(err, axis=0))
                check_tgls(panel, chroms, err, W, pitch_align=32)
                check_tgls(panel, chroms, err, W, ind_begin=37, ind_count=70, pitch_align=1)
                assert panel.chain_kind() == kind_expected
                got = panel.lod_windows(W, 0.001, MG, pitch_align=32)
                assert panel.chain_kind() == kind_expected
                wants = [ol.oracle_calc_lod(g, f, p, cs, ce, W, 0.001, MG) for g, f, p, cs, ce in chroms]
                for c in range(len(chroms)):
                    assert ol.bits_equal(np.ascontiguousarray(got[c]), wants[c]), (kind, c)
                feed, per_chr = panel.lod_feed(W, 0.001, MG, W)
                assert ol.bits_equal(feed, np.concatenate([ol.oracle_flatten(w, W) for w in wants]))
    with make_panel(gpu_ctx, chroms, nind) as panel:      # no switch: 30 terms cannot reach -9999
        panel.lod_windows(W, 0.001, MG, pitch_align=32)
        assert panel.chain_kind() == 0
