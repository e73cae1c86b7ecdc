#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REAL reference (oracle/_ref/libgarlic_ref.so, built from
/root/reference/src by oracle/Makefile).  Runs in the build container only; the fixtures it
writes are data (inputs + the reference's outputs), never reference source.

    make -C oracle ref && python tools/make_golden.py

Every file stores the inputs next to the outputs so the tests need nothing but numpy.
"""
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
MAX_GAP = 200000
ERROR = 0.001


def lod_known_answers():
    g = np.array([0, 1, 2, -9, 3], dtype=np.int16)
    f = np.array([0.0, 1.0, 1e-6, 0.001, 0.01, 0.05, 0.25, 0.5, 0.75, 0.95, 0.99, 0.999], dtype=np.float64)
    e = np.array([1e-16, 1e-3, 0.01, 0.5, 1.0], dtype=np.float64)
    G, F, E = np.meshgrid(g, f, e, indexing="ij")
    out = np.empty(G.shape, dtype=np.float64)
    r = ol.ref()
    for idx in np.ndindex(G.shape):
        out[idx] = r.ref_lod(int(G[idx]), float(F[idx]), float(E[idx]))
    np.savez_compressed(os.path.join(OUT, "lod_known_answers.npz"), genotype=g, freq=f, error=e, lod=out)
    return out.size


def tgls_conversion():
    # README:31 worked example (p = 0.999): GQ=30, GL=-0.000434511774018, PL=0.00434511774018;
    # plus the clamp cases of garlic-data.cpp:1559,1575-1576
    vals = {
        "GQ": [30, 3, 0, 60, 100, 150, -5, 17, 0.5],
        "GL": [-0.000434511774018, 0.0, -1.0, -10.0, -12.0, 0.3, -0.05],
        "PL": [0.00434511774018, 0.0, 10.0, 100.0, 120.0, -3.0, 0.5],
    }
    res = {}
    for t, v in vals.items():
        v = np.array(v, dtype=np.float64)
        nind = len(v)
        with tempfile.NamedTemporaryFile("w", suffix=".tgls", delete=False) as fh:
            fh.write("chrT snp0 0 1 " + " ".join(repr(float(x)) for x in v) + "\n")
            path = fh.name
        out = np.empty((1, nind), dtype=np.float64)
        rc = ol.ref().ref_readTGLS(path.encode(), 1, nind, t.encode(), ol._p(out, ol._dp))
        os.unlink(path)
        assert rc == 0
        res[t + "_in"] = v
        res[t + "_out"] = out[0]
    np.savez_compressed(os.path.join(OUT, "tgls_conversion.npz"), **res)
    return sum(len(v) for v in vals.values())


def unweighted():
    rng = np.random.default_rng(20260101)
    nind = 12
    chroms = []
    # (nloci, centromere known?, gaps)
    for n, known, gaps in [(1200, True, 3), (700, False, 2), (300, True, 1), (45, True, 0), (60, True, 0)]:
        g, f, p, cs, ce = ol.random_panel(rng, n, nind, max_gap=MAX_GAP, gaps=gaps, mono=0.02)
        if not known:
            cs = ce = 0
        chroms.append((g, f, p, cs, ce, known))
    data = {"nchr": len(chroms), "error": ERROR, "max_gap": MAX_GAP}
    wins = [2, 5, 30, 60, 100, 300]
    n = 0
    for c, (g, f, p, cs, ce, known) in enumerate(chroms):
        data[f"geno{c}"], data[f"freq{c}"], data[f"pos{c}"] = g, f, p
        data[f"centro{c}"] = np.array([cs, ce, int(known)], dtype=np.int32)
        # per-SNP term table as the reference's lod() gives it (chain arithmetic can then be
        # checked independently of libm)
        tab = np.empty((g.shape[0], 4), dtype=np.float64)
        for l in range(g.shape[0]):
            for k, code in enumerate((0, 1, 2, -9)):
                tab[l, k] = ol.ref().ref_lod(code, float(f[l]), ERROR)
        data[f"terms{c}"] = tab
        for W in wins:
            win = ol.ref_calc_lod(g, f, p, cs, ce, W, ERROR, MAX_GAP, centro_known=known)
            data[f"win{c}_W{W}"] = win
            n += win.size
    data["winsizes"] = np.array(wins, dtype=np.int32)
    np.savez_compressed(os.path.join(OUT, "unweighted.npz"), **data)
    return n


def tgls_lod():
    rng = np.random.default_rng(20260102)
    g, f, p, cs, ce = ol.random_panel(rng, 500, 10, max_gap=MAX_GAP, gaps=2)
    gq = rng.integers(3, 61, size=g.shape).astype(np.float64)
    gl = np.power(10.0, np.maximum(gq / -10.0, -10.0))  # inputs only; the reference converts below
    # use the reference's own conversion for the error matrix
    with tempfile.NamedTemporaryFile("w", suffix=".tgls", delete=False) as fh:
        for l in range(g.shape[0]):
            fh.write(f"chrT s{l} 0 {int(p[l])} " + " ".join(str(int(x)) for x in gq[l]) + "\n")
        path = fh.name
    err = np.empty(g.shape, dtype=np.float64)
    rc = ol.ref().ref_readTGLS(path.encode(), g.shape[0], g.shape[1], b"GQ", ol._p(err, ol._dp))
    os.unlink(path)
    assert rc == 0
    del gl
    data = dict(geno=g, freq=f, pos=p, centro=np.array([cs, ce], dtype=np.int32), gq=gq, gl_error=err,
                max_gap=MAX_GAP)
    n = 0
    for W in (10, 60):
        win = ol.ref_calc_lod(g, f, p, cs, ce, W, ERROR, MAX_GAP, gl=err)
        data[f"win_W{W}"] = win
        n += win.size
    np.savez_compressed(os.path.join(OUT, "tgls_lod.npz"), **data)
    return n


def wlod():
    rng = np.random.default_rng(20260103)
    g, f, p, cs, ce = ol.random_panel(rng, 600, 10, max_gap=MAX_GAP, gaps=2, mono=0.0)
    gpos = np.cumsum(np.diff(p, prepend=0) * 1e-6 * rng.uniform(0.8, 1.2, size=p.shape[0]))
    data = dict(geno=g, freq=f, pos=p, gpos=gpos, centro=np.array([cs, ce], dtype=np.int32),
                max_gap=MAX_GAP, M=7, mu=1e-9, error=ERROR)
    n = 0
    for W in (10, 30):
        hom, ld = ol.ref_hr2_ld(g, W, threads=2)
        data[f"hom_W{W}"] = hom
        data[f"ld_W{W}"] = ld
        ld_safe = np.where(np.isfinite(ld) & (ld > 0), ld, 1.0)
        data[f"ldsafe_W{W}"] = ld_safe
        ref1 = ol.ref_calc_wlod(g, f, p, gpos, ld_safe, cs, ce, W, ERROR, MAX_GAP, 1e-9, 7, threads=1)
        for t in (3, 8):
            assert ol.bits_equal(ref1, ol.ref_calc_wlod(g, f, p, gpos, ld_safe, cs, ce, W, ERROR, MAX_GAP,
                                                        1e-9, 7, threads=t))
        data[f"win_W{W}"] = ref1
        n += ref1.size
    np.savez_compressed(os.path.join(OUT, "wlod.npz"), **data)
    return n


def phased_ld():
    """--phased LD weights (calcR2LD / r2, garlic-data.cpp:426-535, 585-617) of the real reference,
    all individuals and an --ld-subsample index"""
    rng = np.random.default_rng(20260108)
    g, f, p, cs, ce = ol.random_panel(rng, 500, 14, max_gap=MAX_GAP, gaps=0, mono=0.02, miss=0.06)
    fc = rng.integers(0, 2, size=g.shape).astype(np.uint8)
    sub = np.sort(rng.choice(14, size=6, replace=False)).astype(np.int32)
    data = dict(geno=g, freq=f, first_copy=fc, sub=sub)
    n = 0
    for W in (10, 30):
        data[f"ld_W{W}"] = ol.ref_r2_ld(g, fc, f, W, threads=2)
        data[f"ldsub_W{W}"] = ol.ref_r2_ld(g, fc, f, W, idx=sub, threads=3)
        n += 2 * data[f"ld_W{W}"].size
    np.savez_compressed(os.path.join(OUT, "phased_ld.npz"), **data)
    return n


def flatten():
    d = np.load(os.path.join(OUT, "unweighted.npz"))
    win = d["win0_W30"].copy()
    win[3, 100:110] = np.nan  # NaN is dropped as well as MISSING (garlic-data.cpp:2041)
    data = {"win": win}
    for step in (1, 30):
        data[f"flat_step{step}"] = ol.ref_flatten(win, step)
    np.savez_compressed(os.path.join(OUT, "flatten.npz"), **data)
    return win.size


def main():
    assert ol.have_ref(), "build oracle/_ref first: make -C oracle ref"
    os.makedirs(OUT, exist_ok=True)
    manifest = {
        "generator": "tools/make_golden.py",
        "reference": "szpiech/garlic v1.1.6a sources compiled by oracle/Makefile (-O3 -m64 -mmmx -msse -msse2)",
        "values": {
            "lod_known_answers.npz": lod_known_answers(),
            "tgls_conversion.npz": tgls_conversion(),
            "unweighted.npz": unweighted(),
            "tgls_lod.npz": tgls_lod(),
            "wlod.npz": wlod(),
            "flatten.npz": flatten(),
            "phased_ld.npz": phased_ld(),
        },
    }
    with open(os.path.join(OUT, "MANIFEST.json"), "w") as fh:
        json.dump(manifest, fh, indent=1)
    print(json.dumps(manifest["values"]))


if __name__ == "__main__":
    main()
